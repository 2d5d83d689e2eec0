"""fp32 GEMM under rocprofv3 --pmc: a few back-to-back launches of two shapes (what do the waves wait for?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for M, N, K in ((4096, 4096, 4096), (12800, 3072, 768), (31031, 1536, 512)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(6):
        ops.gemm_nt(a, b, out)
torch.cuda.synchronize()
