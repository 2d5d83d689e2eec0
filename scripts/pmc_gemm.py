"""One launch of each path GEMM shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for name, M, N, K in [("qkv", 12800, 2304, 768), ("out", 12800, 768, 768), ("fc", 12800, 3072, 768), ("proj", 12800, 768, 3072), ("t_qkv", 31031, 1536, 512)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm_nt(a, b, out)
    torch.cuda.synchronize()
