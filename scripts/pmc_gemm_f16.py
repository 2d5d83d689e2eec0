"""One launch set of the f16 x f16 GEMM (for rocprofv3 --pmc passes): sq8192 and two cfg-5 shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for name, M, N, K in [("sq8192", 8192, 8192, 8192), ("qkv", 32768, 3072, 1024), ("proj", 32768, 1024, 4096)]:
    a = torch.randn(M, K, device=dev).half(); b = torch.randn(N, K, device=dev)
    b16 = ops.to_f16(b); out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm_nt(None, b, out, b_planes=b16, a16=a)
    torch.cuda.synchronize()
