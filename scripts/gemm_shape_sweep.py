"""fp32 GEMM time over N / M / K around the cfg-2 shapes (is a shape an outlier, or was it measured cold?)."""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "jittor-clip-fewshot_amd"))
import torch
from clipfs import ops
dev = torch.device("cuda:0")
def t(M, N, K, bias=True):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); bi = torch.randn(N, device=dev) if bias else None
    c = torch.empty(M, N, device=dev)
    for _ in range(3): ops.gemm_nt(a, b, c, bias=bi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(a, b, c, bias=bi)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tiles = ((M + 63) // 64) * ((N + 127) // 128)
    print(f"M={M:6d} N={N:5d} K={K:5d} tiles {tiles:5d} rounds {tiles/512:6.2f}  {us:8.1f} us  {2.0*M*N*K/us/1e6/157.3:5.3f}", flush=True)
wa, wb = torch.randn(8192, 2048, device=dev), torch.randn(4096, 2048, device=dev)
for _ in range(300):
    ops.gemm_nt(wa, wb)  # clocks up before the first measurement
torch.cuda.synchronize()
for N in (1536, 2048, 2304, 2560, 3072, 3584, 4096):
    t(12800, N, 768)
for M in (12288, 12800, 13312, 16384):
    t(M, 2304, 768)
for K in (512, 768, 1024, 1536):
    t(12800, 768, K)
