"""fp32 GEMM time against K at the image tower's M, N: slope = K-loop rate, intercept = per-tile cost outside the loop."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tag = " ".join(f"{k[12:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CLIPFS_GEMM_"))
for M, N in ((12800, 3072), (31031, 2048), (12288, 4096)):
    out = torch.empty(M, N, device=dev)
    res = []
    for K in (128, 256, 512, 768, 1024, 2048, 4096):
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, b, out))
        res.append((K, t))
    (k1, t1), (k2, t2) = res[-2], res[-1]
    slope = (t2 - t1) / (k2 - k1)              # us per unit K
    icpt = res[0][1] - slope * res[0][0]
    tiles = ((M + 63) // 64) * ((N + 127) // 128)
    print(f"[{tag}] {M}x{N}: " + "  ".join(f"K={k}:{t:7.1f}" for k, t in res) +
          f" | slope {2*M*N/slope/1e6/157.3e6*100:5.1f}% of peak, intercept {icpt:6.1f} us = {icpt/ (tiles/512):5.2f} us per round of 512 tiles", flush=True)
