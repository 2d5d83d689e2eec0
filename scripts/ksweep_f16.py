import os, sys, torch
sys.path.insert(0, "/root/repo/jittor-clip-fewshot_amd")
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
tag = os.environ.get("CLIPFS_F16_PHASED", "1")
M, N = 32768, 4096
out = torch.empty(M, N, device=dev)
res = []
for K in (128, 256, 512, 1024, 2048, 4096):
    a16 = torch.randn(M, K, device=dev).half(); b = torch.randn(N, K, device=dev) * K ** -0.5; b16 = ops.to_f16(b)
    t = timeit(lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16))
    res.append((K, t))
print(f"[phased={tag}] " + "  ".join(f"K={k}: {t:7.1f}us" for k, t in res), flush=True)
