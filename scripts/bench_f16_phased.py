"""f16 x f16 GEMM: the 4-phase-per-K-tile 256 x 256 kernel against the 2-phase ping-pong kernel (CLIPFS_F16_PHASED=0 in a
second process) on square and cfg-5 shapes, with a full check of the result against torch on the same f16 operands."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
tag = "phased=" + os.environ.get("CLIPFS_F16_PHASED", "1")
shapes = [("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
          ("img qkv", 32768, 3072, 1024), ("img out", 32768, 1024, 1024), ("img fc", 32768, 4096, 1024), ("img pr", 32768, 1024, 4096),
          ("txt qkv", 28928, 2304, 768), ("txt out", 30976, 768, 768), ("txt fc", 30976, 3072, 768), ("txt pr", 30976, 768, 3072),
          ("k128", 8192, 4096, 128), ("k192", 8192, 4096, 192)]
check = os.environ.get("CHECK", "1") == "1"
for name, M, N, K in shapes:
    g = torch.Generator(device=dev); g.manual_seed(M + N + K)
    a16 = torch.randn(M, K, device=dev, generator=g).half()
    b = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b16 = ops.to_f16(b)
    out = torch.empty(M, N, device=dev)
    ops.gemm_nt(None, b, out, b_planes=b16, a16=a16)
    err = float("nan")
    if check:
        err = 0.0
        for r0 in range(0, M, 4096):
            want = a16[r0:r0 + 4096].float() @ b16.view(torch.float16).reshape(N, K).float().t()
            err = max(err, (out[r0:r0 + 4096] - want).abs().max().item() / want.abs().max().item())
    t = timeit(lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16))
    # epilogue variant: bias + QuickGELU + saved pre-activation, f16 result only (c_fc forward)
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16); aux = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
    ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux, b_planes=b16, a16=a16, out16=out16, only16=True)
    e2 = float("nan")
    if check:
        e2 = 0.0
        for r0 in range(0, M, 4096):
            pre = a16[r0:r0 + 4096].float() @ b16.view(torch.float16).reshape(N, K).float().t() + bias
            e2 = max(e2, (aux[r0:r0 + 4096] - pre).abs().max().item() / pre.abs().max().item())
            y = pre * torch.sigmoid(1.702 * pre)
            e2 = max(e2, (out16[r0:r0 + 4096].float() - y).abs().max().item() / y.abs().max().item())
    t2 = timeit(lambda: ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux, b_planes=b16, a16=a16, out16=out16, only16=True))
    print(f"[{tag}] {name:8s} {M}x{N}x{K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF  err {err:.1e} | fc-epilogue {t2*1e6:8.1f} us {2*M*N*K/t2/1e12:7.1f} TF err {e2:.1e}", flush=True)
    del a16, b, b16, out, out16, aux
