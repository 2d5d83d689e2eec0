"""f16 x f16 GEMM epilogue variants at the cfg-5 c_fc shape (M = 32896, N = 4096, K = 1024)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
M, N, K = 32896, 4096, 1024
a16 = torch.randn(M, K, device=dev).half(); b = torch.randn(N, K, device=dev) * K ** -0.5; b16 = ops.to_f16(b)
out = torch.empty(M, N, device=dev); out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
aux = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
for _ in range(10): ops.gemm_nt(None, b, out, b_planes=b16, a16=a16)
r = {}
r["C32"] = timeit(lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16))
r["C16 only"] = timeit(lambda: ops.gemm_nt(None, b, None, b_planes=b16, a16=a16, out16=out16, only16=True))
r["C16 + bias"] = timeit(lambda: ops.gemm_nt(None, b, None, bias=bias, b_planes=b16, a16=a16, out16=out16, only16=True))
r["C16 + bias + gelu"] = timeit(lambda: ops.gemm_nt(None, b, None, bias=bias, act=1, b_planes=b16, a16=a16, out16=out16, only16=True))
r["C16 + bias + gelu + aux32 (c_fc fwd)"] = timeit(lambda: ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux, b_planes=b16, a16=a16, out16=out16, only16=True))
r["C16 x gelu'(aux32) (c_proj dgrad)"] = timeit(lambda: ops.gemm_nt(None, b, None, act=2, aux_in=aux, b_planes=b16, a16=a16, out16=out16, only16=True))
for k, v in r.items():
    print(f"{k:40s} {v*1e6:8.1f} us  {2*M*N*K/v/1e12:6.1f} TF", flush=True)
