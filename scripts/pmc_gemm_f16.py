"""One launch set of the f16 x f16 GEMM (for rocprofv3 --pmc passes): sq8192 and cfg-5 shapes, each in the epilogue
mode the tower uses (qkv / fc: f16-only result; out / proj: fp32 result + residual) and, for the qkv shape, also with an
fp32 result (what profiles/r02/pmc_gemm_f16_r02.txt measured)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for name, M, N, K, mode in [("sq8192 fp32 C", 8192, 8192, 8192, "c32"), ("qkv fp32 C (r02 mode)", 32768, 3072, 1024, "c32"),
                            ("qkv f16 C + bias (tower mode)", 32768, 3072, 1024, "c16"), ("out fp32 C + residual", 32768, 1024, 1024, "res"),
                            ("fc f16 C + f16 pre-activation", 32768, 4096, 1024, "fc"), ("proj fp32 C + residual", 32768, 1024, 4096, "res")]:
    a = torch.randn(M, K, device=dev).half(); b = torch.randn(N, K, device=dev) * K ** -0.5
    b16 = ops.to_f16(b); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev) if mode in ("c32", "res") else None
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16) if mode in ("c16", "fc") else None
    res = torch.randn(M, N, device=dev) if mode == "res" else None
    aux = torch.empty(M, N, device=dev, dtype=torch.float16) if mode == "fc" else None
    for _ in range(3):
        if mode == "c32":
            ops.gemm_nt(None, b, out, b_planes=b16, a16=a)
        elif mode == "res":
            ops.gemm_nt(None, b, out, bias=bias, residual=res, b_planes=b16, a16=a)
        elif mode == "c16":
            ops.gemm_nt(None, b, None, bias=bias, b_planes=b16, a16=a, out16=out16, only16=True)
        else:
            ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux, aux_f16=True, b_planes=b16, a16=a, out16=out16, only16=True)
    torch.cuda.synchronize()
    print(name, M, N, K, flush=True)
