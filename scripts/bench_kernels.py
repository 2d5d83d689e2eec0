"""Micro-benchmarks of the hot kernels at the path's shapes (cfg-2: B=256, L=50, d=768; text 403x77, d=512).
Prints achieved TFLOP/s (GEMM, vs 157.3 fp32-MFMA peak) or GB/s (memory-bound kernels, vs ~8 TB/s)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    shapes = [("qkv", 12800, 2304, 768), ("out", 12800, 768, 768), ("fc", 12800, 3072, 768), ("proj", 12800, 768, 3072),
              ("t_qkv", 31031, 1536, 512), ("t_out", 31031, 512, 512), ("t_fc", 31031, 2048, 512),
              ("t_proj", 31031, 512, 2048), ("sq4096", 4096, 4096, 4096)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, b, out))
        print(f"gemm {name:7s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TFLOP/s  ({2*M*N*K/t/157.3e12*100:.1f}% of fp32 MFMA peak)")
    x = torch.randn(12800, 768, device=dev)
    g = torch.ones(768, device=dev)
    b = torch.zeros(768, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b))
    print(f"layernorm 12800x768: {t*1e6:.1f} us  {2*x.numel()*4/t/1e9:.0f} GB/s")
    qkv = torch.randn(12800, 2304, device=dev)
    t = timeit(lambda: ops.attention_fwd(qkv, 256, 50, 12, False))
    print(f"attention fwd B=256 L=50 H=12: {t*1e6:.1f} us")
    do = torch.randn(12800, 768, device=dev)
    o_, l_ = ops.attention_fwd(qkv, 256, 50, 12, False, want_lse=True)
    t = timeit(lambda: ops.attention_bwd(qkv, do, 256, 50, 12, False, out=o_, lse=l_))
    print(f"attention bwd B=256 L=50 H=12: {t*1e6:.1f} us")
    qkv = torch.randn(31031, 1536, device=dev)
    t = timeit(lambda: ops.attention_fwd(qkv, 403, 77, 8, True))
    print(f"attention fwd B=403 L=77 H=8 causal: {t*1e6:.1f} us")
    do = torch.randn(31031, 512, device=dev)
    o_, l_ = ops.attention_fwd(qkv, 403, 77, 8, True, want_lse=True)
    t = timeit(lambda: ops.attention_bwd(qkv, do, 403, 77, 8, True, out=o_, lse=l_))
    print(f"attention bwd B=403 L=77 H=8 causal: {t*1e6:.1f} us")
    A = torch.randn(12, 768, device=dev)
    t = timeit(lambda: ops.lora_down(x, A, 4, 3))
    print(f"lora_down 12800x768 r=4x3: {t*1e6:.1f} us")
    t = timeit(lambda: ops.lora_down(x, A, 4, 3, p=0.25, seed=7))
    print(f"lora_down +dropout: {t*1e6:.1f} us")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def lora_bench():
    for M, d, r in ((12800, 768, 4), (31031, 512, 4), (32896, 1024, 16), (31031, 768, 16)):
        x = torch.randn(M, d, device=dev)
        dy = torch.randn(M, 3 * d, device=dev)
        A = torch.randn(3 * r, d, device=dev)
        B = torch.randn(3 * d, r, device=dev)
        t = ops.lora_down(x, A, r, 3)
        dA = torch.zeros_like(A)
        dB = torch.zeros_like(B)
        dx = torch.zeros_like(x)
        for p in (0.0, 0.25):
            td = timeit(lambda: ops.lora_down(x, A, r, 3, p=p, seed=5 if p else 0))
            tm = timeit(lambda: ops.lora_bwd(dy, x, t, A, B, dA, dB, dx=dx, scale=0.5, p=p, seed=5 if p else 0))
            print(f"lora M={M} d={d} r={r} p={p}: down {td*1e6:7.1f} us  bwd {tm*1e6:7.1f} us", flush=True)


if __name__ == "__main__" and "--lora" in sys.argv:
    lora_bench()


def bf16_bench():
    shapes = [("qkv", 12800, 2304, 768), ("out", 12800, 768, 768), ("fc", 12800, 3072, 768), ("proj", 12800, 768, 3072),
              ("t_qkv", 31031, 1536, 512), ("t_out", 31031, 512, 512), ("t_fc", 31031, 2048, 512),
              ("t_proj", 31031, 512, 2048), ("sq4096", 4096, 4096, 4096)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        planes = ops.split_bf16(b)
        t0 = timeit(lambda: ops.gemm_nt(a, b, out))
        t1 = timeit(lambda: ops.gemm_nt(a, b, out, b_planes=planes))
        print(f"gemm {name:7s} fp32 {t0*1e6:8.1f} us {2*M*N*K/t0/1e12:6.1f} TF | bf16x3 {t1*1e6:8.1f} us {2*M*N*K/t1/1e12:6.1f} TF  x{t0/t1:.2f}")


if __name__ == "__main__" and "--bf16" in sys.argv:
    bf16_bench()


def f16_bench():
    """cfg-5 (ViT-L/14, 128 images x 257 tokens) GEMM shapes: fp32-A/f16-B kernel vs the f16 x f16 kernel."""
    shapes = [("qkv", 32896, 3072, 1024), ("out", 32896, 1024, 1024), ("fc", 32896, 4096, 1024),
              ("proj", 32896, 1024, 4096), ("dqkv", 32896, 1024, 3072), ("sq8192", 8192, 8192, 8192)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev) * K ** -0.5
        out = torch.empty(M, N, device=dev)
        out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        a16, b16 = a.half(), ops.to_f16(b)
        t0 = timeit(lambda: ops.gemm_nt(a, b, out, b_planes=b16))
        t1 = timeit(lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16))
        t2 = timeit(lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16, out16=out16))
        t3 = timeit(lambda: ops.gemm_nt(None, b, None, b_planes=b16, a16=a16, out16=out16, only16=True))
        fl = 2 * M * N * K
        print(f"gemm {name:7s} f32A {t0*1e6:8.1f} us {fl/t0/1e12:7.1f} TF | f16A {t1*1e6:8.1f} us {fl/t1/1e12:7.1f} TF "
              f"| +C16 {t2*1e6:8.1f} us {fl/t2/1e12:7.1f} TF | C16 only {t3*1e6:8.1f} us {fl/t3/1e12:7.1f} TF", flush=True)


if __name__ == "__main__" and "--f16" in sys.argv:
    f16_bench()


def attn16_bench():
    """cfg-5 attention (128 images x 16 heads x 257 tokens): fp32 LDS-resident kernels vs the f16 MFMA kernels."""
    B, H, L = 128, 16, 257
    qkv = torch.randn(B * L, 3 * H * 64, device=dev)
    dout = torch.randn(B * L, H * 64, device=dev)
    out, lse = ops.attention_fwd(qkv, B, L, H, False, want_lse=True)
    o16, l16 = ops.attention_f16_fwd(qkv, B, L, H)
    fl = 4 * L * L * 64 * B * H
    t0 = timeit(lambda: ops.attention_fwd(qkv, B, L, H, False, want_lse=True))
    t1 = timeit(lambda: ops.attention_f16_fwd(qkv, B, L, H))
    print(f"attn fwd  fp32 {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | f16 mfma {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF", flush=True)
    t0 = timeit(lambda: ops.attention_bwd(qkv, dout, B, L, H, False, out=out, lse=lse))
    t1 = timeit(lambda: ops.attention_f16_bwd(qkv, dout, o16, l16, B, L, H))
    print(f"attn bwd  fp32 {t0*1e6:8.1f} us {2.5*fl/t0/1e12:6.1f} TF | f16 mfma {t1*1e6:8.1f} us {2.5*fl/t1/1e12:6.1f} TF", flush=True)


if __name__ == "__main__" and "--attn16" in sys.argv:
    attn16_bench()
