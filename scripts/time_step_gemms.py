"""Stand-alone timing of the tower GEMM shapes of one workload (default cfg-2: 256 images, 403 captions), one line per
shape: 64 x 128 tiles, rounds over the 512 workgroup slots, us, TFLOP/s, fraction of the 157.3 TFLOP/s fp32 MFMA peak.

    python scripts/time_step_gemms.py [images captions]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))


def main():
    import torch
    from clipfs import ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 403
    dev = torch.device("cuda:0")
    tot_t = tot_f = 0.0
    # the first seconds of a process run at a lower clock (the first shape measured read 10 % slow): warm up on a big product
    wa, wb = torch.randn(8192, 2048, device=dev), torch.randn(4096, 2048, device=dev)
    for _ in range(300):
        ops.gemm_nt(wa, wb)
    torch.cuda.synchronize()
    for tag, M, d, blocks in (("img", B * 50, 768, 12), ("txt", C * 77, 512, 12)):
        for name, N, K, w in (("qkv", 3 * d, d, 1), ("out", d, d, 2), ("fc", 4 * d, d, 2), ("pr", d, 4 * d, 2), ("dx", d, 3 * d, 1)):
            a = torch.randn(M, K, device=dev)
            b = torch.randn(N, K, device=dev)
            bias = torch.randn(N, device=dev)
            c = torch.empty(M, N, device=dev)
            for _ in range(3):
                ops.gemm_nt(a, b, c, bias=bias)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(a, b, c, bias=bias)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            fl = 2.0 * M * N * K
            tiles = ((M + 63) // 64) * ((N + 127) // 128)
            tot_t += us * w * blocks
            tot_f += fl * w * blocks
            print(f"{tag} {name:4s} M={M:6d} N={N:5d} K={K:5d} tiles {tiles:6d} rounds {tiles / 512:6.2f} {us:8.1f} us "
                  f"{fl / us / 1e6:6.1f} TF/s  {fl / us / 1e6 / 157.3:5.3f}   x{w * blocks} per step", flush=True)
    print(f"weighted: {tot_t / 1e3:.2f} ms per step in these shapes, {tot_f / tot_t / 1e6:.1f} TF/s = {tot_f / tot_t / 1e6 / 157.3:.3f}")


if __name__ == "__main__":
    main()
