"""A/B of the f16 GEMM kernels per epilogue mode on the cfg-5 shapes: python scripts/f16_tile_ab.py   (child processes
with CLIPFS_F16_TILE = 0 (default: 256 x 256 phased + leftovers), 3 (256 x 128, two workgroups per CU), 1 (128 x 128))."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
SHAPES = [("qkv  C16 only", 3072, 1024, "c16"), ("out  fp32 C+res", 1024, 1024, "c32"), ("fc   C16 + aux", 4096, 1024, "fc"),
          ("proj fp32 C+res", 1024, 4096, "c32"), ("du   C16 act2", 4096, 1024, "du"), ("dh   fp32 C", 1024, 4096, "c32n"),
          ("dx   fp32 C", 1024, 3072, "c32n")]


def child():
    import torch
    from clipfs import ops
    dev = torch.device("cuda:0")
    M = 32896
    wa, wb = torch.randn(8192, 2048, device=dev), torch.randn(4096, 2048, device=dev)
    for _ in range(200):
        ops.gemm_nt(wa, wb)
    out = {}
    for name, N, K, mode in SHAPES:
        a16 = torch.randn(M, K, device=dev).half()
        b = torch.randn(N, K, device=dev) * K ** -0.5
        b16 = ops.to_f16(b)
        bias = torch.randn(N, device=dev)
        o32 = torch.empty(M, N, device=dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        res = torch.randn(M, N, device=dev)
        aux16 = torch.randn(M, N, device=dev).half()

        def run():
            if mode == "c16":
                ops.gemm_nt(None, b, None, bias=bias, b_planes=b16, a16=a16, out16=o16, only16=True)
            elif mode == "c32":
                ops.gemm_nt(None, b, o32, bias=bias, residual=res, b_planes=b16, a16=a16)
            elif mode == "c32n":
                ops.gemm_nt(None, b, o32, b_planes=b16, a16=a16)
            elif mode == "fc":
                ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux16, b_planes=b16, a16=a16, out16=o16, only16=True, aux_f16=True)
            else:
                ops.gemm_nt(None, b, None, act=2, aux_in=aux16, b_planes=b16, a16=a16, out16=o16, only16=True, aux_f16=True)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 20 * 1e3
    print("RESULT " + json.dumps(out))


def main():
    if os.environ.get("AB_CHILD"):
        return child()
    res = {}
    for cfg in ("0", "3", "1"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, AB_CHILD="1", CLIPFS_F16_TILE=cfg),
                           capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(cfg, "FAILED", r.stderr[-300:])
            continue
        res[cfg] = json.loads(line[0][7:])
    print(f"{'shape (M = 32896)':24s}" + "".join(f"   tile cfg {c:>2s}" for c in res))
    for name, N, K, _ in SHAPES:
        fl = 2.0 * 32896 * N * K
        print(f"{name:24s}" + "".join(f" {res[c][name]:7.1f} us {fl / res[c][name] / 1e6:5.0f}" for c in res))


if __name__ == "__main__":
    main()
