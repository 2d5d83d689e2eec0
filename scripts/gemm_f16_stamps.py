"""Phase times of the 256 x 256 phased f16 GEMM (diagnostic build: HIPCC_EXTRA=-DCLIPFS_STAMPS build.py --tag stamps;
CLIPFS_LIB_TAG=stamps): s_memtime at entry / after the prologue / after the K loop / after the epilogue per workgroup,
in-kernel clock from s_memrealtime across the K loop."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
M = int(os.environ.get("STAMPS_M", "32768"))  # STAMPS_M=2048: 96 qkv tiles, i.e. fewer writers at a time than CUs
for name, N, K, mode in [("qkv  C16 only", 3072, 1024, "c16"), ("out  fp32 C", 1024, 1024, "c32"), ("fc   C16 + aux", 4096, 1024, "fc"),
                         ("proj fp32 C", 1024, 4096, "c32"), ("du   C16 act2", 4096, 1024, "du"), ("sq8192 fp32", 8192, 8192, "sq")]:
    Mi = 8192 if mode == "sq" else M
    a16 = (torch.randn(Mi, K, device=dev)).half()
    b = torch.randn(N, K, device=dev) * K ** -0.5
    b16 = ops.to_f16(b)
    bias = torch.randn(N, device=dev)
    out = torch.empty(Mi, N, device=dev)
    out16 = torch.empty(Mi, N, device=dev, dtype=torch.float16)
    res = torch.randn(Mi, N, device=dev)
    aux16 = torch.randn(Mi, N, device=dev).half()

    def run():
        if mode == "c16":
            ops.gemm_nt(None, b, None, bias=bias, b_planes=b16, a16=a16, out16=out16, only16=True)
        elif mode in ("c32", "sq"):
            ops.gemm_nt(None, b, out, bias=bias, residual=res, b_planes=b16, a16=a16)
        elif mode == "fc":
            ops.gemm_nt(None, b, None, bias=bias, act=1, aux_out=aux16, b_planes=b16, a16=a16, out16=out16, only16=True, aux_f16=True)
        else:
            ops.gemm_nt(None, b, None, act=2, aux_in=aux16, b_planes=b16, a16=a16, out16=out16, only16=True, aux_f16=True)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) * 1e3
    n = (Mi // 256) * (N // 256)
    buf = (ctypes.c_ulonglong * (n * 10))()
    assert raw.clipfs_debug_read_f16_stamps(buf, n * 10) == 0
    st = np.array(buf, dtype=np.float64).reshape(n, 10)
    pro, kl, ep = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    rt = st[:, 5] - st[:, 4]
    clock = np.median(kl[rt > 0] / rt[rt > 0]) * 0.1
    ideal = K / 64 * 4 * 16 * 16  # 16x16x32: 16 cycles per MFMA on one SIMD, 2 waves per SIMD -> 2 x 64 MFMAs x 16 cyc / 2 ... per wave: 64 MFMAs per K-tile
    print(f"{name:16s} M={Mi} N={N} K={K} tiles={n} event {ev:7.1f} us {2.0 * Mi * N * K / ev / 1e6:7.1f} TF | clock {clock:5.3f} GHz | "
          f"prologue {np.median(pro):6.0f} | K loop {np.median(kl):7.0f} cyc ({np.median(kl) / (K / 64):5.0f}/K-tile; MFMA bound 2048) | "
          f"epilogue {np.median(ep):6.0f} cyc = {np.median(ep) / np.median(st[:, 3] - st[:, 0]) * 100:4.1f}% of the tile "
          f"[stage0 {np.median(st[:, 6] - st[:, 2]):5.0f} | rows0 {np.median(st[:, 7] - st[:, 6]):5.0f} | stage1 {np.median(st[:, 8] - st[:, 7]):5.0f} | rows1 {np.median(st[:, 3] - st[:, 8]):5.0f}]")
