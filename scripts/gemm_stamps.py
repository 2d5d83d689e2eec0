"""Where a workgroup of the fp32 GEMM spends its time (diagnostic build: HIPCC_EXTRA=-DCLIPFS_STAMPS python
jittor-clip-fewshot_amd/build.py --tag stamps; run with CLIPFS_LIB_TAG=stamps).  s_memtime (shader-clock ticks; one
counter per XCD, so only stamps of one XCD are comparable) at kernel entry / after the prologue / after the K loop / after
the epilogue of every workgroup, plus the CU it ran on."""
import ctypes
import os
import sys
from collections import Counter

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
shapes = [("img/8 qkv", 1600, 2304, 768), ("img/8 out", 1600, 768, 768), ("img/8 pr", 1600, 768, 3072),
          ("txt/8 qkv", 3927, 1536, 512), ("img qkv", 12800, 2304, 768), ("img out", 12800, 768, 768),
          ("img fc", 12800, 3072, 768), ("img pr", 12800, 768, 3072), ("sq4096", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    c = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    for _ in range(5):
        ops.gemm_nt(a, b, c, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_nt(a, b, c, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1e3
    tile_cfg = int(os.environ.get('CLIPFS_GEMM_TILE', '0'))
    bm, bn = (128, 128) if tile_cfg == 1 else (64, 128)
    S = lib.clipfs_gemm_splits(M, N, K)
    if tile_cfg == 1:
        S = 1
    units = -(-M // bm) * -(-N // bn) * S
    n = min(units, 16384)
    buf = (ctypes.c_ulonglong * (n * 6))()
    assert raw.clipfs_debug_read_stamps(buf, n * 6) == 0
    st = np.array(buf, dtype=np.float64).reshape(n, 6)
    pro, kl, ep = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    rt = st[:, 5] - st[:, 4]  # s_memrealtime (100 MHz) across the K loop
    clock = np.median(kl[rt > 0] / rt[rt > 0]) * 0.1  # GHz: shader cycles per 10 ns tick
    life = st[:, 3] - st[:, 0]
    wall_k = np.median(rt) / 100.0  # us
    print(f"{name:10s} M={M} N={N} K={K} units={units} S={S} event {ev_us:7.1f} us | in-kernel clock {clock:5.3f} GHz | "
          f"prologue {np.median(pro):6.0f} | K loop med {np.median(kl):7.0f} cyc = {wall_k:6.1f} us, min {kl.min():7.0f} "
          f"({np.median(kl) / (K / 32):5.0f}/K-step; MFMA bound {2 * (bm // 64) * 2048}) | epilogue med {np.median(ep[ep > 0]):6.0f} | life med {np.median(life):7.0f}")
