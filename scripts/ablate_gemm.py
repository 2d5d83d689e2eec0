"""Timing of the stream-K GEMM under tuning knobs (CLIPFS_GEMM_SK_TILE, CLIPFS_GEMM_SK_PER_CU, CLIPFS_GEMM_SK, and the
timing-only CLIPFS_GEMM_ABLATE bits: 1 no LDS-DMA, 4 no barrier, 8 no LDS fragment reads, 16 no epilogue -- WRONG results)."""
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
tag = " ".join(f"{k[12:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CLIPFS_GEMM_"))
shapes = []
for t, M, d in (("img", 12800, 768), ("txt", 31031, 512)):
    shapes += [(f"{t} qkv", M, 3 * d, d), (f"{t} out", M, d, d), (f"{t} fc", M, 4 * d, d), (f"{t} pr", M, d, 4 * d), (f"{t} dqkv", M, d, 3 * d)]
shapes += [("sq4096", 4096, 4096, 4096), ("img/8 fc", 1600, 3072, 768), ("txt/8 out", 3927, 512, 512)]
tt = ff = 0
line = []
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    want = a[:64].double() @ b.double().t()
    ops.gemm_nt(a, b, out)
    err = (out[:64].double() - want).abs().max().item() / want.abs().max().item()
    t = timeit(lambda: ops.gemm_nt(a, b, out))
    if name[:3] in ("img", "txt") and "/" not in name:
        tt += t; ff += 2.0 * M * N * K
    line.append(f"{name}:{2*M*N*K/t/157.3e12*100:5.1f}%" + ("" if err < 1e-5 else f"(ERR {err:.1e})"))
print(f"[{tag}] " + "  ".join(line) + f"  | cfg-2 mix {ff/tt/157.3e12*100:5.1f}%", flush=True)
