"""fp32 GEMM at the path's shapes (forward + dgrad shapes of cfg-2, and the 8-GPU per-rank shapes): time and
correctness (vs an fp64 matmul) of whatever schedule the library picks.  Run twice to compare:
    python scripts/bench_gemm_sk.py            # stream-K (default)
    CLIPFS_GEMM_SK=0 python scripts/bench_gemm_sk.py   # one tile per workgroup"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from clipfs import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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
    shapes = []
    for tag, M, d in (("img", 12800, 768), ("txt", 31031, 512), ("img/8", 1600, 768), ("txt/8", 51 * 77, 512)):
        shapes += [(f"{tag} qkv", M, 3 * d, d), (f"{tag} out", M, d, d), (f"{tag} fc", M, 4 * d, d),
                   (f"{tag} pr", M, d, 4 * d), (f"{tag} dqkv", M, d, 3 * d)]
    shapes += [("sq4096", 4096, 4096, 4096), ("feat", 256, 512, 768), ("ragged", 1000, 200, 96)]
    tot_t = tot_f = 0.0
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        ops.gemm_nt(a, b, out, bias=bias, residual=res)
        out2 = torch.empty(M, N, device=dev)
        ops.gemm_nt(a, b, out2, bias=bias, residual=res)
        want = (a.double() @ b.double().t() + bias.double() + res.double())
        err = (out.double() - want).abs().max().item() / want.abs().max().item()
        same = torch.equal(out, out2)
        t = timeit(lambda: ops.gemm_nt(a, b, out, bias=bias, residual=res))
        fl = 2.0 * M * N * K
        if name.startswith(("img ", "txt ")):
            tot_t += t
            tot_f += fl
        print(f"gemm {name:10s} M={M:6d} N={N:5d} K={K:5d}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF ({fl/t/157.3e12*100:5.1f}%)  "
              f"rel err {err:.1e} reproducible {same}", flush=True)
    print(f"cfg-2 tower shapes together: {tot_f/tot_t/1e12:.1f} TF ({tot_f/tot_t/157.3e12*100:.1f}%)")


if __name__ == "__main__":
    main()
