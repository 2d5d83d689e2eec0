"""Tile height x split-K sweep of the fp32 GEMM on the few-tile shapes of the path (the 8-GPU per-rank shapes of cfg-3
and the one-row-per-sequence products of the last block).  Every configuration runs in a child process (the knobs
CLIPFS_GEMM_BM / CLIPFS_GEMM_SPLITS are read once per process); the parent prints one table, best time per shape.

    python scripts/sweep_small_gemm.py            # parent
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))

SHAPES = []
for tag, M, d in (("img/8", 1600, 768), ("txt/8", 51 * 77, 512), ("img/4", 3200, 768), ("txt/4", 101 * 77, 512),
                  ("img/2", 6400, 768), ("txt/2", 202 * 77, 512), ("img-rows", 256, 768), ("txt-rows", 403, 512),
                  ("img-rows/8", 32, 768), ("txt-rows/8", 51, 512)):
    SHAPES += [(f"{tag} qkv", M, 3 * d, d), (f"{tag} out", M, d, d), (f"{tag} fc", M, 4 * d, d), (f"{tag} pr", M, d, 4 * d),
               (f"{tag} dqkv", M, d, 3 * d)]


def child():
    import torch
    from clipfs import ops
    dev = torch.device("cuda:0")
    out = {}
    for name, M, N, K in SHAPES:
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
        out[name] = e0.elapsed_time(e1) / 20 * 1e3
    print("RESULT " + json.dumps(out))


def main():
    if os.environ.get("SWEEP_CHILD"):
        return child()
    configs = [("auto", {})] + [(f"bm{bm} s{s}", {"CLIPFS_GEMM_BM": str(bm), "CLIPFS_GEMM_SPLITS": str(s)})
                                for bm in (64, 32) for s in (1, 2, 3, 4, 6, 8)]
    res = {}
    for name, env in configs:
        e = dict(os.environ, SWEEP_CHILD="1", **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(f"{name}: FAILED\n{r.stderr[-400:]}")
            continue
        res[name] = json.loads(line[0][7:])
        print(f"[sweep] {name} done", file=sys.stderr, flush=True)
    names = [n for n, _ in configs if n in res]
    print("shape".ljust(18) + "M".rjust(6) + "N".rjust(6) + "K".rjust(6) + "".join(n.rjust(9) for n in names) + "   best")
    for sname, M, N, K in SHAPES:
        row = [res[n][sname] for n in names]
        best = min(range(1, len(row)), key=lambda i: row[i])
        fl = 2.0 * M * N * K
        print(sname.ljust(18) + f"{M:6d}{N:6d}{K:6d}" + "".join(f"{t:9.1f}" for t in row) +
              f"   {names[best]} ({fl / row[best] / 1e6 / 157.3 * 100:.0f}% ; auto {fl / row[0] / 1e6 / 157.3 * 100:.0f}%)")


if __name__ == "__main__":
    main()
