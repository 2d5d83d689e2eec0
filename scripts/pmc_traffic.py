"""Post-process two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) over one
bench.py step into the per-launch HBM traffic of the dominant GEMM kernel (profiles/<round>/gemm_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- \
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-variants --serial-towers
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_write ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv \
        gpurun_out/pmc_write/p_counter_collection.csv profiles/r01/gemm_traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB-units of
the L2's memory-side request counters, summed over the 8 XCDs; on gfx950 FETCH_SIZE reports HALF the bytes of wide
(16 B/lane) streaming reads, which is what both GEMM operands are (global_load_lds_dwordx4) -> doubled.  WRITE_SIZE is
exact for 16 B/lane stores; the GEMM epilogue stores 4 B/lane (128 B contiguous per 32 lanes): "uncalibrated" in the
guide, reported as counted.  Infinity-Cache hits are included in both (the counters sit on the fabric side of L2).
"""
import collections
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stamps():
    """(gemm source stamp of the tree = what the library built from it reports, git HEAD or None).  bench.py reports
    ``roofline.traffic`` only when the stamp equals clipfs_gemm_source_stamp() of the library it loaded."""
    spec = importlib.util.spec_from_file_location("clipfs_build", os.path.join(ROOT, "jittor-clip-fewshot_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        head = None
    return mod.source_stamps()[1], head


def per_launch(path, kernel_substr, counter):
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = list(tot.values())
    return sum(vals) / max(len(vals), 1), len(vals)


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    kern = sys.argv[4] if len(sys.argv) > 4 else "gemm_nt_kernel<64, 128, 3>"
    f_kib, nf = per_launch(fetch_csv, kern, "FETCH_SIZE")
    w_kib, nw = per_launch(write_csv, kern, "WRITE_SIZE")
    gemm_stamp, head = stamps()
    res = {"kernel": kern, "launches_counted": [nf, nw], "gemm_source_stamp": gemm_stamp, "git_head_when_processed": head,
           "fetch_bytes_per_launch": round(2 * f_kib * 1024), "write_bytes_per_launch": round(w_kib * 1024),
           "traffic_bytes_per_launch": round(2 * f_kib * 1024 + w_kib * 1024),
           "corrections": "FETCH_SIZE x2 (gfx950 wide-read undercount), KiB -> bytes, summed over XCDs; includes Infinity-Cache hits"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
