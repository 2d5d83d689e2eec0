"""Per-kernel MFMA utilisation and L2-side (HBM + Infinity Cache) traffic of one bench.py step from three
rocprofv3 --pmc passes (separate runs, --kernel-trace only):

    for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_step/${c%% *} -o p -- \
          python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-variants --serial-towers
    done
    python scripts/pmc_summary.py gpurun_out/pmc_step profiles/r01/pmc_step_summary_r01.txt

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), both summed over the kernel's
dispatches.  Traffic = 2 x FETCH_SIZE (gfx950 wide-read correction, /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE,
KiB -> bytes, summed over XCDs, divided by the kernel's total duration from the trace of the same pass."""
import collections
import csv
import os
import sys


def load(dirname):
    cc = os.path.join(dirname, "p_counter_collection.csv")
    tr = os.path.join(dirname, "p_kernel_trace.csv")
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        vals[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(float)
    calls = collections.Counter()
    for r in csv.DictReader(open(tr)):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        calls[k] += 1
    return vals, dur, calls


def short(name):
    name = name.replace("void ", "").replace("clipfs::", "")
    return name.split("(")[0][:46]


def main():
    root, out = sys.argv[1], sys.argv[2]
    mf, mdur, calls = load(os.path.join(root, "SQ_VALU_MFMA_BUSY_CYCLES"))
    fe, fdur, _ = load(os.path.join(root, "FETCH_SIZE"))
    wr, wdur, _ = load(os.path.join(root, "WRITE_SIZE"))
    rows = []
    for k in sorted(mdur, key=lambda k: -mdur[k]):
        if mdur[k] < 2e-4:
            continue
        busy, act = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mf[k].get("GRBM_GUI_ACTIVE", 0.0)
        util = busy / (act / 8 * 1024) if act else 0.0
        fb = 2 * fe[k].get("FETCH_SIZE", 0.0) * 1024 / fdur[k] / 1e12 if fdur.get(k) else 0.0
        wb = wr[k].get("WRITE_SIZE", 0.0) * 1024 / wdur[k] / 1e12 if wdur.get(k) else 0.0
        rows.append((k, calls[k], mdur[k] * 1e3, util, fb, wb))
    lines = [__doc__.strip().split("\n\n")[0], "",
             f"{'kernel':46s} {'calls':>6s} {'total ms':>9s} {'MFMA util':>9s} {'read TB/s':>9s} {'write TB/s':>10s}"]
    for k, c, ms, u, fb, wb in rows:
        lines.append(f"{k:46s} {c:6d} {ms:9.2f} {u:9.3f} {fb:9.2f} {wb:10.2f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
