"""Per-kernel SQ counters of any command's kernels from one rocprofv3 --pmc pass (csv in DIR):
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -o p -- python3 <script>
    python scripts/pmc_kernels.py DIR [substring]
Prints, per kernel name (summed over dispatches): calls, total us, and every counter divided by SQ_WAVE_CYCLES when
that counter is present (wave-cycle shares: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stalls,
ACTIVE_INST_* = issuing)."""
import collections, csv, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cc = [f for f in os.listdir(d) if f.endswith("counter_collection.csv")][0]
tr = [f for f in os.listdir(d) if f.endswith("kernel_trace.csv")][0]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(d, cc))):
    vals[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(os.path.join(d, tr))):
    k = r["Kernel_Name"].split("(")[0][-60:]
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; calls[k] += 1
for k in sorted(dur, key=lambda k: -dur[k]):
    if sub not in k or dur[k] < 50:
        continue
    v = vals[k]; wc = v.get("SQ_WAVE_CYCLES", 0.0)
    parts = []
    for name in sorted(v):
        if name == "SQ_WAVE_CYCLES":
            continue
        parts.append(f"{name.replace('SQ_', '')} {v[name] / wc:.3f}" if wc else f"{name.replace('SQ_', '')} {v[name]:.3g}")
    print(f"{k:60s} calls {calls[k]:4d} {dur[k]:9.0f} us | " + " | ".join(parts))
