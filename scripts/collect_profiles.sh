#!/bin/bash
# Collects the round's profile evidence on the GPU box (one gpurun call): kernel-trace stats of the three workloads,
# the three --pmc passes over the cfg-2 step (post-processed here: the raw counter CSVs are too large to merge back)
# and the MFMA-busy pass over the f16 GEMM.  Usage:  bash scripts/collect_profiles.sh r03
set -o pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-roofline --no-variants --no-extras --serial-towers"
trace() {  # name, bench args
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o p -- python3 $ROOT/bench.py "$@" > $OUT/tr_$name.log 2>&1 || echo "trace $name failed"
  cp $OUT/tr_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats_$R.csv 2>/dev/null || find $OUT/tr_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats_$R.csv \;
  rm -rf $OUT/tr_$name
  echo "trace $name done"
}
trace bench_step_serial_towers --steps 3 --warmup 1 $B
trace per_rank_32x51_serial_towers --batch 32 --classes 51 --steps 10 --warmup 2 $B
trace l14_fp16_serial_towers --model l14 --batch 128 --precision fp16 --steps 3 --warmup 1 $B
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  d=$OUT/pmc_step/${c%% *}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 $B > $OUT/pmc_${c%% *}.log 2>&1 || echo "pmc $c failed"
  for f in $(find $d -name "p_*.csv"); do mv $f $d/ 2>/dev/null; done
  echo "pmc ${c%% *} done"
done
python3 $ROOT/scripts/pmc_summary.py $OUT/pmc_step $OUT/pmc_step_summary_$R.txt > /dev/null && echo "summary ok"
python3 $ROOT/scripts/pmc_traffic.py $OUT/pmc_step/FETCH_SIZE/p_counter_collection.csv $OUT/pmc_step/WRITE_SIZE/p_counter_collection.csv $OUT/gemm_traffic.json && echo "traffic ok"
rm -rf $OUT/pmc_step
d=$OUT/pmc_f16
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d -o p -- python3 $ROOT/scripts/pmc_gemm_f16.py > $OUT/pmc_f16.log 2>&1 || echo "pmc f16 failed"
for f in $(find $d -name "p_*.csv"); do mv $f $d/ 2>/dev/null; done
python3 - <<PY
import csv, collections
cc = "$d/p_counter_collection.csv"; tr = "$d/p_kernel_trace.csv"
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[int(r["Dispatch_Id"])].setdefault(r["Counter_Name"], 0.0)
    vals[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
names = {}
for r in csv.DictReader(open(cc)):
    names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(tr)):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
labels = [l.strip() for l in open("$OUT/pmc_f16.log") if l[:2] in ("sq", "qk", "ou", "fc", "pr")]
out = ["f16 GEMM (gemm_f16_ph_kernel<true>), rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES over scripts/pmc_gemm_f16.py",
       "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration", ""]
k = 0
for d_id in sorted(vals):
    if "gemm_f16_ph_kernel" not in names[d_id]:
        continue
    v = vals[d_id]; act = v.get("GRBM_GUI_ACTIVE", 0.0)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act / 8 * 1024) if act else 0.0
    lab = labels[k // 3] if k // 3 < len(labels) else "?"
    out.append(f"{lab:45s} dispatch {d_id:4d}  {dur.get(d_id, 0):8.1f} us  MFMA busy {busy:5.3f}  clock {act / 8 / max(dur.get(d_id, 1), 1e-9) / 1e3:5.2f} GHz")
    k += 1
open("$OUT/pmc_gemm_f16_$R.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $d
ls -la $OUT
