"""Ordered kernel sequence of the LAST step in a rocprofv3 kernel trace (csv), with per-kernel duration: a reading aid for
launch-count work.  usage: trace_sequence.py <p_kernel_trace.csv> <marker-kernel-substring>   (a step ends after the
last launch whose name contains the marker, e.g. adamw)"""
import csv
import sys


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(ends) < 2:
        raise SystemExit("fewer than two steps in the trace")
    lo, hi = ends[-2] + 1, ends[-1] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("clipfs::", "").replace("void ", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>2s}  {name[:90]}")
        prev_end = max(prev_end, e)
    print(f"{hi - lo} launches, span {(int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
