"""Idle time inside a rocprofv3 kernel trace: union of the kernel intervals vs the span they cover.

usage: trace_gaps.py <p_kernel_trace.csv> [skip_fraction]
Prints the span, the busy time (union over all queues), the idle time split by gap length, and the kernels that most
often follow a gap.  skip_fraction (default 0.3) drops the warm-up part of the trace by dispatch order.
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    iv = iv[int(len(iv) * skip):]
    span = iv[-1][1] - iv[0][0]
    busy = 0
    cur_s, cur_e = iv[0][0], iv[0][1]
    gaps = []
    for s, e, name in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in iv)
    print(f"kernels {len(iv)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {100 * (span - busy) / span:.1f} %  "
          f"sum of durations {ksum / 1e6:.3f} ms")
    buckets = collections.OrderedDict((k, [0, 0]) for k in ("<2us", "2-5us", "5-10us", "10-50us", ">50us"))
    for g, _ in gaps:
        k = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-50us" if g < 50000 else ">50us"
        buckets[k][0] += 1
        buckets[k][1] += g
    for k, (n, t) in buckets.items():
        print(f"  gaps {k:8s} {n:6d}  {t / 1e6:8.3f} ms")
    after = collections.Counter()
    for g, name in gaps:
        after[name[:70]] += g
    for name, t in after.most_common(12):
        print(f"  idle before {name:70s} {t / 1e6:8.3f} ms")


if __name__ == "__main__":
    main()
