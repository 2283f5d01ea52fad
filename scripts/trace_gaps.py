#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: idle time between consecutive launches of one kernel (by start time) and what ran between.

  python scripts/trace_gaps.py <dir with *_kernel_trace.csv> <kernel substring>
"""
import csv
import glob
import os
import statistics
import sys


def main():
    root, sub = sys.argv[1], sys.argv[2]
    path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(path))]
    rows.sort()
    att = [r for r in rows if sub in r[2]]
    dur = [(e - s) / 1e3 for s, e, _, _ in att]
    gaps = [(att[i + 1][0] - att[i][1]) / 1e3 for i in range(len(att) - 1)]
    small = [g for g in gaps if g < 100.0]
    print(f"{len(att)} launches of *{sub}*: duration median {statistics.median(dur):.1f} us, mean {statistics.mean(dur):.1f} us")
    print(f"gap to the next launch: median {statistics.median(gaps):.1f} us; below 100 us: {len(small)} gaps, median {statistics.median(small):.2f} us, "
          f"mean {statistics.mean(small):.2f} us; 100 us and more: {len(gaps) - len(small)} gaps, sum {sum(g for g in gaps if g >= 100.0) / 1e3:.1f} ms")
    qs = sorted(set(r[3] for r in att))
    print("queues:", qs)
    hist = {}
    for g in small:
        b = int(g)
        hist[b] = hist.get(b, 0) + 1
    print("gap histogram (us: count):", dict(sorted(hist.items())[:25]))
    # busy share of the kernel's CUs: launches of the two halves may overlap (two streams), so the intervals are merged
    merged, span0, span1 = 0, att[0][0], max(e for _, e, _, _ in att)
    cur_s, cur_e = att[0][0], att[0][1]
    for s, e, _, _ in att[1:]:
        if s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            merged += cur_e - cur_s
            cur_s, cur_e = s, e
    merged += cur_e - cur_s
    overlap = sum(e - s for s, e, _, _ in att) - merged
    print(f"from the first launch's start to the last one's end: {(span1 - span0) / 1e6:.2f} ms, some launch of *{sub}* running for "
          f"{merged / 1e6:.2f} ms = {100.0 * merged / (span1 - span0):.1f} % of it; two launches at once for {overlap / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
