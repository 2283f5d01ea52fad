#!/usr/bin/env python3
"""Per-kernel summary (calls, total and average duration) of a rocprofv3 --kernel-trace run.

  python scripts/kernel_stats.py <dir with *_kernel_stats.csv or *_kernel_trace.csv> > profiles/rNN_..._kernel_stats.csv

Uses rocprofv3's own *_kernel_stats.csv when --stats wrote one (durations converted to microseconds), otherwise sums
End_Timestamp - Start_Timestamp per kernel name over *_kernel_trace.csv.
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    rows = {}
    stats = sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        for r in csv.DictReader(open(stats[0])):
            name = r["Name"]
            calls = int(r["Calls"])
            if "TotalDurationNs" in r:
                tot = float(r["TotalDurationNs"]) / 1e3
            else:
                tot = float(r["TotalDurationUs"])
            rows[name] = [calls, tot]
    else:
        traces = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
        if not traces:
            raise SystemExit(f"{root}: no rocprofv3 kernel csv found")
        for r in csv.DictReader(open(traces[0])):
            e = rows.setdefault(r["Kernel_Name"], [0, 0.0])
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(v[1] for v in rows.values()) or 1.0
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, (calls, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, calls, round(tot, 3), round(tot / max(calls, 1), 3), f"{100.0 * tot / total:.4g}"])


if __name__ == "__main__":
    main()
