#!/usr/bin/env python3
"""Turns two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; one counter per pass) over one kernel into the
HBM-bytes-per-launch summary bench.py reads from profiles/.

  python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> \
      "<workload description>" > profiles/rNN_pmc_<kernel>_b<batch>.json
"""
import csv
import json
import sys


def column(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    if not vals:
        raise SystemExit(f"{path}: no {counter} rows for a kernel matching {kernel!r}")
    return vals


def main():
    fetch_csv, write_csv, kernel, workload = sys.argv[1:5]
    f, w = column(fetch_csv, "FETCH_SIZE", kernel), column(write_csv, "WRITE_SIZE", kernel)
    stat = lambda v: {"launches": len(v), "mean_raw_KiB": sum(v) / len(v), "min_raw_KiB": min(v), "max_raw_KiB": max(v)}
    rd = 2.0 * 1024.0 * sum(f) / len(f)     # KiB -> bytes; doubled: 128-byte requests are tallied as 64 bytes on gfx950
    wr = 1024.0 * sum(w) / len(w)
    print(json.dumps({
        "command": "rocprofv3 --pmc <COUNTER> --kernel-include-regex <kernel> -- python3 bench.py --no-cpu-baseline --steps 1 "
                   "--warmup 0 (one pass per counter)",
        "workload": workload, "kernel": kernel,
        "corrections": "FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 "
                       "bytes for wide coalesced reads, so it is doubled",
        "FETCH_SIZE": stat(f), "WRITE_SIZE": stat(w),
        "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}, indent=1))


if __name__ == "__main__":
    main()
