#!/usr/bin/env python3
"""Summarises a rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) into per-kernel
MFMA utilisation: matrix-pipe busy cycles / (SIMDs x kernel cycles).

  python scripts/pmc_mfma_util.py <counter_collection.csv> "<workload>" > profiles/rNN_pmc_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES sums over all SIMDs; GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs, so kernel cycles =
GRBM_GUI_ACTIVE / 8 and the denominator is 1024 SIMDs x that.
"""
import collections
import csv
import json
import sys

SIMDS, XCDS = 1024, 8


def main():
    path, workload = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[name] += 1
    out = []
    for name, c in acc.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        if cyc <= 0:
            continue
        out.append({"kernel": name, "launches": calls[name], "kernel_cycles": round(cyc),
                    "mfma_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
                    "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * cyc), 4)})
    out.sort(key=lambda k: -k["kernel_cycles"])
    print(json.dumps({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...",
                      "workload": workload, "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
