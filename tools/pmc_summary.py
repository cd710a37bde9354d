#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, dispatch count and the sum of every counter collected.
Usage: python tools/pmc_summary.py <rocprofv3 output dir> [--json out.json]
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM):
read bytes = 2 * FETCH_SIZE * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add((f, row["Dispatch_Id"]))
    out = {}
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        d = {"dispatches": len(disp[k])}
        d.update({c: v for c, v in agg[k].items()})
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
        if "FETCH_SIZE" in d:
            d["read_bytes_per_dispatch"] = 2 * d["FETCH_SIZE"] * 1024 / d["dispatches"]
        if "WRITE_SIZE" in d:
            d["write_bytes_per_dispatch"] = d["WRITE_SIZE"] * 1024 / d["dispatches"]
        out[k] = d
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)
    for k, d in out.items():
        print(k[:90], json.dumps(d))


if __name__ == "__main__":
    main()
