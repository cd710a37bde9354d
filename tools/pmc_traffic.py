#!/usr/bin/env python3
"""Fold one workload's PMC passes (tools/prof_workload.sh) into profiles/r2/pmc_traffic.json, the file bench.py reads
`roofline.traffic` from.  Key = <workload>_<graph>_<row order>_windows<P>: the configuration the passes were taken with;
bench.py reports the figure only when it runs exactly that configuration.
  traffic per hop = sum over every k_spmm* dispatch of (2 * FETCH_SIZE + WRITE_SIZE) KiB / hops run
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 128-B requests as 64 B on gfx950; WRITE_SIZE is exact).
Usage: tools/pmc_traffic.py <tag> <workload> <graph> <hops run in a PMC pass> [out.json]"""
import json
import os
import sys


def main():
    tag, name, graph, hops = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else "profiles/r2/pmc_traffic.json"
    bench = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
    fetch = json.load(open(f"gpurun_out/{tag}_fetch.json"))
    write = json.load(open(f"gpurun_out/{tag}_write.json"))
    l2 = json.load(open(f"gpurun_out/{tag}_l2.json"))
    rd = sum(2 * d["FETCH_SIZE"] * 1024 for k, d in fetch.items() if "k_spmm" in k)
    wr = sum(d["WRITE_SIZE"] * 1024 for k, d in write.items() if "k_spmm" in k)
    hit = sum(d.get("TCC_HIT_sum", 0) for k, d in l2.items() if "k_spmm" in k)
    miss = sum(d.get("TCC_MISS_sum", 0) for k, d in l2.items() if "k_spmm" in k)
    cfg = bench["config"]
    key = f"{name}_{graph}_{cfg['row_order']}_windows{cfg['l2_windows']}"
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    entry = {"traffic_bytes_per_hop": (rd + wr) / hops, "read_bytes_per_hop": rd / hops, "write_bytes_per_hop": wr / hops,
             "l2_hit_rate": round(hit / max(hit + miss, 1), 4), "algorithmic_bytes_per_hop": algo,
             "traffic_over_algorithmic": round((rd + wr) / hops / algo, 2), "hops_in_pass": hops,
             "dispatches_per_hop": sum(d["dispatches"] for k, d in fetch.items() if "k_spmm" in k) / hops,
             "ms_per_hop_untraced": bench["roofline"]["avg_launch_ms"],
             "source": f"tools/prof_workload.sh {tag}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, one pass each"}
    allv = json.load(open(out)) if os.path.exists(out) else {}
    allv[key] = entry
    with open(out, "w") as fh:
        json.dump(allv, fh, indent=1, sort_keys=True)
    print(key, json.dumps(entry))


if __name__ == "__main__":
    main()
