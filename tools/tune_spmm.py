#!/usr/bin/env python3
"""Sweep SpMM variants on one graph in ONE process (interleaved rounds, median + min per variant).

Variants: column slices, row order (degree-binned / natural), forced (G, NV), max vector width,
leading-dimension padding and alternative builds of the library (different unroll depths), e.g.
  python tools/tune_spmm.py --workload reddit --libs default,u3_2,u3_8 --slices 1,2,3,4,6
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disco_amd import SparseTensor, _lib, graphgen, utils  # noqa: E402
from disco_amd.build import lib_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--libs", default="default")
    ap.add_argument("--slices", default="1,2,3,4")
    ap.add_argument("--orders", default="binned,natural")
    ap.add_argument("--lds", default="608")
    ap.add_argument("--force", default="auto", help="comma list of G:NV or auto")
    ap.add_argument("--maxvec", default="4")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--skew", type=float, default=None)
    ap.add_argument("--hub", default="0:0", help="comma list of threshold:piece")
    ap.add_argument("--sweep", default="0", help="comma list of 0/1: use the L2-blocked sweep kernel")
    ap.add_argument("--window-kb", default="2048", help="comma list: L2 window size in KiB (sweep)")
    ap.add_argument("--bpc", default="4", help="comma list: sweep blocks per CU")
    ap.add_argument("--mode", default="2", help="comma list: 1 = persistent sweep kernel, 2 = windowed launches")
    ap.add_argument("--rpg", default="4", help="comma list: rows per lane group per windowed launch")
    ap.add_argument("--wincols", default="128", help="comma list: column block width of the windowed launches")
    ap.add_argument("--pct", default="100", help="comma list: sweep_mode 3 pacing threshold (percent of the sync group)")
    ap.add_argument("--polls", default="256", help="comma list: sweep_mode 3 poll cap")
    ap.add_argument("--addr24", default="1", help="comma list of 0/1: 24-bit gather addressing")
    ap.add_argument("--ncols", type=int, default=0, help="gather from a table of this many rows (rectangular, unnormalised)")
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    spec = graphgen.SPECS[args.workload]
    n = int(spec.n_nodes * args.scale)
    nnz_t = int(spec.nnz * args.scale)
    skew = spec.skew if args.skew is None else args.skew
    if args.ncols:
        row, col = graphgen.powerlaw_edges(n, nnz_t, skew, spec.seed, device=dev, symmetric=False, n_cols=args.ncols)
        adj = SparseTensor(row=row, col=col, value=torch.rand(row.numel(), device=dev), sparse_sizes=(n, args.ncols), is_sorted=True)
    else:
        row, col = graphgen.powerlaw_edges(n, nnz_t, skew, spec.seed, device=dev)
        adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
        adj = utils.normalize_adj_csr(adj)
    del row, col
    nnz, F = adj.nnz(), spec.feat
    deg = adj._rowptr[1:] - adj._rowptr[:-1]
    print(json.dumps({"n": n, "nnz": nnz, "F": F, "max_deg": int(deg.max()), "mean_deg": float(deg.float().mean())}), flush=True)

    libs = {}
    for name in args.libs.split(","):
        path = lib_path() if name == "default" else lib_path(f"libdisco_hip_{name}.so")
        libs[name] = _lib.load(path)

    variants = []
    for ld in [int(v) for v in args.lds.split(",")]:
        for libname in libs:
            for order in args.orders.split(","):
                for hub in args.hub.split(","):
                    for force in args.force.split(","):
                        for mv in [int(v) for v in args.maxvec.split(",")]:
                            for s in [int(v) for v in args.slices.split(",")]:
                                for sw in [int(v) for v in args.sweep.split(",")]:
                                    for wkb in ([int(v) for v in args.window_kb.split(",")] if sw else [0]):
                                        for bpc in ([int(v) for v in args.bpc.split(",")] if sw else [0]):
                                            for mode in ([int(v) for v in args.mode.split(",")] if sw else [0]):
                                                for wc in ([int(v) for v in args.wincols.split(",")] if sw and mode >= 2 else [0]):
                                                    for rpg in ([int(v) for v in args.rpg.split(",")] if sw and mode >= 2 else [0]):
                                                        for a24 in [int(v) for v in args.addr24.split(",")]:
                                                          for pct in ([int(v) for v in args.pct.split(",")] if mode == 3 else [0]):
                                                           for polls in ([int(v) for v in args.polls.split(",")] if mode == 3 else [0]):
                                                            variants.append(dict(pct=pct, polls=polls, ld=ld, lib=libname, order=order, hub=hub, force=force, maxvec=mv,
                                                                                 slices=s, sweep=sw, wkb=wkb, bpc=bpc, mode=mode, wc=wc, rpg=rpg, a24=a24))

    bufs = {}
    scheds = {}
    times = {i: [] for i in range(len(variants))}
    stream = _lib.current_stream(dev)
    for rnd in range(args.rounds + 1):
        for i, v in enumerate(variants):
            lib = libs[v["lib"]]
            ld = v["ld"]
            if ld not in bufs:
                x = graphgen.features(args.ncols or n, F, spec.seed, device=dev, ld=ld)
                y = torch.empty((n, ld), dtype=torch.float32, device=dev)[:, :F]
                bufs[ld] = (x, y)
            x, y = bufs[ld]
            thr, piece = (int(t) for t in v["hub"].split(":"))
            key = (v["lib"], v["order"], thr, piece, v["sweep"], v["wkb"], v["mode"], v["wc"])
            if key not in scheds:
                nb = lib.disco_schedule_workspace_bytes(n, nnz, thr, piece)
                ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
                info = _lib.ScheduleInfo()
                t = thr if thr else 2048
                _lib.check(lib.disco_schedule_build(adj._rowptr.data_ptr(), n, nnz, -t if v["order"] == "natural" else thr, piece,
                                                    ws.data_ptr(), ws.numel(), info, stream), "schedule")
                info.n_cols = adj.size(1)
                sb = lib.disco_spmm_workspace_bytes(info, F)
                sws = torch.empty(max(int(sb), 16), dtype=torch.uint8, device=dev)
                swp = None
                if v["sweep"]:
                    cap = 256 if v["mode"] == 1 else v["wc"]
                    nblk = (F + cap - 1) // cap
                    blk = ((F + nblk - 1) // nblk + 31) // 32 * 32
                    wrows = max(64, v["wkb"] * 1024 // (4 * blk))
                    nbs = lib.disco_sweep_workspace_bytes(info, adj.size(1), wrows)
                    swp = torch.empty(int(nbs), dtype=torch.uint8, device=dev)
                    _lib.check(lib.disco_sweep_build(adj._rowptr.data_ptr(), adj._col32.data_ptr(), ws.data_ptr(), info, adj.size(1),
                                                     wrows, swp.data_ptr(), swp.numel(), stream), "sweep_build")
                    print("sweep:", {"windows": info.sweep_windows, "window_rows": info.sweep_window_rows, "MB": swp.numel() >> 20}, flush=True)
                scheds[key] = (ws, info, sws, swp)
            ws, info, sws, swp = scheds[key]
            lib.disco_set_tuning(b"sweep_blocks_per_cu", v["bpc"])
            lib.disco_set_tuning(b"sweep_mode", v["mode"])
            lib.disco_set_tuning(b"win_cols", v["wc"])
            lib.disco_set_tuning(b"win_rpg", v["rpg"])
            lib.disco_set_tuning(b"addr24", v["a24"])
            lib.disco_set_tuning(b"psync_pct", v["pct"])
            lib.disco_set_tuning(b"psync_polls", v["polls"])
            if v["force"] == "auto":
                lib.disco_set_tuning(b"force_g", 0)
                lib.disco_set_tuning(b"force_nv", 0)
            else:
                g, nv = (int(t) for t in v["force"].split(":"))
                lib.disco_set_tuning(b"force_g", g)
                lib.disco_set_tuning(b"force_nv", nv)
            lib.disco_set_tuning(b"max_vec", v["maxvec"])
            s = v["slices"]
            per = ((F + s - 1) // s + 31) // 32 * 32
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c0 in range(0, F, per):
                f = min(per, F - c0)
                rc = lib.disco_spmm_csr_f32(adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n,
                                            x.data_ptr() + 4 * c0, x.stride(0), y.data_ptr() + 4 * c0, y.stride(0), f, 0,
                                            ws.data_ptr(), info, sws.data_ptr(), sws.numel(), stream)
                _lib.check(rc, "spmm")
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[i].append(e0.elapsed_time(e1))
    rows = []
    for i, v in enumerate(variants):
        med, mn = statistics.median(times[i]), min(times[i])
        rows.append((med, mn, v))
    rows.sort(key=lambda r: r[0])
    for med, mn, v in rows:
        print(f"{med:9.3f} ms (min {mn:9.3f})  {nnz * F / med / 1e6:9.1f} G edge*feat/s  gather {((nnz * 8 + nnz * F * 4 + n * F * 4) / med / 1e9):7.2f} TB/s  {v}", flush=True)


if __name__ == "__main__":
    main()
