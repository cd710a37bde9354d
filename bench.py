#!/usr/bin/env python3
"""bench.py -- K-hop GCN-normalised SpMM throughput on MI355X (BASELINE.json metric).

One "step" = one K=2 propagation  Y = Â^2 X  over the whole graph (one pass of the hot path over one
batch of synthetic input).  N=1: the Reddit-shaped configuration (BASELINE.json configs[1]:
233k nodes, 114M edges, F=602).  N>1: the graph is 1-D row-partitioned, every rank owns a
Reddit-shaped row block of an N-times larger graph (weak scaling) and halo feature rows are exchanged
per hop by RCCL all-to-all over xGMI, overlapped with the interior-block SpMM.

Prints ONE JSON line on rank 0 (contract in the task statement): value = edges*feat/s of the whole
job, plus `roofline` (dominant kernel, live HIP-event timing) and `cpu_baseline` (the oracle's OpenMP
C port on a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="reddit", choices=["reddit", "arxiv", "products", "papers100M-shard"])
    ap.add_argument("--hops", type=int, default=2)
    ap.add_argument("--slices", type=int, default=int(os.environ.get("DISCO_BENCH_SLICES", "0")),
                    help="column slices per K-hop (0 = automatic: keep each slice's X under ~190 MB)")
    ap.add_argument("--ld-pad", type=int, default=32, help="pad the feature leading dimension to this many floats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0)
    ap.add_argument("--natural-order", action="store_true", help="keep natural row order (hub split only)")
    ap.add_argument("--hub-threshold", type=int, default=0)
    ap.add_argument("--hub-piece", type=int, default=0)
    ap.add_argument("--skew", type=float, default=None)
    ap.add_argument("--sweep", default="auto", choices=["auto", "on", "off"],
                    help="L2-blocked windowed launches (auto: when the graph re-uses source rows enough)")
    ap.add_argument("--halo-frac", type=float, default=0.25,
                    help="N > 1: share of every rank's entries whose column belongs to another rank (reported in config)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; reported in config)")
    return ap.parse_args()


def auto_slices(n_rows_x: int, F: int) -> int:
    """Column slices per hop.  Slicing X so that one slice fits the 256 MiB Infinity Cache does not pay: a gather
    that misses L2 costs the same whether MALL or HBM serves it, while every slice re-reads
    (col, val) and narrows the lane groups -- products 8.4 -> 7.0 ms, papers100M shard 36.6 -> 26.7 ms per step
    with one slice (profiles/r1/other_workloads.log)."""
    return 1


def b_min_bytes(nnz: int, n_rows: int, n_x_rows: int, F: int) -> float:
    """every CSR entry read once (int32 col + f32 val), rowptr once (4 B as
    nnz < 2^31), X read once, Y written once."""
    return nnz * 8.0 + (n_rows + 1) * 4.0 + (n_x_rows + n_rows) * F * 4.0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # DISCO_DIST_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks share cards,
    # halo rows staged through host memory); the driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("DISCO_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from disco_amd import SparseTensor, _lib, graphgen, utils
    lib = _lib.load()

    name = args.workload
    spec = graphgen.SPECS["papers100M" if name == "papers100M-shard" else name]
    n_local = spec.n_nodes if name != "papers100M-shard" else spec.n_nodes // 8
    nnz_target = spec.nnz if name != "papers100M-shard" else spec.nnz // 8
    if args.scale != 1.0:
        n_local = int(n_local * args.scale)
        nnz_target = int(nnz_target * args.scale * args.scale) if name == "reddit" else int(nnz_target * args.scale)
    F, K = spec.feat, args.hops
    skew = spec.skew if args.skew is None else args.skew
    ld = (F + args.ld_pad - 1) // args.ld_pad * args.ld_pad

    t_setup = time.time()
    if world == 1:
        # ---- single GPU: whole graph ---------------------------------------------------------------
        row, col = graphgen.powerlaw_edges(n_local, nnz_target, skew, spec.seed, device=dev, symmetric=spec.symmetric)
        adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n_local, n_local), is_sorted=True)
        del row, col
        if spec.norm == "normalize_adj":
            adj = utils.normalize_adj_csr(adj)      # utils.py:406-435 semantics, HIP kernels
        else:
            r, c, _ = adj.coo()                     # utils.gcn_norm(add_self_loops=False), _Sampled.py:491
            _, w = utils.gcn_norm(torch.stack([r, c]), None, n_local, add_self_loops=False)
            adj = adj.set_value(w)
            del r, c
        nnz = adj.nnz()
        x = graphgen.features(n_local, F, spec.seed, device=dev, ld=ld)
        y = torch.empty((n_local, ld), dtype=torch.float32, device=dev)[:, :F]
        tmp = torch.empty((n_local, ld), dtype=torch.float32, device=dev)[:, :F]
        n_slices = args.slices if args.slices > 0 else auto_slices(n_local, F)
        thr = -args.hub_threshold if args.natural_order and args.hub_threshold else (-2048 if args.natural_order else args.hub_threshold)
        ws, info = adj.schedule(thr, args.hub_piece, feat=F, sweep={"auto": None, "on": True, "off": False}[args.sweep])
        swept = bool(info.sweep_ws) and int(info.sweep_windows) > 0
        if swept:
            n_slices = 1      # the library cuts the columns into its own blocks (one C-ABI call per hop)
        sws_bytes = lib.disco_spmm_workspace_bytes(info, F)
        sws = torch.empty(max(int(sws_bytes), 16), dtype=torch.uint8, device=dev)
        per = (F + n_slices - 1) // n_slices
        per = (per + 31) // 32 * 32
        col_slices = [(c0, min(per, F - c0)) for c0 in range(0, F, per)]
        stream = _lib.current_stream(dev)
        events = []

        def step(record: bool):
            # same work as disco_khop_csr_f32, issued hop by hop so that each launch can be timed
            for c0, f in col_slices:
                src = x
                for k in range(K):
                    dst = y if k == K - 1 else tmp
                    if record:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    _lib.check(lib.disco_spmm_csr_f32(adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(),
                                                      n_local, src.data_ptr() + 4 * c0, src.stride(0), dst.data_ptr() + 4 * c0,
                                                      dst.stride(0), f, 0, ws.data_ptr(), info, sws.data_ptr(), sws.numel(),
                                                      stream), "disco_spmm_csr_f32")
                    if record:
                        e1.record()
                        events.append((e0, e1, f))
                    src = dst

        units_per_step = float(K) * nnz * F
        algo_bytes = lambda f: b_min_bytes(nnz, n_local, n_local, f)  # noqa: E731
        parallelism = "single"
        halo_info = None
    else:
        from disco_amd import dist as ddist
        part = ddist.build_weak_scaling_partition(spec, n_local, nnz_target, skew, world, rank, dev, ld, halo_frac=args.halo_frac)
        nnz = part.nnz_local
        n_slices = 1
        events = []

        def step(record: bool):
            part.khop(K, events if record else None)

        units_per_step = float(K) * part.nnz_global * F
        algo_bytes = lambda f: b_min_bytes(part.nnz_local, n_local, part.n_x_rows, f)  # noqa: E731
        parallelism = f"row-partition x{world}, halo all-to-all"
        halo_info = part.describe()
        info = part.info_main
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (k_spmm_rows), live HIP-event timing ------------------------
    durs_ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    tot_bytes = sum(algo_bytes(f) for _, _, f in events)
    avg_ms = sum(durs_ms) / max(len(durs_ms), 1)
    achieved = tot_bytes / (sum(durs_ms) * 1e-3) / 1e9 if durs_ms else 0.0
    swept_now = world == 1 and bool(info.sweep_ws) and int(info.sweep_windows) > 0
    kernel_name = ("k_spmm_rows_win: one timed call = one hop = %d window launches x column blocks of <= 128 floats"
                   % int(info.sweep_windows)) if swept_now else "k_spmm_rows (+ k_spmm_hub, same C-ABI call)"
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": kernel_name, "launches_per_step": len(events) // max(args.steps, 1),
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": round(tot_bytes / max(len(events), 1)),
                "gather_model_GBs": round((sum((nnz * 8.0 + nnz * f * 4.0 + n_local * f * 4.0) for _, _, f in events))
                                          / (sum(durs_ms) * 1e-3) / 1e9, 1) if durs_ms else 0.0}

    # HBM/fabric bytes per launch from the PMC passes committed under profiles/ (same command, same config);
    # bench.py cannot read hardware counters itself.
    try:
        with open(os.path.join(ROOT, "profiles", "r1", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        key = "reddit_K2_windowed" if swept_now else "reddit_K2_slices3"
        if world == 1 and name == "reddit" and args.scale == 1.0 and (swept_now or n_slices == 3) and key in pmc:
            roofline["traffic"] = round(pmc[key]["traffic_bytes_per_launch"])
            roofline["traffic_source"] = "profiles/r1/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 read correction)"
    except (OSError, KeyError, ValueError):
        pass

    value = units_per_step * args.steps / elapsed
    out = {
        "metric": "K-hop SpMM edges*feat/s", "value": value, "unit": "edges*feat/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{spec.name} K={K} GCN-normalised propagation, N={n_local}{'/rank' if world > 1 else ''}, "
                               f"nnz={nnz}{'/rank' if world > 1 else ''} (incl. self-loops), F={F}",
                   "K": K, "F": F, "nnz_per_rank": nnz, "rows_per_rank": n_local, "col_slices": n_slices, "ld": ld,
                   "l2_windows": int(info.sweep_windows) if info.sweep_ws else 0,
                   "skew": skew, "parallelism": parallelism, "scale": args.scale,
                   "schedule": {"hub_rows": int(info.n_hub_rows), "hub_pieces": int(info.n_hub_pieces),
                                "max_row_nnz": int(info.max_row_nnz)},
                   "setup_s": round(t_setup, 1)},
        "roofline": roofline,
    }
    if halo_info:
        out["config"]["halo"] = halo_info

    # ---- CPU baseline: the oracle's OpenMP C port on a bounded sample (rank 0, N=1 only) --------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        rows = args.cpu_sample_rows or max(256, min(n_local, int(1.0e10 / max(1.0, nnz / n_local * F))))
        rows = min(rows, n_local)
        rp = adj._rowptr[: rows + 1].cpu().numpy()
        e_end = int(rp[-1])
        cs = adj._col32[:e_end].cpu().numpy().astype("int64")
        vs = adj._value[:e_end].cpu().numpy()
        xh = x.cpu().numpy()
        c_oracle.spmm_csr_f32(rp[:65], cs, vs, xh)  # touch / warm
        tc = time.perf_counter()
        reps = 2
        for _ in range(reps):
            c_oracle.spmm_csr_f32(rp, cs, vs, xh, "sum")
        tc = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": e_end * F * reps / tc, "unit": "edges*feat/s", "cores": c_oracle.num_threads(),
                               "kind": "port",
                               "sample": f"rows [0,{rows}) of the same normalised CSR ({e_end} entries), {reps} SpMM passes at F={F}, "
                                         f"oracle/spmm_ref.c (OpenMP), {tc:.1f} s"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
