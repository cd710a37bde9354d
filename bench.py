#!/usr/bin/env python3
"""bench.py -- K-hop GCN-normalised SpMM throughput on MI355X (BASELINE.json metric).

One "step" = one K=2 propagation  Y = A^2 X  over the whole graph (one pass of the hot path over one batch of
synthetic input).

N=1   the Reddit-shaped configuration (BASELINE.json configs[1]: 233k nodes, 114.6 M edges, F=602).
N>1   Headline = weak scaling: ONE global graph of N x 233k nodes and N x 114.6 M edges with the same mean degree,
      run under two partitions (disco_amd/dist.py), the faster one reported as the headline, the other as an extra:
      features  every rank holds the whole graph and F/N feature columns; A^K X is independent per column, so there is
                NO exchange (the default whenever the graph fits one GPU's 288 GB);
      rows      1-D row partition: every rank generates exactly the entries of its own rows (drawn over the GLOBAL id
                range), halo feature rows are exchanged per hop by RCCL (all-gather of the X shards or all-to-all of
                the needed rows, chosen from the halo that RESULTS) and overlapped with the interior-block SpMM.
      With WORLD_SIZE unset, `--gpus N` starts the N ranks itself (before any GPU call).

Besides the headline the line carries `extra_workloads`: ogbn-products-shaped strong scaling (configs[2]: the
SAME 2.45 M-node graph over N ranks), the papers100M-shaped graph at N = 8 (configs[4]), and the community
variants of the generators (graphgen.py) -- see DESIGN.md section 4.

Prints the JSON line on rank 0 (contract in the task statement): value = edges*feat/s of the whole job, plus
`roofline` (dominant kernel, live HIP-event timing) and `cpu_baseline` (N=1 only: the oracle's OpenMP C port and
BASELINE.md's two torch forms, on a bounded sample).  Exactly one line is printed, compact (the headline in full, extras
in a few dozen bytes each; the complete record goes to stderr and gpurun_out/bench_detail_n<N>.json); a watchdog prints it
with what has been measured if the extra workloads overrun their budget, so that nothing inside an extra can lose the headline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASELINE_CONFIG = {"products": "configs[2]", "papers100M": "configs[4]"}   # BASELINE.json entries that name the row partition
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="reddit", choices=["reddit", "arxiv", "products", "papers100M", "papers100M-shard"])
    ap.add_argument("--graph", default="config", choices=["config", "config-truncated", "community", "community-scrambled"],
                    help="generator variant (graphgen.py): locality-free configuration model [headline], LFR-style "
                         "communities in partitioner order, or the same behind a random relabelling (N = 1 only)")
    ap.add_argument("--max-degree", type=int, default=0,
                    help="truncate the head of the degree law: largest expected degree (0 = untruncated; --graph config-truncated "
                         "uses graphgen.PRODUCTS_MAX_DEGREE)")
    ap.add_argument("--locality-max-group", type=int, default=8192, help="group size bound of --order locality (0: none)")
    ap.add_argument("--locality-sweeps", type=int, default=8, help="label-propagation sweeps of --order locality")
    ap.add_argument("--order", default="auto", choices=["auto", "degree", "locality"],
                    help="row schedule: degree-binned, or grouped by the device-side locality pass first")
    ap.add_argument("--partition", default="auto", choices=["auto", "rows", "features"],
                    help="N > 1: replicate the graph and split the feature columns (no exchange; auto when the graph fits one "
                         "GPU), or 1-D row partition with halo exchange over RCCL")
    ap.add_argument("--slab-pad", type=int, default=16, help="feature partition: pad a slab's leading dimension to this many floats")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="on ONE GPU: run rank 0's share of an N-rank feature-partition job (it has no exchange)")
    ap.add_argument("--feat", type=int, default=0, help="override the feature width (experiments; reported in config)")
    ap.add_argument("--hops", type=int, default=2)
    ap.add_argument("--ld-pad", type=int, default=32, help="pad the feature leading dimension to this many floats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0)
    ap.add_argument("--hub-threshold", type=int, default=0)
    ap.add_argument("--hub-piece", type=int, default=0)
    ap.add_argument("--skew", type=float, default=None)
    ap.add_argument("--sweep", default="auto", choices=["auto", "on", "off"],
                    help="L2-blocked windowed launches (auto: when the graph re-uses source rows enough)")
    ap.add_argument("--window-kb", type=int, default=0, help="L2 window size of the windowed launches in KiB (0 = library default)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "need", "dense"], help="N > 1: halo exchange")
    ap.add_argument("--extras", default="auto", choices=["auto", "none", "all"],
                    help="extra workloads in the line (auto: products strong scaling, community variants, papers100M at N=8)")
    ap.add_argument("--extras-budget-s", type=float, default=300.0, help="skip the remaining extras after this many seconds")
    ap.add_argument("--tune", default="", help="comma-separated key=value pairs for disco_set_tuning (experiments)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workloads (debug only; reported in config)")
    return ap.parse_args()


# ---- launcher ---------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) before anything touches the GPU."""
    n = args.gpus
    backend = os.environ.get("DISCO_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()          # does not initialise the GPU
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (set DISCO_DIST_BACKEND=gloo to rehearse "
              f"with ranks sharing a card)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---- byte models ----------------------------------------------------------------------
def b_min_bytes(nnz: int, n_rows: int, F: int) -> float:
    """Every CSR entry read once (int32 col + f32 val), rowptr once (4 B as nnz < 2^31), X read once, Y written once."""
    return nnz * 8.0 + (n_rows + 1) * 4.0 + 2.0 * n_rows * F * 4.0


def b_gather_bytes(nnz: int, n_rows: int, F: int) -> float:
    return nnz * 8.0 + (n_rows + 1) * 4.0 + nnz * F * 4.0 + n_rows * F * 4.0


class Ctx:
    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("DISCO_DIST_BACKEND", "nccl")
        idx = self.local_rank if self.backend == "nccl" else self.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(idx)
        self.dev = torch.device("cuda", idx)

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, warmup: int, steps: int) -> float:
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(warmup):
            step(False)
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        return self.max_over_ranks(time.perf_counter() - t0)


def graph_params(name: str, args, world: int):
    """(spec, n_global, nnz_global, scaling) of a workload at this world size."""
    from disco_amd import graphgen
    shard = name == "papers100M-shard"
    spec = graphgen.SPECS["papers100M" if shard else name]
    n, nnz = spec.n_nodes, spec.nnz
    if shard:                     # single-GPU stand-in: a square graph of 1/8 the size (r1 workload, kept for comparison)
        n, nnz = n // 8, nnz // 8
    if args.scale != 1.0:
        n = int(n * args.scale)
        nnz = int(nnz * args.scale * args.scale) if name == "reddit" else int(nnz * args.scale)
    if name == "reddit" and world > 1:
        return spec, n * world, nnz * world, "weak"
    return spec, n, nnz, ("weak" if world == 1 else "strong")


def run_spmm(ctx: Ctx, name: str, graph: str, steps: int, warmup: int, want_cpu: bool = False, partition: str = None):
    """Build one workload, time it, return the result dict (collective: every rank calls it)."""
    from disco_amd import SparseTensor, _lib, graphgen, utils
    from disco_amd import dist as ddist
    args, dev, world, rank = ctx.args, ctx.dev, ctx.world, ctx.rank
    # --emulate-world N on ONE GPU: do exactly what rank 0 of an N-rank feature-partition job does (that partition has
    # no exchange, so a rank's time IS the job's time); reported as n_gpus = 1 with the emulation named in config
    gworld, grank = (args.emulate_world, 0) if (world == 1 and args.emulate_world > 1) else (world, rank)
    lib = _lib.load()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        _lib.check(lib.disco_set_tuning(k.encode(), int(v)), f"disco_set_tuning({k})")
    spec, n_global, nnz_target, scaling = graph_params(name, args, gworld)
    F, K = (args.feat or spec.feat), args.hops
    skew = spec.skew if args.skew is None else args.skew
    partition = partition or args.partition
    if gworld == 1:
        partition = "single"
    elif partition == "auto":
        # Replicate the graph and split the FEATURE columns (no exchange at all) whenever the whole graph fits one GPU:
        # generator temporaries + CSR ~ 48 B per entry, three column slabs of X; else split the rows (halo exchange).
        need = nnz_target * 40 + 3 * n_global * ((F + gworld - 1) // gworld + 16) * 4
        partition = "features" if need < 0.8 * torch.cuda.get_device_properties(dev).total_memory or world == 1 else "rows"
    Fg = F                                  # global width; F below is what THIS rank propagates
    if partition == "features":
        c0, c1 = ddist.feature_slabs(Fg, gworld)[grank]
        F = c1 - c0
        assert F > 0, "more ranks than 4-float column groups"
    ld = (F + args.ld_pad - 1) // args.ld_pad * args.ld_pad
    if partition == "features":
        ld = (F + args.slab_pad - 1) // args.slab_pad * args.slab_pad
    comm, mu = (graphgen.COMMUNITY_SIZE, graphgen.COMMUNITY_MU) if graph.startswith("community") else (0, 1.0)
    max_deg = graphgen.PRODUCTS_MAX_DEGREE if graph == "config-truncated" else args.max_degree
    events = []
    t_setup = time.time()
    cfg = {"K": K, "F": Fg, "ld": ld, "skew": skew, "scale": args.scale,
           "graph": {"config": "configuration model, no locality" + (f", head truncated at expected degree {max_deg}" if max_deg else ""),
                     "config-truncated": f"configuration model, no locality, head truncated at expected degree {max_deg}",
                     "community": f"LFR-style communities of {comm} nodes, mixing mu={mu}, partitioner order",
                     "community-scrambled": f"LFR-style communities of {comm} nodes, mixing mu={mu}, ids scrambled"}[graph]}
    if partition in ("single", "features"):
        row, col = graphgen.powerlaw_edges(n_global, nnz_target, skew, spec.seed, device=dev, comm_size=comm, mu=mu,
                                           scramble=graph.endswith("scrambled"), max_degree=max_deg)
        adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n_global, n_global), is_sorted=True)
        del row, col
        if spec.norm == "normalize_adj":
            adj = utils.normalize_adj_csr(adj)      # utils.py:406-435 semantics, HIP kernels
        else:
            r, c, _ = adj.coo()                     # utils.gcn_norm(add_self_loops=False), _Sampled.py:491
            _, w = utils.gcn_norm(torch.stack([r, c]), None, n_global, add_self_loops=False)
            adj = adj.set_value(w)
            del r, c
        n_local, nnz, nnz_global = n_global, adj.nnz(), adj.nnz()
        x = graphgen.features(n_local, F, spec.seed + 31 * grank, device=dev, ld=ld)
        bufs = [torch.empty((n_local, ld), dtype=torch.float32, device=dev)[:, :F] for _ in range(min(K, 3))]
        order = args.order
        if order == "auto":         # community structure present and no L2 windows: let the locality pass find it
            windows_pay = args.sweep == "on" or (args.sweep == "auto" and adj.sweep_pays(F))
            order = "locality" if graph.startswith("community") and not windows_pay else "degree"
        if order == "locality":
            torch.cuda.synchronize()
            t0 = time.time()
            groups = adj.locality_groups(args.locality_sweeps, args.locality_max_group)
            torch.cuda.synchronize()
            t_pass = time.time() - t0
            r, c, _ = adj.coo()
            cfg["locality_pass"] = {"sweeps": args.locality_sweeps, "max_group": args.locality_max_group, "seconds": round(t_pass, 3),
                                    "groups": int(torch.unique(groups).numel()),
                                    "edges_inside_a_group": round(float((groups[r] == groups[c]).float().mean()), 4)}
            del r, c
            adj.set_row_groups(groups)
        ws, info = adj.schedule(args.hub_threshold, args.hub_piece, feat=F, sweep={"auto": None, "on": True, "off": False}[args.sweep],
                                window_bytes=(args.window_kb << 10) or None)
        sws_bytes = int(lib.disco_spmm_workspace_bytes(info, F))
        sws = torch.empty(max(sws_bytes, 16), dtype=torch.uint8, device=dev)
        stream = _lib.current_stream(dev)

        def step(record: bool):
            # the work of disco_khop_csr_f32, issued hop by hop so that each hop can be timed
            src = x
            for k in range(K):
                dst = bufs[0] if k == K - 1 else bufs[1 + (k & 1)]
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(lib.disco_spmm_csr_f32(adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(),
                                                  n_local, src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), F, 0,
                                                  ws.data_ptr(), info, sws.data_ptr(), sws.numel(), stream), "disco_spmm_csr_f32")
                if record:
                    e1.record()
                    events.append((e0, e1))
                src = dst

        windows = int(info.sweep_windows) if info.sweep_ws else 0
        par = "single" if partition == "single" else (
            f"feature partition x{gworld}: graph replicated on every rank, {F} of {Fg} columns per rank, no exchange" +
            (f" [rank 0's share run alone on one GPU: --emulate-world {gworld}]" if world == 1 else ""))
        cfg.update({"parallelism": par, "row_order": order, "l2_windows": windows, "xcd_run": int(info.xcd_run),
                    "schedule": {"hub_rows": int(info.n_hub_rows), "hub_pieces": int(info.n_hub_pieces),
                                 "max_row_nnz": int(info.max_row_nnz)}})
        kernel = (f"k_spmm_rows_win: one timed call = one hop = {windows} window launches x column blocks of <= 128 floats"
                  if windows else "k_spmm_rows (+ k_spmm_hub, same C-ABI call)")
    else:
        g = ddist.build_partitioned_graph(n_global, nnz_target, F, skew, spec.seed, spec.norm, world, rank, dev, ld,
                                          comm_size=comm, mu=mu, exchange=args.exchange, K=K, max_degree=max_deg)
        n_local, nnz, nnz_global = g.part.n_local, g.nnz_local, g.nnz_global
        order = args.order
        if order == "auto":
            order = "locality" if graph.startswith("community") and not g.part.adj_int.sweep_pays(F) else "degree"
        if order == "locality" and g.part.adj_int.nnz():    # interior block: square, local ids
            g.part.adj_int.set_row_groups(g.part.adj_int.locality_groups(args.locality_sweeps, args.locality_max_group))
        cfg["row_order"] = order
        g.part.adj_int.schedule(feat=F)
        if g.part.adj_halo.nnz():
            g.part.adj_halo.schedule(feat=F)
        ev3 = []

        def step(record: bool):
            g.part.time_waits = record        # events around every wait for the exchange: the part compute did not cover
            g.khop(K, ev3 if record else None)
            if record:
                events.extend((a, b) for a, b, _ in ev3)
                ev3.clear()

        cfg.update({"parallelism": f"1-D row partition x{world} of one global graph", "halo": g.describe(),
                    "rows_global": n_global})
        kernel = "k_spmm_rows on the interior block + halo block (y += A_halo x_recv), exchange overlapped"
        adj, x = None, None
    torch.cuda.synchronize()
    ctx.barrier()
    t_setup = time.time() - t_setup

    elapsed = ctx.timed(step, warmup, steps)
    durs_ms = [a.elapsed_time(b) for a, b in events]
    hop_ms = sum(durs_ms) / max(len(durs_ms), 1)
    if partition == "rows":
        waits = g.part.exposed_wait_ms()
        cfg["halo"]["exposed_exchange_ms_per_hop"] = round(sum(waits) / max(len(durs_ms), 1), 4)
        cfg["halo"]["exchange_GBs_if_fully_exposed"] = round(cfg["halo"]["recv_bytes_per_hop"] / max(sum(waits) / max(len(durs_ms), 1), 1e-6) / 1e6, 1)
    algo = b_min_bytes(nnz, n_local, F)                      # this rank's share per hop
    achieved = algo / (hop_ms * 1e-3) / 1e9 if durs_ms else 0.0
    value = float(K) * nnz_global * Fg * steps / elapsed
    if world == 1 and gworld > 1:
        value /= gworld                   # one rank's share of the emulated job
    cfg.update({"nnz_per_rank": nnz, "rows_per_rank": n_local, "nnz_global": nnz_global, "setup_s": round(t_setup, 1)})
    res = {
        "value": value, "ms_per_step": elapsed / steps * 1e3, "scaling": scaling, "n_gpus": world, "partition": partition,
        "workload": f"{spec.name}{' (1/8-size square stand-in)' if name == 'papers100M-shard' else ''} K={K} GCN-normalised "
                    f"propagation, N={n_global}, nnz={nnz_global} (incl. self-loops), F={Fg}",
        "config": cfg,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "traffic_measured_this_run": False, "kernel": kernel,
                     "launches_per_step": len(events) // max(steps, 1), "avg_launch_ms": round(hop_ms, 4),
                     "algorithmic_bytes_per_launch": round(algo),
                     "gather_model_GBs": round(b_gather_bytes(nnz, n_local, F) / (hop_ms * 1e-3) / 1e9, 1) if durs_ms else 0.0},
    }
    # HBM / fabric bytes per hop from the PMC passes committed under profiles/ for exactly this configuration
    # (bench.py cannot read hardware counters itself); null when no matching measurement exists.
    if world == 1 and gworld == 1 and args.scale == 1.0 and not args.feat:
        key = f"{name}_{graph}_{cfg['row_order']}_windows{cfg['l2_windows']}"
        for rnd in ("r3", "r2"):             # newest committed measurement of exactly this configuration
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                    pmc = json.load(fh)
                if key in pmc:
                    res["roofline"]["traffic"] = round(pmc[key]["traffic_bytes_per_hop"])
                    res["roofline"]["traffic_source"] = f"profiles/{rnd}/pmc_traffic.json[{key}] (committed PMC passes, not this run)"
                    break
            except (OSError, KeyError, ValueError):
                pass
    if want_cpu and world == 1 and gworld == 1:
        res["cpu_baseline"] = cpu_baselines(adj, x, F, args)
    return res


def cpu_baselines(adj, x, F, args):
    """CPU figures timed on this box's host cores, each on a bounded sample of the SAME normalised CSR (rows [0, R)):
      port      the oracle's OpenMP C restatement of torch_sparse's CPU spmm_sum (oracle/spmm_ref.c)   [headline object]
      torch_csr BASELINE.md section 2 "CSR form":     torch.sparse_csr @ X
      scatter   BASELINE.md section 2 "scatter form": index_select -> mul -> index_add_  (PyG's non-fused route)"""
    from oracle import c_oracle
    n_local, nnz = adj.size(0), adj.nnz()
    rows = args.cpu_sample_rows or max(256, min(n_local, int(6.0e10 / max(1.0, nnz / n_local * F))))
    rows = min(rows, n_local)
    rp_t = adj._rowptr[: rows + 1].cpu()
    e_end = int(rp_t[-1])
    col_t = adj._col32[:e_end].cpu().to(torch.int64)
    val_t = adj._value[:e_end].cpu()
    x_t = x.cpu().contiguous()
    rp, cs, vs, xh = rp_t.numpy(), col_t.numpy(), val_t.numpy(), x_t.numpy()
    c_oracle.spmm_csr_f32(rp[:65], cs, vs, xh)  # touch / warm
    reps = 2
    tc = time.perf_counter()
    for _ in range(reps):
        c_oracle.spmm_csr_f32(rp, cs, vs, xh, "sum")
    tc = time.perf_counter() - tc
    sample = f"rows [0,{rows}) of the same normalised CSR ({e_end} entries) at F={F}"
    out = {"value": e_end * F * reps / tc, "unit": "edges*feat/s", "cores": c_oracle.num_threads(), "kind": "port",
           "sample": f"{sample}, {reps} SpMM passes, oracle/spmm_ref.c (OpenMP), {tc:.1f} s"}
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    # bounded sub-samples for the two torch forms (they are 1-2 orders of magnitude slower than the port)
    forms = []
    try:
        r2 = max(64, rows // 8)
        e2 = int(rp_t[r2])
        a = torch.sparse_csr_tensor(rp_t[: r2 + 1], col_t[:e2], val_t[:e2], (r2, x_t.shape[0]))
        a @ x_t[:, :8]
        t0 = time.perf_counter()
        a @ x_t
        t1 = time.perf_counter() - t0
        forms.append({"form": "torch.sparse_csr @ X (BASELINE.md 2, CSR form)", "kind": "port", "value": e2 * F / t1,
                      "unit": "edges*feat/s", "cores": threads, "sample": f"rows [0,{r2}) ({e2} entries), 1 pass, {t1:.1f} s"})
        r3 = max(64, rows // 32)
        e3 = int(rp_t[r3])
        rowid = torch.repeat_interleave(torch.arange(r3), rp_t[1: r3 + 1] - rp_t[:r3])
        t0 = time.perf_counter()
        msg = x_t.index_select(0, col_t[:e3])
        msg.mul_(val_t[:e3, None])
        torch.zeros(r3, F).index_add_(0, rowid, msg)
        t2 = time.perf_counter() - t0
        forms.append({"form": "index_select -> mul -> index_add_ (BASELINE.md 2, scatter form)", "kind": "port",
                      "value": e3 * F / t2, "unit": "edges*feat/s", "cores": threads,
                      "sample": f"rows [0,{r3}) ({e3} entries), 1 pass, {t2:.1f} s"})
    except Exception as exc:  # noqa: BLE001  (a baseline that cannot run must not lose the GPU result)
        forms.append({"error": repr(exc)})
    out["torch_forms"] = forms
    return out


def run_scorer(ctx: Ctx, steps: int, warmup: int):
    """configs[3]: PGE_Edge over all pairs of 25 000 condensed nodes (MFMA-bound); step = one edge_construction."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    args, dev = ctx.args, ctx.dev
    n, d, H = max(256, int(25000 * args.scale)), 128, 256
    torch.manual_seed(20250226)
    pge = PGE_Edge(nfeat=2 * d, nhid=H, device=dev).to(dev).eval()
    Z = torch.randn(n, 2 * d, device=dev)
    events = []

    def step(record: bool):
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        pge.edge_construction(Z, 0.99, "transductive")
        if record:
            e1.record()
            events.append((e0, e1))

    elapsed = ctx.timed(step, warmup, steps)
    ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
    flops = 2.0 * n * n * (H * H + H) + 2.0 * n * 2 * (2 * d) * H
    peak = 157.3
    return {"workload": f"PGE_Edge pair scoring + symmetrise/threshold/COO, n={n}, nfeat={2 * d}, hidden={H}", "n_gpus": 1,
            "value": n * n * steps / elapsed, "unit": "pairs/s", "ms_per_step": elapsed / steps * 1e3,
            "roofline": {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4), "traffic": None,
                         "kernel": "k_scorer_pairs<256> (v_mfma_f32_32x32x2_f32), whole edge_construction timed"}}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ctx = Ctx(args)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(ctx.backend, **({"device_id": ctx.dev} if ctx.backend == "nccl" else {}))
    t_start = time.time()

    def headline(h):
        return {"value": h["value"], "ms_per_step": h["ms_per_step"], "scaling": h["scaling"],
                "config": dict(h["config"], workload=h["workload"]), "roofline": h["roofline"]}

    # N > 1, --partition auto: the feature partition first (no collective on its data path: nothing in it can hang) ...
    head = run_spmm(ctx, args.workload, args.graph, args.steps, args.warmup, want_cpu=not args.no_cpu_baseline)
    out = {"metric": "K-hop SpMM edges*feat/s", "value": None, "unit": "edges*feat/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
           "scaling": None, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": None, "roofline": None}
    out.update(headline(head))
    if "cpu_baseline" in head:
        out["cpu_baseline"] = head["cpu_baseline"]

    extras = []
    run_extras = args.extras != "none" and args.workload == "reddit" and args.graph == "config"
    # ONE JSON line is printed.  Nothing after the first measurement may lose it: if the rest overruns its budget by two
    # minutes (a hung collective, a rank that died), rank 0 prints the line with what it has -- marked "aborted" -- and
    # EVERY rank leaves with WATCHDOG_EXIT_CODE: a hang never ends with status 0.
    watchdog = arm_watchdog(max(1.0, args.extras_budget_s + 120.0 - (time.time() - t_start)) + (0 if rank0(ctx) else 5),
                            lambda: emit(dict(out, aborted=True), extras + [{"aborted": "the remaining workloads overran their time budget"}], world)
                            if rank0(ctx) else None, start=False)
    if world > 1 or run_extras:
        watchdog.start()
    if run_extras:
        # (workload, generator, partition); partition None = --partition (auto: feature columns when the graph fits)
        plan = [("products", "config", None), ("products", "config-truncated", None), ("products", "community", None)]
        if world > 1:
            big = world == 8 or args.extras == "all"
            # Order = what must not fall to the time budget comes first.  The headline has been measured under the
            # exchange-free feature partition (nothing in it can hang); next the 1-D row partition with halo exchange over
            # RCCL (north_star's design for graphs that exceed one GPU): the headline workload again -- the faster of its two
            # partitions becomes the headline, the other one stays in this list -- then BASELINE configs[2] (products,
            # row-partitioned) and configs[4] (papers100M-shaped, halo exchange over xGMI), which are ALWAYS reported under the
            # row partition as named entries, whichever partition wins a headline.  The feature-partition extras and the
            # community variants follow.
            rows_first = []
            if args.partition == "auto" and head.get("partition") == "features":
                rows_first.append(("headline", args.graph, "rows"))
            rows_first.append(("products", "config", "rows"))
            if big:
                rows_first.append(("papers100M", "config", "rows"))
            plan = rows_first + [("reddit", "community", None)] + plan
            if big:
                plan += [("papers100M", "config", None), ("papers100M", "community", None)]
            plan += [("products", "community", "rows")]
            if big:
                plan += [("papers100M", "community", "rows")]
        else:
            # N = 1: the pair scorer (configs[3]) and, last, the FULL papers100M-shaped graph on this one GPU (configs[4]'s
            # single-GPU anchor: 111 M nodes, 1.6 G entries, ~185 GB resident)
            plan += [("scorer", "", None), ("papers100M", "config", None)]
        for name, graph, part in plan:
            go = 1 if time.time() - t_start < args.extras_budget_s else 0
            if world > 1:       # rank 0 decides for everybody
                flag = torch.tensor([go], dtype=torch.int64, device=ctx.dev if ctx.backend == "nccl" else "cpu")
                dist.broadcast(flag, src=0)
                go = int(flag.item())
            if not go:
                extras.append({"workload": f"{name} ({graph})", "skipped": "time budget"})
                continue
            torch.cuda.empty_cache()
            try:
                if name == "scorer":
                    r = run_scorer(ctx, max(1, min(args.steps, 3)), 1)
                elif name == "headline":
                    r = run_spmm(ctx, args.workload, graph, args.steps, args.warmup, partition=part)
                    r["unit"] = "edges*feat/s"
                    if r["value"] > head["value"]:        # the same on every rank: times are maxima over the ranks
                        head, r = r, head
                        out.update(headline(head))
                    r["note"] = "the headline workload under the other partition"
                else:
                    r = run_spmm(ctx, name, graph, args.steps, args.warmup, partition=part)
                    r["unit"] = "edges*feat/s"
                    if part == "rows" and name in BASELINE_CONFIG:
                        r["baseline_config"] = BASELINE_CONFIG[name]
            except Exception as e:       # reported, never hidden; at N > 1 the other ranks meet the watchdog
                r = {"workload": f"{name} ({graph})", "failed": f"{type(e).__name__}: {e}"[:400]}
                extras.append(r)
                if world > 1:
                    break
                continue
            extras.append(r)
    watchdog.cancel()
    if rank0(ctx):
        emit(out, extras, world)
    if world > 1:
        dist.destroy_process_group()


WATCHDOG_EXIT_CODE = 3


def arm_watchdog(seconds: float, on_fire, start: bool = True):
    """Timer that, when it fires, runs ``on_fire()`` (rank 0: print the line with what has been measured, marked
    "aborted") and then ends the process with WATCHDOG_EXIT_CODE (non-zero: the launcher and the driver must see that
    the run did not finish; os._exit because a hung collective cannot be unwound).  cancel() it when the work is done."""
    import threading

    def fire():
        try:
            on_fire()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(WATCHDOG_EXIT_CODE)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    if start:
        t.start()
    return t


def rank0(ctx: Ctx) -> bool:
    return ctx.rank == 0


def compact_extra(e: dict) -> dict:
    """One extra workload in a few dozen bytes: whoever captures stdout may keep only its tail, so the ONE line
    stays short; everything else goes to stderr and gpurun_out/bench_detail_n<N>.json."""
    if "failed" in e or "skipped" in e or "aborted" in e:
        return {k: (str(v)[:120]) for k, v in e.items() if k in ("workload", "failed", "skipped", "aborted")}
    c = e.get("config", {})
    par = c.get("parallelism", "")
    out = {"workload": e.get("workload", "").split(" K=")[0].split(",")[0][:44], "ms_per_step": round(e["ms_per_step"], 3),
           "value": float(f"{e['value']:.4g}"), "frac": e.get("roofline", {}).get("frac")}
    if e.get("unit", "edges*feat/s") != "edges*feat/s":
        out["unit"] = e["unit"]
    if "graph" in c:
        out["graph"] = c["graph"].split(",")[0][:28] + (" (head truncated)" if "truncated" in c["graph"] else "")
    if par and par != "single":
        out["partition"] = "features" if par.startswith("feature") else "rows"
    if c.get("row_order") == "locality":
        out["row_order"] = "locality"
    if "note" in e:
        out["note"] = e["note"]
    if "baseline_config" in e:
        out["cfg"] = e["baseline_config"]
        halo = c.get("halo", {})
        if "exposed_exchange_ms_per_hop" in halo:
            out["exposed_xchg_ms"] = halo["exposed_exchange_ms_per_hop"]
    return out


def emit(out: dict, extras: list, n_gpus: int) -> None:
    """Print THE line (compact) on stdout; the full record goes to stderr and, when possible, to a file."""
    full = dict(out)
    if extras:
        full["extra_workloads"] = extras
    line = dict(out)
    if extras:
        line["extra_workloads"] = [compact_extra(e) for e in extras]
    cb = line.get("cpu_baseline")
    if cb and "torch_forms" in cb:
        cb = dict(cb)
        cb["sample"] = cb["sample"].split(" at F=")[0] + ", oracle/spmm_ref.c (OpenMP)"
        cb["torch_forms"] = [{"form": t["form"].split(" (")[0], "value": float(f"{t['value']:.4g}"), "cores": t["cores"]} for t in cb["torch_forms"]]
        line["cpu_baseline"] = cb
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_detail_n{n_gpus}.json"), "w") as fh:
            json.dump(full, fh)
        line["detail"] = f"gpurun_out/bench_detail_n{n_gpus}.json (also on stderr)"
    except OSError:
        line["detail"] = "stderr"
    print("BENCH_DETAIL " + json.dumps(full), file=sys.stderr, flush=True)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
