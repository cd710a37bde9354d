#!/usr/bin/env python3
"""BASELINE.json config 4: edge-translation pair scoring over n condensed nodes (default 25 000, d=128,
nfeat = 2d).  Reports pairs/s, achieved fp32 FLOP/s of the layer-2 GEMM (2*H*H per pair) against the
fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) and the time of the post-processing kernels."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd.models.parametrized_adj_lp import PGE_Edge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=25000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threshold", type=float, default=0.99)
    ap.add_argument("--ref-rows", type=int, default=8, help="rows of the reference-style torch loop to time")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    nfeat, H = 2 * args.d, 256
    m = PGE_Edge(nfeat=nfeat).to(dev).eval()
    with torch.no_grad():
        m.layers[-1].weight.mul_(4.0)
    Z = torch.randn(args.n, nfeat, device=dev)
    S = m.score_all_pairs(Z)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        S = m.score_all_pairs(Z)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = min(ts)
    pairs = float(args.n) ** 2
    flops = pairs * (2.0 * H * H + 2.0 * H)
    t0 = time.perf_counter()
    ei, ew = m.edge_construction(Z, args.threshold, "transductive")
    torch.cuda.synchronize()
    t_edges = time.perf_counter() - t0
    # the reference's loop: one n x 4d batch per row i through the torch module (LargeScaleCondensing.py:304-305)
    import numpy as np
    t0 = time.perf_counter()
    for i in range(args.ref_rows):
        row = m.inference(torch.cat([Z[[i] * args.n], Z[np.arange(args.n)]], axis=1)).detach().cpu()
    t_ref = (time.perf_counter() - t0) / max(args.ref_rows, 1) * args.n
    print(json.dumps({"n": args.n, "d": args.d, "pairs_per_s": pairs / t, "score_s": t, "tflops_layer2": flops / t / 1e12,
                      "frac_fp32_mfma_peak": flops / t / 157.3e12, "edge_construction_total_s": t_edges, "edges_kept": int(ew.numel()),
                      "reference_style_torch_loop_s_extrapolated": t_ref, "speedup_vs_loop": t_ref / t_edges}))


if __name__ == "__main__":
    main()
