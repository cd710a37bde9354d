"""Synthetic graphs shaped like the datasets BASELINE.json names.

There is no network on the build or GPU boxes, so ogbn-arxiv / Reddit / ogbn-products /
ogbn-papers100M are replaced by seeded generators with the same node count, edge count and feature
width.  G(N, nnz, skew, seed): every node gets a weight ``rank^-skew`` (ranks are a seeded random
permutation, so hubs are scattered over the id range), both endpoints of each edge are drawn
from that weight distribution by inverse-CDF sampling (a configuration-model draw), self pairs are
dropped, the list is symmetrised when the reference symmetrises that dataset
(/root/reference/utils.py:99-100) and duplicates are removed.  skew = 0 is the uniform, worst
locality variant.  Pure torch, runs on CPU (tests) or on the GPU (bench); the exact nnz after
de-duplication is what every throughput formula uses.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass(frozen=True)
class GraphSpec:
    name: str
    n_nodes: int
    nnz: int          # directed entries the dataset has (before self-loops)
    feat: int
    skew: float
    symmetric: bool
    norm: str         # "normalize_adj" (utils.py:406-435: +I, row degree) or "gcn_norm" (utils.py:228-247)
    seed: int


# BASELINE.json configs (C4 is the pair scorer and has no graph)
SPECS = {
    "arxiv": GraphSpec("ogbn-arxiv-shaped", 169_343, 1_166_243, 128, 0.667, True, "normalize_adj", 20250223),
    "reddit": GraphSpec("reddit-shaped", 232_965, 114_615_892, 602, 0.4, True, "normalize_adj", 20250224),
    "products": GraphSpec("ogbn-products-shaped", 2_449_029, 61_859_140, 100, 0.9, True, "normalize_adj", 20250225),
    "papers100M": GraphSpec("ogbn-papers100M-shaped", 111_059_956, 1_615_685_872, 128, 0.9, True, "gcn_norm", 20250227),
}


def _sample_nodes(cdf: torch.Tensor, m: int, gen: torch.Generator) -> torch.Tensor:
    u = torch.rand(m, generator=gen, device=cdf.device, dtype=torch.float64)
    return torch.searchsorted(cdf, u).clamp_(max=cdf.numel() - 1)


def powerlaw_edges(n_nodes: int, nnz: int, skew: float, seed: int, device="cpu", symmetric: bool = True,
                   n_cols: Optional[int] = None, add_diag: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Unique (row, col) pairs, row-major sorted.  ``n_cols`` > n_nodes makes a rectangular row block
    (used for the row-partitioned multi-GPU shards; ``symmetric`` must then be False)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    n_cols = n_nodes if n_cols is None else n_cols
    assert not (symmetric and n_cols != n_nodes)

    def cdf_for(n):
        rank = torch.randperm(n, generator=gen, device=device).to(torch.float64) + 1.0
        w = rank.pow(-skew) if skew > 0 else torch.ones(n, dtype=torch.float64, device=device)
        c = torch.cumsum(w, 0)
        return c / c[-1]

    cdf_r = cdf_for(n_nodes)
    cdf_c = cdf_r if n_cols == n_nodes else cdf_for(n_cols)
    m = nnz // 2 if symmetric else nnz
    keys = []
    # draw in chunks to bound peak memory
    chunk = 1 << 26
    done = 0
    while done < m:
        k = min(chunk, m - done)
        r = _sample_nodes(cdf_r, k, gen)
        c = _sample_nodes(cdf_c, k, gen)
        if n_cols == n_nodes:
            keep = r != c
            r, c = r[keep], c[keep]
        keys.append(r * n_cols + c)
        if symmetric:
            keys.append(c * n_cols + r)
        done += k
    if add_diag:
        d = torch.arange(min(n_nodes, n_cols), device=device, dtype=torch.int64)
        keys.append(d * n_cols + d)
    key = torch.unique(torch.cat(keys))  # sorted
    del keys
    return key // n_cols, key % n_cols


def features(n: int, F: int, seed: int, device="cpu", ld: Optional[int] = None) -> torch.Tensor:
    """X ~ N(0,1) float32 (standardised features as in /root/reference/utils.py:71-77)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 7919)
    ld = F if ld is None else ld
    buf = torch.zeros((n, ld), dtype=torch.float32, device=device)
    # fill in row blocks to bound the temporary
    step = max(1, (1 << 28) // max(F, 1))
    for i in range(0, n, step):
        j = min(n, i + step)
        buf[i:j, :F] = torch.randn((j - i, F), generator=gen, device=device, dtype=torch.float32)
    return buf[:, :F]
