"""Synthetic graphs shaped like the datasets BASELINE.json names.

There is no network on the build or GPU boxes, so ogbn-arxiv / Reddit / ogbn-products /
ogbn-papers100M are replaced by seeded generators with the same node count, EXACTLY the same entry count
and the same feature width.

Model.  Every node gets a weight ``rank^-skew`` (ranks are a seeded random permutation, so hubs are
scattered over the id range).  An undirected edge {u, v} is drawn

  * "config" (default, ``mu = 1``): both endpoints from the weight distribution (a configuration-model
    draw).  No locality whatsoever: the worst case for caches and for a row partition;
  * "community" (``comm_size > 0``, mixing parameter ``mu`` as in the LFR benchmark): with probability
    ``1 - mu`` v is drawn from u's community (a run of ``comm_size`` consecutive ids) by the same weights,
    with probability ``mu`` from the whole graph.  ``scramble=True`` hides the structure behind a random
    relabelling (the "as downloaded" order a locality pass has to undo); ``scramble=False`` is the order a
    partitioner hands back (communities contiguous).

Self pairs are dropped, duplicates removed, and the draw is topped up / trimmed until the number of
undirected edges is exactly ``nnz // 2`` -- the entry count every throughput formula uses is the one
BASELINE names, not "whatever survived de-duplication".

Row partitions (multi-GPU): the id range is cut into ``world`` equal blocks and the edge set is the union of
independent draws per BLOCK PAIR (a, b), each with its own seed and an exact quota, so rank g can generate
precisely the entries of its own rows -- pair (g, h) for every h -- without ever holding the global graph,
and rank h derives the mirrored entries from the same seed.  ``world = 1`` is the single-GPU generator.
Communities never straddle a block.  Pure torch; runs on CPU (tests) or on the GPU (bench).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

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

# community variant reported next to the headline: LFR-style mixing
COMMUNITY_SIZE, COMMUNITY_MU = 4096, 0.3
# truncated-head variant of the products-shaped graph (round-2 verdict item 7): largest expected degree of the order
# of a real co-purchase graph instead of the untruncated power law's 680 k
PRODUCTS_MAX_DEGREE = 20_000


def block_offsets(n_global: int, world: int) -> List[int]:
    return [n_global * g // world for g in range(world + 1)]


def node_weight_cdf(n: int, skew: float, seed: int, device, max_degree: int = 0, nnz: int = 0) -> torch.Tensor:
    """Unnormalised cumulative node weights (float64), identical on every rank for the same arguments.
    ``max_degree`` > 0 (with the graph's entry count ``nnz``) truncates the head of the power law: weights are capped
    so that no node's EXPECTED degree (nnz * w_i / sum w) exceeds ``max_degree`` -- the untruncated law gives the
    products-shaped graph one row with 28 % of all nodes as neighbours, far heavier than a co-purchase graph's head
    (order 10^4)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if skew > 0:
        w = (torch.randperm(n, generator=gen, device=device).to(torch.float64) + 1.0).pow_(-skew)
    else:
        w = torch.ones(n, dtype=torch.float64, device=device)
    if max_degree > 0 and nnz > 0 and skew > 0:
        cap = float(w.max())
        for _ in range(32):             # fixed point of cap = max_degree * sum(min(w, cap)) / nnz (monotone, converges fast)
            new_cap = max_degree * float(torch.clamp(w, max=cap).sum()) / nnz
            if new_cap >= cap * (1 - 1e-12):
                break
            cap = new_cap
        w = torch.clamp(w, max=cap)
    return torch.cumsum(w, 0)


def _cdf_at(cdf: torch.Tensor, idx) -> torch.Tensor:
    """Cumulative weight of all ids < idx (idx: python int or int64 tensor)."""
    if isinstance(idx, int):
        return cdf[idx - 1] if idx > 0 else cdf.new_zeros(())
    return torch.where(idx > 0, cdf[(idx - 1).clamp_(min=0)], cdf.new_zeros(()))


def _sample_range(cdf, lo, hi, k, gen):
    """k ids in [lo, hi) drawn by weight; lo / hi are python ints or per-draw int64 tensors."""
    base, top = _cdf_at(cdf, lo), _cdf_at(cdf, hi)
    t = base + torch.rand(k, generator=gen, device=cdf.device, dtype=torch.float64) * (top - base)
    ids = torch.searchsorted(cdf, t)
    if isinstance(lo, int):
        return ids.clamp_(min=lo, max=hi - 1)
    return torch.minimum(torch.maximum(ids, lo), hi - 1)


def pair_quotas(cdf: torch.Tensor, offs: List[int], m_total: int, comm_size: int, mu: float):
    """Exact number of undirected edges of every block pair (a <= b); sums to m_total."""
    world = len(offs) - 1
    tot = float(cdf[-1])
    W = [float(_cdf_at(cdf, offs[g + 1]) - _cdf_at(cdf, offs[g])) / tot for g in range(world)]
    mu_eff = mu if comm_size > 0 else 1.0
    share = {}
    for a in range(world):
        share[(a, a)] = (1.0 - mu_eff) * W[a] + mu_eff * W[a] * W[a]
        for b in range(a + 1, world):
            share[(a, b)] = mu_eff * 2.0 * W[a] * W[b]
    keys = sorted(share)
    raw = [share[k] * m_total for k in keys]
    q = [int(r) for r in raw]
    # largest remainders get the units lost to rounding down: the total is exact
    order = sorted(range(len(keys)), key=lambda i: raw[i] - q[i], reverse=True)
    for i in order[: m_total - sum(q)]:
        q[i] += 1
    return dict(zip(keys, q)), W


def pair_edges(cdf: torch.Tensor, offs: List[int], a: int, b: int, quota: int, seed: int, comm_size: int, mu: float,
               W: List[float]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exactly ``quota`` distinct undirected edges (u in block a, v in block b; u != v) as GLOBAL ids, the same
    tensors on whichever rank asks.  a == b: both endpoints in the block, returned with u < v."""
    dev = cdf.device
    gen = torch.Generator(device=dev)
    gen.manual_seed((seed * 1_000_003 + a * 65_537 + b) % (2 ** 62))
    n = int(cdf.numel())
    lo_a, hi_a, lo_b, hi_b = offs[a], offs[a + 1], offs[b], offs[b + 1]
    if quota <= 0:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return e, e
    cap = (hi_a - lo_a) * (hi_a - lo_a - 1) // 2 if a == b else (hi_a - lo_a) * (hi_b - lo_b)
    if quota > cap:
        raise ValueError(f"block pair ({a},{b}) cannot hold {quota} distinct edges")
    mu_eff = mu if comm_size > 0 else 1.0
    # share of this pair's draws that are intra-community (only the diagonal pair has any)
    p_intra = 0.0
    if a == b and mu_eff < 1.0:
        p_intra = (1.0 - mu_eff) * W[a] / ((1.0 - mu_eff) * W[a] + mu_eff * W[a] * W[a])
    have = torch.empty(0, dtype=torch.int64, device=dev)
    need, rounds = quota, 0
    while have.numel() < quota:
        k = min(max(int(need * 1.08) + 1024, 1024), 1 << 27)
        u = _sample_range(cdf, lo_a, hi_a, k, gen)
        v = _sample_range(cdf, lo_b, hi_b, k, gen)
        if p_intra > 0.0:
            c0 = lo_a + (u - lo_a) // comm_size * comm_size
            c1 = torch.clamp(c0 + comm_size, max=hi_a)
            vi = _sample_range(cdf, c0, c1, k, gen)
            intra = torch.rand(k, generator=gen, device=dev) < p_intra
            v = torch.where(intra, vi, v)
        if a == b:
            keep = u != v
            u, v = u[keep], v[keep]
            u, v = torch.minimum(u, v), torch.maximum(u, v)
        have = torch.unique(torch.cat([have, u * n + v]))
        need = quota - have.numel()
        rounds += 1
        if rounds > 200:
            raise RuntimeError("edge quota not reached: graph too dense for this generator")
    if have.numel() > quota:   # drop the surplus at random (seeded), keep the set sorted
        sel = torch.randperm(have.numel(), generator=gen, device=dev)[:quota]
        have = have[torch.sort(sel).values]
    return have // n, have % n


def partition_edges(n_global: int, nnz: int, skew: float, seed: int, world: int = 1, rank: int = 0, device="cpu",
                    comm_size: int = 0, mu: float = 1.0, add_diag: bool = False, scramble: bool = False,
                    cdf: Optional[torch.Tensor] = None, max_degree: int = 0):
    """The entries of rank ``rank``'s rows of the symmetric global graph: (row LOCAL to the block, col GLOBAL),
    unique, row-major sorted, plus the block offsets.  Summed over the ranks the graph has exactly
    ``2 * (nnz // 2)`` off-diagonal entries (+ n_global diagonal ones with ``add_diag``)."""
    device = torch.device(device)
    offs = block_offsets(n_global, world)
    if cdf is None:
        cdf = node_weight_cdf(n_global, skew, seed, device, max_degree, nnz)
    quotas, W = pair_quotas(cdf, offs, nnz // 2, comm_size, mu)
    lo, hi = offs[rank], offs[rank + 1]
    keys = []
    for h in range(world):
        a, b = min(rank, h), max(rank, h)
        u, v = pair_edges(cdf, offs, a, b, quotas[(a, b)], seed, comm_size, mu, W)
        if a == b:
            keys.append((u - lo) * n_global + v)
            keys.append((v - lo) * n_global + u)
        elif rank == a:
            keys.append((u - lo) * n_global + v)
        else:
            keys.append((v - lo) * n_global + u)
        del u, v
    if add_diag:
        d = torch.arange(hi - lo, device=device, dtype=torch.int64)
        keys.append(d * n_global + (d + lo))
    key = torch.sort(torch.cat(keys)).values          # pairs are disjoint and internally distinct: no duplicates
    del keys
    row, col = key // n_global, key % n_global
    if scramble:
        if world != 1:
            raise ValueError("scramble relabels the whole id range: single-block graphs only")
        gen = torch.Generator(device=device)
        gen.manual_seed(seed + 104_729)
        p = torch.randperm(n_global, generator=gen, device=device)
        key = torch.sort(p[row] * n_global + p[col]).values
        row, col = key // n_global, key % n_global
    return row, col, offs


def powerlaw_edges(n_nodes: int, nnz: int, skew: float, seed: int, device="cpu", symmetric: bool = True,
                   n_cols: Optional[int] = None, add_diag: bool = False, comm_size: int = 0, mu: float = 1.0,
                   scramble: bool = False, max_degree: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Unique (row, col) pairs, row-major sorted, EXACTLY ``nnz`` of them (``2 * (nnz // 2)`` when symmetric,
    + the diagonal with ``add_diag``).  ``symmetric=False`` (optionally rectangular, ``n_cols``) draws directed
    entries: rows and columns by weight, no self-pair rule."""
    device = torch.device(device)
    if symmetric:
        assert n_cols is None or n_cols == n_nodes
        row, col, _ = partition_edges(n_nodes, nnz, skew, seed, 1, 0, device, comm_size, mu, add_diag, scramble,
                                      max_degree=max_degree)
        return row, col
    n_cols = n_nodes if n_cols is None else n_cols
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    cdf_r = node_weight_cdf(n_nodes, skew, seed, device)
    cdf_c = cdf_r if n_cols == n_nodes else node_weight_cdf(n_cols, skew, seed + 1, device)
    if nnz > n_nodes * n_cols:
        raise ValueError("more entries than the shape holds")
    have = torch.empty(0, dtype=torch.int64, device=device)
    while have.numel() < nnz:
        k = min(int((nnz - have.numel()) * 1.08) + 1024, 1 << 27)
        r = _sample_range(cdf_r, 0, n_nodes, k, gen)
        c = _sample_range(cdf_c, 0, n_cols, k, gen)
        have = torch.unique(torch.cat([have, r * n_cols + c]))
    if have.numel() > nnz:
        sel = torch.randperm(have.numel(), generator=gen, device=device)[:nnz]
        have = have[torch.sort(sel).values]
    if add_diag:
        d = torch.arange(min(n_nodes, n_cols), device=device, dtype=torch.int64)
        have = torch.unique(torch.cat([have, d * n_cols + d]))
    return have // n_cols, have % n_cols


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
