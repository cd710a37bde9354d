"""Host-side logic on CPU: the CSR container, the synthetic generators and the no-fallback rule."""
import numpy as np
import pytest
import torch

import json
import os
import subprocess
import sys
import time

from helpers import ROOT, orc
from disco_amd import SparseTensor, graphgen, ops
from disco_amd.models.convs_large import GCNConv, SGConv
from disco_amd.models.convs_large.message_passing import _EdgeListCache


def test_sparse_tensor_matches_oracle_layout():
    rng = np.random.default_rng(0)
    n, E = 50, 400
    row, col = rng.integers(0, n, E), rng.integers(0, n, E)
    val = rng.random(E).astype(np.float32)
    st = SparseTensor(row=torch.from_numpy(row), col=torch.from_numpy(col), value=torch.from_numpy(val), sparse_sizes=(n, n))
    rp, c, v = orc.coo_to_csr(row, col, val, n)
    assert np.array_equal(st.storage.rowptr().numpy(), rp)
    assert np.array_equal(st.storage.col().numpy(), c)
    assert np.array_equal(st.storage.value().numpy(), v)
    t = st.t()
    rp2, c2, v2 = orc.adj_t_from_coo(row, col, val, n)
    assert np.array_equal(t.csr()[0].numpy(), rp2) and np.array_equal(t.csr()[1].numpy(), c2)
    assert np.array_equal(np.sort(t.csr()[2].numpy()), np.sort(v2))
    assert t.t() is st and st.size(0) == n and st.nnz() == E and st.sparse_sizes() == (n, n)
    r3, c3, v3 = st.coo()
    assert np.array_equal(r3.numpy() * n + c3.numpy(), np.sort(row * n + col))


def test_sparse_tensor_rejects_bad_indices():
    with pytest.raises(ValueError):
        SparseTensor(row=torch.tensor([0, 1]), col=torch.tensor([0, 9]), sparse_sizes=(2, 3))
    with pytest.raises(ValueError):
        SparseTensor(rowptr=torch.tensor([0, 1, 5]), col=torch.tensor([0, 1]), sparse_sizes=(2, 3))


def test_edge_list_cache_is_stable_in_edge_order():
    ei = torch.tensor([[3, 1, 2, 0, 1], [0, 0, 1, 0, 1]])
    struct, perm = _EdgeListCache().get(ei, 4, 4)
    assert struct.csr()[0].tolist() == [0, 3, 5, 5, 5]
    assert struct.csr()[1].tolist() == [3, 1, 0, 2, 1]      # sources in original edge order per target
    assert perm.tolist() == [0, 1, 3, 2, 4]


def test_generator_is_deterministic_and_symmetric():
    r1, c1 = graphgen.powerlaw_edges(2000, 30000, 0.8, seed=5)
    r2, c2 = graphgen.powerlaw_edges(2000, 30000, 0.8, seed=5)
    assert torch.equal(r1, r2) and torch.equal(c1, c2)
    k = set((r1 * 2000 + c1).tolist())
    assert all((c * 2000 + r) in k for r, c in zip(r1.tolist()[:500], c1.tolist()[:500]))
    assert not bool((r1 == c1).any())
    assert len(k) == r1.numel()                             # no duplicates
    deg = torch.bincount(r1, minlength=2000)
    assert deg.max() > 8 * deg.float().mean()               # skewed
    r3, c3 = graphgen.powerlaw_edges(500, 4000, 0.0, seed=1, symmetric=False, n_cols=4000)
    assert int(c3.max()) < 4000 and int(r3.max()) < 500 and r3.numel() == 4000
    assert r1.numel() == 30000                              # exactly the entry count asked for, not what survived de-duplication


def test_partitioned_generator_gives_one_consistent_global_graph():
    """Every rank generates only its own rows; the union over the ranks is one symmetric graph with exactly the
    requested number of entries, whatever the world size; the community variant keeps most entries inside a
    block and scrambling hides that without changing the counts."""
    n, nnz = 5000, 60000
    for world in (1, 2, 3, 4):
        keys = []
        for rank in range(world):
            row, col, offs = graphgen.partition_edges(n, nnz, 0.8, 11, world, rank, comm_size=256, mu=0.3, add_diag=True)
            assert offs == graphgen.block_offsets(n, world) and int(row.max()) < offs[rank + 1] - offs[rank]
            assert bool((row[1:] * n + col[1:] > row[:-1] * n + col[:-1]).all())        # unique, row-major sorted
            keys.append((row + offs[rank]) * n + col)
        key = torch.cat(keys)
        rr, cc = key // n, key % n
        assert key.numel() == nnz + n and int((rr != cc).sum()) == nnz
        s = set(key.tolist())
        assert all((int(c) * n + int(r)) in s for r, c in zip(rr[::7], cc[::7]))
        if world > 1:
            same = ((rr * world // n) == (cc * world // n)).float().mean()
            assert float(same) > 0.7                        # mu = 0.3: at most ~30 % of the draws leave the block
    row, col, _ = graphgen.partition_edges(n, nnz, 0.8, 11, comm_size=256, mu=0.3)
    rs, cs, _ = graphgen.partition_edges(n, nnz, 0.8, 11, comm_size=256, mu=0.3, scramble=True)
    assert rs.numel() == row.numel() == nnz
    assert float(((row // 256) == (col // 256)).float().mean()) > 0.5 > 0.1 > float(((rs // 256) == (cs // 256)).float().mean())
    assert torch.equal(torch.bincount(row, minlength=n).sort().values, torch.bincount(rs, minlength=n).sort().values)


def test_no_cpu_fallback():
    st = SparseTensor(row=torch.tensor([0, 1]), col=torch.tensor([1, 0]), sparse_sizes=(2, 2))
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.spmm(st, torch.ones(2, 4))
    with pytest.raises(RuntimeError, match="no CPU"):
        SGConv(4, 3, K=2)(torch.ones(2, 4), st)


def test_module_state_dict_keys_match_reference():
    # convs.{i}.lin.weight, convs.{i}.bias (GCN) / lin.{weight,bias} (SGC)
    assert sorted(GCNConv(8, 4).state_dict()) == ["bias", "lin.weight"]
    assert sorted(SGConv(8, 4, K=2).state_dict()) == ["lin.bias", "lin.weight"]
    assert SGConv(8, 4).K == 3


def test_negative_sampling_host_logic():
    """negative_sampling is index bookkeeping in stock torch: no returned pair is an edge or a self pair, pairs are
    distinct, and an (almost) complete graph yields fewer samples than asked for instead of looping forever."""
    import numpy as np
    import torch
    from disco_amd.link_prediction import negative_sampling
    n = 60
    rng = np.random.default_rng(0)
    ei = torch.from_numpy(rng.integers(0, n, (2, 400)))
    neg = negative_sampling(ei, n, 900, torch.Generator().manual_seed(3)).numpy()
    assert neg.shape == (2, 900)
    pos = set((ei[0] * n + ei[1]).tolist())
    keys = neg[0] * n + neg[1]
    assert not (set(keys.tolist()) & pos) and len(set(keys.tolist())) == 900 and np.all(neg[0] != neg[1])
    full = torch.tensor([[i, j] for i in range(6) for j in range(6) if i != j and (i, j) != (0, 1)]).T
    few = negative_sampling(full, 6, 10, torch.Generator().manual_seed(1))
    assert few.shape[1] <= 1 and all((int(a), int(b)) == (0, 1) for a, b in few.T)


def test_add_self_loops_and_dense_normalisation_contract():
    """add_self_loops appends n loops after the edges (reference interface, utils.py:457-471); the dense branch of
    normalize_adj_tensor equals diag(r) (A + I) diag(r) with r = rowsum^-1/2 and inf -> 0 (utils.py:486-493)."""
    from disco_amd import utils
    ei = torch.tensor([[0, 2, 2], [1, 2, 0]])
    w = torch.tensor([0.5, 2.0, 3.0])
    ei2, w2 = utils.add_self_loops(ei, w, fill_value=7, num_nodes=4)
    assert ei2.tolist() == [[0, 2, 2, 0, 1, 2, 3], [1, 2, 0, 0, 1, 2, 3]] and w2.tolist() == [0.5, 2.0, 3.0, 7, 7, 7, 7]
    ei3, w3 = utils.add_self_loops(ei, None, num_nodes=3)
    assert w3 is None and ei3.shape == (2, 6)
    with pytest.raises(ValueError):
        utils.add_self_loops(ei, w[:2], num_nodes=3)
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(9, 9, generator=g) < 0.3).float()
    a[4] = 0
    a[4, 4] = -1.0                                    # row sum of A + I is 0 -> r = inf -> 0
    got = utils.normalize_adj_tensor(a)
    mx = a + torch.eye(9)
    r = mx.sum(1).pow(-0.5)
    r[torch.isinf(r)] = 0
    assert torch.equal(got, torch.diag(r) @ mx @ torch.diag(r)) and float(got[4].abs().sum()) == 0.0


def test_pge_edge_shell_contract():
    """Constructor / state_dict contract of the scorer's torch shell and that `inference` is the
    eval-mode sigmoid of the logits the oracle restates."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    m = PGE_Edge(nfeat=5, nhid=16, nlayers=3)
    keys = sorted(m.state_dict())
    assert keys == sorted([f"layers.{i}.{p}" for i in range(3) for p in ("weight", "bias")] +
                          [f"bns.{i}.{p}" for i in range(2) for p in ("weight", "bias", "running_mean", "running_var",
                                                                      "num_batches_tracked")])
    assert m.layers[0].in_features == 10 and m.layers[2].out_features == 1
    assert len(PGE_Edge(4, nlayers=4).layers) == 4 and len(PGE_Edge(4, nlayers=4).bns) == 3
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.normal_()
            bn.running_var.uniform_(0.5, 2.0)
    e = torch.randn(33, 10)
    p = m.inference(e)
    assert not m.training and p.shape == (33,)
    params = {k: v.numpy() for k, v in m.state_dict().items()}
    ref = orc.sigmoid(orc.pge_edge_logits(params, e.numpy(), np.float64))
    assert np.allclose(p.numpy(), ref, atol=2e-6)


def test_feature_slabs_cover_the_columns_in_vector_units():
    from disco_amd.dist import feature_slabs
    for F, world in ((602, 8), (128, 8), (100, 8), (100, 3), (13, 3), (5, 4), (128, 1)):
        slabs = feature_slabs(F, world)
        assert len(slabs) == world and slabs[0][0] == 0 and slabs[-1][1] == F
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
        assert all(c0 % 4 == 0 and c1 >= c0 for c0, c1 in slabs)
        widths = [c1 - c0 for c0, c1 in slabs]
        assert max(widths) - min(widths) <= 4 + (F % 4 != 0) * 3
    assert feature_slabs(128, 8) == [(16 * g, 16 * g + 16) for g in range(8)]


def test_bench_refuses_to_run_fewer_gpus_than_asked():
    """`bench.py --gpus N` without a launcher starts its own ranks -- or exits non-zero BEFORE any GPU call when fewer
    GPUs are visible; it never silently measures one GPU (round-1 verdict)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DISCO_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "only 0 GPU" in p.stderr and p.stdout.strip() == ""


def test_bench_byte_models_and_graph_sizes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # SURVEY 8d, Reddit: B_min = nnz*8 + (N+1)*4 + 2*N*F*4
    assert bench.b_min_bytes(114_848_857, 232_965, 602) == 114_848_857 * 8 + 232_966 * 4 + 2 * 232_965 * 602 * 4

    class A:
        scale = 1.0
    s, n, nnz, scaling = bench.graph_params("reddit", A, 8)
    assert (n, nnz, scaling) == (8 * 232_965, 8 * 114_615_892, "weak")
    s, n, nnz, scaling = bench.graph_params("products", A, 4)
    assert (n, nnz, scaling) == (2_449_029, 61_859_140, "strong")
    s, n, nnz, scaling = bench.graph_params("papers100M", A, 1)
    assert (n, nnz) == (111_059_956, 1_615_685_872)


def test_bench_line_stays_short(capsys, tmp_path, monkeypatch):
    """Whoever captures bench.py's stdout may keep only its tail: the ONE line is compact (extras in a few dozen bytes
    each), the full record goes to stderr and a file."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = json.loads(open(os.path.join(ROOT, "profiles", "r2", "bench_default_n1.json")).read().strip().splitlines()[-1])
    extras = full.pop("extra_workloads")
    extras = extras * 4 + [{"workload": "x", "failed": "RuntimeError: " + "y" * 500}, {"workload": "z", "skipped": "time budget"}]
    bench.emit(full, extras, 8)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4200, len(lines[0])
    line = json.loads(lines[0])
    assert line["value"] == full["value"] and line["roofline"] == full["roofline"] and len(line["extra_workloads"]) == len(extras)
    assert all(len(json.dumps(e)) < 260 for e in line["extra_workloads"])
    assert cap.err.startswith("BENCH_DETAIL ") and json.loads(cap.err[len("BENCH_DETAIL "):])["extra_workloads"][0] == extras[0]
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail_n8.json"))["value"] == full["value"]


def test_bench_watchdog_prints_the_line_and_exits_nonzero(tmp_path):
    """A hang (an extra workload that never returns: hung collective, dead rank) must not end with status 0: the
    watchdog prints the ONE line with what has been measured, marked "aborted", and the process leaves with
    bench.WATCHDOG_EXIT_CODE."""
    script = tmp_path / "hang.py"
    script.write_text(
        "import importlib.util, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod3', {os.path.join(ROOT, 'bench.py')!r})\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        f"bench.ROOT = {str(tmp_path)!r}\n"
        "out = {'metric': 'K-hop SpMM edges*feat/s', 'value': 1.0e12, 'n_gpus': 1, 'ms_per_step': 38.0}\n"
        "extras = [{'workload': 'products', 'ms_per_step': 8.1, 'value': 1.5e12, 'roofline': {'frac': 0.07}}]\n"
        "bench.arm_watchdog(0.3, lambda: bench.emit(dict(out, aborted=True), extras + [{'aborted': 'overran'}], 1))\n"
        "time.sleep(60)          # the fake extra that sleeps past the budget\n"
        "print('never reached')\n")
    t0 = time.time()
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=50)
    assert time.time() - t0 < 40
    assert p.returncode == 3, p.stderr[-400:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "never reached" not in p.stdout
    line = json.loads(lines[0])
    assert line["aborted"] is True and line["value"] == 1.0e12
    assert any("aborted" in e for e in line["extra_workloads"])


def test_anchor_max_has_no_cpu_fallback_and_validates_before_launch():
    """A8 entry point: CPU tensors raise (no fallback); the C ABI rejects bad arguments before touching the device."""
    from disco_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.anchor_max(torch.zeros(4, 8), torch.zeros(2, 3, dtype=torch.long))
    lib = _lib.load()
    # width beyond the row pitch, k < 1, negative sizes; empty work is fine
    assert lib.disco_anchor_max_f32(None, 10, 4, 8, None, 0, None, 5, 3, None, 5, None, 8, None, None) == -1
    assert lib.disco_anchor_max_f32(None, 10, 8, 8, None, 0, None, 5, 0, None, 5, None, 8, None, None) == -1
    assert lib.disco_anchor_max_f32(None, -1, 8, 8, None, 0, None, 5, 3, None, 5, None, 8, None, None) == -1
    assert lib.disco_anchor_max_f32(None, 10, 8, 8, None, 0, None, 0, 3, None, 0, None, 8, None, None) == 0
    assert lib.disco_anchor_max_f32(None, 10, 8, 8, None, 0, None, 5, 3, None, 5, None, 8, None, None) == -1   # null pointers


def test_truncated_head_generator_caps_the_expected_degree():
    """graphgen max_degree (round-2 verdict item 7): same node and entry counts, no node far above the cap."""
    n, nnz = 60_000, 1_500_000
    r0, _ = graphgen.powerlaw_edges(n, nnz, 0.9, 3)
    r1, c1 = graphgen.powerlaw_edges(n, nnz, 0.9, 3, max_degree=600)
    assert r0.numel() == r1.numel() == nnz
    d0, d1 = torch.bincount(r0, minlength=n), torch.bincount(r1, minlength=n)
    assert int(d0.max()) > 5000 and int(d1.max()) < 1.25 * 600
    key = r1 * n + c1
    assert bool((key[1:] > key[:-1]).all())                       # still unique and row-major sorted
    rt, ct, offs = graphgen.partition_edges(n, nnz, 0.9, 3, world=2, rank=1, max_degree=600)
    assert int(torch.bincount(rt).max()) < 1.25 * 600


def test_grid_helpers_never_hand_out_a_dimension_that_wraps(tmp_path):
    """disco_internal.h: grid_for() must cover every workgroup with each dimension below 2^32 work-items (a 1-D launch of
    2^32 work-items silently runs only part of them, profiles/r3/r3_gridlimit.log), keep x a multiple of 8 once it is 2-D,
    and grid_1d() must refuse exactly what would wrap.  Host-only program, compiled here with hipcc (no GPU needed)."""
    import shutil
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not found")
    src = tmp_path / "grids.hip"
    src.write_text('''
#include "disco_internal.h"
#include <cstdio>
int main() {
  const int64_t cases[] = {0, 1, 7, 4194304, 4194305, 16777215, 16777216, 17000000, 27764989, 100000000, 2147483647LL,
                           274877906944LL, 70368744177664LL};
  for (int threads : {64, 256, 512, 1024}) for (int64_t b : cases) {
    dim3 g, h;
    const bool ok = disco::grid_for(b, threads, &g), ok1 = disco::grid_1d(b, threads, &h);
    printf("%d %lld %d %u %u %d %u\\n", threads, (long long)b, (int)ok, ok ? g.x : 0u, ok ? g.y : 0u, (int)ok1, ok1 ? h.x : 0u);
  }
  return 0;
}
''')
    exe = tmp_path / "grids"
    subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-w", f"-I{os.path.join(ROOT, 'disco_amd', 'csrc')}", "-o", str(exe), str(src)],
                   check=True, capture_output=True, timeout=300)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=60).stdout.split("\n")
    seen = 0
    for ln in filter(None, lines):
        threads, b, ok, x, y, ok1, x1 = (int(v) for v in ln.split())
        need = max(b, 1)
        if ok:
            assert x * y >= need and x * threads <= 1 << 30 and 1 <= y <= 65535, ln
            assert (x - 1) * y < need + 8 * y, ln                        # no more than the rounding of x to a multiple of 8 is idle
            assert y == 1 or x % 8 == 0, ln
        else:
            assert need > (((1 << 30) // threads) // 8 * 8) * 65535, ln    # refused only when 65535 rows of blocks cannot hold it
        assert bool(ok1) == (need * threads < 1 << 32), ln
        assert not ok1 or x1 == need
        seen += 1
    assert seen == 4 * 13


def test_partition_helpers_cover_their_range_exactly():
    """Property sweep (hypothesis) of the pure partition helpers of disco_amd.dist / graphgen / sparse: every cut list
    starts at 0, ends at the size, never decreases; slabs and chunks respect their alignment; the nnz-balanced row cuts
    never give a rank more than its share plus one row's entries."""
    from hypothesis import given, settings, strategies as st
    from disco_amd import dist as dd
    from disco_amd.sparse import sweep_window_rows

    @settings(max_examples=300, deadline=None)
    @given(F=st.integers(1, 5000), world=st.integers(1, 64))
    def slabs(F, world):
        s = dd.feature_slabs(F, world)
        assert len(s) == world and s[0][0] == 0 and s[-1][1] == F
        for (a0, a1), (b0, b1) in zip(s, s[1:]):
            assert a1 == b0
        for c0, c1 in s:
            assert 0 <= c0 <= c1 <= F and c0 % 4 == 0 and (c1 % 4 == 0 or c1 == F)
        widths = [c1 - c0 for c0, c1 in s]
        assert max(widths) - min(widths) <= 4 + 3                  # as equal as 4-float units allow (the last unit may be partial)

    @settings(max_examples=300, deadline=None)
    @given(F=st.integers(1, 5000))
    def chunks(F):
        c = dd.RowPartition._col_chunks(F)
        assert c[0][0] == 0 and c[-1][1] == F and all(a[1] == b[0] for a, b in zip(c, c[1:]))
        assert all(c0 % 32 == 0 and 0 < c1 - c0 <= 160 for c0, c1 in c)

    @settings(max_examples=200, deadline=None)
    @given(n=st.integers(0, 10 ** 9), world=st.integers(1, 64))
    def blocks(n, world):
        o = graphgen.block_offsets(n, world)
        assert o[0] == 0 and o[-1] == n and all(a <= b for a, b in zip(o, o[1:])) and max(b - a for a, b in zip(o, o[1:])) - min(b - a for a, b in zip(o, o[1:])) <= 1

    @settings(max_examples=200, deadline=None)
    @given(deg=st.lists(st.integers(0, 50), min_size=0, max_size=200), world=st.integers(1, 9))
    def balanced(deg, world):
        rowptr = torch.zeros(len(deg) + 1, dtype=torch.int64)
        if deg:
            rowptr[1:] = torch.tensor(deg).cumsum(0)
        o = dd.balanced_row_offsets(rowptr, world)
        assert len(o) == world + 1 and o[0] == 0 and o[-1] == len(deg) and all(a <= b for a, b in zip(o, o[1:]))
        nnz = int(rowptr[-1])
        for g in range(world):
            share = int(rowptr[o[g + 1]] - rowptr[o[g]])
            assert share <= nnz // world + 1 + 2 * (max(deg) if deg else 0)

    @settings(max_examples=200, deadline=None)
    @given(F=st.integers(1, 4000), wb=st.integers(1 << 10, 1 << 26))
    def window(F, wb):
        r = sweep_window_rows(F, wb)
        assert r >= 64
        blk = min(128, (F + 31) // 32 * 32)
        assert r == 64 or r * 4 * 32 <= wb                           # a window of the narrowest block fits the budget

    slabs(); chunks(); blocks(); balanced(); window()


def test_sparse_tensor_host_construction_properties():
    """Property sweep of the host-side SparseTensor (the reference builds its adjacency from host tensors,
    LargeScaleCondensing.py:486): row-major order, duplicates kept in input order (torch_sparse's constructor), transpose
    stable in the original entry order, dense forms equal to scipy's."""
    import scipy.sparse as sp
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(m=st.integers(1, 12), n=st.integers(1, 12), data=st.data())
    def check(m, n, data):
        e = data.draw(st.integers(0, 40))
        row = np.array(data.draw(st.lists(st.integers(0, m - 1), min_size=e, max_size=e)), dtype=np.int64)
        col = np.array(data.draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e)), dtype=np.int64)
        val = np.arange(1, e + 1, dtype=np.float32)                 # distinct: the order of duplicates is visible
        a = SparseTensor(row=torch.from_numpy(row), col=torch.from_numpy(col), value=torch.from_numpy(val), sparse_sizes=(m, n))
        rp, c, v = a.csr()
        assert rp.tolist() == np.concatenate([[0], np.cumsum(np.bincount(row, minlength=m))]).tolist()
        order = np.lexsort((np.arange(e), col, row))                  # by row, then column, then input position
        assert c.tolist() == col[order].tolist() and v.tolist() == val[order].tolist()
        dense = sp.coo_matrix((val, (row, col)), shape=(m, n)).toarray()
        assert np.array_equal(a.to_dense().numpy(), dense)
        t = a.t()
        assert t.sparse_sizes() == (n, m) and np.array_equal(t.to_dense().numpy(), dense.T)
        rt, ct, vt = t.csr()
        order_t = np.lexsort((np.arange(e), row[order], col[order]))  # stable in the CSR entry order
        assert ct.tolist() == row[order][order_t].tolist() and vt.tolist() == val[order][order_t].tolist()
        assert t.t() is a
        r2, c2, v2 = a.coo()
        assert r2.tolist() == row[order].tolist() and c2.tolist() == col[order].tolist()

    check()


def test_sparse_tensors_are_freed_by_reference_counting_alone():
    """No reference cycles: dropping the last name of a SparseTensor (or of the untransposed half of `SparseTensor(...).t()`,
    the reference's idiom) frees it at once, with the cycle collector switched off -- on the GPU that is 13.8 GB per copy at
    papers100M size."""
    import gc
    import weakref
    gc.collect()
    gc.disable()
    try:
        a = SparseTensor(row=torch.tensor([0, 1, 2, 2]), col=torch.tensor([1, 0, 0, 2]), value=torch.arange(4.0), sparse_sizes=(3, 3))
        wa = weakref.ref(a)
        t = a.t()
        assert t.t() is a and a.t() is t and a.storage.value() is a._value
        b = a.set_value(torch.ones(4))
        wb = weakref.ref(b)
        del b
        assert wb() is None, "a set_value copy survived its last reference"
        del a
        assert wa() is None, "the untransposed matrix survived `SparseTensor(...).t()`"
        assert t.t().t() is t                                   # the origin is gone: transposing again builds a new pair
        assert torch.equal(t.t().to_dense(), t.to_dense().t())
        wt = weakref.ref(t)
        del t
        assert wt() is None
    finally:
        gc.enable()
