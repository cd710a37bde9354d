"""GPU parity of the HIP pair scorer (A9) with (a) the vectors produced by the REFERENCE's own PGE_Edge
(tests/golden/scorer_golden.npz), (b) the oracle's post-processing variants, (c) this package's torch
module run the way the reference's loop runs it.  Tolerance: logits within 2e-5 (abs + rel) of the
reference's float32 forward -- both sides are float32 dot products of length <= 4d + 256 in different
orders; probabilities within 2e-6."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, orc

pytestmark = pytest.mark.gpu
SCORER = np.load(os.path.join(ROOT, "tests", "golden", "scorer_golden.npz"))
DEV = "cuda:0"


def load_model(tag):
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    params = {k[len(tag) + 7:]: torch.from_numpy(SCORER[k]) for k in SCORER.files if k.startswith(f"{tag}_param_")}
    nfeat = params["layers.0.weight"].shape[1] // 2
    nhid = params["layers.0.weight"].shape[0]
    m = PGE_Edge(nfeat=nfeat, nhid=nhid, nlayers=3)
    missing = m.load_state_dict(params, strict=True)      # same state_dict keys as the reference's checkpoints
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_scores_match_reference_vectors(tag):
    m = load_model(tag)
    Z = torch.from_numpy(SCORER[f"{tag}_Z"]).to(DEV)
    L = m.score_all_pairs(Z, logits=True).cpu().numpy()
    S = m.score_all_pairs(Z).cpu().numpy()
    assert np.allclose(L, SCORER[f"{tag}_L"], rtol=2e-5, atol=2e-5)
    assert np.allclose(S, SCORER[f"{tag}_S"], rtol=0, atol=2e-6)
    # sub-blocks address the same pairs
    blk = m.score_block(Z, (3, 11), (5, 24)).cpu().numpy()
    assert np.array_equal(blk, S[3:11, 5:24])


@pytest.mark.parametrize("variant", ["transductive", "inductive", "sampled"])
@pytest.mark.parametrize("thr", [0.5, 0.9, 0.99])
def test_edge_construction_variants(variant, thr):
    m = load_model("a")
    Z = torch.from_numpy(SCORER["a_Z"]).to(DEV)
    ei, ew = m.edge_construction(Z, thr, variant)
    ei, ew = ei.cpu().numpy(), ew.cpu().numpy()
    S = m.score_all_pairs(Z).cpu().numpy()
    rei, rew = orc.edge_construction(S, thr, variant)     # same post-processing on the same scores: exact
    assert np.array_equal(ei, rei) and np.array_equal(ew, rew)
    # against the reference's own scores: identical away from the threshold
    gei, gew = orc.edge_construction(SCORER["a_S"], thr, variant)
    A = (SCORER["a_S"] + SCORER["a_S"].T) / 2
    safe = np.abs(A - thr) > 1e-5
    n = S.shape[0]
    got = np.zeros((n, n), np.float32); got[ei[0], ei[1]] = ew
    ref = np.zeros((n, n), np.float32); ref[gei[0], gei[1]] = gew
    assert np.array_equal((got != 0) & safe, (ref != 0) & safe)
    assert np.allclose(got[safe], ref[safe], atol=2e-6)


def test_against_module_loop_random():
    """The reference's own loop (LargeScaleCondensing.py:304-305) run with this package's torch module on
    the GPU vs. the fused HIP path, n not a multiple of the tile, d = 128 (nfeat = 2d as with aggr)."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    torch.manual_seed(3)
    n, nfeat = 333, 256
    m = PGE_Edge(nfeat=nfeat).to(DEV)
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.3, 1.7)
            bn.weight.normal_(1.0, 0.3)
            bn.bias.normal_(0, 0.2)
    Z = torch.randn(n, nfeat, device=DEV)
    ref = torch.zeros(n, n, device=DEV)
    for i in range(n):
        ref[i] = m.inference(torch.cat([Z[[i] * n], Z[np.arange(n)]], axis=1)).detach()
    got = m.score_all_pairs(Z)
    assert torch.allclose(got, ref, rtol=0, atol=3e-6)
    # float64 oracle on the logits
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    Zc = Z.cpu().numpy()
    rows = [0, 17, 332]
    emb = np.concatenate([np.repeat(Zc[rows], n, axis=0), np.tile(Zc, (len(rows), 1))], axis=1)
    l64 = orc.pge_edge_logits(params, emb, np.float64).reshape(len(rows), n)
    lg = m.score_all_pairs(Z, logits=True)[rows].cpu().numpy()
    assert np.allclose(lg, l64, rtol=2e-5, atol=2e-5)


def test_full_size_c4_properties():
    """BASELINE config 4 at full size (n = 25 000 condensed nodes, d = 128 with aggr -> nfeat 256, H = 256):
    three rows of the 6.25e8-pair score matrix against the reference-style module call and the float64
    oracle, and the structural properties of the edge list (symmetric, unit diagonal, torch.nonzero order)."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    torch.manual_seed(5)
    n, nfeat = 25000, 256
    m = PGE_Edge(nfeat=nfeat).to(DEV).eval()
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.3, 1.7)
    Z = torch.randn(n, nfeat, device=DEV) * 0.5
    S = m.score_all_pairs(Z)
    rows = [0, 12345, n - 1]
    for i in rows:
        ref = m.inference(torch.cat([Z[[i] * n], Z], axis=1)).detach()      # LargeScaleCondensing.py:305
        assert torch.allclose(S[i], ref, rtol=0, atol=3e-6)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    Zc = Z.cpu().numpy()
    cols = np.arange(0, n, 97)
    emb = np.concatenate([np.repeat(Zc[rows], len(cols), axis=0), np.tile(Zc[cols], (len(rows), 1))], axis=1)
    p64 = orc.sigmoid(orc.pge_edge_logits(params, emb, np.float64)).reshape(len(rows), len(cols))
    assert np.allclose(S[rows][:, cols].cpu().numpy(), p64, rtol=0, atol=2e-6)
    thr = float(torch.quantile(S[:2000, :2000].flatten(), 0.999))          # keep ~0.1 % of the pairs
    ei, ew = m.edges_from_scores(S, thr, "transductive")
    E = ei.shape[1]
    assert n <= E < n * n // 100
    key = ei[0] * n + ei[1]
    assert bool((key[1:] > key[:-1]).all())                                # row-major, strictly increasing
    tkey, order = torch.sort(ei[1] * n + ei[0])
    assert torch.equal(tkey, key) and torch.equal(ew[order], ew)           # symmetric pattern and weights
    diag = ei[0] == ei[1]
    assert int(diag.sum()) == n and bool((ew[diag] == 1).all())
    off = ~diag
    a = (S[ei[0][off], ei[1][off]] + S[ei[1][off], ei[0][off]]) / 2.0
    assert torch.equal(ew[off], a) and bool((a >= thr).all())


@pytest.mark.parametrize("n,nfeat,nhid", [(333, 12, 64), (1000, 20, 256), (64, 7, 32), (130, 5, 128)])
@pytest.mark.parametrize("variant", ["transductive", "inductive", "sampled"])
def test_fused_edge_construction_equals_dense_path(n, nfeat, nhid, variant):
    """disco_pairscore_edges_f32 (scores both directions of a pair, symmetrises, thresholds and emits edges without
    the n x n matrix) must give the dense path's edge list bit for bit, in torch.nonzero order -- at several
    thresholds (incl. 1.5: an inductive graph loses its diagonal), with n not a multiple of the 64-node tile, and when
    the first pass overflows its edge buffer."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    torch.manual_seed(n + nhid)
    m = PGE_Edge(nfeat=nfeat, nhid=nhid).to(DEV).eval()
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 1.5)
        m.layers[-1].weight.mul_(4.0)
    Z = torch.randn(n, nfeat, device=DEV)
    for thr in (0.5, 0.9, 1.5, 0.0):
        ei_d, ew_d = m.edge_construction(Z, thr, variant, fused=False)
        ei_f, ew_f = m.edge_construction(Z, thr, variant)
        assert torch.equal(ei_d, ei_f) and torch.equal(ew_d, ew_f), (thr, ei_d.shape, ei_f.shape)
    ei_s, ew_s = m.edge_construction(Z, 0.5, variant, capacity=10)         # overflow -> second pass
    ei_d, ew_d = m.edge_construction(Z, 0.5, variant, fused=False)
    assert ei_d.shape[1] > 10 and torch.equal(ei_d, ei_s) and torch.equal(ew_d, ew_s)
