"""Link-predictor pre-training on the device: the batch gather is bit exact against numpy
indexing, and the whole loop reproduces a reference-style loop (the indexing / concat / host round trip of
/root/reference/LargeScaleCondensing.py:188-200 written with stock torch) from the same seeds."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("w,ld", [(256, 256), (6, 6), (7, 7), (602, 608), (1204, 1204)])
def test_pair_gather_bit_exact(w, ld):
    from disco_amd import ops
    rng = np.random.default_rng(w)
    n, e, p = 500, 4000, 1777
    z = rng.standard_normal((n, ld)).astype(np.float32)
    ei = rng.integers(0, n, (2, e))
    sel = rng.integers(0, e, p)
    zd = torch.from_numpy(z).to(dev())[:, :w]
    got = ops.pair_gather(zd, torch.from_numpy(ei).to(dev()), torch.from_numpy(sel).to(dev())).cpu().numpy()
    ref = np.concatenate([z[ei[0][sel], :w], z[ei[1][sel], :w]], axis=1)
    assert np.array_equal(got, ref)
    got_all = ops.pair_gather(zd, torch.from_numpy(ei).to(dev())).cpu().numpy()
    assert np.array_equal(got_all, np.concatenate([z[ei[0], :w], z[ei[1], :w]], axis=1))
    bad = ei.copy()
    bad[1, sel[5]] = n
    with pytest.raises(IndexError):
        ops.pair_gather(zd, torch.from_numpy(bad).to(dev()), torch.from_numpy(sel).to(dev()))
    with pytest.raises(IndexError):
        ops.pair_gather(zd, torch.from_numpy(ei).to(dev()), torch.tensor([e], device=dev()))
    assert ops.pair_gather(zd, torch.from_numpy(ei).to(dev()), torch.empty(0, dtype=torch.int64, device=dev())).shape == (0, 2 * w)


def test_negative_sampling_properties():
    from disco_amd.link_prediction import negative_sampling
    n = 300
    rng = np.random.default_rng(3)
    ei = rng.integers(0, n, (2, 5000))
    g = torch.Generator(device=dev()).manual_seed(7)
    neg = negative_sampling(torch.from_numpy(ei).to(dev()), n, 15000, g).cpu().numpy()
    assert neg.shape == (2, 15000) and neg.min() >= 0 and neg.max() < n
    pos = set((ei[0] * n + ei[1]).tolist())
    keys = neg[0] * n + neg[1]
    assert not (set(keys.tolist()) & pos)
    assert np.unique(keys).size == keys.size and np.all(neg[0] != neg[1])


def _reference_style_loop(pge, feat_transform, edge_index, neg_edge_index, nedges, epochs, lr, g):
    """LargeScaleCondensing.py:187-200 with stock torch, including the .cpu()/.cuda() round trip."""
    opt = torch.optim.Adam(pge.parameters(), lr=lr)
    pge.train()
    losses = []
    for _ in range(epochs):
        index = torch.randint(0, edge_index.shape[1], (nedges,), device=dev(), generator=g)
        neg_index = torch.randint(0, neg_edge_index.shape[1], (3 * nedges,), device=dev(), generator=g)
        pos = torch.concat((feat_transform[edge_index[0][index]], feat_transform[edge_index[1][index]]), dim=1).cpu()
        neg = torch.concat((feat_transform[neg_edge_index[0][neg_index]], feat_transform[neg_edge_index[1][neg_index]]), dim=1).cpu()
        batch = torch.concat((pos, neg), dim=0).to(dev())
        y = torch.concat((torch.ones(nedges), torch.zeros(3 * nedges))).to(dev())
        opt.zero_grad()
        pred = pge.forward(batch)
        loss = torch.nn.BCELoss()(pred, y.float())
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


def test_link_prediction_reproduces_reference_style_loop():
    from disco_amd import SparseTensor
    from disco_amd.link_prediction import edge_translation_features, link_prediction, negative_sampling
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    torch.manual_seed(0)
    n, d, nedges, epochs = 600, 16, 256, 30
    rng = np.random.default_rng(5)
    comm = rng.integers(0, 6, n)
    feat = (np.eye(6, d)[comm] * 3 + rng.standard_normal((n, d)) * 0.3).astype(np.float32)
    src = rng.integers(0, n, 6000)
    dst = np.array([rng.choice(np.flatnonzero(comm == comm[s])) for s in src])      # edges inside communities
    ei = torch.from_numpy(np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])])).to(dev())
    x = torch.from_numpy(feat).to(dev())
    adj_t = SparseTensor(row=ei[0], col=ei[1], value=None, sparse_sizes=(n, n)).t()
    ft = edge_translation_features(adj_t, x)                                            # [n, 2d]
    assert ft.shape == (n, 2 * d) and torch.equal(ft[:, :d], x)
    neg = negative_sampling(ei, n, 3 * ei.shape[1], torch.Generator(device=dev()).manual_seed(1))
    a = PGE_Edge(nfeat=2 * d, nhid=32, device=dev()).to(dev())
    b = copy.deepcopy(a)
    ga = torch.Generator(device=dev()).manual_seed(11)
    gb = torch.Generator(device=dev()).manual_seed(11)
    ref_losses = _reference_style_loop(b, ft, ei, neg, nedges, epochs, 0.01, gb)
    best_acc, last = link_prediction(a, ft, ei, nedges, epochs, lr=0.01, neg_edge_index=neg, generator=ga, log_every=1)
    assert abs(last - ref_losses[-1]) <= 1e-5 * max(1.0, abs(ref_losses[-1])), (last, ref_losses[-1])
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6)
    assert ref_losses[-1] < 0.7 * ref_losses[0] and best_acc > 0.8
