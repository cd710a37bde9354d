"""A8: anchor neighbour features of edge_construction (/root/reference/LargeScaleCondensing.py:293-300,
_induct.py:284-294, _Sampled.py:252-262) -- ``feat_transform[index[c]][anchor, :L*d].max(dim=1).values`` as one HIP
kernel (disco_anchor_max_f32), bit exact against the oracle's numpy restatement (a maximum has no rounding)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _case(n, cols, n_c, k, seed, repeats=True):
    rng = np.random.default_rng(seed)
    T = rng.standard_normal((n, cols)).astype(np.float32)
    T[rng.random((n, cols)) < 0.05] = 0.0
    anchor = rng.integers(0, n, (n_c, k), dtype=np.int64)
    if repeats and n_c > 2:
        anchor[0, :] = anchor[0, 0]            # the same anchor three times
        if k > 1:
            anchor[1, 1] = anchor[1, 0]
    return T, anchor


@pytest.mark.parametrize("cols,width", [(256, 256), (256, 128), (602, 301), (100, 100), (7, 5), (1204, 602)])
@pytest.mark.parametrize("k", [3, 1, 5])
def test_anchor_max_matches_oracle_bit_for_bit(cols, width, k):
    from disco_amd import ops
    T, anchor = _case(900, cols, 411, k, seed=cols + k)
    got = ops.anchor_max(torch.from_numpy(T).to(DEV), torch.from_numpy(anchor).to(DEV), width)
    want = orc.anchor_max(T, anchor, width)
    assert got.shape == want.shape
    assert np.array_equal(got.cpu().numpy().view(np.int32), want.view(np.int32))


def test_anchor_max_with_class_subset_and_scatter_rows():
    """The reference's loop over classes: rows index[c] of feat_transform, anchors as positions inside the class,
    results written to feat_syn_neighbor[index_syn[c]]; classes without synthetic nodes keep zeros."""
    from disco_amd import link_prediction as lp
    rng = np.random.default_rng(5)
    n_train, d, L, nclass, neighbor = 3000, 64, 2, 6, 3
    feat_transform = rng.standard_normal((n_train, L * d + 32)).astype(np.float32)     # wider than the cut
    labels = rng.integers(0, nclass, n_train)
    index = {c: np.flatnonzero(labels == c) for c in range(nclass)}
    n_syn = 240
    labels_syn = np.sort(rng.integers(0, nclass - 1, n_syn))                            # the last class has no synthetic node
    index_syn = {c: np.flatnonzero(labels_syn == c) for c in range(nclass - 1)}
    anchors = {c: rng.integers(0, len(index[c]), (len(index_syn[c]), neighbor)) for c in index_syn}
    feat_syn = rng.standard_normal((n_syn, d)).astype(np.float32)
    got = lp.anchor_neighbor_features(torch.from_numpy(feat_syn).to(DEV), torch.from_numpy(feat_transform).to(DEV),
                                      anchors, index, index_syn, width=L * d).cpu().numpy()
    want = np.zeros((n_syn, L * d), np.float32)
    for c in index_syn:                                    # LargeScaleCondensing.py:296-299
        want[index_syn[c]] = orc.anchor_max(feat_transform, anchors[c], L * d, index=index[c])
    assert np.array_equal(got[:, :d], feat_syn)
    assert np.array_equal(got[:, d:].view(np.int32), want.view(np.int32))


def test_anchor_max_nan_propagates_and_bad_index_raises():
    from disco_amd import ops
    T, anchor = _case(50, 8, 20, 3, seed=1, repeats=False)
    T[anchor[4, 1], 2] = np.nan
    got = ops.anchor_max(torch.from_numpy(T).to(DEV), torch.from_numpy(anchor).to(DEV)).cpu().numpy()
    want = torch.from_numpy(T)[torch.from_numpy(anchor)].max(dim=1).values.numpy()      # torch.max(dim) itself
    assert np.isnan(got[4, 2]) and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    anchor[7, 0] = 50
    with pytest.raises(IndexError):
        ops.anchor_max(torch.from_numpy(T).to(DEV), torch.from_numpy(anchor).to(DEV))


def test_anchor_max_through_the_raw_c_abi_leaves_other_rows_alone():
    from disco_amd import _lib
    lib = _lib.load()
    T, anchor = _case(300, 128, 40, 3, seed=9)
    t = torch.from_numpy(T).to(DEV)
    a = torch.from_numpy(anchor).to(DEV)
    rows = torch.arange(0, 80, 2, device=DEV)                  # every second row of an 80-row output
    out = torch.full((80, 160), 7.0, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = lib.disco_anchor_max_f32(t.data_ptr(), 300, 128, 96, None, 0, a.data_ptr(), 40, 3, rows.data_ptr(), 80,
                                  out.data_ptr(), 160, err.data_ptr(), _lib.current_stream(DEV))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert int(err.item()) == 0
    assert np.array_equal(o[0::2, :96], orc.anchor_max(T, anchor, 96))
    assert (o[1::2] == 7.0).all() and (o[:, 96:] == 7.0).all()
    # argument validation happens before any launch
    assert lib.disco_anchor_max_f32(t.data_ptr(), 300, 64, 96, None, 0, a.data_ptr(), 40, 3, None, 80, out.data_ptr(), 160, None, None) == -1
    assert lib.disco_anchor_max_f32(t.data_ptr(), 300, 128, 96, None, 0, a.data_ptr(), 40, 0, None, 80, out.data_ptr(), 160, None, None) == -1
    assert lib.disco_anchor_max_f32(t.data_ptr(), 300, 128, 96, None, 0, a.data_ptr(), 40, 3, None, 10, out.data_ptr(), 160, None, None) == -1


def test_anchor_max_empty_and_misaligned_operands():
    """Edge cases: no synthetic node in a class, zero width, a feature table whose rows are not 16-byte aligned (copied
    once into a padded buffer by the host side) and an output that is a column view of a wider matrix."""
    from disco_amd import ops
    T, anchor = _case(40, 7, 9, 3, seed=2)
    t = torch.from_numpy(T).to(DEV)
    a = torch.from_numpy(anchor).to(DEV)
    assert ops.anchor_max(t, a[:0]).shape == (0, 7)
    assert ops.anchor_max(t, a, 0).shape == (9, 0)
    got = ops.anchor_max(t, a, 5)                                   # pitch of 7 floats: scalar path or padded copy
    assert np.array_equal(got.cpu().numpy(), orc.anchor_max(T, anchor, 5))
    wide = torch.full((9, 23), -2.0, device=DEV)
    ops.anchor_max(t, a, 6, out=wide[:, 11:17])                     # unaligned column view of a wider output
    w = wide.cpu().numpy()
    assert np.array_equal(w[:, 11:17], orc.anchor_max(T, anchor, 6))
    assert (w[:, :11] == -2.0).all() and (w[:, 17:] == -2.0).all()
    with pytest.raises(ValueError):
        ops.anchor_max(t, a, 8)                                     # wider than the table
