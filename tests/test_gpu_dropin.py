"""The reference drivers' own call shapes, with HOST tensors where the reference has host tensors
(/root/reference/LargeScaleCondensing.py:478-491, :330-355; LargeScaleCondensing_Sampled.py:491-499): only the imports are
swapped.  Results against the float64 oracle / the device-resident path."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import random_csr

pytestmark = pytest.mark.gpu
device = "cuda:0"


def _host_graph(n, seed):
    rowptr, col, _ = random_csr(n, n, 7, seed=seed, hubs=[(5, 300)], weighted=False)
    a = sp.csr_matrix((np.ones(len(col), dtype=np.float32), col, rowptr), shape=(n, n))
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    return a


@pytest.mark.parametrize("model_name", ["SGC", "GCN", "SAGE"])
def test_driver_call_shapes_with_host_adjacency(model_name):
    import disco_amd.utils as utils
    from disco_amd import SparseTensor
    from disco_amd.models import basicgnn_large as M
    torch.manual_seed(0)
    n, d, nclass = 900, 24, 6
    adj_sp = _host_graph(n, 31)
    labels = torch.randint(0, nclass, (n,))
    feat = torch.randn(n, d)
    adj = utils.to_tensor(adj_sp, device="cpu")                                     # utils.py:322-341: torch sparse COO on the host
    assert utils.is_sparse_tensor(adj) and adj.device.type == "cpu"
    if model_name in ("GCN", "SGC"):
        adj = utils.normalize_adj_tensor(adj, sparse=True)                          # :480 -- host in, host out, GPU arithmetic
        assert adj.device.type == "cpu" and adj.is_sparse
        adj = SparseTensor(row=adj._indices()[0], col=adj._indices()[1], value=adj._values(), sparse_sizes=adj.size()).t()   # :486
    else:
        adj = SparseTensor(row=torch.concat((adj._indices()[0], torch.arange(len(labels)))),
                           col=torch.concat((adj._indices()[1], torch.arange(len(labels)))),
                           value=torch.concat((adj._values(), torch.ones((len(labels),)))), sparse_sizes=adj.size()).t()     # :488
    assert adj.device.type == "cpu"
    if model_name == "SGC":
        model = M.SGC(nfeat=d, nhid=32, nclass=nclass, dropout=0, nlayers=2, sgc=True).cuda()
    elif model_name == "GCN":
        model = M.GCN(nfeat=d, nhid=32, nclass=nclass, dropout=0.0, nlayers=2, norm="BatchNorm", act="relu").cuda()
    else:
        model = M.GraphSAGE(nfeat=d, nhid=32, nclass=nclass, dropout=0.0, nlayers=2, norm="BatchNorm", act="relu").cuda()
    model.initialize()
    outs = []
    for _ in range(3):                                                             # the evaluations of train_on_syn_graph, :344
        outs.append(model.predict(feat.to(device), adj.to(device)))
    assert adj.to(device) is adj.to(device) and torch.equal(outs[0], outs[2])
    # the same evaluation with everything built on the device from the start
    adj_dev = utils.to_tensor(adj_sp, device=device)
    if model_name in ("GCN", "SGC"):
        adj_dev = utils.normalize_adj_tensor(adj_dev, sparse=True)
        adj_dev = SparseTensor(row=adj_dev._indices()[0], col=adj_dev._indices()[1], value=adj_dev._values(), sparse_sizes=adj_dev.size()).t()
    else:
        ar = torch.arange(n, device=device)
        adj_dev = SparseTensor(row=torch.concat((adj_dev._indices()[0], ar)), col=torch.concat((adj_dev._indices()[1], ar)),
                               value=torch.concat((adj_dev._values(), torch.ones(n, device=device))), sparse_sizes=adj_dev.size()).t()
    ref = model.predict(feat.to(device), adj_dev)
    assert torch.allclose(outs[0], ref, rtol=1e-5, atol=1e-6)
    # dense float64 check of the whole evaluation for SGC: log_softmax(A^2 X W^T + b)
    if model_name == "SGC":
        A = adj_sp.toarray().astype(np.float64)
        if A[0, 0] == 0:
            A = A + np.eye(n)
        r = np.power(A.sum(1), -0.5)
        r[np.isinf(r)] = 0
        A = (r[:, None] * A) * r[None, :]
        W = model.convs[0].lin.weight.detach().cpu().double().numpy()
        b = model.convs[0].lin.bias.detach().cpu().double().numpy()
        z = A @ (A @ feat.double().numpy()) @ W.T + b
        z = z - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - z.max(1, keepdims=True)
        assert np.allclose(outs[0].cpu().double().numpy(), z, rtol=1e-4, atol=1e-4)


def test_gcn_norm_and_normalize_sparse_tensor_accept_host_operands():
    """LargeScaleCondensing_Sampled.py:491: `utils.gcn_norm(data.edge_index, None, N, add_self_loops=False)` on host tensors."""
    import disco_amd.utils as utils
    rng = np.random.default_rng(3)
    n, E = 500, 6000
    ei = torch.from_numpy(rng.integers(0, n, (2, E)))
    w = torch.from_numpy(rng.random(E).astype(np.float32) + 0.1)
    for weights in (None, w):
        for loops in (False, True):
            a_i, a_w = utils.gcn_norm(ei, weights, n, add_self_loops=loops)
            b_i, b_w = utils.gcn_norm(ei.to(device), None if weights is None else weights.to(device), n, add_self_loops=loops)
            assert a_w.device.type == "cpu" and a_i.device.type == "cpu"
            assert torch.equal(a_i, b_i.cpu()) and torch.equal(a_w, b_w.cpu())
            if not loops:
                assert a_i is ei
    adj = torch.sparse_coo_tensor(ei, w, (n, n))
    a = utils.normalize_sparse_tensor(adj)
    b = utils.normalize_sparse_tensor(adj.to(device))
    assert a.device.type == "cpu" and torch.equal(a._indices(), b._indices().cpu()) and torch.equal(a._values(), b._values().cpu())
