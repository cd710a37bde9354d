"""The allocation-free entry point (disco_spmm_csr_f32_noalloc: hub partial rows in a file-scope device array) gives
the same bits as disco_spmm_csr_f32 with a caller workspace -- through the raw C ABI, two calls in a row, the second
on a smaller input (the array keeps the first call's contents), and rejects problems beyond its scratch."""
import numpy as np
import pytest
import torch

from helpers import random_csr

pytestmark = pytest.mark.gpu


def _run_both(n, F, hubs, seed, reduce):
    from disco_amd import SparseTensor, _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rowptr, col, val = random_csr(n, n, 12, seed=seed, hubs=hubs, weighted=True)
    adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev),
                       value=torch.from_numpy(val).to(dev), sparse_sizes=(n, n), is_sorted=True)
    ws, info = adj.schedule(256, 1024)
    assert info.n_hub_partials > 0 and info.n_multi_rows > 0, "the case must use the partial-row scratch"
    need = int(lib.disco_spmm_workspace_bytes(info, F))
    assert 0 < need <= _lib.NOALLOC_SCRATCH_BYTES
    ld = (F + 3) // 4 * 4
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((n, ld)).astype(np.float32)).to(dev)
    sws = torch.empty(need, dtype=torch.uint8, device=dev)
    y_ref = torch.full((n, ld), -3.0, device=dev)
    y_new = torch.full((n, ld), -3.0, device=dev)
    stream = _lib.current_stream(dev)
    args = (adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n, x.data_ptr(), ld)
    _lib.check(lib.disco_spmm_csr_f32(*args, y_ref.data_ptr(), ld, F, reduce, ws.data_ptr(), info, sws.data_ptr(), need, stream), "ref")
    torch.cuda.synchronize()
    # device-wide free memory (hipMemGetInfo): torch's caching allocator never returns blocks to the device on
    # its own, so a drop here would be an allocation made by the call itself
    free_before = torch.cuda.mem_get_info(dev)[0]
    _lib.check(lib.disco_spmm_csr_f32_noalloc(*args, y_new.data_ptr(), ld, F, reduce, ws.data_ptr(), info, stream), "noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] >= free_before, "the allocation-free entry point allocated device memory"
    return y_ref, y_new, info


@pytest.mark.parametrize("reduce", [0, 1, 2])
def test_noalloc_entry_matches_workspace_entry_twice(reduce):
    # first call: large hub rows (many partial rows in the scratch); second call: smaller graph and width
    y_ref, y_new, info = _run_both(6000, 602, [(3, 60000), (10, 9000), (4000, 2500)], 11, reduce)
    assert int(info.n_hub_partials) >= 60
    assert torch.equal(y_ref, y_new)
    y_ref2, y_new2, info2 = _run_both(1500, 100, [(7, 5000), (8, 1100)], 12, reduce)
    assert int(info2.n_hub_partials) < int(info.n_hub_partials)
    assert torch.equal(y_ref2, y_new2)


def test_noalloc_entry_rejects_oversized_problems():
    from disco_amd import _lib
    lib = _lib.load()
    info = _lib.ScheduleInfo()
    info.n_rows, info.n_hub_partials = 10, 7000            # 7000 partial rows x 604 floats > 16 MiB
    assert lib.disco_spmm_workspace_bytes(info, 602) > _lib.NOALLOC_SCRATCH_BYTES
    rc = lib.disco_spmm_csr_f32_noalloc(None, None, None, 10, None, 608, None, 608, 602, 0, None, info, None)
    assert rc == -5 and b"scratch" in lib.disco_status_string(rc)


def test_noalloc_entry_under_hip_graph_capture():
    """The header offers this entry point to code under HIP graph capture: capture a K = 2 propagation made of
    disco_spmm_csr_f32_noalloc calls on a hub-bearing graph (first call made OUTSIDE the capture, as the header
    demands: it resolves the scratch address), replay the graph twice -- the second time on new input values in
    the same buffers -- and compare bits with the workspace entry point."""
    from disco_amd import SparseTensor, _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, F, K = 5000, 256, 2
    rowptr, col, val = random_csr(n, n, 14, seed=21, hubs=[(5, 40000), (77, 6000), (1234, 2100)], weighted=True)
    val = (val * 0.05).astype(np.float32)
    adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev),
                       value=torch.from_numpy(val).to(dev), sparse_sizes=(n, n), is_sorted=True)
    ws, info = adj.schedule(256, 1024)
    assert info.n_hub_partials > 0 and info.n_multi_rows > 0
    need = int(lib.disco_spmm_workspace_bytes(info, F))
    assert 0 < need <= _lib.NOALLOC_SCRATCH_BYTES
    x = torch.randn(n, F, device=dev)
    t_ref, y_ref = torch.empty(n, F, device=dev), torch.empty(n, F, device=dev)
    t_new, y_new = torch.zeros(n, F, device=dev), torch.zeros(n, F, device=dev)
    sws = torch.empty(need, dtype=torch.uint8, device=dev)
    csr = (adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n)

    def khop_noalloc(stream):
        _lib.check(lib.disco_spmm_csr_f32_noalloc(*csr, x.data_ptr(), F, t_new.data_ptr(), F, F, 0, ws.data_ptr(), info, stream), "hop 1")
        _lib.check(lib.disco_spmm_csr_f32_noalloc(*csr, t_new.data_ptr(), F, y_new.data_ptr(), F, F, 0, ws.data_ptr(), info, stream), "hop 2")

    def khop_ref():
        s = _lib.current_stream(dev)
        _lib.check(lib.disco_spmm_csr_f32(*csr, x.data_ptr(), F, t_ref.data_ptr(), F, F, 0, ws.data_ptr(), info, sws.data_ptr(), need, s), "ref 1")
        _lib.check(lib.disco_spmm_csr_f32(*csr, t_ref.data_ptr(), F, y_ref.data_ptr(), F, F, 0, ws.data_ptr(), info, sws.data_ptr(), need, s), "ref 2")

    khop_noalloc(_lib.current_stream(dev))          # outside any capture: resolves the scratch address of this device
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        free_before = torch.cuda.mem_get_info(dev)[0]
        with torch.cuda.graph(graph, stream=side):
            khop_noalloc(_lib.current_stream(dev))  # the capturing stream
        assert torch.cuda.mem_get_info(dev)[0] >= free_before - (2 << 20), "capture of the allocation-free K-hop allocated device memory"
    torch.cuda.current_stream(dev).wait_stream(side)
    for trial in range(2):
        if trial:
            x.copy_(torch.randn(n, F, device=dev))  # new values, same buffers: what a replayed graph sees
        t_new.fill_(-1.0)
        y_new.fill_(-1.0)
        graph.replay()
        khop_ref()
        torch.cuda.synchronize()
        assert torch.equal(t_new, t_ref) and torch.equal(y_new, y_ref), f"replay {trial}"
        assert bool(torch.isfinite(y_new).all())
