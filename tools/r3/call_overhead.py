"""Per-call cost of the binding at small sizes (round-2 verdict, boundary row: "per-call overhead unmeasured at C1 sizes").
Host wall time per call over many back-to-back calls (the queue hides nothing once it is full), for: the raw C-ABI call through
ctypes, ops.spmm_raw (operand checks + schedule lookup + ctypes), SGConv.forward at K = 2, and the same K = 2 propagation
replayed from a HIP graph captured over the allocation-free entry point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disco_amd import SparseTensor, graphgen, ops, utils, _lib
from disco_amd.models.convs_large import SGConv

dev = torch.device("cuda", 0)
lib = _lib.load()

def wall(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for name, n, nnz, F in (("tiny", 2000, 20000, 128), ("arxiv-shaped (C1)", 169343, 1166243, 128)):
    row, col = graphgen.powerlaw_edges(n, nnz, 0.667, 5, device=dev)
    adj = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
    x = graphgen.features(n, F, 1, device=dev)
    y, t = torch.empty_like(x), torch.empty_like(x)
    ws, info = adj.schedule(feat=F)
    need = int(lib.disco_spmm_workspace_bytes(info, F))
    sws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    stream = _lib.current_stream(dev)
    args = (adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n)
    raw = lambda: lib.disco_spmm_csr_f32(*args, x.data_ptr(), F, y.data_ptr(), F, F, 0, ws.data_ptr(), info, sws.data_ptr(), need, stream)
    reps = 2000 if n < 10000 else 300
    t_raw = wall(raw, reps)
    t_ops = wall(lambda: ops.spmm_raw(adj, x, "sum", out=y), reps)
    conv = SGConv(F, 16, K=2).to(dev)
    with torch.no_grad():
        t_conv = wall(lambda: conv(x, adj), reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); raw(); e1.record(); torch.cuda.synchronize()
    line = f"{name}: N={n} nnz={adj.nnz()} F={F}: one hop on the device {e0.elapsed_time(e1) * 1e3:.1f} us; host wall per call: raw C ABI {t_raw:.1f} us, ops.spmm_raw {t_ops:.1f} us, SGConv(K=2).forward {t_conv:.1f} us"
    if need <= _lib.NOALLOC_SCRATCH_BYTES:
        na = lambda s: (lib.disco_spmm_csr_f32_noalloc(*args, x.data_ptr(), F, t.data_ptr(), F, F, 0, ws.data_ptr(), info, s),
                        lib.disco_spmm_csr_f32_noalloc(*args, t.data_ptr(), F, y.data_ptr(), F, F, 0, ws.data_ptr(), info, s))
        na(stream); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                na(_lib.current_stream(dev))
        torch.cuda.current_stream(dev).wait_stream(side)
        t_graph = wall(g.replay, reps)
        t_two = wall(lambda: na(stream), reps)
        line += f"; K=2 as two raw calls {t_two:.1f} us, as one HIP-graph replay {t_graph:.1f} us"
    print(line, flush=True)
