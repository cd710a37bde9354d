"""Experiment (round 3): does the windowed Reddit kernel suffer from the STRIDED source window (a 128-float column block
of rows with a 2432-byte pitch: 4 cache lines out of every 19) rather than from the window size?  Times one hop's five
column blocks with x / y as column views of the [N, 608] buffers against packed [N, 128] blocks, over window sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disco_amd import SparseTensor, graphgen, ops, utils, _lib

dev = torch.device("cuda", 0)
spec = graphgen.SPECS["reddit"]
n, F = spec.n_nodes, spec.feat
row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
adj0 = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
del row, col
x = graphgen.features(n, F, spec.seed, device=dev, ld=608)
xfull = x.as_strided((n, 608), (608, 1))
yfull = torch.empty((n, 608), device=dev)
blocks = [(c0, min(c0 + 128, 608)) for c0 in range(0, 608, 128)]     # 4 x 128 + 96 (zero padded tail)
xp = [xfull[:, a:b].contiguous() for a, b in blocks]
yp = [torch.empty((n, b - a), device=dev) for a, b in blocks]
lib = _lib.load()
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    _lib.check(lib.disco_set_tuning(k.encode(), int(v)), k)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for wkb in (2048, 2560, 3072, 3584, 4096, 5632):
    adj = adj0.set_value(adj0.storage.value())
    ws, info = adj.schedule(8192, 4096, feat=128, sweep=True, window_bytes=wkb << 10)
    def strided():
        for a, b in blocks:
            ops.spmm_raw(adj, xfull[:, a:b], "sum", out=yfull[:, a:b])
    def packed():
        for i in range(len(blocks)):
            ops.spmm_raw(adj, xp[i], "sum", out=yp[i])
    ts, tp = timeit(strided), timeit(packed)
    ok = all(torch.equal(yfull[:, a:b], yp[i]) for i, (a, b) in enumerate(blocks))
    print(f"window {wkb:5d} KiB  windows {int(info.sweep_windows):3d}  strided {ts:7.3f} ms/hop   packed {tp:7.3f} ms/hop   same bits {ok}", flush=True)
