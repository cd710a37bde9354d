import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import random_csr
from disco_amd import SparseTensor, ops
dev = torch.device("cuda", 0)
def to_adj(rowptr, col, val, n):
    return SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev), value=torch.from_numpy(val).to(dev), sparse_sizes=(n, n), is_sorted=True)
n, F = 3000, 96
rowptr, col, val = random_csr(n, n, 90, seed=77, hubs=[(5, 2900)], weighted=True)
x = np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)
xd = torch.from_numpy(x).to(dev)
plain = to_adj(rowptr, col, val, n); plain.schedule(8192, 4096, feat=F, sweep=False)
yp = ops.spmm_raw(plain, xd, "sum").cpu().numpy()
# sequential fma chain on the host
ref = np.zeros((n, F), np.float32)
for r in range(n):
    a = np.zeros(F, np.float64)
    for e in range(rowptr[r], rowptr[r + 1]):
        a = (np.float64(val[e]) * x[col[e]].astype(np.float64) + a).astype(np.float32).astype(np.float64)
    ref[r] = a
print("plain vs host fma chain: mismatching elements", int((yp != ref).sum()))
for wb_rows in (3000, 1500, 1000, 512, 64):
    win = to_adj(rowptr, col, val, n)
    _, iw = win.schedule(8192, 4096, feat=F, sweep=True, window_bytes=wb_rows * 4 * F)
    yw = ops.spmm_raw(win, xd, "sum").cpu().numpy()
    bad = (yw != yp)
    rows = np.nonzero(bad.any(1))[0]
    print("window_rows", wb_rows, "P", int(iw.sweep_windows), "mismatch elems", int(bad.sum()), "rows", len(rows), rows[:10], "deg", np.diff(rowptr)[rows[:10]],
          "vs host chain", int((yw != ref).sum()))
    if len(rows):
        r = rows[0]; cbad = np.nonzero(bad[r])[0]
        print("  row", r, "cols", cbad[:8], yw[r, cbad[:4]], yp[r, cbad[:4]], ref[r, cbad[:4]])
