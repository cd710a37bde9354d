#!/usr/bin/env python3
"""Host-to-device upload of the Reddit-shaped CSR + X from pinned memory (what the reference pays before each of
its 31 evaluations, LargeScaleCondensing.py:344): the PCIe-inclusive figure quoted in DESIGN.md section 4."""
import time, torch, json
n, F, nnz = 232965, 602, 114521225
x = torch.empty((n, 608), dtype=torch.float32).pin_memory()
col = torch.empty(nnz, dtype=torch.int32).pin_memory()
val = torch.empty(nnz, dtype=torch.float32).pin_memory()
rp = torch.empty(n + 1, dtype=torch.int64).pin_memory()
torch.cuda.synchronize()
ts = []
for i in range(4):
    t0 = time.perf_counter()
    a = x.cuda(non_blocking=True); b = col.cuda(non_blocking=True); c = val.cuda(non_blocking=True); d = rp.cuda(non_blocking=True)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
nbytes = x.numel()*4 + nnz*8 + (n+1)*8
print(json.dumps({"bytes": nbytes, "upload_ms_pinned": [round(t*1e3,2) for t in ts], "GBs": nbytes/min(ts)/1e9}))
