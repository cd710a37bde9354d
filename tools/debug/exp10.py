import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disco_amd.models.parametrized_adj_lp import PGE_Edge
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(1)
m = PGE_Edge(nfeat=256, nhid=256).to(dev).eval()
with torch.no_grad():
    m.layers[-1].weight.mul_(3.0)
Z = torch.randn(n, 256, device=dev)
out = {}
for fused in (True, False):
    for thr in (0.99, 0.5):
        m.edge_construction(Z[:512], thr, "transductive", fused=fused)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ei, ew = m.edge_construction(Z, thr, "transductive", fused=fused)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        flops = 2.0 * n * n * (256 * 256 + 256)
        out[f"fused={fused} thr={thr}"] = {"s": round(min(ts), 4), "edges": int(ei.shape[1]), "TFLOPs": round(flops / min(ts) / 1e12, 1),
                                            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
        torch.cuda.reset_peak_memory_stats()
print(json.dumps(out, indent=1))
