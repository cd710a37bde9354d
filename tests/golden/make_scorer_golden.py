"""Generates tests/golden/scorer_golden.npz by IMPORTING the reference's own PGE_Edge
(/root/reference/models/parametrized_adj_lp.py -- pure torch, the only importable module of the
reference) in the build container.  The reference itself never travels: only
the inputs (Z, parameters) and its outputs (probabilities, logits) are committed.

Run:  python tests/golden/make_scorer_golden.py     (needs /root/reference; CPU only)
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/models/parametrized_adj_lp.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scorer_golden.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_parametrized_adj_lp", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(20250222)
    out = {}
    for tag, (n, d, nhid) in {"a": (24, 12, 256), "b": (40, 6, 32)}.items():
        model = mod.PGE_Edge(nfeat=d, nhid=nhid, nlayers=3, device="cpu")
        # non-trivial BatchNorm state, as after training
        with torch.no_grad():
            for bn in model.bns:
                bn.running_mean.copy_(torch.randn_like(bn.running_mean) * 0.3)
                bn.running_var.copy_(torch.rand_like(bn.running_var) * 1.5 + 0.25)
                bn.weight.copy_(torch.randn_like(bn.weight) * 0.5 + 1.0)
                bn.bias.copy_(torch.randn_like(bn.bias) * 0.2)
            model.layers[-1].weight.mul_(6.0)  # spread the logits so thresholds near 1 keep some edges
        Z = torch.randn(n, d)
        S = torch.zeros(n, n)
        for i in range(n):  # the reference's pair loop, LargeScaleCondensing.py:304-305
            S[i] = model.inference(torch.cat([Z[[i] * n], Z[np.arange(n)]], axis=1)).detach()
        # logits: same forward without the final sigmoid
        model.eval()
        with torch.no_grad():
            L = torch.zeros(n, n)
            for i in range(n):
                h = torch.cat([Z[[i] * n], Z[np.arange(n)]], axis=1)
                for ix, layer in enumerate(model.layers):
                    h = layer(h)
                    if ix != len(model.layers) - 1:
                        h = torch.relu(model.bns[ix](h))
                L[i] = h.reshape(-1)
        out[f"{tag}_Z"] = Z.numpy()
        out[f"{tag}_S"] = S.numpy()
        out[f"{tag}_L"] = L.numpy()
        for k, v in model.state_dict().items():
            out[f"{tag}_param_{k}"] = v.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
