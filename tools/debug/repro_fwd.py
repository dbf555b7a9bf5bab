#!/usr/bin/env python3
"""First tensor of the STEM training forward that differs between identical runs (fresh model per run, as the reproducibility test)."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.optim import configure_optimizers  # noqa: E402
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
runs = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    torch.manual_seed(7)
    stem = SpatioTemporalPriorModel_Res().to(dev).train()
    opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    y_cur = torch.randn(16, 192, 16, 16, device=dev, generator=g) * 3
    y_cond = y_cur + torch.rand(16, 192, 16, 16, device=dev, generator=g) - 0.5
    eng = stem.engine()
    y_hat, lik_y, lik_z, k = eng.forward(y_cur, y_cond, True)
    torch.cuda.synchronize()
    snap = {n: k[n].clone() for n in ("he_in", "he0", "he2", "z_hat", "hd0", "hd2", "tp0", "tp2", "epm_in", "e0", "e2", "gp") if isinstance(k.get(n), torch.Tensor)}
    for n, p in k["planes"].items():
        snap["planes:" + n] = p.merge().clone()
        snap["record:" + n] = torch.tensor(p.record())
    snap["lik_y"] = lik_y.clone()
    snap["wp:TPM.0"] = eng.TPM[0].wp6_fwd.clone()
    snap["bias:TPM.0"] = eng.TPM[0].mod.bias.detach().clone()
    again, againp = eng.TPM[0].fwd6(k["planes"]["yd"], F.ACT_LRELU, planes=True)
    torch.cuda.synchronize()
    snap["tp0:recomputed"] = again.clone()
    print(f"run {r}: tp0 in the flow == tp0 recomputed afterwards: {torch.equal(again, k['tp0'])}", flush=True)
    if not torch.equal(again, k["tp0"]):
        d = (again - k["tp0"]).abs() > 0                     # logical [B, C, H, W]
        print("   wrong elements:", int(d.sum()), "of", d.numel(), "| images", sorted(set(d.nonzero()[:, 0].tolist())),
              "| channels", int(d.nonzero()[:, 1].min()), "..", int(d.nonzero()[:, 1].max()),
              "| rows", sorted(set(d.nonzero()[:, 2].tolist())), "| cols", sorted(set(d.nonzero()[:, 3].tolist())))
        dd = (again - k["tp0"])
        idx = d.nonzero()[:8]
        for i in idx:
            b_, c_, y_, x_ = i.tolist()
            print(f"      [{b_},{c_},{y_},{x_}] flow {float(k['tp0'][b_, c_, y_, x_]):+.5f} recomputed {float(again[b_, c_, y_, x_]):+.5f}")
    runs.append(snap)
for r in range(1, len(runs)):
    diff = [n for n in runs[0] if not torch.equal(runs[0][n], runs[r][n])]
    print(f"run {r} vs 0: differing: {diff if diff else 'none'}")
    for n in diff[:6]:
        a, b = runs[0][n].double(), runs[r][n].double()
        print(f"     {n}: max |diff| {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3e}" + (f"  values {runs[0][n].tolist()} vs {runs[r][n].tolist()}" if n.startswith('record') else ""))
