#!/usr/bin/env python3
"""Distance of the fp16 route and of the fp32-MFMA route of the STEM training pass from the CPU oracle (double accumulation),
per parameter gradient, with the discrete decisions (leaky-ReLU sides, likelihood bound) compared -- the numbers behind the
gates of tests/test_hip_f16x3.py::test_engine_schedule_with_and_without_bf16_layers."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_pass as OP  # noqa: E402
from spatiotemporalentropymodel_amd import engine as E  # noqa: E402
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input  # noqa: E402

d = torch.device("cuda:0")
for (ebc, cin, B, ls, lo, hi, res) in ((256, 192, 2, 16, -6, 6, None), (256, 192, 2, 16, -4, 4, 1.5), (64, 64, 2, 8, -4, 4, 1.5), (64, 24, 2, 8, -4, 4, 1.5),
                                       (64, 48, 2, 8, -4, 4, 1.5)):
    y_cond = closed_form_input("eng:c", (B, cin, ls, ls), lo, hi).to(d)
    y_cur = closed_form_input("eng:y", (B, cin, ls, ls), lo, hi).to(d) if res is None else y_cond + closed_form_input("eng:r", (B, cin, ls, ls), -res, res).to(d)
    runs = {}
    for tag, on in (("fp16", True), ("fp32", False)):
        E.StemEngine.use_fx3 = on
        m = closed_form_fill_(SpatioTemporalPriorModel_Res(ebc, cin)).to(d).train()
        runs[tag] = OP.hip_train_pass(m, y_cur, y_cond, f"rvo{cin}")
    ref, rgrads, racts = OP.oracle_train_pass(m, y_cur, y_cond, runs["fp32"][4], True)
    print(f"\n== ebc {ebc} cin {cin} B {B} latents {ls}x{ls} inputs [{lo},{hi}] residual {res}")
    for tag in ("fp16", "fp32"):
        out, loss, grads, acts, _ = runs[tag]
        fl = OP.decisions_flipped(acts, racts, OP.host(out["likelihoods"]["y"]), ref["lik_y"])
        dist = sorted(((OP.grad_distance(grads[n], g), n) for n, g in rgrads.items() if n in grads), reverse=True)
        lik = float(np.max(np.abs(OP.host(out["likelihoods"]["y"]) - ref["lik_y"]) / np.maximum(ref["lik_y"], 0.1 * ref["lik_y"].max())))
        print(f"  {tag}: flips vs oracle {fl}; lik_y {lik:.2e}; worst gradients " + ", ".join(f"{n} {v:.1e}" for v, n in dist[:5]))
    fl = OP.decisions_flipped(runs["fp16"][3], runs["fp32"][3], OP.host(runs["fp16"][0]["likelihoods"]["y"]), OP.host(runs["fp32"][0]["likelihoods"]["y"]))
    dist = sorted(((OP.grad_distance(runs["fp16"][2][n], g), n) for n, g in runs["fp32"][2].items()), reverse=True)
    print(f"  fp16 vs fp32: flips {fl}; worst gradients " + ", ".join(f"{n} {v:.1e}" for v, n in dist[:5]))
