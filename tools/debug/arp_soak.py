#!/usr/bin/env python3
"""Soak of the persistent decoder (csrc/ar_persistent.hip) against the per-position loop: random latent sizes, both model widths,
with and without the temporal prior, batches decoded concurrently -- every reconstruction bit for bit.  usage: arp_soak.py [rounds]"""
import os
import random
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spatiotemporalentropymodel_amd.models as Mo  # noqa: E402
from spatiotemporalentropymodel_amd import config  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
rng = random.Random(7)
models = {}
for cls, widths in (("SpatioTemporalPriorModel_Res", (64, 96)), ("SpatioTemporalPriorModelWithoutTPM", (64, 96)), ("SpatioTemporalPriorModel_Res", (256, 192))):
    m = closed_form_fill_(getattr(Mo, cls)(*widths)).to(dev).eval()
    m.update(force=True)
    models[(cls, widths)] = m
bad = 0
for i in range(rounds):
    (cls, widths), m = rng.choice(list(models.items()))
    big = widths[1] == 192
    H, W = 4 * rng.randint(1, 4 if big else 8), 4 * rng.randint(1, 5 if big else 10)
    B = rng.choice([1, 1, 2, 3, 5]) if not big else rng.choice([1, 2])
    y_cur = closed_form_input(f"soak:y{i}", (B, widths[1], H, W), -8, 8).to(dev)
    y_cond = closed_form_input(f"soak:c{i}", (B, widths[1], H, W), -8, 8).to(dev)
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        os.environ.pop("STEM_AR_PERSISTENT", None)
        os.environ.pop("STEM_AR_NO_BATCH", None)
        config.runtime()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            a = m.decompress(enc["strings"], enc["shape"], y_cond)
            a = (a["y_hat"] if isinstance(a, dict) else a).clone()
        os.environ["STEM_AR_PERSISTENT"] = "0"
        os.environ["STEM_AR_NO_BATCH"] = "1"
        config.runtime()
        b = m.decompress(enc["strings"], enc["shape"], y_cond)
        b = (b["y_hat"] if isinstance(b, dict) else b).clone()
    ok = bool(torch.equal(a, b))
    bad += not ok
    print(f"{i:3d} {cls:36s} M={widths[1]:3d} B={B} {H:2d}x{W:2d}  {'ok' if ok else 'MISMATCH max %.3g' % float((a - b).abs().max())}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
