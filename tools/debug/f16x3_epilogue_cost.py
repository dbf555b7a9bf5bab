#!/usr/bin/env python3
"""What the epilogue forms of the 192-column kernel cost at the g_a.2 shape: conv / conv + GDN, fp32 / planes output, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, K, H = 16, 192, 192, 128
x = torch.randn(B, C, H, H, device=dev)
w = torch.randn(K, C, 5, 5, device=dev) / (C * 25) ** 0.5
b = torch.randn(K, device=dev) * 0.1
beta, gamma = torch.rand(K, device=dev) + 0.5, torch.rand(K, K, device=dev) * 0.1
xp, wp, gp = F.F16Planes.split(x), F.pack_weight_f16x2(w), F.pack_gdn_gamma_f16x2(gamma)
variants = {"conv -> fp32": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2),
            "conv -> planes": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, planes_out=True),
            "conv + GDN -> fp32": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta=beta, gamma=gamma, gp=gp),
            "conv + GDN -> planes": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta=beta, gamma=gamma, planes_out=True, gp=gp)}
ev = {k: [] for k in variants}
for it in range(12):
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev[k].append((e0, e1))
torch.cuda.synchronize()
for k, l in ev.items():
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in l[2:])
    print(f"{k:22s} median {t[len(t) // 2]:6.1f} us   min {t[0]:6.1f} us")
