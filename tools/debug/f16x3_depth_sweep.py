#!/usr/bin/env python3
"""g_a.2 / g_a.4 / g_a.6-sized launches of the 192-column kernel (+ fused GDN, planes out) per pixel tile, main-loop form (fx3_depth: LDS stages) and MFMA shape (fx3_mfma)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, H in (("g_a.2", 128), ("g_a.4", 64), ("g_a.6", 32)):
    B, C, K = 16, 192, 192
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(K, C, 5, 5, device=dev) / (C * 25) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    beta, gamma = torch.rand(K, device=dev) + 0.5, torch.rand(K, K, device=dev) * 0.1
    xp, wp, gp = F.F16Planes.split(x), F.pack_weight_f16x2(w), F.pack_gdn_gamma_f16x2(gamma)
    line = f"{name} ({B}x{H}x{H}):"
    for tile in (64, 128):
        for depth, mfma in ((2, 32), (3, 32), (3, 16)):
            with F.tuning(fx3_tile=tile, fx3_depth=depth, fx3_mfma=mfma):
                t = timeit(lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta=beta, gamma=gamma, planes_out=True, gp=gp))
            line += f"   tile {tile} stages {depth} mfma {mfma}: {t:6.1f} us"
    print(line)
