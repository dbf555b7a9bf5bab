#!/usr/bin/env python3
"""192-column kernel on layer-wise-model shapes (3x3, stride 1, full resolution) per MFMA shape and LDS stages (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, C, K, R, st, relu) in ((16, 256, 64, 64, 3, 1, True), (16, 256, 160, 192, 3, 1, True), (16, 128, 192, 192, 5, 2, False),
                                  (16, 128, 192, 192, 5, 2, True), (16, 128, 192, 192, 3, 1, True)):
    x = torch.randn(B, C, H, H, device=dev)
    if relu:
        x = torch.relu(x)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2(w)
    line = f"B{B} {H}x{H} C{C}->K{K} k{R} s{st} {'relu-in' if relu else 'randn-in'}:"
    for depth, mfma in ((2, 32), (3, 32), (3, 16)):
        with F.tuning(fx3_tile=128, fx3_depth=depth, fx3_mfma=mfma):
            t = timeit(lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, R, R, st, R // 2, planes_out=True))
        Ho = (H + 2 * (R // 2) - R) // st + 1
        line += f"   stages {depth} mfma {mfma}: {t:8.1f} us ({2 * B * Ho * Ho * K * C * R * R / t / 1e6:6.1f} TF)"
    print(line)
