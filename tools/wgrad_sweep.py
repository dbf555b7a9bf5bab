#!/usr/bin/env python3
"""Time conv2d_wgrad (without bias gradient) for a few shapes under the tile config forced by STEM_WGRAD_CFG. Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402
from spatiotemporalentropymodel_amd import _lib  # noqa: E402

SHAPES = [(8, 128, 128, 128, 128, 3, 1), (8, 192, 256, 256, 160, 3, 1), (8, 160, 256, 256, 128, 3, 1), (8, 128, 64, 64, 128, 3, 1),
          (8, 128, 32, 32, 128, 3, 1), (8, 128, 16, 16, 192, 3, 1), (8, 192, 16, 16, 192, 3, 1), (8, 4, 256, 256, 192, 3, 1),
          (16, 256, 16, 16, 320, 5, 1), (16, 192, 64, 64, 192, 5, 2), (16, 1152, 16, 16, 768, 1, 1)]
for (B, C, H, W, K, R, st) in SHAPES:
    pd = R // 2
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    Ho, Wo = F.conv_out_hw(H, W, R, R, st, pd)
    dy = torch.randn(B, K, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last)
    fn = lambda: F.conv2d_wgrad(x, dy, K, R, R, st, pd, need_db=False)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gf = 2.0 * B * Ho * Wo * K * C * R * R / 1e9
    sp = _lib.hip().stem_wgrad_splits(B, Ho, Wo, C, K, R, R)
    print(f"cfg {os.environ.get('STEM_WGRAD_CFG', 'auto'):>4} B{B} C{C} {H}x{W} K{K} k{R} s{st}: splits {sp:3d} {ms * 1e3:8.1f} us {gf / ms:6.1f} TF/s")
