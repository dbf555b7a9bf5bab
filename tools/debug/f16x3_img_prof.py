#!/usr/bin/env python3
"""Profiling driver: N launches of the general f16x3 convolution on one STEM layer shape (B=16), image-tile or 128-pixel form.
    python3 tools/debug/f16x3_img_prof.py [layer] [img|gen128] [split] [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

LAYERS = {"TPM.0": (192, 256, 5), "TPM.2": (256, 320, 5), "TPM.4": (320, 384, 5), "HE.0": (384, 256, 3), "EPM.0": (1152, 768, 1), "EPM.4": (576, 384, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "TPM.4"
form = sys.argv[2] if len(sys.argv) > 2 else "img"
split = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
C_, K, R = LAYERS[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(16, C_, 16, 16, device=dev)
w = torch.randn(K, C_, R, R, device=dev) / (C_ * R * R) ** 0.5
b = torch.randn(K, device=dev) * 0.1
xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2_gen(w)
tune = dict(fx3_gen_img=2 if form == "img" else 1)
if split:
    tune["fx3_split"] = split
with F.tuning(**tune):
    for _ in range(n):
        F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    torch.cuda.synchronize()
