#!/usr/bin/env python3
"""Profiling driver for csrc/conv_f16x3.hip at the bench's g_a.2 / g_a.4 sizes: f16x3_prof.py [variant] [iters] [layer]
variant: planes | fp32 | conv | split | ref (fp32-MFMA kernel); layer: 2 (128^2 -> 64^2) or 4 (64^2 -> 32^2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "planes"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
layer = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, K = 16, 192, 192
H = W = 128 if layer == 2 else 64
x = torch.randn(B, C, H, W, device=dev)
w = torch.randn(K, C, 5, 5, device=dev) / (C * 25) ** 0.5
b = torch.randn(K, device=dev) * 0.1
beta = torch.rand(K, device=dev) + 0.5
gamma = torch.rand(K, K, device=dev) * 0.1
xp = F.F16Planes.split(x)
wp = F.pack_weight_f16x2(w)
wp32 = F.pack_weight(w, F.PACK_CONV_FWD)
xn = F.to_nhwc(x)
gp = F.pack_gdn_gamma_f16x2(gamma)
fn = {"planes": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta, gamma, planes_out=True, gp=gp),
      "fp32": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta, gamma, gp=gp),
      "conv": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2),
      "convplanes": lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, planes_out=True),
      "split": lambda: F.F16Planes.split(x),
      "c4planes": None, "c4fp32": None,
      "ref": lambda: F.conv2d_gdn_fwd(xn, wp32, b, beta, gamma, K, 5, 5, 2, 2)}[variant]
if variant in ("c4planes", "c4fp32"):      # g_a.0 + GDN: 3 -> 192 channels, 256^2 -> 128^2
    w0 = torch.randn(K, 3, 5, 5, device=dev) / 75 ** 0.5
    x4 = torch.rand(B, 256, 256, 4, device=dev)
    wp0 = F.pack_weight(w0, F.PACK_CONV_FWD_C4)
    fn = (lambda: F.conv2d_fwd_c4_gdn_planes(x4, wp0, b, beta, gamma, K, 5, 5, 2, 2)) if variant == "c4planes" else \
        (lambda: F.conv2d_fwd_c4_gdn(x4, wp0, b, beta, gamma, K, 5, 5, 2, 2))
for _ in range(iters):
    out = fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = fn()
e1.record()
torch.cuda.synchronize()
print(f"{variant} layer {layer}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per launch")
