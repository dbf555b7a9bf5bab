#!/usr/bin/env python3
"""Spread of the image-tile kernel's main-loop time over the workgroups of one launch, by split / N tile / image / XCD slot.
Needs the experiments build (see f16x3_img_phases.py)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import _lib, functional as F  # noqa: E402

LAYERS = {"TPM.0": (192, 256, 5), "TPM.2": (256, 320, 5), "TPM.4": (320, 384, 5), "HE.0": (384, 256, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "TPM.4"
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
C_, K, R = LAYERS[name]
dev = torch.device("cuda:0")
x = torch.randn(16, C_, 16, 16, device=dev)
w = torch.randn(K, C_, R, R, device=dev) / (C_ * R * R) ** 0.5
b = torch.randn(K, device=dev) * 0.1
xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2_gen(w)
lib = _lib.hip()
lib.stem_exper_img_stamps.argtypes = [C.c_void_p]
lib.stem_exper_img_stamps.restype = None
NW = 4096
stamps = torch.zeros(NW * 8, dtype=torch.int64, device=dev)
tune = dict(fx3_gen_img=2)
if split:
    tune["fx3_split"] = split
with F.tuning(**tune):
    for _ in range(20):
        F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    torch.cuda.synchronize()
    lib.stem_exper_img_stamps(stamps.data_ptr())
    F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    torch.cuda.synchronize()
    lib.stem_exper_img_stamps(None)
s = stamps.view(NW, 8).cpu()
n = int((s[:, 0] > 0).sum())
gx, gy = 16, (K + 127) // 128
gz = n // (gx * gy)
hw = s[:n, 7].clone()
s = s[:n].double() / 100.0
loop = (s[:, 3] - s[:, 2]).view(gz, gy, gx)
start = (s[:, 0] - s[:, 0].min()).view(gz, gy, gx)
print(f"{name}: grid x={gx} (images) y={gy} z={gz}; loop us: min {float(loop.min()):.1f} median {float(loop.median()):.1f} max {float(loop.max()):.1f}")
print("by split z:", [round(float(loop[z].mean()), 1) for z in range(gz)])
print("by N tile y:", [round(float(loop[:, y].mean()), 1) for y in range(gy)])
print("by image x:", [round(float(loop[:, :, i].mean()), 1) for i in range(gx)])
lin = torch.arange(n).view(gz, gy, gx)
print("by linear id % 8 (XCD slot):", [round(float(loop[lin % 8 == k].mean()), 1) for k in range(8)])
for z in range(gz):
    for y in range(gy):
        print(f" z={z} y={y}:", " ".join(f"{float(v):5.0f}" for v in loop[z, y]))

xcc = (hw >> 60) & 0xF
hid = (hw >> 44) & 0xFFFF
cyc = (hw & 0xFFFFFFFFFFF).double()
cu = (hid >> 8) & 0xF
sh = (hid >> 12) & 0x1
se = (hid >> 13) & 0x7
mhz = cyc / loop.reshape(-1)
print("shader clock in the loop by XCC (MHz):", [(k, round(float(mhz[xcc == k].mean()))) for k in sorted(set(xcc.tolist()))])
lp = loop.reshape(-1)
print("by XCC_ID:", [(k, int((xcc == k).sum()), round(float(lp[xcc == k].mean()), 1)) for k in sorted(set(xcc.tolist()))])
print("by SE:", [(k, int((se == k).sum()), round(float(lp[se == k].mean()), 1)) for k in sorted(set(se.tolist()))])
print("by SH:", [(k, int((sh == k).sum()), round(float(lp[sh == k].mean()), 1)) for k in sorted(set(sh.tolist()))])
print("by CU:", [(k, int((cu == k).sum()), round(float(lp[cu == k].mean()), 1)) for k in sorted(set(cu.tolist()))])
for k in sorted(set(xcc.tolist())):
    sel = (xcc == k).nonzero().flatten().tolist()
    print(f" xcc {k}:", " ".join(f"{i}:se{int(se[i])}sh{int(sh[i])}cu{int(cu[i])}={float(lp[i]):.0f}" for i in sel))
