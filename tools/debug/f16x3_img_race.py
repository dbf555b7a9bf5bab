#!/usr/bin/env python3
"""Race screen for the image-tile kernel: every STEM layer shape launched `n` times on the same operands, outputs compared bit
for bit with the first launch (forward with planes, input gradient), with a second stream keeping the chip unevenly loaded."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CASES = [("TPM.0", 16, 192, 16, 16, 256, 5, 0), ("TPM.2", 16, 256, 16, 16, 320, 5, 0), ("TPM.4", 16, 320, 16, 16, 384, 5, 0),
         ("HE.0", 16, 384, 16, 16, 256, 3, 0), ("HD.4", 16, 256, 16, 16, 384, 3, 0), ("ctx", 16, 192, 16, 16, 384, 5, 12),
         ("EPM.0", 16, 1152, 16, 16, 768, 1, 0), ("EPM.4", 16, 576, 16, 16, 384, 1, 0), ("odd", 3, 96, 20, 31, 160, 5, 0), ("odd3", 2, 64, 37, 18, 96, 3, 0),
         ("roi64", 4, 64, 64, 64, 64, 3, 0), ("roi128", 2, 96, 128, 128, 96, 3, 0), ("roi32", 8, 192, 32, 32, 192, 3, 0)]
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
bad = 0
for name, B, C, H, W, K, R, taps in CASES:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xp = F.F16Planes.split(x)
    wp = F.pack_weight_f16x2_gen(w, taps=taps) if taps else F.pack_weight_f16x2_gen(w)
    kw = dict(epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    xn = F.to_nhwc(x)
    dyp = F.F16Planes.split(torch.randn(B, K, H, W, device=dev)) if K % 32 == 0 else None
    wpd = F.pack_weight_f16x2_gen(w, flip=True) if not taps and K % 32 == 0 else None
    if taps:
        kw["taps"] = taps
    for split in (0, 1, 2, 3, 7):
        tune = dict(fx3_gen_img=2)
        if split:
            tune["fx3_split"] = split
        with F.tuning(**tune):
            y0, yp0 = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, **kw)
            y0, p0 = y0.clone(), yp0.merge().clone()           # (the planes buffer itself has unwritten record slots)
            nbad = 0
            for i in range(n):
                if i % 3 == 0:
                    with torch.cuda.stream(side):
                        junk @ junk
                y, yp = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, **kw)
                if not (torch.equal(y, y0) and torch.equal(yp.merge(), p0)):
                    nbad += 1
            torch.cuda.synchronize()
            print(f"{name:6s} split {split}: {nbad} of {n} launches differ from the first", flush=True)
            bad += nbad
            if taps or K % 32:
                continue
            # input gradient: dy planes, mirrored weight, DACT epilogue on x, no bias, fp32 + planes out
            d0, dp0 = F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, R // 2, epi=F.GEN_EPI_DACT, slope=0.01, z=xn, want_planes=True)
            d0, dp0 = d0.clone(), dp0.merge().clone()
            nbad = 0
            for i in range(n):
                if i % 3 == 0:
                    with torch.cuda.stream(side):
                        junk @ junk
                d, dp = F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, R // 2, epi=F.GEN_EPI_DACT, slope=0.01, z=xn, want_planes=True)
                if not (torch.equal(d, d0) and torch.equal(dp.merge(), dp0)):
                    nbad += 1
            torch.cuda.synchronize()
            print(f"{name:6s} split {split} dgrad: {nbad} of {n} launches differ from the first", flush=True)
            bad += nbad
print("RACE SCREEN", "FAILED" if bad else "clean")
