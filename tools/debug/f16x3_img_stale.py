#!/usr/bin/env python3
"""Does a launch of the image-tile kernel depend on what the PREVIOUS launch left behind (workspace slabs, LDS, records)?
conv(x1), conv(x2), conv(x1) again: the third result must equal the first bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
CASES = [("TPM.0", 16, 192, 16, 16, 256, 5, 0), ("TPM.2", 16, 256, 16, 16, 320, 5, 0), ("TPM.4", 16, 320, 16, 16, 384, 5, 0),
         ("HE.0", 16, 384, 16, 16, 256, 3, 0), ("HD.4", 16, 256, 16, 16, 384, 3, 0), ("ctx", 16, 192, 16, 16, 384, 5, 12)]
bad = 0
for form in (2, 1):
    for name, B, C, H, W, K, R, taps in CASES:
        w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
        b = torch.randn(K, device=dev) * 0.1
        wp = F.pack_weight_f16x2_gen(w, taps=taps) if taps else F.pack_weight_f16x2_gen(w)
        kw = dict(epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
        if taps:
            kw["taps"] = taps
        xs = [F.F16Planes.split(torch.randn(B, C, H, W, device=dev) * s) for s in (1.0, 3.0)]
        for split in (0, 2, 5):
            tune = dict(fx3_gen_img=form)
            if split:
                tune["fx3_split"] = split
            with F.tuning(**tune):
                res = []
                for i in (0, 1, 0, 1, 0):
                    y, yp = F.conv2d_f16x3_gen(xs[i], wp, b, K, R, R, 1, R // 2, **kw)
                    res.append((i, y.clone(), yp.merge().clone()))
                torch.cuda.synchronize()
                ok = all(torch.equal(res[j][1], res[0][1]) and torch.equal(res[j][2], res[0][2]) for j in (2, 4)) and torch.equal(res[3][1], res[1][1])
                d = max(float((res[j][1] - res[0][1]).abs().max()) for j in (2, 4))
                print(f"form {form} {name:6s} split {split}: {'same' if ok else 'DIFFERENT'} (max diff {d:.3e} of {float(res[0][1].abs().max()):.3e})", flush=True)
                bad += not ok
print("STALE-STATE SCREEN", "FAILED" if bad else "clean")
