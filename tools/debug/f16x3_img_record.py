#!/usr/bin/env python3
"""Scale record written by the image-tile kernel: header, scale, one slot per (pixel block, N tile) = max |y| of that tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for name, B, C, H, W, K, R in [("TPM.0", 16, 192, 16, 16, 256, 5), ("TPM.2", 16, 256, 16, 16, 320, 5), ("HE.0", 16, 384, 16, 16, 256, 3), ("odd", 3, 96, 20, 31, 160, 5)]:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2_gen(w)
    for split in (0, 1):
        for rep in range(3):
            with F.tuning(fx3_gen_img=2, **({"fx3_split": split} if split else {})):
                y, yp = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
            torch.cuda.synchronize()
            n = (yp.data.numel() - yp.q_offset) // 4
            q = yp.data[yp.q_offset:].view(torch.float32)[:n].cpu()
            ns = int(q[:1].view(torch.int32)[0])
            ty, tx, ntn = (H + 15) // 16, (W + 15) // 16, (K + 127) // 128
            ya = y.abs().cpu()                                   # logical [B, K, H, W]
            want = []
            for nt in range(ntn):
                for bi in range(B):
                    for iy in range(ty):
                        for ix in range(tx):
                            want.append(float(ya[bi, nt * 128:(nt + 1) * 128, iy * 16:(iy + 1) * 16, ix * 16:(ix + 1) * 16].max()))
            want = torch.tensor(want)
            ok = ns == len(want) and torch.equal(q[16:16 + ns], want) and float(q[2:16].abs().max()) == 0.0
            print(f"{name} split {split} rep {rep}: nslots {ns} (expected {len(want)}), inv {float(q[1]):.3e}, slots {'match' if ok else 'MISMATCH'}"
                  + ("" if ok else f" first bad {[(i, float(q[16 + i]), float(want[i])) for i in range(min(ns, len(want))) if float(q[16 + i]) != float(want[i])][:4]}"), flush=True)
            bad += not ok
print("RECORD CHECK", "FAILED" if bad else "ok")
