#!/usr/bin/env python3
"""Split-K hand-off screen: the slab region of the workspace is filled with NaN before every launch -- a last arriver that reads
a slab before its writer's stores have landed shows up as NaN (re-running on a warm workspace hides it: the stale values are the
right ones)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
form = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tot = 0
for name, B, C, H, W, K, R in [("TPM.0", 16, 192, 16, 16, 256, 5), ("HE.0", 16, 384, 16, 16, 256, 3), ("TPM.4", 16, 320, 16, 16, 384, 5)]:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2_gen(w)
    with F.tuning(fx3_gen_img=form):
        y0, _ = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
        torch.cuda.synchronize()
        y0 = y0.clone()
        ws = [v for k, v in F._WS.items() if k[1] == F._stream()][0]
        bad = 0
        for i in range(n):
            ws[16384:].fill_(float("nan"))                      # counters (first 64 KB) stay zero
            y, _ = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
            if not torch.equal(y, y0):
                bad += 1
        torch.cuda.synchronize()
        print(f"form {form} {name}: {bad} of {n} launches on a poisoned workspace differ", flush=True)
        tot += bad
print("POISON SCREEN", "FAILED" if tot else "clean")
