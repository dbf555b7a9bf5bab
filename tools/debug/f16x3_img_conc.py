#!/usr/bin/env python3
"""Two image-tile launches running CONCURRENTLY on two streams (the TPM chain beside the hyper branch in the training step):
results against the same launches run alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
form = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def mk(B, C, H, W, K, R):
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    return F.F16Planes.split(x), F.pack_weight_f16x2_gen(w), b, K, R


def run(c):
    xp, wp, b, K, R = c
    return F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)


A, Bc = mk(16, 192, 16, 16, 256, 5), mk(16, 384, 16, 16, 256, 3)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with F.tuning(fx3_gen_img=form):
    ya, _ = run(A)
    yb, _ = run(Bc)
    torch.cuda.synchronize()
    ya, yb = ya.clone(), yb.clone()
    bad = 0
    for it in range(200):
        with F.on_stream(s1):
            y1, _ = run(A)
        with F.on_stream(s2):
            y2, _ = run(Bc)
        torch.cuda.synchronize()
        if not torch.equal(y1, ya) or not torch.equal(y2, yb):
            bad += 1
            if bad <= 5:
                print(f"iteration {it}: A diff {float((y1 - ya).abs().max()):.3e}  B diff {float((y2 - yb).abs().max()):.3e}", flush=True)
print(f"form {form}: {bad} of 200 concurrent pairs differ from the launches run alone")
