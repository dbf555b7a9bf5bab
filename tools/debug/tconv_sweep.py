#!/usr/bin/env python3
"""Transposed-face launches of the training step (HD.0 / HD.2 forward, HE.4 / HE.2 input gradient at B = 16) alone: time per launch
over the chunks-per-workgroup plan (stem_tuning_set("tconv_cps")), against igemm.hip's fp32-MFMA deconvolution."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, C, H, W, N, R in (("HD.0", 16, 256, 4, 4, 256, 5), ("HD.2", 16, 256, 8, 8, 256, 5)):
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, N, R, R, device=dev) / (C * R * R / 4) ** 0.5
    b = torch.randn(N, device=dev) * 0.1
    xp = F.F16Planes.split(x)
    wp = F.pack_weight_f16x2_tconv(w)
    w32 = F.pack_weight(w, F.PACK_DECONV_FWD)
    xn = F.to_nhwc(x)
    t32 = timeit(lambda: F.deconv2d_fwd(xn, w32, b, N, R, R, 2, 2, 1, F.ACT_LRELU))
    line = f"{name}: igemm fp32 {t32:6.1f} us | tconv planner {timeit(lambda: F.tconv2d_f16x3(xp, wp, b, N, R, epi=F.GEN_EPI_LRELU, want_planes=True)):6.1f} us | cps:"
    for cps in (4, 6, 8, 12, 16, 24, 36, 72):
        with F.tuning(tconv_cps=cps):
            t = timeit(lambda: F.tconv2d_f16x3(xp, wp, b, N, R, epi=F.GEN_EPI_LRELU, want_planes=True))
        line += f"  {cps}:{t:.0f}"
    for tile in (64, 128):
        with F.tuning(fx3_gen_tile=tile):
            t = timeit(lambda: F.tconv2d_f16x3(xp, wp, b, N, R, epi=F.GEN_EPI_LRELU, want_planes=True))
        line += f"  | tile{tile}: {t:.0f}"
    print(line, flush=True)
