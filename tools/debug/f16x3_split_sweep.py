#!/usr/bin/env python3
"""Split-factor sweep of the general f16x3 kernel and the f16x3 weight-gradient kernel on the STEM layer shapes (B = 16, 16x16
latents): time per split factor next to the planner's own choice (fx3_split / wg3_split = 0)."""
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


LAYERS = [("TPM.0", 192, 256, 5), ("TPM.2", 256, 320, 5), ("TPM.4", 320, 384, 5), ("HE.0", 384, 256, 3), ("HD.4", 256, 384, 3),
          ("EPM.0", 1152, 768, 1), ("EPM.2", 768, 576, 1), ("EPM.4", 576, 384, 1), ("g_a.6", 192, 192, 5)]
which = sys.argv[1] if len(sys.argv) > 1 else "gen"
for name, C, K, R in LAYERS:
    B, H, W = 16, 16, 16
    stride = 1
    if name == "g_a.6":
        H = W = 32
        stride = 2
    pad = R // 2
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xp = F.F16Planes.split(x)
    res = {}
    if which == "gen":
        wp = F.pack_weight_f16x2_gen(w)
        for tile in (64, 128):
            res = {}
            for s in [0] + list(range(1, 17)):
                with F.tuning(fx3_split=s, fx3_gen_tile=tile):
                    res[s] = timeit(lambda: F.conv2d_f16x3_gen(xp, wp, b, K, R, R, stride, pad, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=K % 32 == 0))
            best = min((v, k) for k, v in res.items() if k)
            print(f"{name:7s} tile {tile:3d} planner {res[0]:6.1f} us   best split {best[1]:2d}: {best[0]:6.1f} us   " + " ".join(f"{k}:{v:.0f}" for k, v in res.items() if k))
        with F.tuning():
            t0 = timeit(lambda: F.conv2d_f16x3_gen(xp, wp, b, K, R, R, stride, pad, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=K % 32 == 0))
        print(f"{name:7s} library's own plan: {t0:6.1f} us")
        continue
    else:
        if stride != 1:
            continue
        dyp = F.F16Planes.split(torch.randn(B, K, H, W, device=dev))
        dw = torch.zeros_like(w)
        db = torch.zeros(K, device=dev)
        for s in [0] + list(range(1, 17)):
            with F.tuning(wg3_split=s):
                res[s] = timeit(lambda: F.conv2d_wgrad_f16x3_into(xp, dyp, K, R, R, pad, dw, db))
    best = min((v, k) for k, v in res.items() if k)
    print(f"{name:7s} planner {res[0]:6.1f} us   best split {best[1]:2d}: {best[0]:6.1f} us   " + " ".join(f"{k}:{v:.0f}" for k, v in res.items() if k))
