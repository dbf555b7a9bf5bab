#!/usr/bin/env python3
"""Dev check of csrc/wgrad_f16x3.hip: weight gradients of the stride-1 STEM layers against fp64 and the fp32-MFMA kernel, timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, B, C, H, W, K, R, timing=True):
    pad = R // 2
    x = torch.randn(B, C, H, W, device=dev)
    dy = torch.randn(B, K, H, W, device=dev)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (K, C, R, R), dy.double().cpu(), padding=pad)
    xp, dyp = F.F16Planes.split(x), F.F16Planes.split(dy)
    splits, elems = F.wgrad_f16x3_plan(x.shape, K, R, R, pad)
    dwp = torch.empty(elems, device=dev)
    F.conv2d_wgrad_f16x3(xp, dyp, K, R, R, pad, dwp, splits)
    dw6 = dwp.view(splits, R * R, K, C).sum(0).permute(1, 2, 0).reshape(K, C, R, R)
    xn, dyn = F.to_nhwc(x), F.to_nhwc(dy)
    dw32, _ = F.conv2d_wgrad(xn, dyn, K, R, R, 1, pad)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    e6, e32 = float((dw6.double().cpu() - ref).abs().max()) / sc, float((dw32.double().cpu() - ref).abs().max()) / sc
    line = f"{name:8s} splits {splits:2d}  err f16x3 {e6:.2e}  fp32 {e32:.2e}"
    # full path (slab sums + bias gradient) against fp64, and the per-tap kernel (wg3_row=1) next to the filter-row form
    dwf, dbf = torch.zeros(K, C, R, R, device=dev), torch.zeros(K, device=dev)
    F.conv2d_wgrad_f16x3_into(xp, dyp, K, R, R, pad, dwf, dbf)
    eb = float((dbf.double().cpu() - dy.double().cpu().sum((0, 2, 3))).abs().max()) / float(dy.double().sum((0, 2, 3)).abs().max())
    ef = float((dwf.double().cpu() - ref).abs().max()) / sc
    line += f"  into {ef:.2e} bias {eb:.2e}"
    with F.tuning(wg3_row=1):
        s1, el1 = F.wgrad_f16x3_plan(x.shape, K, R, R, pad)
        dwp1 = torch.empty(el1, device=dev)
        F.conv2d_wgrad_f16x3(xp, dyp, K, R, R, pad, dwp1, s1)
        d1 = dwp1.view(s1, R * R, K, C).sum(0).permute(1, 2, 0).reshape(K, C, R, R)
        line += f"  vs per-tap {float((d1 - dw6).abs().max()) / sc:.1e}"
        if timing:
            t1 = timeit(lambda: F.conv2d_wgrad_f16x3(xp, dyp, K, R, R, pad, dwp1, s1))
            line += f" ({s1} splits {t1:6.1f} us)"
    if timing:
        t6 = timeit(lambda: F.conv2d_wgrad_f16x3(xp, dyp, K, R, R, pad, dwp, splits))
        t32 = timeit(lambda: F.conv2d_wgrad(xn, dyn, K, R, R, 1, pad, unpack=False, need_db=False))
        gf = 2 * B * H * W * K * C * R * R / 1e9
        line += f"   {t6:7.1f} us ({gf / t6 * 1e3:5.1f} TF)  vs fp32 {t32:7.1f} us ({gf / t32 * 1e3:5.1f} TF)"
    print(line)


only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
_case = case
case = lambda name, *a, **k: _case(name, *a, **k) if only is None or name in only else None
case("small", 2, 64, 9, 11, 96, 3, timing=False)
case("odd", 1, 96, 13, 7, 160, 5, timing=False)
case("w32", 2, 96, 8, 32, 160, 5, timing=False)
case("w16r3", 3, 64, 5, 16, 96, 3, timing=False)
case("w48", 1, 96, 3, 48, 64, 5, timing=False)
case("w16one", 1, 32, 1, 16, 32, 3, timing=False)
case("TPM.0", 16, 192, 16, 16, 256, 5)
case("TPM.2", 16, 256, 16, 16, 320, 5)
case("TPM.4", 16, 320, 16, 16, 384, 5)
case("HE.0", 16, 384, 16, 16, 256, 3)
case("HD.4", 16, 256, 16, 16, 384, 3)
case("CTX", 16, 192, 16, 16, 384, 5)
case("EPM.0", 16, 1152, 16, 16, 768, 1)
case("EPM.2", 16, 768, 16, 16, 576, 1)
case("EPM.4", 16, 576, 16, 16, 384, 1)
