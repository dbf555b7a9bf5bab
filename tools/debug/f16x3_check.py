#!/usr/bin/env python3
"""Dev check of csrc/conv_f16x3.hip: split/merge exactness, conv(+GDN) against fp64 and against the fp32-MFMA kernel, timing."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def ref64(x, w, b, beta, gamma, stride, pad, beta_min=1e-6):
    x, w, b = x.double().cpu(), w.double().cpu(), b.double().cpu()
    v = torch.nn.functional.conv2d(x, w, b, stride=stride, padding=pad)
    if beta is None:
        return v
    ped = 2.0 ** -36
    bb = torch.clamp(beta.double().cpu(), min=(beta_min + ped) ** 0.5) ** 2 - ped
    gg = torch.clamp(gamma.double().cpu(), min=2.0 ** -18) ** 2 - ped
    nrm = torch.nn.functional.conv2d(v * v, gg[:, :, None, None], bb)
    return v / torch.sqrt(nrm)


def case(B, H, W, C, K, R, stride, gdn, planes_out):
    x = torch.randn(B, C, H, W, device=dev) * 2.0
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    beta = (torch.rand(K, device=dev) + 0.5) if gdn else None
    gamma = (torch.rand(K, K, device=dev) * 0.1 + 0.1 * torch.eye(K, device=dev)) if gdn else None
    xp = F.F16Planes.split(x)
    assert (lambda _p, _y: bool(((_p.merge().double() - _y.double()).abs() <= _y.double().abs() * 2.0 ** -22 + _p.record()[0] * 2.0 ** -25).all()))(xp, F.to_nhwc(x)), "split/merge beyond 2^-22"
    wp = F.pack_weight_f16x2(w)
    out = F.conv2d_f16x3_fwd(xp, wp, b, K, R, R, stride, R // 2, beta, gamma, planes_out=planes_out)
    y = out.merge() if planes_out else out
    r = ref64(x, w, b, beta, gamma, stride, R // 2)
    wp32 = F.pack_weight(w, F.PACK_CONV_FWD)
    y32 = F.conv2d_gdn_fwd(F.to_nhwc(x), wp32, b, beta, gamma, K, R, R, stride, R // 2) if gdn else \
        F.conv2d_fwd(F.to_nhwc(x), wp32, b, K, R, R, stride, R // 2)
    torch.cuda.synchronize()
    scale = float(r.abs().max())
    e6 = float((y.double().cpu() - r).abs().max()) / scale
    e32 = float((y32.double().cpu() - r).abs().max()) / scale
    print(f"B{B} {H}x{W} C{C}->K{K} k{R} s{stride} gdn={int(gdn)} planes_out={int(planes_out)}: max err / max|ref|  f16x3 {e6:.2e}   fp32-MFMA {e32:.2e}")
    return e6, e32


case(1, 16, 16, 64, 64, 3, 1, False, False)
case(2, 20, 28, 192, 192, 5, 2, False, False)
case(2, 20, 28, 192, 192, 5, 2, True, False)
case(2, 20, 28, 192, 192, 5, 2, True, True)
case(1, 33, 47, 96, 160, 5, 2, True, True)
case(3, 16, 16, 192, 192, 5, 2, False, True)

# timing at the g_a.2 size of the bench: B=16, 128x128 -> 64x64
B, H, W, C, K = 16, 128, 128, 192, 192
x = torch.randn(B, C, H, W, device=dev)
w = torch.randn(K, C, 5, 5, device=dev) / (C * 25) ** 0.5
b = torch.randn(K, device=dev) * 0.1
beta = torch.rand(K, device=dev) + 0.5
gamma = torch.rand(K, K, device=dev) * 0.1
xp = F.F16Planes.split(x)
wp = F.pack_weight_f16x2(w)
wp32 = F.pack_weight(w, F.PACK_CONV_FWD)
xn = F.to_nhwc(x)
gp = F.pack_gdn_gamma_f16x2(gamma)          # kept while gamma does not change, as layers._PackCache does
for name, fn in (("f16x3 conv+GDN -> planes", lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta, gamma, planes_out=True, gp=gp)),
                 ("f16x3 conv+GDN -> fp32", lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2, beta, gamma, gp=gp)),
                 ("f16x3 conv only -> fp32", lambda: F.conv2d_f16x3_fwd(xp, wp, b, K, 5, 5, 2, 2)),
                 ("fp32-MFMA conv+GDN", lambda: F.conv2d_gdn_fwd(xn, wp32, b, beta, gamma, K, 5, 5, 2, 2)),
                 ("split 16x128x128x192", lambda: F.F16Planes.split(x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gf = 2 * B * 64 * 64 * K * (C * 25 + (K if "GDN" in name else 0)) / 1e9
    print(f"{name:32s} {dt * 1e6:8.1f} us" + (f"  {gf / dt / 1e3:7.1f} TFLOP/s (fp32-equivalent)" if "split" not in name else ""))
