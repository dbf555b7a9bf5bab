#!/usr/bin/env python3
"""Dev check of the general f16x3 kernel on the TPM layer shapes: forward (+ leaky ReLU) and input-gradient (+ DACT) against
fp64 and against the fp32-MFMA kernels; timing of both."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SL = 0.01


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
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    xn = F.to_nhwc(x)
    # forward + leaky ReLU
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=pad), SL)
    xp = F.F16Planes.split(x)
    wp = F.pack_weight_f16x2_gen(w)
    y, yp = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, pad, epi=F.GEN_EPI_LRELU, slope=SL, want_planes=K % 32 == 0)
    wp32 = F.pack_weight(w, F.PACK_CONV_FWD)
    y32 = F.conv2d_fwd(xn, wp32, b, K, R, R, 1, pad, F.ACT_LRELU, slope=SL)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    e6, e32 = float((y.double().cpu() - ref).abs().max()) / sc, float((y32.double().cpu() - ref).abs().max()) / sc
    okp = yp is None or (lambda _p, _y: bool(((_p.merge().double() - _y.double()).abs() <= _y.double().abs() * 2.0 ** -22 + _p.record()[0] * 2.0 ** -25).all()))(yp, y)
    line = f"{name:8s} fwd  err f16x3 {e6:.2e} fp32 {e32:.2e} planes_ok={okp}"
    if timing:
        t6 = timeit(lambda: F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, pad, epi=F.GEN_EPI_LRELU, slope=SL, want_planes=K % 32 == 0))
        t32 = timeit(lambda: F.conv2d_fwd(xn, wp32, b, K, R, R, 1, pad, F.ACT_LRELU, slope=SL))
        gf = 2 * B * H * W * K * C * R * R / 1e9
        line += f"   {t6:7.1f} us ({gf / t6 * 1e3:5.1f} TF)  vs fp32 {t32:7.1f} us ({gf / t32 * 1e3:5.1f} TF)"
    print(line)
    # input gradient with the activation derivative of the INPUT x (as if x = lrelu(u))
    dy = torch.randn(B, K, H, W, device=dev)
    dref = torch.nn.grad.conv2d_input(x.shape, w.double().cpu(), dy.double().cpu(), padding=pad)
    dref = torch.where(x.double().cpu() > 0, dref, dref * SL)
    dyp = F.F16Planes.split(dy)
    wpd = F.pack_weight_f16x2_gen(w, flip=True)
    d6, _ = F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, pad, epi=F.GEN_EPI_DACT, slope=SL, z=xn)
    wpd32 = F.pack_weight(w, F.PACK_CONV_DGRAD)
    d32 = F.conv2d_dgrad(F.to_nhwc(dy), wpd32, x.shape, K, R, R, 1, pad, xact=xn)
    torch.cuda.synchronize()
    sc = float(dref.abs().max())
    e6, e32 = float((d6.double().cpu() - dref).abs().max()) / sc, float((d32.double().cpu() - dref).abs().max()) / sc
    line = f"{name:8s} dgrad err f16x3 {e6:.2e} fp32 {e32:.2e}"
    if timing:
        dyn = F.to_nhwc(dy)
        t6 = timeit(lambda: F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, pad, epi=F.GEN_EPI_DACT, slope=SL, z=xn))
        t32 = timeit(lambda: F.conv2d_dgrad(dyn, wpd32, x.shape, K, R, R, 1, pad, xact=xn))
        line += f"   {t6:7.1f} us ({gf / t6 * 1e3:5.1f} TF)  vs fp32 {t32:7.1f} us ({gf / t32 * 1e3:5.1f} TF)"
    print(line)


only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
if only is None:
    case("small", 2, 64, 9, 11, 96, 3, timing=False)
    case("odd", 1, 96, 13, 7, 160, 5, timing=False)
_case = case
case = lambda name, *a, **k: _case(name, *a, **k) if only is None or name in only else None
case("TPM.0", 16, 192, 16, 16, 256, 5)
case("TPM.2", 16, 256, 16, 16, 320, 5)
case("TPM.4", 16, 320, 16, 16, 384, 5)
case("HE.0", 16, 384, 16, 16, 256, 3)
case("HD.4", 16, 256, 16, 16, 384, 3)
case("EPM.0", 16, 1152, 16, 16, 768, 1)
case("EPM.2", 16, 768, 16, 16, 576, 1)
case("EPM.4", 16, 576, 16, 16, 384, 1)
