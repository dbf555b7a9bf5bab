#!/usr/bin/env python3
"""Image-tile form of the general f16x3 kernel (csrc/conv_f16x3_img.hip) against the 128-pixel form and fp64 on the STEM layer
shapes at B=16: forward (+ leaky ReLU, planes out) and input gradient (+ DACT); error, time per launch, split sweep.

    python3 tools/debug/f16x3_img_check.py [layer,layer,...] [--sweep]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SL = 0.01
VARIANTS = {"gen128": dict(fx3_gen_img=1), "img": dict(fx3_gen_img=2)}


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


def case(name, B, C, H, W, K, R, taps=0, sweep=False):
    pad = R // 2
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5
    if taps:
        w.view(K, C, R * R)[:, :, taps:] = 0
    b = torch.randn(K, device=dev) * 0.1
    xn = F.to_nhwc(x)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=pad), SL)
    sc = float(ref.abs().max())
    xp = F.F16Planes.split(x)
    wp = F.pack_weight_f16x2_gen(w, taps=taps) if taps else F.pack_weight_f16x2_gen(w)
    gf = 2 * B * H * W * K * C * (taps if taps else R * R) / 1e9
    kw = dict(epi=F.GEN_EPI_LRELU, slope=SL, want_planes=K % 32 == 0)
    if taps:
        kw["taps"] = taps
    outs = {}
    for vn, tune in VARIANTS.items():
        with F.tuning(**tune):
            y, yp = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, pad, **kw)
            torch.cuda.synchronize()
            err = float((y.double().cpu() - ref).abs().max()) / sc
            okp = yp is None or bool(((yp.merge().double() - y.double()).abs() <= y.double().abs() * 2.0 ** -22 + yp.record()[0] * 2.0 ** -25).all())
            t = timeit(lambda: F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, pad, **kw))
            outs[vn] = y
            print(f"{name:8s} fwd   {vn:8s} err {err:.2e} planes_ok={okp}  {t:7.1f} us ({gf / t * 1e3:6.1f} TF fp32-equivalent)", flush=True)
    d = float((outs["img"] - outs["gen128"]).abs().max()) / sc
    print(f"{name:8s} fwd   img vs gen128: {d:.2e} of max")
    if sweep:
        for vn in ("gen128", "img"):
            line = f"{name:8s} fwd   {vn:8s} split sweep:"
            for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
                with F.tuning(fx3_split=sp, **VARIANTS[vn]):
                    t = timeit(lambda: F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, pad, **kw), n=20)
                line += f"  {sp}:{t:.0f}"
            print(line, flush=True)
    if taps or K % 32:
        return
    dy = torch.randn(B, K, H, W, device=dev)
    dref = torch.nn.grad.conv2d_input(x.shape, w.double().cpu(), dy.double().cpu(), padding=pad)
    dref = torch.where(x.double().cpu() > 0, dref, dref * SL)
    dsc = float(dref.abs().max())
    dyp = F.F16Planes.split(dy)
    wpd = F.pack_weight_f16x2_gen(w, flip=True)
    for vn, tune in VARIANTS.items():
        with F.tuning(**tune):
            d6, _ = F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, pad, epi=F.GEN_EPI_DACT, slope=SL, z=xn)
            torch.cuda.synchronize()
            err = float((d6.double().cpu() - dref).abs().max()) / dsc
            t = timeit(lambda: F.conv2d_f16x3_gen(dyp, wpd, None, C, R, R, 1, pad, epi=F.GEN_EPI_DACT, slope=SL, z=xn))
            print(f"{name:8s} dgrad {vn:8s} err {err:.2e}  {t:7.1f} us ({gf / t * 1e3:6.1f} TF)", flush=True)


args = [a for a in sys.argv[1:] if not a.startswith("--")]
only = args[0].split(",") if args else None
sweep = "--sweep" in sys.argv
CASES = [
    ("small", 2, 64, 16, 16, 96, 3, 0), ("odd", 1, 96, 20, 31, 160, 5, 0), ("two", 3, 32, 33, 30, 136, 1, 0), ("ctxs", 2, 64, 17, 29, 96, 5, 12),
    ("TPM.0", 16, 192, 16, 16, 256, 5, 0), ("TPM.2", 16, 256, 16, 16, 320, 5, 0), ("TPM.4", 16, 320, 16, 16, 384, 5, 0),
    ("HE.0", 16, 384, 16, 16, 256, 3, 0), ("HD.4", 16, 256, 16, 16, 384, 3, 0), ("ctx", 16, 192, 16, 16, 384, 5, 12),
    ("EPM.0", 16, 1152, 16, 16, 768, 1, 0), ("EPM.2", 16, 768, 16, 16, 576, 1, 0), ("EPM.4", 16, 576, 16, 16, 384, 1, 0),
]
for cs in CASES:
    if only is None or cs[0] in only:
        case(*cs[:7], taps=cs[7], sweep=sweep and cs[0] not in ("small", "odd", "two", "ctxs"))
