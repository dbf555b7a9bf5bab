#!/usr/bin/env python3
"""Per-layer-shape microbenchmark of the variable-rate P model (stem_roi) at training size: every distinct convolution
shape is timed forward / dgrad / wgrad (incl. bias gradient) and weighted by how often it runs in one GOP iteration
(7 forwards, 28 frame-backwards: selfcheck.roi_gop_step).  Dev tool.

    python tools/roi_layer_bench.py [--batch 8] [--size 256] [--top 30]
"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402
from spatiotemporalentropymodel_amd.layers import Conv2d, ConvTranspose2d  # noqa: E402


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    from spatiotemporalentropymodel_amd.models import stem_roi
    dev = torch.device("cuda:0")
    m = stem_roi().to(dev).train()
    shapes = OrderedDict()

    def hook(mod, inp, out):
        x = inp[0]
        key = (type(mod).__name__, x.shape[1], x.shape[2], x.shape[3], mod.out_channels, mod.kernel_size, mod.stride, mod.padding,
               getattr(mod, "output_padding", 0))
        shapes.setdefault(key, []).append(mod)

    for name, mod in m.named_modules():
        if isinstance(mod, (Conv2d, ConvTranspose2d)):
            mod._bench_name = name
            mod.register_forward_hook(hook)
    x = torch.rand(a.batch, 3, a.size, a.size, device=dev)
    q = torch.rand(a.batch, 1, a.size, a.size, device=dev)
    with torch.no_grad():
        m(x, x, q)
    rows = []
    B = a.batch
    for (kind, C, H, W, K, R, st, pd, op), mods in shapes.items():
        n = len(mods)
        xin = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        if C <= 4:
            continue
        if kind == "Conv2d":
            w = torch.randn(K, C, R, R, device=dev) * 0.02
            Ho, Wo = F.conv_out_hw(H, W, R, R, st, pd)
            flop = 2.0 * B * Ho * Wo * K * C * R * R
            wp, wd = F.pack_weight(w, F.PACK_CONV_FWD), F.pack_weight(w, F.PACK_CONV_DGRAD)
            dy = torch.randn(B, K, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
            b = torch.zeros(K, device=dev)
            tf = timeit(lambda: F.conv2d_fwd(xin, wp, b, K, R, R, st, pd))
            td = timeit(lambda: F.conv2d_dgrad(dy, wd, xin.shape, K, R, R, st, pd))
            tw = timeit(lambda: F.conv2d_wgrad(xin, dy, K, R, R, st, pd))
            tw0 = timeit(lambda: F.conv2d_wgrad(xin, dy, K, R, R, st, pd, need_db=False))
        else:
            w = torch.randn(C, K, R, R, device=dev) * 0.02
            Ho, Wo = F.deconv_out_hw(H, W, R, R, st, pd, op)
            flop = 2.0 * B * H * W * K * C * R * R
            wp, wd = F.pack_weight(w, F.PACK_DECONV_FWD), F.pack_weight(w, F.PACK_DECONV_DGRAD)
            dy = torch.randn(B, K, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
            b = torch.zeros(K, device=dev)
            tf = timeit(lambda: F.deconv2d_fwd(xin, wp, b, K, R, R, st, pd, op))
            td = timeit(lambda: F.deconv2d_dgrad(dy, wd, xin.shape, K, R, R, st, pd, op))
            tw = timeit(lambda: F.deconv2d_wgrad(xin, dy, K, R, R, st, pd, op))
            tw0 = timeit(lambda: F.deconv2d_wgrad(xin, dy, K, R, R, st, pd, op, need_db=False))
        total = n * (7 * tf + 28 * (td + tw))
        rows.append((total, kind, C, H, W, K, R, st, n, flop, tf, td, tw, tw0, mods[0]._bench_name))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"sum over conv layers: {tot:.1f} ms per GOP iteration (B={B}, {a.size}x{a.size})")
    print("  total ms  n  kind   C    HxW   K  k s |    GF |  fwd us (TF/s) | dgrad us (TF/s) | wgrad us (TF/s) | wgrad w/o db | first")
    for (total, kind, C, H, W, K, R, st, n, flop, tf, td, tw, tw0, nm) in rows[: a.top]:
        g = flop / 1e9
        print(f"{total:9.1f} {n:2d} {kind[:6]:6s} {C:4d} {H:3d}x{W:<3d} {K:4d} {R} {st} | {g:5.1f} | {tf * 1e3:7.1f} ({g / tf:5.1f}) | "
              f"{td * 1e3:7.1f} ({g / td:5.1f}) | {tw * 1e3:7.1f} ({g / tw:5.1f}) | {tw0 * 1e3:7.1f} | {nm}")


if __name__ == "__main__":
    main()
