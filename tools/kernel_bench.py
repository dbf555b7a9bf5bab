#!/usr/bin/env python3
"""Per-layer kernel microbenchmark on the MI355X (dev tool; used under rocprofv3 --pmc as well).

    python tools/kernel_bench.py [--only NAME] [--iters N]
Prints, per layer shape of the big config (B=16), the average launch time and TFLOP/s (algorithmic flops)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

LAYERS = {
    # name: (kind, B, C, H, W, K, R, stride, pad)
    "g_a.2": ("conv", 16, 192, 128, 128, 192, 5, 2, 2),
    "g_a.2+gdn": ("conv_gdn", 16, 192, 128, 128, 192, 5, 2, 2),       # the fused kernel bench.py's roofline is quoted on
    "g_a.0+gdn": ("conv_c4_gdn", 16, 3, 256, 256, 192, 5, 2, 2),
    "g_a.4+gdn": ("conv_gdn", 16, 192, 64, 64, 192, 5, 2, 2),
    "g_a.4": ("conv", 16, 192, 64, 64, 192, 5, 2, 2),
    "g_a.6": ("conv", 16, 192, 32, 32, 192, 5, 2, 2),
    "gdn.1": ("gdn", 16, 192, 128, 128, 192, 1, 1, 0),
    "TPM.0": ("conv", 16, 192, 16, 16, 256, 5, 1, 2),
    "TPM.2": ("conv", 16, 256, 16, 16, 320, 5, 1, 2),
    "TPM.4": ("conv", 16, 320, 16, 16, 384, 5, 1, 2),
    "HE.0": ("conv", 16, 384, 16, 16, 256, 3, 1, 1),
    "HE.2": ("conv", 16, 256, 16, 16, 256, 5, 2, 2),
    "HE.4": ("conv", 16, 256, 8, 8, 256, 5, 2, 2),
    "HD.0": ("deconv", 16, 256, 4, 4, 256, 5, 2, 2),
    "HD.2": ("deconv", 16, 256, 8, 8, 256, 5, 2, 2),
    "EPM.0": ("conv", 16, 1152, 16, 16, 768, 1, 1, 0),
    "EPM.2": ("conv", 16, 768, 16, 16, 576, 1, 1, 0),
    "TPM.2.dgrad": ("dgrad", 16, 256, 16, 16, 320, 5, 1, 2),
    "TPM.2.wgrad": ("wgrad", 16, 256, 16, 16, 320, 5, 1, 2),
    "EPM.0.wgrad": ("wgrad", 16, 1152, 16, 16, 768, 1, 1, 0),
}


def run(name, iters):
    kind, B, C, H, W, K, R, st, pd = LAYERS[name]
    dev = "cuda"
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    if kind == "deconv":
        w = torch.randn(C, K, R, R, device=dev) * 0.02
        Ho, Wo = F.deconv_out_hw(H, W, R, R, st, pd, 1)
    else:
        w = torch.randn(K, C, R, R, device=dev) * 0.02
        Ho, Wo = F.conv_out_hw(H, W, R, R, st, pd)
    b = torch.zeros(K, device=dev)
    flop = 2.0 * B * Ho * Wo * K * C * R * R
    if kind == "conv":
        wp = F.pack_weight(w, F.PACK_CONV_FWD)
        fn = lambda: F.conv2d_fwd(x, wp, b, K, R, R, st, pd)
    elif kind == "conv_gdn":
        wp = F.pack_weight(w, F.PACK_CONV_FWD)
        beta, gamma = torch.ones(K, device=dev), (0.1 * torch.eye(K, device=dev) + 0.01).sqrt()
        flop += 2.0 * B * Ho * Wo * K * K + 3.0 * B * Ho * Wo * K
        fn = lambda: F.conv2d_gdn_fwd(x, wp, b, beta, gamma, K, R, R, st, pd)
    elif kind == "conv_c4_gdn":
        wp = F.pack_weight(w, F.PACK_CONV_FWD_C4)
        x4 = torch.rand(B, H, W, 4, device=dev)
        beta, gamma = torch.ones(K, device=dev), (0.1 * torch.eye(K, device=dev) + 0.01).sqrt()
        flop += 2.0 * B * Ho * Wo * K * K + 3.0 * B * Ho * Wo * K
        fn = lambda: F.conv2d_fwd_c4_gdn(x4, wp, b, beta, gamma, K, R, R, st, pd)
    elif kind == "deconv":
        wp = F.pack_weight(w, F.PACK_DECONV_FWD)
        flop = 2.0 * B * H * W * K * C * R * R
        fn = lambda: F.deconv2d_fwd(x, wp, b, K, R, R, st, pd, 1)
    elif kind == "gdn":
        beta, gamma = torch.ones(C, device=dev), (0.1 * torch.eye(C, device=dev) + 0.01).sqrt()
        fn = lambda: F.gdn_fwd(x, beta, gamma)
    elif kind == "dgrad":
        wp = F.pack_weight(w, F.PACK_CONV_DGRAD)
        dy = torch.randn(B, K, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
        fn = lambda: F.conv2d_dgrad(dy, wp, x.shape, K, R, R, st, pd)
    elif kind == "wgrad":
        dy = torch.randn(B, K, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
        fn = lambda: F.conv2d_wgrad(x, dy, K, R, R, st, pd)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:14s} {kind:7s} {ms * 1e3:9.1f} us  {flop / ms / 1e9:7.1f} TFLOP/s  ({flop / 1e9:.1f} GF)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    for n in LAYERS:
        if a.only is None or n in a.only.split(","):
            run(n, a.iters)
