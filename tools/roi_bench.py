#!/usr/bin/env python3
"""Time one GOP iteration of the variable-rate training loop (BASELINE.json configs[4]; selfcheck.roi_gop_step) on cuda:0.

    python tools/roi_bench.py [--batch 8] [--size 256] [--frames 7] [--iters 3]
"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    from spatiotemporalentropymodel_amd.losses import PixelwiseRateDistortionLoss
    from spatiotemporalentropymodel_amd.models import stem_roi, stem_roi_i
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.selfcheck import roi_gop_step
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    imodel, pmodel = stem_roi_i().to(dev).train(), stem_roi().to(dev).train()
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opts = configure_optimizers(imodel, args, max_norm=None) + configure_optimizers(pmodel, args, max_norm=None)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    frames = [torch.rand(a.batch, 3, a.size, a.size, device=dev, generator=g) for _ in range(a.frames)]
    qmap = torch.rand(a.batch, 1, a.size, a.size, device=dev, generator=g)
    crit = PixelwiseRateDistortionLoss()
    for _ in range(a.warmup):
        roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        log = roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"roi GOP iteration: batch {a.batch} x {a.frames} frames {a.size}x{a.size}: {dt * 1e3:.1f} ms  "
          f"({a.batch * a.frames / dt:.1f} frames/s), last loss {float(log[-1][0]['loss'].detach()):.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
