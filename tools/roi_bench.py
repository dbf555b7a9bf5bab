#!/usr/bin/env python3
"""Time GOP iterations of the variable-rate training loop (BASELINE.json configs[4]; selfcheck.roi_gop_step).

    python tools/roi_bench.py [--batch 8] [--size 256] [--frames 7] [--iters 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/roi_bench.py

One process per GPU; with N > 1 each rank trains its own `--batch` GOPs (weak scaling) and the frame gradients are
exchanged through distributed.GopGradAccumulator (RCCL).  Rank 0 prints a JSON line shaped like bench.py's.
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
    from spatiotemporalentropymodel_amd import distributed as D
    rank, world, local = D.init_from_env()
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    imodel, pmodel = stem_roi_i().to(dev).train(), stem_roi().to(dev).train()
    D.broadcast_parameters(imodel)
    D.broadcast_parameters(pmodel)
    for i, m in enumerate((imodel, pmodel)):
        m.entropy_bottleneck.noise_seed = m.gaussian_conditional.noise_seed = D.shard_seed(1234 + 100 * i, rank)
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opts = configure_optimizers(imodel, args, max_norm=None) + configure_optimizers(pmodel, args, max_norm=None)
    acc = D.GopGradAccumulator([opts[0].flat, opts[2].flat], [opts[1].flat, opts[3].flat]) if world > 1 else None
    g = torch.Generator(device=dev)
    g.manual_seed(D.shard_seed(1, rank))
    frames = [torch.rand(a.batch, 3, a.size, a.size, device=dev, generator=g) for _ in range(a.frames)]
    qmap = torch.rand(a.batch, 1, a.size, a.size, device=dev, generator=g)
    crit = PixelwiseRateDistortionLoss()
    for _ in range(a.warmup):
        roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0, accumulator=acc)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        log = roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0, accumulator=acc)
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev) / a.iters
    if rank == 0:
        import json
        print(f"roi GOP iteration: {world} GPU(s), batch {a.batch} x {a.frames} frames {a.size}x{a.size} per GPU: {dt * 1e3:.1f} ms  "
              f"({world * a.batch * a.frames / dt:.1f} frames/s), last loss {float(log[-1][0]['loss'].detach()):.4f}, "
              f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        print(json.dumps({"metric": "frames/s", "value": world * a.batch * a.frames / dt, "unit": "frames/s", "n_gpus": world,
                          "steps": a.iters, "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "configs[4]: stem_roi_i + stem_roi GOP training iteration (I + P frames, BPTT across the "
                                     "GOP, per-frame clip, one step of 4 Adam optimisers)", "per_gpu_batch": a.batch, "frames": a.frames,
                                     "crop": a.size, "parallelism": f"dp{world}"}}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
