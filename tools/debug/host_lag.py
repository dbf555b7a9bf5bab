#!/usr/bin/env python3
"""Is the host ahead of the GPU inside bench.py's default loop (latents prefetched, fused schedule)?  For every P-frame step of
a few bench steps: host clock when step() is entered, GPU clock (event on the compute stream) when the GPU gets there, both
relative to the start of the measurement.  lag = gpu - host: ~0 means the GPU waits for the host's launches (host-bound at that
point), large means the host is running ahead."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.optim import configure_optimizers  # noqa: E402
from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep, LatentPrefetcher  # noqa: E402
from spatiotemporalentropymodel_amd.zoo import models  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
imodel = models["mbt2018"](quality=4).to(dev).eval()
stem = SpatioTemporalPriorModel_Res().to(dev).train()
opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
frames = bench.synthetic_septuplet(bench.BATCH, bench.SIZE, 1234, dev)
reducer = None
if os.environ.get("STEM_DIST_SINGLE"):                       # one rank in a real RCCL group: the data-parallel path's host cost
    from spatiotemporalentropymodel_amd import distributed as D  # noqa: E402
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    D.init_from_env(single=True)
    reducer = D.OverlappedGradReducer(opt.flat).attach(stem.engine())
fused = FusedPFrameStep(stem, opt, aux_opt)
if os.environ.get("STEM_HOST_TAPE", "1") != "0":           # the native executor (bench.py's default); STEM_HOST_TAPE=0: the Python schedule
    from spatiotemporalentropymodel_amd.tape import TapedPFrameStep  # noqa: E402
    fused = TapedPFrameStep(fused)
prefetch = LatentPrefetcher(imodel, ahead=1)
npix = bench.BATCH * bench.SIZE * bench.SIZE
marks = []


def one_step(record):
    prefetch.start(frames, frames_ready=True)
    y_cond = prefetch.get(0)[1]
    for t in range(1, bench.FRAMES):
        y_cur = prefetch.get(t)[0]
        if record:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((time.perf_counter(), e))
        out, oc, aux, gn = fused.step(y_cur, y_cond, npix, reducer=reducer)
        y_cond = out["y_hat"]


for _ in range(3):
    one_step(False)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
t0 = time.perf_counter()
NSTEPS = int(os.environ.get("STEPS", 3))
for _ in range(NSTEPS):
    one_step(True)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{NSTEPS} bench steps: host enqueue {t_enq * 1e3:.1f} ms, wall {t_all * 1e3:.1f} ms")
for i, (th, e) in enumerate(marks):
    tg = e0.elapsed_time(e)
    print(f"P-step {i:2d}: host {1e3 * (th - t0):7.2f} ms   gpu {tg:7.2f} ms   lag {tg - 1e3 * (th - t0):6.2f} ms")
if os.environ.get("STEM_HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        one_step(False)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
