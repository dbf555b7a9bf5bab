#!/usr/bin/env python3
"""How long the host needs to ENQUEUE one bench step (no synchronisation inside) vs how long the GPU needs to run it. Dev tool.
STEM_HOST_TINY=1: B=1, 64x64 (the GPU never back-pressures: wall = pure host cost); STEM_HOST_GENERIC=1: the nn.Module / autograd
route (selfcheck.p_frame_step) instead of the explicit fused schedule bench.py runs by default (trainer.FusedPFrameStep)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spatiotemporalentropymodel_amd.losses import EMLoss  # noqa: E402
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.optim import configure_optimizers  # noqa: E402
from spatiotemporalentropymodel_amd.selfcheck import p_frame_step  # noqa: E402
from spatiotemporalentropymodel_amd.zoo import models  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
imodel = models["mbt2018"](quality=4).to(dev).eval()
stem = SpatioTemporalPriorModel_Res().to(dev).train()
opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
crit = EMLoss()
tiny = bool(os.environ.get("STEM_HOST_TINY"))          # B=1, 64x64: the GPU is idle most of the time, wall = pure host cost
frames = bench.synthetic_septuplet(1 if tiny else bench.BATCH, 64 if tiny else bench.SIZE, 1234, dev)


generic = bool(os.environ.get("STEM_HOST_GENERIC"))
if not generic:
    from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep  # noqa: E402
    fused = FusedPFrameStep(stem, opt, aux_opt)
    if os.environ.get("STEM_HOST_TAPE", "1") != "0":    # the native executor (tape.TapedPFrameStep), bench.py's default; 0: plain Python enqueue
        from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
        fused = TapedPFrameStep(fused)
npix = frames[0].shape[0] * frames[0].shape[2] * frames[0].shape[3]


def one_step():
    with torch.no_grad():
        ys = [imodel.getY(f) for f in frames]
    y_cond = ys[0][1]
    for t in range(1, bench.FRAMES):
        if generic:
            out, oc, aux, gn = p_frame_step(imodel, stem, crit, opt, aux_opt, frames[t], y_cond, y_cur=ys[t][0])
        else:
            out, oc, aux, gn = fused.step(ys[t][0], y_cond, npix)
        y_cond = out["y_hat"]


for _ in range(3):
    one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    one_step()
t_enq = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print(f"{'generic' if generic else 'fused'} route{' (tiny problem)' if tiny else ''}: host enqueue {t_enq * 1e3:.1f} ms/step, wall {t_all * 1e3:.1f} ms/step")

if os.environ.get("STEM_HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        one_step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats(os.environ.get("STEM_HOST_PROFILE_SORT", "tottime")).print_stats(int(os.environ.get("STEM_HOST_PROFILE_LINES", "28")))
