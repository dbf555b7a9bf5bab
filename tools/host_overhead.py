#!/usr/bin/env python3
"""How long the host needs to ENQUEUE one bench step (no synchronisation inside) vs how long the GPU needs to run it. Dev tool."""
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


def one_step():
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    for t in range(1, bench.FRAMES):
        out, oc, aux, gn = p_frame_step(imodel, stem, crit, opt, aux_opt, frames[t], y_cond)
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
print(f"host enqueue {t_enq * 1e3:.1f} ms/step, wall {t_all * 1e3:.1f} ms/step")

if os.environ.get("STEM_HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        one_step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
