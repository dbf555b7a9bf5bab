#!/usr/bin/env python3
"""Soak of the launch tape: 90 P-frame steps (15 GOPs, latents prefetched, tuned schedule) through tape.TapedPFrameStep against the
plain trainer.FusedPFrameStep on the same model and frames: parameters, auxiliary parameters, latents and the logged scalars must
be bit-identical (dev tool; the 9-step version is tests/test_hip_trainer.py)."""
import sys, types, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import trainer, selfcheck as S
from spatiotemporalentropymodel_amd.optim import configure_optimizers
from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
dev = torch.device("cuda:0")
def run(taped, nsteps=90):
    torch.manual_seed(11)
    imodel, stem = S.build_models(64, 96, 64, 96, dev, closed_form=False, inject_noise=False)
    stem.train()
    opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    g = torch.Generator(device=dev).manual_seed(3)
    frames = [torch.rand(2, 3, 256, 256, device=dev, generator=g) for _ in range(7)]
    step = trainer.FusedPFrameStep(stem, opt, aux)
    if taped: step = TapedPFrameStep(step)
    pf = trainer.LatentPrefetcher(imodel)
    sched = trainer.tuned_schedule(dev)
    log = []
    with sched:
        for it in range(nsteps // 6):
            pf.start(frames, frames_ready=True)
            y_cond = pf.get(0)[1]
            for t in range(1, 7):
                out, oc, al, gn = step.step(pf.get(t)[0], y_cond, 2 * 256 * 256)
                y_cond = out["y_hat"]
            log.append((float(oc["loss"]), float(gn), float(al)))
        step.finish()
    torch.cuda.synchronize()
    return opt.flat.data.clone(), aux.flat.data.clone(), y_cond.clone(), log
a = run(False); b = run(True)
print("params equal", torch.equal(a[0], b[0]), "aux equal", torch.equal(a[1], b[1]), "y_hat equal", torch.equal(a[2], b[2]), "logs equal", a[3] == b[3])
print(a[3][-1], b[3][-1])
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
print("SOAK OK")
