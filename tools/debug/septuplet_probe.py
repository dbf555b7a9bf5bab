#!/usr/bin/env python3
"""why do SeptupletTrainer(route=taped, prefetch) and (route=fused, latents first) differ at the first step?"""
import os, sys, types, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatiotemporalentropymodel_amd import selfcheck as S
from spatiotemporalentropymodel_amd.optim import configure_optimizers
from spatiotemporalentropymodel_amd.trainer import SeptupletTrainer

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
items = [[torch.rand(1, 3, 64, 64, device=dev, generator=g) for _ in range(7)] for _ in range(2)]
torch.cuda.synchronize()


def pair():
    torch.manual_seed(11)
    imodel, stem = S.build_models(64, 96, 64, 96, dev, closed_form=False, inject_noise=False)
    stem.train()
    opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    for i, m in enumerate((imodel.gaussian_conditional, stem.entropy_bottleneck, stem.gaussian_conditional)):
        m.noise_seed = 1000 + i
    return imodel, stem, opt, aux


for route, prefetch, ready in (("fused", False, True), ("fused", True, True), ("fused", True, False), ("taped", True, True), ("taped", False, True)):
    im, stem, opt, aux = pair()
    tr = SeptupletTrainer(im, stem, opt, aux, route=route, prefetch=prefetch, rng=random.Random(5))
    seen = []
    tr.on_step = lambda t, out, oc, a, gn: seen.append((float(out["y_hat"].abs().sum()), float(oc["loss"]), float(gn)))
    for frames in items:
        tr.train_septuplet(frames, rand=0.1, frames_ready=ready)
    tr.finish()
    torch.cuda.synchronize()
    print(route, prefetch, ready, [tuple(round(v, 9) for v in s) for s in seen[:4]], float(opt.flat.data.double().sum()))
