"""GPU: the hipGraph-captured P-frame step (graphs.GraphedPFrameStep) against the eager step it replaces.

Same model state, same inputs, same noise (a device-side noise_source cannot be captured, so both sides use the Philox
stream with the device-side epoch the graph uses): gradients must be bit-identical (same kernels, same order), the
Adam-stepped parameters equal up to the last bit of the bias-correction scalars (host pow vs device pow), the step
counter / learning-rate / fresh-noise mechanics must survive replays, and eager calls after replays must see the
current weights (pack caches invalidated)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(seed=3):
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    stem = SpatioTemporalPriorModel_Res(64, 96).to(dev).train()
    opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    return stem, opt, aux


def _eager_step(stem, opt, aux, crit, y_cur, y_cond, target, epoch):
    from spatiotemporalentropymodel_amd import functional as F
    F.counter_add_(epoch, 1)
    opt.zero_grad(), aux.zero_grad()
    out = stem(y_cur, y_cond)
    oc = crit(out, target)
    oc["loss"].backward()
    gn = opt.grad_norm()
    grad = opt.flat.grad.clone()
    opt.step(norm_is_current=True)
    al = stem.aux_loss()
    al.backward()
    aux.step()
    return out, oc, al, gn, grad


def test_graphed_step_equals_eager_step():
    from spatiotemporalentropymodel_amd.graphs import GraphedPFrameStep
    from spatiotemporalentropymodel_amd.losses import EMLoss
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    ys = [torch.randn(2, 96, 8, 8, device=dev, generator=g) * 3 for _ in range(5)]
    target = torch.empty(2, 3, 128, 128, device=dev)
    crit = EMLoss()
    # eager reference: device-state optimiser + device epoch noise, exactly what the graph captures
    stem_e, opt_e, aux_e = _build()
    opt_e.enable_device_state(), aux_e.enable_device_state()
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    stem_e.entropy_bottleneck.noise_epoch = stem_e.gaussian_conditional.noise_epoch = epoch
    # graphed
    stem_g, opt_g, aux_g = _build()
    gs = GraphedPFrameStep(stem_g, crit, opt_g, aux_g, (128, 128))
    assert torch.equal(opt_e.flat.data, opt_g.flat.data)
    # the graph's warm-up advances the modules' HOST noise offsets before capture; replicate on the eager side so that both
    # draw from the same Philox counters: (offset at capture) + epoch * 2^40
    n_steps = 4
    res_g = []
    for t in range(n_steps):
        if t == 2:                                       # a scheduler halves the learning rate between two replays
            opt_g.param_groups[0]["lr"] *= 0.5
        out, oc, al, gn = gs.step(ys[t + 1], ys[t])
        res_g.append((out["y_hat"].clone(), float(oc["loss"]), float(al), float(gn), opt_g.flat.grad.clone(), opt_g.flat.data.clone()))
    assert opt_g.t == n_steps and aux_g.t == n_steps and int(gs._epoch.item()) == n_steps
    assert int(opt_g._dev["step"].item()) == n_steps
    for t in range(n_steps):
        if t == 2:
            opt_e.param_groups[0]["lr"] *= 0.5
        for m_g, m_e in ((stem_g.entropy_bottleneck, stem_e.entropy_bottleneck), (stem_g.gaussian_conditional, stem_e.gaussian_conditional)):
            m_e._noise_offset = gs._capture_offsets[id(m_g)]           # the graph re-uses the captured offsets on every replay
        out, oc, al, gn, grad = _eager_step(stem_e, opt_e, aux_e, crit, ys[t + 1], ys[t], target, epoch)
        yh, loss, a, n, grad_g, data_g = res_g[t]
        assert torch.equal(out["y_hat"], yh), f"step {t}: y_hat (noise stream) differs"
        if t == 0:
            assert torch.equal(grad, grad_g), "step 0: gradients differ between eager and captured schedule"
        assert abs(float(oc["loss"]) - loss) <= 1e-6 * abs(loss), (t, float(oc["loss"]), loss)
        assert abs(float(gn) - n) <= 1e-5 * n and abs(float(al) - a) <= 1e-6 * abs(a)
        err = float((opt_e.flat.data - data_g).abs().max())
        assert err <= 2e-7 * (t + 1) + 1e-9, (t, err)           # bias-correction scalars: host pow vs device pow, last bit
    # fresh noise on every replay: the same inputs twice give different y_hat (= y_cur + noise)
    o1 = gs.step(ys[1], ys[0])[0]["y_hat"].clone()
    o2 = gs.step(ys[1], ys[0])[0]["y_hat"].clone()
    assert not torch.equal(o1, o2) and float((o1 - o2).abs().max()) <= 1.0
    # eager forward after replays sees the CURRENT weights (pack caches were invalidated)
    stem_g.eval()
    with torch.no_grad():
        a = stem_g(ys[1], ys[0])["likelihoods"]["y"].clone()
        stem_g.engine()._pack_key = None                  # force a repack: must give the same numbers
        b = stem_g(ys[1], ys[0])["likelihoods"]["y"]
    assert torch.equal(a, b)
    # geometry is fixed per instance; host-injected noise is refused
    with pytest.raises(RuntimeError):
        gs.step(ys[1][:1], ys[0][:1])
    stem_x, opt_x, aux_x = _build()
    stem_x.gaussian_conditional.noise_source = lambda shape, device: torch.zeros(shape, device=device)
    with pytest.raises(RuntimeError):
        GraphedPFrameStep(stem_x, crit, opt_x, aux_x, (128, 128)).step(ys[1], ys[0])


def test_graphed_step_state_dict_and_resume():
    """optimizer.state_dict() after replays carries the replayed step count; loading it into a fresh graphed optimiser resumes"""
    from spatiotemporalentropymodel_amd.graphs import GraphedPFrameStep
    from spatiotemporalentropymodel_amd.losses import EMLoss
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(6)
    ys = [torch.randn(1, 96, 4, 4, device=dev, generator=g) * 3 for _ in range(4)]
    stem, opt, aux = _build(7)
    gs = GraphedPFrameStep(stem, EMLoss(), opt, aux, (64, 64))
    for t in range(3):
        gs.step(ys[t + 1], ys[t])
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 3.0
    stem2, opt2, aux2 = _build(7)
    stem2.load_state_dict(stem.state_dict())
    opt2.load_state_dict(sd)
    aux2.load_state_dict(aux.state_dict())
    assert opt2.t == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.flat.data, opt.flat.data)
