"""GPU: the explicit-schedule P-frame step (trainer.FusedPFrameStep: fused glue kernels, no autograd) against the generic
nn.Module / autograd / optimiser route (selfcheck.p_frame_step) that the reference's goldens pin.

Same weights, same inputs, same noise (both draw the same Philox counters, or both are fed the goldens' injected noise):
the forward tensors are bit-identical, gradients agree to fp32 rounding of the loss-gradient scalar (coef / lik vs
(1 / lik) * (g / ln 2)), and the optimiser trajectories stay together."""
import types

import numpy as np
import pytest
import torch

from conftest import f64_gate

pytestmark = pytest.mark.gpu


def _pair(ebc, cin, N, M, closed_form, inject):
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    imodel, stem = S.build_models(ebc, cin, N, M, dev, closed_form=closed_form, inject_noise=inject)
    stem.train()
    opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    return imodel, stem, opt, aux


def test_fused_step_tracks_generic_step_philox_noise():
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.losses import EMLoss
    from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(4)]
    im_a, stem_a, opt_a, aux_a = _pair(64, 96, 64, 96, False, False)
    im_b, stem_b, opt_b, aux_b = _pair(64, 96, 64, 96, False, False)
    assert torch.equal(opt_a.flat.data, opt_b.flat.data)
    fused = FusedPFrameStep(stem_b, opt_b, aux_b)
    assert fused.overwrite_grads                          # the gradients stay in place after a step: step 1's are inspected below
    fused.clear_grad_in_adam = False
    crit = EMLoss()
    with torch.no_grad():
        _, y_cond_a = im_a.getY(frames[0])
        _, y_cond_b = im_b.getY(frames[0])
    assert torch.equal(y_cond_a, y_cond_b)
    for t in range(1, 4):
        out_a, oc_a, aux_la, gn_a = S.p_frame_step(im_a, stem_a, crit, opt_a, aux_a, frames[t], y_cond_a)
        grad_a = opt_a.flat.grad.clone()
        with torch.no_grad():
            y_cur, _ = im_b.getY(frames[t])
        out_b, oc_b, aux_lb, gn_b = fused.step(y_cur, y_cond_b, 2 * 128 * 128)
        if t == 2:
            fused.clear_grad_in_adam = True               # the default: Adam clears the buffer, the next step skips its memset
        if t == 1:          # identical parameters: identical forward, down to the noise
            assert torch.equal(out_a["y_hat"], out_b["y_hat"])
            assert torch.equal(out_a["likelihoods"]["y"], out_b["likelihoods"]["y"])
            assert torch.equal(out_a["likelihoods"]["z"], out_b["likelihoods"]["z"])
            assert abs(float(oc_a["loss"]) - float(oc_b["loss"])) <= 1e-12 * abs(float(oc_b["loss"]))
            e = float((grad_a - opt_b.flat.grad).abs().max()) / float(grad_a.abs().max())
            assert e <= 2e-6, e
        assert abs(float(oc_a["loss"]) - float(oc_b["loss"])) <= 1e-5 * abs(float(oc_b["loss"])), t
        assert abs(float(oc_a["y_bpp_loss"]) - float(oc_b["y_bpp_loss"])) <= 1e-5 * abs(float(oc_b["y_bpp_loss"]))
        assert abs(float(gn_a) - float(gn_b)) <= 1e-4 * float(gn_b), (t, float(gn_a), float(gn_b))
        assert abs(float(aux_la) - float(aux_lb)) <= 1e-5 * abs(float(aux_lb)), (t, float(aux_la), float(aux_lb))
        y_cond_a, y_cond_b = out_a["y_hat"], out_b["y_hat"]
    # three Adam steps later the parameters are still together (Adam's normalised update may flip noise-level elements)
    fused.finish()                       # the auxiliary stream's last update of the quantiles, before they are read here
    err = (opt_a.flat.data - opt_b.flat.data).abs()
    assert float(err.max()) <= 6.3e-4 and float((err <= 2e-6).float().mean()) >= 0.97, (float(err.max()), float((err <= 2e-6).float().mean()))
    assert float((aux_a.flat.data - aux_b.flat.data).abs().max()) <= 1e-4
    assert opt_b.t == 3 and aux_b.t == 3
    assert float(opt_b.flat.grad.abs().max()) > 0.0       # overwrite mode (the default): nothing clears the buffer, the last step's gradients stay


@pytest.mark.parametrize("tag", ["small", "big"])
def test_fused_step_against_reference_fixtures(golden, tag):
    """The fused route on the reference's own training case (injected noise, closed-form weights): loss, exact gradient norm
    and every parameter gradient of step 1 against the float64 run of the reference (tests/golden/stem_f64.npz) at 1e-4."""
    from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    g, f64 = golden(f"stem_train_{tag}.npz"), golden("stem_f64.npz")
    ebc, cin, N, M, batch, size, steps = (int(v) for v in g["cfg"])
    dev = torch.device("cuda:0")
    imodel, stem, opt, aux = _pair(ebc, cin, N, M, True, True)
    fused = FusedPFrameStep(stem, opt, aux)
    frames = [f.to(dev) for f in smooth_frames("train:" + tag, batch, steps + 1, size)]
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
        y_cur, _ = imodel.getY(frames[1])
    # look at the gradients before Adam consumes them: run the schedule by hand up to the norm
    eng = stem.engine()
    opt.flat.zero_grad()
    npix = batch * size * size
    y_hat, lik_y, lik_z, k = eng.forward(y_cur, y_cond, True, rate_coef=(-1.0 / (np.log(2.0) * npix), -1.0 / npix))
    eng.backward(k, k["dlik_y"], k["dlik_z"])
    gn = float(opt.grad_norm())
    x_loss, x_ybpp, x_zbpp, _, x_gn = f64[f"{tag}:s1:scalars"]
    r32 = f64[f"{tag}:ref32:s1:scalars"]
    f64_gate([float(k["loss3"][2]), float(k["loss3"][0]), float(k["loss3"][1]), gn], [x_loss, x_ybpp, x_zbpp, x_gn], r32[[0, 1, 2, 4]],
             f"{tag} fused step: loss / y_bpp / z_bpp / grad norm", floor=0.0)
    f64_gate(lik_y.cpu().contiguous().numpy(), f64[f"{tag}:s1:lik_y"], f64[f"{tag}:ref32:lik_y"], f"{tag} fused lik_y", atol=1e-9)
    clip = min(1.0, 1.0 / (gn + 1e-6))
    worst = 0.0
    for name, p in stem.named_parameters():
        if name.endswith(".quantiles"):
            continue
        gd = p.grad.double() * clip
        ex = f64[f"{tag}:s1:gsum:{name}"]
        assert abs(float(gd.sum()) - ex[0]) <= 1e-4 * ex[1] + 1e-12, name
        rms = float(np.sqrt(ex[2] / p.numel()))
        xs = f64[f"{tag}:s1:gslice:{name}"]
        sl = gd.reshape(-1)[:: max(1, gd.numel() // 64)][:64].cpu().numpy()
        e = float((np.abs(sl - xs) / np.maximum(np.abs(xs), rms)).max())
        worst = max(worst, e)
        assert e <= 1e-4, (name, e)
    print(f"[f64 gate] {tag} fused-step gradients: HIP vs exact {worst:.2e}   reference-fp32 vs exact "
          f"{float(f64[f'{tag}:ref32:grad_slice'][0]):.2e}   bound 1e-04")
    # and the whole step() (a fresh pair: the hand-run above consumed noise draws)
    imodel2, stem2, opt2, aux2 = _pair(ebc, cin, N, M, True, True)
    with torch.no_grad():
        _, y_cond2 = imodel2.getY(frames[0])
        y_cur2, _ = imodel2.getY(frames[1])
    fused2 = FusedPFrameStep(stem2, opt2, aux2)
    out, oc, aux_l, gnl = fused2.step(y_cur2, y_cond2, npix)
    fused2.finish()
    aux_ref = g["s1:scalars"][3]
    assert abs(float(oc["loss"]) - x_loss) <= 1e-4 * x_loss and abs(float(gnl) - x_gn) <= 1e-4 * x_gn
    assert abs(float(aux_l) - aux_ref) <= 1e-4 * abs(aux_ref), (float(aux_l), aux_ref)
    np.testing.assert_allclose(stem2.entropy_bottleneck.quantiles.grad.cpu().numpy(), g["s1:dquantiles"], rtol=1e-4, atol=1e-6)


def test_latent_prefetcher_equals_direct_calls():
    """trainer.LatentPrefetcher computes getY of the frames on its own stream, ahead of the consumer: same latents as direct
    calls (the transform is deterministic), noise within [-1/2, 1/2], every frame delivered in order for ahead = 1 and 3."""
    from spatiotemporalentropymodel_amd.trainer import LatentPrefetcher
    from spatiotemporalentropymodel_amd.zoo import models
    torch.manual_seed(4)
    d = torch.device("cuda:0")
    imodel = models["mbt2018"](quality=4).to(d).eval()
    frames = [torch.rand(2, 3, 128, 128, device=d) for _ in range(5)]
    with torch.no_grad():
        direct = [imodel.getY(f)[0].clone() for f in frames]
    for ahead in (1, 3):
        pf = LatentPrefetcher(imodel, ahead=ahead).start(frames)
        for t in range(len(frames)):
            y, yq = pf.get(t)
            scratch = torch.zeros(1 << 20, device=d).sum()          # unrelated work on the consumer stream between the gets
            assert torch.equal(y, direct[t]), (ahead, t)
            assert float((yq - y).abs().max()) <= 0.5
            del scratch
    torch.cuda.synchronize()


def test_latent_prefetcher_across_sequences_equals_direct_calls():
    """start(frames, next_frames=following): the following sequence's first two latents are computed during this sequence's last
    two consumer steps and adopted by start(following).  Same call order as the unpipelined loop, so y AND the noisy copy (the
    noise counters advance per call) equal direct sequential calls bit for bit -- three sequences, the last without a successor."""
    from spatiotemporalentropymodel_amd.trainer import LatentPrefetcher
    from spatiotemporalentropymodel_amd.zoo import models
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(3)
    seqs = [[torch.rand(2, 3, 128, 128, device=d, generator=g) for _ in range(5)] for _ in range(3)]
    runs = []
    for piped in (False, True):
        torch.manual_seed(4)
        imodel = models["mbt2018"](quality=4).to(d).eval()
        imodel.gaussian_conditional.noise_seed = 77
        out = []
        if not piped:
            with torch.no_grad():
                for fr in seqs:
                    out.append([tuple(t.clone() for t in imodel.getY(f)) for f in fr])
        else:
            pf = LatentPrefetcher(imodel)
            for i, fr in enumerate(seqs):
                nxt = seqs[i + 1] if i + 1 < len(seqs) else None
                pf.start(fr, frames_ready=True, next_frames=nxt)
                if i > 0:
                    assert pf._next >= 2                     # frames 0 and 1 were adopted, not recomputed
                got = [pf.get(0)]
                for t in range(1, len(fr)):
                    got.append(pf.get(t))
                    if nxt is not None and t == len(fr) - 1:
                        assert pf._nxt["n"] == 2
                torch.cuda.synchronize()
                out.append([(y.clone(), yq.clone()) for y, yq in got])
        runs.append(out)
    for sa, sb in zip(*runs):
        for (ya, qa), (yb, qb) in zip(sa, sb):
            assert torch.equal(ya, yb) and torch.equal(qa, qb)


def test_tuned_schedule_changes_no_result(monkeypatch):
    """trainer.tuned_schedule (what bench.py runs under: step streams at high priority on a dedicated compute stream, the
    prefetch stream masked to 192 CUs through hipExtStreamCreateWithCUMask) is scheduling only: three P-frame steps with the
    latents prefetched give bit-identical parameters, losses and latents with and without it."""
    from spatiotemporalentropymodel_amd import functional as F
    from spatiotemporalentropymodel_amd import trainer
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(4)]
    results = []
    for tuned in (False, True):
        for k in trainer.SCHEDULE_DEFAULTS:
            monkeypatch.delenv(k, raising=False)
        if not tuned:
            for k in trainer.SCHEDULE_DEFAULTS:
                monkeypatch.setenv(k, "")                     # explicitly none: every stream at one priority, no mask
        monkeypatch.setattr(F, "_STREAM_PRIO", None)          # parsed again by the next make_stream
        imodel, stem, opt, aux = _pair(64, 96, 64, 96, False, False)
        sched = trainer.tuned_schedule(dev)
        if tuned:
            assert F._STREAM_PRIO == {"latents": 0, "side": -1, "compute": -1} and F._cu_mask("latents") is not None
        else:
            assert F._STREAM_PRIO == {} and F._cu_mask("latents") is None
        fused = trainer.FusedPFrameStep(stem, opt, aux)
        pf = trainer.LatentPrefetcher(imodel)
        losses = []
        torch.cuda.synchronize()
        with sched:
            pf.start(frames, frames_ready=True)
            y_cond = pf.get(0)[1]
            for t in range(1, 4):
                out, oc, aux_l, gn = fused.step(pf.get(t)[0], y_cond, 2 * 128 * 128)
                losses.append((float(oc["loss"]), float(gn), float(aux_l)))
                y_cond = out["y_hat"]
            fused.finish()
        torch.cuda.synchronize()
        if tuned:
            assert isinstance(pf._stream, torch.cuda.ExternalStream)       # the masked stream (hipExtStreamCreateWithCUMask)
        results.append((opt.flat.data.clone(), aux.flat.data.clone(), y_cond.clone(), losses))
    a, b = results
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == b[3], (a[3], b[3])


@pytest.mark.parametrize("tuned", [False, True])
def test_taped_step_is_bit_identical_to_the_fused_schedule(monkeypatch, tuned):
    """tape.TapedPFrameStep (the native executor: the schedule recorded once, re-issued by stem_tape_replay) against
    trainer.FusedPFrameStep over nine P-frame steps -- two ordinary, two recorded, five replayed -- with the latents prefetched
    on their own stream: identical losses, gradient norms, auxiliary losses, latents and parameters, bit for bit, with and without
    the tuned schedule (stream priorities / CU mask).  Also: the counters the replays advance on the host side (Adam's step, the
    noise offsets) match the ordinary run's."""
    from spatiotemporalentropymodel_amd import functional as F
    from spatiotemporalentropymodel_amd import trainer
    from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(10)]
    results = []
    for taped in (False, True):
        for k in trainer.SCHEDULE_DEFAULTS:
            monkeypatch.delenv(k, raising=False)
            if not tuned:
                monkeypatch.setenv(k, "")
        monkeypatch.setattr(F, "_STREAM_PRIO", None)
        imodel, stem, opt, aux = _pair(64, 96, 64, 96, False, False)
        sched = trainer.tuned_schedule(dev)
        step = trainer.FusedPFrameStep(stem, opt, aux)
        if taped:
            step = TapedPFrameStep(step)
        pf = trainer.LatentPrefetcher(imodel)
        log = []
        torch.cuda.synchronize()
        with sched:
            pf.start(frames, frames_ready=True)
            y_cond = pf.get(0)[1]
            for t in range(1, 10):
                out, oc, aux_l, gn = step.step(pf.get(t)[0], y_cond, 2 * 128 * 128)
                log.append((float(oc["loss"]), float(oc["y_bpp_loss"]), float(gn), float(aux_l), out["y_hat"].clone(),
                            out["likelihoods"]["y"].clone()))
                y_cond = out["y_hat"]
            step.finish()
        torch.cuda.synchronize()
        eb, gc = stem.entropy_bottleneck, stem.gaussian_conditional
        results.append((opt.flat.data.clone(), aux.flat.data.clone(), log, (opt.t, aux.t, eb._noise_offset, gc._noise_offset)))
        if taped:
            assert step.tape is not None and step.replays == 6 and len(step.tape) > 60 and step.tape.dynamic_args >= 3
    a, b = results
    for t, (la, lb) in enumerate(zip(a[2], b[2])):
        assert la[:4] == lb[:4], (t, la[:4], lb[:4])
        assert torch.equal(la[4], lb[4]) and torch.equal(la[5], lb[5]), t
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[3] == b[3], (a[3], b[3])


def _taped_run(taped, steps, make, frames, npix, edit=None, tuned=True, monkeypatch=None, mutate=None):
    """`steps` P-frame steps with the latents prefetched, through trainer.FusedPFrameStep (taped=False) or tape.TapedPFrameStep;
    edit(t, opt, aux) runs after step t (a scheduler's hook); mutate(step_object) before the first step.
    -> (parameters, aux parameters, per-step log, counters, the step object)"""
    from spatiotemporalentropymodel_amd import functional as F
    from spatiotemporalentropymodel_amd import trainer
    from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
    dev = frames[0].device
    if monkeypatch is not None:
        for k in trainer.SCHEDULE_DEFAULTS:
            monkeypatch.delenv(k, raising=False)
            if not tuned:
                monkeypatch.setenv(k, "")
        monkeypatch.setattr(F, "_STREAM_PRIO", None)
    imodel, stem, opt, aux = make()
    sched = trainer.tuned_schedule(dev)
    step = trainer.FusedPFrameStep(stem, opt, aux)
    if mutate is not None:
        mutate(step, opt, aux)
    if taped:
        step = TapedPFrameStep(step)
    pf = trainer.LatentPrefetcher(imodel)
    log = []
    torch.cuda.synchronize()
    with sched:
        t = 0
        while t < steps:
            pf.start(frames, frames_ready=True)
            y_cond = pf.get(0)[1]
            for f in range(1, len(frames)):
                if t == steps:
                    break
                out, oc, aux_l, gn = step.step(pf.get(f)[0], y_cond, npix)
                t += 1
                log.append((float(oc["loss"]), float(oc["y_bpp_loss"]), float(gn), float(aux_l), out["y_hat"].clone(),
                            out["likelihoods"]["y"].clone()))
                y_cond = out["y_hat"]
                if edit is not None:
                    edit(t, opt, aux)
        step.finish()
    torch.cuda.synchronize()
    eb, gc = stem.entropy_bottleneck, stem.gaussian_conditional
    return opt.flat.data.clone(), aux.flat.data.clone(), log, (opt.t, aux.t, eb._noise_offset, gc._noise_offset), step


def _assert_same_run(a, b):
    for t, (la, lb) in enumerate(zip(a[2], b[2])):
        assert la[:4] == lb[:4], (t, la[:4], lb[:4])
        assert torch.equal(la[4], lb[4]) and torch.equal(la[5], lb[5]), t
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[3] == b[3], (a[3], b[3])


def test_taped_step_follows_the_scheduler(monkeypatch):
    """stem/trainSTEM.py:123,290: ReduceLROnPlateau rewrites param_groups[0]["lr"] while training runs.  Nine P-frame steps --
    two ordinary, two recorded, five replayed -- with the main learning rate cut to a tenth after step 6 (by a real
    ReduceLROnPlateau attached to the fused optimiser), the aux learning rate halved after step 7 and max_norm tightened after
    step 8: the taped run equals the untaped one bit for bit, and differs from a taped run nobody edited (so the replays did
    read the new values)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(10)]
    scheds = {}

    def edit(t, opt, aux):
        if t == 1:
            scheds[id(opt)] = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", patience=0)
            scheds[id(opt)].step(1.0)
        if t == 6:
            scheds[id(opt)].step(2.0)               # no improvement with patience 0: lr x 0.1
            assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12
        if t == 7:
            aux.param_groups[0]["lr"] *= 0.5
        if t == 8:
            opt.max_norm = 0.05

    make = lambda: _pair(64, 96, 64, 96, False, False)
    plain = _taped_run(False, 9, make, frames, 2 * 128 * 128, edit, monkeypatch=monkeypatch)
    taped = _taped_run(True, 9, make, frames, 2 * 128 * 128, edit, monkeypatch=monkeypatch)
    still = _taped_run(True, 9, make, frames, 2 * 128 * 128, None, monkeypatch=monkeypatch)
    assert taped[4].taped and taped[4].replays == 6 and taped[4].refused is None
    assert len(taped[4].tape.float_bindings) == 2            # the two optimiser launches
    _assert_same_run(plain, taped)
    assert not torch.equal(still[0], taped[0]) and not torch.equal(still[1], taped[1])
    for t in range(6):                                       # ... and identical up to the first edit
        assert still[2][t][:4] == taped[2][t][:4]


def test_taped_step_holds_the_memset_when_adam_does_not_clear(monkeypatch):
    """`clear_grad_in_adam = False` (gradients left in place for inspection, cleared at the start of the next step): the clearing is
    a library call (stem_zero_bytes), so the tape holds it -- the bias gradients, which ACCUMULATE into the flat buffer, would
    otherwise grow from replay to replay."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(19)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(9)]

    def keep_grads(step, opt, aux):
        step.overwrite_grads = False                    # the clear-then-accumulate form, clearing at the start of the next step
        step.clear_grad_in_adam = False
    make = lambda: _pair(64, 96, 64, 96, False, False)
    plain = _taped_run(False, 8, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=keep_grads)
    taped = _taped_run(True, 8, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=keep_grads)
    assert taped[4].taped and taped[4].replays == 5
    assert any(e[0] == "call" and e[1] == "stem_zero_bytes" for e in taped[4].tape.entries)
    _assert_same_run(plain, taped)


def test_overwriting_backward_equals_clear_then_accumulate(monkeypatch):
    """FusedPFrameStep's default backward OVERWRITES the flat gradient buffer (every gradient is produced exactly once per step:
    no clearing pass, no read of the old value); `overwrite_grads = False` clears (inside the Adam pass, or with a memset) and
    accumulates as autograd does.  Same losses, norms, latents and parameters bit for bit over five steps -- untaped and taped."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(31)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(7)]
    make = lambda: _pair(64, 96, 64, 96, False, False)

    def accumulate(step, opt, aux):
        step.overwrite_grads = False

    def accumulate_memset(step, opt, aux):
        step.overwrite_grads = False
        step.clear_grad_in_adam = False
    ref = _taped_run(False, 6, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch)
    for taped, mutate in ((False, accumulate), (False, accumulate_memset), (True, None), (True, accumulate)):
        got = _taped_run(taped, 6, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=mutate)
        _assert_same_run(ref, got)
        assert not taped or got[4].taped


def test_taped_step_refuses_what_it_cannot_replay(monkeypatch):
    """Schedules a tape cannot hold keep running on the ordinary route, loudly, with the ordinary route's results: (1) torch
    operators on device memory inside the step (here: a reducer that is NOT announced through functional.tape_py scales the
    gradient buffer with Tensor.mul_) are seen by the recording's guard; (2) device-resident optimiser state; (3) latents that are
    neither NHWC nor contiguous."""
    import warnings
    from spatiotemporalentropymodel_amd import trainer
    from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(23)
    frames = [torch.rand(2, 3, 128, 128, device=dev, generator=g) for _ in range(7)]
    make = lambda: _pair(64, 96, 64, 96, False, False)

    def sneaky(step, opt, aux):                     # a torch op in the middle of the step, outside the library and outside tape_py
        real = step.eng.backward

        def backward(*a, **kw):
            r = real(*a, **kw)
            from spatiotemporalentropymodel_amd.layers import join_wgrad_stream
            join_wgrad_stream()
            opt.flat.grad.mul_(0.5)
            return r
        step.eng.backward = backward
    plain = _taped_run(False, 6, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=sneaky)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        taped = _taped_run(True, 6, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=sneaky)
    assert not taped[4].taped and "aten::mul_" in taped[4].refused, taped[4].refused
    assert any("ordinary schedule" in str(x.message) for x in w)
    _assert_same_run(plain, taped)

    def dev_state(step, opt, aux):
        opt.enable_device_state()
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        taped = _taped_run(True, 5, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=dev_state)
    assert not taped[4].taped and "device memory" in taped[4].refused
    plain = _taped_run(False, 5, make, frames, 2 * 128 * 128, monkeypatch=monkeypatch, mutate=dev_state)
    _assert_same_run(plain, taped)

    imodel, stem, opt, aux = make()
    step = TapedPFrameStep(trainer.FusedPFrameStep(stem, opt, aux))
    with torch.no_grad():
        y0, y1 = imodel.getY(frames[0])[1], imodel.getY(frames[1])[0]
    odd = torch.empty(y1.shape[0], y1.shape[1], y1.shape[2], 2 * y1.shape[3], device=dev)[:, :, :, ::2]
    odd.copy_(y1)
    assert not odd.is_contiguous()
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        for _ in range(5):
            step.step(odd, y0, 2 * 128 * 128)
    step.finish()
    torch.cuda.synchronize()
    assert not step.taped and "neither NHWC nor contiguous" in step.refused


def test_taped_step_at_the_bench_geometry_is_bit_identical(monkeypatch):
    """The route bench.py times, at ITS geometry: SpatioTemporalPriorModel_Res() (N = M = 192, 256-channel hyper path) on B = 16
    septuplets of 256 x 256 with the latent prefetcher and the tuned schedule -- where the planner picks the image-tile / ring
    kernels, split-K workspaces and the multi-tensor tables the tape keeps pointers into.  Twelve P-frame steps = two septuplets
    (two ordinary, two recorded, eight replayed, the prefetcher restarting in between) taped against untaped: losses, norms,
    y_hat, likelihoods and every parameter bit for bit."""
    import types
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.zoo import models
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(29)
    frames = [torch.rand(16, 3, 256, 256, device=dev, generator=g) for _ in range(7)]

    def make():
        torch.manual_seed(1234)
        imodel = models["mbt2018"](quality=4).to(dev).eval()
        stem = SpatioTemporalPriorModel_Res().to(dev).train()
        for m in (imodel.gaussian_conditional, stem.entropy_bottleneck, stem.gaussian_conditional):
            m.noise_seed = 4242
        opt, aux = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
        return imodel, stem, opt, aux
    plain = _taped_run(False, 12, make, frames, 16 * 256 * 256, monkeypatch=monkeypatch)
    taped = _taped_run(True, 12, make, frames, 16 * 256 * 256, monkeypatch=monkeypatch)
    assert taped[4].taped and taped[4].replays == 9 and taped[4].refused is None and len(taped[4].tape) > 60
    _assert_same_run(plain, taped)
    assert plain[3][0] == 12


# the switches of config.StemRuntimeConfig that stay supported next to the defaults (round 4: the others were deleted)
_SCHEDULE_SWITCHES = [("overlap_wgrad", False), ("branch_streams", False), ("split_pack", False), ("defer_bias_final", False),
                      ("tpm_first", False), ("tpm_first_bwd", False), ("tpm_wgrad_inline", False), ("pack_first", True), ("pack_pair", False), ("share_in_planes", False), ("ctx_split_on_side", False)]
_ROUTE_SWITCHES = [("epm_dgrad_by_prior", True), ("fuse_gc_backward", False), ("use_wg3", False), ("use_fx3t", False), ("use_fx3s", False), ("use_ctx3", False), ("use_records", False), ("use_fx3", False)]


def _two_steps(monkeypatch, switch):
    """two P-frame steps of the fused schedule at 16x16 latents (the image-tile kernel and the filter-row weight gradient are the
    forms that run there) with one engine switch away from its default -> (parameters, aux parameters, y_hat, losses, norms)"""
    from spatiotemporalentropymodel_amd import engine as E
    from spatiotemporalentropymodel_amd import trainer
    dev = torch.device("cuda:0")
    if switch is not None:
        monkeypatch.setattr(E.StemEngine, switch[0], switch[1])
    g = torch.Generator(device=dev).manual_seed(9)
    frames = [torch.rand(2, 3, 256, 256, device=dev, generator=g) for _ in range(3)]
    imodel, stem, opt, aux = _pair(64, 96, 64, 96, False, False)
    fused = trainer.FusedPFrameStep(stem, opt, aux)
    with torch.no_grad():
        ys = [imodel.getY(f) for f in frames]
    y_cond, losses, norms = ys[0][1], [], []
    for t in (1, 2):
        out, oc, aux_l, gn = fused.step(ys[t][0], y_cond, 2 * 256 * 256)
        losses.append(float(oc["loss"]))
        norms.append(float(gn))
        y_cond = out["y_hat"]
    fused.finish()
    torch.cuda.synchronize()
    if switch is not None:
        monkeypatch.undo()
    return opt.flat.data.clone(), aux.flat.data.clone(), y_cond.clone(), losses, norms


def test_schedule_switches_change_no_bit(monkeypatch):
    """every scheduling switch that is still supported (weight gradients on the compute stream, no hyper branch, one packing
    launch per role, bias second stages per layer, hyper branch enqueued first) gives the default's bits"""
    ref = _two_steps(monkeypatch, None)
    for sw in _SCHEDULE_SWITCHES:
        got = _two_steps(monkeypatch, sw)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2]), sw
        assert ref[3] == got[3] and ref[4] == got[4], (sw, ref[3:], got[3:])


@pytest.mark.parametrize("switch", _ROUTE_SWITCHES, ids=[s[0] for s in _ROUTE_SWITCHES])
def test_route_switches_stay_within_the_gates(monkeypatch, switch):
    """each arithmetic route that is still selectable (fp32-MFMA weight gradients / strided faces / context convolution, maxima
    measured instead of recorded, all STEM layers on fp32 MFMA) against the default fp16 route: losses to 1e-5, gradient norms to
    1e-3 (single flipped leaky-ReLU decisions move individual gradients, tests/test_hip_f16x3.py; the norm absorbs them)"""
    ref = _two_steps(monkeypatch, None)
    got = _two_steps(monkeypatch, switch)
    for a, b in zip(ref[3], got[3]):
        assert abs(a - b) <= 1e-5 * abs(a), (switch, ref[3], got[3])
    for a, b in zip(ref[4], got[4]):
        assert abs(a - b) <= 1e-3 * abs(a), (switch, ref[4], got[4])
    assert float((ref[2] - got[2]).abs().max()) <= 1e-4 * float(ref[2].abs().max())


def test_septuplet_loop_with_temporal_subsampling_is_the_same_on_every_route():
    """trainer.SeptupletTrainer.train_septuplet -- the loop body of stem/trainSTEM.py:174-226 that bench.py times -- over five
    loader items whose subsampling draws (:175-182) pick frames 1,3,5,7 / 1,4,7 / 1,7 / all seven / all seven: through the launch
    tape (with the latent prefetch) and through the plain explicit schedule (latents first) the per-step losses, clipped norms and
    every parameter after the 19 optimisation steps are bit-identical; the frame selection is the reference's slices."""
    import random
    from spatiotemporalentropymodel_amd.trainer import SeptupletTrainer, subsample_septuplet
    seven = list(range(7))
    assert subsample_septuplet(seven, 0.25) == [0, 2, 4, 6] and subsample_septuplet(seven, 0.2500001) == [0, 3, 6]
    assert subsample_septuplet(seven, 0.50) == [0, 3, 6] and subsample_septuplet(seven, 0.75) == [0, 6] and subsample_septuplet(seven, 0.76) == seven
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    items = [[torch.rand(1, 3, 64, 64, device=dev, generator=g) for _ in range(7)] for _ in range(5)]
    draws = [0.1, 0.4, 0.6, 0.9, 1.0]
    results = []
    for route, prefetch in (("taped", True), ("fused", False)):
        im, stem, opt, aux = _pair(64, 96, 64, 96, False, False)
        for i, m in enumerate((im.gaussian_conditional, stem.entropy_bottleneck, stem.gaussian_conditional)):
            m.noise_seed = 1000 + i
        tr = SeptupletTrainer(im, stem, opt, aux, route=route, prefetch=prefetch, rng=random.Random(5))
        log = []
        # a replayed step returns the recorded step's static buffers (tape.TapedPFrameStep): values are read step by step, as a
        # training loop's logging does (stem/trainSTEM.py:220-224), not after the item
        tr.on_step = lambda t, out, oc, aux_l, gn: log.append((float(oc["loss"]), float(gn), float(aux_l)))
        for frames, r in zip(items, draws):
            assert len(tr.train_septuplet(frames, rand=r)) == len(subsample_septuplet(seven, r)) - 1
        tr.finish()
        torch.cuda.synchronize()
        results.append((log, opt.flat.data.clone(), aux.flat.data.clone(), tr))
    (la, pa, qa, ta), (lb, pb, qb, _) = results
    assert len(la) == len(lb) == 3 + 2 + 1 + 6 + 6
    assert ta.step_fn.taped and ta.step_fn.replays >= 10
    for x, y in zip(la, lb):
        assert x[1] == y[1] and abs(x[0] - y[0]) <= 1e-12 * abs(y[0]) and abs(x[2] - y[2]) <= 1e-12 * abs(y[2]), (x, y)
    assert torch.equal(pa, pb) and torch.equal(qa, qb)
    # a draw from the trainer's own generator (rand=None) is the seeded sequence
    r = random.Random(5)
    assert len(ta.train_septuplet(items[0])) == len(subsample_septuplet(seven, r.random())) - 1
