"""Shared harness pieces for smoke(), the GPU tests and bench.py: noise injection, the restated
per-P-frame training step of stem/trainSTEM.py:194-218, and smoke_check()."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from .weights import closed_form_fill_, closed_form_input, smooth_frames

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NoiseFeed:
    """noise_source hook: the k-th draw of role R is closed_form_input("noise:R:k") -- the same numbers
    tests/golden/make_golden.py fed to the reference through EntropyModel._get_noise_cached."""

    def __init__(self, role):
        self.role, self.k = role, 0

    def __call__(self, shape, device):
        name = f"noise:{self.role}:{self.k}"
        self.k += 1
        return closed_form_input(name, tuple(shape), -0.5, 0.5).to(device)


class SlicedNoise:
    """noise_source for rank `rank` of `world`: the rows this rank's `per_rank` samples own in the noise tensor ONE process would
    draw for the concatenated batch (same closed-form stream as NoiseFeed) -- what makes a data-parallel step comparable, number
    for number, with the single-process step over the global batch (tests/dp_worker.py, dp_step_vs_full_batch)."""

    def __init__(self, role, rank, world, per_rank, batch_last=False):
        self.role, self.rank, self.world, self.per, self.k, self.batch_last = role, rank, world, per_rank, 0, batch_last

    def __call__(self, shape, device):
        name = f"noise:{self.role}:{self.k}"
        self.k += 1
        lo, hi = self.rank * self.per, (self.rank + 1) * self.per
        if self.batch_last:                       # EntropyBottleneck asks for [C, 1, H*W*B] with B innermost
            Cc, one, n = shape
            hw = n // self.per
            full = closed_form_input(name, (Cc, 1, hw * self.per * self.world), -0.5, 0.5)
            return full.reshape(Cc, hw, self.per * self.world)[:, :, lo:hi].reshape(Cc, 1, n).contiguous().to(device)
        full = closed_form_input(name, (shape[0] * self.world,) + tuple(shape[1:]), -0.5, 0.5)
        return full[lo:hi].contiguous().to(device)


def dp_step_vs_full_batch(make_models, frames_of_rank, rank, world, device, size):
    """The self-check of a data-parallel run (bench.py STEM_BENCH_VERIFY=1, tests): ONE P-frame optimisation step of the explicit
    schedule (trainer.FusedPFrameStep) through the overlapped reducer on every rank's shard, against the same step over the
    GLOBAL batch computed by rank 0 alone from the same initial weights -- same frames, the same noise numbers (SlicedNoise).
    The loop body is stem/trainSTEM.py:194-218; the reference has no parallel form to compare with.

    make_models() -> (imodel, stem) with identical weights on every call and rank; frames_of_rank(r) -> [frame0, frame1] of
    rank r's shard ([B,3,size,size] on `device`).  Collective (every rank calls it).  Returns on every rank
    {"loss_dp", "loss_full", "loss_rel", "grad_rel", "rccl_nranks", "route"}; loss_full / *_rel are None off rank 0."""
    import torch.distributed as dist
    from . import distributed as D
    from .optim import configure_optimizers
    from .trainer import FusedPFrameStep

    def one(rank_, world_, frames, reducer_on):
        imodel, stem = make_models()
        stem.train()
        B = frames[0].shape[0]
        imodel.gaussian_conditional.noise_source = SlicedNoise("iframe_gc", rank_, world_, B)
        stem.entropy_bottleneck.noise_source = SlicedNoise("stem_eb", rank_, world_, B, batch_last=True)
        stem.gaussian_conditional.noise_source = SlicedNoise("stem_gc", rank_, world_, B)
        opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
        red = D.OverlappedGradReducer(opt.flat).attach(stem.engine()) if reducer_on else None
        fused = FusedPFrameStep(stem, opt, aux_opt)
        fused.clear_grad_in_adam = False
        with torch.no_grad():
            _, y_cond = imodel.getY(frames[0])
            y_cur, _ = imodel.getY(frames[1])
        out, oc, aux, gn = fused.step(y_cur, y_cond, B * size * size, grad_scale=1.0 / world_ if reducer_on else 1.0, reducer=red)
        fused.finish()
        torch.cuda.synchronize(device)
        if red is not None:
            red.check()
        g = opt.flat.grad.detach().clone() * (1.0 / world_ if reducer_on else 1.0)
        info = (red.rccl_nranks, red.route) if red is not None else (None, "none")
        stem.engine().grad_ready_hook = None
        return float(oc["loss"]), g, info

    loss_r, g_dp, (nranks, route) = one(rank, world, frames_of_rank(rank), dist.is_initialized())
    t = torch.tensor([loss_r], dtype=torch.float64, device="cpu" if (dist.is_initialized() and dist.get_backend() == "gloo") else device)
    if dist.is_initialized():
        dist.all_reduce(t)
    res = {"loss_dp": float(t.item()) / world, "loss_full": None, "loss_rel": None, "grad_rel": None, "rccl_nranks": nranks, "route": route}
    if rank == 0:
        full = [torch.cat([frames_of_rank(r)[i] for r in range(world)]) for i in range(2)]
        loss_f, g_full, _ = one(0, 1, full, False)
        res["loss_full"] = loss_f
        res["loss_rel"] = abs(res["loss_dp"] - loss_f) / max(abs(loss_f), 1e-30)
        res["grad_rel"] = float((g_dp - g_full).abs().max() / g_full.abs().max().clamp_min(1e-30))
    if dist.is_initialized():
        dist.barrier()
    return res


def build_models(ebc, cin, N, M, device, cls=None, closed_form=True, inject_noise=True):
    from .models import JointAutoregressiveHierarchicalPriors, SpatioTemporalPriorModel_Res
    cls = cls or SpatioTemporalPriorModel_Res
    imodel = JointAutoregressiveHierarchicalPriors(N, M)
    stem = cls(ebc, cin)
    if closed_form:
        closed_form_fill_(imodel)
        closed_form_fill_(stem)
    imodel, stem = imodel.to(device).eval(), stem.to(device)
    if inject_noise:
        imodel.gaussian_conditional.noise_source = NoiseFeed("iframe_gc")
        stem.entropy_bottleneck.noise_source = NoiseFeed("stem_eb")
        stem.gaussian_conditional.noise_source = NoiseFeed("stem_gc")
    return imodel, stem


def p_frame_step(imodel, stem, criterion, optimizer, aux_optimizer, x, y_cond, grad_scale=1.0, reducer=None, y_cur=None):
    """One P-frame optimisation step, the body of stem/trainSTEM.py:203-218 with the fused optimiser:
    zero_grad -> getY -> stem forward -> EMLoss -> backward -> [all-reduce] -> clip+Adam -> aux loss/step.
    `y_cur`: the frame's latents when the caller has already run the (frozen, no-grad) analysis transform -- e.g. for
    all frames of the septuplet up front, which lets the host run ahead of the GPU (bench.py)."""
    optimizer.zero_grad()
    aux_optimizer.zero_grad()
    if y_cur is None:
        with torch.no_grad():
            y_cur, _ = imodel.getY(x)
    out = stem(y_cur, y_cond)
    oc = criterion(out, x)
    oc["loss"].backward()                 # an attached OverlappedGradReducer exchanges slices during this call
    if reducer is not None:
        reducer.finish() if hasattr(reducer, "finish") else reducer.all_reduce()
    gn = optimizer.grad_norm() * grad_scale if hasattr(optimizer, "grad_norm") else None
    optimizer.step(grad_scale=grad_scale, norm_is_current=True) if hasattr(optimizer, "grad_norm") else optimizer.step()
    aux = stem.aux_loss()
    aux.backward()
    aux_optimizer.step()
    return out, oc, aux, gn


def roi_gop_step(imodel, pmodel, criterion, optimizers, frames, qmap, clip_max_norm=1.0, max_loss=None, accumulator=None):
    """One GOP iteration of the variable-rate training loop, stem_roi/train_stem_roi.py:509-631:

        zero all four gradients; lmbdamap = quality2lambda(Qmap)
        frame 0:  I(x0, Q)          -> loss.backward(retain_graph) -> clip(I grads)  -> I.aux_loss().backward()
        frame t:  P(xt, x_hat, Q)   -> loss.backward(retain_graph) -> clip(P grads)  -> P.aux_loss().backward()
        step optimizer_i, aux_optimizer_i, optimizer_p, aux_optimizer_p            (once, no further clipping)

    x_hat is NOT detached between frames, so a P frame's loss back-propagates through every earlier frame into both
    models; each model's running gradient (its `.quantiles` included, as clip_grad_norm_(model.parameters()) does) is
    clipped right after its own frame's backward.  `optimizers` = (opt_i, aux_i, opt_p, aux_p) from configure_optimizers
    (max_norm=None).  `max_loss` reproduces the script's "skip invalid loss" break (NaN/Inf/loss > max_loss; costs a host
    sync per frame; upstream compares the I frame against the previous GOP's P loss, here each frame checks its own;
    with an accumulator the verdict is OR-ed over the ranks so that all of them leave the GOP at the same frame).
    `accumulator` (distributed.GopGradAccumulator over the four flat buffers) makes the loop data parallel: the frame
    gradient is all-reduced before it joins the running sum that gets clipped.
    Returns the per-frame criterion dictionaries, clip norms and aux losses."""
    from .losses import quality2lambda
    from .optim import clip_grad_norm_
    opt_i, aux_i, opt_p, aux_p = optimizers
    if accumulator is not None:
        accumulator.begin()
    else:
        for o in optimizers:
            o.zero_grad()
    lmbdamap = quality2lambda(qmap)
    log, x_cond = [], None
    for idx, x in enumerate(frames):
        model, opts = (imodel, (opt_i, aux_i)) if idx == 0 else (pmodel, (opt_p, aux_p))
        out = model(x, qmap) if idx == 0 else model(x, x_cond, qmap)
        x_cond = out["x_hat"]
        oc = criterion(out, x, lmbdamap)
        if max_loss is not None:
            lv = float(oc["loss"].detach())
            bad = (not np.isfinite(lv)) or lv > max_loss
            if accumulator is not None:
                # data parallel: the decision must be the same on every rank, or the ranks that continue would wait
                # forever in end_frame()'s all-reduce for the one that left (and the replicas would step differently)
                bad = accumulator.any_rank(bad)
            if bad:
                break
        oc["loss"].backward(retain_graph=True)
        gn = None
        if accumulator is not None:
            accumulator.end_frame()
        if clip_max_norm and clip_max_norm > 0:
            bufs = [accumulator.running(o.flat) for o in opts] if accumulator is not None else None
            gn = clip_grad_norm_(opts, clip_max_norm, tensors=bufs)
        aux = model.aux_loss()
        aux.backward()
        if accumulator is not None:
            accumulator.end_aux()
        log.append((oc, gn, aux))
    if accumulator is not None:
        accumulator.finish()
    for o in optimizers:
        o.step()
    return log


def smoke_check(verbose=False):
    """One small training step on cuda:0 checked against (a) the golden vectors of the reference and
    (b) the CPU oracle evaluated on the very same inputs."""
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    from . import _lib
    _lib.hip()                                    # fail loudly if the HIP extension is missing
    from .losses import EMLoss
    from .optim import configure_optimizers
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc

    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(REPO, "tests", "golden", "stem_train_small.npz")))
    ebc, cin, N, M, batch, size, steps = (int(v) for v in g["cfg"])
    imodel, stem = build_models(ebc, cin, N, M, dev)
    stem.train()
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opt, aux_opt = configure_optimizers(stem, args)
    frames = [f.to(dev) for f in smooth_frames("train:small", batch, steps + 1, size)]
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    out, oc, aux, gn = p_frame_step(imodel, stem, EMLoss(), opt, aux_opt, frames[1], y_cond)
    torch.cuda.synchronize()
    # Gate values: the float64 run of the reference on the same inputs (tests/golden/stem_f64.npz, written by
    # make_golden.py:gen_f64) -- the reference's own fp32 numbers are up to 1e-4 away from it (its fp32 clip_grad_norm_:
    # 9.9e-5), so they are only the secondary check.
    x = dict(np.load(os.path.join(REPO, "tests", "golden", "stem_f64.npz")))
    loss, ybpp, zbpp, aux_ref, gn_ref = x["small:s1:scalars"]

    def rel(a, b):
        return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)

    def ratio(a, b):      # tests/conftest.py:close_ratio with floor 0.1, atol 1e-9 (the likelihood bound)
        return float(np.max(np.maximum(np.abs(a - b) - 1e-9, 0.0) / np.maximum(np.abs(b), 0.1 * np.abs(b).max())))

    checks = {"loss": rel(oc["loss"], loss), "y_bpp": rel(oc["y_bpp_loss"], ybpp), "z_bpp": rel(oc["z_bpp_loss"], zbpp),
              "grad_norm": rel(gn, gn_ref), "aux_loss": rel(aux, aux_ref)}
    lik_y = out["likelihoods"]["y"].detach().cpu().contiguous().numpy().astype(np.float64)
    checks["lik_y_vs_f64"] = ratio(lik_y, x["small:s1:lik_y"])
    checks["loss_vs_ref_fp32"] = rel(oc["loss"], g["s1:scalars"][0])
    # oracle on the same y_cur / y_cond (g_a through the oracle as well)
    isd = {k: v.detach().cpu().numpy() for k, v in imodel.state_dict().items() if v.dtype == torch.float32}
    y_ref = orc.g_a(isd, frames[1].cpu().numpy())
    with torch.no_grad():
        y_hip, _ = imodel.getY(frames[1])
    checks["g_a_vs_oracle"] = float(np.max(np.abs(y_hip.cpu().contiguous().numpy() - y_ref)) / np.abs(y_ref).max())
    # the yardstick beside each float64 gate: how far the REFERENCE'S OWN fp32 run is from the exact value in the same metric
    # (make_golden.py:gen_f64) -- lik_y at 7.6e-5 reads differently next to the reference's 5.0e-5 than next to nothing
    r32 = x["small:ref32:s1:scalars"]
    yard = {"loss": float(r32[0]), "y_bpp": float(r32[1]), "z_bpp": float(r32[2]), "aux_loss": float(r32[3]), "grad_norm": float(r32[4]),
            "lik_y_vs_f64": float(x["small:ref32:lik_y"][0])}
    if verbose:
        for k, v in checks.items():
            print(f"smoke: {k:18s} rel err {v:.3e}" + (f"   (reference-fp32 vs float64: {yard[k]:.3e})" if k in yard else ""))
    bad = {k: v for k, v in checks.items() if not v < 1e-4}
    if bad:
        raise AssertionError(f"smoke(): HIP path disagrees with the reference/oracle: {bad}")
    return checks
