#!/usr/bin/env python3
"""One data-parallel rank of the 2-rank GPU tests (tests/test_hip_dp2.py).  NOT a pytest module.

Two of these processes share cuda:0 and exchange gradients through torch.distributed's gloo backend (RCCL refuses
two ranks on one device), so that distributed.OverlappedGradReducer / GopGradAccumulator run against the REAL HIP
backward -- hooks fired from StemEngine.backward on the weight-gradient stream, slices of the flat gradient buffer
all-reduced while the rest of backward is still queued -- on the single-GPU test box.  The parent test compares what the
ranks dump with a single-process run over the concatenated batch.

    python tests/dp_worker.py --case train|train_fused|train_taped|train_untaped|gop|rccl1_train_fused|rccl1_gop|rccl1_train_taped|rccl2_verify|rccl2_train_taped --rank R --world W --port P --out DIR

The `rccl1_*` cases are ONE rank in a world-size-1 process group on the RCCL ("nccl") backend: the collectives are
identities, but every RCCL call of the reducers, their side-stream ordering and the device-tensor reductions of bench.py
(`max_over_ranks`, `any_rank`) execute for real -- as far as the RCCL path can go on a one-GPU box.
"""
import argparse
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


from spatiotemporalentropymodel_amd.selfcheck import SlicedNoise  # noqa: E402,F401  (the rows a rank owns in the global batch's noise)


DEVICE_INDEX = 0          # rccl2_* cases: the rank's own device


def flat_np(t):
    return t.detach().cpu().numpy().copy()


def case_train(rank, world, out_dir, steps=2):
    """`steps` P-frame optimisation steps of SpatioTemporalPriorModel_Res (small config) on ONE sample per rank, with the
    overlapped reducer attached to the engine: dumps losses, the exchanged (averaged) gradient of step 1, and the
    parameters after the last step."""
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.losses import EMLoss
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    dev = torch.device("cuda", DEVICE_INDEX)
    imodel, stem = S.build_models(64, 96, 64, 96, dev, inject_noise=False)
    stem.train()
    imodel.gaussian_conditional.noise_source = SlicedNoise("iframe_gc", rank, world, 1)
    stem.entropy_bottleneck.noise_source = SlicedNoise("stem_eb", rank, world, 1, batch_last=True)
    stem.gaussian_conditional.noise_source = SlicedNoise("stem_gc", rank, world, 1)
    D.broadcast_parameters(stem)
    opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    red = D.OverlappedGradReducer(opt.flat).attach(stem.engine())
    crit = EMLoss()
    frames = [f[rank:rank + 1].contiguous().to(dev) for f in smooth_frames("dp2:train", world, steps + 1, 64)]
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    dump = {}
    for t in range(1, steps + 1):
        if t == 1:                                  # by hand, to look at the exchanged gradient before Adam consumes it
            opt.zero_grad(), aux_opt.zero_grad()
            with torch.no_grad():
                y_cur, _ = imodel.getY(frames[t])
            out = stem(y_cur, y_cond)
            oc = crit(out, frames[t])
            oc["loss"].backward()
            red.finish()
            dump["grad_avg"] = flat_np(opt.flat.grad) * red.grad_scale
            gn = opt.grad_norm() * red.grad_scale
            opt.step(grad_scale=red.grad_scale, norm_is_current=True)
            aux = stem.aux_loss()
            aux.backward()
            aux_opt.step()
        else:
            out, oc, aux, gn = S.p_frame_step(imodel, stem, crit, opt, aux_opt, frames[t], y_cond, grad_scale=red.grad_scale,
                                              reducer=red)
        y_cond = out["y_hat"]
        dump[f"s{t}:loss"] = np.array([float(oc["loss"]), float(gn), float(aux)])
    torch.cuda.synchronize()
    dump["params"] = flat_np(opt.flat.data)
    dump["quantiles"] = flat_np(aux_opt.flat.data)
    dump["reducer_calls"] = np.array([red.calls])
    dump["issuer"] = np.array([type(getattr(red, "_issuer", None)).__name__])
    np.savez(os.path.join(out_dir, f"train_rank{rank}.npz"), **dump)


def case_train_fused(rank, world, out_dir, steps=2, tag="train_fused"):
    """The same two steps through the explicit schedule bench.py times (trainer.FusedPFrameStep) with the overlapped
    reducer attached -- the code path of `bench.py --gpus N`."""
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    dev = torch.device("cuda", DEVICE_INDEX)
    imodel, stem = S.build_models(64, 96, 64, 96, dev, inject_noise=False)
    stem.train()
    imodel.gaussian_conditional.noise_source = SlicedNoise("iframe_gc", rank, world, 1)
    stem.entropy_bottleneck.noise_source = SlicedNoise("stem_eb", rank, world, 1, batch_last=True)
    stem.gaussian_conditional.noise_source = SlicedNoise("stem_gc", rank, world, 1)
    D.broadcast_parameters(stem)
    opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    red = D.OverlappedGradReducer(opt.flat).attach(stem.engine())
    fused = FusedPFrameStep(stem, opt, aux_opt)
    fused.clear_grad_in_adam = False
    frames = [f[rank:rank + 1].contiguous().to(dev) for f in smooth_frames("dp2:train", world, steps + 1, 64)]
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    dump = {}
    for t in range(1, steps + 1):
        with torch.no_grad():
            y_cur, _ = imodel.getY(frames[t])
        out, oc, aux, gn = fused.step(y_cur, y_cond, 64 * 64, grad_scale=red.grad_scale, reducer=red)
        if t == 1:
            dump["grad_avg"] = flat_np(opt.flat.grad) * red.grad_scale
        y_cond = out["y_hat"]
        dump[f"s{t}:loss"] = np.array([float(oc["loss"]), float(gn), float(aux)])
    fused.finish()
    torch.cuda.synchronize()
    dump["params"] = flat_np(opt.flat.data)
    dump["quantiles"] = flat_np(aux_opt.flat.data)
    dump["reducer_calls"] = np.array([red.calls])
    dump["issuer"] = np.array([type(getattr(red, "_issuer", None)).__name__])
    import torch.distributed as dist
    dump["backend"] = np.array([dist.get_backend() if dist.is_initialized() else "none"])
    dump["max_over_ranks"] = np.array([D.max_over_ranks(3.25, dev)])
    np.savez(os.path.join(out_dir, f"{tag}_rank{rank}.npz"), **dump)


def case_train_taped(rank, world, out_dir, steps=8, tag="train_taped", taped=True):
    """Eight P-frame steps with the overlapped reducer attached, through the native executor (tape.TapedPFrameStep: two ordinary
    steps, two recorded, four replayed -- the reducer's torch.distributed calls are re-run from the tape's Python entries) or,
    taped=False, through the plain explicit schedule.  Philox noise (a tape replays the counters, not injected tensors)."""
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.tape import TapedPFrameStep
    from spatiotemporalentropymodel_amd.trainer import FusedPFrameStep
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    dev = torch.device("cuda", DEVICE_INDEX)
    torch.manual_seed(5)
    imodel, stem = S.build_models(64, 96, 64, 96, dev, inject_noise=False)
    stem.train()
    for i, m in enumerate((imodel, stem)):
        m.entropy_bottleneck.noise_seed = m.gaussian_conditional.noise_seed = D.shard_seed(77 + 100 * i, rank)
    D.broadcast_parameters(stem)
    opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    red = D.OverlappedGradReducer(opt.flat).attach(stem.engine())
    step = FusedPFrameStep(stem, opt, aux_opt)
    if taped:
        step = TapedPFrameStep(step)
    frames = [f[rank:rank + 1].contiguous().to(dev) for f in smooth_frames("dp2:taped", world, steps + 1, 64)]
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    dump = {}
    for t in range(1, steps + 1):
        with torch.no_grad():
            y_cur, _ = imodel.getY(frames[t])
        out, oc, aux, gn = step.step(y_cur, y_cond, 64 * 64, grad_scale=red.grad_scale, reducer=red)
        y_cond = out["y_hat"].clone()
        dump[f"s{t}:loss"] = np.array([float(oc["loss"]), float(gn), float(aux)])
        if t == 6:                                  # a scheduler's edit between two replayed steps (stem/trainSTEM.py:123,290)
            opt.param_groups[0]["lr"] *= 0.5
    step.finish()
    torch.cuda.synchronize()
    dump["params"] = flat_np(opt.flat.data)
    dump["quantiles"] = flat_np(aux_opt.flat.data)
    dump["collectives"] = np.array([red.collectives])
    dump["replays"] = np.array([step.replays if taped else 0])
    dump["taped"] = np.array([bool(taped and step.taped)])
    dump["issuer"] = np.array([type(getattr(red, "_issuer", None)).__name__])
    dump["nranks"] = np.array([red.rccl_nranks or 0])
    dump["replicas_identical"] = np.array([D.replicas_identical(opt.flat.data)[0]])
    import torch.distributed as dist
    dump["backend"] = np.array([dist.get_backend() if dist.is_initialized() else "none"])
    np.savez(os.path.join(out_dir, f"{tag}_rank{rank}.npz"), **dump)


def case_gop(rank, world, out_dir, frames_n=3, tag="gop"):
    """One GOP iteration of the variable-rate loop (selfcheck.roi_gop_step) with GopGradAccumulator, one sample per rank."""
    out_tag = tag
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.losses import PixelwiseRateDistortionLoss
    from spatiotemporalentropymodel_amd.models import stem_roi, stem_roi_i
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_scaled_, closed_form_input, smooth_frames
    dev = torch.device("cuda", DEVICE_INDEX)
    imodel, pmodel = stem_roi_i(), stem_roi()
    closed_form_fill_scaled_(imodel, "stem_roi_i", 0.7)
    closed_form_fill_scaled_(pmodel, "stem_roi", 0.7)
    imodel, pmodel = imodel.to(dev).train(), pmodel.to(dev).train()
    for m, tag in ((imodel, "i"), (pmodel, "p")):
        m.entropy_bottleneck.noise_source = SlicedNoise(f"roi_{tag}_eb", rank, world, 1, batch_last=True)
        m.gaussian_conditional.noise_source = SlicedNoise(f"roi_{tag}_gc", rank, world, 1)
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opt_i, aux_i = configure_optimizers(imodel, args, max_norm=None)
    opt_p, aux_p = configure_optimizers(pmodel, args, max_norm=None)
    acc = D.GopGradAccumulator([opt_i.flat, opt_p.flat], [aux_i.flat, aux_p.flat])
    frames = [f[rank:rank + 1].contiguous().to(dev) for f in smooth_frames("dp2:gop", world, frames_n, 64)]
    qmap = closed_form_input("dp2:qmap", (world, 1, 64, 64), 0.0, 1.0)[rank:rank + 1].contiguous().to(dev)
    log = S.roi_gop_step(imodel, pmodel, PixelwiseRateDistortionLoss(), (opt_i, aux_i, opt_p, aux_p), frames, qmap,
                         clip_max_norm=float(os.environ.get("DP2_CLIP", "1.0")), accumulator=acc)
    torch.cuda.synchronize()
    dump = {"losses": np.array([[float(oc["loss"]), float(gn) if gn is not None else 0.0, float(aux)] for oc, gn, aux in log]),
            "params_i": flat_np(opt_i.flat.data), "params_p": flat_np(opt_p.flat.data),
            "grad_i": flat_np(opt_i.flat.grad), "grad_p": flat_np(opt_p.flat.grad),
            "any_rank": np.array([acc.any_rank(False), acc.any_rank(True)]), "active": np.array([acc.active])}
    np.savez(os.path.join(out_dir, f"{out_tag}_rank{rank}.npz"), **dump)


def case_rccl2_verify(rank, world, out_dir, tag="rccl2_verify"):
    """TWO real RCCL ranks, one device each (needs >= 2 visible GPUs): (a) one P-frame step through the overlapped reducer against
    the same step over the global batch on rank 0 alone (selfcheck.dp_step_vs_full_batch -- bench.py's STEM_BENCH_VERIFY check);
    (b) replicas_identical on parameters that went through it."""
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd import selfcheck as S
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    dev = torch.device("cuda", DEVICE_INDEX)
    torch.cuda.set_device(dev)
    allf = smooth_frames("dp2:verify", world, 2, 64)
    v = S.dp_step_vs_full_batch(lambda: S.build_models(64, 96, 64, 96, dev, inject_noise=False),
                                lambda r: [f[r:r + 1].contiguous().to(dev) for f in allf], rank, world, dev, 64)
    same, csum = D.replicas_identical(torch.full((1000,), 1.5, device=dev))
    diff, _ = D.replicas_identical(torch.full((1000,), 1.5 + rank, device=dev))
    np.savez(os.path.join(out_dir, f"{tag}_rank{rank}.npz"), loss_dp=np.array([v["loss_dp"]]),
             loss_rel=np.array([v["loss_rel"] if v["loss_rel"] is not None else -1.0]),
             grad_rel=np.array([v["grad_rel"] if v["grad_rel"] is not None else -1.0]),
             nranks=np.array([v["rccl_nranks"] or 0]), route=np.array([v["route"]]), same=np.array([same]), diff=np.array([diff]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rccl1 = a.case.startswith("rccl1_")
    rccl2 = a.case.startswith("rccl2_")
    global DEVICE_INDEX
    base = a.case
    if "@" in a.case:                               # rccl1_<case>@<n>: the reducer's issue mode (STEM_DP_THREADED=n)
        base, mode = a.case.split("@")
        os.environ["STEM_DP_THREADED"] = mode
    if rccl2:                                       # one device per rank, RCCL between them
        DEVICE_INDEX = a.rank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world),
                      LOCAL_RANK=str(DEVICE_INDEX), STEM_DIST_BACKEND="nccl" if (rccl1 or rccl2) else "gloo")
    from spatiotemporalentropymodel_amd import distributed as D
    D.init_from_env(single=rccl1)
    if rccl2:
        assert a.world == 2 and torch.distributed.get_backend() == "nccl" and torch.cuda.device_count() >= 2
        {"rccl2_verify": case_rccl2_verify, "rccl2_train_taped": case_train_taped}[base](a.rank, a.world, a.out, tag=a.case)
    elif rccl1:
        assert a.world == 1 and torch.distributed.get_backend() == "nccl"
        {"rccl1_train_fused": case_train_fused, "rccl1_gop": case_gop, "rccl1_train_taped": case_train_taped}[base](a.rank, a.world, a.out, tag=a.case)
    else:
        {"train": case_train, "train_fused": case_train_fused, "gop": case_gop, "train_taped": case_train_taped,
         "train_untaped": lambda r, w, o: case_train_taped(r, w, o, tag="train_untaped", taped=False)}[a.case](a.rank, a.world, a.out)
    D.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
