#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: frames/s on synthetic 7x256x256 septuplets, STEM training step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 16 synthetic septuplets per GPU (configs[1]):
7x I-frame analysis transform g_a (getY) + 6 P-frame optimisation steps of SpatioTemporalPriorModel_Res(256,192)
(forward, EMLoss, backward, [RCCL all-reduce], fused clip+Adam, aux loss + aux Adam) = the loop body of
stem/trainSTEM.py:174-226 with all 7 frames used.  value = 7*16*N*K / t  frames/s (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BATCH, SIZE, FRAMES = 16, 256, 7
# the 192-ch 5x5 stride-2 analysis conv g_a.2 (7.550 GF/frame) + the GDN contraction fused
# into its epilogue (g_a.3, 0.302 GF/frame): one kernel launch per frame batch
GA2_FLOP_PER_FRAME = 2 * 192 * 192 * 25 * 64 * 64 + 2 * 192 * 192 * 64 * 64


def synthetic_septuplet(batch, size, seed, device):
    """7 x [B,3,size,size] in [0,1]: low-frequency sinusoid images translated by (t, 2t) px + N(0, 0.01^2)
    (SURVEY.md §8(d)); generated on the device, shape contract of stem/dataset_vidseq.py:57-88."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    fy = torch.rand(batch, 3, 8, 1, 1, device=device, generator=g) * 6 + 1
    fx = torch.rand(batch, 3, 8, 1, 1, device=device, generator=g) * 6 + 1
    ph = torch.rand(batch, 3, 8, 1, 1, device=device, generator=g) * 2 * math.pi
    amp = torch.rand(batch, 3, 8, 1, 1, device=device, generator=g) * 0.08 + 0.04
    yy = torch.arange(size, device=device, dtype=torch.float32).view(1, 1, 1, size, 1)
    xx = torch.arange(size, device=device, dtype=torch.float32).view(1, 1, 1, 1, size)
    frames = []
    for t in range(FRAMES):
        img = 0.5 + (amp * torch.sin(2 * math.pi * (fy * (yy + t) + fx * (xx + 2 * t)) / size + ph)).sum(2)
        img = img + 0.01 * torch.randn(img.shape, device=device, generator=g)
        frames.append(img.clamp_(0, 1).contiguous())
    return frames


def cpu_baseline():
    """The CPU oracle (port of the reference's arithmetic, oracle/stem_oracle.c) timed on this box's host
    cores on a bounded sample: ONE frame of g_a and ONE P-frame STEM forward+backward at B=1, big config,
    256x256.  A septuplet costs 7 g_a + 6 P-steps for 7 frames."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc
    from spatiotemporalentropymodel_amd.weights import closed_form_tensor
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)

    def W(*s):
        return (rng.standard_normal(s) * math.sqrt(2.0 / np.prod(s[1:]))).astype(np.float32)

    isd = {}
    ch = [3, 192, 192, 192, 192]
    for i in range(4):
        isd[f"g_a.{2 * i}.weight"], isd[f"g_a.{2 * i}.bias"] = W(ch[i + 1], ch[i], 5, 5), np.zeros(ch[i + 1], np.float32)
        if i < 3:
            isd[f"g_a.{2 * i + 1}.beta"] = np.ones(192, np.float32)
            isd[f"g_a.{2 * i + 1}.gamma"] = np.sqrt(0.1 * np.eye(192) + 2.0 ** -36).astype(np.float32)
    ssd = {}
    conv = {"TPM.0": (256, 192, 5), "TPM.2": (320, 256, 5), "TPM.4": (384, 320, 5), "HE.0": (256, 384, 3), "HE.2": (256, 256, 5),
            "HE.4": (256, 256, 5), "HD.0": (256, 256, 5), "HD.2": (256, 256, 5), "HD.4": (384, 256, 3),
            "context_prediction": (384, 192, 5), "EPM.0": (768, 1152, 1), "EPM.2": (576, 768, 1), "EPM.4": (384, 576, 1)}
    for n, (o, i, k) in conv.items():
        ssd[n + ".weight"], ssd[n + ".bias"] = W(o, i, k, k), np.zeros(256 if n in ("HD.0", "HD.2") else o, np.float32)
    for i, (fo, fi) in enumerate([(3, 1), (3, 3), (3, 3), (3, 3), (1, 3)]):
        ssd[f"entropy_bottleneck._matrix{i}"] = closed_form_tensor(f"entropy_bottleneck._matrix{i}", (256, fo, fi)).numpy()
        ssd[f"entropy_bottleneck._bias{i}"] = closed_form_tensor(f"entropy_bottleneck._bias{i}", (256, fo, 1)).numpy()
        if i < 4:
            ssd[f"entropy_bottleneck._factor{i}"] = closed_form_tensor(f"entropy_bottleneck._factor{i}", (256, fo, 1)).numpy()
    ssd["entropy_bottleneck.quantiles"] = closed_form_tensor("entropy_bottleneck.quantiles", (256, 1, 3)).numpy()
    x = rng.random((1, 3, SIZE, SIZE)).astype(np.float32)
    t0 = time.perf_counter()
    y = orc.g_a(isd, x)
    t_ga = time.perf_counter() - t0
    y_cond = y + rng.uniform(-0.5, 0.5, y.shape).astype(np.float32)
    noise = {"z": rng.uniform(-0.5, 0.5, (1, 256, 4, 4)).astype(np.float32),
             "q": rng.uniform(-0.5, 0.5, y.shape).astype(np.float32), "lik": rng.uniform(-0.5, 0.5, y.shape).astype(np.float32)}
    t0 = time.perf_counter()
    keep = {}
    out = orc.stem_forward(ssd, y, y_cond, residual=True, training=True, noise=noise, keep=keep)
    orc.stem_backward(ssd, keep, out["lik_y"], out["lik_z"], SIZE * SIZE)
    t_p = time.perf_counter() - t0
    t_sept = FRAMES * t_ga + (FRAMES - 1) * (t_ga * 0 + t_p)
    return {"value": FRAMES / t_sept, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/stem_oracle.c (OpenMP, {cores} threads): 1 frame g_a ({t_ga:.2f} s) + 1 P-frame STEM fwd+bwd "
                      f"({t_p:.2f} s) at B=1, 256x256, N=M=192; septuplet = 7 g_a + 6 P-steps, extrapolated"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from spatiotemporalentropymodel_amd import _lib
    from spatiotemporalentropymodel_amd import distributed as D
    _lib.hip()                                    # no HIP library -> fail loudly, nothing to measure
    rank, world, local = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from spatiotemporalentropymodel_amd.losses import EMLoss
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.selfcheck import p_frame_step
    from spatiotemporalentropymodel_amd.zoo import models

    torch.manual_seed(1234)                       # identical random-init weights on every rank
    imodel = models["mbt2018"](quality=4).to(dev).eval()          # stem/trainSTEM.py:113,128
    stem = SpatioTemporalPriorModel_Res().to(dev).train()         # stem/trainSTEM.py:115,130
    D.broadcast_parameters(stem)
    D.broadcast_parameters(imodel)
    seed = D.shard_seed(1234, rank)
    for m in (imodel.gaussian_conditional, stem.entropy_bottleneck, stem.gaussian_conditional):
        m.noise_seed = seed * 7919 + id(m) % 1000
    opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    # gradient slices are all-reduced (RCCL, side stream) as backward finishes each module group
    reducer = D.OverlappedGradReducer(opt.flat).attach(stem.engine()) if world > 1 else None
    crit = EMLoss()
    frames = synthetic_septuplet(BATCH, SIZE, seed, dev)

    # HIP-event probe around the dominant kernel (g_a.2 + fused GDN, the 192-ch 5x5 stride-2 analysis conv) on the
    # stream it is launched on (= torch's current stream, which is what the C ABI receives)
    probe = []
    imodel.g_a.probe = (2, probe)

    def one_step():
        with torch.no_grad():
            _, y_cond = imodel.getY(frames[0])
        last = None
        for t in range(1, FRAMES):
            out, oc, aux, gn = p_frame_step(imodel, stem, crit, opt, aux_opt, frames[t], y_cond,
                                            grad_scale=1.0 / world, reducer=reducer)
            y_cond = out["y_hat"]
            last = oc
        return last

    for _ in range(args.warmup):
        one_step()
    probe.clear()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in probe])) if probe else float("nan")
    loss = float(last["loss"].detach())
    if rank != 0:
        return
    flop = GA2_FLOP_PER_FRAME * BATCH
    achieved = flop / (kern_ms * 1e-3) / 1e12
    traffic = None
    tf = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        traffic = json.load(open(tf)).get("g_a2_bytes_per_launch")
    res = {
        "metric": "frames/s", "value": FRAMES * BATCH * world * args.steps / dt, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: SpatioTemporalPriorModel_Res(256,192) training step on mbt2018(192,192) latents, "
                               "16 septuplets x 7 frames x 256x256 per GPU, EMLoss, clip 1.0 + Adam 1e-4 / aux Adam 1e-3",
                   "per_gpu_batch": BATCH, "global_batch": BATCH * world, "frames_per_step": FRAMES * BATCH * world,
                   "p_frame_steps_per_step": FRAMES - 1, "parallelism": f"dp{world}", "final_loss_bpp": loss},
        "roofline": {"bound": "mfma", "kernel": "igemm_kernel<128,192,32,96,FUSE> = g_a.2 conv (192->192, 5x5 s2, 128^2->64^2, B=16) + fused GDN g_a.3",
                     "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "flop_per_launch": flop, "avg_launch_ms": kern_ms, "launches_timed": len(probe), "traffic": traffic},
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
