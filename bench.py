#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: frames/s on synthetic 7x256x256 septuplets, STEM training step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N rank processes (launch_ranks:
plain child processes with torch.distributed.run's environment contract, started before this process has made any GPU call)
and relays rank 0's JSON line; with more ranks than GPUs (the one-GPU test box) the ranks share devices and exchange through
gloo instead of RCCL (distributed.init_from_env).

One "step" = one pass of the hot path over one batch of 16 synthetic septuplets per GPU (configs[1]):
7x I-frame analysis transform g_a (getY) + 6 P-frame optimisation steps of SpatioTemporalPriorModel_Res(256,192)
(forward, EMLoss, backward, [RCCL all-reduce], fused clip+Adam, aux loss + aux Adam) = the loop body of
stem/trainSTEM.py:174-226 with all 7 frames used.  value = 7*16*N*K / t  frames/s (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# A rank of a data-parallel run has FIVE active streams (compute, weight gradients, hyper branch, latent prefetch + the process
# group's): with the runtime's default of four hardware queues per priority every high-priority stream gets a queue of its own, and
# the fifth active queue costs 2.2 ms per step on one MI355X (a stream that only waits for events and records one reproduces it:
# DESIGN.md 8).  With GPU_MAX_HW_QUEUES=2 the streams share queues and a ONE-rank RCCL run takes 12.84 instead of 13.85 ms (11.54
# without a group; profiles/r05_ab_hwq2.log) -- but a one-rank collective is a no-op, whereas a real one is a kernel that waits for
# its peers and would then sit in a queue it shares with one of the step's streams.  That cannot be measured on a one-GPU box, so
# multi-rank runs keep the runtime's default; STEM_BENCH_HW_QUEUES=<n> sets GPU_MAX_HW_QUEUES for an experiment (it has to be in the
# environment before the HIP runtime starts, i.e. before torch is imported).
if os.environ.get("STEM_BENCH_HW_QUEUES", "").strip():
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["STEM_BENCH_HW_QUEUES"].strip()

import numpy as np  # noqa: E402
import torch  # noqa: E402


def RUNTIME():
    """the package's configuration object (routes, schedule, data parallel; config.StemRuntimeConfig)"""
    from spatiotemporalentropymodel_amd import config
    return config.runtime()

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BATCH, SIZE, FRAMES = 16, 256, 7
# the 192-ch 5x5 stride-2 analysis conv g_a.2 (7.550 GF/frame) + the GDN contraction fused
# into its epilogue (g_a.3, 0.302 GF/frame): one kernel launch per frame batch
GA2_FLOP_PER_FRAME = 2 * 192 * 192 * 25 * 64 * 64 + 2 * 192 * 192 * 64 * 64
GA2_CONV_FLOP_PER_FRAME = 2 * 192 * 192 * 25 * 64 * 64           # the convolution alone (the part that runs as 3 fp16 MFMAs per product)
GA0_FLOP_PER_FRAME = 2 * 192 * 3 * 25 * 128 * 128 + 2 * 192 * 192 * 128 * 128      # g_a.0 (3 -> 192, 5x5 s2) + fused GDN g_a.1
# useful (algorithmic) flop of one bench step: 187.1 GF per septuplet (7 x g_a 11.966 GF + 6 P-frame steps of
# 17.228 GF = STEM forward 6.418 + weight gradients 6.418 + input gradients 4.392), 16 septuplets per GPU
USEFUL_FLOP_PER_STEP = 187.1e9 * BATCH
PEAK_F16_MFMA_TFLOPS = 2516.6        # same guide: v_mfma_f32_32x32x16_f16 / 16x16x32 (equal flop per cycle), 1024 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz (dense)
F16_PRODUCTS = 3                     # fp16 MFMA products per fp32 product in the split-operand kernels (csrc/stem_common.h)
PINNED_CPUS = None                   # host cores this rank pinned itself to (multi-GPU runs: distributed.pin_rank_to_gpu_cores)


def _apply_bench_tuning(_lib):
    """sweep aid: STEM_BENCH_TUNING="fx3_tile=64,fx3_depth=2" forces plan selectors of the library (stem_tuning_set: validated names
    and values, none of them changes a result); STEM_BENCH_FX3_TILE=64 is the older spelling of fx3_tile=64"""
    spec = os.environ.get("STEM_BENCH_TUNING", "")
    if os.environ.get("STEM_BENCH_FX3_TILE"):
        spec = f"fx3_tile={os.environ['STEM_BENCH_FX3_TILE']}," + spec
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        _lib.check(_lib.hip().stem_tuning_set(k.strip().encode(), int(v)))


def _masked_cus():
    """CUs the latent-prefetch stream may use (256 without a mask)"""
    from spatiotemporalentropymodel_amd import functional as F
    words = F._cu_mask("latents")
    return 256 if words is None else sum(bin(w).count("1") for w in words)


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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """distinct (socket, core) pairs of /proc/cpuinfo; os.cpu_count() (logical CPUs) when that cannot be read"""
    try:
        cores, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cores.add((phys, line.split(":", 1)[1].strip()))
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


def _median_time(fn, repeats=3, warmup=1, budget_s=None):
    """`warmup` untimed calls, then the median wall time of up to `repeats` calls (at least 3; stops early once `budget_s`
    seconds of timed calls have been spent)"""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if budget_s is not None and len(ts) >= 3 and sum(ts) + ts[-1] > budget_s:
            break
    return float(np.median(ts)), ts


def _baseline_weights(rng):
    """random-init weights of the config-2 / config-1 models in the reference's state-dict layout (numpy)"""
    from spatiotemporalentropymodel_amd.weights import closed_form_tensor

    def W(*s):
        return (rng.standard_normal(s) * math.sqrt(2.0 / np.prod(s[1:]))).astype(np.float32)

    def transforms(N, M):
        sd, ch = {}, [3, N, N, N, M]
        for i in range(4):
            sd[f"g_a.{2 * i}.weight"], sd[f"g_a.{2 * i}.bias"] = W(ch[i + 1], ch[i], 5, 5), np.zeros(ch[i + 1], np.float32)
            sd[f"g_s.{2 * i}.weight"], sd[f"g_s.{2 * i}.bias"] = W(ch[4 - i], ch[3 - i], 5, 5), np.zeros(ch[3 - i], np.float32)
            if i < 3:
                for t in ("g_a", "g_s"):
                    sd[f"{t}.{2 * i + 1}.beta"] = np.ones(N, np.float32)
                    sd[f"{t}.{2 * i + 1}.gamma"] = np.sqrt(0.1 * np.eye(N) + 2.0 ** -36).astype(np.float32)
        return sd

    def stem_weights(ebc, cin):
        sd = {}
        conv = {"TPM.0": (256, cin, 5), "TPM.2": (320, 256, 5), "TPM.4": (2 * cin, 320, 5), "HE.0": (256, 2 * cin, 3),
                "HE.2": (256, 256, 5), "HE.4": (ebc, 256, 5), "HD.0": (ebc, 256, 5), "HD.2": (256, 256, 5), "HD.4": (2 * cin, 256, 3),
                "context_prediction": (2 * cin, cin, 5), "EPM.0": (768, 6 * cin, 1), "EPM.2": (576, 768, 1), "EPM.4": (2 * cin, 576, 1)}
        for n, (o, i, k) in conv.items():
            sd[n + ".weight"], sd[n + ".bias"] = W(o, i, k, k), np.zeros(256 if n in ("HD.0", "HD.2") else o, np.float32)
        for i, (fo, fi) in enumerate([(3, 1), (3, 3), (3, 3), (3, 3), (1, 3)]):
            sd[f"entropy_bottleneck._matrix{i}"] = closed_form_tensor(f"entropy_bottleneck._matrix{i}", (ebc, fo, fi)).numpy()
            sd[f"entropy_bottleneck._bias{i}"] = closed_form_tensor(f"entropy_bottleneck._bias{i}", (ebc, fo, 1)).numpy()
            if i < 4:
                sd[f"entropy_bottleneck._factor{i}"] = closed_form_tensor(f"entropy_bottleneck._factor{i}", (ebc, fo, 1)).numpy()
        sd["entropy_bottleneck.quantiles"] = closed_form_tensor("entropy_bottleneck.quantiles", (ebc, 1, 3)).numpy()
        return sd

    return transforms, stem_weights


def _cpu_baseline_torch(budget_s):
    """(i) of cpu_baseline on torch CPU operators: what the reference itself executes on a CPU (oracle/stem_torch_cpu.py, pinned
    against the C oracle by tests/test_oracle_vs_golden.py::test_torch_cpu_leg_matches_oracle)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_torch_cpu as tc
    cores = _physical_cores()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        rng = np.random.default_rng(0)
        transforms, stem_weights = _baseline_weights(rng)
        tr = tc.PFrameTrainer(transforms(192, 192), stem_weights(256, 192))
        B = BATCH
        ga_times = []

        def p_step():
            x = torch.from_numpy(rng.random((B, 3, SIZE, SIZE), dtype=np.float32))
            lat = (B, 192, SIZE // 16, SIZE // 16)
            noise = {"z": torch.from_numpy(rng.uniform(-0.5, 0.5, (B, 256, SIZE // 64, SIZE // 64)).astype(np.float32)),
                     "q": torch.from_numpy(rng.uniform(-0.5, 0.5, lat).astype(np.float32)),
                     "lik": torch.from_numpy(rng.uniform(-0.5, 0.5, lat).astype(np.float32))}
            y_noise = torch.from_numpy(rng.uniform(-0.5, 0.5, lat).astype(np.float32))
            ga_times.append(tr.step(x, noise, y_noise)[1])

        t_step, step_times = _median_time(p_step, repeats=10, warmup=2, budget_s=budget_s)
        t_ga = float(np.median(ga_times[2:]))
        threads = torch.get_num_threads()
    finally:
        torch.set_num_threads(old_threads)
    t_sept = FRAMES * t_ga + (FRAMES - 1) * (t_step - t_ga)
    return {"frames_per_s": FRAMES * B / t_sept, "threads": int(threads), "p_step_s": t_step, "g_a_s": t_ga, "batch": B,
            "warmup_runs": 2, "timed_runs": len(step_times), "p_step_runs_s": [round(t, 4) for t in step_times]}


def _cpu_baseline_port(budget_s):
    """the same step on the numpy im2col + SGEMM port of the path (oracle/stem_port_blas.py on the host BLAS, elementwise
    entropy-model pieces from oracle/stem_oracle.c): the round-1..3 baseline, kept beside the torch-CPU leg; plus config 1
    (forward of one 7x256x256 septuplet through the small model: 7 g_a, 6 STEM forwards, 6 g_s)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc
    import stem_port_blas as port
    cores = _physical_cores()
    limiter = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limiter = threadpool_limits(limits=cores, user_api="blas")
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    transforms, stem_weights = _baseline_weights(rng)

    with port.installed():
        isd, ssd = transforms(192, 192), stem_weights(256, 192)
        names = sorted(k for k in ssd if not k.endswith("quantiles"))
        adam = {k: (np.zeros_like(ssd[k]), np.zeros_like(ssd[k])) for k in names}
        state = {"t": 0}

        def p_step(B):
            x = rng.random((B, 3, SIZE, SIZE), dtype=np.float32)
            t0 = time.perf_counter()
            y = orc.g_a(isd, x)
            t_ga = time.perf_counter() - t0
            y_cond = y + rng.uniform(-0.5, 0.5, y.shape).astype(np.float32)
            noise = {"z": rng.uniform(-0.5, 0.5, (B, 256, 4, 4)).astype(np.float32),
                     "q": rng.uniform(-0.5, 0.5, y.shape).astype(np.float32), "lik": rng.uniform(-0.5, 0.5, y.shape).astype(np.float32)}
            keep = {}
            out = orc.stem_forward(ssd, y, y_cond, residual=True, training=True, noise=noise, keep=keep)
            g = orc.stem_backward(ssd, keep, out["lik_y"], out["lik_z"], B * SIZE * SIZE)
            norm = math.sqrt(sum(float(np.vdot(g[k], g[k])) for k in names))              # clip_grad_norm_(1.0) + Adam
            clip = min(1.0, 1.0 / (norm + 1e-6))
            state["t"] += 1
            c1, c2 = 1 - 0.9 ** state["t"], 1 - 0.999 ** state["t"]
            for k in names:
                m, v = adam[k]
                gk = g[k].reshape(ssd[k].shape) * np.float32(clip)
                m *= np.float32(0.9); m += np.float32(0.1) * gk
                v *= np.float32(0.999); v += np.float32(0.001) * gk * gk
                ssd[k] -= np.float32(1e-4 / c1) * m / (np.sqrt(v / np.float32(c2)) + np.float32(1e-8))
            return t_ga

        B = BATCH
        ga_times = []
        t_step, step_times = _median_time(lambda: ga_times.append(p_step(B)), repeats=3, warmup=1, budget_s=budget_s)
        t_ga = float(np.median(ga_times[1:]))
        # config 1: small model, one septuplet forward (eval)
        isd1, ssd1 = transforms(64, 96), stem_weights(64, 96)
        frames = [rng.random((1, 3, SIZE, SIZE), dtype=np.float32) for _ in range(FRAMES)]

        def septuplet_forward():
            y_cond = orc.g_a(isd1, frames[0])
            for t in range(1, FRAMES):
                y = orc.g_a(isd1, frames[t])
                out = orc.stem_forward(ssd1, y, y_cond, residual=False, training=False)
                orc.g_s(isd1, out["y_hat"])
                y_cond = out["y_hat"]

        t_sept1, _ = _median_time(septuplet_forward)
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
    t_sept = FRAMES * t_ga + (FRAMES - 1) * (t_step - t_ga)
    return {"frames_per_s": FRAMES * B / t_sept, "blas_threads": int(blas_threads), "p_step_s": t_step, "g_a_s": t_ga, "batch": B,
            "warmup_runs": 1, "timed_runs": len(step_times), "config1_septuplet_forward_s": t_sept1}


def cpu_baseline(budget_s=45.0):
    """The reference's CPU path on this box's host cores, config 2: one P-frame
    optimisation step at B = 16 (g_a of the frame, STEM forward, EMLoss, backward, global-norm clip + Adam, auxiliary loss +
    Adam on the quantiles), 2 warm-ups, then the median of up to 10 runs -- fewer (never under 3) when 10 would take more than
    `budget_s` seconds, because the default bench run has to finish within minutes; the record says how many.

    `value` = the torch-CPU leg (kind "torch-cpu": torch.nn.functional convolutions through MKL-DNN, autograd,
    clip_grad_norm_, torch.optim.Adam with torch.set_num_threads(physical cores) -- what spatiotemporalpriors.py:807-868 +
    stem/trainSTEM.py:203-218 execute on a CPU), frames/s over a septuplet schedule (7 g_a + 6 P-steps per 7 frames) like the GPU
    number.  `port` = the numpy im2col + SGEMM port of rounds 1-3 beside it (one warm-up, 3 runs)."""
    tl = _cpu_baseline_torch(budget_s)
    pl = _cpu_baseline_port(30.0)
    cores = _physical_cores()
    return {"value": tl["frames_per_s"], "unit": "frames/s", "cores": tl["threads"], "kind": "torch-cpu",
            "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(), "host_physical_cores": cores, "batch": tl["batch"],
            "warmup_runs": tl["warmup_runs"], "timed_runs": tl["timed_runs"], "p_step_s": tl["p_step_s"], "p_step_runs_s": tl["p_step_runs_s"],
            "g_a_s": tl["g_a_s"],
            "port": {"value": pl["frames_per_s"], "kind": "port", "cores": pl["blas_threads"], "p_step_s": pl["p_step_s"], "g_a_s": pl["g_a_s"],
                     "warmup_runs": pl["warmup_runs"], "timed_runs": pl["timed_runs"], "config1_septuplet_forward_s": pl["config1_septuplet_forward_s"]},
            "sample": f"oracle/stem_torch_cpu.py (torch {torch.__version__} CPU operators, {tl['threads']} threads) on {_cpu_model()}: config-2 P-frame "
                      f"step at B={tl['batch']} (g_a {tl['g_a_s']:.2f} s + STEM fwd / bwd / clip / Adam / aux {tl['p_step_s'] - tl['g_a_s']:.2f} s; "
                      f"2 warm-ups, median of {tl['timed_runs']} runs), septuplet = 7 g_a + 6 P-steps -> {tl['frames_per_s']:.2f} frames/s; "
                      f"beside it the numpy im2col + SGEMM port ({pl['blas_threads']} BLAS threads): P-step {pl['p_step_s']:.2f} s -> "
                      f"{pl['frames_per_s']:.2f} frames/s; config-1 septuplet forward (small model, port) {pl['config1_septuplet_forward_s']:.2f} s"}


def bench_roi(args):
    """--config roi: BASELINE.json configs[4] -- one GOP training iteration of the variable-rate pair (stem_roi_i, stem_roi;
    stem_roi/train_stem_roi.py:509-631 = selfcheck.roi_gop_step) per step on B=16 GOPs of 7 frames 256x256 per GPU with the
    four lambda points in one batch (uniform quality maps 0.30 / 0.45 / 0.55 / 0.70, four samples each,
    stem_roi/eval_stem_roi.py:368-376).  roofline: the forward launch of the full-resolution 3x3 convolution 192 -> 160 of
    the quality-feature net (qmap_feature_ga1.2: the largest single layer of the model, 2*192*160*9 flop per pixel)."""
    from spatiotemporalentropymodel_amd import _lib
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd.losses import PixelwiseRateDistortionLoss
    from spatiotemporalentropymodel_amd.models import stem_roi, stem_roi_i
    from spatiotemporalentropymodel_amd.optim import configure_optimizers
    from spatiotemporalentropymodel_amd.selfcheck import roi_gop_step
    _lib.hip()
    _apply_bench_tuning(_lib)
    rank, world, local = D.init_from_env()
    assert world == args.gpus and torch.cuda.is_available()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B = args.roi_batch
    torch.manual_seed(1234)
    imodel, pmodel = stem_roi_i().to(dev).train(), stem_roi().to(dev).train()
    D.broadcast_parameters(imodel)
    D.broadcast_parameters(pmodel)
    for i, m in enumerate((imodel, pmodel)):
        m.entropy_bottleneck.noise_seed = m.gaussian_conditional.noise_seed = D.shard_seed(1234 + 100 * i, rank)
    a = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opts = configure_optimizers(imodel, a, max_norm=None) + configure_optimizers(pmodel, a, max_norm=None)
    acc = D.GopGradAccumulator([opts[0].flat, opts[2].flat], [opts[1].flat, opts[3].flat]) if D.dist.is_initialized() else None
    frames = synthetic_septuplet(B, SIZE, D.shard_seed(1234, rank), dev)
    levels = (0.30, 0.45, 0.55, 0.70)
    qmap = torch.cat([torch.full((B // 4, 1, SIZE, SIZE), q) for q in levels]).to(dev)
    crit = PixelwiseRateDistortionLoss()
    probe = []
    pmodel.qmap_feature_ga1.probe = (2, probe)
    for _ in range(args.warmup):
        roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0, accumulator=acc)
    probe.clear()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        log = roi_gop_step(imodel, pmodel, crit, opts, frames, qmap, 1.0, accumulator=acc)
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    dp_report = None
    if acc is not None:                              # the replicas' own evidence (see main(): config.data_parallel)
        torch.cuda.synchronize()
        same = [D.replicas_identical(o.flat.data) for o in opts]
        dp_report = {"rccl_nranks": D.dist.get_world_size(), "exchange_route": f"torch.distributed {D.dist.get_backend()} (GopGradAccumulator)",
                     "replicas_identical": all(ok for ok, _ in same), "replica_checksums_rank0": [f"{c & (2 ** 64 - 1):016x}" for _, c in same]}
    if rank != 0:
        return
    kern_ms = float(np.mean([x.elapsed_time(y) for x, y in probe]))
    flop = 2.0 * 192 * 160 * 9 * SIZE * SIZE * B
    f16_layers = RUNTIME().layers_f16x3
    # the probed layer runs on the 192-column split-operand kernel (three fp16 MFMAs per fp32 product: executed = 3 x algorithmic,
    # against the fp16 peak); with STEM_LAYERS_F16X3=0 on the fp32-MFMA kernel against its own peak
    work, peak = (F16_PRODUCTS * flop, PEAK_F16_MFMA_TFLOPS) if f16_layers else (flop, PEAK_FP32_MFMA_TFLOPS)
    achieved = work / (kern_ms * 1e-3) / 1e12
    _emit({
        "metric": "frames/s", "value": FRAMES * B * world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[4]: variable-rate stem_roi_i + stem_roi GOP training iteration (I + 6 P frames 256x256, BPTT across the "
                               "GOP, clip after every frame, one step of 4 Adam optimisers), 4 lambda points (quality 0.30/0.45/0.55/0.70) in one batch",
                   "per_gpu_batch": B, "global_batch": B * world, "frames_per_step": FRAMES * B * world, "parallelism": f"dp{world}",
                   "stride1_convolutions": "fp16 matrix cores, two fp16 planes per operand, 3 products per fp32 product (layers.Conv2dFunction)" if f16_layers else "fp32 MFMA",
                   "final_loss": float(log[-1][0]["loss"].detach()), **({"data_parallel": dp_report} if dp_report else {})},
        "roofline": {"bound": "mfma",
                     "kernel": ("conv_f16x3_kernel<128> with activation epilogue" if f16_layers else "igemm") +
                               " = conv3x3 192->160 at 256x256 (stem_roi.qmap_feature_ga1.2, forward), B=%d" % B,
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "flop_per_launch": work, "useful_flop_per_launch": flop, "useful_tflops": flop / (kern_ms * 1e-3) / 1e12,
                     "avg_launch_ms": kern_ms, "launches_timed": len(probe), "traffic": None}})


def _cpu_baseline_eval(stem, enc, y_cond, y_hat_gpu, budget_s=15.0):
    """cpu_baseline of --config eval: the raster-order decoding loop as the reference executes it on a CPU
    (oracle/stem_torch_cpu.decode_positions: torch CPU conv2d per position + the reference's own RansDecoder from oracle/_ref) on
    the FIRST positions of the very P frame the GPU just decoded (its strings, its tables, its hyper / temporal prior tensors),
    for `budget_s` seconds; the positions it decodes are compared with the GPU's."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc
    import stem_torch_cpu as tc
    dec = orc.reference_rans_decoder()
    if dec is None:
        return None
    cores = _physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        with torch.no_grad():
            dev = y_cond.device
            z_hat = stem.entropy_bottleneck.decompress(enc["strings"][1], enc["shape"]).to(dev).float()
            hp = stem.HD(z_hat).float().cpu().contiguous()
            tp = stem.TPM(y_cond).float().cpu().contiguous()
        gc = stem.gaussian_conditional
        tables = {"gc_cdf": gc._quantized_cdf.cpu().numpy(), "gc_cdf_length": gc._cdf_length.cpu().numpy(), "gc_offset": gc._offset.cpu().numpy(),
                  "gc_scale_table": gc.scale_table.cpu().numpy()}
        ssd = {k: v.detach().float().cpu().numpy() for k, v in stem.state_dict().items() if k.startswith(("context_prediction.", "EPM."))}
        H, W = hp.shape[-2:]
        yc = y_cond.float().cpu().contiguous()
        # Short runs pick the thread count (one position is a handful of tiny convolutions: on a 128-core host torch's full
        # thread pool costs more than it computes -- 30 ms per position with 128 threads against a few ms with 8) and size the
        # sample; the baseline is the FASTEST setting, all of them are recorded
        tried = {}
        for nt in sorted({1, 8, 32, cores}):
            if nt > cores:
                continue
            torch.set_num_threads(nt)
            tc.decode_positions(ssd, yc, hp, tp, enc["strings"][0][0], tables, orc.reference_rans_decoder(), max_positions=4)
            _, n0, t0 = tc.decode_positions(ssd, yc, hp, tp, enc["strings"][0][0], tables, orc.reference_rans_decoder(), max_positions=16)
            tried[nt] = t0 / max(n0, 1)
        best = min(tried, key=tried.get)
        torch.set_num_threads(best)
        per = tried[best]
        npos = int(max(64, min(H * W, budget_s / per)))
        res, n, dt = tc.decode_positions(ssd, yc, hp, tp, enc["strings"][0][0], tables, orc.reference_rans_decoder(), max_positions=npos)
        # The sample against the GPU's decode of the same string, row by row.  The two evaluate the scales in different fp32 orders: a
        # scale within fp32 noise of a table threshold picks another CDF on one side, after which the two rANS decoders read different
        # symbols for the rest of the image (the reference warns about exactly this between its own CPU and GPU runs,
        # spatiotemporalpriors.py:966-970).  Reported: how many leading rows agree to 1e-4, and the worst difference inside them.
        rows = n // W                                        # complete rows decoded by the sample
        got = (res[:, :, :rows] + yc[:, :, :rows]).numpy()
        ref = y_hat_gpu[:, :, :rows].float().cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-30) if rows else 1.0
        row_err = np.abs(got - ref).max(axis=(0, 1, 3)) / scale if rows else np.zeros(0)
        agree = int(np.argmax(row_err > 1e-4)) if (row_err > 1e-4).any() else rows
        err = float(row_err[:agree].max()) if agree else None
    finally:
        torch.set_num_threads(old)
    per = dt / n
    return {"value": 1.0 / (per * H * W), "unit": "frames/s", "cores": best, "kind": "torch-cpu", "cpu_model": _cpu_model(), "host_physical_cores": cores,
            "us_per_position_by_threads": {str(k): v * 1e6 for k, v in tried.items()},
            "positions_timed": n, "seconds": dt, "us_per_position": per * 1e6, "positions_per_frame": H * W,
            "decode_loop_s_per_frame_extrapolated": per * H * W, "sample_rows": rows, "sample_rows_agreeing_with_gpu": agree,
            "agreeing_rows_max_rel_diff": err,
            "sample": f"the first {n} of {H * W} positions of one 1080p P frame's raster-order decoding loop (spatiotemporalpriors.py:1015-1054) on torch "
                      f"{torch.__version__} CPU operators, {best} threads (the fastest of {sorted(tried)} on this {cores}-core host), with the reference's own RansDecoder (oracle/_ref): {per * 1e6:.0f} us per position "
                      f"-> {per * H * W:.1f} s for the loop of one frame; value = 1 / that (decode loop ONLY: the reference's encoder walks the same loop, "
                      f"SURVEY.md 3.2: 13 s encode + 39 s decode per frame), i.e. an upper bound of the reference's CPU frames/s"}


def bench_eval(args):
    """--config eval: BASELINE.json configs[3] -- the evaluation loop of stem/evalSTEM.py (evaluation.eval_gop) on ONE synthetic
    1920x1080 sequence, GOP 12: a step = one GOP = the I frame through mbt2018's compress / decompress + 11 P frames through
    getY -> forward -> compress -> decompress -> getX with host rANS.  value = frames/s of the whole loop (encode AND decode of
    every frame, as the script runs them back to back).  The path is bound by the LATENCY of one raster position (four dependent
    matrix-vector products + a host round trip for the symbols), not by HBM or the matrix pipe: `roofline.bound` says so."""
    from spatiotemporalentropymodel_amd import _lib, evaluation
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    from spatiotemporalentropymodel_amd.zoo import models
    _lib.hip()
    assert args.gpus == 1, "the evaluation loop is sequential in the frames of a sequence: one GPU (replicas only)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    Hh, Ww, GOP = args.eval_height, args.eval_width, args.eval_gop
    imodel = closed_form_fill_(models["mbt2018"](quality=4)).to(dev).eval()
    if args.eval_latent_scale != 1.0:
        # a lower operating point without trained weights: the last analysis layer scaled down (and the first synthesis layer up), so
        # that most of a position's 192 symbols are zero as at a trained model's rates; the arithmetic per position is unchanged
        with torch.no_grad():
            imodel.g_a[6].weight.mul_(args.eval_latent_scale)
            imodel.g_a[6].bias.mul_(args.eval_latent_scale)
            imodel.g_s[0].weight.mul_(1.0 / args.eval_latent_scale)
    imodel.update(force=True)
    stem = closed_form_fill_(SpatioTemporalPriorModel_Res()).to(dev).eval()
    stem.update(force=True)
    yy, xx = torch.meshgrid(torch.arange(Hh, device=dev), torch.arange(Ww, device=dev), indexing="ij")
    frames = [torch.stack([0.5 + 0.4 * torch.sin((xx + 3 * t) / (40.0 + 10 * c)) * torch.cos((yy + t) / (55.0 - 5 * c)) for c in range(3)])
              for t in range(GOP)]
    for _ in range(args.warmup):
        evaluation.eval_gop(imodel, stem, frames, gop=GOP, with_msssim=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runs = [evaluation.eval_gop(imodel, stem, frames, gop=GOP, with_msssim=False) for _ in range(args.steps)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [f for r in runs for f in r["frames"]]
    P = [f for f in per if f["type"] == "P"]
    I = [f for f in per if f["type"] == "I"]
    mean = lambda xs: float(np.mean(xs)) if xs else None        # noqa: E731
    npos = ((Hh + 63) // 64 * 4) * ((Ww + 63) // 64 * 4)
    dec_p = mean([f["decoding_time"] for f in P])
    last = runs[-1]["frames"]
    res = {
        "metric": "frames/s", "value": GOP * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[3]: evaluation loop of stem/evalSTEM.py (evaluation.eval_gop), one {Ww}x{Hh} sequence, GOP {GOP}: I frame by mbt2018(192,192) "
                               "compress / decompress, P frames by SpatioTemporalPriorModel_Res(256,192) getY / forward / compress / decompress / getX, host rANS; "
                               "closed-form (untrained) weights: every one of the 192 symbols of a position is non-trivial (5.7 bpp), nothing like a trained model's 0.08 bpp",
                   "frames_per_step": GOP, "p_frames_per_step": GOP - 1, "latent_positions_per_frame": npos, "latent_scale": args.eval_latent_scale,
                   "not_in_the_timed_loop": "MS-SSIM (evaluation.ms_ssim: a host-side reporting metric of the script, next to the codec path; PSNR is computed)",
                   "decoder": "persistent kernel (csrc/ar_persistent.hip)" if RUNTIME().ar_persistent else "per-position loop (csrc/ar.hip)",
                   "i_frame": {"encode_s": mean([f["encoding_time"] for f in I]), "decode_s": mean([f["decoding_time"] for f in I]), "bpp": mean([f["bpp"] for f in I])},
                   "p_frame": {"encode_s": mean([f["encoding_time"] for f in P]), "decode_s": dec_p, "bpp": mean([f["bpp"] for f in P]),
                               "estimate_bpp": mean([f["estimate_bpp"] for f in P]), "psnr_db": mean([f["psnr"] for f in P])},
                   "psnr_ave": runs[-1]["psnr_ave"], "bpp_ave": runs[-1]["bpp_ave"]},
        "roofline": {"bound": "latency", "kernel": "ar_decode_persistent_kernel: the raster-order loop of one P frame (context 5x5 product, EPM.0, EPM.2, EPM.4 per position, "
                                                   "symbols from the host decoder through a pinned mailbox); the frame's decode time also holds HD / TPM / g_s (a few ms)",
                     "achieved": npos / dec_p, "peak": 1e6 / 2.6, "unit": "positions/s", "frac": (npos / dec_p) / (1e6 / 2.6),
                     "us_per_position": dec_p / npos * 1e6,
                     "peak_definition": "one position's ARITHMETIC alone, 2.6 us: the four dependent products on 32 resident workgroups with the weights in registers "
                                        "(phase stamps of the instrumented library, profiles/r05f_eval_1080p_persistent_phases.log: ctx 0.42 + h1 0.83 + h2 0.74 + gp 0.62 us); "
                                        "the rest of a position is hand-overs between the products (4.9 us) and the host's symbol decoding + mailbox round trip (6.5 us)",
                     "traffic": None}}
    if not args.no_cpu_baseline:
        k = next(i for i, f in enumerate(last) if f["type"] == "P")
        pf = last[k]
        enc = {"strings": pf["strings"], "shape": pf["shape"]}
        res["cpu_baseline"] = _cpu_baseline_eval(stem, enc, last[k - 1]["y_conditioned"], pf["y_conditioned"])
    _emit(res)


def _emit(res):
    """the ONE JSON line, as the LAST line of stdout: RCCL prints its version banner through C stdio, which (not a terminal) is
    flushed at exit, i.e. after a Python print -- flush it first"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(res), flush=True)


def launch_ranks(n, argv):
    """`bench.py --gpus N` typed without a launcher: start the N ranks here.  This parent never touches the GPU (no HIP call,
    no torch.cuda query): it only spawns `python bench.py <same arguments>` N times with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (children are fresh processes, nothing is exec'ed over a GPU-initialised one), forwards
    rank 0's stdout (the ONE JSON line), sends the other ranks' output to stderr, and returns non-zero if any rank failed --
    the remaining ranks are then terminated by PID so that a dead peer cannot leave the others hanging in a collective."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, pending = 0, set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    out = procs[0].stdout.read().decode()              # rank 0 prints one line at the very end: the pipe cannot fill up before
    for line in out.splitlines():                      # stdout carries the JSON line only (gloo / RCCL banners go to stderr)
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr)
    sys.stdout.flush()
    if rc == 0 and not any(line.lstrip().startswith("{") for line in out.splitlines()):
        print("bench.py: rank 0 finished without a result line", file=sys.stderr)
        rc = 1
    return rc


def rendezvous_only(args):
    """--rendezvous-only: the distributed plumbing of a run without the workload -- process group from the environment, one
    sum all-reduce through distributed.all_reduce_sum_ (device tensor when there is a GPU), the max-over-ranks reduction
    the timing uses, a barrier; rank 0 prints one JSON line.  This is what the CPU test of the launcher runs (gloo)."""
    import torch.distributed as dist
    from spatiotemporalentropymodel_amd import distributed as D
    rank, world, local = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    t = torch.full((1 << 16,), float(rank + 1), device=dev)
    D.all_reduce_sum_(t)
    slowest = D.max_over_ranks(float(rank), dev)
    D.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "n_gpus": world, "backend": dist.get_backend() if dist.is_initialized() else None,
                          "device": str(dev), "all_reduce_sum": float(t[0]), "expected_sum": world * (world + 1) / 2,
                          "max_over_ranks": slowest}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="stem", choices=["stem", "roi", "eval"], help="stem = BASELINE configs[1] (the metric's workload, default); "
                    "roi = configs[4], the variable-rate GOP iteration; eval = configs[3], the evaluation loop (compress / decompress with host rANS) on one 1080p GOP")
    ap.add_argument("--eval-height", type=int, default=1080)
    ap.add_argument("--eval-width", type=int, default=1920)
    ap.add_argument("--eval-gop", type=int, default=12)
    ap.add_argument("--eval-latent-scale", type=float, default=1.0, help="--config eval: scale of the I-frame model's last analysis layer (1.0: the closed-form "
                    "weights as they are, ~5.7 bpp; 0.02: most symbols zero, a trained model's operating range)")
    ap.add_argument("--roi-batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10; 2 for --config roi)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 6 for --config roi: the first process that runs the variable-rate models on a box needs about that many iterations before an iteration takes what it takes in every later process -- 1.40 s after two, 0.95-1.0 s after six)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latents", default="prefetch", choices=["first", "prefetch"], help="prefetch (default): getY of frame t + 1 runs on a second "
                    "stream while P-frame step t runs (trainer.LatentPrefetcher); first: getY of all 7 frames before the P-frame steps")
    ap.add_argument("--latents-ahead", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("STEM_BENCH_PIPELINE", "0")), help="1: the next septuplet's first two "
                    "latents are computed during the current septuplet's last two P-frame steps (a training loop whose loader holds the next batch); "
                    "0 (default): every septuplet enqueues its own two opening transforms -- with the launch tape the host runs a P-frame step ahead of "
                    "the GPU, so they overlap the previous septuplet's tail anyway (12.12 against 12.15 ms per step, profiles/r05_ab_pipeline.log)")
    ap.add_argument("--generic", action="store_true", help="run the P-frame step through nn.Module / autograd / torch-style optimiser "
                    "calls (selfcheck.p_frame_step) instead of the explicit fused schedule (trainer.FusedPFrameStep)")
    ap.add_argument("--graph", action="store_true", help="replay the P-frame step from its hipGraph (graphs.GraphedPFrameStep) instead of "
                    "issuing it kernel by kernel: 2 ms instead of 11-22 ms of host time per step, same GPU time (DESIGN.md §7)")
    ap.add_argument("--tape", type=int, default=int(os.environ.get("STEM_BENCH_TAPE", "1")), help="1: the P-frame step through the launch tape "
                    "(tape.TapedPFrameStep: recorded once, replayed by csrc/tape.hip; bit-identical, ~3 instead of ~11 ms of host time per step)")
    ap.add_argument("--rendezvous-only", action="store_true", help="set up the ranks, run one all-reduce and the timing reduction, exit "
                    "(launcher / process-group check without the workload; works without a GPU)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"roi": 2, "eval": 2}.get(args.config, 10)
    if args.warmup is None:
        args.warmup = {"roi": 6, "eval": 1}.get(args.config, 3)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # a rank of a multi-GPU run pins itself to the cores of its GPU's NUMA node before anything touches the GPU (sysfs only)
    global PINNED_CPUS
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from spatiotemporalentropymodel_amd.distributed import pin_rank_to_gpu_cores
        PINNED_CPUS = pin_rank_to_gpu_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])))
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.config == "roi":
        return bench_roi(args)
    if args.config == "eval":
        return bench_eval(args)

    from spatiotemporalentropymodel_amd import _lib
    from spatiotemporalentropymodel_amd import distributed as D
    _lib.hip()                                    # no HIP library -> fail loudly, nothing to measure
    _apply_bench_tuning(_lib)
    rank, world, local = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    # The library's tuned schedule (trainer.tuned_schedule; the environment overrides its two settings):
    #  * stream priorities: the P-frame step's own streams -- a dedicated compute stream instead of torch's default stream, the
    #    weight-gradient / branch / auxiliary streams -- at HIP's high priority, the latent-prefetch stream at normal priority: the
    #    command processor then dispatches the step's small kernels ahead of the long analysis-transform kernels (22.4 against
    #    22.7 ms per step on the same box).  STEM_STREAM_PRIO="" runs everything at one priority;
    #  * the latent-prefetch stream is confined to 160 of the 256 CUs (192 until round 4; hipExtStreamCreateWithCUMask): a running workgroup of the
    #    long analysis-transform kernels cannot be pre-empted, so without the mask the step's short, high-priority launches wait for
    #    CUs to drain (HE.2's 25 us launch took 200 us next to g_a.2).  Same box: 15.74 -> 15.39-15.43 ms per step (208 CUs 15.73,
    #    160 CUs 15.58, 144 CUs 16.4); with the round's earlier, slower kernels the same mask cost time.  STEM_STREAM_CUMASK="" removes it.
    from spatiotemporalentropymodel_amd.trainer import tuned_schedule
    torch.cuda.set_device(dev)
    compute = tuned_schedule(dev)          # before anything creates a stream; entered around the steps below

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
    for i, m in enumerate((imodel.gaussian_conditional, stem.entropy_bottleneck, stem.gaussian_conditional)):
        m.noise_seed = seed * 7919 + 131 * i          # one Philox stream per noise source, the same in every run of this rank
    opt, aux_opt = configure_optimizers(stem, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))
    # gradient slices are all-reduced (RCCL, side stream) as backward finishes each module group
    # (also at world size 1 when STEM_DIST_SINGLE=1 created a one-rank RCCL group: same calls, same stream ordering)
    reducer = D.OverlappedGradReducer(opt.flat).attach(stem.engine()) if D.dist.is_initialized() else None
    crit = EMLoss()
    frames = synthetic_septuplet(BATCH, SIZE, seed, dev)

    # HIP-event probe around the dominant kernel (g_a.2 + fused GDN, the 192-ch 5x5 stride-2 analysis conv) on the
    # stream it is launched on (= torch's current stream, which is what the C ABI receives)
    probe, probe0 = [], []
    if os.environ.get("STEM_BENCH_NOPROBE", "0") != "1":
        imodel.g_a.probe = {2: probe, 0: probe0}
    f16_chain = RUNTIME().analysis_f16x3

    # The loop body of stem/trainSTEM.py:174-226 is the package's (trainer.SeptupletTrainer.train_septuplet: getY of the frames
    # on the prefetch stream, one P-frame optimisation step per later frame through the launch tape / the explicit schedule / the
    # generic nn.Module route); rand = 1.0 keeps all seven frames of every septuplet (the metric's unit), i.e. the script's
    # temporal subsampling (:175-182, trainer.subsample_septuplet) is not drawn.
    # --graph (single device): the P-frame step (zero_grad .. aux Adam, ~160 launches) is replayed from ONE hipGraph per
    # step (graphs.GraphedPFrameStep); getY stays eager so that the HIP-event probe can bracket its dominant kernel.  Data
    # parallel runs are always eager: the RCCL exchanges are issued from inside backward.
    use_graph = world == 1 and args.graph
    from spatiotemporalentropymodel_amd.trainer import SeptupletTrainer
    route = "generic" if args.generic else ("taped" if args.tape else "fused")
    septuplets = SeptupletTrainer(imodel, stem, opt, aux_opt, route="fused" if use_graph else route, prefetch=args.latents == "prefetch",
                                  ahead=args.latents_ahead, reducer=reducer, grad_scale=1.0 / world, criterion=crit)
    fused_step, prefetch = septuplets.step_fn, septuplets.prefetch
    graphed = None
    if use_graph:
        from spatiotemporalentropymodel_amd.graphs import GraphedPFrameStep
        graphed = GraphedPFrameStep(stem, crit, opt, aux_opt, (SIZE, SIZE))
        fused_step = None

    def _mark(t, out, oc, aux, gn):
        if TIMELINE is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            TIMELINE.append(ev)
    septuplets.on_step = _mark

    def one_step():
        if graphed is None:
            # the synthetic frames were generated before the timed region.  --pipeline 1: the NEXT septuplet (a loader's following
            # batch; here the same synthetic tensors) is named too, so that its frames 0 and 1 go through the analysis transform
            # during this septuplet's last two P-frame steps instead of in front of its own first one
            return septuplets.train_septuplet(frames, rand=1.0, next_images=frames if (args.pipeline and prefetch is not None) else None)[-1][0]
        if prefetch is not None:
            prefetch.start(frames, frames_ready=True, next_frames=frames if args.pipeline else None)
            ys = None
            y_cond = prefetch.get(0)[1]
        else:
            with torch.no_grad():
                ys = [imodel.getY(f) for f in frames]
            y_cond = ys[0][1]
        last = None
        for t in range(1, FRAMES):
            y_cur = prefetch.get(t)[0] if prefetch is not None else ys[t][0]
            out, oc, aux, gn = graphed.step(y_cur, y_cond)
            y_cond = out["y_hat"]                # graph mode: static output buffer, copied into the y_cond input by the next step()
            last = oc
            _mark(t, out, oc, aux, gn)
        return last

    # STEM_BENCH_TIMELINE=1 (dev): where the analysis-transform launches of the prefetch stream fall between the P-frame steps'
    # ends -- HIP events only, no profiler in the way
    TIMELINE = [] if os.environ.get("STEM_BENCH_TIMELINE") else None

    with compute:
        for _ in range(args.warmup):
            one_step()
        probe.clear()
        probe0.clear()
        # device idle BEFORE the barrier too: the gradient exchanges of the last warm-up step run on libstem_dp's own communicator,
        # the barrier on torch's -- two communicators must not have collectives in flight in rank-dependent order
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if TIMELINE is not None:
            TIMELINE.clear()
            tl0 = torch.cuda.Event(enable_timing=True)
            tl0.record()
        for _ in range(args.steps):
            last = one_step()
        torch.cuda.synchronize()
        if TIMELINE is not None and rank == 0:
            marks = [("P-step end", tl0.elapsed_time(e)) for e in TIMELINE] + [(f"g_a.2 [{tl0.elapsed_time(a):8.3f} .. ", tl0.elapsed_time(b)) for a, b in probe]
            for name, t in sorted(marks, key=lambda m: m[1])[:60]:
                print(f"timeline {name} {t:8.3f} ms", file=sys.stderr)
        D.barrier()
        dt = D.max_over_ranks(time.perf_counter() - t0, dev)

    # launch durations inside the timed region: with --latents prefetch these launches share the chip with the P-frame step running
    # on the compute stream, so they measure the schedule, not the kernel
    kern_ms_overlap = float(np.mean([a.elapsed_time(b) for a, b in probe])) if probe else float("nan")
    kern0_ms_overlap = float(np.mean([a.elapsed_time(b) for a, b in probe0])) if probe0 else float("nan")
    n_overlap = len(probe)
    loss = float(last["loss"].detach())
    # the kernels' own rate: the same launches (same frames, same weights) with the chip to themselves, timed by the same HIP events on
    # the launching stream, right after the timed region (prefetch: 3 x 7 launches; latents first: the timed region's launches already are)
    if prefetch is not None:
        probe.clear()
        probe0.clear()
        with torch.no_grad():
            for _ in range(3):
                for f in frames:
                    imodel.getY(f)
        torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in probe])) if probe else float("nan")
    # ---- what a data-parallel run says about ITSELF (every rank takes part; outside the timed region) -------------------------------
    # rccl_nranks: the rank count the communicator that carried the gradients reports (ncclCommCount of libstem_dp's own
    # communicator; the torch group's size on the other routes) -- n_gpus alone is only WORLD_SIZE read back from the environment;
    # replicas_identical: MIN == MAX over ranks of a 64-bit checksum of the parameters' bits after the timed region;
    # STEM_BENCH_VERIFY=1: one P-frame step through the reducer on every shard against the same step over the global batch on rank 0.
    dp_report = None
    if reducer is not None:
        if fused_step is not None:
            fused_step.finish()
        torch.cuda.synchronize()
        reducer.check()                       # a failed exchange aborts this rank (non-zero exit; the launcher stops the peers)
        same, csum = D.replicas_identical(opt.flat.data)
        dp_report = {"rccl_nranks": reducer.rccl_nranks, "exchange_route": reducer.route, "collectives_per_step": None,
                     "replicas_identical": same, "replica_checksum_rank0": f"{csum & (2 ** 64 - 1):016x}", "loss_vs_single_rank": None}
        dp_report["collectives_per_step"] = reducer.collectives / max(1, (args.steps + args.warmup) * (FRAMES - 1))
        if os.environ.get("STEM_BENCH_VERIFY", "0") == "1":
            from spatiotemporalentropymodel_amd.selfcheck import dp_step_vs_full_batch

            def make_models():
                torch.manual_seed(1234)
                return models["mbt2018"](quality=4).to(dev).eval(), SpatioTemporalPriorModel_Res().to(dev)

            v = dp_step_vs_full_batch(make_models, lambda r: synthetic_septuplet(BATCH, SIZE, D.shard_seed(1234, r), dev)[:2], rank, world, dev, SIZE)
            dp_report["loss_vs_single_rank"] = {"loss_dp_mean_over_ranks": v["loss_dp"], "loss_full_batch_rank0_alone": v["loss_full"],
                                                "rel_diff": v["loss_rel"], "averaged_gradient_max_rel_diff": v["grad_rel"],
                                                "global_batch": BATCH * world, "rccl_nranks": v["rccl_nranks"],
                                                "what": "one P-frame step (stem/trainSTEM.py:194-218) from identical initial weights, "
                                                        "frames of every rank's seed, closed-form noise sliced per rank"}
    if rank != 0:
        return
    kern0_ms = float(np.mean([a.elapsed_time(b) for a, b in probe0])) if probe0 else float("nan")
    flop = GA2_FLOP_PER_FRAME * BATCH
    tfile = os.path.join(REPO, "profiles", "hbm_traffic.json")
    tj = json.load(open(tfile)) if os.path.exists(tfile) else {}
    # g_a.0 + GDN g_a.1 (csrc/c4gdn_f16x3.hip since round 3; igemm.hip's fp32-MFMA kernel with STEM_C4GDN_F16X3=0), measured the same
    # way; the probe brackets the NCHW -> NHWC4 layout kernel (~20 us) and the convolution kernel
    flop0 = GA0_FLOP_PER_FRAME * BATCH
    c4_f16 = RUNTIME().first_layer_f16x3 and f16_chain
    exec0 = F16_PRODUCTS * (2 * 192 * 80 * 128 * 128 + 2 * 192 * 192 * 128 * 128) * BATCH       # conv K: 75 real products in 80 slots (128 until round 5), GDN K = 192; x3 products
    peak0 = PEAK_F16_MFMA_TFLOPS if c4_f16 else PEAK_FP32_MFMA_TFLOPS
    work0 = exec0 if c4_f16 else flop0
    first_line = {"bound": "mfma",
                  "kernel": ("c4gdn_f16x3_kernel<6> = g_a.0 conv (3->192, 5x5 s2, 256^2->128^2, B=16) + GDN g_a.1 in one kernel: transposed "
                             "contractions, squared outputs handed accumulator -> B operand in registers, 3 fp16 MFMAs per fp32 product") if c4_f16 else
                            "igemm_kernel<128,192,32,96,C4,FUSE> = g_a.0 + fused GDN g_a.1 (v_mfma_f32_32x32x2_f32)",
                  "achieved": work0 / (kern0_ms_overlap * 1e-3) / 1e12, "peak": peak0, "unit": "TFLOP/s",
                  "frac": work0 / (kern0_ms_overlap * 1e-3) / 1e12 / peak0, "flop_per_launch": work0,
                  "avg_launch_ms": kern0_ms_overlap, "launches_timed": len(probe0) if prefetch is None else n_overlap,
                  "isolated": {"achieved": work0 / (kern0_ms * 1e-3) / 1e12, "frac": work0 / (kern0_ms * 1e-3) / 1e12 / peak0, "avg_launch_ms": kern0_ms,
                               "launches_timed": len(probe0)},
                  "useful_flop_per_launch": flop0, "useful_tflops": flop0 / (kern0_ms_overlap * 1e-3) / 1e12,
                  "useful_tflops_isolated": flop0 / (kern0_ms * 1e-3) / 1e12, "traffic": tj.get("g_a0_c4gdn_bytes_per_launch")}
    if f16_chain:
        # Dominant kernel: g_a.2 + GDN on the fp16 matrix cores.  Every fp32 product is THREE fp16 MFMA products (conv_f16x3.hip), so
        # the matrix pipe executes 3x the algorithmic flop (convolution and GDN contraction); `achieved` / `frac` are that executed fp16 rate against
        # the dense fp16 peak (never mixed into an fp32 fraction), taken on the launches INSIDE the timed region (the
        # headline is the in-region figure); the same launches alone on the chip are under "isolated", the algorithmic
        # (fp32-equivalent, "useful") rate next to both.
        executed = F16_PRODUCTS * GA2_FLOP_PER_FRAME * BATCH        # convolution and the fused GDN contraction, both on the fp16 instruction
        in_ms = kern_ms_overlap if prefetch is not None else kern_ms
        roof = {"bound": "mfma", "kernel": "conv_f16x3_kernel<128> = g_a.2 conv (192->192, 5x5 s2, 128^2->64^2, B=16) + fused GDN g_a.3: operands "
                                           "pre-split into 2 scaled fp16 planes, 3 fp16 MFMAs (v_mfma_f32_16x16x32_f16 in this launch) per fp32 product, fp32 accumulate",
                "achieved": executed / (in_ms * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": executed / (in_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                "flop_per_launch": executed, "flop_definition": "executed fp16 MFMA flop = 3 x algorithmic flop of the convolution and of the fused GDN contraction",
                "avg_launch_ms": in_ms, "launches_timed": n_overlap if prefetch is not None else len(probe),
                # the launches of the timed region run on the latent-prefetch stream, which the schedule confines to a subset of the
                # CUs (trainer.SCHEDULE_DEFAULTS): the same in-region rate against the peak of THOSE CUs
                "cu_mask_cus": _masked_cus(),
                "frac_of_masked_cus": executed / (in_ms * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS * _masked_cus() / 256.0),
                "isolated": {"achieved": executed / (kern_ms * 1e-3) / 1e12, "frac": executed / (kern_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                             "avg_launch_ms": kern_ms, "launches_timed": len(probe)},
                "useful_flop_per_launch": flop, "useful_tflops": flop / (in_ms * 1e-3) / 1e12, "useful_tflops_isolated": flop / (kern_ms * 1e-3) / 1e12,
                "useful_isolated_vs_fp32_mfma_peak": flop / (kern_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                "timing_note": ("achieved / frac / avg_launch_ms: the launches of the timed region, which run on the latent-prefetch stream "
                                "(confined to the CUs of config.stream_cu_masks; the isolated launches below run on the unmasked compute stream) next to "
                                "a P-frame step (HIP events on the launching stream); isolated: the same launches (same frames, weights) repeated 3 x 7 "
                                "times right after the timed region with the chip to themselves") if prefetch is not None else
                               "the launches of the timed region run alone (latents first): in-region = isolated",
                "clock_note": "counters of this kernel: profiles/r06_pmc_ga2.csv (DESIGN.md 7); the guide's sustained 16-bit MFMA rate on "
                              "random data is ~1250 TFLOP/s (power-limited clock)",
                "traffic": tj.get("g_a2_f16x3_bytes_per_launch"),
                "traffic_source": ("profiles/hbm_traffic.json (tools/update_hbm_traffic.py): %s, measured in %s from %s -- separate rocprofv3 --pmc passes "
                                   "(FETCH_SIZE x2 + WRITE_SIZE, tools/debug/prof_tcc.sh over tools/debug/f16x3_prof.py planes); NOT re-measured in this run"
                                   % (tj.get("g_a2_f16x3", {}).get("kernel", "?").split(" = ")[0], tj.get("g_a2_f16x3", {}).get("round", "round 3"),
                                      tj.get("g_a2_f16x3", {}).get("source", "the round's profile run")))}
    else:
        achieved = flop / (kern_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "igemm_kernel<128,192,32,96,FUSE> = g_a.2 conv (192->192, 5x5 s2, 128^2->64^2, B=16) + fused GDN g_a.3",
                "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "flop_per_launch": flop, "avg_launch_ms": kern_ms, "launches_timed": len(probe), "traffic": tj.get("g_a2_bytes_per_launch"),
                "traffic_source": "profiles/hbm_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + "
                                  "WRITE_SIZE, tools/kernel_bench.py --only g_a.2+gdn), NOT re-measured in this run"}
    res = {
        "metric": "frames/s", "value": FRAMES * BATCH * world * args.steps / dt, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: SpatioTemporalPriorModel_Res(256,192) training step on mbt2018(192,192) latents, "
                               "16 septuplets x 7 frames x 256x256 per GPU, EMLoss, clip 1.0 + Adam 1e-4 / aux Adam 1e-3",
                   "per_gpu_batch": BATCH, "global_batch": BATCH * world, "frames_per_step": FRAMES * BATCH * world,
                   "p_frame_steps_per_step": FRAMES - 1, "parallelism": f"dp{world}", "final_loss_bpp": loss,
                   "latents": "getY of frame t + 1 on a second stream during P-frame step t (trainer.LatentPrefetcher)" if prefetch is not None
                              else "getY of all 7 frames before the P-frame steps",
                   "analysis_transform": "fp16 matrix cores, two fp16 planes per operand, 3 products per fp32 product (conv_f16x3.hip)" if f16_chain else "fp32 MFMA",
                   "stream_priorities": RUNTIME().stream_prio,
                   "stream_cu_masks": RUNTIME().stream_cumask or "none",
                   "plan_selectors": os.environ.get("STEM_BENCH_TUNING", "") or "library defaults",
                   "rank0_host_cores": (f"{len(PINNED_CPUS)} cores of the GPU's NUMA node ({PINNED_CPUS[0]}..{PINNED_CPUS[-1]})" if PINNED_CPUS else "not pinned"),
                   "loop": "trainer.SeptupletTrainer.train_septuplet (stem/trainSTEM.py:174-226), all 7 frames of every septuplet",
                   "launch": "hipGraph replay per P-frame step" if use_graph else
                             (("explicit fused schedule (trainer.FusedPFrameStep)" + (" replayed from its launch tape by the native executor (tape.TapedPFrameStep, csrc/tape.hip)"
                                                                                          if args.tape else "")) if fused_step is not None else "generic nn.Module / autograd route")},
        "roofline": roof,
        "roofline_first_layer": first_line,
        "useful_tflops_per_gpu": USEFUL_FLOP_PER_STEP / (dt / args.steps) / 1e12,
    }
    if dp_report is not None:
        res["config"]["data_parallel"] = dp_report
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    _emit(res)
    if dp_report is not None:
        lv = dp_report["loss_vs_single_rank"]
        if not dp_report["replicas_identical"] or dp_report["rccl_nranks"] != world or (lv is not None and not (lv["rel_diff"] < 1e-5 and lv["averaged_gradient_max_rel_diff"] < 1e-4)):
            print("bench.py: the data-parallel run failed its own check (config.data_parallel)", file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
