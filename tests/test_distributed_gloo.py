"""CPU, world_size 2, gloo: the data-parallel plumbing of bench.py / distributed.py.

Checks (a) bucketed sum all-reduce of the flat gradient buffer + grad_scale, (b) parameter broadcast,
(c) max-over-ranks timing aggregation, and (d) the claim DP relies on: because EMLoss normalises by the LOCAL
pixel count, the mean of per-rank gradients equals the gradient of the global batch (verified with the CPU
oracle's STEM forward/backward on a 2-sample batch split over 2 ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    torch.set_num_threads(2)
    from spatiotemporalentropymodel_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    out = {}
    # (a) bucketed all-reduce
    n = 1_000_003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = D.FlatGradReducer(g, n_buckets=4, min_bucket_elems=1000)
    assert len(red.ranges) == 4 and red.ranges[0][0] == 0 and red.ranges[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(red.ranges, red.ranges[1:]))
    red.all_reduce()
    out["allreduce_ok"] = bool(torch.equal(g, torch.arange(n, dtype=torch.float32) * 3)) and red.grad_scale == 0.5
    # (b) broadcast
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    D.broadcast_parameters(lin)
    out["bcast_ok"] = bool((lin.weight == 1.0).all())
    # (c) timing aggregation and per-rank seeds
    out["max"] = D.max_over_ranks(1.0 + rank, "cpu")
    out["seed"] = D.shard_seed(1234, rank)
    # (d) mean of per-rank gradients == global-batch gradient (oracle)
    import stem_oracle as orc
    from spatiotemporalentropymodel_amd.weights import closed_form_input, closed_form_tensor
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_oracle_vs_golden import _stem_keys
    ssd = {k: closed_form_tensor(k, s).numpy() for k, s in _stem_keys(64, 96).items()}
    B, ls = 2, 4
    y_cur = closed_form_input("dp:y", (B, 96, ls, ls), -4, 4).numpy()
    y_cond = closed_form_input("dp:c", (B, 96, ls, ls), -4, 4).numpy()
    noise = {"z": closed_form_input("dp:nz", (B, 64, 1, 1), -.5, .5).numpy(), "q": closed_form_input("dp:nq", (B, 96, ls, ls), -.5, .5).numpy(),
             "lik": closed_form_input("dp:nl", (B, 96, ls, ls), -.5, .5).numpy()}

    def grads(sl):
        keep = {}
        o = orc.stem_forward(ssd, y_cur[sl], y_cond[sl], residual=True, training=True, noise={k: v[sl] for k, v in noise.items()}, keep=keep)
        npix = y_cur[sl].shape[0] * 64 * 64
        return orc.stem_backward(ssd, keep, o["lik_y"], o["lik_z"], npix)

    mine = grads(slice(rank, rank + 1))
    names = sorted(mine)
    flat = torch.from_numpy(np.concatenate([mine[k].ravel() for k in names]))
    D.FlatGradReducer(flat, n_buckets=3, min_bucket_elems=1000).all_reduce()
    flat *= 0.5
    if rank == 0:
        full = grads(slice(0, 2))
        ref = np.concatenate([full[k].ravel() for k in names])
        err = np.abs(flat.numpy() - ref).max() / np.abs(ref).max()
        out["dp_grad_rel_err"] = float(err)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


@pytest.mark.timeout(600)
def test_dp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["allreduce_ok"] and res[r]["bcast_ok"]
        assert res[r]["max"] == 2.0 and res[r]["seed"] == 1234 + r
    assert res[0]["dp_grad_rel_err"] < 1e-5, res[0]


def _worker_overlap(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    torch.set_num_threads(2)
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.optim import FlatParameters
    D.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = SpatioTemporalPriorModel_Res(64, 96)
    main = sorted([(n, p) for n, p in m.named_parameters() if not n.endswith(".quantiles")], key=lambda t: t[0])
    flat = FlatParameters(main)
    red = D.OverlappedGradReducer(flat, min_bytes=0)            # every module group is exchanged as soon as it is reported
    eng = m.engine()
    red.attach(eng)
    flat.grad.copy_(torch.arange(flat.numel, dtype=torch.float32) % 1000 * (rank + 1))
    expect = torch.arange(flat.numel, dtype=torch.float32) % 1000 * 3
    # replay the order in which StemEngine.backward reports finished module groups
    eb = m.entropy_bottleneck
    groups = [[p for l in eng.EPM for p in (l.mod.weight, l.mod.bias)], [eng.CTX.mod.weight, eng.CTX.mod.bias],
              [p for l in eng.TPM for p in (l.mod.weight, l.mod.bias)],
              [p for l in eng.HD for p in (l.mod.weight, l.mod.bias)] + eb._tensors14(),
              [p for l in eng.HE for p in (l.mod.weight, l.mod.bias)]]
    for grp in groups[:3]:
        eng.grad_ready_hook(grp)
    # HD and entropy_bottleneck are NOT adjacent in the flat (name-sorted) buffer: they must go as two runs, and the
    # HE / TPM / context tensors lying between them must not be touched by that call
    he_slice = slice(flat.offsets[[n for n, _ in main].index("HE.0.bias")], None)
    before = flat.grad[he_slice][:256].clone()
    calls = red.calls
    eng.grad_ready_hook(groups[3])
    ok_runs = red.calls - calls == 2 and torch.equal(flat.grad[he_slice][:256], before)
    eng.grad_ready_hook(groups[4])
    red.finish()
    ok_a, n_a = bool(torch.equal(flat.grad, expect)), red.collectives
    # default mode: neighbouring groups travel together (a run is exchanged once it holds min_bytes, the rest at finish());
    # nothing may be exchanged before it was reported, everything exactly once
    red2 = D.OverlappedGradReducer(flat, min_bytes=1 << 40).attach(eng)       # nothing reaches the threshold: one run at finish()
    flat.grad.copy_(torch.arange(flat.numel, dtype=torch.float32) % 1000 * (rank + 1))
    for grp in groups[:3]:
        eng.grad_ready_hook(grp)
    untouched = bool(torch.equal(flat.grad[he_slice][:256], before))          # HE not reported yet: still this rank's values
    eng.grad_ready_hook(groups[3])
    eng.grad_ready_hook(groups[4])
    red2.finish()
    ok_b = bool(torch.equal(flat.grad, expect)) and untouched and red2.collectives == 1 and n_a == 6 and red2.calls == 6
    # the layout configure_optimizers uses (optim.backward_layout_key): groups sit in backward-completion order, so HD and the
    # bottleneck are ONE run, nothing small is left over for finish(), and the parameter ORDER (names / params / state dicts) is
    # still the name-sorted one
    from spatiotemporalentropymodel_amd.optim import backward_layout_key
    m3 = SpatioTemporalPriorModel_Res(64, 96)
    main3 = sorted([(n, p) for n, p in m3.named_parameters() if not n.endswith(".quantiles")], key=lambda t: t[0])
    flat3 = FlatParameters(main3, layout_key=backward_layout_key)
    eng3 = m3.engine()
    red3 = D.OverlappedGradReducer(flat3, min_bytes=0).attach(eng3)
    flat3.grad.copy_(torch.arange(flat3.numel, dtype=torch.float32) % 1000 * (rank + 1))
    eb3 = m3.entropy_bottleneck
    groups3 = [[p for l in eng3.EPM for p in (l.mod.weight, l.mod.bias)], [eng3.CTX.mod.weight, eng3.CTX.mod.bias],
               [p for l in eng3.TPM for p in (l.mod.weight, l.mod.bias)],
               [p for l in eng3.HD for p in (l.mod.weight, l.mod.bias)] + eb3._tensors14(),
               [p for l in eng3.HE for p in (l.mod.weight, l.mod.bias)]]
    lows = []
    for grp in groups3:
        eng3.grad_ready_hook(grp)
        lows.append(min(flat3.offsets[flat3.params.index(p)] if False else next(o for q_, o in zip(flat3.params, flat3.offsets) if q_ is p) for p in grp))
    red3.finish()
    ok_c = (bool(torch.equal(flat3.grad, torch.arange(flat3.numel, dtype=torch.float32) % 1000 * 3)) and red3.calls == 5 and red3.collectives == 5
            and lows == sorted(lows) and lows[0] == 0 and flat3.names == [n for n, _ in main3])
    q.put((rank, {"ok": ok_a and ok_b and ok_c, "runs": bool(ok_runs), "calls": red.calls, "scale": red.grad_scale}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_overlapped_group_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["ok"], "every element must be summed exactly once"
        assert res[r]["runs"] and res[r]["calls"] == 6 and res[r]["scale"] == 0.5


class _Flat:
    """Minimal stand-in for optim.FlatParameters (grad buffer + zero_grad) so the accumulator logic runs on CPU."""

    def __init__(self, n):
        self.grad = torch.zeros(n)

    def zero_grad(self):
        self.grad.zero_()


def _clip(bufs, max_norm):
    total = torch.sqrt(sum((b.double() ** 2).sum() for b in bufs))
    coef = min(1.0, max_norm / (float(total) + 1e-6))
    for b in bufs:
        b.mul_(coef)
    return float(total)


def _gop_reference(frame_grads, aux_grads, max_norm):
    """Single-process semantics (train_stem_roi.py:533-541): G += g_t; clip(G, Gaux); Gaux += a_t."""
    G, A = torch.zeros_like(frame_grads[0]), torch.zeros_like(aux_grads[0])
    norms = []
    for g, a in zip(frame_grads, aux_grads):
        G += g
        norms.append(_clip([G, A], max_norm))
        A += a
    return G, A, norms


def _worker_gop(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    torch.set_num_threads(2)
    from spatiotemporalentropymodel_amd import distributed as D
    D.init_from_env(backend="gloo")
    gen = torch.Generator().manual_seed(7)
    T, n, na = 4, 1003, 48
    per_rank = [[torch.randn(n, generator=gen) * (3.0 if t == 1 else 0.3) for _ in range(world)] for t in range(T)]
    aux = [torch.randn(na, generator=gen) * 0.2 for _ in range(T)]
    main, side = _Flat(n), _Flat(na)
    acc = D.GopGradAccumulator([main], [side])
    acc.begin()
    norms = []
    for t in range(T):
        main.grad += per_rank[t][rank]                       # "backward" of frame t on this rank's samples
        acc.end_frame()
        norms.append(_clip([acc.running(main), acc.running(side)], 1.0))
        side.grad += aux[t]                                  # aux loss: parameters only, identical on every rank
        acc.end_aux()
    acc.finish()
    # collective control flow: a rank-local "bad loss" verdict must become everybody's verdict
    any_ok = acc.any_rank(rank == 1) is True and acc.any_rank(False) is False
    G, A, ref_norms = _gop_reference([sum(per_rank[t]) / world for t in range(T)], aux, 1.0)
    ok = bool(torch.allclose(main.grad, G, rtol=1e-6, atol=1e-7)) and bool(torch.allclose(side.grad, A, rtol=1e-6, atol=1e-7))
    q.put((rank, {"ok": ok and any_ok, "norms": bool(np.allclose(norms, ref_norms, rtol=1e-6)), "clipped": sum(v > 1.0 for v in ref_norms)}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gop_gradient_accumulator_world2_gloo():
    """Per-frame exchange + clip of the running sum == the single-process full-batch loop (variable-rate training)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gop, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["ok"] and res[r]["norms"] and res[r]["clipped"] >= 2, res


def test_overlapped_reducer_refuses_partial_or_double_coverage():
    """VERDICT r1 weak #9 / ADVICE: finish() used to re-reduce the whole buffer when a group was missing (double-counting
    the slices already exchanged); it now raises, and so does a slice reported twice.  world_size 1, CPU."""
    from spatiotemporalentropymodel_amd import distributed as D
    from spatiotemporalentropymodel_amd.optim import FlatParameters
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    named = sorted(net.named_parameters(), key=lambda t: t[0])
    flat = FlatParameters(named)
    red = D.OverlappedGradReducer(flat)
    ps = [p for _, p in named]
    red.reduce_params(ps[:2])
    red.reduce_params(ps[2:])
    red.finish()                                       # full, single coverage: fine
    red.reduce_params(ps[:2])
    with pytest.raises(RuntimeError, match="never reported.*1\\.(bias|weight)"):
        red.finish()
    red.reduce_params(ps[:3])
    red.reduce_params(ps[2:])
    with pytest.raises(RuntimeError, match="summed twice"):
        red.finish()
    red.reduce_params(ps)                              # state was reset by the failed finish()
    red.finish()


def _run_bench(*argv, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_gloo():
    """`python bench.py --gpus 2` typed bare (the driver's N=1 command form with another N): bench.launch_ranks starts the
    two ranks, which rendezvous (gloo here: no GPU), run one all_reduce_sum_ and the timing reduction; stdout of the parent
    is exactly rank 0's JSON line."""
    import json
    p = _run_bench("--gpus", "2", "--rendezvous-only")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["rendezvous"] == "ok" and rec["n_gpus"] == 2 and rec["backend"] == "gloo"
    assert rec["all_reduce_sum"] == rec["expected_sum"] == 3.0 and rec["max_over_ranks"] == 1.0


def test_bench_launcher_reports_a_failed_rank():
    """Without a GPU the workload itself refuses to run: every rank exits non-zero, and so must the launcher (no JSON line)."""
    p = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")], p.stdout


def test_backend_choice_follows_device_count(monkeypatch):
    """init_from_env: RCCL only when every rank has its own GPU; more ranks than devices (or no device) -> gloo."""
    from spatiotemporalentropymodel_amd import distributed as D
    seen = {}
    monkeypatch.setattr(D.dist, "is_initialized", lambda: False)
    monkeypatch.setattr(D.dist, "init_process_group", lambda backend, rank, world_size, **kw: seen.update(b=backend, r=rank, w=world_size, kw=kw))
    monkeypatch.setattr(D.torch.cuda, "set_device", lambda d: seen.update(dev=d))
    monkeypatch.delenv("STEM_DIST_BACKEND", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    for ndev, world, rank, want, local in ((8, 8, 5, "nccl", 5), (1, 2, 1, "gloo", 0), (0, 2, 1, "gloo", 1), (1, 1, 0, "nccl", 0)):
        monkeypatch.setattr(D.torch.cuda, "device_count", lambda n=ndev: n)
        monkeypatch.setenv("WORLD_SIZE", str(world)); monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("LOCAL_RANK", str(rank))
        seen.clear()
        assert D.init_from_env(single=True) == (rank, world, local)
        assert seen["b"] == want and seen["w"] == world, (ndev, world, seen)
        if want == "nccl" and "pg_options" in seen["kw"]:          # RCCL's own stream at high priority, like the schedule's side streams
            assert seen["kw"]["pg_options"].is_high_priority_stream
    # two nodes of 8 GPUs under torch.distributed.run (WORLD_SIZE=16, LOCAL_WORLD_SIZE=8): nothing is shared, RCCL it is;
    # 16 ranks on ONE 8-GPU node share devices and go through the host
    monkeypatch.setattr(D.torch.cuda, "device_count", lambda: 8)
    for lws, want, local in ((8, "nccl", 3), (16, "gloo", 3)):
        monkeypatch.setenv("WORLD_SIZE", "16"); monkeypatch.setenv("RANK", "11"); monkeypatch.setenv("LOCAL_RANK", "3")
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(lws))
        seen.clear()
        assert D.init_from_env() == (11, 16, local)
        assert seen["b"] == want and seen["w"] == 16, (lws, seen)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    # world size 1 without `single`: no process group at all
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "1"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.delenv("STEM_DIST_SINGLE", raising=False)
    D.init_from_env()
    assert not seen


def test_rank_pinning_reads_the_gpu_numa_topology(tmp_path, monkeypatch):
    """distributed.pin_rank_to_gpu_cores: GPU nodes of a (fake) KFD topology in node order, each with the cores of its PCI
    device's NUMA node; ranks whose GPUs share a node split its cores; unknown topology / STEM_PIN_RANKS=0 leave the mask alone."""
    from spatiotemporalentropymodel_amd import distributed as D
    root = tmp_path / "sys"
    nodes = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    # two CPU nodes, then four GPUs: two on NUMA node 0 (cpus 0-3), two on node 1 (cpus 4-7)
    layout = [(0, None), (0, None), (64, "0-3"), (64, "0-3"), (64, "4-7"), (64, "4-5,6-7")]
    for i, (simd, cpus) in enumerate(layout):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\ndrm_render_minor {128 + i}\n")
        if cpus is not None:
            d = root / "class" / "drm" / f"renderD{128 + i}" / "device"
            d.mkdir(parents=True)
            (d / "local_cpulist").write_text(cpus + "\n")
    assert D.gpu_cpu_lists(str(root)) == [[0, 1, 2, 3], [0, 1, 2, 3], [4, 5, 6, 7], [4, 5, 6, 7]]
    monkeypatch.delenv("STEM_PIN_RANKS", raising=False)
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(D.os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    got = [D.pin_rank_to_gpu_cores(r, 4, str(root), apply=False) for r in range(4)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert D.pin_rank_to_gpu_cores(2, 1, str(root), apply=False) == [4, 5, 6, 7]      # alone on its node: all of it
    assert D.pin_rank_to_gpu_cores(5, 8, str(root), apply=False) is None              # no such device
    assert D.pin_rank_to_gpu_cores(0, 4, str(tmp_path / "nothing"), apply=False) is None
    monkeypatch.setenv("STEM_PIN_RANKS", "0")
    assert D.pin_rank_to_gpu_cores(0, 4, str(root), apply=False) is None
    monkeypatch.delenv("STEM_PIN_RANKS", raising=False)
    # a launcher that re-maps devices: HIP device r is NOT KFD GPU r any more.  unique ids for the UUID form
    for i, uid in ((2, 0x1111), (3, 0x2222), (4, 0xABCD), (5, 0xF00D)):
        with open(nodes / str(i) / "properties", "a") as f:
            f.write(f"unique_id {uid}\n")
    pin = lambda r, w, env: D.pin_rank_to_gpu_cores(r, w, str(root), apply=False, environ=env)
    assert D.visible_device_map(4, {}) == [0, 1, 2, 3]
    assert D.visible_device_map(4, {"HIP_VISIBLE_DEVICES": "3,1"}) == [3, 1]
    assert D.visible_device_map(4, {"CUDA_VISIBLE_DEVICES": "2"}) == [2]
    assert D.visible_device_map(4, {"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "2,0"}) == [3, 1]      # HIP indexes what ROCr exposes
    assert D.visible_device_map(4, {"HIP_VISIBLE_DEVICES": "1,7,2"}) == [1]                                       # stops at the first invalid entry
    assert D.visible_device_map(4, {"ROCR_VISIBLE_DEVICES": "GPU-abcd,GPU-1111"}, ["1111", "2222", "abcd", "f00d"]) == [2, 0]
    assert pin(0, 2, {"HIP_VISIBLE_DEVICES": "3,1"}) == [4, 5, 6, 7] and pin(1, 2, {"HIP_VISIBLE_DEVICES": "3,1"}) == [0, 1, 2, 3]
    assert pin(0, 2, {"HIP_VISIBLE_DEVICES": "2,3"}) == [4, 5] and pin(1, 2, {"HIP_VISIBLE_DEVICES": "2,3"}) == [6, 7]       # both on node 1: split
    assert pin(0, 1, {"ROCR_VISIBLE_DEVICES": "GPU-f00d"}) == [4, 5, 6, 7]
    assert pin(1, 2, {"HIP_VISIBLE_DEVICES": "2"}) is None                                                        # one visible device, rank 1 has none
    # applying really narrows the mask (to cores this process is allowed to use)
    monkeypatch.undo()
    allowed = sorted(os.sched_getaffinity(0))
    d = root / "class" / "drm" / "renderD130" / "device"
    (d / "local_cpulist").write_text(f"{allowed[0]}\n")
    try:
        # (environ={}: this container exports an EMPTY HIP_VISIBLE_DEVICES -- no device visible, which the map honours too)
        assert D.pin_rank_to_gpu_cores(0, 1, str(root), environ={"HIP_VISIBLE_DEVICES": ""}) is None
        assert D.pin_rank_to_gpu_cores(0, 1, str(root), environ={}) == [allowed[0]] and sorted(os.sched_getaffinity(0)) == [allowed[0]]
    finally:
        os.sched_setaffinity(0, allowed)


# ---- the native issue path's construction protocol (distributed._NativeIssuer) with a stand-in for libstem_dp.so -------------------
class _FakeDp:
    """what _NativeIssuer calls on libstem_dp.so, recording the calls; `fail` names the entry point that fails on this rank"""

    def __init__(self, fail=None, nranks=2):
        self.fail, self.n, self.calls, self.status = fail, nranks, [], 0

    def _rc(self, name):
        self.calls.append(name)
        return -2 if self.fail == name else 0

    def stem_dp_prepare(self, href, device):
        rc = self._rc("prepare")
        if rc == 0:
            import ctypes
            ctypes.cast(href, ctypes.POINTER(ctypes.c_void_p))[0] = 0x1000
        return rc

    def stem_dp_unique_id(self, ident):
        ident[0] = 42
        return self._rc("unique_id")

    def stem_dp_connect(self, h, ident, world, rank):
        self.ident0 = ident[0]
        return self._rc("connect")

    def stem_dp_nranks(self, h):
        self.calls.append("nranks")
        return self.n

    def stem_dp_status(self, h):
        return self.status

    def stem_dp_destroy(self, h):
        self.calls.append("destroy")
        return 0

    def stem_dp_last_error(self):
        return f"fake: {self.fail} failed".encode()


def _worker_native_protocol(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    torch.set_num_threads(1)
    from spatiotemporalentropymodel_amd import distributed as D
    D.init_from_env(backend="gloo")
    out = {}

    def attempt(fake):
        try:
            iss = D._NativeIssuer("cpu", lib=fake)
            return "ok", iss
        except D.NativeRouteUnavailable as e:
            return "unavailable", str(e)
        except RuntimeError as e:
            return "error", str(e)

    # (a) rank 1 cannot prepare: BOTH ranks learn it before anybody connects; the prepared rank's handle is destroyed
    fake = _FakeDp(fail="prepare" if rank == 1 else None)
    out["a"] = (attempt(fake)[0], list(fake.calls))
    # (b) rank 0 cannot draw the id
    fake = _FakeDp(fail="unique_id" if rank == 0 else None)
    out["b"] = (attempt(fake)[0], list(fake.calls))
    # (c) connect fails on rank 0 only: past the agreement there is no fallback -- RuntimeError on both
    fake = _FakeDp(fail="connect" if rank == 0 else None)
    kind, msg = attempt(fake)
    out["c"] = (kind, list(fake.calls))
    # (d) RCCL reports another rank count than the process group's
    fake = _FakeDp(nranks=2 if rank == 0 else 1)
    out["d"] = (attempt(fake)[0], list(fake.calls))
    # (e) everything fine: the id drawn on rank 0 reached rank 1, nranks is what the communicator says
    fake = _FakeDp()
    kind, iss = attempt(fake)
    out["e"] = (kind, list(fake.calls), fake.ident0, iss.nranks if kind == "ok" else None)
    if kind == "ok":
        iss.check()
        fake.status = -3
        fake.fail = "collective"
        try:
            iss.check()
            out["check"] = "silent"
        except RuntimeError as e:
            out["check"] = str(e)
        iss.h = None                                   # nothing real to destroy at exit
    # replica checksums: equal tensors agree, one flipped low bit is seen by every rank
    t = torch.arange(10000, dtype=torch.float32) * 0.37
    out["same"] = D.replicas_identical(t)[0]
    if rank == 1:
        t.view(torch.int32)[7777] ^= 1
    out["diff"] = D.replicas_identical(t)[0]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


@pytest.mark.timeout(300)
def test_native_issuer_construction_is_agreed_by_all_ranks_gloo():
    """ADVICE r5 (distributed.py:297): a rank-local failure while the native RCCL path is set up must never leave the peers inside
    ncclCommInitRank or reducing on different communicators.  Two gloo ranks, libstem_dp.so replaced by a recording stand-in."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_native_protocol, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["a"][0] == "unavailable" and "connect" not in res[r]["a"][1]
        assert res[r]["b"][0] == "unavailable" and "connect" not in res[r]["b"][1]
        assert res[r]["c"][0] == "error" and "connect" in res[r]["c"][1] and "destroy" in res[r]["c"][1]
        assert res[r]["d"][0] == "error"
        assert res[r]["e"][0] == "ok" and res[r]["e"][2] == 42 and res[r]["e"][3] == 2
        assert "collective failed" in res[r]["check"]
        assert res[r]["same"] is True and res[r]["diff"] is False
    assert "destroy" in res[0]["a"][1] and "prepare" in res[1]["a"][1]          # rank 0 had prepared fine: its handle is released
