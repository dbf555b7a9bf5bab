import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dp2: two data-parallel ranks sharing cuda:0 (workers start at collection time)")
    config.addinivalue_line("markers", "bench_gpus2: `python bench.py --gpus 2` launched bare at collection time")


# ---- 2-rank data-parallel GPU tests ---------------------------------------------------------------------------------
# The rank processes are started HERE, right after collection and before any test has run, i.e. while this pytest
# process has not initialised the GPU yet (a GPU-initialised process must not fork+exec children on the GPU pool).
# They run concurrently with the first tests and write .npz dumps that tests/test_hip_dp2.py waits for and checks.
DP2 = {"procs": {}, "dir": None}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_collection_finish(session):
    cases = sorted({m.args[0] for item in session.items for m in item.iter_markers("dp2") if m.args})
    want_bench = any(item.get_closest_marker("bench_gpus2") for item in session.items)
    if (not cases and not want_bench) or session.config.option.collectonly:
        return
    import subprocess
    import tempfile
    DP2["dir"] = tempfile.mkdtemp(prefix="stem_dp2_")
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    if torch.cuda.device_count() < 2:                    # counting devices does not initialise the GPU
        cases = [c for c in cases if not c.startswith("rccl2_")]      # two real RCCL ranks need a device each: those tests skip
    for case in cases:                                   # one pair at a time would serialise; pairs are small, run them all
        port = _free_port()
        world = 1 if case.startswith("rccl1_") else 2    # rccl1_*: ONE rank in a world-size-1 RCCL ("nccl") process group
        DP2["procs"][case] = [
            subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), "--case", case, "--rank", str(r),
                              "--world", str(world), "--port", str(port), "--out", DP2["dir"]], env=env,
                             stdout=open(os.path.join(DP2["dir"], f"{case}_rank{r}.log"), "w"), stderr=subprocess.STDOUT)
            for r in range(world)]


    # `python bench.py --gpus 2` typed bare on the GPU box (the command form the driver uses): started here for the same reason
    if any(item.get_closest_marker("bench_gpus2") for item in session.items):
        env_b = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        env_b["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        env_b["STEM_BENCH_VERIFY"] = "1"
        DP2["bench2"] = subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                          "--no-cpu-baseline"], env=env_b, stdout=open(os.path.join(DP2["dir"], "bench2.out"), "w"),
                                         stderr=open(os.path.join(DP2["dir"], "bench2.err"), "w"))


def pytest_sessionfinish(session, exitstatus):
    for procs in DP2["procs"].values():
        for p in procs:
            if p.poll() is None:
                p.kill()
    b = DP2.get("bench2")
    if b is not None and b.poll() is None:
        b.kill()


@pytest.fixture(scope="session")
def dp2_results():
    """wait(case) -> [rank0 dump, rank1 dump] (dicts of numpy arrays); fails with the worker logs if a rank died."""
    def wait(case, timeout=900):
        procs = DP2["procs"].get(case)
        assert procs, f"dp2 workers for {case!r} were not started (collection hook)"
        for r, p in enumerate(procs):
            try:
                rc = p.wait(timeout=timeout)
            except Exception:
                p.kill()
                rc = "timeout"
            if rc != 0:
                log = open(os.path.join(DP2["dir"], f"{case}_rank{r}.log")).read()[-4000:]
                pytest.fail(f"dp2 worker {case} rank {r} exited with {rc}:\n{log}")
        return [dict(np.load(os.path.join(DP2["dir"], f"{case}_rank{r}.npz"))) for r in range(len(procs))]
    return wait


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


def close_ratio(a, b, floor, atol=0.0):
    """max over elements of (|a - b| - atol) / max(|b|, floor * max|b|): the quantity assert_close bounds by rtol."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float((np.maximum(np.abs(a - b) - atol, 0.0) / np.maximum(np.abs(b), floor * scale)).max())


def assert_close(a, b, rtol=1e-4, atol=0.0, what="", *, floor):
    """north_star tolerance: 1e-4 relative (fp32).  `atol` is stated per call site where the quantity has a natural
    absolute floor (e.g. likelihoods are floored at 1e-9).  `floor` is stated per call site too: fp32 summation error is
    relative to the magnitude of the summed terms, not of a cancelling result, so elements smaller than `floor` x max|b|
    are held to the absolute bound rtol x floor x max|b| (floor=0.1 -> 1e-5 of the tensor's maximum, about ten times the
    fp32 noise of the dot products on this path; floor=0 is a pure element-wise relative test)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    tol = atol + rtol * np.maximum(np.abs(b), floor * scale)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()}/{bad.size} out of tolerance; max err {err.max():.3e} "
                           f"(scale {scale:.3e}) at {np.unravel_index(err.argmax(), err.shape)}")


def f64_gate(ours, exact, ref32_err, what, rtol=1e-4, atol=0.0, floor=0.1):
    """Parity gate against the float64 run of the reference (tests/golden/stem_f64.npz, make_golden.py:gen_f64): the HIP
    result must be within `rtol` (north_star: 1e-4) of the EXACT value; the reference's own fp32 error in the same
    metric is printed next to ours (it is the yardstick: e.g. torch's fp32 clip_grad_norm_ is ~1e-4 off the exact norm,
    so two correct fp32 implementations can legitimately sit 2e-4 apart)."""
    r = close_ratio(ours, exact, floor, atol)
    print(f"[f64 gate] {what}: HIP vs exact {r:.2e}   reference-fp32 vs exact {float(np.max(ref32_err)):.2e}   bound {rtol:.0e}")
    assert r <= rtol, f"{what}: HIP path is {r:.3e} from the float64 reference (bound {rtol:.0e}; the reference's own fp32 run: {float(np.max(ref32_err)):.3e})"
    return r
