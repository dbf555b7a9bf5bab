"""GPU: libstem_dp.so (include/stem_dp.h) on its own -- a one-rank RCCL communicator (RCCL refuses two ranks on one device), the
helper thread's ordering guarantees checked directly through the C ABI."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from spatiotemporalentropymodel_amd import _lib
    return _lib.dp()


def test_native_exchange_orders_producers_collective_and_consumer():
    """stem_dp_submit: the all-reduce starts only after everything enqueued on the producer streams has finished;
    stem_dp_fence: the consumer stream continues only after the collective.  A producer stream fills the buffer at the END of a long
    queue of work; the consumer copies the buffer right after the fence.  At one rank the sum is the identity: the copy must hold
    the producer's final values -- for several rounds (the flag counts fences), with one and with two producer streams, and an
    empty exchange in between."""
    lib = _lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ident = (C.c_ubyte * 128)()
    assert lib.stem_dp_unique_id(ident) == 0
    h = C.c_void_p()
    assert lib.stem_dp_create(C.byref(h), ident, 1, 0, 0) == 0, lib.stem_dp_last_error()
    try:
        p1, p2, cons = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        n = 1 << 22
        buf = torch.zeros(2 * n, device=dev)
        big = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        for rnd in range(1, 6):
            with torch.cuda.stream(p1):
                x = big
                for _ in range(12):                      # ~ms of queued work in front of the final write
                    x = x @ big
                    x = x / x.abs().max().clamp_min(1.0)
                buf[:n].fill_(float(rnd))
            with torch.cuda.stream(p2):
                y = big
                for _ in range(6):
                    y = y @ big
                    y = y / y.abs().max().clamp_min(1.0)
                buf[n:].fill_(float(-rnd))
            streams = (C.c_void_p * 2)(p1.cuda_stream, p2.cuda_stream)
            if rnd % 2:
                assert lib.stem_dp_submit(h, streams, 2, buf.data_ptr(), 2 * n) == 0, lib.stem_dp_last_error()
            else:                                        # one slice per producer stream
                assert lib.stem_dp_submit(h, (C.c_void_p * 1)(p1.cuda_stream), 1, buf.data_ptr(), n) == 0
                assert lib.stem_dp_submit(h, (C.c_void_p * 1)(p2.cuda_stream), 1, buf.data_ptr() + 4 * n, n) == 0
            assert lib.stem_dp_submit(h, streams, 2, buf.data_ptr(), 0) == 0          # nothing to exchange: accepted, no-op
            assert lib.stem_dp_fence(h, cons.cuda_stream) == 0, lib.stem_dp_last_error()
            with torch.cuda.stream(cons):
                snap = buf.clone()
            cons.synchronize()
            assert lib.stem_dp_status(h) == 0
            assert float(snap[:n].min()) == float(snap[:n].max()) == float(rnd), rnd
            assert float(snap[n:].min()) == float(snap[n:].max()) == float(-rnd), rnd
            torch.cuda.synchronize()
    finally:
        assert lib.stem_dp_destroy(h) == 0


def test_native_exchange_rejects_bad_arguments():
    lib = _lib()
    assert lib.stem_dp_fence(None, None) != 0 and b"null handle" in lib.stem_dp_last_error()
    ident = (C.c_ubyte * 128)()
    h = C.c_void_p()
    assert lib.stem_dp_create(C.byref(h), ident, 2, 5, 0) != 0 and b"bad arguments" in lib.stem_dp_last_error()
    assert lib.stem_dp_connect(None, ident, 1, 0) != 0 and lib.stem_dp_nranks(None) < 0 and lib.stem_dp_abort(None, -1, None) != 0
    assert lib.stem_dp_destroy(None) == 0


def _prepared(lib, connect=True):
    ident = (C.c_ubyte * 128)()
    assert lib.stem_dp_unique_id(ident) == 0
    h = C.c_void_p()
    assert lib.stem_dp_prepare(C.byref(h), 0) == 0, lib.stem_dp_last_error()
    assert h.value
    if connect:
        assert lib.stem_dp_connect(h, ident, 1, 0) == 0, lib.stem_dp_last_error()
    return h, ident


def test_prepare_is_local_and_connect_reports_the_communicators_rank_count():
    """The two construction steps (include/stem_dp.h): stem_dp_prepare touches no communicator (nranks / submit / fence refuse the
    handle), stem_dp_connect builds it and stem_dp_nranks reads the rank count back from RCCL (ncclCommCount), a second connect
    is refused."""
    lib = _lib()
    torch.cuda.set_device(0)
    h, ident = _prepared(lib, connect=False)
    try:
        assert lib.stem_dp_nranks(h) < 0 and b"not connected" in lib.stem_dp_last_error()
        buf = torch.zeros(16, device="cuda:0")
        assert lib.stem_dp_submit(h, None, 0, buf.data_ptr(), 16) != 0 and b"not connected" in lib.stem_dp_last_error()
        assert lib.stem_dp_fence(h, None) != 0
        assert lib.stem_dp_connect(h, ident, 1, 0) == 0, lib.stem_dp_last_error()
        assert lib.stem_dp_nranks(h) == 1
        assert lib.stem_dp_connect(h, ident, 1, 0) != 0 and b"connected already" in lib.stem_dp_last_error()
        assert lib.stem_dp_status(h) == 0
    finally:
        assert lib.stem_dp_destroy(h) == 0


def test_helper_failure_is_abort_all_and_nothing_is_left_waiting(monkeypatch):
    """A collective the helper thread cannot issue (fault injected at exchange 1 by STEM_DP_FAULT, read at prepare): the status
    carries the first failure and its message, the communicator is gone (stem_dp_nranks refuses), a fence that was already queued
    is released from the host -- the consumer stream finishes instead of waiting for a flag nobody writes -- and every later
    submit / fence returns the status, so that the rank leaves with an error instead of stepping on unreduced gradients."""
    lib = _lib()
    torch.cuda.set_device(0)
    monkeypatch.setenv("STEM_DP_FAULT", "1")
    h, _ = _prepared(lib)
    monkeypatch.delenv("STEM_DP_FAULT")
    try:
        cons = torch.cuda.Stream()
        buf = torch.ones(1 << 20, device="cuda:0")
        torch.cuda.synchronize()
        st = (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
        assert lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()) == 0            # exchange 0: fine
        assert lib.stem_dp_fence(h, cons.cuda_stream) == 0
        cons.synchronize()
        assert lib.stem_dp_status(h) == 0
        assert lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()) == 0            # exchange 1: refused inside the helper
        rc_fence = lib.stem_dp_fence(h, cons.cuda_stream)                                 # queued before or after the helper noticed
        with torch.cuda.stream(cons):
            snap = buf.clone()
        cons.synchronize()                                                                # must return either way
        import time
        t0 = time.time()
        while lib.stem_dp_status(h) == 0 and time.time() - t0 < 10:
            time.sleep(0.01)
        assert lib.stem_dp_status(h) == -3 and b"injected fault" in lib.stem_dp_last_error()
        assert rc_fence in (0, -3)
        assert lib.stem_dp_nranks(h) < 0
        assert lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()) == -3 and b"failed earlier" in lib.stem_dp_last_error()
        assert lib.stem_dp_fence(h, cons.cuda_stream) == -3
        assert float(snap.sum()) == float(buf.numel())
    finally:
        assert lib.stem_dp_destroy(h) == 0


def test_host_abort_has_the_same_effect():
    lib = _lib()
    torch.cuda.set_device(0)
    h, _ = _prepared(lib)
    try:
        assert lib.stem_dp_abort(h, -7, b"rank 3 lost its loader") == 0
        assert lib.stem_dp_status(h) == -7 and b"rank 3 lost its loader" in lib.stem_dp_last_error()
        assert lib.stem_dp_abort(h, -8, b"second") == 0 and lib.stem_dp_status(h) == -7          # the first failure stands
        assert lib.stem_dp_fence(h, torch.cuda.current_stream().cuda_stream) == -7
        torch.cuda.synchronize()
    finally:
        assert lib.stem_dp_destroy(h) == 0


def test_reducer_raises_after_a_failed_exchange(monkeypatch):
    """through the Python binding: distributed._NativeIssuer.check() (called by OverlappedGradReducer.finish() after the step's
    fence, and by bench.py after its synchronise) turns the status into an exception"""
    from spatiotemporalentropymodel_amd import distributed as D
    lib = _lib()
    torch.cuda.set_device(0)
    h, _ = _prepared(lib)
    iss = D._NativeIssuer.__new__(D._NativeIssuer)
    iss.lib, iss.h = lib, h
    try:
        iss.check()
        iss.abort("peer died")
        with pytest.raises(RuntimeError, match="peer died"):
            iss.check()
        with pytest.raises(RuntimeError, match="failed earlier"):
            iss.fence()
    finally:
        iss.close()
