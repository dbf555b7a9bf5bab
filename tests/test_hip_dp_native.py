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
    assert lib.stem_dp_destroy(None) == 0
