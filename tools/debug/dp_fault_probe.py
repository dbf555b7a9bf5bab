#!/usr/bin/env python3
"""Where does libstem_dp's failure path spend its time?  Stage-by-stage log of the fault-injection sequence of
tests/test_hip_dp_native.py::test_helper_failure_is_abort_all_and_nothing_is_left_waiting (run under `timeout`)."""
import ctypes as C
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(60, exit=True)
import torch  # noqa: E402
from spatiotemporalentropymodel_amd import _lib  # noqa: E402


def log(*a):
    print(f"[{time.time() - T0:7.3f}]", *a, file=sys.stderr, flush=True)


T0 = time.time()
lib = _lib.dp()
torch.cuda.set_device(0)
os.environ["STEM_DP_FAULT"] = sys.argv[1] if len(sys.argv) > 1 else "1"
ident = (C.c_ubyte * 128)()
assert lib.stem_dp_unique_id(ident) == 0
h = C.c_void_p()
assert lib.stem_dp_prepare(C.byref(h), 0) == 0
log("prepared")
assert lib.stem_dp_connect(h, ident, 1, 0) == 0
log("connected, nranks", lib.stem_dp_nranks(h))
cons = torch.cuda.Stream()
buf = torch.ones(1 << 20, device="cuda:0")
torch.cuda.synchronize()
st = (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
log("submit 0 ->", lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()))
log("fence 0 ->", lib.stem_dp_fence(h, cons.cuda_stream))
cons.synchronize()
log("consumer synchronised; status", lib.stem_dp_status(h))
log("submit 1 ->", lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()))
log("fence 1 ->", lib.stem_dp_fence(h, cons.cuda_stream))
t = time.time()
while lib.stem_dp_status(h) == 0 and time.time() - t < 10:
    time.sleep(0.01)
log("status", lib.stem_dp_status(h), lib.stem_dp_last_error())
cons.synchronize()
log("consumer synchronised after the failure")
log("nranks", lib.stem_dp_nranks(h), "submit", lib.stem_dp_submit(h, st, 1, buf.data_ptr(), buf.numel()), "fence", lib.stem_dp_fence(h, cons.cuda_stream))
log("destroy ->", lib.stem_dp_destroy(h))
log("done")
