#!/usr/bin/env python3
"""g_a.0 + GDN at the bench shape (B=16, 256x256): the fp16 kernel of csrc/c4gdn_f16x3.hip next to the fp32-MFMA kernel of
igemm.hip (STEM_C4GDN_F16X3=0), isolated launches timed with HIP events; max difference of the two results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

B, K = int(os.environ.get("B", 16)), 192
ABLATE = int(sys.argv[1]) if len(sys.argv) > 1 else 0            # experiments build only: 1 = 5 instead of 8 conv k-steps, 2 = wait + barrier at every second step, 3 = both
if ABLATE:
    import ctypes
    from spatiotemporalentropymodel_amd import _lib
    fn_ = _lib.hip().stem_exper_c4gdn_ablate
    fn_.argtypes, fn_.restype = [ctypes.c_int], None
    fn_(ABLATE)
    print(f"ABLATION {ABLATE} (results are wrong, timing only)")
torch.manual_seed(0)
x = torch.rand(B, 3, 256, 256, device="cuda")
w = torch.randn(K, 3, 5, 5, device="cuda") * 0.1
b = torch.randn(K, device="cuda") * 0.1
beta = torch.rand(K, device="cuda") + 0.5
gamma = torch.rand(K, K, device="cuda") * 0.1 + 0.1 * torch.eye(K, device="cuda")
x4 = F.nchw3_to_nhwc4(x)
wp = F.pack_weight(w, F.PACK_CONV_FWD_C4)
ast = F.c4gdn_stream(wp, gamma, K, 5, 5)
flop = (2 * 192 * 3 * 25 + 2 * 192 * 192) * 128 * 128 * B
res = {}
for route in (("1",) if ABLATE else ("1", "0")):
    os.environ["STEM_C4GDN_F16X3"] = route
    for planes in (True, False):
        fn = (lambda: F.conv2d_fwd_c4_gdn_planes(x4, wp, b, beta, gamma, K, 5, 5, 2, 2, astream=ast)) if planes else \
             (lambda: F.conv2d_fwd_c4_gdn(x4, wp, b, beta, gamma, K, 5, 5, 2, 2, astream=ast))
        for _ in range(3):
            y = fn()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[(route, planes)] = y.merge() if planes else y
        print(f"route {'f16x3 ' if route == '1' else 'fp32-mfma'} {'planes' if planes else 'fp32  '}: median {ts[5] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us  "
              f"{flop / ts[5] / 1e9:6.1f} TFLOP/s algorithmic")
if not ABLATE:
    d = (res[("1", True)] - res[("0", True)]).abs().max() / res[("0", True)].abs().max()
    print(f"max |f16x3 - fp32-mfma| / max = {float(d):.2e}")
