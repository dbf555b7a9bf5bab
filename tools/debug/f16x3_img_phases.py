#!/usr/bin/env python3
"""Where a workgroup of the image-tile kernel spends its time: phase stamps (s_memrealtime, 100 MHz) of every workgroup of one
launch.  Needs the experiments build: `make -C spatiotemporalentropymodel_amd/csrc experiments`, then
    STEM_HIP_LIBRARY=spatiotemporalentropymodel_amd/libstem_hip_exper.so python3 tools/debug/f16x3_img_phases.py [layer] [split] [ablate]
ablate = 1: every second chunk's barrier left out (wrong results; the upper bound of what one barrier per two chunks could buy);
ablate = 2: the workgroups return right after the main loop (what everything behind it costs at most);
ablate = 3: one-dimensional launch with the splits of a tile on ONE XCD, partial tiles stored / read without sc1 (they stay in the
XCD's L2; results not guaranteed); the launch's duration by HIP events and the difference from the normal mode's result are printed
in every mode.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import _lib, functional as F  # noqa: E402

LAYERS = {"TPM.0": (192, 256, 5), "TPM.2": (256, 320, 5), "TPM.4": (320, 384, 5), "HE.0": (384, 256, 3), "EPM.0": (1152, 768, 1), "EPM.4": (576, 384, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "TPM.4"
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ablate = int(sys.argv[3]) if len(sys.argv) > 3 else 0
C_, K, R = LAYERS[name]
dev = torch.device("cuda:0")
x = torch.randn(16, C_, 16, 16, device=dev)
w = torch.randn(K, C_, R, R, device=dev) / (C_ * R * R) ** 0.5
b = torch.randn(K, device=dev) * 0.1
xp, wp = F.F16Planes.split(x), F.pack_weight_f16x2_gen(w)
lib = _lib.hip()
lib.stem_exper_img_stamps.argtypes = [C.c_void_p]
lib.stem_exper_img_stamps.restype = None
with F.tuning(**dict(fx3_gen_img=2, **({"fx3_split": split} if split else {}))):
    ref_out = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)[0].clone()
if ablate:
    lib.stem_exper_img_ablate.argtypes = [C.c_int]
    lib.stem_exper_img_ablate.restype = None
    lib.stem_exper_img_ablate(ablate)
    print(f"ABLATION {ablate} (results are wrong, timing only)")
NW = 4096
stamps = torch.zeros(NW * 8, dtype=torch.int64, device=dev)
tune = dict(fx3_gen_img=2)
if split:
    tune["fx3_split"] = split
with F.tuning(**tune):
    for _ in range(5):
        F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    torch.cuda.synchronize()
    lib.stem_exper_img_stamps(stamps.data_ptr())
    lib.stem_exper_img_waits.argtypes = [C.c_void_p, C.c_int]
    lib.stem_exper_img_waits.restype = None
    lib.stem_exper_img_waits(None, 1)
    F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
    torch.cuda.synchronize()
    lib.stem_exper_img_stamps(None)
    wbuf = (C.c_uint64 * 4)()
    lib.stem_exper_img_waits(wbuf, 0)
    if wbuf[2]:
        print(f"  wavefront cycles in the main loop: {100.0 * wbuf[0] / wbuf[2]:.1f} % waiting for the weight DMA (vmcnt), {100.0 * wbuf[1] / wbuf[2]:.1f} % at the barrier")
with F.tuning(**tune):                # the launch by HIP events, without the stamps
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name}: launch by HIP events: median {ts[10]:.1f} us, min {ts[0]:.1f} us")
    out = F.conv2d_f16x3_gen(xp, wp, b, K, R, R, 1, R // 2, epi=F.GEN_EPI_LRELU, slope=0.01, want_planes=True)[0]
    print(f"{name}: max |this mode - normal mode| / max = {float((out - ref_out).abs().max() / ref_out.abs().max()):.2e}")
s = stamps.view(NW, 8).cpu()
s = s[s[:, 0] > 0].double() / 100.0          # microseconds
t0 = float(s[:, 0].min())
print(f"{name}: {s.shape[0]} workgroups; first start 0, last start {float(s[:, 0].max()) - t0:.1f} us, last end {float(s[:, 1:7].max()) - t0:.1f} us")
names = ["start -> loads issued", "scales (records, bias)", "main loop", "partials + ticket", "last arriver: slab read", "epilogue"]
last = s[:, 6] > 0
for i, nm in enumerate(names):
    a, bb = s[:, i], s[:, i + 1]
    if i == 4:
        ok = last & (s[:, 5] > 0)
    elif i == 5:
        ok = last
        a = torch.where(s[:, 5] > 0, s[:, 5], s[:, 3])
    elif i == 3:
        ok = s[:, 4] > 0
    else:
        ok = bb > 0
    if ok.any():
        d = (bb - a)[ok]
        print(f"  {nm:28s} n={int(ok.sum()):4d} median {float(d.median()):7.2f} us  max {float(d.max()):7.2f}  min {float(d.min()):7.2f}")
print(f"  workgroup lifetime: median {float((s[:, 1:7].max(dim=1).values - s[:, 0]).median()):.1f} us")
