#!/usr/bin/env python3
"""Where a 1080p P-frame compress() spends its time: hyper path, wavefront loop on the GPU, copies, host rANS."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import codec, functional as F  # noqa: E402
from spatiotemporalentropymodel_amd.entropy_models import BufferedRansEncoder  # noqa: E402
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input  # noqa: E402

dev = torch.device("cuda:0")
m = closed_form_fill_(SpatioTemporalPriorModel_Res()).to(dev).eval()
m.update(force=True)
H, W, M = 68, 120, 192
y_cur = closed_form_input("es:y", (1, M, H, W), -6, 6).to(dev)
y_cond = closed_form_input("es:c", (1, M, H, W), -6, 6).to(dev)


def t():
    torch.cuda.synchronize()
    return time.perf_counter()


with torch.no_grad():
    for rep in range(3):
        t0 = t()
        z_strings, zshape, hp, tp = codec._hyper(m, y_cur, y_cond)
        t1 = t()
        yc, yd = F.to_nhwc(y_cur), F.to_nhwc(y_cond)
        target = F.sub(codec._dense(yc), codec._dense(yd))
        ar = codec._ARContext(m, dev)
        tables = m.gaussian_conditional.host_tables()
        buf = codec._padded(target, H, W, M, dev)
        sym = torch.empty((H * W, M), device=dev, dtype=torch.int32)
        idx = torch.empty((H * W, M), device=dev, dtype=torch.int32)
        t2 = t()
        ar.encode_wavefront(buf, H, W, tp.data_ptr(), hp.data_ptr(), sym, idx)
        t3 = t()
        s_h, i_h = sym.cpu().numpy(), idx.cpu().numpy()
        t4 = t()
        enc = BufferedRansEncoder()
        enc.encode_with_indexes(s_h, i_h, tables)
        out = enc.flush()
        t5 = t()
        print(f"hyper path {1e3 * (t1 - t0):.1f} ms, set-up {1e3 * (t2 - t1):.1f}, wavefront loop ({W + 3 * (H - 1)} steps) {1e3 * (t3 - t2):.1f}, "
              f"copies to the host {1e3 * (t4 - t3):.1f}, host rANS {1e3 * (t5 - t4):.1f} ms ({len(out)} bytes)")
