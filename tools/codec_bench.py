#!/usr/bin/env python3
"""configs[3]-shaped codec timing: one 1080p P-frame latent (y [1,192,68,120]) through compress / decompress."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_  # noqa: E402

m = SpatioTemporalPriorModel_Res()
closed_form_fill_(m)
m = m.cuda().eval()
m.update(force=True)
g = torch.Generator(device="cuda")
g.manual_seed(3)
y_cond = torch.randn(1, 192, 68, 120, device="cuda", generator=g) * 3
y_cur = y_cond + torch.randn(1, 192, 68, 120, device="cuda", generator=g) * 2
with torch.no_grad():
    enc = m.compress(y_cur, y_cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = m.compress(y_cur, y_cond)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec = m.decompress(enc["strings"], enc["shape"], y_cond)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
nbytes = len(enc["strings"][0][0]) + len(enc["strings"][1][0])
print("1080p P-frame latent: compress %.3f s, decompress %.3f s, %d bytes (%.3f bpp)" % (t1 - t0, t2 - t1, nbytes, 8 * nbytes / (1088 * 1920)))
print("reference (survey container, torch CPU): 13 s + 39 s per frame (SURVEY.md 3.2)")
