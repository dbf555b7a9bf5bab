#!/usr/bin/env python3
"""stem_adam_step_zero against stem_adam_step_bmax on the bench model's 18 M parameters, alone on the chip (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

n = 18_070_000
dev = torch.device("cuda:0")
p, g, m, v = (torch.randn(n, device=dev) * s for s in (0.1, 1.0, 0.0, 0.0))
bm = torch.empty(4 * ((n + F.adam_chunk() - 1) // F.adam_chunk()), device=dev)


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


t0 = timeit(lambda: F.adam_step(p, g, m, v, None, 0.0, 1.0, 1e-4, 0.9, 0.999, 1e-8, 5, zero_grad=True))
t1 = timeit(lambda: F.adam_step_bmax(p, g, m, v, None, 0.0, 1.0, 1e-4, 0.9, 0.999, 1e-8, 5, bm, zero_grad=True))
print(f"adam_step_zero {t0:.1f} us ({n * 32 / t0 / 1e6:.2f} TB/s)   adam_step_bmax {t1:.1f} us ({n * 32 / t1 / 1e6:.2f} TB/s)")
