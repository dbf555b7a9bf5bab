#!/usr/bin/env python3
"""Soak: 240 random shapes (tools/debug/f16x3_fuzz.py, two seeds) through the split-operand kernels under the library's own plan
and under every forced main-loop form (pixel tile x LDS stages x MFMA shape); fails above 1e-5 of max|ref| against fp64 (dev tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "debug"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import f16x3_fuzz
from spatiotemporalentropymodel_amd import functional as F
worst = {}
for name, plan in (("default", {}), ("t128-3-16", dict(fx3_tile=128, fx3_depth=3, fx3_mfma=16, fx3_gen_tile=128, fx3_gen_mfma=16)),
                   ("t64-3-16", dict(fx3_tile=64, fx3_depth=3, fx3_mfma=16, fx3_gen_tile=64, fx3_gen_mfma=16)), ("t128-3-32", dict(fx3_tile=128, fx3_depth=3, fx3_mfma=32)),
                   ("t64-3-32", dict(fx3_tile=64, fx3_depth=3, fx3_mfma=32)), ("t128-2", dict(fx3_tile=128, fx3_depth=2))):
    with F.tuning(**plan):
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            w = max(f16x3_fuzz.run(120, seed, verbose=False) for seed in (21, 22))
        flagged = [l for l in buf.getvalue().splitlines() if "!!" in l]
    worst[name] = w
    print(name, f"worst {w:.2e}", "flagged:", flagged[:3], flush=True)
assert max(worst.values()) <= 1e-5, worst
print("SOAK OK")
