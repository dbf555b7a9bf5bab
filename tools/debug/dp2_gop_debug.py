"""debug: spawn the two gop workers, then compare with the single-process B=2 run per tensor (ratio statistics)."""
import os, subprocess, sys, tempfile, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
out = tempfile.mkdtemp()
clip = sys.argv[1] if len(sys.argv) > 1 else "1.0"
env = dict(os.environ, PYTHONPATH=REPO, DP2_CLIP=clip)
procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), "--case", "gop", "--rank", str(r), "--world", "2",
                           "--port", "29533", "--out", out], env=env) for r in range(2)]
assert all(p.wait() == 0 for p in procs)
import numpy as np, torch
from dp_worker import SlicedNoise
from spatiotemporalentropymodel_amd import selfcheck as S
from spatiotemporalentropymodel_amd.losses import PixelwiseRateDistortionLoss
from spatiotemporalentropymodel_amd.models import stem_roi, stem_roi_i
from spatiotemporalentropymodel_amd.optim import configure_optimizers
from spatiotemporalentropymodel_amd.weights import closed_form_fill_scaled_, closed_form_input, smooth_frames
r0 = dict(np.load(os.path.join(out, "gop_rank0.npz")))
dev = torch.device("cuda:0")
imodel = closed_form_fill_scaled_(stem_roi_i(), "stem_roi_i", 0.7).to(dev).train()
pmodel = closed_form_fill_scaled_(stem_roi(), "stem_roi", 0.7).to(dev).train()
for m, tag in ((imodel, "i"), (pmodel, "p")):
    m.entropy_bottleneck.noise_source = SlicedNoise(f"roi_{tag}_eb", 0, 1, 2, batch_last=True)
    m.gaussian_conditional.noise_source = SlicedNoise(f"roi_{tag}_gc", 0, 1, 2)
args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
opts = configure_optimizers(imodel, args, max_norm=None) + configure_optimizers(pmodel, args, max_norm=None)
class NoStep:
    def __init__(self, o): self.o, self.flat, self._sumsq = o, o.flat, o._sumsq
    def zero_grad(self): self.o.zero_grad()
    def step(self, *a, **k): pass
frames = [f.to(dev) for f in smooth_frames("dp2:gop", 2, 3, 64)]
qmap = closed_form_input("dp2:qmap", (2, 1, 64, 64), 0.0, 1.0).to(dev)
log = S.roi_gop_step(imodel, pmodel, PixelwiseRateDistortionLoss(), tuple(NoStep(o) for o in opts), frames, qmap, float(clip))
print("norms single", [float(l[1]) if l[1] is not None else None for l in log], "dp", r0["losses"][:, 1])
for key, o in (("grad_i", opts[0]), ("grad_p", opts[2])):
    full = o.flat.grad.detach().cpu().double().numpy()
    rows = []
    for name, p, off in zip(o.flat.names, o.flat.params, o.flat.offsets):
        n = p.numel(); a = r0[key][off:off+n].astype(np.float64); b = full[off:off+n]
        ratio = float((a * b).sum() / max((b * b).sum(), 1e-300))
        rows.append((float(np.abs(a - b).max() / (np.abs(b).max() or 1)), ratio, name))
    rows.sort(reverse=True)
    print(key, "worst:", [(f"{e:.2e}", f"{r:.4f}", n) for e, r, n in rows[:8]])
    print(key, "ratio quantiles", np.quantile([r[1] for r in rows], [0, .1, .5, .9, 1]))
