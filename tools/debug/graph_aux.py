import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_hip_graph import _build, _eager_step
from spatiotemporalentropymodel_amd.graphs import GraphedPFrameStep
from spatiotemporalentropymodel_amd.losses import EMLoss
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
ys = [torch.randn(2, 96, 8, 8, device=dev, generator=g) * 3 for _ in range(5)]
target = torch.empty(2, 3, 128, 128, device=dev)
crit = EMLoss()
stem_e, opt_e, aux_e = _build(); opt_e.enable_device_state(); aux_e.enable_device_state()
epoch = torch.zeros(1, dtype=torch.int64, device=dev)
stem_e.entropy_bottleneck.noise_epoch = stem_e.gaussian_conditional.noise_epoch = epoch
stem_g, opt_g, aux_g = _build()
gs = GraphedPFrameStep(stem_g, crit, opt_g, aux_g, (128, 128))
out, oc, al, gn = gs.step(ys[1], ys[0])
print("graph aux", float(al), "recomputed eagerly from graph model after replay (post aux step)", float(stem_g.aux_loss()))
for m_g, m_e in ((stem_g.entropy_bottleneck, stem_e.entropy_bottleneck), (stem_g.gaussian_conditional, stem_e.gaussian_conditional)):
    m_e._noise_offset = gs._capture_offsets[id(m_g)]
pre = float(stem_e.aux_loss())
out, oc, al_e, gn, grad = _eager_step(stem_e, opt_e, aux_e, crit, ys[1], ys[0], target, epoch)
print("eager aux before any step", pre, "eager aux after main step", float(al_e), "after aux step", float(stem_e.aux_loss()))
print("param diff main", float((opt_e.flat.data - opt_g.flat.data).abs().max()), "aux", float((aux_e.flat.data - aux_g.flat.data).abs().max()))
