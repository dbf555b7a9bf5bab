import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spatiotemporalentropymodel_amd.models as M
from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
dev = torch.device("cuda:0")
m = closed_form_fill_(M.SpatioTemporalPriorModel_Res(64, 96)).to(dev).eval()
m.update(force=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
y_cur = closed_form_input("lb:y", (B, 96, 8, 12), -6, 6).to(dev)
y_cond = closed_form_input("lb:c", (B, 96, 8, 12), -6, 6).to(dev)
with torch.no_grad():
    enc = m.compress(y_cur, y_cond)
    a = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
    os.environ["STEM_AR_NO_BATCH"] = "1"
    b = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
d = (a - b).abs()
for g in range(B):
    nz = (d[g] > 0).nonzero()
    print("image", g, "max diff", float(d[g].max()), "count", int((d[g] > 0).sum()), "first", nz[0].tolist() if len(nz) else None)
