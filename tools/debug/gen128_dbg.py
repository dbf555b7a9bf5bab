import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfgs = [(2, 64, 9, 11, 96, 3), (1, 32, 8, 16, 128, 1), (1, 32, 8, 16, 256, 1), (1, 64, 8, 16, 128, 1)]
for B, C, H, W, K, R in cfgs:
  x = torch.randn(B, C, H, W, device=dev); w = torch.randn(K, C, R, R, device=dev) * 0.05; b = torch.zeros(K, device=dev)
  ref = torch.nn.functional.conv2d(x, w, b, padding=R // 2)
  print("cfg", B, C, H, W, K, R)
  for tile in (128,):
    with F.tuning(fx3_gen_tile=tile):
        y, _ = F.conv2d_f16x3_gen(F.F16Planes.split(x), F.pack_weight_f16x2_gen(w), b, K, R, R, 1, R // 2)
    torch.cuda.synchronize()
    d = (y - ref).abs()
    bad = ~(d < 1e-3)
    print(tile, "bad", int(bad.sum()), "nan", int(torch.isnan(y).sum()))
    if bad.any():
        bc = bad.permute(1, 0, 2, 3).reshape(K, -1)
        print(" bad channels:", [i for i, v in enumerate(bc.sum(1)) if int(v) > 3])
