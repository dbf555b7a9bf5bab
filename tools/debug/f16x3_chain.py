#!/usr/bin/env python3
"""Error of the whole analysis transform g_a (mbt2018, N = M = 192) against an fp64 evaluation, for the fp32-MFMA kernels and for
the split-operand fp16 kernels (two planes, three products per fp32 product)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd.layers import GDN  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_  # noqa: E402
from spatiotemporalentropymodel_amd.zoo import models  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16      # 16 = the bench batch: the split-operand chain with its 128-pixel tiles (below 12288 output pixels a layer stays on fp32)


def ga64(imodel, x):
    h = x.double().cpu()
    for m in imodel.g_a:
        if isinstance(m, GDN):
            ped = 2.0 ** -36
            bb = torch.clamp(m.beta.double().cpu(), min=(m.beta_min + ped) ** 0.5) ** 2 - ped
            gg = torch.clamp(m.gamma.double().cpu(), min=2.0 ** -18) ** 2 - ped
            h = h / torch.sqrt(torch.nn.functional.conv2d(h * h, gg[:, :, None, None], bb))
        else:
            h = torch.nn.functional.conv2d(h, m.weight.double().cpu(), m.bias.double().cpu(), stride=m.stride, padding=m.padding)
    return h


for fill in ("closed_form", "default_init"):
    torch.manual_seed(3)
    imodel = models["mbt2018"](quality=4)
    if fill == "closed_form":
        closed_form_fill_(imodel)
    imodel = imodel.to(dev).eval()
    yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
    x = torch.stack([torch.stack([0.5 + 0.4 * torch.sin((xx + 17 * g) / (9.0 + 3 * c + g)) * torch.cos(yy / (7.0 - c)) for c in range(3)])
                     for g in range(B)]) + 0.05 * torch.rand(B, 3, 256, 256, device=dev)
    x = x.clamp(0, 1)
    ref = ga64(imodel, x)
    scale = float(ref.abs().max())
    rms = float(ref.pow(2).mean().sqrt())
    print(f"{fill} (B = {B}): max|y| {scale:.3f} rms {rms:.3f}")
    for name, env in (("fp32-MFMA", {"STEM_F16X3": "0"}), ("fp16 x3", {})):
        os.environ.pop("STEM_F16X3", None)
        os.environ.update(env)
        with torch.no_grad():
            y = imodel.g_a(x)
        torch.cuda.synchronize()
        d = (y.double().cpu() - ref).abs()
        print(f"  {name:10s} max abs err {float(d.max()):.3e}  = {float(d.max()) / scale:.2e} of max|y|;  rms err {float(d.pow(2).mean().sqrt()):.3e} = {float(d.pow(2).mean().sqrt()) / rms:.2e} of rms")
