#!/usr/bin/env python3
"""20 launches of g_a.0 + GDN (csrc/c4gdn_f16x3.hip, planes out) at the bench shape: the target of tools/debug/prof_pmc.sh"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import functional as F  # noqa: E402

B, K = int(os.environ.get("B", 16)), 192
torch.manual_seed(0)
x4 = F.nchw3_to_nhwc4(torch.rand(B, 3, 256, 256, device="cuda"))
wp = F.pack_weight(torch.randn(K, 3, 5, 5, device="cuda") * 0.1, F.PACK_CONV_FWD_C4)
b, beta = torch.randn(K, device="cuda") * 0.1, torch.rand(K, device="cuda") + 0.5
gamma = torch.rand(K, K, device="cuda") * 0.1 + 0.1 * torch.eye(K, device="cuda")
ast = F.c4gdn_stream(wp, gamma, K, 5, 5)
for _ in range(20):
    y = F.conv2d_c4_gdn_f16x3(x4, ast, b, beta, K, 5, 5, 2, 2, planes_out=True)
torch.cuda.synchronize()
