"""time of one 64x64-tile igemm launch vs number of workgroups (no split-K): conv 5x5 s1, C=64 -> N=256, i.e. 50 chunks per
workgroup, 4 n-tiles; M = B*16*16 varied so that workgroups = 4*M/64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatiotemporalentropymodel_amd import functional as F
C, K = int(os.environ.get("CIN", 64)), 256
w = torch.randn(K, C, 5, 5, device="cuda") * 0.02
wp = F.pack_weight(w, F.PACK_CONV_FWD)
b = torch.zeros(K, device="cuda")
for B in (4, 8, 12, 15, 16, 17, 20, 24, 32, 48, 64, 128):
    x = torch.randn(B, C, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
    fn = lambda: F.conv2d_fwd(x, wp, b, K, 5, 5, 1, 2)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    wgs = B * 256 // 64 * 4
    flop = 2.0 * B * 256 * K * C * 25
    print(f"B={B:4d} workgroups={wgs:5d} ({wgs/256:5.2f}/CU)  {us:8.1f} us  {flop/us/1e6:6.1f} TF  us per (WG/CU) {us/(wgs/256):7.1f}")
