"""Spatial feature transform blocks of the variable-rate (ROI) STEM models, with the reference's module and
parameter names (compressai/models/stem_utils.py:24-63):

    SFT(x_nc, prior_nc, ks, nhidden)    mlp_shared.0, mlp_gamma, mlp_beta      out = x * (1 + gamma(q)) + beta(q)
    SFTResblk(x_nc, prior_nc, ks)       conv_0, conv_1, norm_0, norm_1         out = x + conv_1(a(norm_1(conv_0(a(norm_0(x))))))

The modulation (and the leaky-ReLU that follows it inside SFTResblk / the hyper encoder) is one HIP pass
(stem_sft_fwd / stem_sft_bwd); the convolutions are the implicit-GEMM kernels with the ReLU folded in.
"""
import torch.nn as nn

from ..layers import AddFunction, Conv2d, FusedSequential, SFTFunction, adaptive_avg_pool2d

__all__ = ["SFT", "SFTResblk"]


class SFT(nn.Module):
    def __init__(self, x_nc, prior_nc=1, ks=3, nhidden=128):
        super().__init__()
        pw = ks // 2
        self.mlp_shared = FusedSequential(Conv2d(prior_nc, nhidden, kernel_size=ks, padding=pw), nn.ReLU())
        self.mlp_gamma = Conv2d(nhidden, x_nc, kernel_size=ks, padding=pw)
        self.mlp_beta = Conv2d(nhidden, x_nc, kernel_size=ks, padding=pw)

    def forward(self, x, qmap, slope=1.0):
        """`slope` != 1 applies leaky_relu(., slope) to the result in the same pass (callers that follow the
        SFT with an activation: stem_utils.py:56-57, stem_roi.py:566-573)."""
        qmap = adaptive_avg_pool2d(qmap, x.shape[2:])
        actv = self.mlp_shared(qmap)
        return SFTFunction.apply(x, self.mlp_gamma(actv), self.mlp_beta(actv), float(slope))


class SFTResblk(nn.Module):
    ACTV_SLOPE = 2e-1                                    # stem_utils.py:62-63

    def __init__(self, x_nc, prior_nc, ks=3):
        super().__init__()
        self.conv_0 = Conv2d(x_nc, x_nc, kernel_size=3, padding=1)
        self.conv_1 = Conv2d(x_nc, x_nc, kernel_size=3, padding=1)
        self.norm_0 = SFT(x_nc, prior_nc, ks=ks)
        self.norm_1 = SFT(x_nc, prior_nc, ks=ks)

    def forward(self, x, qmap):
        dx = self.conv_0(self.norm_0(x, qmap, slope=self.ACTV_SLOPE))
        dx = self.conv_1(self.norm_1(dx, qmap, slope=self.ACTV_SLOPE))
        return AddFunction.apply(x, dx)
