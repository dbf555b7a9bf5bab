"""Training criteria with the reference's names and output dictionaries (root utils.py:8-74).

EMLoss is the trainSTEM criterion: rate only.  The log-sum reduction and its gradient are HIP kernels
(stem_log2_sum / stem_dlog); the scalar bookkeeping around them is torch.
"""
import math

import torch
import torch.nn as nn

from . import functional as F


class _Log2SumFunction(torch.autograd.Function):
    """sum(log2(lik)) as one fp64-accumulating reduction kernel; backward: dlik = g / (lik * ln2)."""

    @staticmethod
    def forward(ctx, lik):
        likd = F.to_nhwc(lik)
        if F.nhwc_ld(likd) != likd.shape[1]:
            likd = F.copy_channels(likd, F.empty_nhwc(*likd.shape, likd.device))
        acc = torch.zeros(1, dtype=torch.float64, device=lik.device)
        F.log2_sum(likd, acc)
        ctx.save_for_backward(likd)
        return acc.reshape(())

    @staticmethod
    def backward(ctx, g):
        (likd,) = ctx.saved_tensors
        # g is a 0-dim fp64 tensor that stays on the device (no host sync): coef = g / ln 2
        return F.dlog(likd, 1.0) * (g.float() / math.log(2.0))


def log2_sum(lik):
    return _Log2SumFunction.apply(lik)


class EMLoss(nn.Module):
    """Entropy-model loss (utils.py:8-27): bpp of y and z, summed."""

    def forward(self, stpm_out, target):
        N, _, H, W = target.size()
        num_pixels = N * H * W
        out = {}
        out["y_bpp_loss"] = log2_sum(stpm_out["likelihoods"]["y"]) / (-num_pixels)
        out["z_bpp_loss"] = log2_sum(stpm_out["likelihoods"]["z"]) / (-num_pixels)
        out["loss"] = out["y_bpp_loss"] + out["z_bpp_loss"]
        return out


class RateDistortionLoss(nn.Module):
    """utils.py:30-50: bpp + lmbda * 255^2 * MSE(x_hat, target).  The distortion term is the same fp64-accumulating
    squared-error kernel the variable-rate criterion uses (stem_weighted_sqerr_sum / _bwd) with a unit weight map."""

    def __init__(self, lmbda=1e-2):
        super().__init__()
        self.lmbda = lmbda
        self._ones = None

    def _unit_map(self, target):
        B, _, H, W = target.shape
        o = self._ones
        if o is None or o.shape != (B, 1, H, W) or o.device != target.device:
            o = self._ones = torch.ones(B, 1, H, W, device=target.device, dtype=torch.float32)
        return o

    def forward(self, output, target):
        N, _, H, W = target.size()
        num_pixels = N * H * W
        out = {}
        out["bpp_loss"] = sum(log2_sum(l) / (-num_pixels) for l in output["likelihoods"].values())
        out["mse_loss"] = _WeightedMSEFunction.apply(output["x_hat"], target, self._unit_map(target))
        out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        return out


class _WeightedMSEFunction(torch.autograd.Function):
    """mean(lambda * (x_hat - x)^2) over [B,C,H,W] as one fp64-accumulating kernel; backward in one pass."""

    @staticmethod
    def forward(ctx, x_hat, target, lmbdamap):
        x_hat, target, lmbdamap = x_hat.contiguous(), target.contiguous(), lmbdamap.contiguous()
        ctx.save_for_backward(x_hat, target, lmbdamap)
        return F.weighted_sqerr_sum(x_hat, target, lmbdamap) / x_hat.numel()

    @staticmethod
    def backward(ctx, g):
        x_hat, target, lmbdamap = ctx.saved_tensors
        return F.weighted_sqerr_bwd(x_hat, target, lmbdamap, g, 1.0 / x_hat.numel()), None, None


class PixelwiseRateDistortionLoss(nn.Module):
    """Rate + spatially weighted distortion of the variable-rate models (utils.py:53-74): lmbdamap is [B,1,H,W]."""

    def forward(self, output, target, lmbdamap):
        N, _, H, W = target.size()
        num_pixels = N * H * W
        out = {}
        out["bpp_loss"] = sum(log2_sum(l) / (-num_pixels) for l in output["likelihoods"].values())
        out["mse_loss"] = _WeightedMSEFunction.apply(output["x_hat"], target, lmbdamap)
        out["loss"] = 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        return out


def quality2lambda(qmap):
    """utils.py:97-101"""
    return 0.002 * torch.exp(3.4409 * qmap)
