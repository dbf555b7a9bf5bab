"""Flat parameter / gradient buffers and the fused clip + Adam step.

The reference's per-step optimiser work (stem/trainSTEM.py:213-218, utils.py:104-135) is
`clip_grad_norm_` over 41 tensors followed by two torch Adam steps: ~250 small kernels.  Here every
STEM parameter is a view into ONE flat fp32 buffer (and its gradient a view into a second one), so the
global norm is one reduction, clip + Adam one elementwise pass, and the data-parallel gradient
exchange one RCCL all-reduce over the same buffer (distributed.py).
"""
from __future__ import annotations

import torch

from . import functional as F
from .layers import bump_weight_epoch, join_wgrad_stream


class FlatParameters:
    """Re-homes `params` into one contiguous buffer; `.grad`s become views of a second buffer."""

    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned for the float4 kernels
        self.numel = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p._flat_grad_view = self.grad[o:o + n].view(p.shape)
            p.grad = p._flat_grad_view
        bump_weight_epoch(self.params)

    def zero_grad(self):
        join_wgrad_stream()
        self.grad.zero_()
        for p in self.params:
            p.grad = p._flat_grad_view


class FusedClipAdam:
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps) over a FlatParameters in two kernel launches.
    `max_norm=None` disables clipping (the reference does not clip the aux optimiser's `.quantiles`)."""

    def __init__(self, flat: FlatParameters, lr, max_norm=None, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.max_norm, self.betas, self.eps = flat, float(lr), max_norm, betas, float(eps)
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.t = 0
        self._sumsq = F.sumsq_accumulator(flat.data.device)          # [0] = sum of squares, rest = reduction scratch
        self.param_groups = [{"lr": self.lr, "params": flat.params}]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def grad_norm(self):
        """Global L2 norm of the current flat gradient as a 0-dim device tensor (no host sync)."""
        join_wgrad_stream()
        self._sumsq[:1].zero_()
        F.sumsq(self.flat.grad, self._sumsq)
        return self._sumsq[0].sqrt().reshape(())

    def step(self, grad_scale: float = 1.0, norm_is_current: bool = False):
        """grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).  norm_is_current: the caller
        has just called grad_norm() on these very gradients (as the training loop does to report the norm), so the
        reduction is not repeated."""
        join_wgrad_stream()
        self.t += 1
        self.lr = self.param_groups[0]["lr"]
        use_clip = self.max_norm is not None and self.max_norm > 0
        if use_clip and not norm_is_current:
            self._sumsq[:1].zero_()
            F.sumsq(self.flat.grad, self._sumsq)
        F.adam_step(self.flat.data, self.flat.grad, self.m, self.v, self._sumsq if use_clip else None,
                    float(self.max_norm) if use_clip else 0.0, float(grad_scale), self.lr, self.betas[0], self.betas[1],
                    self.eps, self.t)
        bump_weight_epoch(self.flat.params)

    def state_dict(self):
        return {"t": self.t, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.t, self.lr = int(sd["t"]), float(sd["lr"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups[0]["lr"] = self.lr


def clip_grad_norm_(optimizers, max_norm, tensors=None):
    """torch.nn.utils.clip_grad_norm_ over the union of the optimisers' flat gradient buffers, without a host sync:
    one reduction per buffer into a shared fp64 accumulator, then one scaling pass per buffer.  Returns the norm
    before clipping as a 0-dim device tensor.  (The variable-rate training loop clips after every frame while the
    gradients of a GOP keep accumulating: stem_roi/train_stem_roi.py:536,563.)  `tensors` overrides which flat
    buffers are clipped (the data-parallel loop clips the running global sums, distributed.GopGradAccumulator)."""
    join_wgrad_stream()
    optimizers = [o for o in optimizers if o is not None]
    bufs = list(tensors) if tensors is not None else [o.flat.grad for o in optimizers]
    acc = optimizers[0]._sumsq
    acc[:1].zero_()
    for g in bufs:
        F.sumsq(g, acc)
    for g in bufs:
        F.clip_scale(g, acc, max_norm)
    return acc[0].sqrt().reshape(())


def configure_optimizers(net, args, fused=True, max_norm=1.0):
    """utils.py:104-135: main Adam over everything but `.quantiles`, aux Adam over `.quantiles`.
    fused=True returns FusedClipAdam objects (clip folded into the main step); fused=False returns the
    reference's plain torch.optim.Adam pair."""
    named = list(net.named_parameters())
    main = sorted([(n, p) for n, p in named if not n.endswith(".quantiles") and p.requires_grad], key=lambda t: t[0])
    aux = sorted([(n, p) for n, p in named if n.endswith(".quantiles") and p.requires_grad], key=lambda t: t[0])
    assert len(main) + len(aux) == len([1 for _, p in named if p.requires_grad])
    if not fused:
        return (torch.optim.Adam((p for _, p in main), lr=args.learning_rate),
                torch.optim.Adam((p for _, p in aux), lr=args.aux_learning_rate))
    return (FusedClipAdam(FlatParameters(main), args.learning_rate, max_norm=max_norm),
            FusedClipAdam(FlatParameters(aux), args.aux_learning_rate, max_norm=None))
