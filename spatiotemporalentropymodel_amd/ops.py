"""LowerBound / NonNegativeParametrizer: the reference's small operator modules
(compressai/ops/bound_ops.py:19-53, compressai/ops/parametrizers.py:21-45).

On the HIP path both are fused into kernels (the parametrizer into the GDN kernel's operand
staging and epilogue, the lower bounds with their pass-through gradient rule into the likelihood
kernels), so these classes mainly carry the registered buffers that keep state_dict keys identical
to the reference's (`…beta_reparam.pedestal`, `…lower_bound.bound`, `likelihood_lower_bound.bound`).
Their forward is kept for host-side use (e.g. `update()`), on whatever device the input lives.
"""
import torch
import torch.nn as nn


class LowerBoundFunction(torch.autograd.Function):
    """max(x, bound) with the gradient passed through when x >= bound or when it pushes x up."""

    @staticmethod
    def forward(ctx, input_, bound):
        ctx.save_for_backward(input_, bound)
        return torch.max(input_, bound)

    @staticmethod
    def backward(ctx, grad_output):
        input_, bound = ctx.saved_tensors
        keep = (input_ >= bound) | (grad_output < 0)
        return keep.type(grad_output.dtype) * grad_output, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return LowerBoundFunction.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((self.minimum + self.reparam_offset ** 2) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal
