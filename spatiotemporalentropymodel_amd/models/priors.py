"""CompressionModel base class and the I-frame transforms of JointAutoregressiveHierarchicalPriors
("mbt2018": compressai/models/priors.py:42-106, 406-694).

On the STEM training path only g_a / g_s (getY / getX) are executed.  The evaluation loop also codes its I frames with this
model (stem/evalSTEM.py:54-59: compress / decompress; priors.py:476-716): forward (inference), compress and decompress run the
hyper-prior modules and the same raster-order coder as the STEM models (codec.py: the entropy-parameter network sees
cat(params, ctx) exactly like a STEM model without temporal prior).
"""
import torch
import torch.nn as nn

from .. import functional as F
from ..entropy_models import EntropyBottleneck, GaussianConditional
from ..layers import GDN, Conv2d, ConvTranspose2d, FusedSequential, LeakyReLU, MaskedConv2d, cat, conv, deconv
from .utils import update_registered_buffers

__all__ = ["CompressionModel", "JointAutoregressiveHierarchicalPriors"]


class CompressionModel(nn.Module):
    def __init__(self, entropy_bottleneck_channels, init_weights=True):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)
        if init_weights:
            # As in the reference this runs BEFORE subclasses create their layers (priors.py:51-56), so
            # convolutions keep torch's default nn.Conv2d initialisation; kept for behavioural parity.
            self._initialize_weights()

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def _initialize_weights(self):
        """kaiming_normal_ weights, zero biases for every (transposed) convolution present (priors.py:67-72)."""
        for m in self.modules():
            if isinstance(m, (Conv2d, ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, *args):
        raise NotImplementedError()

    def update(self, force=False):
        updated = False
        for m in self.children():
            if not isinstance(m, EntropyBottleneck):
                continue
            updated |= m.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        update_registered_buffers(self.entropy_bottleneck, "entropy_bottleneck",
                                  ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
        return super().load_state_dict(state_dict, strict=strict)


class JointAutoregressiveHierarchicalPriors(CompressionModel):
    def __init__(self, N=192, M=192, **kwargs):
        super().__init__(entropy_bottleneck_channels=N, **kwargs)
        self.g_a = FusedSequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = FusedSequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True),
                                   deconv(N, N), GDN(N, inverse=True), deconv(N, 3))
        self.h_a = FusedSequential(conv(M, N, stride=1, kernel_size=3), LeakyReLU(inplace=True),
                                   conv(N, N, stride=2, kernel_size=5), LeakyReLU(inplace=True),
                                   conv(N, N, stride=2, kernel_size=5))
        self.h_s = FusedSequential(deconv(N, M, stride=2, kernel_size=5), LeakyReLU(inplace=True),
                                   deconv(M, M * 3 // 2, stride=2, kernel_size=5), LeakyReLU(inplace=True),
                                   conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))
        self.entropy_parameters = FusedSequential(Conv2d(M * 12 // 3, M * 10 // 3, 1), LeakyReLU(inplace=True),
                                                  Conv2d(M * 10 // 3, M * 8 // 3, 1), LeakyReLU(inplace=True),
                                                  Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)

    # ---- the two entry points the STEM scripts use (priors.py:686-694, 397-402) ---------------
    def getY(self, x):
        """-> (y, y + U(-1/2,1/2)); the mbt2018 variant adds noise in eval mode too (priors.py:691)."""
        y = self.g_a(x)
        y_quantized = self.gaussian_conditional.quantize(y, "noise")
        return y, y_quantized

    def getX(self, y_hat):
        """g_s + clamp(0,1); returns a contiguous NCHW image batch (layout change and clamp fused)."""
        return F.to_nchw(self.g_s(y_hat), clamp01=True)

    # ---- the I-frame codec (priors.py:476-716; stem/evalSTEM.py:54-59) ---------------------------------------------------------
    #: what codec.py's raster-order coder asks a model: a spatial prior, no temporal prior, the latents themselves are coded
    HAS_SPM, HAS_TPM, RESIDUAL = True, False, False

    @property
    def in_channels(self):
        return self.M

    @property
    def EPM(self):
        return self.entropy_parameters

    def forward(self, x):
        """priors.py:476-508 -> {"y", "y_hat", "x_hat", "likelihoods": {"y", "z"}, "entropy_params": {"scales_hat", "means_hat"}}"""
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_likelihoods = self.entropy_bottleneck(z)
        params = self.h_s(z_hat)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx_params = self.context_prediction(y_hat)
        gaussian_params = self.entropy_parameters(cat([params, ctx_params]))
        scales_hat, means_hat = gaussian_params.chunk(2, 1)
        _, y_likelihoods = self.gaussian_conditional(y, scales_hat, means=means_hat)
        x_hat = self.g_s(y_hat)
        return {"y": y, "y_hat": y_hat, "x_hat": x_hat, "likelihoods": {"y": y_likelihoods, "z": z_likelihoods},
                "entropy_params": {"scales_hat": scales_hat, "means_hat": means_hat}}

    def compress(self, x):
        """priors.py:544-584 -> {"strings": [y_strings, z_strings], "shape": z.shape[-2:]}"""
        from ..codec import iframe_compress
        with torch.no_grad():
            return iframe_compress(self, x)

    def decompress(self, strings, shape):
        """priors.py:633-674 -> {"x_hat", "y_hat"}"""
        from ..codec import iframe_decompress
        with torch.no_grad():
            return iframe_decompress(self, strings, shape)

    def update(self, scale_table=None, force=False):
        """the Gaussian tables next to the bottleneck's (priors.py's MeanScaleHyperprior.update, which mbt2018 inherits)"""
        from .spatiotemporalpriors import get_scale_table
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= super().update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        update_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                  ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        return super().load_state_dict(state_dict, strict=strict)
