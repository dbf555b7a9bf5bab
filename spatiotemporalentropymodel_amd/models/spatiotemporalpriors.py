"""The five STEM P-frame entropy models with the reference's constructor signature, module names,
state-dict keys and method contracts (compressai/models/spatiotemporalpriors.py:33-1072):

    forward(y_cur, y_conditioned)    -> {"y_hat", "likelihoods": {"y", "z"}}
    compress(y_cur, y_conditioned)   -> {"strings": [y_strings, z_strings], "shape": z.shape[-2:]}
    decompress(strings, shape, y_conditioned) -> Tensor  (dict {"y_hat"} for the _Res model, as upstream)
    update(scale_table=None, force=False), load_state_dict, aux_loss

forward / backward run the fused HIP schedule of `engine.StemEngine`; the bitstream side keeps the
rANS coder on the host (codec.py).
"""
import math
import warnings

import torch
import torch.nn as nn

from .. import functional as F
from ..engine import StemEngine, StemFunction
from ..entropy_models import GaussianConditional
from ..layers import Conv2d, ConvTranspose2d, FusedSequential, LeakyReLU, MaskedConv2d
from .priors import CompressionModel
from .utils import update_registered_buffers

__all__ = ["get_scale_table", "SpatioTemporalPriorModelWithoutSPMTPM", "SpatioTemporalPriorModelWithoutSPM",
           "SpatioTemporalPriorModelWithoutTPM", "SpatioTemporalPriorModel", "SpatioTemporalPriorModel_Res"]

# From Balle's tensorflow compression examples (spatiotemporalpriors.py:22-30)
SCALES_MIN = 0.11
SCALES_MAX = 256
SCALES_LEVELS = 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):  # pylint: disable=W0622
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def _tpm(cin):
    return FusedSequential(Conv2d(cin, 256, kernel_size=5, padding=2, stride=1), LeakyReLU(),
                           Conv2d(256, 320, kernel_size=5, padding=2, stride=1), LeakyReLU(),
                           Conv2d(320, cin * 2, kernel_size=5, padding=2, stride=1))


def _he(cin, zc):
    return FusedSequential(Conv2d(cin * 2, 256, kernel_size=3, padding=1, stride=1), LeakyReLU(),
                           Conv2d(256, 256, kernel_size=5, padding=2, stride=2), LeakyReLU(),
                           Conv2d(256, zc, kernel_size=5, padding=2, stride=2))


def _hd(cin, zc):
    return FusedSequential(ConvTranspose2d(zc, 256, kernel_size=5, padding=2, stride=2, output_padding=1), LeakyReLU(),
                           ConvTranspose2d(256, 256, kernel_size=5, padding=2, stride=2, output_padding=1), LeakyReLU(),
                           Conv2d(256, cin * 2, kernel_size=3, padding=1, stride=1))


def _epm(cin, nprior):
    return FusedSequential(Conv2d(cin * 2 * nprior, 768, kernel_size=1), LeakyReLU(),
                           Conv2d(768, 576, kernel_size=1), LeakyReLU(),
                           Conv2d(576, cin * 2, kernel_size=1))


class _StemBase(CompressionModel):
    HAS_TPM = HAS_SPM = RESIDUAL = False
    HARD_CODED_Z = False          # the two ablations without SPM hard-code 256 hyper-latent channels (:44-58,150-164)
    DECOMPRESS_RETURNS_DICT = False

    def __init__(self, entropy_bottleneck_channels=256, in_channels=192):
        super().__init__(entropy_bottleneck_channels=entropy_bottleneck_channels)
        zc = 256 if self.HARD_CODED_Z else entropy_bottleneck_channels
        # module creation order follows the reference so that parameter / state-dict order is identical
        if self.HAS_TPM:
            self.TPM = _tpm(in_channels)
        self.HE = _he(in_channels, zc)
        self.HD = _hd(in_channels, zc)
        if self.HAS_SPM:
            self.context_prediction = MaskedConv2d(in_channels, in_channels * 2, kernel_size=5, padding=2, stride=1)
        self.EPM = _epm(in_channels, 1 + int(self.HAS_TPM) + int(self.HAS_SPM))
        self.gaussian_conditional = GaussianConditional(None)
        self.in_channels = in_channels
        self._engine = None
        self._engine_params = None

    # ---- fused forward -----------------------------------------------------------------------
    def engine(self) -> StemEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", StemEngine(self, self.HAS_TPM, self.HAS_SPM, self.RESIDUAL))
            object.__setattr__(self, "_engine_params", [p for n, p in self.named_parameters() if not n.endswith(".quantiles")])
        return self._engine

    def forward(self, y_cur, y_conditioned):
        eng = self.engine()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._engine_params):
            y_hat, lik_y, lik_z = StemFunction.apply(eng, self.training, y_cur, y_conditioned, *self._engine_params)
        else:
            y_hat, lik_y, lik_z, _ = eng.forward(y_cur, y_conditioned, self.training)
        return {"y_hat": y_hat, "likelihoods": {"y": lik_y, "z": lik_z}}

    # ---- bitstream side ----------------------------------------------------------------------
    def compress(self, y_cur, y_conditioned):
        from ..codec import stem_compress
        return stem_compress(self, y_cur, y_conditioned)

    def decompress(self, strings, shape, y_conditioned):
        from ..codec import stem_decompress
        assert isinstance(strings, list) and len(strings) == 2
        y_hat = stem_decompress(self, strings, shape, y_conditioned)
        return {"y_hat": y_hat} if self.DECOMPRESS_RETURNS_DICT else y_hat

    def load_state_dict(self, state_dict, strict=True):
        update_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                  ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        return super().load_state_dict(state_dict, strict=strict)

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= super().update(force=force)
        return updated


class SpatioTemporalPriorModelWithoutSPMTPM(_StemBase):
    """Hyper-prior only (spatiotemporalpriors.py:33-129)."""
    HARD_CODED_Z = True


class SpatioTemporalPriorModelWithoutSPM(_StemBase):
    """Hyper-prior + temporal prior (spatiotemporalpriors.py:132-243)."""
    HAS_TPM = True
    HARD_CODED_Z = True


class SpatioTemporalPriorModelWithoutTPM(_StemBase):
    """Hyper-prior + spatial (masked-conv) prior (spatiotemporalpriors.py:246-505)."""
    HAS_SPM = True


class SpatioTemporalPriorModel(_StemBase):
    """Hyper + temporal + spatial priors on y_cur (spatiotemporalpriors.py:508-788)."""
    HAS_TPM = HAS_SPM = True


class SpatioTemporalPriorModel_Res(_StemBase):
    """Hyper + temporal + spatial priors on the residual y_cur - y_conditioned (spatiotemporalpriors.py:791-1072)."""
    HAS_TPM = HAS_SPM = RESIDUAL = True
    DECOMPRESS_RETURNS_DICT = True
