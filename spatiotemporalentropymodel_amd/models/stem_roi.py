"""Pixel-domain STEM video models, including the variable-rate / ROI family driven by a quality map
(compressai/models/stem_roi.py), with the reference's class names, constructor arguments, module names and
state-dict keys:

    stem_baseline      :21-179    shared PEncoder for the current and the conditioning frame
    stem_baselinev2    :182-350   separate ConditionEncoder
    stem_roi           :353-699   SFT-modulated encoder / hyper-encoder / decoder (P frames)
    stem_roi_wo_gsc    :702-1014  as stem_roi without the decoder-side conditioning
    stem_roi_i         :1017-1325 the I-frame counterpart (no temporal prior)

    forward(x_cur[, x_conditioned][, Qmap]) -> {"x_hat", "y_hat", "likelihoods": {"y", "z"}}
    compress(...) -> {"strings": [y_strings, z_strings], "shape"};  decompress(strings, shape[, x_conditioned])
        -> {"x_hat", "y_hat", "entropy_params": {"scales_hat", "means_hat"}}

All tensors between the modules stay NHWC in HBM; torch.cat / chunk are channel slices of one buffer.  Every
convolution, GDN, SFT, pooling and likelihood is a HIP kernel behind the C ABI (autograd Functions in ..layers /
..entropy_models); there is no torch compute on this path besides scalar bookkeeping.
"""
import torch.nn as nn

from .. import functional as F
from ..entropy_models import GaussianConditional
from ..layers import GDN, Conv2d, ConvTranspose2d, FusedSequential, adaptive_avg_pool2d, cat, conv, deconv, to_nchw
from .priors import CompressionModel
from .spatiotemporalpriors import get_scale_table
from .stem_utils import SFT, SFTResblk
from .utils import update_registered_buffers

__all__ = ["stem_baseline", "stem_baselinev2", "stem_roi", "stem_roi_wo_gsc", "stem_roi_i"]


# ----------------------------------------------------------------------------- sub-network factories
def _transform(chs, layer, inverse):
    """conv/GDN (or deconv/IGDN) ladder: stem_roi.py:31-39, 47-55."""
    mods = []
    for i in range(len(chs) - 1):
        mods.append(layer(chs[i], chs[i + 1]))
        if i < len(chs) - 2:
            mods.append(GDN(chs[i + 1], inverse=inverse))
    return FusedSequential(*mods)


def _lrelu_chain(layers, slope=None):
    mods = []
    for i, m in enumerate(layers):
        mods.append(m)
        if i < len(layers) - 1:
            mods.append(nn.LeakyReLU() if slope is None else nn.LeakyReLU(slope, True))
    return FusedSequential(*mods)


def _tpm(c):
    return _lrelu_chain([Conv2d(c, 256, 5, 1, 2), Conv2d(256, 320, 5, 1, 2), Conv2d(320, c * 2, 5, 1, 2)])


def _epm(cin, c):
    return _lrelu_chain([Conv2d(cin, 768, 1), Conv2d(768, 576, 1), Conv2d(576, c * 2, 1)])


def _up2(cin, cmid, cout):
    """Two stride-2 transposed convolutions and a 3x3 convolution (hs / wmap_generator / HD)."""
    return _lrelu_chain([ConvTranspose2d(cin, cmid[0], 5, 2, 2, 1), ConvTranspose2d(cmid[0], cmid[1], 5, 2, 2, 1),
                         Conv2d(cmid[1], cout, 3, 1, 1)])


def _qmap_head(cin, c1, c2, cout):
    return _lrelu_chain([conv(cin, c1, 3, 1), conv(c1, c2, 3, 1), conv(c2, cout, 3, 1)], 0.1)


def _qmap_step(cin, cmid, cout, up=False):
    """Quality features one resolution level down (3x3 stride-2 conv) or up (3x3 stride-2 deconv), then 1x1."""
    return _lrelu_chain([(deconv if up else conv)(cin, cmid, 3), conv(cmid, cout, 1, 1)], 0.1)


# ----------------------------------------------------------------------------- shared coding logic
class _PixelStem(CompressionModel):
    """forward / compress / decompress shared by the five models; subclasses provide the four transforms."""

    TEMPORAL = True          # has x_conditioned + TPM
    QMAP = False             # takes a quality map

    def __init__(self, entropy_bottleneck_channels=256, in_channels=192):
        super().__init__(entropy_bottleneck_channels=entropy_bottleneck_channels)
        self.in_channels = int(in_channels)

    # hooks ------------------------------------------------------------------------------------
    def _analysis(self, x, Qmap):
        raise NotImplementedError

    def _condition(self, x_conditioned):
        raise NotImplementedError

    def _hyper_analysis(self, yy, Qmap):
        raise NotImplementedError

    def _synthesis(self, y_hat, z_hat):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------
    def _split_args(self, args):
        want = 1 + int(self.TEMPORAL) + int(self.QMAP)
        if len(args) != want:
            raise TypeError(f"{type(self).__name__} takes {want} tensors, got {len(args)}")
        x_cur = args[0]
        x_cond = args[1] if self.TEMPORAL else None
        Qmap = args[-1] if self.QMAP else None
        return x_cur, x_cond, Qmap

    def _gaussian_params(self, z_hat, y_conditioned):
        hyper = self.HD(z_hat)
        if self.TEMPORAL:
            return self.EPM(cat([self.TPM(y_conditioned), hyper]))
        return self.EPM(hyper)

    def _latents(self, x_cur, x_cond, Qmap):
        y_cur = self._analysis(x_cur, Qmap)
        y_conditioned = self._condition(x_cond) if self.TEMPORAL else None
        z = self._hyper_analysis(cat([y_cur, y_conditioned]) if self.TEMPORAL else y_cur, Qmap)
        return y_cur, y_conditioned, z

    def forward(self, *args):
        x_cur, x_cond, Qmap = self._split_args(args)
        y_cur, y_conditioned, z = self._latents(x_cur, x_cond, Qmap)
        z_hat, z_likelihoods = self.entropy_bottleneck(z)
        scales_hat, means_hat = self._gaussian_params(z_hat, y_conditioned).chunk(2, 1)
        y_hat, y_likelihoods = self.gaussian_conditional(y_cur, scales_hat, means=means_hat)
        x_hat = to_nchw(self._synthesis(y_hat, z_hat))         # plain NCHW image batch, differentiable
        return {"x_hat": x_hat, "y_hat": y_hat, "likelihoods": {"y": y_likelihoods, "z": z_likelihoods}}

    def compress(self, *args):
        x_cur, x_cond, Qmap = self._split_args(args)
        y_cur, y_conditioned, z = self._latents(x_cur, x_cond, Qmap)
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        scales_hat, means_hat = self._gaussian_params(z_hat, y_conditioned).chunk(2, 1)
        indexes = self.gaussian_conditional.build_indexes(scales_hat)
        y_strings = self.gaussian_conditional.compress(y_cur, indexes, means=means_hat)
        return {"strings": [y_strings, z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape, x_conditioned=None):
        assert isinstance(strings, list) and len(strings) == 2
        if self.TEMPORAL and x_conditioned is None:
            raise TypeError(f"{type(self).__name__}.decompress needs x_conditioned")
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        y_conditioned = self._condition(x_conditioned) if self.TEMPORAL else None
        scales_hat, means_hat = self._gaussian_params(z_hat, y_conditioned).chunk(2, 1)
        indexes = self.gaussian_conditional.build_indexes(scales_hat)
        y_hat = self.gaussian_conditional.decompress(strings[0], indexes, means=means_hat)
        x_hat = F.to_nchw(self._synthesis(y_hat, z_hat), clamp01=True)
        return {"x_hat": x_hat, "y_hat": y_hat, "entropy_params": {"scales_hat": scales_hat, "means_hat": means_hat}}

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


# ----------------------------------------------------------------------------- fixed-rate baselines
class stem_baseline(_PixelStem):
    """One analysis transform for both frames; TPM + hyper-prior fused by the EPM (stem_roi.py:21-179)."""

    SHARED_ENCODER = True

    def __init__(self, entropy_bottleneck_channels=256, in_channels=192):
        super().__init__(entropy_bottleneck_channels, in_channels)
        c = in_channels
        self.PEncoder = _transform([3, 128, 128, 128, c], conv, False)
        if not self.SHARED_ENCODER:
            self.ConditionEncoder = _transform([3, 128, 128, 128, c], conv, False)
        self.PDecoder = _transform([c, 128, 128, 128, 3], deconv, True)
        self.TPM = _tpm(c)
        self.HE = _lrelu_chain([Conv2d(c * 2, 256, 3, 1, 1), Conv2d(256, 256, 5, 2, 2), Conv2d(256, 256, 5, 2, 2)])
        self.HD = _up2(256, (256, 256), c * 2)
        self.EPM = _epm(c * 4, c)
        self.gaussian_conditional = GaussianConditional(None)

    def _analysis(self, x, Qmap):
        return self.PEncoder(x)

    def _condition(self, x_conditioned):
        return (self.PEncoder if self.SHARED_ENCODER else self.ConditionEncoder)(x_conditioned)

    def _hyper_analysis(self, yy, Qmap):
        return self.HE(yy)

    def _synthesis(self, y_hat, z_hat):
        return self.PDecoder(y_hat)

    def getY(self, x, isEval=False):
        """stem_roi.py:141-160: analysis transform; isEval zero-pads H, W to multiples of 64 (centred)."""
        if isEval:
            h, w = x.size(2), x.size(3)
            p = 64
            new_h, new_w = (h + p - 1) // p * p, (w + p - 1) // p * p
            left, top = (new_w - w) // 2, (new_h - h) // 2
            x = nn.functional.pad(x, (left, new_w - w - left, top, new_h - h - top), mode="constant", value=0)
        return self.PEncoder(x)


class stem_baselinev2(stem_baseline):
    """stem_baseline with its own ConditionEncoder for the previous reconstruction (stem_roi.py:182-350)."""

    SHARED_ENCODER = False


# ----------------------------------------------------------------------------- variable-rate (ROI) family
class _RoiStem(_PixelStem):
    QMAP = True
    GSC = True               # decoder-side conditioning on w = wmap_generator(z_hat)

    def __init__(self, entropy_bottleneck_channels=256, in_channels=192):
        super().__init__(entropy_bottleneck_channels, in_channels)
        c = in_channels
        yy = c * 2 if self.TEMPORAL else c
        # -- encoder (creation order as in the reference: stem_roi.py:357-399)
        for i, cin in enumerate((3, 128, 128), start=1):
            setattr(self, f"ga{i}", FusedSequential(conv(cin, 128), GDN(128)))
            setattr(self, f"ga{i}_SFT", SFT(x_nc=128, prior_nc=128, ks=3))
        self.ga4 = conv(128, c)
        self.ga4_SFTResB1 = SFTResblk(c, prior_nc=c, ks=3)
        self.ga4_SFTResB2 = SFTResblk(c, prior_nc=c, ks=3)
        self.qmap_feature_ga1 = _qmap_head(4, 192, 160, 128)
        self.qmap_feature_ga2 = _qmap_step(128, 128, 128)
        self.qmap_feature_ga3 = _qmap_step(128, 128, 128)
        self.qmap_feature_ga4 = _qmap_step(128, 128, c)
        # -- hyper encoder (:403-437)
        self.ha1 = conv(yy, 256, 3, 1)
        self.ha1_SFT = SFT(x_nc=256, prior_nc=256, ks=3)
        self.ha1_act = nn.LeakyReLU()
        self.ha2 = conv(256, 256, 5, 2)
        self.ha2_SFT = SFT(x_nc=256, prior_nc=256, ks=3)
        self.ha2_act = nn.LeakyReLU()
        self.ha3 = conv(256, 256, 5, 2)
        self.ha3_ResB1 = SFTResblk(256, 256, ks=3)
        self.ha3_ResB2 = SFTResblk(256, 256, ks=3)
        self.qmap_feature_ha1 = _qmap_head(yy + 1, 128, 192, 256)
        self.qmap_feature_ha2 = _qmap_step(256, 256, 256)
        self.qmap_feature_ha3 = _qmap_step(256, 256, 256)
        # -- hyper decoder (:441-447)
        self.hs = _up2(256, (256, 256), c * 2)
        # -- decoder (:452-498)
        if self.GSC:
            self.wmap_generator = _up2(256, (192, 128), 64)
            self.gs0_SFTResB1 = SFTResblk(c, prior_nc=c, ks=3)
            self.gs0_SFTResB2 = SFTResblk(c, prior_nc=c, ks=3)
        for i, cin in enumerate((c, 128, 128), start=1):
            setattr(self, f"gs{i}", FusedSequential(deconv(cin, 128), GDN(128, inverse=True)))
            if self.GSC:
                setattr(self, f"gs{i}_SFT", SFT(x_nc=128, prior_nc=128, ks=3))
        self.gs4 = deconv(128, 3)
        if self.GSC:
            self.qmap_feature_gs0 = _qmap_head(64 + c, 192, 192, 192)
            self.qmap_feature_gs1 = _qmap_step(192, 128, 128, up=True)
            self.qmap_feature_gs2 = _qmap_step(128, 128, 128, up=True)
            self.qmap_feature_gs3 = _qmap_step(128, 128, 128, up=True)
        # -- priors (:503-531)
        self.ConditionEncoder = _transform([3, 128, 128, 128, c], conv, False)
        if self.TEMPORAL:
            self.TPM = _tpm(c)
        self.EPM = _epm(c * 4 if self.TEMPORAL else c * 2, c)
        self.gaussian_conditional = GaussianConditional(None)

    # ---- transforms (method names as in the reference: PEncoder / PDecoder / HE / HD) ----------
    def PEncoder(self, x, Qmap):
        """stem_roi.py:534-550: each analysis stage is modulated by quality-map features at its resolution."""
        q = self.qmap_feature_ga1(cat([x, Qmap]))
        for i in (1, 2, 3):
            if i > 1:
                q = getattr(self, f"qmap_feature_ga{i}")(q)
            x = getattr(self, f"ga{i}_SFT")(getattr(self, f"ga{i}")(x), q)
        q = self.qmap_feature_ga4(q)
        x = self.ga4(x)
        return self.ga4_SFTResB2(self.ga4_SFTResB1(x, q), q)

    def PDecoder(self, x, z):
        """stem_roi.py:552-573 (with GSC) / :843-852 (without)."""
        w = None
        if self.GSC:
            w = self.qmap_feature_gs0(cat([self.wmap_generator(z), x]))
            x = self.gs0_SFTResB2(self.gs0_SFTResB1(x, w), w)
        for i in (1, 2, 3):
            x = getattr(self, f"gs{i}")(x)
            if self.GSC:
                w = getattr(self, f"qmap_feature_gs{i}")(w)
                x = getattr(self, f"gs{i}_SFT")(x, w)
        return self.gs4(x)

    def HE(self, x, Qmap):
        """stem_roi.py:575-593"""
        q = self.qmap_feature_ha1(cat([adaptive_avg_pool2d(Qmap, x.shape[2:]), x]))
        x = self.ha1_SFT(self.ha1(x), q, slope=self.ha1_act.negative_slope)
        q = self.qmap_feature_ha2(q)
        x = self.ha2_SFT(self.ha2(x), q, slope=self.ha2_act.negative_slope)
        q = self.qmap_feature_ha3(q)
        x = self.ha3(x)
        return self.ha3_ResB2(self.ha3_ResB1(x, q), q)

    def HD(self, x):
        return self.hs(x)

    def _analysis(self, x, Qmap):
        return self.PEncoder(x, Qmap)

    def _condition(self, x_conditioned):
        return self.ConditionEncoder(x_conditioned)

    def _hyper_analysis(self, yy, Qmap):
        return self.HE(yy, Qmap)

    def _synthesis(self, y_hat, z_hat):
        return self.PDecoder(y_hat, z_hat)


class stem_roi(_RoiStem):
    """P-frame model: forward(x_cur, x_conditioned, Qmap) (stem_roi.py:353-699)."""

    def forward_compress(self, x_cur, x_conditioned, Qmap):
        """stem_roi.py:622-641: forward without the reconstruction in the returned dictionary."""
        out = self.forward(x_cur, x_conditioned, Qmap)
        return {"y_hat": out["y_hat"], "likelihoods": out["likelihoods"]}


class stem_roi_wo_gsc(_RoiStem):
    """stem_roi without wmap_generator / decoder SFTs (stem_roi.py:702-1014)."""

    GSC = False


class stem_roi_i(_RoiStem):
    """I-frame model: forward(x_cur, Qmap); keeps an (unused) ConditionEncoder like the reference so that
    checkpoints load key-for-key (stem_roi.py:1017-1325)."""

    TEMPORAL = False
