"""Layer modules with the reference's names, constructor arguments and state-dict keys, whose
forward/backward run the HIP kernels through the C ABI.

  Conv2d / ConvTranspose2d   torch.nn.Conv2d / ConvTranspose2d as built by compressai/models/utils.py:112-130
  MaskedConv2d               compressai/layers/layers.py:21-47
  GDN                        compressai/layers/gdn.py:22-67
  FusedSequential            nn.Sequential that fuses Conv -> LeakyReLU pairs into the conv epilogue
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import config as _config
from . import functional as F
from .ops import NonNegativeParametrizer

_WEIGHT_EPOCH = [0]


def bump_weight_epoch(params=None):
    """Called by the fused optimiser: parameters changed behind torch's version counters.  With `params` only those
    tensors are marked (each carries its own counter), so packed copies of other models' weights -- e.g. the frozen
    I-frame transforms next to a training STEM -- stay valid; without, every cache is invalidated."""
    if params is None:
        _WEIGHT_EPOCH[0] += 1
        return
    for p in params:
        p._stem_epoch = getattr(p, "_stem_epoch", 0) + 1


def weight_epoch(w):
    return (_WEIGHT_EPOCH[0], getattr(w, "_stem_epoch", 0))


PACK_F16X2 = -3          # _PackCache role of the pre-split fp16 image (csrc/conv_f16x3.hip); not a stem_pack_* role
PACK_F16X2_GEN = -4      # ... in the layout of the general (128-column tiles, split-K) kernel
PACK_F16X2_FLIP = -7     # PACK_F16X2 of the mirrored, transposed weight (input gradient on the 192-column kernel)
PACK_F16X2_GEN_FLIP = -6 # ... of the mirrored, transposed weight: the input-gradient of a stride-1 convolution as a convolution
PACK_C4GDN = -5           # A-operand stream of csrc/c4gdn_f16x3.hip: first-layer weight AND the following GDN's gamma
PACK_GDN_GAMMA = -8       # reparametrised gamma of the GDN fused into conv_f16x3_kernel, packed as a 1x1 weight image


class _PackCache:
    """Packed weight copies, rebuilt only when the parameter changed."""

    def __init__(self):
        self._c = {}

    def get(self, w: torch.Tensor, role: int, masked: int = 0):
        key = (w._version, w.data_ptr(), weight_epoch(w), tuple(w.shape))
        hit = self._c.get(role)
        if hit is not None and hit[0] == key:
            return hit[1]
        if role == PACK_F16X2:
            wp = F.pack_weight_f16x2(w)
        elif role == PACK_F16X2_FLIP:
            wp = F.pack_weight_f16x2(w, flip=True)
        elif role == PACK_F16X2_GEN:
            wp = F.pack_weight_f16x2_gen(w)
        elif role == PACK_F16X2_GEN_FLIP:
            wp = F.pack_weight_f16x2_gen(w, flip=True)
        elif role == PACK_GDN_GAMMA:
            wp = F.pack_gdn_gamma_f16x2(w)
        else:
            wp = F.pack_weight(w, role, masked)
        if (masked & 3) == 2:                    # the kernel zeroed taps of w in place
            key = (w._version, w.data_ptr(), weight_epoch(w), tuple(w.shape))
        self._c[role] = (key, wp)
        return wp

    def get_c4gdn(self, w: torch.Tensor, gamma: torch.Tensor, K: int, R: int):
        """the combined (first-layer weight, GDN gamma) stream of F.conv2d_c4_gdn_f16x3, rebuilt when either parameter changed"""
        key = tuple((t._version, t.data_ptr(), weight_epoch(t), tuple(t.shape)) for t in (w, gamma))
        hit = self._c.get(PACK_C4GDN)
        if hit is not None and hit[0] == key:
            return hit[1]
        st = F.c4gdn_stream(self.get(w, F.PACK_CONV_FWD_C4), gamma, K, R, R)
        self._c[PACK_C4GDN] = (key, st)
        return st


def _flat_grad(p):
    """The parameter's slot in a FlatParameters gradient buffer when `p.grad` currently IS that slot, else None.
    Weight / bias gradients are then accumulated straight into the slot by the unpack / column-sum kernels and the
    autograd Function returns None for them: exactly what AccumulateGrad's `p.grad += g` would do, without the
    temporary and the extra pass.  (Only `.backward()` accumulation is served this way; torch.autograd.grad() with
    these parameters as inputs would see None -- use plain parameters, i.e. no FlatParameters, for that.)"""
    view = getattr(p, "_flat_grad_view", None)
    return view if view is not None and p.grad is view else None


# Weight gradients are off the critical path of back-propagation (nothing downstream consumes them), so when they are
# accumulated straight into a flat gradient buffer they are launched on a side stream: the many small layers of the
# variable-rate models (16x16 .. 4x4 feature maps) then overlap their latency-bound wgrad / column-sum / unpack kernels
# with the dgrad chain.  All weight-gradient work shares that one stream (ordered among itself, so the shared
# geometry-keyed workspaces and repeated accumulation into one parameter stay race-free); the compute stream re-joins
# at the end of every backward pass (autograd engine callback) and wherever gradients are consumed (optim.py).
_WGRAD_SIDE = {"enabled": True, "streams": {}, "queued": False, "keep": []}


def wgrad_side_stream(device):
    st = _WGRAD_SIDE["streams"].get(device)
    if st is None:
        st = _WGRAD_SIDE["streams"][device] = F.make_stream(device, "side")
    return st


def join_wgrad_stream():
    """Order all outstanding side-stream weight-gradient work before whatever the current stream does next."""
    _WGRAD_SIDE["queued"] = False
    for dev, st in _WGRAD_SIDE["streams"].items():
        F.stream_wait(F.cur_stream(dev), st)
    _WGRAD_SIDE["keep"].clear()          # the current stream is now ordered after every reader (see _on_side_stream)


def _on_side_stream(fn, *tensors):
    dev = tensors[0].device
    side = wgrad_side_stream(dev)
    F.stream_wait(side, F.cur_stream(dev))
    with F.on_stream(side):
        fn()
    for t in tensors:
        t.record_stream(side)            # the allocator must not hand the memory out again before the side stream is done
    # ... and nobody may WRITE it before then either.  autograd sums fan-out gradients in place when it holds the last
    # reference to a buffer (InputBuffer::accumulate): the gradient of a residual add reaches a convolution's backward
    # AND, as the very same tensor, the skip connection, where the engine later does `dy.add_(dx_branch)` on the compute
    # stream -- while the weight-gradient kernel queued here may not have read dy yet.  (Observed as wrong conv_1
    # gradients of the residual blocks as soon as a second process competed for the GPU and delayed the side stream;
    # tests/test_hip_dp2.py.)  Holding a reference until the streams are joined keeps such buffers out of place.
    _WGRAD_SIDE["keep"].extend(tensors)
    if not _WGRAD_SIDE["queued"]:
        _WGRAD_SIDE["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)


# ----------------------------------------------------------------------------- autograd functions
def _layers_f16x3_enabled():
    """stride-1 convolutions of the layer-wise (autograd) models -- the variable-rate family of models/stem_roi.py -- on the fp16
    matrix cores (three fp16 products per fp32 product on two-plane operands, ~2^-21 per product: csrc/conv_f16x3.hip / wgrad_f16x3.hip); STEM_LAYERS_F16X3=0: fp32 MFMA"""
    return _config.runtime().layers_f16x3


def _conv_f16x3_route(weight, stride, pad, masked, x_shape):
    """forward, input gradient and weight gradient of this convolution on the fp16 kernels: stride 1, 'same' padding, channel
    counts that are multiples of 32, operands within the kernels' 2 GiB buffer views"""
    K, Cc, R, S = weight.shape
    B, _, H, W = x_shape
    # odd windows only: with an even R the output is (H + 1) x (W + 1) and the input gradient needs pad R - 1 - pad, which the
    # planes hand-over between layers ('same' shapes) does not carry
    return (stride == 1 and R == S and R % 2 == 1 and pad == R // 2 and not masked and Cc % 32 == 0 and K % 32 == 0 and R * S <= 25
            and B * H * W <= _config.runtime().layers_f16x3_maxpix
            and _planes_fit(B * H * W, max(Cc, K)) and B * H * W * ((max(K, Cc) + 127) // 128) * 512 < 0x7FFFFF00)


#: The general fp16 kernel streams its weight tile once per 64-pixel workgroup and the fp16 weight-gradient kernel re-reads both
#: operands once per tap: at full-resolution feature maps (a million pixels per batch) both are bound by L2 -> LDS traffic and
#: lose to the 128x128-tile fp32-MFMA kernels; `config.layers_f16x3_maxpix` caps the fp16 route's pixel count (sweep: DESIGN.md section 9)


def _wide_kernel(n_out, x_shape):
    """large pixel counts with at most 192 output channels: the 192-column kernel (128-pixel workgroups, one weight stream per
    128 pixels) instead of the general one (64-pixel workgroups x 128-column tiles, built for the 16x16 latents)"""
    B, _, H, W = x_shape
    return n_out <= 192 and B * H * W >= _config.runtime().layers_wide_minpix



def planes_of(t):
    """the fp16 planes copy a producing kernel left next to an activation tensor (same values), if any"""
    return getattr(t, "_stem_planes", None)


class Conv2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, masked, cache, slope=F.LRELU_SLOPE, want_planes=False):
        K, Cc, R, S = weight.shape
        # <=4 input channels: the image (NCHW, fused layout change: g_a.0) or image+quality map (stem_roi.py:529)
        first = (Cc == 3 and F.nhwc_ld(x) is None) or (Cc == 4 and R * S <= 32)
        if not first and x.is_cuda and _layers_f16x3_enabled() and _conv_f16x3_route(weight, stride, pad, masked, x.shape):
            # fp16 route: the input as planes (left by the producer, or split here), kept for the weight gradient instead
            # of the fp32 input; the activation is the kernel's epilogue
            xp = planes_of(x)
            if xp is None or tuple(xp.shape) != tuple(x.shape):
                xp = F.F16Planes.split(x)
            if _wide_kernel(K, x.shape) and xp.dense:
                y, yp = F.conv2d_f16x3_act(xp, cache.get(weight, PACK_F16X2), bias, K, R, S, 1, pad, bool(act), slope, want_planes)
            else:
                y, yp = F.conv2d_f16x3_gen(xp, cache.get(weight, PACK_F16X2_GEN), bias, K, R, S, 1, pad,
                                            epi=F.GEN_EPI_LRELU if act else F.GEN_EPI_BIAS, slope=slope, want_planes=want_planes)
            if yp is not None:
                y._stem_planes = yp
            ctx.cfg = (stride, pad, act, masked, cache, False, tuple(x.shape), slope)
            ctx.params = (weight, bias)
            ctx.fx3 = (xp.q_offset, xp.pix_bytes, xp.byte_offset)
            ctx.save_for_backward(xp.data, weight, y if act else None)
            return y
        ctx.fx3 = None
        if first:
            xin = F.nchw3_to_nhwc4(x) if Cc == 3 else F.dense_nhwc(x).permute(0, 2, 3, 1)
            y = F.conv2d_fwd_c4(xin, cache.get(weight, F.PACK_CONV_FWD_C4), bias, K, R, S, stride, pad)
            xin = xin.permute(0, 3, 1, 2)               # [B,4,H,W] NHWC view for the weight gradient
        else:
            xin = F.to_nhwc(x)
            y = F.conv2d_fwd(xin, cache.get(weight, F.PACK_CONV_FWD, masked), bias, K, R, S, stride, pad,
                             act | (F.CONV_MASKED_A if masked and not masked & 4 else 0), slope=slope)
        if first and act:
            y = F.lrelu_fwd(y, slope)
        ctx.cfg = (stride, pad, act, masked, cache, first, tuple(x.shape), slope)
        ctx.params = (weight, bias)
        ctx.save_for_backward(xin, weight, y if act else None)
        return y

    @staticmethod
    def _backward_fx3(ctx, dy):
        stride, pad, act, masked, cache, first, xshape, slope = ctx.cfg
        xdata, weight, y = ctx.saved_tensors
        K, Cc, R, S = weight.shape
        xp = F.F16Planes(xdata, xshape, *ctx.fx3)
        dy = F.to_nhwc(dy)
        # the gradient as planes, with this layer's leaky-ReLU derivative applied in the splitting pass
        dyp = F.F16Planes.split_dact(dy, y, slope) if act else F.F16Planes.split(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            if _wide_kernel(Cc, xshape):
                dx = F.conv2d_f16x3_act(dyp, cache.get(weight, PACK_F16X2_FLIP), None, Cc, R, S, 1, pad)[0]
            else:
                dx = F.conv2d_f16x3_gen(dyp, cache.get(weight, PACK_F16X2_GEN_FLIP), None, Cc, R, S, 1, pad, epi=F.GEN_EPI_BIAS)[0]
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            need_db = bool(ctx.needs_input_grad[2])
            gw, gb = _flat_grad(ctx.params[0]), (_flat_grad(ctx.params[1]) if need_db else None)
            if gw is not None and (gb is not None or not need_db):
                def run():
                    F.conv2d_wgrad_f16x3_into(xp, dyp, K, R, S, pad, gw, gb if need_db else None, accumulate=True)
                _on_side_stream(run, dyp.data, xdata) if _WGRAD_SIDE["enabled"] else run()
            else:
                dw = torch.zeros((K, Cc, R, S), device=dy.device, dtype=torch.float32)
                db = torch.zeros(K, device=dy.device, dtype=torch.float32) if need_db else None
                F.conv2d_wgrad_f16x3_into(xp, dyp, K, R, S, pad, dw, db, accumulate=True)
        return dx, dw, db, None, None, None, None, None, None, None

    @staticmethod
    def backward(ctx, dy):
        if ctx.fx3 is not None:
            return Conv2dFunction._backward_fx3(ctx, dy)
        stride, pad, act, masked, cache, first, xshape, slope = ctx.cfg
        xin, weight, y = ctx.saved_tensors
        K, Cc, R, S = weight.shape
        dy = F.to_nhwc(dy)
        if F.nhwc_ld(dy) != K:
            dy = F.copy_channels(dy, F.empty_nhwc(*dy.shape, dy.device))
        if act:
            dy = F.lrelu_bwd(y, dy, slope)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = F.conv2d_dgrad(dy, cache.get(weight, F.PACK_CONV_DGRAD, (1 | (masked & 4)) if masked else 0), xshape, K, R, S, stride, pad)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            need_db = bool(ctx.needs_input_grad[2])
            gw, gb = _flat_grad(ctx.params[0]), (_flat_grad(ctx.params[1]) if need_db else None)
            if gw is not None and (gb is not None or not need_db) and not (first and Cc == 3):
                def run():
                    F.conv2d_wgrad(xin, dy, K, R, S, stride, pad, dw_out=gw, db_out=gb, need_db=need_db, accumulate=True)
                _on_side_stream(run, dy, xin) if _WGRAD_SIDE["enabled"] else run()
            else:
                dw, db = F.conv2d_wgrad(xin, dy, K, R, S, stride, pad, need_db=need_db)
                if first and Cc == 3:
                    dw = dw[:, :3].contiguous()
        return dx, dw, db, None, None, None, None, None, None, None


class ConvTranspose2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, opad, act, cache, slope=F.LRELU_SLOPE):
        Cc, K, R, S = weight.shape
        xin = F.to_nhwc(x)
        y = F.deconv2d_fwd(xin, cache.get(weight, F.PACK_DECONV_FWD), bias, K, R, S, stride, pad, opad, act, slope=slope)
        ctx.cfg = (stride, pad, opad, act, cache, tuple(x.shape), slope)
        ctx.params = (weight, bias)
        ctx.save_for_backward(xin, weight, y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, opad, act, cache, xshape, slope = ctx.cfg
        xin, weight, y = ctx.saved_tensors
        Cc, K, R, S = weight.shape
        dy = F.to_nhwc(dy)
        if K == 3 and not act and R * S <= 32:
            return ConvTranspose2dFunction._backward_rgb(ctx, dy)
        if F.nhwc_ld(dy) != K:
            dy = F.copy_channels(dy, F.empty_nhwc(*dy.shape, dy.device))
        if act:
            dy = F.lrelu_bwd(y, dy, slope)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = F.deconv2d_dgrad(dy, cache.get(weight, F.PACK_DECONV_DGRAD), xshape, K, R, S, stride, pad, opad)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            need_db = bool(ctx.needs_input_grad[2])
            gw, gb = _flat_grad(ctx.params[0]), (_flat_grad(ctx.params[1]) if need_db else None)
            if gw is not None and (gb is not None or not need_db):
                def run():
                    F.deconv2d_wgrad(xin, dy, K, R, S, stride, pad, opad, dw_out=gw, db_out=gb, need_db=need_db, accumulate=True)
                _on_side_stream(run, dy, xin) if _WGRAD_SIDE["enabled"] else run()
            else:
                dw, db = F.deconv2d_wgrad(xin, dy, K, R, S, stride, pad, opad, need_db=need_db)
        return dx, dw, db, None, None, None, None, None, None

    @staticmethod
    def _backward_rgb(ctx, dy):
        """The synthesis transform's last layer (-> 3 image channels): the image gradient is padded to 4 channels so that
        the input gradient is the 4-channel-input convolution kernel (it IS Conv2d(weight [C,3,R,S], stride, pad) applied
        to dY) and the weight gradient uses the folded-tap mode, instead of 3-wide operands in 64-wide MFMA tiles."""
        stride, pad, opad, act, cache, xshape, slope = ctx.cfg
        xin, weight, _ = ctx.saved_tensors
        Cc, K, R, S = weight.shape
        B, _, Ho, Wo = dy.shape
        dy4 = torch.zeros((B, Ho, Wo, 4), device=dy.device, dtype=torch.float32)
        F.copy_channels(dy, dy4.permute(0, 3, 1, 2)[:, :3])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = F.conv2d_fwd_c4(dy4, cache.get(weight, F.PACK_CONV_FWD_C4), None, Cc, R, S, stride, pad)
            assert tuple(dx.shape) == tuple(xshape), (dx.shape, xshape)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            need_db = bool(ctx.needs_input_grad[2])
            dw4, db4 = F.deconv2d_wgrad(xin, dy4.permute(0, 3, 1, 2), 4, R, S, stride, pad, opad, need_db=need_db)
            dw = dw4[:, :3].contiguous()
            db = db4[:3].contiguous() if need_db else None
        return dx, dw, db, None, None, None, None, None, None


class GDNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_min):
        xin = F.to_nhwc(x)
        ctx.cfg = (inverse, beta_min)
        ctx.save_for_backward(xin, beta, gamma)
        return F.gdn_fwd(xin, beta, gamma, inverse, beta_min)

    @staticmethod
    def backward(ctx, dy):
        inverse, beta_min = ctx.cfg
        xin, beta, gamma = ctx.saved_tensors
        dy = F.to_nhwc(dy)
        if F.nhwc_ld(dy) % 4 or F.nhwc_ld(xin) % 4:
            raise NotImplementedError("GDN backward needs channel counts that are multiples of 4")
        dx, dbeta, dgamma = F.gdn_bwd(xin, dy, beta.detach().contiguous(), gamma.detach().contiguous(), inverse, beta_min)
        return dx, dbeta, dgamma, None, None


class SFTFunction(torch.autograd.Function):
    """act(x * (1 + gamma) + beta), act = leaky-ReLU(slope) or identity (slope 1): stem_utils.py:41,56-57."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope):
        x, gamma, beta = (F.dense_nhwc(t) for t in (x, gamma, beta))
        out = F.sft_fwd(x, gamma, beta, slope)
        ctx.slope = slope
        ctx.save_for_backward(x, gamma, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, out = ctx.saved_tensors
        dx, dg, db = F.sft_bwd(x, gamma, out, F.dense_nhwc(dout), ctx.slope)
        return dx, dg, db, None


class AvgPoolFunction(torch.autograd.Function):
    """F.adaptive_avg_pool2d(x, (Ho, Wo)) for integer ratios (stem_utils.py:37, stem_roi.py:563)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        ctx.hw = tuple(x.shape[2:])
        return F.avgpool(F.to_nhwc(x), Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        return F.avgpool_bwd(F.to_nhwc(dy), *ctx.hw), None, None


class CatFunction(torch.autograd.Function):
    """torch.cat(xs, dim=1) into one NHWC buffer; the gradient hands back channel-slice views."""

    @staticmethod
    def forward(ctx, *xs):
        B, _, H, W = xs[0].shape
        ctx.widths = [int(t.shape[1]) for t in xs]
        buf = F.empty_nhwc(B, sum(ctx.widths), H, W, xs[0].device)
        c = 0
        for t, n in zip(xs, ctx.widths):
            F.copy_channels(F.to_nhwc(t), buf[:, c:c + n])
            c += n
        return buf

    @staticmethod
    def backward(ctx, dy):
        dy = F.to_nhwc(dy)
        outs, c = [], 0
        for n in ctx.widths:
            outs.append(dy[:, c:c + n])
            c += n
        return tuple(outs)


class AddFunction(torch.autograd.Function):
    """Residual sum x + dx (stem_utils.py:58) on dense NHWC tensors."""

    @staticmethod
    def forward(ctx, a, b):
        return F.add(F.dense_nhwc(a), F.dense_nhwc(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class ToNCHWFunction(torch.autograd.Function):
    """NHWC feature map -> contiguous NCHW tensor (the image handed back to the caller)."""

    @staticmethod
    def forward(ctx, x):
        return F.to_nchw(F.to_nhwc(x))

    @staticmethod
    def backward(ctx, dy):
        return F.to_nhwc(dy)


def cat(xs):
    return CatFunction.apply(*xs)


def to_nchw(x):
    return ToNCHWFunction.apply(x)


def adaptive_avg_pool2d(x, size):
    if tuple(x.shape[2:]) == tuple(size):
        return x
    return AvgPoolFunction.apply(x, int(size[0]), int(size[1]))


# ----------------------------------------------------------------------------- modules
def _pair(v):
    return v if isinstance(v, int) else v[0]


class Conv2d(nn.Module):
    """nn.Conv2d(in, out, kernel_size, stride, padding) with square kernels, bias, dilation 1."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.kernel_size, self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packs = _PackCache()
        self._masked = 0
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))      # nn.Conv2d default; models re-init (priors.py:67-72)
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight[0].numel())
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=F.ACT_NONE, slope=F.LRELU_SLOPE, planes=False):
        """planes=True: the consumer is another fp16-routed convolution -- leave the pre-split copy next to the output"""
        return Conv2dFunction.apply(x, self.weight, self.bias, self.stride, self.padding, act, self._masked, self._packs, slope, planes)

    def f16x3_route(self, x_shape):
        return _layers_f16x3_enabled() and _conv_f16x3_route(self.weight, self.stride, self.padding, self._masked, x_shape)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}"


class ConvTranspose2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.output_padding = _pair(output_padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, self.kernel_size, self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packs = _PackCache()
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight.shape[0] * self.kernel_size ** 2)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=F.ACT_NONE, slope=F.LRELU_SLOPE):
        return ConvTranspose2dFunction.apply(x, self.weight, self.bias, self.stride, self.padding, self.output_padding,
                                             act, self._packs, slope)


class MaskedConv2d(Conv2d):
    """PixelCNN-style masked convolution; like the reference the masked taps of `weight` are zeroed
    *in place* at every forward (the pack kernel does it) while their gradients stay unmasked."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        if mask_type not in ("A", "B"):
            raise ValueError(f'Invalid "mask_type" value "{mask_type}"')
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0
        # pack-kernel mode 2 (zero the masked taps of `weight` in place, as the reference's forward does) | 4 for type B.
        # Type A (the STEM context model, spatiotemporalpriors.py:546,830) additionally lets the kernel skip the dead taps.
        self._masked = 2 | (4 if mask_type == "B" else 0)


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_min = float(beta_min)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        return GDNFunction.apply(x, self.beta, self.gamma, self.inverse, self.beta_min)


class LeakyReLU(nn.LeakyReLU):
    """Placeholder that keeps nn.Sequential indices (`HE.0`, `HE.2`, …) identical to the reference;
    FusedSequential folds it into the preceding convolution's epilogue."""


def _conv_gdn_fused(conv_mod, gdn, x):
    """Conv2d / ConvTranspose2d followed by GDN / IGDN in ONE kernel (inference only: no autograd graph)."""
    K = conv_mod.out_channels
    R = conv_mod.kernel_size
    w, b = conv_mod.weight, conv_mod.bias
    if isinstance(conv_mod, ConvTranspose2d):
        return F.deconv2d_gdn_fwd(F.to_nhwc(x), conv_mod._packs.get(w, F.PACK_DECONV_FWD), b, gdn.beta, gdn.gamma, K, R, R,
                                  conv_mod.stride, conv_mod.padding, conv_mod.output_padding, gdn.inverse, gdn.beta_min)
    if conv_mod.in_channels == 3 and F.nhwc_ld(x) is None:
        ast = conv_mod._packs.get_c4gdn(w, gdn.gamma, K, R) if F.c4gdn_supported(K, R, R, gdn.inverse) else None
        return F.conv2d_fwd_c4_gdn(F.nchw3_to_nhwc4(x), conv_mod._packs.get(w, F.PACK_CONV_FWD_C4), b, gdn.beta, gdn.gamma, K, R, R,
                                   conv_mod.stride, conv_mod.padding, gdn.inverse, gdn.beta_min, astream=ast)
    return F.conv2d_gdn_fwd(F.to_nhwc(x), conv_mod._packs.get(w, F.PACK_CONV_FWD, conv_mod._masked), b, gdn.beta, gdn.gamma, K, R, R,
                            conv_mod.stride, conv_mod.padding, gdn.inverse, gdn.beta_min)


def _f16x3_enabled():
    """fp32-accurate convolutions on the fp16 matrix cores for inference-only chains (csrc/conv_f16x3.hip).  STEM_F16X3=0
    selects the fp32-MFMA kernels everywhere."""
    return _config.runtime().analysis_f16x3


#: fewest output pixels for which the fp16 kernel beats the fp32-MFMA one (64-pixel tiles, no split-K: below ~3/4 of the CUs
#: the split-K fp32 kernel wins; measured on g_a.6 at B=16: 4096 pixels)
_F16X3_MIN_PIXELS = 12288


def _f16x3_eligible(m, in_shape):
    """Can conv `m`, applied to an input of logical shape `in_shape` = (B, C, H, W), run on csrc/conv_f16x3.hip?"""
    if not (type(m) is Conv2d and not m._masked and m.in_channels % 32 == 0 and m.out_channels <= 192
            and m.kernel_size * m.kernel_size <= 25 and m.weight.is_cuda):
        return False
    return _f16x3_shape_ok(m, in_shape)


def _f16x3_shape_ok(m, in_shape):
    B, _, H, W = in_shape
    Ho, Wo = F.conv_out_hw(H, W, m.kernel_size, m.kernel_size, m.stride, m.padding)
    if not (_planes_fit(B * H * W, m.in_channels) and _planes_fit(B * Ho * Wo, m.out_channels)):
        return False          # the kernels address their operands through 2 GiB buffer views: such a batch stays on the fp32 kernels
    return B * Ho * Wo >= _F16X3_MIN_PIXELS


def _planes_fit(npix, channels):
    """does a planes tensor of this size (4 bytes per element) stay inside one 2 GiB buffer view?"""
    return npix * ((channels + 31) // 32) * F.PLANES_SLAB_BYTES < 0x7FFFFF00


def _f16x3_gen_eligible(m, follows_gdn, in_shape=None):
    """Small layers that end a planes chain (the last convolution of the analysis transform: 4096 output pixels at the bench
    size) go to the general split-K kernel, which has no fused GDN."""
    if in_shape is not None and not _planes_fit(in_shape[0] * in_shape[2] * in_shape[3], m.in_channels):
        return False
    return (type(m) is Conv2d and not m._masked and not follows_gdn and m.in_channels % 32 == 0 and m.out_channels % 4 == 0
            and m.kernel_size * m.kernel_size <= 25 and m.weight.is_cuda)


def _conv_out_shape(m, in_shape):
    B, _, H, W = in_shape
    Ho, Wo = F.conv_out_hw(H, W, m.kernel_size, m.kernel_size, m.stride, m.padding)
    return (B, m.out_channels, Ho, Wo)


class FusedSequential(nn.Sequential):
    #: optional (index, list) pair -- or a dict {index: list} -- set by bench.py: HIP events are recorded on the launching stream
    #: around the kernel(s) of child `index`
    probe = None

    def _timed(self, i, fn):
        sink = None
        if isinstance(self.probe, dict):
            sink = self.probe.get(i)
        elif self.probe is not None and self.probe[0] == i:
            sink = self.probe[1]
        if sink is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        sink.append((e0, e1))
        return out

    def forward(self, x):
        mods = list(self)
        nograd = not torch.is_grad_enabled()
        fx3 = nograd and _f16x3_enabled()
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if fx3 and type(m) is Conv2d:
                # frozen / inference chain of convolutions (the analysis transform): operands pre-split into fp16 planes, the
                # following GDN fused, the output written as planes again when the next convolution takes them.  A chain
                # starts where the next convolution is eligible too -- either at the 3-channel first layer, whose fp32 kernel
                # then writes planes, or with a split pass over an fp32 tensor -- and runs until one is not eligible.
                gdn = nxt if isinstance(nxt, GDN) and not nxt.inverse else None
                j = i + (2 if gdn is not None else 1)
                K, R = m.out_channels, m.kernel_size
                out_shape = _conv_out_shape(m, x.shape)
                chain = False
                if K % 32 == 0 and j < len(mods) and mods[j].__class__ is Conv2d and mods[j].in_channels == K:
                    nxt_gdn = j + 1 < len(mods) and isinstance(mods[j + 1], GDN)
                    chain = _f16x3_eligible(mods[j], out_shape) or _f16x3_gen_eligible(mods[j], nxt_gdn, out_shape)
                if (chain and gdn is not None and m.in_channels == 3 and not isinstance(x, F.F16Planes) and F.nhwc_ld(x) is None
                        and K <= 192):
                    wp = m._packs.get(m.weight, F.PACK_CONV_FWD_C4)
                    ast = m._packs.get_c4gdn(m.weight, gdn.gamma, K, R) if F.c4gdn_supported(K, R, R) else None
                    x = self._timed(i, lambda: F.conv2d_fwd_c4_gdn_planes(F.nchw3_to_nhwc4(x), wp, m.bias, gdn.beta, gdn.gamma, K, R, R,
                                                                          m.stride, m.padding, gdn.beta_min, astream=ast))
                    i = j
                    continue
                if _f16x3_eligible(m, x.shape) and (isinstance(x, F.F16Planes) or (chain and x.is_cuda)):
                    xin = x if isinstance(x, F.F16Planes) else F.F16Planes.split(x)
                    wp = m._packs.get(m.weight, PACK_F16X2)
                    gp = m._packs.get(gdn.gamma, PACK_GDN_GAMMA) if gdn is not None else None     # kept in the convolution's cache
                    x = self._timed(i, lambda: F.conv2d_f16x3_fwd(xin, wp, m.bias, K, R, R, m.stride, m.padding,
                                                                   gdn.beta if gdn is not None else None,
                                                                   gdn.gamma if gdn is not None else None,
                                                                   gdn.beta_min if gdn is not None else 1e-6, planes_out=chain, gp=gp))
                    i = j
                    continue
                if isinstance(x, F.F16Planes) and _f16x3_gen_eligible(m, gdn is not None):
                    wp = m._packs.get(m.weight, PACK_F16X2_GEN)
                    x = self._timed(i, lambda: F.conv2d_f16x3_gen(x, wp, m.bias, K, R, R, m.stride, m.padding, want_fp32=not chain,
                                                                   want_planes=chain)[1 if chain else 0])
                    i = j
                    continue
            if (isinstance(m, (Conv2d, ConvTranspose2d)) and isinstance(nxt, GDN) and nograd and m.out_channels <= 192
                    and m.out_channels % 4 == 0 and m.in_channels % 4 in (0, 3)):
                x = self._timed(i, lambda: _conv_gdn_fused(m, nxt, x))
                i += 2
                continue
            if isinstance(m, (Conv2d, ConvTranspose2d)) and isinstance(nxt, (nn.LeakyReLU, nn.ReLU)):
                # LeakyReLU(slope) or ReLU (= slope 0) folded into the conv epilogue
                slope = float(nxt.negative_slope) if isinstance(nxt, nn.LeakyReLU) else 0.0
                if type(m) is Conv2d and x.is_cuda:
                    # hand planes to the next convolution when both run on the fp16 kernels (a conv -> LeakyReLU -> conv chain)
                    after = mods[i + 2] if i + 2 < len(mods) else None
                    out_shape = _conv_out_shape(m, x.shape)
                    hand = type(after) is Conv2d and m.f16x3_route(x.shape) and after.f16x3_route(out_shape)
                    x = self._timed(i, lambda: m(x, act=F.ACT_LRELU, slope=slope, planes=hand))
                else:
                    x = self._timed(i, lambda: m(x, act=F.ACT_LRELU, slope=slope))
                i += 2
            else:
                x = m(x)
                i += 1
        return x


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    """compressai/models/utils.py:112-120"""
    return Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    """compressai/models/utils.py:122-130"""
    return ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                           output_padding=stride - 1, padding=kernel_size // 2)
