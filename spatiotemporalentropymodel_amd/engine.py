"""StemEngine: the whole-model forward / backward schedule of a STEM entropy model as a flat sequence
of C-ABI kernel launches (no per-layer autograd nodes, no concatenations, no activation passes).

Mirrors, for all five model variants, the dataflow of
  compressai/models/spatiotemporalpriors.py:70-83, 176-194, 292-311, 561-585, 845-868 (forward)
and what torch autograd derives from it for `EMLoss` (utils.py:18-27) with y_cur / y_conditioned
detached (stem/trainSTEM.py:208).

Memory plan (NHWC fp32): `he_in` = [y_cur | y_cond] and `epm_in` = [tp | hp | ctx]
are single buffers whose channel slices are written in place by their producers; the EPM output
`gp` is read as (scales | means) slices by the Gaussian kernel.  LeakyReLU is fused into the
producing convolution (forward) and into the consuming dgrad's epilogue (backward).
"""
from __future__ import annotations

import torch

from . import _lib
from . import config as _config
from . import functional as F
from . import layers as _layers

class _Switch:
    """A StemEngine switch backed by a field of config.StemRuntimeConfig, read when it is LOOKED UP: `config.override(...)` takes
    effect at once, a changed `STEM_*` variable when the next engine is built (StemEngine.__init__ parses the environment again;
    a look-up itself reads the parsed object -- 0.1 us instead of 15 -- because the schedule asks ~50 times per step).  The route
    switches are looked up when an engine is built, the scheduling ones at every step.  A plain value assigned on the class or on
    an instance (`StemEngine.use_fx3 = False`, tests' monkeypatch) wins."""

    def __init__(self, field):
        self.field = field

    def __get__(self, obj, owner=None):
        return getattr(_config._RUNTIME or _config.runtime(), self.field)


class _RangePlanes:
    """the planes of a tensor that was produced range by range (one F16Planes per channel range, each with its own scale record):
    `channels(c0, c1)` hands out the part that covers exactly that range"""

    def __init__(self, parts):
        self.parts = dict(parts)            # (c0, c1) -> F16Planes

    def channels(self, c0, c1):
        return self.parts[(c0, c1)]


class _Layer:
    """One convolution of the schedule: parameters, persistent packed-weight buffers and wgrad slabs."""

    def __init__(self, mod, kind, eng):
        self.mod, self.kind, self.eng = mod, kind, eng      # kind: "conv" | "deconv"
        self.R = mod.kernel_size
        self.stride, self.pad = mod.stride, mod.padding
        self.opad = getattr(mod, "output_padding", 0)
        self.K, self.C = mod.out_channels, mod.in_channels
        self.masked = getattr(mod, "_masked", 0)
        self.wp_fwd = self.wp_dgrad = None
        self.need_dgrad = True
        # forward / input-gradient on the fp16 matrix cores (csrc/conv_f16x3.hip, general variant): stride-1 convolutions whose
        # contraction channels are multiples of 32; decided once by the engine (StemEngine._select_fx3)
        self.fx3 = False
        self.wg3 = False
        self.taps = 0             # masked convolution on the fp16 kernel: the number of live taps (a prefix of the row-major order)
        # fx3s: the STRIDED-CONVOLUTION face of a stride-2 layer on the general fp16 kernel -- the forward of a strided Conv2d
        # (HE.2 / HE.4), the input gradient of a ConvTranspose2d (HD.0 / HD.2: a strided convolution of dy with the stored weight
        # [C][K][R][S] read as a Conv2d weight with C outputs); the transposed face stays on igemm.hip's sub-pixel phases
        self.fx3s = False
        # fx3t: ... and the TRANSPOSED face too (forward of a ConvTranspose2d, input gradient of a strided Conv2d) as one launch of the
        # same kernel over its four sub-pixel phases (F.tconv2d_f16x3), and the layer's weight gradient on the per-tap fp16 kernel
        # with a strided gather (F.conv2d_wgrad_f16x3_strided): no fp32-MFMA launch is left in the layer
        self.fx3t = False
        self.wp6_fwd = self.wp6_dgrad = None
        self._slabs = {}
        self.pending = None       # (dwp, splits) of the last wgrad, consumed by StemEngine.unpack_all
        self.lane = 0             # which weight-gradient stream this layer's wgrad / unpack runs on (StemEngine.side_stream)

    def fx3_eligible(self):
        return (self.kind == "conv" and self.stride == 1 and not self.masked and self.C % 32 == 0 and self.K % 32 == 0
                and self.R % 2 == 1 and self.R * self.R <= 25 and self.pad == self.R // 2)     # odd windows: 'same' shapes both ways

    def fx3_masked_eligible(self):
        """the context model's masked convolution, forward only (its input is data + noise: no input gradient): the general
        kernel runs over the live taps alone -- 12 of 25 for the 5x5 type-A mask (layers.py:21-47)"""
        return (self.kind == "conv" and self.stride == 1 and bool(self.masked) and not self.need_dgrad and self.C % 32 == 0
                and self.K % 4 == 0 and self.R % 2 == 1 and self.R * self.R <= 25 and self.pad == self.R // 2)

    def fx3s_eligible(self):
        n_out, n_red = (self.K, self.C) if self.kind == "conv" else (self.C, self.K)       # outputs / contraction channels of that face
        return (self.stride == 2 and not self.masked and n_red % 32 == 0 and n_out % 4 == 0 and self.R * self.R <= 25
                and (self.kind == "conv" or self.need_dgrad))

    def fx3t_eligible(self):
        """both faces and the weight gradient on the fp16 kernels: 5x5 / 3x3, stride 2, padding R // 2, the ConvTranspose2d with
        output_padding 1 (fine grid = twice the coarse grid), channel counts in whole 32-channel slabs"""
        return (self.fx3s and self.R % 2 == 1 and self.R >= 3 and self.pad == self.R // 2 and self.C % 32 == 0 and self.K % 32 == 0
                and (self.kind == "conv" or self.opad == 1))

    def alloc_packs(self, device):
        if self.fx3s and self.fx3t:
            conv = self.kind == "conv"
            # (zeros: the pair pack never writes the padding rows of a 128-row tile)
            strided = torch.zeros(F.f16x2_gen_weight_bytes(*((self.K, self.C) if conv else (self.C, self.K)), self.R, self.R), device=device, dtype=torch.uint8)
            phases = torch.zeros(F.f16x2_gen_weight_bytes(*((self.C, self.K) if conv else (self.K, self.C)), self.R, self.R), device=device, dtype=torch.uint8)
            self.wp_fwd = torch.empty(0, device=device)              # never read: marks the layer as allocated
            self.wp_dgrad = None
            if conv:
                self.wp6_fwd, self.wp6_dgrad = strided, (phases if self.need_dgrad else None)
            else:
                self.wp6_fwd, self.wp6_dgrad = phases, (strided if self.need_dgrad else None)
            return
        if self.fx3s:
            n = self.K * self.C * self.R * self.R
            nb = F.f16x2_gen_weight_bytes(*((self.K, self.C) if self.kind == "conv" else (self.C, self.K)), self.R, self.R)
            img = torch.empty(nb, device=device, dtype=torch.uint8)
            if self.kind == "conv":      # forward on the fp16 kernel, input gradient on igemm.hip
                self.wp6_fwd, self.wp6_dgrad = img, None
                self.wp_fwd = torch.empty(0, device=device)          # never read: marks the layer as allocated
                self.wp_dgrad = torch.empty(n, device=device, dtype=torch.float32) if self.need_dgrad else None
            else:                        # forward on igemm.hip, input gradient on the fp16 kernel
                self.wp6_fwd, self.wp6_dgrad = None, img
                self.wp_fwd = torch.empty(n, device=device, dtype=torch.float32)
                self.wp_dgrad = None
            return
        if self.fx3:
            self.wp6_fwd = torch.zeros(F.f16x2_gen_weight_bytes(self.K, self.C, self.R, self.R), device=device, dtype=torch.uint8)
            self.wp6_dgrad = torch.zeros(F.f16x2_gen_weight_bytes(self.C, self.K, self.R, self.R), device=device,
                                         dtype=torch.uint8) if self.need_dgrad else None
            return
        n = self.K * self.C * self.R * self.R
        self.wp_fwd = torch.empty(n, device=device, dtype=torch.float32)
        self.wp_dgrad = torch.empty(n, device=device, dtype=torch.float32) if self.need_dgrad else None

    def role_descs(self, role):
        """(fp32 descriptors, fp16 descriptors) of this layer's packed copies for role 0 (forward) / 1 (input gradient)"""
        if self.fx3s and self.fx3t:
            w, conv = self.mod.weight, self.kind == "conv"
            # strided face: the torch weight read as a Conv2d weight [outputs][contraction]; transposed face: read as w[c][n] (flip = 2)
            if role == 0:
                return [], [_lib.F16PackDesc(w.data_ptr(), self.wp6_fwd.data_ptr(), self.K, self.C, self.R, self.R, 0 if conv else 2, 0)]
            if not self.need_dgrad:
                return [], []
            return [], [_lib.F16PackDesc(w.data_ptr(), self.wp6_dgrad.data_ptr(), self.C, self.K, self.R, self.R, 2 if conv else 0, 0)]
        if self.fx3s:
            w, conv = self.mod.weight, self.kind == "conv"
            if role == 0:
                return ([], [_lib.F16PackDesc(w.data_ptr(), self.wp6_fwd.data_ptr(), self.K, self.C, self.R, self.R, 0, 0)]) if conv \
                    else ([self._desc32(0)], [])
            if not self.need_dgrad:
                return [], []
            return ([self._desc32(1)], []) if conv \
                else ([], [_lib.F16PackDesc(w.data_ptr(), self.wp6_dgrad.data_ptr(), self.C, self.K, self.R, self.R, 0, 0)])
        return ([], self.pack_descs6()[role:role + 1]) if self.fx3 else (self.pack_descs()[role:role + 1], [])

    def pair_desc(self):
        """both fp16 images of this layer as ONE descriptor of the pair pack (F.pack_weights_f16x2_pair_multi), or None when the
        layer keeps an fp32 copy of a role"""
        a, b = self.role_descs(0), self.role_descs(1)
        if a[0] or b[0] or not a[1]:
            return None
        w = self.mod.weight
        A, Bd = int(w.shape[0]), int(w.shape[1])
        roles = []
        for d in (a[1][0], b[1][0] if b[1] else None):
            if d is None:
                roles += [None, 0, 0]
                continue
            rows_b = int(d.flip != 0)
            assert (d.N, d.C) == ((Bd, A) if rows_b else (A, Bd)), "pair pack: a role image must be a transpose of the tensor itself"
            roles += [d.wp, rows_b | (int(d.flip) << 1), int(d.taps)]
        return _lib.F16PairDesc(w.data_ptr(), A, Bd, self.R, self.R, roles[0], roles[1], roles[2], roles[3], roles[4], roles[5], None, 0, 0)

    def pack_descs6(self):
        w = self.mod.weight
        out = [_lib.F16PackDesc(w.data_ptr(), self.wp6_fwd.data_ptr(), self.K, self.C, self.R, self.R, 0, self.taps)]
        if self.need_dgrad:       # the input-gradient of a stride-1 convolution is a convolution with the mirrored, transposed weight
            out.append(_lib.F16PackDesc(w.data_ptr(), self.wp6_dgrad.data_ptr(), self.C, self.K, self.R, self.R, 1, 0))
        return out

    def fwd6(self, xp, act=F.ACT_NONE, out=None, planes=False):
        """-> (fp32 output, planes output or None); `xp` a F16Planes (possibly a channel view)"""
        return F.conv2d_f16x3_gen(xp, self.wp6_fwd, self.mod.bias, self.K, self.R, self.R, self.stride, self.pad,
                                   epi=F.GEN_EPI_LRELU if act == F.ACT_LRELU else F.GEN_EPI_BIAS, out=out, want_planes=planes, taps=self.taps)

    def dgrad6(self, dyp, xact=None, planes=False):
        """-> (dx fp32, dx planes or None); xact: the activated input of this layer (leaky-ReLU derivative folded in)"""
        return F.conv2d_f16x3_gen(dyp, self.wp6_dgrad, None, self.C, self.R, self.R, 1, self.pad,
                                   epi=F.GEN_EPI_DACT if xact is not None else F.GEN_EPI_BIAS, z=xact, want_planes=planes)

    def dgrad6s(self, dyp, xact=None, planes=False):
        """input gradient of a ConvTranspose2d = the strided convolution of dy with the stored weight (fx3s)"""
        return F.conv2d_f16x3_gen(dyp, self.wp6_dgrad, None, self.C, self.R, self.R, self.stride, self.pad,
                                   epi=F.GEN_EPI_DACT if xact is not None else F.GEN_EPI_BIAS, z=xact, want_planes=planes)

    def fwd6t(self, xp, act=F.ACT_NONE, planes=False):
        """forward of a ConvTranspose2d on the fp16 kernel (four sub-pixel phases, one launch) -> (fp32, planes or None)"""
        return F.tconv2d_f16x3(xp, self.wp6_fwd, self.mod.bias, self.K, self.R,
                               epi=F.GEN_EPI_LRELU if act == F.ACT_LRELU else F.GEN_EPI_BIAS, want_planes=planes)

    def dgrad6t(self, dyp, xact=None, planes=False):
        """input gradient of a strided Conv2d = the transposed face, leaky-ReLU derivative of the layer's input folded in"""
        return F.tconv2d_f16x3(dyp, self.wp6_dgrad, None, self.C, self.R, epi=F.GEN_EPI_DACT if xact is not None else F.GEN_EPI_BIAS,
                               z=xact, want_planes=planes, fine_hw=tuple(xact.shape[2:]) if xact is not None else None)

    def wgrad_t(self, fine_p, coarse_p, dy):
        """weight + bias gradient of a stride-2 layer from planes: `fine_p` the fine-grid operand (a Conv2d's input, a
        ConvTranspose2d's output gradient), `coarse_p` the coarse-grid one; dy: the fp32 output gradient (bias column sums of
        the transposed layer).  On the weight-gradient stream like every other weight gradient."""
        side = self.eng.side_stream(dy.device, self.lane)
        if side is not None:
            F.stream_wait(side, F.cur_stream(dy.device))
            with F.on_stream(side):
                self._wgrad_t(fine_p, coarse_p, dy)
            for t in (fine_p.data, coarse_p.data, dy):
                t.record_stream(side)
        else:
            self._wgrad_t(fine_p, coarse_p, dy)

    def _wgrad_t(self, fine_p, coarse_p, dy):
        conv = self.kind == "conv"
        Kk = self.K if conv else self.C                  # rows of the slabs = channels of the coarse operand
        key = ("fp16t",) + tuple(fine_p.shape)
        if key not in self._slabs:
            splits, elems = F.wgrad_f16x3_strided_plan(fine_p.shape, Kk, self.R, self.R, self.stride, self.pad)
            self._slabs[key] = (torch.empty(elems, device=dy.device, dtype=torch.float32), splits,
                                torch.empty(splits * Kk, device=dy.device, dtype=torch.float32))
        dwp, splits, bpart = self._slabs[key]
        gb = _grad_of(self.mod.bias) if self.mod.bias is not None else None
        F.conv2d_wgrad_f16x3_strided(fine_p, coarse_p, Kk, self.R, self.R, self.stride, self.pad, dwp, splits,
                                     bias_part=bpart if (conv and gb is not None) else None)
        self.pending_bias = None
        if gb is not None and conv:                      # column sums of dy came out of the kernel: second stage
            desc = _lib.BiasFinalDesc(bpart.data_ptr(), gb.data_ptr(), self.K, splits, int(self.eng.accumulate_grads), 0)
            if self.eng.defer_bias_final:
                self.pending_bias = desc
            else:
                F.bias_grad_final_multi([desc])
        elif gb is not None:                             # the transposed layer's bias sees the FINE tensor: its own column sums
            F.bias_grad(dy, gb, accumulate=self.eng.accumulate_grads)
        self.pending = (dwp, splits)

    def wg3_eligible(self):
        """weight gradient on csrc/wgrad_f16x3.hip: stride-1 convolutions (all taps are produced, as autograd does for the
        masked context convolution too)"""
        return (self.kind == "conv" and self.stride == 1 and self.C % 32 == 0 and self.K % 32 == 0 and self.R % 2 == 1
                and self.R * self.R <= 25 and self.pad == self.R // 2)

    def wgrad_any(self, x, dy, xp=None, dyp=None):
        """weight + bias gradient from the planes operands when this layer runs its weight gradient on the fp16 kernel and both
        are at hand, else from the fp32 tensors"""
        if not (self.wg3 and xp is not None and dyp is not None):
            return self.wgrad(x, dy)
        side = self.eng.side_stream(dy.device, self.lane)
        if side is not None:
            F.stream_wait(side, F.cur_stream(dy.device))
            with F.on_stream(side):
                self._wgrad3(xp, dyp, dy)
            for t in (xp.data, dyp.data, dy):
                t.record_stream(side)
        else:
            self._wgrad3(xp, dyp, dy)

    def _wgrad3(self, xp, dyp, dy):
        key = ("fp16",) + tuple(xp.shape)
        if key not in self._slabs:
            splits, elems = F.wgrad_f16x3_plan(xp.shape, self.K, self.R, self.R, self.pad)
            self._slabs[key] = (torch.empty(elems, device=dy.device, dtype=torch.float32), splits,
                                torch.empty(splits * self.K, device=dy.device, dtype=torch.float32))
        dwp, splits, bpart = self._slabs[key]
        gb = _grad_of(self.mod.bias) if self.mod.bias is not None else None
        self.pending_bias = F.conv2d_wgrad_f16x3(xp, dyp, self.K, self.R, self.R, self.pad, dwp, splits, db=gb, bias_part=bpart,
                                                 accumulate_db=self.eng.accumulate_grads, defer_bias=self.eng.defer_bias_final)
        self.pending = (dwp, splits)

    def _desc32(self, role):
        w = self.mod.weight
        conv = self.kind == "conv"
        if role == 0:
            return _lib.PackDesc(w.data_ptr(), self.wp_fwd.data_ptr(), self.K, self.C, self.R, self.R,
                                 F.PACK_CONV_FWD if conv else F.PACK_DECONV_FWD, self.masked)
        return _lib.PackDesc(w.data_ptr(), self.wp_dgrad.data_ptr(), self.K, self.C, self.R, self.R,
                             F.PACK_CONV_DGRAD if conv else F.PACK_DECONV_DGRAD, (1 | (self.masked & 4)) if self.masked else 0)

    def pack_descs(self):
        return [self._desc32(0)] + ([self._desc32(1)] if self.need_dgrad else [])

    def fwd(self, x, act=F.ACT_NONE, out=None):
        self.eng.ensure_packed()
        self.eng._wait_fwd_rest()
        if self.fx3 or (self.fx3s and self.kind == "conv"):          # callers outside the training schedule (codec.py) hand over fp32 tensors
            return self.fwd6(F.F16Planes.split(x), act, out=out)[0]
        if self.fx3t and self.kind == "deconv":
            y = self.fwd6t(F.F16Planes.split(x), act)[0]
            return y if out is None else F.copy_channels(y, out)
        m = self.mod
        self.eng._wait_fwd32_packs()
        if self.kind == "conv":
            if self.masked and not self.masked & 4:
                act |= F.CONV_MASKED_A            # the masked taps (type A) are zeros: skip them
            return F.conv2d_fwd(x, self.wp_fwd, m.bias, self.K, self.R, self.R, self.stride, self.pad, act, out=out)
        return F.deconv2d_fwd(x, self.wp_fwd, m.bias, self.K, self.R, self.R, self.stride, self.pad, self.opad, act, out=out)

    def dgrad(self, dy, x_shape, xact=None):
        if self.kind == "conv":
            return F.conv2d_dgrad(dy, self.wp_dgrad, x_shape, self.K, self.R, self.R, self.stride, self.pad, xact=xact)
        return F.deconv2d_dgrad(dy, self.wp_dgrad, x_shape, self.K, self.R, self.R, self.stride, self.pad, self.opad, xact=xact)

    def wgrad(self, x, dy):
        """packed slabs now, bias gradient straight into .grad; StemEngine.unpack_all() turns every layer's
        slabs into .grad tensors with one launch.  Runs on the engine's weight-gradient stream (nothing on the dgrad
        chain consumes it), ordered after everything the compute stream has queued so far."""
        side = self.eng.side_stream(x.device, self.lane)
        if side is not None:
            F.stream_wait(side, F.cur_stream(x.device))
            with F.on_stream(side):
                self._wgrad(x, dy)
            x.record_stream(side)
            dy.record_stream(side)
        else:
            self._wgrad(x, dy)

    def _wgrad(self, x, dy):
        m = self.mod
        gb = _grad_of(m.bias)
        deconv = self.kind == "deconv"
        key = tuple(x.shape)
        key = key + (F.nhwc_ld(x), F.nhwc_ld(dy))          # the gather table bakes in the pitch of the gathered tensor
        fresh = key not in self._slabs
        if fresh:
            splits, elems = F.wgrad_plan(x.shape, self.K, self.R, self.R, self.stride, self.pad, deconv=deconv)
            self._slabs[key] = (torch.empty(elems, device=x.device, dtype=torch.float32), splits)
        dwp, splits = self._slabs[key]
        defer = self.eng.defer_bias_final and gb is not None
        if deconv:
            _, b = F.deconv2d_wgrad(x, dy, self.K, self.R, self.R, self.stride, self.pad, self.opad, db_out=gb, dwp=dwp, unpack=False,
                                    table_valid=not fresh, accumulate_db=self.eng.accumulate_grads, defer_bias=defer)
        else:
            _, b = F.conv2d_wgrad(x, dy, self.K, self.R, self.R, self.stride, self.pad, db_out=gb, dwp=dwp, unpack=False,
                                  table_valid=not fresh, accumulate_db=self.eng.accumulate_grads, defer_bias=defer)
        self.pending_bias = b if defer else None          # the second stage joins the module group's launch (_group_ready_on_stream)
        self.pending = (dwp, splits)

    def unpack_desc(self):
        dwp, splits = self.pending
        self.pending = None
        return _lib.UnpackDesc(dwp.data_ptr(), _grad_of(self.mod.weight).data_ptr(), self.K, self.C, self.R, self.R, splits,
                               (F.UNPACK_DECONV if self.kind == "deconv" else 0) | (F.UNPACK_ACCUMULATE if self.eng.accumulate_grads else 0))


def _attach_block_maxima(descs, maxima, flat):
    """point every fp16 pack descriptor whose weight lies inside the flat parameter buffer at the optimiser's chunk maxima"""
    ch = F.adam_chunk()
    base, n = flat.data_ptr(), flat.numel()
    for d in descs:
        off = (d.w - base) // 4
        numel = d.N * d.C * d.R * d.S
        if (d.w - base) % 4 == 0 and 0 <= off and off + numel <= n:
            d.bmax, d.b0 = maxima.data_ptr(), off // ch
            d.nb = (off + numel - 1) // ch - d.b0 + 1


def _qp(rec):
    """address of a scale record tensor, or None"""
    return None if rec is None else rec.data_ptr()


def _grad_of(p):
    if p.grad is None:
        owner = getattr(p, "_flat_grad_view", None)
        # a fresh slot starts at zero: every producer below ADDS into it (autograd's `.grad +=`)
        p.grad = owner if owner is not None else torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class StemEngine:
    def __init__(self, model, has_tpm: bool, has_spm: bool, residual: bool):
        _config.runtime()                   # parse the environment now if a STEM_* variable changed (the switches read the parsed object)
        self.m = model
        self.has_tpm, self.has_spm, self.residual = has_tpm, has_spm, residual
        L = lambda mod: _Layer(mod, "conv", self)
        D = lambda mod: _Layer(mod, "deconv", self)
        self.HE = [L(model.HE[0]), L(model.HE[2]), L(model.HE[4])]
        self.HD = [D(model.HD[0]), D(model.HD[2]), L(model.HD[4])]
        self.TPM = [L(model.TPM[0]), L(model.TPM[2]), L(model.TPM[4])] if has_tpm else None
        self.CTX = L(model.context_prediction) if has_spm else None
        self.EPM = [L(model.EPM[0]), L(model.EPM[2]), L(model.EPM[4])]
        self.nprior = 1 + int(has_tpm) + int(has_spm)
        self.layers = self.HE + self.HD + (self.TPM or []) + ([self.CTX] if has_spm else []) + self.EPM
        # no input gradient is ever needed for the first layer of a chain fed by (detached) data
        for first in [self.HE[0]] + ([self.TPM[0]] if has_tpm else []) + ([self.CTX] if has_spm else []):
            first.need_dgrad = False
        self._pack_key = None
        self._pack_descs = None
        self._side = {}
        self._checked = False
        self._dgrad_pack_event = None
        self._fwd32_pack_event = None
        self._fwd_rest_event = None
        #: backward ADDS into .grad (autograd's semantics; several backward passes between two zero_grad() calls accumulate).  An explicit
        #: schedule that produces every gradient exactly once per step sets it False for its backward: the producers then OVERWRITE
        #: (no clearing pass, no read of the old value: 144 MB less HBM traffic per P-frame step of the big model)
        self.accumulate_grads = True
        self._events = {}
        self._select_fx3()

    def _select_fx3(self):
        """Which layers run on the fp16 kernels.  HE.0 and HD.4 are routed one by one; the TPM and EPM chains hand planes from
        layer to layer (and the EPM input gradient is read through channel views at multiples of P = 2 * Cin), so each chain is
        routed as a whole: fp16 only if EVERY layer of it is eligible (channel counts multiples of 32) and, for the EPM, the
        views are 32-aligned -- otherwise the whole chain stays on the fp32-MFMA kernels, which only need C % 4 == 0."""
        for l in self.layers:
            l.fx3 = self.use_fx3 and l.fx3_eligible()
            l.taps = 0
            l.fx3s = False
        if self.use_fx3 and self.use_fx3s:
            # the hyper path's strided faces; each needs its neighbour's planes (HE.0 -> HE.2 -> HE.4, HD.4 -> HD.2 -> HD.0)
            if self.HE[0].fx3 and self.HE[1].fx3s_eligible():
                self.HE[1].fx3s = True
                self.HE[2].fx3s = self.HE[2].fx3s_eligible() and self.HE[1].K % 32 == 0
            if self.HD[2].fx3 and self.HD[1].fx3s_eligible():
                self.HD[1].fx3s = True
                self.HD[0].fx3s = self.HD[0].fx3s_eligible() and self.HD[1].C % 32 == 0
        if self.use_fx3t and self.use_wg3:
            for l in self.HE[1:] + self.HD[:2]:
                l.fx3t = l.fx3t_eligible()
            # planes travel down each chain: all of a chain's stride-2 layers or none
            if not (self.HE[1].fx3t and self.HE[2].fx3t and self.HE[0].fx3):
                self.HE[1].fx3t = self.HE[2].fx3t = False
            if not (self.HD[0].fx3t and self.HD[1].fx3t and self.HD[2].fx3):
                self.HD[0].fx3t = self.HD[1].fx3t = False
        if self.has_spm and self.use_fx3 and self.use_ctx3 and self.CTX.fx3_masked_eligible():
            self.CTX.fx3 = True
            self.CTX.taps = F.masked_live_taps(self.CTX.R, self.CTX.R, "B" if self.CTX.masked & 4 else "A")
        P = self.HE[0].C                                # HE.0 reads cat(y_cur, y_cond): its C is 2 * Cin = P
        for group, need_aligned in ((self.TPM, False), (self.EPM, True)):
            if group and not (all(l.fx3 for l in group) and (not need_aligned or P % 32 == 0)):
                for l in group:
                    l.fx3 = False
        for l in self.layers:
            # weight gradients take whatever planes the forward / input-gradient route left behind (wgrad_any falls back to the
            # fp32 kernel when there are none), so they follow the layer's own eligibility
            l.wg3 = self.use_fx3 and self.use_wg3 and l.wg3_eligible()

    #: forward and input-gradient of the stride-1 layers (TPM, HE.0, HD.4, EPM) on the fp16 matrix cores: three fp16 products per fp32 product on operands split into two scaled fp16 planes, ~2^-21 relative per product (tests: 1e-4 gates; measured 0.4-1.6e-6 of max per layer against fp64)
    #: (three fp16 MFMAs per fp32 product, csrc/conv_f16x3.hip); STEM_ENGINE_F16X3=0 keeps every layer on the fp32-MFMA kernels
    use_fx3 = _Switch("engine_f16x3")
    #: the entropy glue (prologue, Gaussian backward) records the maxima of the fp32 tensors it writes, so that their fp16 splits
    #: skip the maximum pass (four launches per P-frame step); STEM_ENGINE_RECORDS=0: every split measures its input itself
    use_records = _Switch("engine_records")
    #: the strided-convolution faces of the hyper path's stride-2 layers (HE.2 / HE.4 forward, HD.2 / HD.0 input gradient) on the
    #: general fp16 kernel; STEM_ENGINE_STRIDED_F16X3=0: igemm.hip
    use_fx3s = _Switch("engine_strided_f16x3")
    #: ... and their TRANSPOSED faces (HD.0 / HD.2 forward, HE.2 / HE.4 input gradient: one launch over the four sub-pixel phases)
    #: plus the four layers' weight gradients (per-tap kernel, strided gather); STEM_ENGINE_TRANSPOSED_F16X3=0: igemm.hip / wgrad.hip
    use_fx3t = _Switch("engine_transposed_f16x3")
    #: the masked context convolution's forward on the same kernel over its live taps; STEM_ENGINE_CTX_F16X3=0: igemm.hip
    use_ctx3 = _Switch("engine_ctx_f16x3")
    #: ... and their weight gradients (csrc/wgrad_f16x3.hip); STEM_ENGINE_WGRAD_F16X3=0 keeps those on wgrad.hip
    use_wg3 = _Switch("engine_wgrad_f16x3")

    #: weight gradients (wgrad + bias column sums + unpack + the data-parallel exchange hook) run on their own stream
    #: and overlap the latency-bound parts of the dgrad chain; set False to keep everything on the compute stream
    overlap_wgrad = _Switch("engine_overlap")

    #: (a second weight-gradient stream for the hyper path, the slab sums on a stream of their own and the context model on a third
    #: forward stream were switches until round 4: +0.8, +0.6 and +1.6 ms per bench step -- removed)
    #: the next forward's weight packing is split: forward-role copies on the compute stream (the forward waits for them), the
    #: input-gradient copies on a weight-gradient stream (only backward waits): 22.48-22.62 ms against 22.67-22.82 ms per bench
    #: step; STEM_ENGINE_SPLIT_PACK=0: one launch each as before
    split_pack = _Switch("engine_split_pack")
    #: ... and of the forward-role images only those of the layers that open the forward are packed on the compute stream, the others
    #: on the weight-gradient stream under the opening kernels.  Shortens the P-frame step's critical path by ~60 us when the step
    #: runs alone; inside the bench step (chip shared with the latent prefetch) it measured 12.39 against 12.32 ms
    #: (profiles/r05_ab_pack_first.log): off by default, STEM_ENGINE_PACK_FIRST=1 enables it
    pack_first = _Switch("engine_pack_first")

    #: the temporal-prior chain's weight gradients (three filter-row launches + slab sums, ~300 us of the ~900 us the
    #: weight-gradient stream carries per P-frame step) are issued on the COMPUTE stream, behind the chain's own input gradients:
    #: that stream has nothing else to do once the TPM input gradients are out (the hyper chain runs on its branch stream), while the
    #: weight-gradient stream was the last to finish by ~250 us (profiles/r05_gantt_palone.txt).  Scheduling only: lane -1 = "the
    #: stream the backward runs on".  STEM_ENGINE_TPM_WGRAD_INLINE=0: every weight gradient on the side stream, as in round 4
    tpm_wgrad_inline = _Switch("engine_tpm_wgrad_inline")

    def side_stream(self, device, lane=0):
        if not self.overlap_wgrad or device.type != "cuda" or lane < 0:
            return None
        st = self._side.get(lane)
        if st is None or st.device != device:
            st = self._side[lane] = F.make_stream(device, "side")
        return st

    #: the hyper path (HE -> bottleneck -> HD) and the temporal / spatial priors are independent until the entropy-parameter
    #: network joins them: the hyper path runs on its own stream in forward and backward so that the ramp-up / drain of its
    #: small launches overlaps the other branch's kernels (30.90 -> 30.75 ms per bench step; STEM_ENGINE_BRANCH=0 disables)
    branch_streams = _Switch("engine_branch")
    #: the temporal-prior chain is enqueued before the hyper branch (scheduling only)
    tpm_first = _Switch("engine_tpm_first")
    tpm_first_bwd = _Switch("engine_tpm_first_bwd")
    #: the context model's forward on the weight-gradient stream (opt-in experiment, STEM_ENGINE_CTX_ON_SIDE=1)
    ctx_on_side = _Switch("engine_ctx_on_side")
    ctx_split_on_side = _Switch("engine_ctx_split_on_side")
    #: one planes tensor for he_in = [y_cur | y_cond] and the TPM chain's input (its second half): a split launch less per step
    share_in_planes = _Switch("engine_share_in_planes")
    #: EPM.0's input gradient as one launch per prior range, the hyper chain's range first, so that the chain that ends the backward
    #: starts a third of the launch earlier.  Three 128-row-tile launches instead of one: 11.80 against 11.61 ms per bench step
    #: (profiles/r05_ab_epm_by_prior.log) -- three smaller launches cost more than the earlier start returns.  Off; STEM_ENGINE_EPM_DGRAD_BY_PRIOR=1
    epm_dgrad_by_prior = _Switch("engine_epm_dgrad_by_prior")
    #: the GaussianConditional's backward computed by the fused forward glue (one launch less per P-frame step)
    fuse_gc_backward = _Switch("engine_fuse_gc_backward")

    def _branch(self, device, which=0):
        if not self.branch_streams or device.type != "cuda":
            return None
        bs = getattr(self, "_bstreams", None)
        if bs is None:
            bs = self._bstreams = {}
        st = bs.get(which)
        if st is None or st.device != device:
            st = bs[which] = F.make_stream(device, "branch")
        return st

    def ensure_packed(self, block_max=None):
        """(Re)build every layer's packed weight copies with ONE kernel launch when any weight changed.  Inside
        StemEngine.forward the check has already run for the whole schedule (`_checked`).
        block_max = (maxima, flat): the per-chunk maxima an optimiser pass just left for the flat parameter buffer `flat`
        (optim.step(block_max=True)); the fp16 images of weights inside that buffer take their scales from them instead of a
        maximum launch.  Only meaningful in the call that directly follows that optimiser step."""
        if self._checked:
            return
        key = tuple((_layers.weight_epoch(l.mod.weight), l.mod.weight._version, l.mod.weight.data_ptr()) for l in self.layers)
        if key == self._pack_key:
            return
        stale = False
        for l in self.layers:
            have = l.wp_fwd if (l.fx3s or not l.fx3) else l.wp6_fwd
            stale = stale or have is None or have.device != l.mod.weight.device
        if stale:
            for l in self.layers:
                l.alloc_packs(l.mod.weight.device)
            self._pack_descs = None
        dev = self.layers[0].mod.weight.device
        if block_max is not None and self.pack_pair and self._pack_pairs(*block_max):
            self._pack_key = tuple((_layers.weight_epoch(l.mod.weight), l.mod.weight._version, l.mod.weight.data_ptr()) for l in self.layers)
            return
        side = self.side_stream(dev) if self.split_pack and not torch.cuda.is_current_stream_capturing() else None
        roles = ((0, 1), (1, 2)) if side is not None else ((0, 2),)
        for lo, hi in roles:                     # descriptor 0 of a layer = forward role, descriptor 1 = input-gradient role
            on_side = side is not None and lo == 1
            if on_side:
                F.stream_wait(side, F.cur_stream(dev))      # the optimiser step that changed the weights
            with F.on_stream(side if on_side else None):
                both = [l.role_descs(r) for l in self.layers for r in range(lo, hi)]
                descs = [d for a, _ in both for d in a]
                descs6 = [d for _, b in both for d in b]
                if descs6 and block_max is not None:
                    _attach_block_maxima(descs6, *block_max)
                first6 = []
                if descs6 and side is not None and not on_side and self.pack_first:
                    # The forward cannot start before its first kernels' images exist, and the optimiser pass cannot overlap anything:
                    # only the images of the layers that OPEN the forward (TPM.0, HE.0, the context model: 3.9 of 17 M weights) are
                    # packed on the compute stream; the others follow on the weight-gradient stream, in front of its input-gradient
                    # images, while the opening kernels run.  Their consumers wait for `fwd_rest` (_wait_fwd_rest).
                    opening = {id(l) for l in ([self.HE[0]] + ([self.TPM[0]] if self.has_tpm else []) + ([self.CTX] if self.has_spm else []))}
                    fwd_pairs = [(l, d) for l in self.layers for d in l.role_descs(0)[1]]
                    first6 = [d for l, d in fwd_pairs if id(l) in opening]
                    rest6 = [d for l, d in fwd_pairs if id(l) not in opening]
                    if first6 and rest6:
                        if block_max is not None:
                            _attach_block_maxima(first6 + rest6, *block_max)
                        F.pack_weights_f16x2_multi((_lib.F16PackDesc * len(first6))(*first6))
                        F.stream_wait(side, F.cur_stream(dev))
                        with F.on_stream(side):
                            F.pack_weights_f16x2_multi((_lib.F16PackDesc * len(rest6))(*rest6))
                            self._fwd_rest_event = self._events.setdefault("fwd_rest", torch.cuda.Event())
                            F.event_record(self._fwd_rest_event, side)
                    else:
                        first6 = []
                if descs6 and not first6:          # the forward's first kernels (HE.0, TPM.0, the context model) wait for these
                    if not on_side:
                        self._fwd_rest_event = None
                    F.pack_weights_f16x2_multi((_lib.F16PackDesc * len(descs6))(*descs6))
                # the fp32 copies of the forward role (the transposed hyper-decoder layers: consumed on the hyper branch, half a
                # forward later) are packed on that branch's stream, off the compute stream's optimiser -> forward chain
                bs = self._branch(dev) if (descs and side is not None and not on_side and lo == 0 and hi == 1) else None
                if descs and bs is not None:
                    F.stream_wait(bs, F.cur_stream(dev))
                    with F.on_stream(bs):
                        F.pack_weights_multi((_lib.PackDesc * len(descs))(*descs))
                        # one event object for the lifetime of the engine: a launch tape replays the record and the wait on it
                        self._fwd32_pack_event = self._events.setdefault("fwd32", torch.cuda.Event())
                        F.event_record(self._fwd32_pack_event, bs)
                elif descs:
                    F.pack_weights_multi((_lib.PackDesc * len(descs))(*descs))
                    if not on_side:
                        self._fwd32_pack_event = None
                if on_side:
                    self._dgrad_pack_event = self._events.setdefault("dgrad", torch.cuda.Event())
                    F.event_record(self._dgrad_pack_event, side)
        # masked == 2 zeroed taps of the context weight in place: refresh its version in the key
        self._pack_key = tuple((_layers.weight_epoch(l.mod.weight), l.mod.weight._version, l.mod.weight.data_ptr()) for l in self.layers)

    #: after an optimiser pass that left chunk maxima, both images of every layer come from ONE launch that reads each weight once
    #: (F.pack_weights_f16x2_pair_multi) on the compute stream, instead of one launch per role (the input-gradient role on the
    #: weight-gradient stream, under the forward).  STEM_ENGINE_PACK_PAIR=0: one launch per role as in round 4
    pack_pair = _Switch("engine_pack_pair")

    def _pack_pairs(self, maxima, flat):
        """the pair pack of every layer, if every layer's images are fp16 transposes of its tensor and every tensor lies inside the
        flat buffer the optimiser pass measured; else False (the per-role path runs)"""
        descs = [l.pair_desc() for l in self.layers]
        if any(d is None for d in descs):
            return False
        ch = F.adam_chunk()
        base, n = flat.data_ptr(), flat.numel()
        for d in descs:
            off = (d.w - base) // 4
            numel = d.A * d.B * d.R * d.S
            if (d.w - base) % 4 or off < 0 or off + numel > n:
                return False
            d.bmax, d.b0 = maxima.data_ptr(), off // ch
            d.nb = (off + numel - 1) // ch - d.b0 + 1
        F.pack_weights_f16x2_pair_multi((_lib.F16PairDesc * len(descs))(*descs))
        self._dgrad_pack_event = self._fwd_rest_event = None
        return True

    def unpack_all(self):
        for lane in sorted({l.lane for l in self.layers}):
            self._group_ready([l for l in self.layers if l.lane == lane], [])

    #: optional callable(list_of_parameters): invoked during backward as soon as the gradients of a module group
    #: (EPM, context_prediction, TPM, HD + entropy_bottleneck, HE -- the order backward produces them) are final,
    #: so a data-parallel reducer can start exchanging that slice while the rest of backward still runs
    grad_ready_hook = None

    #: one launch for a group's bias-gradient second stages (stem_bias_grad_final_multi); STEM_ENGINE_BIAS_MULTI=0: one per layer
    defer_bias_final = _Switch("engine_bias_multi")

    def _group_ready(self, layers, extra_params):
        dev = layers[0].mod.weight.device
        side = self.side_stream(dev, layers[0].lane)
        if side is None:
            return self._group_ready_on_stream(layers, extra_params)
        # extra_params (entropy-bottleneck gradients) were produced on the compute stream: order them before the hook
        F.stream_wait(side, F.cur_stream(dev))
        with F.on_stream(side):
            self._group_ready_on_stream(layers, extra_params)

    def _group_ready_on_stream(self, layers, extra_params):
        bdescs = [l.pending_bias for l in layers if getattr(l, "pending_bias", None) is not None]
        for l in layers:
            l.pending_bias = None
        if bdescs:
            F.bias_grad_final_multi(bdescs)
        descs = [l.unpack_desc() for l in layers if l.pending is not None]
        if descs:
            arr = (_lib.UnpackDesc * len(descs))(*descs)
            F.unpack_wgrads_multi(arr)
        if self.grad_ready_hook is not None:
            params = [p for l in layers for p in (l.mod.weight, l.mod.bias) if p is not None] + list(extra_params)
            if params:
                hook = self.grad_ready_hook
                F.tape_py(lambda: hook(params))          # torch.distributed calls: a launch tape re-runs them at this point

    def join_side_stream(self):
        for st in self._side.values():
            F.stream_wait(F.cur_stream(st.device), st)

    def _wait_fwd32_packs(self):
        """a consumer of a forward-role fp32 weight copy: order it after the packing on the hyper branch's stream (a no-op
        for the hyper branch itself, which runs on that stream)"""
        if self._fwd32_pack_event is not None:
            F.event_wait(F.cur_stream(), self._fwd32_pack_event)

    def _wait_fwd_rest(self):
        """a consumer of a forward-role image that was packed on the weight-gradient stream (every layer but the ones that open
        the forward): order the current stream behind that packing"""
        if self._fwd_rest_event is not None:
            F.event_wait(F.cur_stream(), self._fwd_rest_event)

    def _wait_dgrad_packs(self):
        """backward's first consumer of an input-gradient weight copy: order it after the side-stream packing"""
        if self._dgrad_pack_event is not None:
            F.event_wait(F.cur_stream(), self._dgrad_pack_event)
            self._dgrad_pack_event = None

    # -------------------------------------------------------------------------------------------
    def forward(self, y_cur, y_cond, training: bool, rate_coef=None):
        """rate_coef = (coef, scale) selects the fused training glue (training only): the elementwise work between the
        convolutions runs as 4 kernels that also produce dlik = coef / lik and (y_bpp, z_bpp, loss) = scale * sum log2 lik,
        i.e. EMLoss forward AND backward (utils.py:18-27): `k` then carries "dlik_y", "dlik_z", "loss3"."""
        self.ensure_packed()
        self._checked = True          # weights cannot change inside one forward: skip the per-layer checks
        try:
            return self._forward(y_cur, y_cond, training, rate_coef)
        finally:
            self._checked = False

    def _forward(self, y_cur, y_cond, training: bool, rate_coef=None):
        m = self.m
        yc, yd = F.to_nhwc(y_cur.detach()), F.to_nhwc(y_cond.detach())
        B, Cin, H, W = yc.shape
        dev = yc.device
        eb, gc = m.entropy_bottleneck, m.gaussian_conditional
        fused = rate_coef is not None
        assert not fused or training, "the fused glue is the TRAINING forward"
        k = {}
        target = t_hat = y_hat = None
        ctx_go = None
        rec = self._rec = {}    # scale records left by the producers of fp32 tensors that are split for the fp16 kernels below
        if fused:
            # one kernel: he_in = [y_cur | y_cond], target, t_hat = target + noise, y_hat = t_hat (+ y_cond); it also records
            # max |y_cur|, |y_cond| and max |t_hat| per workgroup: the splits of he_in, y_cond and t_hat need no maximum pass
            slot = gc._noise_slot(yc) if self.has_spm else {}
            he_in, target, t_hat, y_hat = F.prior_prologue(yc, yd, self.residual, True, self.has_spm, records=rec if self.use_fx3 and self.use_records else None, **slot)
            if self.has_spm and self.ctx_on_side and self.side_stream(dev) is not None and self._branch(dev) is not None:
                ctx_go = self._events.setdefault("ctx_go", torch.cuda.Event())       # t_hat exists from here on
                F.event_record(ctx_go, F.cur_stream(dev))
        else:
            # hyper encoder on cat(y_cur, y_cond): the two halves are written into one buffer
            he_in = F.empty_nhwc(B, 2 * Cin, H, W, dev)
            F.copy_channels(yc, he_in[:, :Cin])
            F.copy_channels(yd, he_in[:, Cin:])
        P = 2 * Cin
        epm_in = F.empty_nhwc(B, self.nprior * P, H, W, dev)
        o_tp, o_hp = (0, P) if self.has_tpm else (None, 0)
        o_ctx = o_hp + P
        bs = self._branch(dev)
        main = F.cur_stream(dev) if bs is not None else None
        if bs is not None:
            F.stream_wait(bs, main)
        pl = {}             # planes copies of activations, kept for the weight gradients
        split = F.F16Planes.split
        tp0 = tp2 = None
        ctx_split_done = None
        if (fused and self.has_spm and self.CTX.fx3 and self.ctx_split_on_side and ctx_go is None and rec.get("t_hat") is not None
                and self.side_stream(dev) is not None):
            # t_hat exists since the prologue and the context model runs behind the TPM chain: its planes are made meanwhile on the
            # weight-gradient stream, which has nothing to do during the forward (one launch less between TPM.4 and the context model)
            side = self.side_stream(dev)
            F.stream_wait(side, F.cur_stream(dev))
            with F.on_stream(side):
                pl["t_hat"] = split(t_hat, src_q=_qp(rec.get("t_hat")))
                ctx_split_done = self._events.setdefault("ctx_split", torch.cuda.Event())
                F.event_record(ctx_split_done, side)
            t_hat.record_stream(side)

        if self.share_in_planes and fused and self.HE[0].fx3 and self.has_tpm and self.TPM[0].fx3 and Cin % 32 == 0 and rec.get("in") is not None:
            # he_in = [y_cur | y_cond] and the TPM chain's input y_cond share ONE planes tensor (same record: max(|y_cur|, |y_cond|)):
            # one split on the compute stream, the TPM chain reads its second half as a channel view, the hyper branch (which waits
            # for this stream anyway) the whole -- a launch less, the same values
            pl["he_in"] = split(he_in, src_q=_qp(rec.get("in")))
            pl["yd"] = pl["he_in"].channels(Cin, 2 * Cin)
            if bs is not None:
                F.stream_wait(bs, main)

        def tpm_chain():
            nonlocal tp0, tp2
            if self.has_tpm and self.TPM[0].fx3:
                # planes travel from layer to layer (written by the producing epilogue next to the fp32 copy backward needs)
                if "yd" not in pl:
                    pl["yd"] = split(yd, src_q=_qp(rec.get("in")))       # max(|y_cur|, |y_cond|) bounds y_cond
                tp0, pl["tp0"] = self.TPM[0].fwd6(pl["yd"], F.ACT_LRELU, planes=True)
                self._wait_fwd_rest()
                tp2, pl["tp2"] = self.TPM[1].fwd6(pl["tp0"], F.ACT_LRELU, planes=True)
                self.TPM[2].fwd6(pl["tp2"], out=epm_in[:, o_tp:o_tp + P])
            elif self.has_tpm:
                self._wait_fwd_rest()
                tp0 = self.TPM[0].fwd(yd, F.ACT_LRELU)
                tp2 = self.TPM[1].fwd(tp0, F.ACT_LRELU)
                self.TPM[2].fwd(tp2, out=epm_in[:, o_tp:o_tp + P])

        if self.tpm_first:           # enqueued ahead of the hyper branch's ~14 launches: the TPM chain is the forward's critical path
            tpm_chain()
        with F.on_stream(bs):
            if self.HE[0].fx3:
                if "he_in" not in pl:
                    pl["he_in"] = split(he_in, src_q=_qp(rec.get("in")))
                he0, he0p = self.HE[0].fwd6(pl["he_in"], F.ACT_LRELU, planes=self.HE[1].fx3s)
            else:
                he0 = self.HE[0].fwd(he_in, F.ACT_LRELU)
            self._wait_fwd_rest()
            if self.HE[1].fx3s:           # the strided forwards on the general fp16 kernel, planes handed down
                he2, he2p = self.HE[1].fwd6(he0p, F.ACT_LRELU, planes=self.HE[2].fx3s)
                z = self.HE[2].fwd6(he2p)[0] if self.HE[2].fx3s else self.HE[2].fwd(he2)
                if self.HE[1].fx3t:       # the weight gradients read them again
                    pl["he0"], pl["he2"] = he0p, he2p
            else:
                he2 = self.HE[1].fwd(he0, F.ACT_LRELU)
                z = self.HE[2].fwd(he2)
            pack = F.eb_pack(eb._tensors14())
            if fused:
                if self.HD[0].fx3t and self.use_records:         # the kernel leaves max |z_hat| for the split below: no maximum pass
                    z_hat, lik_z, k["dlik_z"], part_z, rec["z_hat"] = F.eb_forward_train(z, pack, rate_coef[0], bound=eb._lik_bound, record=True,
                                                                                         **eb._noise_slot(z))
                else:
                    z_hat, lik_z, k["dlik_z"], part_z = F.eb_forward_train(z, pack, rate_coef[0], bound=eb._lik_bound, **eb._noise_slot(z))
            elif training:
                z_hat, lik_z = F.eb_forward(z, pack, noise=eb._noise_like(z))
            else:
                z_hat, lik_z = F.eb_forward(z, pack, medians=eb._medians_vec())
            # hyper decoder; its last conv writes the `hp` slice of the EPM input
            if self.HD[0].fx3t:          # the transposed layers on the fp16 kernel: planes in, planes out, no maximum / split passes
                pl["z_hat"] = split(z_hat, src_q=_qp(rec.get("z_hat")))
                hd0, pl["hd0"] = self.HD[0].fwd6t(pl["z_hat"], F.ACT_LRELU, planes=True)
                hd2, pl["hd2"] = self.HD[1].fwd6t(pl["hd0"], F.ACT_LRELU, planes=True)
                self.HD[2].fwd6(pl["hd2"], out=epm_in[:, o_hp:o_hp + P])
            else:
                hd0 = self.HD[0].fwd(z_hat, F.ACT_LRELU)
                hd2 = self.HD[1].fwd(hd0, F.ACT_LRELU)
            if self.HD[0].fx3t:
                pass
            elif self.HD[2].fx3:
                pl["hd2"] = split(hd2)
                self.HD[2].fwd6(pl["hd2"], out=epm_in[:, o_hp:o_hp + P])
            else:
                self.HD[2].fwd(hd2, out=epm_in[:, o_hp:o_hp + P])
        if not self.tpm_first:
            tpm_chain()
        if not fused:
            target = F.sub(yc, yd) if self.residual else (yc if F.nhwc_ld(yc) == Cin else F.copy_channels(yc, F.empty_nhwc(B, Cin, H, W, dev)))
        if self.has_spm:
            # gaussian_conditional.quantize(target, "noise" | "dequantize") with no means (:570-572, :853-855)
            if not fused:
                t_hat = F.add(target, gc._noise_like(target)) if training else F.round_(target)
            if ctx_go is not None:
                # the context model only needs t_hat: on the weight-gradient stream (idle during the forward) next to the TPM chain
                # and the hyper branch, instead of behind the TPM chain
                side = self.side_stream(dev)
                F.event_wait(side, ctx_go)
                with F.on_stream(side):
                    self._ctx_forward(t_hat, epm_in[:, o_ctx:o_ctx + P], pl)
                t_hat.record_stream(side)
                epm_in.record_stream(side)
                F.stream_wait(main, side)
            else:
                if ctx_split_done is not None:
                    F.event_wait(F.cur_stream(dev), ctx_split_done)
                self._ctx_forward(t_hat, epm_in[:, o_ctx:o_ctx + P], pl)
        if bs is not None:
            F.stream_wait(main, bs)
        self._wait_fwd_rest()
        if self.EPM[0].fx3:
            pl["epm_in"] = split(epm_in)
            e0, pl["e0"] = self.EPM[0].fwd6(pl["epm_in"], F.ACT_LRELU, planes=True)
            e2, pl["e2"] = self.EPM[1].fwd6(pl["e0"], F.ACT_LRELU, planes=True)
            gp = self.EPM[2].fwd6(pl["e2"])[0]                     # [B, 2*Cin, H, W] = scales | means
        else:
            e0 = self.EPM[0].fwd(epm_in, F.ACT_LRELU)
            e2 = self.EPM[1].fwd(e0, F.ACT_LRELU)
            gp = self.EPM[2].fwd(e2)                               # [B, 2*Cin, H, W] = scales | means
        scales, means = gp[:, :Cin], gp[:, Cin:]
        if fused:
            # the GaussianConditional's backward in the same launch (d loss / d likelihood = coef / lik is known here): backward()
            # uses it when it is handed this very dlik_y
            if self.fuse_gc_backward:
                dgp = F.empty_nhwc(B, 2 * Cin, H, W, dev)
                gc_out, lik_y, k["dlik_y"], part_y, k["qg"] = F.gc_forward_train(
                    target, scales, means, rate_coef[0], scale_bound=gc._scale_bound, lik_bound=gc._lik_bound,
                    backward=(dgp[:, :Cin], dgp[:, Cin:]), record=self.EPM[0].fx3 and self.use_records, **gc._noise_slot(target))
                k["dgp"] = dgp
            else:
                gc_out, lik_y, k["dlik_y"], part_y = F.gc_forward_train(target, scales, means, rate_coef[0], scale_bound=gc._scale_bound,
                                                                        lik_bound=gc._lik_bound, **gc._noise_slot(target))
            k["loss3"] = F.em_loss_finalize(part_y, part_z, rate_coef[1])
            if not self.has_spm:
                y_hat = gc_out
        else:
            noise = gc._noise_like(target) if training else None
            gc_out, lik_y = F.gc_forward(target, scales, means, noise=noise, scale_bound=gc._scale_bound, lik_bound=gc._lik_bound)
            if self.has_spm:
                y_hat = F.add(t_hat, yd if F.nhwc_ld(yd) == Cin else F.copy_channels(yd, F.empty_nhwc(B, Cin, H, W, dev))) if self.residual else t_hat
            else:
                y_hat = gc_out
        k.update(he_in=he_in, he0=he0, he2=he2, z_hat=z_hat, pack=pack, hd0=hd0, hd2=hd2, epm_in=epm_in, tp0=tp0, tp2=tp2,
                 yd=yd, t_hat=t_hat, e0=e0, e2=e2, gp=gp, gc_out=gc_out, offs=(o_tp, o_hp, o_ctx), P=P, Cin=Cin, planes=pl)
        return y_hat, lik_y, lik_z, k

    # -------------------------------------------------------------------------------------------
    def backward(self, k, dlik_y, dlik_z):
        """Parameter gradients of a scalar that depends on (lik_y, lik_z), ADDED into .grad as autograd does (the
        training loop zeroes them before every backward, stem/trainSTEM.py:203; a slot that does not exist yet starts
        at zero), so several backward passes between two zero_grad() calls accumulate -- the same semantics as the
        layer-wise Functions of the variable-rate models (layers.py)."""
        m = self.m
        gc = m.gaussian_conditional
        Cin, P = k["Cin"], k["P"]
        o_tp, o_hp, o_ctx = k["offs"]
        gp = k["gp"]
        B, _, H, W = gp.shape
        self._wait_dgrad_packs()
        if k.get("dgp") is not None and dlik_y is k.get("dlik_y"):          # computed by the fused forward glue
            dgp, qg = k["dgp"], k["qg"]
        else:
            dgp = F.empty_nhwc(B, 2 * Cin, H, W, gp.device)
            qg = F.gc_backward(k["gc_out"], gp[:, :Cin], gp[:, Cin:], dlik_y, dgp[:, :Cin], dgp[:, Cin:], dy=None,
                               scale_bound=gc._scale_bound, lik_bound=gc._lik_bound, record=self.EPM[0].fx3 and self.use_records)
        # EPM (1x1 chain)
        dprip = None
        pl = k.get("planes", {})
        if self.EPM[0].fx3:
            dgpp = F.F16Planes.split(dgp, src_q=_qp(qg))
            self.EPM[2].wgrad_any(k["e2"], dgp, pl.get("e2"), dgpp)
            de2, de2p = self.EPM[2].dgrad6(dgpp, xact=k["e2"], planes=True)
            self.EPM[1].wgrad_any(k["e0"], de2, pl.get("e0"), de2p)
            de0, de0p = self.EPM[1].dgrad6(de2p, xact=k["e0"], planes=True)
            self.EPM[0].wgrad_any(k["epm_in"], de0, pl.get("epm_in"), de0p)
            if self.epm_dgrad_by_prior and P % 128 == 0 and self.branch_streams and self._branch(gp.device) is not None:
                # EPM.0's input gradient feeds three independent consumers (hyper chain, TPM chain, context weight gradient): computed
                # range by range, the hyper chain's first -- the chain that ends the backward -- which then starts a third of the
                # launch earlier (rows [n0, n0 + P) of the flipped weight image = whole 128-row tiles)
                l0 = self.EPM[0]
                dpri = F.empty_nhwc(B, self.nprior * P, H, W, gp.device)
                order = [o_hp] + ([o_tp] if self.has_tpm else []) + ([o_ctx] if self.has_spm else [])
                parts = {}
                for j, o in enumerate(order):
                    _, parts[(o, o + P)] = F.conv2d_f16x3_gen(de0p, l0.wp6_dgrad, None, P, l0.R, l0.R, 1, l0.pad, out=dpri[:, o:o + P],
                                                              want_planes=True, rows=(self.nprior * P, o))
                    if j == 0:
                        hp_done = self._events.setdefault("bwd_epm", torch.cuda.Event())
                        F.event_record(hp_done, F.cur_stream(gp.device))
                dprip = _RangePlanes(parts)
            else:
                dpri, dprip = self.EPM[0].dgrad6(de0p, planes=True)    # the prior branches read 32-aligned channel views of the planes
        else:
            self.EPM[2].wgrad(k["e2"], dgp)
            de2 = self.EPM[2].dgrad(dgp, k["e2"].shape, xact=k["e2"])
            self.EPM[1].wgrad(k["e0"], de2)
            de0 = self.EPM[1].dgrad(de2, k["e0"].shape, xact=k["e0"])
            self.EPM[0].wgrad(k["epm_in"], de0)
            dpri = self.EPM[0].dgrad(de0, k["epm_in"].shape)
        self._group_ready(self.EPM, [])
        bs = self._branch(gp.device)
        main = F.cur_stream(gp.device)

        if bs is not None:                 # the hyper chain depends on the EPM input gradient only: its point on the compute stream
            epm_done = self._events.setdefault("bwd_epm", torch.cuda.Event())
            if not isinstance(dprip, _RangePlanes):          # (range by range: recorded right behind the hyper chain's range)
                F.event_record(epm_done, main)

        def hyper_branch():                # hyper chain (HD -> bottleneck -> HE) on its own stream, next to the TPM chain
            F.event_wait(bs, epm_done)
            with F.on_stream(bs):
                self._backward_hyper(k, dpri, dlik_z, dprip)

        if bs is not None and not self.tpm_first_bwd:
            hyper_branch()
        # spatial prior: weight gradient of all 25 taps, no input gradient (its input is data + noise)
        if self.has_spm:
            if self.CTX.wg3 and dprip is not None:
                thp = pl.get("t_hat") or F.F16Planes.split(k["t_hat"])         # left by the forward when it ran on the fp16 kernel
                self.CTX.wgrad_any(k["t_hat"], dpri[:, o_ctx:o_ctx + P], thp, dprip.channels(o_ctx, o_ctx + P))
            else:
                self.CTX.wgrad(k["t_hat"], dpri[:, o_ctx:o_ctx + P])
            self._group_ready([self.CTX], [])
        if self.has_tpm:
            for l in self.TPM:
                l.lane = -1 if (self.tpm_wgrad_inline and bs is not None) else 0
            dtp = dpri[:, o_tp:o_tp + P]
            if self.TPM[2].fx3:
                dtpp = dprip.channels(o_tp, o_tp + P) if dprip is not None else F.F16Planes.split(dtp)
                self.TPM[2].wgrad_any(k["tp2"], dtp, pl.get("tp2"), dtpp)
                d, dp = self.TPM[2].dgrad6(dtpp, xact=k["tp2"], planes=True)
                self.TPM[1].wgrad_any(k["tp0"], d, pl.get("tp0"), dp)
                d, dp = self.TPM[1].dgrad6(dp, xact=k["tp0"], planes=self.TPM[0].wg3)
                self.TPM[0].wgrad_any(k["yd"], d, pl.get("yd"), dp)
            else:
                self.TPM[2].wgrad(k["tp2"], dtp)
                d = self.TPM[2].dgrad(dtp, k["tp2"].shape, xact=k["tp2"])
                self.TPM[1].wgrad(k["tp0"], d)
                d = self.TPM[1].dgrad(d, k["tp0"].shape, xact=k["tp0"])
                self.TPM[0].wgrad(k["yd"], d)
            self._group_ready(self.TPM, [])
        if bs is not None and self.tpm_first_bwd:
            hyper_branch()
        if bs is None:
            self._backward_hyper(k, dpri, dlik_z, dprip)
        else:
            F.stream_wait(main, bs)
        self.join_side_stream()          # gradients are complete for whatever the compute stream does next

    def _ctx_forward(self, t_hat, out, pl):
        """context_prediction(t_hat) -> its channel slice of the EPM input (spatiotemporalpriors.py:857): on the fp16 kernel over
        the live taps of the mask (the planes of t_hat stay for the weight gradient), else on igemm.hip's masked form"""
        if self.CTX.fx3:
            if "t_hat" not in pl:
                pl["t_hat"] = F.F16Planes.split(t_hat, src_q=_qp(self._rec.get("t_hat")))
            self.CTX.fwd6(pl["t_hat"], out=out)
        else:
            self.CTX.fwd(t_hat, out=out)

    def _backward_hyper(self, k, dpri, dlik_z, dprip=None):
        m = self.m
        P = k["P"]
        o_tp, o_hp, o_ctx = k["offs"]
        # hyper decoder
        dhp = dpri[:, o_hp:o_hp + P]
        pl = k.get("planes", {})
        if self.HD[2].fx3:
            dhpp = dprip.channels(o_hp, o_hp + P) if dprip is not None else F.F16Planes.split(dhp)
            self.HD[2].wgrad_any(k["hd2"], dhp, pl.get("hd2"), dhpp)
            d, dp = self.HD[2].dgrad6(dhpp, xact=k["hd2"], planes=self.HD[1].fx3s)
        else:
            self.HD[2].wgrad(k["hd2"], dhp)
            d = self.HD[2].dgrad(dhp, k["hd2"].shape, xact=k["hd2"])
        if self.HD[1].fx3t:
            self.HD[1].wgrad_t(dp, pl["hd0"], d)
        else:
            self.HD[1].wgrad(k["hd0"], d)
        if self.HD[1].fx3s:               # input gradients of the transposed layers = strided convolutions of dy, planes handed down
            d, dp = self.HD[1].dgrad6s(dp, xact=k["hd0"], planes=self.HD[0].fx3s)
        else:
            d = self.HD[1].dgrad(d, k["hd0"].shape, xact=k["hd0"])
        if self.HD[0].fx3t:
            self.HD[0].wgrad_t(dp, pl["z_hat"], d)
        else:
            self.HD[0].wgrad(k["z_hat"], d)
        dz_hat = self.HD[0].dgrad6s(dp)[0] if self.HD[0].fx3s else self.HD[0].dgrad(d, k["z_hat"].shape)
        # entropy bottleneck: d/dz = dz_hat + likelihood path; 58 parameter gradients per channel
        eb = m.entropy_bottleneck
        qdz = None
        if self.HE[2].fx3t and self.use_records:
            dz, dpack, qdz = F.eb_backward(k["z_hat"], k["pack"], dlik_z, dzhat_in=dz_hat, bound=eb._lik_bound, record=True)
        else:
            dz, dpack = F.eb_backward(k["z_hat"], k["pack"], dlik_z, dzhat_in=dz_hat, bound=eb._lik_bound)
        F.eb_unpack_grads(dpack, [_grad_of(p) for p in eb._tensors14()], accumulate=self.accumulate_grads)
        self._group_ready(self.HD, eb._tensors14())
        # hyper encoder
        if self.HE[2].fx3t:               # transposed faces (input gradients of the strided convolutions) and weight gradients on the fp16 kernels
            dzp = F.F16Planes.split(dz, src_q=_qp(qdz))
            self.HE[2].wgrad_t(pl["he2"], dzp, dz)
            d, dp = self.HE[2].dgrad6t(dzp, xact=k["he2"], planes=True)
            self.HE[1].wgrad_t(pl["he0"], dp, d)
            d, dp = self.HE[1].dgrad6t(dp, xact=k["he0"], planes=self.HE[0].wg3 and "he_in" in pl)
            self.HE[0].wgrad_any(k["he_in"], d, pl.get("he_in"), dp)
        else:
            self.HE[2].wgrad(k["he2"], dz)
            d = self.HE[2].dgrad(dz, k["he2"].shape, xact=k["he2"])
            self.HE[1].wgrad(k["he0"], d)
            d = self.HE[1].dgrad(d, k["he0"].shape, xact=k["he0"])
            self.HE[0].wgrad_any(k["he_in"], d, pl.get("he_in"), F.F16Planes.split(d) if self.HE[0].wg3 and "he_in" in pl else None)
        self._group_ready(self.HE, [])


class StemFunction(torch.autograd.Function):
    """Autograd node wrapping a whole STEM forward: inputs are the model parameters (so that
    `loss.backward()` reaches us), outputs (y_hat, lik_y, lik_z).  Parameter gradients are written
    directly into `.grad` by the engine and `None` is returned to autograd."""

    @staticmethod
    def forward(ctx, engine, training, y_cur, y_cond, *params):
        y_hat, lik_y, lik_z, k = engine.forward(y_cur, y_cond, training)
        ctx.engine, ctx.keep = engine, k
        ctx.mark_non_differentiable(y_hat)
        return y_hat, lik_y, lik_z

    @staticmethod
    def backward(ctx, _dy_hat, dlik_y, dlik_z):
        eng, k = ctx.engine, ctx.keep
        if k is None:
            raise RuntimeError("StemFunction: backward through this forward a second time -- its saved activations were "
                               "released by the first backward (retain_graph is not supported by the fused schedule; "
                               "run the forward again)")
        ctx.keep = None
        gp = k["gp"]
        B, _, H, W = gp.shape
        if dlik_y is None:
            dlik_y = torch.zeros_like(k["gc_out"])
        if dlik_z is None:
            dlik_z = torch.zeros_like(k["z_hat"])
        eng.backward(k, _dense(dlik_y), _dense(dlik_z))
        return (None,) * (4 + len(eng.m._engine_params))


def _dense(t):
    t = F.to_nhwc(t)
    if F.nhwc_ld(t) != t.shape[1]:
        t = F.copy_channels(t, F.empty_nhwc(*t.shape, t.device))
    return t
