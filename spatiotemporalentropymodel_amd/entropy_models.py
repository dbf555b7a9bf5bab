"""EntropyModel / EntropyBottleneck / GaussianConditional with the reference's API
(compressai/entropy_models/entropy_models.py:68-604).

Device side (forward / likelihood / quantize / build_indexes, forward AND backward) = fused HIP
kernels.  Host side (`update()`: CDF tables once per model, `compress()` / `decompress()`: the
rANS coder) = numpy + libstem_rans.so, as north_star prescribes ("the rANS coder stays on host").
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import scipy.stats
import torch
import torch.nn as nn

from . import _lib
from . import functional as F
from .ops import LowerBound


# ----------------------------------------------------------------------------- host codec
def pmf_to_quantized_cdf(pmf, precision=16):
    """compressai._CXX.pmf_to_quantized_cdf (cpp_exts/ops/ops.cpp:24-81) -> torch.IntTensor"""
    p = np.ascontiguousarray(pmf.detach().cpu().numpy() if torch.is_tensor(pmf) else pmf, dtype=np.float32)
    cdf = np.empty(p.size + 1, np.uint32)
    lib = _lib.rans()
    if lib.stem_pmf_to_quantized_cdf(p.ctypes.data, p.size, int(precision), cdf.ctypes.data) != 0:
        raise ValueError(lib.stem_rans_last_error().decode())
    return torch.from_numpy(cdf.astype(np.int32))


class _Tables:
    """Dense int32 views of (cdf, cdf_length, offset) handed to libstem_rans.so without per-call conversion."""

    def __init__(self, cdf, sizes, offsets):
        self.cdf = np.ascontiguousarray(cdf.detach().cpu().numpy() if torch.is_tensor(cdf) else cdf, dtype=np.int32)
        self.sizes = np.ascontiguousarray(sizes.detach().cpu().numpy() if torch.is_tensor(sizes) else sizes, dtype=np.int32).reshape(-1)
        self.offsets = np.ascontiguousarray(offsets.detach().cpu().numpy() if torch.is_tensor(offsets) else offsets, dtype=np.int32).reshape(-1)
        assert self.cdf.ndim == 2 and self.cdf.shape[0] == self.sizes.size == self.offsets.size

    def args(self):
        return (self.cdf.ctypes.data, self.cdf.shape[0], self.cdf.shape[1], self.sizes.ctypes.data, self.offsets.ctypes.data)


def _i32(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int32)


def _as_tables(cdfs, sizes, offsets):
    return cdfs if isinstance(cdfs, _Tables) else _Tables(np.asarray(cdfs, dtype=np.int32), sizes, offsets)


def _rans_err():
    return RuntimeError(_lib.rans().stem_rans_last_error().decode())


class RansEncoder:
    """compressai.ans.RansEncoder (cpp_exts/rans/rans_interface.cpp:193-204)."""

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes=None, offsets=None) -> bytes:
        t = _as_tables(cdfs, cdfs_sizes, offsets)
        sym, idx = _i32(symbols), _i32(indexes)
        if sym.size != idx.size:
            raise ValueError("symbols and indexes differ in length")
        cap = 8 * sym.size + 16
        out = np.empty(cap, np.uint8)
        n = _lib.rans().stem_rans_encode(sym.ctypes.data, idx.ctypes.data, sym.size, *t.args(), out.ctypes.data, cap)
        if n < 0:
            raise _rans_err()
        return out[:n].tobytes()


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder (rans_interface.cpp:99-191)."""

    def __init__(self):
        self._h = _lib.rans().stem_rans_encoder_create()
        self._destroy = _lib.rans().stem_rans_encoder_destroy      # bound now: module globals vanish at shutdown

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes=None, offsets=None) -> None:
        t = _as_tables(cdfs, cdfs_sizes, offsets)
        sym, idx = _i32(symbols), _i32(indexes)
        if _lib.rans().stem_rans_encoder_push(self._h, sym.ctypes.data, idx.ctypes.data, sym.size, *t.args()) != 0:
            raise _rans_err()

    def flush(self) -> bytes:
        lib = _lib.rans()
        cap = lib.stem_rans_encoder_pending_bytes(self._h)
        out = np.empty(cap, np.uint8)
        n = lib.stem_rans_encoder_flush(self._h, out.ctypes.data, cap)
        if n < 0:
            raise _rans_err()
        return out[:n].tobytes()


class RansDecoder:
    """compressai.ans.RansDecoder (rans_interface.cpp:206-350)."""

    def __init__(self):
        self._h = _lib.rans().stem_rans_decoder_create()
        self._destroy = _lib.rans().stem_rans_decoder_destroy

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes=None, offsets=None):
        return self.decode_with_indexes_np(encoded, indexes, cdfs, cdfs_sizes, offsets).tolist()

    def decode_with_indexes_np(self, encoded, indexes, cdfs, cdfs_sizes=None, offsets=None):
        t = _as_tables(cdfs, cdfs_sizes, offsets)
        buf = np.frombuffer(encoded, np.uint8)
        idx = _i32(indexes)
        out = np.empty(idx.size, np.int32)
        if _lib.rans().stem_rans_decode(buf.ctypes.data, buf.size, idx.ctypes.data, idx.size, *t.args(), out.ctypes.data) != 0:
            raise _rans_err()
        return out

    def set_stream(self, encoded) -> None:
        buf = np.frombuffer(encoded, np.uint8)
        if _lib.rans().stem_rans_decoder_set_stream(self._h, buf.ctypes.data, buf.size) != 0:
            raise _rans_err()

    def decode_stream(self, indexes, cdfs, cdfs_sizes=None, offsets=None):
        return self.decode_stream_np(indexes, cdfs, cdfs_sizes, offsets).tolist()

    def decode_stream_np(self, indexes, cdfs, cdfs_sizes=None, offsets=None):
        t = _as_tables(cdfs, cdfs_sizes, offsets)
        idx = _i32(indexes)
        out = np.empty(idx.size, np.int32)
        if _lib.rans().stem_rans_decoder_decode(self._h, idx.ctypes.data, idx.size, *t.args(), out.ctypes.data) != 0:
            raise _rans_err()
        return out


_ENTROPY_CODER = ["ans"]


def available_entropy_coders():
    """compressai/__init__.py:58-62 (the optional `rangecoder` pip package is not provided)."""
    return ["ans"]


def set_entropy_coder(entropy_coder):
    if entropy_coder not in available_entropy_coders():
        raise ValueError(f'Invalid entropy coder "{entropy_coder}", choose from({", ".join(available_entropy_coders())}).')
    _ENTROPY_CODER[0] = entropy_coder


def get_entropy_coder():
    return _ENTROPY_CODER[0]


class _EntropyCoder:
    def __init__(self, method):
        if not isinstance(method, str):
            raise ValueError(f'Invalid method type "{type(method)}"')
        if method not in available_entropy_coders():
            raise ValueError(f'Unknown entropy coder "{method}" (available: {", ".join(available_entropy_coders())})')
        self._encoder, self._decoder = RansEncoder(), RansDecoder()

    def encode_with_indexes(self, *args, **kwargs):
        return self._encoder.encode_with_indexes(*args, **kwargs)

    def decode_with_indexes(self, *args, **kwargs):
        return self._decoder.decode_with_indexes(*args, **kwargs)


# ----------------------------------------------------------------------------- autograd glue
class _EBFunction(torch.autograd.Function):
    """EntropyBottleneck.forward (entropy_models.py:424-452) as one kernel each way."""

    @staticmethod
    def forward(ctx, x, noise, medians, *tensors14):
        pack = F.eb_pack(list(tensors14))
        z_hat, lik = F.eb_forward(F.to_nhwc(x), pack, medians=medians, noise=noise)
        ctx.save_for_backward(z_hat, pack)
        return z_hat, lik

    @staticmethod
    def backward(ctx, dzhat, dlik):
        z_hat, pack = ctx.saved_tensors
        dlik = _dense(dlik)
        dzin = None if dzhat is None else _dense(dzhat)
        dz, dpack = F.eb_backward(z_hat, pack, dlik, dzhat_in=dzin)
        grads = [torch.empty((pack.shape[0], *s), device=pack.device) for s in _EB_SHAPES]
        F.eb_unpack_grads(dpack, grads)
        return (dz, None, None, *grads)


_EB_SHAPES = [(3, 1), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (1, 3), (1, 1)]


def _dense(t):
    t = F.to_nhwc(t)
    if F.nhwc_ld(t) != t.shape[1]:
        t = F.copy_channels(t, F.empty_nhwc(*t.shape, t.device))
    return t


class _GCFunction(torch.autograd.Function):
    """GaussianConditional.forward (entropy_models.py:588-596): quantize + likelihood + both LowerBounds."""

    @staticmethod
    def forward(ctx, y, scales, means, noise, scale_bound, lik_bound):
        y = _dense(y)
        B, Cc, H, W = y.shape
        # scales / means are usually the two halves of one EPM output (chunk(2,1)): equal pitch views
        sc, mu = F.to_nhwc(scales), F.to_nhwc(means)
        if F.nhwc_ld(sc) != F.nhwc_ld(mu):
            sc, mu = _dense(sc), _dense(mu)
        out, lik = F.gc_forward(y, sc, mu, noise=noise, scale_bound=scale_bound, lik_bound=lik_bound)
        ctx.bounds = (scale_bound, lik_bound)
        ctx.save_for_backward(out, sc, mu)
        return out, lik

    @staticmethod
    def backward(ctx, dout, dlik):
        out, sc, mu = ctx.saved_tensors
        B, Cc, H, W = out.shape
        dgp = F.empty_nhwc(B, 2 * Cc, H, W, out.device)
        dy = F.empty_nhwc(B, Cc, H, W, out.device) if ctx.needs_input_grad[0] else None
        F.gc_backward(out, sc, mu, _dense(dlik), dgp[:, :Cc], dgp[:, Cc:], dy=dy, scale_bound=ctx.bounds[0], lik_bound=ctx.bounds[1])
        if dy is not None and dout is not None:
            dy = F.add(dy, _dense(dout))
        return dy, dgp[:, :Cc], dgp[:, Cc:], None, None, None


# ----------------------------------------------------------------------------- modules
class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        if entropy_coder is None:
            entropy_coder = get_entropy_coder()
        self.entropy_coder = _EntropyCoder(entropy_coder)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self._lik_bound = float(likelihood_bound)
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        # noise: counter-based Philox stream (seed, offset); `noise_source(shape, device)` overrides it (parity tests)
        self.noise_seed = 0x5713
        self._noise_offset = 0
        self.noise_source = None
        self.noise_epoch = None       # 1-element int64 device tensor: device-side draw count (captured hipGraphs)
        self._tables = None

    offset = property(lambda self: self._offset)
    quantized_cdf = property(lambda self: self._quantized_cdf)
    cdf_length = property(lambda self: self._cdf_length)

    def forward(self, *args):
        raise NotImplementedError()

    def _noise_like(self, x_nhwc):
        """U(-1/2, 1/2) with x's logical [B,C,H,W] shape, NHWC memory (replaces _get_noise_cached, :112-120)."""
        if self.noise_source is not None:
            n = self.noise_source(tuple(x_nhwc.shape), x_nhwc.device)
            return _dense(n.to(x_nhwc.device))
        out = F.uniform_noise_like(x_nhwc, self.noise_seed, self._noise_offset, epoch=self.noise_epoch)
        self._noise_offset += (x_nhwc.numel() + 3) // 4
        return out

    def _noise_slot(self, x_nhwc):
        """For kernels that draw their noise inline: either {"noise": tensor} (injected source) or the Philox coordinates
        {"seed", "offset", "epoch"} the next _noise_like(x) would have used -- and advance the stream past them."""
        if self.noise_source is not None:
            return {"noise": self._noise_like(x_nhwc)}
        slot = {"seed": self.noise_seed, "offset": self._noise_offset, "epoch": self.noise_epoch}
        self._noise_offset += (x_nhwc.numel() + 3) // 4
        return slot

    def quantize(self, inputs, mode, means=None):
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        x = _dense(inputs.detach() if not inputs.requires_grad else inputs)
        if mode == "noise":
            if inputs.requires_grad:
                return inputs + self._noise_like(x)      # differentiable identity path (torch add; off the hot path)
            return F.add(x, self._noise_like(x))
        if means is not None:
            m = _dense(means.detach().expand_as(inputs).contiguous() if means.shape != inputs.shape else means.detach())
            x = F.sub(x, m)
        out = F.round_(x)
        if mode == "dequantize":
            return F.add(out, m) if means is not None else out
        return out.int()

    @staticmethod
    def dequantize(inputs, means=None):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.float()
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """One quantised CDF row per distribution: its first pmf_length[i] masses followed by the tail mass, through
        pmf_to_quantized_cdf at the coder's precision, zero-padded to max_length + 2 entries (entropy_models.py:170-176)."""
        rows = torch.zeros((len(pmf_length), int(max_length) + 2), dtype=torch.int32)
        for row, masses, tail, n in zip(rows, pmf, tail_mass, pmf_length):
            q = pmf_to_quantized_cdf(torch.cat((masses[: int(n)], tail), dim=0), self.entropy_coder_precision)
            row[: q.numel()] = q
        return rows

    @staticmethod
    def _require_table(t, ndim, missing, what):
        # the reference's messages (entropy_models.py:178-199): callers and tests match on them
        if t.numel() == 0:
            raise ValueError(f"Uninitialized {missing}. Run update() first")
        if t.dim() != ndim:
            raise ValueError(f"Invalid {what} size {t.size()}")

    def _check_cdf_size(self):
        self._require_table(self._quantized_cdf, 2, "CDFs", "CDF")

    def _check_offsets_size(self):
        self._require_table(self._offset, 1, "offsets", "offsets")

    def _check_cdf_length(self):
        self._require_table(self._cdf_length, 1, "CDF lengths", "offsets")

    def host_tables(self) -> _Tables:
        """(cdf, length, offset) as dense host arrays, cached until the next update()/load_state_dict."""
        key = (self._quantized_cdf.data_ptr(), self._quantized_cdf._version, tuple(self._quantized_cdf.shape))
        if self._tables is None or self._tables[0] != key:
            self._check_cdf_size(), self._check_cdf_length(), self._check_offsets_size()
            self._tables = (key, _Tables(self._quantized_cdf, self._cdf_length, self._offset))
        return self._tables[1]

    def compress(self, inputs, indexes, means=None):
        """symbols = round(inputs - means) coded with the host rANS, one string per batch element (:201-233)."""
        if len(inputs.size()) != 4:
            raise ValueError("Invalid `inputs` size. Expected a 4-D tensor.")
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        symbols = self.quantize(inputs, "symbols", means)
        t = self.host_tables()
        sym = symbols.cpu().contiguous().numpy()          # logical NCHW order, as the reference flattens
        idx = indexes.int().cpu().contiguous().numpy()
        enc = RansEncoder()
        return [enc.encode_with_indexes(sym[i], idx[i], t) for i in range(sym.shape[0])]

    @staticmethod
    def _check_decompress_args(strings, indexes, means):
        # means may be per element or one value per (batch, channel) broadcast over the spatial dims
        problems = (
            (not isinstance(strings, (tuple, list)), "Invalid `strings` parameter type."),
            (isinstance(strings, (tuple, list)) and len(strings) != indexes.shape[0], "Invalid strings or indexes parameters"),
            (indexes.dim() != 4, "Invalid `indexes` size. Expected a 4-D tensor."),
        )
        for bad, msg in problems:
            if bad:
                raise ValueError(msg)
        if means is None:
            return
        if tuple(means.shape[:-2]) != tuple(indexes.shape[:-2]):
            raise ValueError("Invalid means or indexes parameters")
        per_channel = means.shape[2] == 1 and means.shape[3] == 1
        if tuple(means.shape) != tuple(indexes.shape) and not per_channel:
            raise ValueError("Invalid means parameters")

    def decompress(self, strings, indexes, means=None):
        """one string per batch element -> dequantised values (:235-279); argument errors are ValueError as upstream"""
        self._check_decompress_args(strings, indexes, means)
        t = self.host_tables()
        idx = indexes.int().cpu().contiguous().numpy()
        dec = RansDecoder()
        vals = np.stack([dec.decode_with_indexes_np(s, idx[i], t).reshape(idx[i].shape) for i, s in enumerate(strings)])
        outputs = torch.from_numpy(vals).to(indexes.device)
        return self.dequantize(outputs, means)


class EntropyBottleneck(EntropyModel):
    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        if self.filters != (3, 3, 3, 3):
            raise NotImplementedError("the HIP bottleneck kernel is specialised for filters=(3,3,3,3) (the only STEM use)")
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        # Per-channel MLP 1 -> 3 -> 3 -> 3 -> 3 -> 1 of the cumulative (entropy_models.py:303-340 upstream).  Layer i maps
        # widths[i] -> widths[i+1]; the initial values make softplus(matrix) a constant 1 / (s * fan_out) with
        # s = init_scale^(1/5), biases are U(-1/2, 1/2) (drawn layer by layer: the only RNG use), gate factors 0.
        widths = (1, *self.filters, 1)
        n_layers = len(widths) - 1
        per_layer_scale = self.init_scale ** (1.0 / n_layers)
        C = self.channels
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            raw = float(np.log(np.expm1(1.0 / per_layer_scale / fan_out)))           # softplus^-1 of the target value
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((C, fan_out, fan_in), raw)))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(C, fan_out, 1).uniform_(-0.5, 0.5)))
            if i < n_layers - 1:
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(C, fan_out, 1)))
        # quantiles start at (-init_scale, 0, +init_scale); the aux loss pulls them to the tail_mass/2, 1/2, 1-tail_mass/2
        # points of the learned density, expressed as logits in `target`
        q0 = torch.tensor([-self.init_scale, 0.0, self.init_scale])
        self.quantiles = nn.Parameter(q0.expand(C, 1, 3).clone())
        tail_logit = float(np.log(2.0 / self.tail_mass - 1.0))
        self.register_buffer("target", torch.tensor([-tail_logit, 0.0, tail_logit]))

    def _tensors14(self):
        return [getattr(self, n) for n in F.EB_TENSORS]

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def _medians_vec(self):
        return self.quantiles.detach()[:, 0, 1].contiguous()

    # ---- host side -------------------------------------------------------------------------
    def _logits_cumulative_host(self, inputs):
        """entropy_models.py:388-407 on CPU tensors, used only by update() (once per model)."""
        logits = inputs
        for i in range(len(self.filters) + 1):
            matrix = getattr(self, f"_matrix{i:d}").detach().cpu()
            logits = torch.matmul(torch.nn.functional.softplus(matrix), logits)
            logits = logits + getattr(self, f"_bias{i:d}").detach().cpu()
            if i < len(self.filters):
                factor = getattr(self, f"_factor{i:d}").detach().cpu()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def update(self, force=False):
        """Host side, once per model: the integer CDF tables of the learned density on the support the quantiles span
        (entropy_models.py:341-381; the arithmetic order is the reference's, the tables are compared entry by entry)."""
        if self._offset.numel() > 0 and not force:
            return False
        lo_q, med, hi_q = self.quantiles.detach().cpu()[:, 0, :].unbind(dim=1)
        left = torch.clamp(torch.ceil(med - lo_q).int(), min=0)           # integer steps covered below / above the median
        right = torch.clamp(torch.ceil(hi_q - med).int(), min=0)
        length = right + left + 1
        longest = length.max()
        grid = torch.arange(longest)[None, :] + (med - left)[:, None, None]      # [C, 1, longest] sample positions
        lower, upper = self._logits_cumulative_host(grid - 0.5), self._logits_cumulative_host(grid + 0.5)
        flip = -torch.sign(lower + upper)                  # evaluate the difference of sigmoids on the numerically safe side
        mass = torch.abs(torch.sigmoid(flip * upper) - torch.sigmoid(flip * lower))[:, 0, :]
        tails = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        dev = self.quantiles.device
        self._quantized_cdf = self._pmf_to_cdf(mass, tails, length, longest).to(dev)
        self._offset = (-left).to(dev)
        self._cdf_length = (length + 2).to(dev)
        self._tables = None
        return True

    # ---- device side -----------------------------------------------------------------------
    def _noise_like(self, x_nhwc):
        """The reference draws bottleneck noise in [C,1,H*W*B] order (entropy_models.py:426-434); an injected
        noise_source is asked for that shape and re-laid out so parity tests can feed the same numbers."""
        if self.noise_source is not None:
            B, Cc, H, W = x_nhwc.shape
            n = self.noise_source((Cc, 1, H * W * B), x_nhwc.device)
            n = n.reshape(Cc, H, W, B).permute(3, 0, 1, 2)
            return _dense(n.to(x_nhwc.device).contiguous())
        return super()._noise_like(x_nhwc)

    def loss(self):
        """Auxiliary loss (entropy_models.py:383-386); gradient flows to `quantiles` only."""
        return _EBAuxFunction.apply(self.quantiles, self.target, *self._tensors14())

    def forward(self, x):
        xin = F.to_nhwc(x)
        if self.training:
            noise, med = self._noise_like(xin), None
        else:
            noise, med = None, self._medians_vec()
        return _EBFunction.apply(x, noise, med, *self._tensors14())

    @staticmethod
    def _build_indexes(size):
        N, Cc, H, W = size
        indexes = torch.arange(Cc).view(1, -1, 1, 1)
        return indexes.int().repeat(N, 1, H, W)

    def compress(self, x):
        indexes = self._build_indexes(x.size()).to(x.device)
        medians = self._get_medians().detach().expand(x.size(0), -1, 1, 1)
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), size[0], size[1])
        indexes = self._build_indexes(output_size).to(self._quantized_cdf.device)
        medians = self._get_medians().detach().expand(len(strings), -1, 1, 1)
        return super().decompress(strings, indexes, medians)


class _EBAuxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quantiles, target, *tensors14):
        pack = F.eb_pack(list(tensors14))
        loss, dq = F.eb_aux_loss(quantiles, pack, target)
        ctx.save_for_backward(dq)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return (dq * g, None) + (None,) * 14


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = float(scale_table[0])
            self.lower_bound_scale = LowerBound(scale_bound)
            register_bound = None
        elif scale_bound is not None and scale_bound > 0:
            self.lower_bound_scale = LowerBound(scale_bound)
            register_bound = torch.Tensor([float(scale_bound)])
        else:
            raise ValueError("Invalid parameters")
        self._scale_bound = float(scale_bound)
        self.register_buffer("scale_table", self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", register_bound)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(inputs):
        return float(0.5) * torch.erfc(float(-(2 ** -0.5)) * inputs)

    @staticmethod
    def _standardized_quantile(quantile):
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        """Install a scale table and rebuild the CDF tables; a no-op (False) when tables exist and force is not set
        (entropy_models.py:535-541)."""
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = self._prepare_scale_table(scale_table).to(self.scale_table.device)
        self.update()
        return True

    def update(self):
        """Host side, once per model (entropy_models.py:543-568): per table scale the zero-mean Gaussian's mass on the integers
        within the tail_mass quantile, symmetric about 0, as integer CDF rows."""
        sigma = self.scale_table.detach().cpu()
        reach = torch.ceil(sigma * (-self._standardized_quantile(self.tail_mass / 2))).int()      # half width of the support
        length = 2 * reach + 1
        longest = torch.max(length).item()
        dist = torch.abs(torch.arange(longest).int() - reach[:, None]).float()                    # |k - centre|
        s = sigma.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - dist) / s)
        lower = self._standardized_cumulative((-0.5 - dist) / s)
        dev = self.scale_table.device
        self._quantized_cdf = self._pmf_to_cdf(upper - lower, 2 * lower[:, :1], length, longest).to(dev)
        self._offset = (-reach).to(dev)
        self._cdf_length = (length + 2).to(dev)
        self._tables = None

    def forward(self, inputs, scales, means=None):
        if means is None:
            # entropy_models.py:570-596 with means=None: quantize(inputs) without an offset, likelihood of the values themselves --
            # exactly what a zero mean computes (x - 0 and round(x - 0) + 0 are x and round(x) in fp32); the kernel takes the tensor
            means = torch.zeros_like(inputs, memory_format=torch.preserve_format).detach()
        noise = self._noise_like(_dense(inputs.detach())) if self.training else None
        return _GCFunction.apply(inputs, scales, means, noise, self._scale_bound, self._lik_bound if self.use_likelihood_bound else 0.0)

    def build_indexes(self, scales):
        return F.build_indexes(F.to_nhwc(scales.detach()), self.scale_table, self._scale_bound)
