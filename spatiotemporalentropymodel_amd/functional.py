"""Thin Python veneer over the C ABI (include/stem_hip.h): tensor plumbing only.

Activations are torch CUDA tensors of logical shape [B,C,H,W] whose memory is NHWC
("channels_last"), optionally a channel slice of a wider buffer (pixel pitch `ld` > C).
Every function launches hand-written HIP kernels on torch's current stream; nothing here
computes with torch ops.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from . import config as _config

PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_DECONV_FWD, PACK_DECONV_DGRAD, PACK_CONV_FWD_C4 = 0, 1, 2, 3, 4
ACT_NONE, ACT_LRELU = 0, 1
CONV_MASKED_A = 0x100        # OR into `act`: skip the taps a type-A MaskedConv2d zeroes (include/stem_hip.h)
LRELU_SLOPE = 0.01
EB_NPARAM = 58


def _stream():
    """Raw handle of torch's current stream on the current device (the C ABI's `stream` argument).  The private fast
    path costs ~0.3 us instead of ~3 us for torch.cuda.current_stream().cuda_stream -- this runs once per launch."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:          # torch without these private hooks
        return torch.cuda.current_stream().cuda_stream



# ---- stream bookkeeping of the launch schedules.  torch's own helpers resolve device indices through several Python layers
# (torch.cuda.current_stream(device) ~5 us, the `torch.cuda.stream` context manager ~10 us, Stream.wait_stream ~8 us with a
# fresh Event each); a P-frame step asks ~50 times per step, a bench step ~300 times.  Same semantics, fewer layers:
def cur_stream(device=None):
    """torch.cuda.current_stream(device)"""
    try:
        idx = device.index if (device is not None and device.index is not None) else torch._C._cuda_getDevice()
        sid, di, dt = torch._C._cuda_getCurrentStream(idx)
        return torch.cuda.Stream(stream_id=sid, device_index=di, device_type=dt)
    except (AttributeError, TypeError):          # torch without these private hooks
        return torch.cuda.current_stream(device)


class on_stream:
    """`with torch.cuda.stream(s):` for a stream of the current device; on_stream(None) is a no-op"""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        if self.s is not None:
            self.prev = cur_stream(self.s.device)
            torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        if self.s is not None:
            torch.cuda.set_stream(self.prev)
        return False


_WAIT_EVENTS = {}


def stream_wait(dst, src):
    """dst.wait_stream(src): work queued on `dst` from now on starts after everything `src` holds now.  One persistent event per
    stream pair instead of a fresh one per call (a wait refers to the record that precedes it; recording the event again later
    does not move it).  Under stream capture torch's own call is used (captured events must not outlive the capture)."""
    if torch.cuda.is_current_stream_capturing():
        dst.wait_stream(src)
        return
    key = (dst.cuda_stream, src.cuda_stream)
    ev = _WAIT_EVENTS.get(key)
    if ev is None:
        ev = _WAIT_EVENTS[key] = torch.cuda.Event()
    ev.record(src)
    ev.wait(dst)
    if _TAPE is not None:
        _TAPE.entries.append(("wait", dst.cuda_stream, src.cuda_stream))


#: the launch tape that is recording (tape.recording), or None.  The schedule's synchronisation goes through the helpers below so
#: that a recording sees it; outside a recording they are the plain torch calls.
_TAPE = None


def event_record(ev, stream):
    """ev.record(stream) for an event the schedule re-uses every step (a tape replays the record on the same event)"""
    ev.record(stream)
    if _TAPE is not None:
        _TAPE.entries.append(("evrec", ev.cuda_event, stream.cuda_stream))
        _TAPE.keep.append(ev)


def event_wait(stream, ev):
    """stream.wait_event(ev)"""
    stream.wait_event(ev)
    if _TAPE is not None:
        _TAPE.entries.append(("evwait", stream.cuda_stream, ev.cuda_event))
        _TAPE.keep.append(ev)


def tape_py(fn):
    """call fn() now; under a recording also log it, so that a replay calls it at this point of the schedule, with the stream
    that is current now (the data-parallel exchange: torch.distributed calls in the middle of backward).  Whatever torch ops fn
    issues are fn's own business -- a replay re-runs them -- so the recording's foreign-op guard looks away meanwhile."""
    if _TAPE is None:
        return fn()
    _TAPE.entries.append(("py", fn, cur_stream()))
    _TAPE.in_py += 1
    try:
        return fn()
    finally:
        _TAPE.in_py -= 1


def tape_bind_floats(provider):
    """Under a recording: the launch just recorded takes its float arguments from `provider()` (a tuple, in the order of the
    entry point's float / double parameters) on every replay -- values a caller may edit between steps (the optimiser's
    param_groups: stem/trainSTEM.py:123,290).  No-op otherwise."""
    if _TAPE is not None:
        _TAPE.bind_floats(provider)


def zero_bytes(t):
    """t.zero_() for a dense tensor as a library call on the current stream: recordable (tape.LaunchTape), unlike Tensor.zero_()"""
    _chk(_lib.hip().stem_zero_bytes(t.data_ptr(), t.numel() * t.element_size(), _stream()))
    return t


def copy_d2d(dst, src):
    """dst <- src (same byte count, both dense) as a library call on the current stream: recordable, unlike Tensor.copy_"""
    assert dst.numel() * dst.element_size() == src.numel() * src.element_size()
    _chk(_lib.hip().stem_copy_d2d(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), _stream()))
    return dst


_STREAM_PRIO = None


def make_stream(device, role):
    """A HIP stream for one of the schedule's roles -- "side" (weight gradients, hyper branch, auxiliary loss: work on or next to
    the critical path of a P-frame step) or "latents" (the frozen analysis transform of the NEXT frame: long, throughput-only
    kernels).  STEM_STREAM_PRIO="side=-1,latents=0" maps roles to HIP stream priorities (lower = more urgent; torch's default
    stream has 0): with it the command processor dispatches the step's own small kernels ahead of the prefetch stream's."""
    global _STREAM_PRIO
    if _STREAM_PRIO is None:
        _STREAM_PRIO = {}
        for kv in _config.runtime().stream_prio.split(","):
            if "=" in kv:
                k, v = kv.split("=")
                _STREAM_PRIO[k.strip()] = int(v)
    mask = _cu_mask(role)
    if mask is not None:
        try:
            return _masked_stream(device, mask)
        except (RuntimeError, OSError, AttributeError) as e:      # a scheduling hint, not a result: run unmasked rather than not at all
            import warnings
            warnings.warn(f"CU mask for the '{role}' stream not applied ({e}); using an unmasked stream")
    prio = _STREAM_PRIO.get(role, _STREAM_PRIO.get("side", 0) if role == "branch" else 0)      # "branch" (the hyper path's stream) defaults to side's
    return torch.cuda.Stream(device=device, priority=prio)


def _cu_mask(role):
    """STEM_STREAM_CUMASK="latents=block:96" / "side=tail:96" / "latents=mod8:3": restrict a role's stream to a subset of the 256 CUs
    -- the first n CU bits, the last n, or the bits i with i % 8 < k (whole XCDs if the mask enumerates CUs XCD-interleaved).  Keeps the long
    analysis-transform kernels of the prefetch stream off part of the chip so that the P-frame step's short kernels always find
    free CUs (bench.py's default: latents=block:160).  A masked stream is created at the default priority."""
    for kv in _config.runtime().stream_cumask.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            if k.strip() == role:
                kind, n = v.split(":")
                n = int(n)
                if kind == "block":
                    bits = [i < n for i in range(256)]
                elif kind == "tail":                      # the LAST n CU bits: the complement of block:(256 - n)
                    bits = [i >= 256 - n for i in range(256)]
                else:
                    bits = [(i % 8) < n for i in range(256)]
                words = [sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)]
                return words
    return None


def _masked_stream(device, words):
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    with torch.cuda.device(device):
        rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=device)


def _chk(rc):
    if rc != 0:
        _lib.check(rc)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class tuning:
    """`with F.tuning(fx3_split=3): ...` -- force a plan selector of the library (stem_tuning_set: "fx3_tile", "fx3_split",
    "wg3_split") for the calls inside the block; tests and sweep tools only.  The workspace-size cache of the general fp16
    kernel depends on the split factor and is dropped on entry and exit."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        lib = _lib.hip()
        for k, v in self.kv.items():
            self.old[k] = int(lib.stem_tuning_get(k.encode()))
            _chk(lib.stem_tuning_set(k.encode(), int(v)))
        _WS_GEN_BYTES.clear()
        return self

    def __exit__(self, *exc):
        lib = _lib.hip()
        for k, v in self.old.items():
            if v >= 0:
                lib.stem_tuning_set(k.encode(), v)
        _WS_GEN_BYTES.clear()
        return False


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("STEM HIP kernels need CUDA/ROCm tensors (no CPU fallback exists)")
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"STEM HIP kernels are fp32, got {t.dtype}")


# ----------------------------------------------------------------------------- layout helpers
def nhwc_ld(t: torch.Tensor):
    """Pixel pitch of a [B,C,H,W] tensor laid out NHWC (possibly a channel slice), or None."""
    if t.dim() != 4:
        return None
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if Cc > 1 and sc != 1:
        return None
    ld = None
    if W > 1:
        ld = sw
    elif H > 1:
        ld = sh
    elif B > 1:
        ld = sb
    else:
        ld = Cc
    if ld < Cc:
        return None
    if W > 1 and sw != ld:
        return None
    if H > 1 and sh != W * ld:
        return None
    if B > 1 and sb != H * W * ld:
        return None
    return ld


def empty_nhwc(B, Cc, H, W, device, ld=None):
    """Uninitialised [B,C,H,W] tensor with NHWC memory; ld > C allocates a wider pitch.  (One allocator call: this runs
    for every kernel output, ~500 times per training step.)"""
    if ld is None or ld == Cc:
        return torch.empty_strided((B, Cc, H, W), (H * W * Cc, 1, W * Cc, Cc), device=device, dtype=torch.float32)
    return torch.empty((B, H, W, ld), device=device, dtype=torch.float32).permute(0, 3, 1, 2)[:, :Cc]


def channel_slice(buf: torch.Tensor, c0: int, c1: int):
    """[B,C,H,W] NHWC buffer -> view of channels [c0,c1) sharing memory (replaces torch.cat / chunk)."""
    return buf[:, c0:c1]


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return t unchanged if its memory already is NHWC, else convert with the HIP transpose kernel."""
    _require_cuda(t)
    if nhwc_ld(t) is not None:
        return t
    src = t.contiguous()          # NCHW
    B, Cc, H, W = src.shape
    out = empty_nhwc(B, Cc, H, W, t.device)
    _chk(_lib.hip().stem_nchw_to_nhwc(src.data_ptr(), out.data_ptr(), Cc, B, Cc, H, W, _stream()))
    return out


def to_nchw(t: torch.Tensor, clamp01: bool = False) -> torch.Tensor:
    """NHWC-memory tensor -> contiguous NCHW tensor (optionally clamped to [0,1], getX: priors.py:399)."""
    _require_cuda(t)
    ld = nhwc_ld(t)
    if ld is None:
        raise RuntimeError("to_nchw expects an NHWC-memory tensor")
    B, Cc, H, W = t.shape
    out = torch.empty((B, Cc, H, W), device=t.device, dtype=torch.float32)
    _chk(_lib.hip().stem_nhwc_to_nchw(t.data_ptr(), ld, out.data_ptr(), B, Cc, H, W, int(clamp01), _stream()))
    return out


def copy_channels(src: torch.Tensor, dst: torch.Tensor):
    """dst (a channel slice of a wider NHWC buffer) <- src (any NHWC view), same [B,C,H,W]."""
    assert src.shape == dst.shape
    B, Cc, H, W = src.shape
    _chk(_lib.hip().stem_copy_channels(src.data_ptr(), nhwc_ld(src), dst.data_ptr(), nhwc_ld(dst), B * H * W, Cc, _stream()))
    return dst


def dense_nhwc(t: torch.Tensor) -> torch.Tensor:
    """NHWC tensor whose pixel pitch equals its channel count (copies channel-slice views)."""
    t = to_nhwc(t)
    if nhwc_ld(t) != t.shape[1]:
        t = copy_channels(t, empty_nhwc(*t.shape, t.device))
    return t


def nchw3_to_nhwc4(x: torch.Tensor) -> torch.Tensor:
    _require_cuda(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    assert Cc == 3
    out = torch.empty((B, H, W, 4), device=x.device, dtype=torch.float32)
    # the image's scale record (max |x| per workgroup) rides along as an attribute: the first-layer kernel scales its fp16 split by it
    # (the returned tensor is a temporary of the analysis transform: whoever writes into it afterwards must drop `_stem_q`)
    q = torch.empty(16 + (B * H * W + 1023) // 1024, device=x.device, dtype=torch.float32)
    _chk(_lib.hip().stem_nchw3_to_nhwc4(x.data_ptr(), out.data_ptr(), B, H, W, q.data_ptr(), _stream()))
    out._stem_q = q
    return out


def _nhwc4_record(x4: torch.Tensor) -> torch.Tensor:
    """scale record of an NHWC4 image: the one nchw3_to_nhwc4 attached, or measured here (stem_amax_nhwc)"""
    q = getattr(x4, "_stem_q", None)
    if q is None:
        B, H, W, _ = x4.shape
        q = torch.empty(16 + 256, device=x4.device, dtype=torch.float32)
        _chk(_lib.hip().stem_amax_nhwc(x4.data_ptr(), 4, B * H * W, 4, q.data_ptr(), 256, _stream()))
        x4._stem_q = q
    return q


# ----------------------------------------------------------------------------- weights
def pack_weight(w: torch.Tensor, role: int, masked: int = 0) -> torch.Tensor:
    """w: Conv2d [K,C,R,S] or ConvTranspose2d [C,K,R,S] (contiguous) -> packed copy for `role`."""
    _require_cuda(w)
    w = w.detach().contiguous()
    if role in (PACK_DECONV_FWD, PACK_DECONV_DGRAD):
        Cc, K, R, S = w.shape
    else:
        K, Cc, R, S = w.shape
    n = _lib.hip().stem_packed_weight_elems(K, Cc, R, S, role)
    out = torch.empty(n, device=w.device, dtype=torch.float32)
    _chk(_lib.hip().stem_pack_weight(w.data_ptr(), out.data_ptr(), K, Cc, R, S, role, int(masked), _stream()))
    return out


# ----------------------------------------------------------------------------- convolutions
_WS = {}
_WS_BYTES = {}


def _workspace(kind, dims, device):
    """(ptr, nbytes) of the split-K scratch for this layer shape.  One growing buffer per (device, launching stream):
    kernels of a stream are ordered, so consecutive layers can share it; concurrent streams get their own."""
    key = (kind, dims)
    need = _WS_BYTES.get(key)
    if need is None:
        need = _WS_BYTES[key] = int(_lib.hip().stem_conv_workspace_bytes(kind, *dims))
    if need == 0:
        return 0, 0
    slot = (device, _stream())
    buf = _WS.get(slot)
    if buf is None or buf.numel() * 4 < need:
        # zero-filled: the head of the buffer holds the per-tile arrival counters of the split-K protocol, which every
        # launch leaves at zero again (include/stem_hip.h)
        buf = _WS[slot] = torch.zeros((need + 3) // 4, device=device, dtype=torch.float32)
    return buf.data_ptr(), need


def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def deconv_out_hw(H, W, R, S, stride, pad, opad):
    return (H - 1) * stride - 2 * pad + R + opad, (W - 1) * stride - 2 * pad + S + opad


def conv2d_fwd(x, wp, bias, K, R, S, stride, pad, act=ACT_NONE, out=None, slope=LRELU_SLOPE):
    _require_cuda(x, wp, bias)
    B, Cc, H, W = x.shape
    ldx = nhwc_ld(x)
    assert ldx is not None, "conv2d_fwd: input must be NHWC memory"
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x.device)
    ws, wsb = _workspace(0 | (act & CONV_MASKED_A), (B, H, W, Cc, K, R, S, stride, pad, 0), x.device)
    _chk(_lib.hip().stem_conv2d_fwd(x.data_ptr(), ldx, wp.data_ptr(), _ptr(bias), out.data_ptr(), nhwc_ld(out),
                                    B, H, W, Cc, K, R, S, stride, pad, act, slope, ws, wsb, _stream()))
    return out


def conv2d_fwd_c4(x4, wp, bias, K, R, S, stride, pad, out=None):
    """First analysis layer: x4 is the [B,H,W,4] buffer from nchw3_to_nhwc4."""
    B, H, W, _ = x4.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x4.device)
    _chk(_lib.hip().stem_conv2d_fwd_c4(x4.data_ptr(), wp.data_ptr(), _ptr(bias), out.data_ptr(), nhwc_ld(out),
                                       B, H, W, K, R, S, stride, pad, _stream()))
    return out


def conv2d_dgrad(dy, wp_dgrad, x_shape, K, R, S, stride, pad, xact=None, out=None):
    """dX of Conv2d; xact (the layer input, an LReLU output) fuses the activation backward."""
    B, Cc, H, W = x_shape
    if out is None:
        out = empty_nhwc(B, Cc, H, W, dy.device)
    ws, wsb = _workspace(1, (B, H, W, Cc, K, R, S, stride, pad, 0), dy.device)
    _chk(_lib.hip().stem_conv2d_dgrad(dy.data_ptr(), nhwc_ld(dy), wp_dgrad.data_ptr(), out.data_ptr(), nhwc_ld(out),
                                      _ptr(xact), 0 if xact is None else nhwc_ld(xact), LRELU_SLOPE,
                                      B, H, W, Cc, K, R, S, stride, pad, ws, wsb, _stream()))
    return out


def wgrad_plan(x_shape, K, R, S, stride, pad, deconv=False):
    """(splits, slab-buffer elements) for a weight-gradient launch on this input shape."""
    B, Cc, H, W = x_shape
    lib = _lib.hip()
    if deconv:
        splits = lib.stem_wgrad_splits(B, H, W, K, Cc, R, S)
        npix = B * H * W
    else:
        Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
        splits = lib.stem_wgrad_splits(B, Ho, Wo, Cc, K, R, S)
        npix = B * Ho * Wo
    return splits, int(lib.stem_wgrad_workspace_elems(splits, Cc, K, R, S, npix))


_WGRAD_WS = {}
WGRAD_TABLE_VALID, WGRAD_ACCUMULATE_DB, WGRAD_DEFER_DB = 1, 4, 8
UNPACK_DECONV, UNPACK_ACCUMULATE = 1, 2


def _wgrad_workspace(key, elems, device):
    """Slab / partial-sum / gather-table scratch of one layer geometry.  Kernels of a stream are ordered, so every layer
    with this geometry shares the buffer, and its gather table (a function of the geometry only) is built once."""
    key = key + (_stream(),)                 # one buffer per launching stream: ordering is only guaranteed within a stream
    hit = _WGRAD_WS.get(key)
    if hit is not None and hit.numel() >= elems:
        return hit, True
    buf = _WGRAD_WS[key] = torch.empty(elems, device=device, dtype=torch.float32)
    return buf, False


def _deferred_bias(lib, x, dy, dwp, db, K, Cc, R, S, splits, flags, acc, deconv):
    """descriptor of the bias gradient's pending second stage after a WGRAD_DEFER_DB call (parts behind the slabs)"""
    B, _, Ho, Wo = dy.shape
    parts = int(lib.stem_wgrad_bias_parts(x.data_ptr(), nhwc_ld(x), dy.data_ptr(), nhwc_ld(dy), B * Ho * Wo, Cc, K, splits, flags, int(deconv)))
    return _lib.BiasFinalDesc(dwp.data_ptr() + 4 * splits * R * S * K * Cc, db.data_ptr(), K, parts, int(acc), 0)


def conv2d_wgrad(x, dy, K, R, S, stride, pad, dw_out=None, db_out=None, need_db=True, dwp=None, unpack=True, table_valid=False,
                 accumulate=False, accumulate_db=False, defer_bias=False):
    """-> (dW [K,C,R,S], db [K]) in the reference's layouts.  With unpack=False only the packed slabs in `dwp`
    are produced (the caller sums/transposes all layers at once with unpack_wgrads_multi).  accumulate=True adds into
    dw_out / db_out (gradient accumulation over several backward passes)."""
    B, Cc, H, W = x.shape
    lib = _lib.hip()
    splits, elems = wgrad_plan(x.shape, K, R, S, stride, pad)
    if dwp is None:
        dwp, table_valid = _wgrad_workspace((x.device, 0, tuple(x.shape), nhwc_ld(x), K, R, S, stride, pad), elems, x.device)
    assert not accumulate or (dw_out is not None and (db_out is not None or not need_db))
    db = (db_out if db_out is not None else torch.empty(K, device=x.device, dtype=torch.float32)) if need_db else None
    assert not accumulate_db or db_out is not None
    flags = (WGRAD_TABLE_VALID if table_valid else 0) | (WGRAD_ACCUMULATE_DB if (accumulate or accumulate_db) else 0)
    defer_bias = defer_bias and db is not None and not unpack
    if defer_bias:
        flags |= WGRAD_DEFER_DB
    _chk(lib.stem_conv2d_wgrad(x.data_ptr(), nhwc_ld(x), dy.data_ptr(), nhwc_ld(dy), dwp.data_ptr(), _ptr(db),
                               B, H, W, Cc, K, R, S, stride, pad, splits, flags, _stream()))
    if defer_bias:          # -> (None, descriptor): the caller runs the second stage (bias_grad_final_multi)
        return None, _deferred_bias(lib, x, dy, dwp, db, K, Cc, R, S, splits, flags, accumulate or accumulate_db, False)
    if not unpack:
        return None, db
    dw = dw_out if dw_out is not None else torch.empty((K, Cc, R, S), device=x.device, dtype=torch.float32)
    _chk(lib.stem_unpack_wgrad(dwp.data_ptr(), dw.data_ptr(), K, Cc, R, S, splits, UNPACK_ACCUMULATE if accumulate else 0, _stream()))
    return dw, db


def deconv2d_fwd(x, wp, bias, K, R, S, stride, pad, opad, act=ACT_NONE, out=None, slope=LRELU_SLOPE):
    _require_cuda(x, wp, bias)
    B, Cc, H, W = x.shape
    Ho, Wo = deconv_out_hw(H, W, R, S, stride, pad, opad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x.device)
    ws, wsb = _workspace(2, (B, H, W, Cc, K, R, S, stride, pad, opad), x.device)
    _chk(_lib.hip().stem_deconv2d_fwd(x.data_ptr(), nhwc_ld(x), wp.data_ptr(), _ptr(bias), out.data_ptr(), nhwc_ld(out),
                                      B, H, W, Cc, K, R, S, stride, pad, opad, act, slope, ws, wsb, _stream()))
    return out


def deconv2d_dgrad(dy, wp_dgrad, x_shape, K, R, S, stride, pad, opad, xact=None, out=None):
    B, Cc, H, W = x_shape
    if out is None:
        out = empty_nhwc(B, Cc, H, W, dy.device)
    ws, wsb = _workspace(3, (B, H, W, Cc, K, R, S, stride, pad, opad), dy.device)
    _chk(_lib.hip().stem_deconv2d_dgrad(dy.data_ptr(), nhwc_ld(dy), wp_dgrad.data_ptr(), out.data_ptr(), nhwc_ld(out),
                                        _ptr(xact), 0 if xact is None else nhwc_ld(xact), LRELU_SLOPE,
                                        B, H, W, Cc, K, R, S, stride, pad, opad, ws, wsb, _stream()))
    return out


def deconv2d_wgrad(x, dy, K, R, S, stride, pad, opad, dw_out=None, db_out=None, need_db=True, dwp=None, unpack=True, table_valid=False,
                   accumulate=False, accumulate_db=False, defer_bias=False):
    """-> (dW [C,K,R,S], db [K]) in nn.ConvTranspose2d's layout."""
    B, Cc, H, W = x.shape
    lib = _lib.hip()
    splits, elems = wgrad_plan(x.shape, K, R, S, stride, pad, deconv=True)
    if dwp is None:
        dwp, table_valid = _wgrad_workspace((x.device, 1, tuple(x.shape), nhwc_ld(dy), K, R, S, stride, pad, opad), elems, x.device)
    assert not accumulate or (dw_out is not None and (db_out is not None or not need_db))
    db = (db_out if db_out is not None else torch.empty(K, device=x.device, dtype=torch.float32)) if need_db else None
    assert not accumulate_db or db_out is not None
    flags = (WGRAD_TABLE_VALID if table_valid else 0) | (WGRAD_ACCUMULATE_DB if (accumulate or accumulate_db) else 0)
    defer_bias = defer_bias and db is not None and not unpack
    if defer_bias:
        flags |= WGRAD_DEFER_DB
    _chk(lib.stem_deconv2d_wgrad(x.data_ptr(), nhwc_ld(x), dy.data_ptr(), nhwc_ld(dy), dwp.data_ptr(), _ptr(db),
                                 B, H, W, Cc, K, R, S, stride, pad, opad, splits, flags, _stream()))
    if defer_bias:
        return None, _deferred_bias(lib, x, dy, dwp, db, K, Cc, R, S, splits, flags, accumulate or accumulate_db, True)
    if not unpack:
        return None, db
    dw = dw_out if dw_out is not None else torch.empty((Cc, K, R, S), device=x.device, dtype=torch.float32)
    _chk(lib.stem_unpack_wgrad(dwp.data_ptr(), dw.data_ptr(), K, Cc, R, S, splits,
                               UNPACK_DECONV | (UNPACK_ACCUMULATE if accumulate else 0), _stream()))
    return dw, db


def pack_weights_multi(descs):
    """descs: ctypes array of _lib.PackDesc"""
    _chk(_lib.hip().stem_pack_weights_multi(descs, len(descs), _stream()))       # the array object itself: a launch tape keeps it alive


def unpack_wgrads_multi(descs):
    _chk(_lib.hip().stem_unpack_wgrads_multi(descs, len(descs), _stream()))


def gdn_fwd(x, beta, gamma, inverse=False, beta_min=1e-6, out=None):
    _require_cuda(x, beta, gamma)
    B, Cc, H, W = x.shape
    if out is None:
        out = empty_nhwc(B, Cc, H, W, x.device)
    _chk(_lib.hip().stem_gdn_fwd(x.data_ptr(), nhwc_ld(x), beta.data_ptr(), gamma.data_ptr(), out.data_ptr(), nhwc_ld(out),
                                 B, H, W, Cc, int(inverse), beta_min, _stream()))
    return out


def conv2d_gdn_fwd(x, wp, bias, beta, gamma, K, R, S, stride, pad, inverse=False, beta_min=1e-6, out=None):
    """Conv2d with the following GDN/IGDN fused into the epilogue (inference path)."""
    B, Cc, H, W = x.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x.device)
    _chk(_lib.hip().stem_conv2d_gdn_fwd(x.data_ptr(), nhwc_ld(x), wp.data_ptr(), _ptr(bias), beta.data_ptr(), gamma.data_ptr(),
                                        out.data_ptr(), nhwc_ld(out), B, H, W, Cc, K, R, S, stride, pad, int(inverse), beta_min, _stream()))
    return out


#: bytes per pixel and 32-channel slab of the planes layout: two fp16 planes of 32 values
PLANES_SLAB_BYTES = 128


class F16Planes:
    """An NHWC activation tensor pre-split for the 16-bit matrix cores (csrc/conv_f16x3.hip): every fp32 value v is stored as two
    fp16 numbers whose sum is v * 2^e (to 2^-22 |v|), [pixel][C/32][2][32], followed in the same buffer by the
    tensor's scale record (2^-e and the measured max |v| per producing workgroup: include/stem_hip.h).  Only produced and
    consumed by the split-operand convolution kernels; `shape` is the logical [B,C,H,W].  `channels(c0, c1)` is a view of a
    32-aligned channel range (same storage, same pixel pitch, same record), accepted as an INPUT by conv2d_f16x3_gen."""
    __slots__ = ("data", "shape", "pix_bytes", "byte_offset", "q_offset")

    def __init__(self, data, shape, q_offset, pix_bytes=None, byte_offset=0):
        self.data, self.shape = data, tuple(shape)
        self.pix_bytes = (self.shape[1] // 32) * PLANES_SLAB_BYTES if pix_bytes is None else pix_bytes
        self.byte_offset = byte_offset
        self.q_offset = q_offset

    @property
    def dense(self):
        return self.byte_offset == 0 and self.pix_bytes == (self.shape[1] // 32) * PLANES_SLAB_BYTES

    def data_ptr(self):
        return self.data.data_ptr() + self.byte_offset

    def q_ptr(self):
        """device address of the scale record"""
        return self.data.data_ptr() + self.q_offset

    def channels(self, c0, c1):
        if c0 % 32 or c1 % 32 or not 0 <= c0 < c1 <= self.shape[1]:
            raise ValueError(f"planes views are 32-channel aligned, got [{c0}, {c1}) of {self.shape[1]}")
        B, _, H, W = self.shape
        return F16Planes(self.data, (B, c1 - c0, H, W), self.q_offset, self.pix_bytes, self.byte_offset + (c0 // 32) * PLANES_SLAB_BYTES)

    @staticmethod
    def nbytes(npix, Cc):
        """(payload bytes = offset of the scale record, total bytes) = stem_f16x2_planes_qrec_offset / _planes_bytes"""
        payload = npix * (Cc // 32) * PLANES_SLAB_BYTES
        return payload, payload + (((16 + ((npix + 63) // 64) * ((Cc + 127) // 128)) * 4 + 15) & ~15)

    @staticmethod
    def empty(B, Cc, H, W, device, min_slots=0):
        """min_slots: room for at least this many producer slots in the scale record (a producer whose workgroups do not tile the
        tensor 64 pixels x 128 channels at a time: the four-phase transposed launch)"""
        if Cc % 32:
            raise ValueError(f"the planes layout needs a channel count that is a multiple of 32, got {Cc}")
        payload, total = F16Planes.nbytes(B * H * W, Cc)            # in Python: this runs ~60 times per training step
        if min_slots:
            total = max(total, payload + (((16 + min_slots) * 4 + 15) & ~15))
        return F16Planes(torch.empty(total, device=device, dtype=torch.uint8), (B, Cc, H, W), payload)

    @staticmethod
    def split(x, src_q=None):
        """src_q: data_ptr of a scale record whose slots hold max |x| already (left by the kernel that produced x)"""
        x = to_nhwc(x)
        B, Cc, H, W = x.shape
        out = F16Planes.empty(B, Cc, H, W, x.device)
        _chk(_lib.hip().stem_f16x2_split_nhwc(x.data_ptr(), nhwc_ld(x), out.data.data_ptr(), out.q_ptr(), src_q, B * H * W, Cc, _stream()))
        return out

    @staticmethod
    def split_dact(dy, z, slope, src_q=None):
        """planes of dy * (z > 0 ? 1 : slope): the leaky-ReLU derivative applied while splitting (z = the activated output)"""
        dy, z = to_nhwc(dy), to_nhwc(z)
        B, Cc, H, W = dy.shape
        out = F16Planes.empty(B, Cc, H, W, dy.device)
        _chk(_lib.hip().stem_f16x2_split_dact_nhwc(dy.data_ptr(), nhwc_ld(dy), z.data_ptr(), nhwc_ld(z), float(slope), out.data.data_ptr(),
                                                     out.q_ptr(), src_q, B * H * W, Cc, _stream()))
        return out

    def merge(self):
        assert self.dense, "merge() of a channel view is not implemented"
        B, Cc, H, W = self.shape
        out = empty_nhwc(B, Cc, H, W, self.data.device)
        _chk(_lib.hip().stem_f16x2_merge_nhwc(self.data.data_ptr(), self.q_ptr(), out.data_ptr(), Cc, B * H * W, Cc, _stream()))
        return out

    def record(self):
        """(2^-e, max |v| over the slots) -- synchronises; tests / debugging"""
        n = (self.data.numel() - self.q_offset) // 4
        q = self.data[self.q_offset:].view(torch.float32)[:n].cpu()
        ns = int(q[:1].view(torch.int32)[0])
        return float(q[1]), float(q[16:16 + ns].max()) if ns else 0.0


def pack_weight_f16x2(w: torch.Tensor, flip: bool = False) -> torch.Tensor:
    """torch Conv2d weight [K,C,R,S] -> the chunked, pre-split LDS image conv2d_f16x3_fwd streams; flip=True: the operand of
    the input gradient of a stride-1 convolution (rows = input channels, contraction = output channels, taps mirrored)."""
    _require_cuda(w)
    K, Cc, R, S = w.shape
    N, Cin = (Cc, K) if flip else (K, Cc)
    nbytes = _lib.hip().stem_f16x2_conv_weight_bytes(Cin, R, S)
    if nbytes == 0:
        raise ValueError(f"f16x2 weights need a contraction channel count that is a multiple of 32, got {Cin}")
    out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    fn = _lib.hip().stem_f16x2_pack_conv_weight_flip if flip else _lib.hip().stem_f16x2_pack_conv_weight
    _chk(fn(w.detach().contiguous().data_ptr(), out.data_ptr(), N, Cin, R, S, _stream()))
    return out


def conv2d_f16x3_act(xp: F16Planes, wp, bias, K, R, S, stride, pad, act=False, slope=LRELU_SLOPE, want_planes=False):
    """Conv2d (+ leaky ReLU) of a planes tensor on the 192-column kernel (K <= 192): the large-pixel-count layers of the layer-wise
    models.  -> (NHWC fp32 tensor, planes copy or None)"""
    B, Cc, H, W = xp.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    dev = xp.data.device
    assert xp.dense, "the 192-column kernel takes whole planes tensors"
    y = empty_nhwc(B, K, Ho, Wo, dev)
    yp = F16Planes.empty(B, K, Ho, Wo, dev) if want_planes else None
    _chk(_lib.hip().stem_conv2d_f16x3_fwd_act(xp.data.data_ptr(), xp.q_ptr(), wp.data_ptr(), _ptr(bias), 1 if act else 0, float(slope), y.data_ptr(), nhwc_ld(y),
                                               yp.data.data_ptr() if yp is not None else None, yp.q_ptr() if yp is not None else None,
                                               B, H, W, Cc, K, R, S, stride, pad, _stream()))
    return y, yp


def pack_gdn_gamma_f16x2(gamma: torch.Tensor) -> torch.Tensor:
    """The STORED gamma [N, N] of a GDN -> what conv2d_f16x3_fwd's fused GDN streams: gamma' = max(gamma, 2^-18)^2 - 2^-36
    (parametrizers.py:42-45) packed as the weight image of a 1x1 convolution [N][ceil32(N)].  A handful of small launches:
    callers on a hot path keep the result while gamma does not change (layers._PackCache, role PACK_GDN_GAMMA)."""
    N = gamma.shape[0]
    g = torch.clamp(gamma.detach(), min=2.0 ** -18)
    g = g * g - 2.0 ** -36
    cpad = (N + 31) // 32 * 32
    if cpad != N:
        g = torch.nn.functional.pad(g, (0, cpad - N))
    return pack_weight_f16x2(g.reshape(N, cpad, 1, 1).contiguous())


def conv2d_f16x3_fwd(xp: F16Planes, wp, bias, K, R, S, stride, pad, beta=None, gamma=None, beta_min=1e-6, planes_out=False, gp=None):
    """Conv2d (+ GDN when beta / gamma are given) of a planes tensor with fp32 accuracy on the fp16 matrix cores.
    Returns an NHWC fp32 tensor, or a F16Planes for the next convolution of the chain."""
    B, Cc, H, W = xp.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    dev = xp.data.device
    if planes_out:
        out = F16Planes.empty(B, K, Ho, Wo, dev)
        y, ldy, yp, yq = None, 0, out.data.data_ptr(), out.q_ptr()
    else:
        out = empty_nhwc(B, K, Ho, Wo, dev)
        y, ldy, yp, yq = out.data_ptr(), nhwc_ld(out), None, None
    assert xp.dense, "the analysis-transform kernel takes whole planes tensors"
    if gp is None and gamma is not None:      # gp: a kept pack_gdn_gamma_f16x2(gamma)
        gp = pack_gdn_gamma_f16x2(gamma)
    _chk(_lib.hip().stem_conv2d_f16x3_fwd(xp.data.data_ptr(), xp.q_ptr(), wp.data_ptr(), _ptr(bias), _ptr(beta), _ptr(gp), beta_min,
                                           y, ldy, yp, yq, B, H, W, Cc, K, R, S, stride, pad, _stream()))
    return out


_WS_GEN_BYTES = {}
GEN_EPI_BIAS, GEN_EPI_LRELU, GEN_EPI_DACT = 0, 1, 2


def masked_live_taps(R, S, mask_type="A"):
    """taps of an R x S type-A / type-B masked convolution that are not zeroed (layers.py:21-47): a prefix of the row-major order"""
    return (R // 2) * S + S // 2 + (1 if mask_type == "B" else 0)


def pack_weight_f16x2_gen(w: torch.Tensor, flip: bool = False, out=None, taps: int = 0) -> torch.Tensor:
    """torch Conv2d weight [K,C,R,S] -> the image conv2d_f16x3_gen streams; flip=True packs the operand of the input-gradient of
    a stride-1 convolution (rows = input channels, packed channels = output channels, taps mirrored)."""
    _require_cuda(w)
    K, Cc, R, S = w.shape
    N, Cin = (Cc, K) if flip else (K, Cc)
    nbytes = _lib.hip().stem_f16x2_conv_weight_gen_bytes(N, Cin, R, S)
    if nbytes == 0:
        raise ValueError(f"f16x2 weights need a contraction channel count that is a multiple of 32, got {Cin}")
    if out is None:
        out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    assert out.numel() == nbytes
    wc = w.detach()
    assert taps == 0 or wc.is_contiguous(), "a masked weight is zeroed in place: it must be contiguous"
    _chk(_lib.hip().stem_f16x2_pack_conv_weight_gen(wc.contiguous().data_ptr(), out.data_ptr(), N, Cin, R, S, int(flip), int(taps), _stream()))
    return out


def f16x2_gen_weight_bytes(N, C, R, S):
    return int(_lib.hip().stem_f16x2_conv_weight_gen_bytes(N, C, R, S))


def pack_weights_f16x2_multi(descs):
    """descs: ctypes array of _lib.F16PackDesc (at most 24 per call: the table is a kernel argument)"""
    for i in range(0, len(descs), 24):
        n = min(24, len(descs) - i)
        part = descs if (i == 0 and n == len(descs)) else (_lib.F16PackDesc * n).from_buffer(descs, i * C.sizeof(_lib.F16PackDesc))
        _chk(_lib.hip().stem_f16x2_pack_conv_weights_multi(part, n, _stream()))      # (an array object, not byref: a launch tape keeps it alive)


def pack_weights_f16x2_pair_multi(descs):
    """descs: ctypes array of _lib.F16PairDesc (at most 20 per call): both images of every layer from one read of its weights"""
    for i in range(0, len(descs), 20):
        n = min(20, len(descs) - i)
        part = descs if (i == 0 and n == len(descs)) else (_lib.F16PairDesc * n).from_buffer(descs, i * C.sizeof(_lib.F16PairDesc))
        _chk(_lib.hip().stem_f16x2_pack_conv_weights_pair_multi(part, n, _stream()))


def conv2d_f16x3_gen(xp: F16Planes, wp, bias, N, R, S, stride, pad, epi=GEN_EPI_BIAS, slope=LRELU_SLOPE, z=None, out=None,
                      want_fp32=True, want_planes=False, taps=0, rows=None):
    """General f16x3 convolution (training-time STEM layers): returns (fp32 NHWC tensor or None, F16Planes or None).
    `out` may be a channel slice of a wider NHWC buffer; epi = GEN_EPI_DACT multiplies by the leaky-ReLU derivative at z;
    taps > 0: a masked convolution whose weight image holds only its first `taps` taps (pack_weight_f16x2_gen(..., taps=)).
    rows = (N_image, n0): `wp` holds N_image rows and this call computes its rows [n0, n0 + N) (whole 128-row tiles)."""
    B, Cc, H, W = xp.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    dev = xp.data.device
    y = None
    if want_fp32 or out is not None:
        y = out if out is not None else empty_nhwc(B, N, Ho, Wo, dev)
    yp = F16Planes.empty(B, N, Ho, Wo, dev) if want_planes else None
    dims = (B, H, W, Cc, N, R, S, stride, pad, taps)
    need = _WS_GEN_BYTES.get(dims)
    if need is None:
        need = _WS_GEN_BYTES[dims] = int(_lib.hip().stem_conv2d_f16x3_gen_workspace_bytes(*dims))
    ws_ptr = 0
    if need:
        slot = (dev, _stream())
        buf = _WS.get(slot)
        if buf is None or buf.numel() * 4 < need:
            buf = _WS[slot] = torch.zeros((need + 3) // 4, device=dev, dtype=torch.float32)       # zero head: arrival counters
        ws_ptr = buf.data_ptr()
    if rows is not None:
        _chk(_lib.hip().stem_conv2d_f16x3_gen_fwd_rows(xp.data_ptr(), xp.q_ptr(), xp.pix_bytes, wp.data_ptr(), int(rows[0]), int(rows[1]), _ptr(bias),
                                                        epi, slope, _ptr(z), nhwc_ld(z) if z is not None else 0, _ptr(y),
                                                        nhwc_ld(y) if y is not None else 0, yp.data.data_ptr() if yp is not None else None,
                                                        yp.q_ptr() if yp is not None else None, B, H, W, Cc, N, R, S, stride, pad, taps, ws_ptr, need,
                                                        _stream()))
        return y, yp
    _chk(_lib.hip().stem_conv2d_f16x3_gen_fwd(xp.data_ptr(), xp.q_ptr(), xp.pix_bytes, wp.data_ptr(), _ptr(bias), epi, slope, _ptr(z),
                                               nhwc_ld(z) if z is not None else 0, _ptr(y), nhwc_ld(y) if y is not None else 0,
                                               yp.data.data_ptr() if yp is not None else None, yp.q_ptr() if yp is not None else None,
                                               B, H, W, Cc, N, R, S, stride, pad, taps, ws_ptr, need, _stream()))
    return y, yp


def pack_weight_f16x2_tconv(w: torch.Tensor) -> torch.Tensor:
    """torch weight read as w[c][n][r][s] (an nn.ConvTranspose2d weight [Cin, Cout, R, R] for its forward; an nn.Conv2d weight
    [K, C, R, R] for its input gradient: rows n = the face's outputs, c = its contraction channels) -> the four sub-pixel phase
    images tconv2d_f16x3 streams (one buffer, one scale record)."""
    _require_cuda(w)
    Cin, N, R, S = w.shape
    assert R == S and R % 2 == 1
    nbytes = f16x2_gen_weight_bytes(N, Cin, R, S)
    if nbytes == 0:
        raise ValueError(f"f16x2 weights need a contraction channel count that is a multiple of 32, got {Cin}")
    out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    wc = w.detach().contiguous()
    pack_weights_f16x2_multi((_lib.F16PackDesc * 1)(_lib.F16PackDesc(wc.data_ptr(), out.data_ptr(), N, Cin, R, S, 2, 0)))
    return out


_WS_TCONV_BYTES = {}


def tconv2d_f16x3(xp: F16Planes, wp, bias, N, R, epi=GEN_EPI_BIAS, slope=LRELU_SLOPE, z=None, out=None, want_fp32=True, want_planes=False,
                  fine_hw=None):
    """The transposed face of a stride-2, R x R, padding R // 2 layer on the fp16 matrix cores, one launch over its four sub-pixel
    phases: the forward of nn.ConvTranspose2d(C, N, R, stride=2, padding=R//2, output_padding=1) or the input gradient of
    nn.Conv2d(N, C, R, stride=2, padding=R//2) on an even-sized input.  xp: planes [B, C, H, W] of the coarse tensor; wp:
    pack_weight_f16x2_tconv / the engine's phase images.  -> (fp32 NHWC [B, N, 2H, 2W] or None, planes or None); z (GEN_EPI_DACT):
    the activated tensor of the FINE grid.  fine_hw = (2H - 1 or 2H, 2W - 1 or 2W): the input gradient of a strided convolution
    whose input had odd sizes."""
    B, Cc, H, W = xp.shape
    dev = xp.data.device
    Hf, Wf = fine_hw if fine_hw is not None else (2 * H, 2 * W)
    y = None
    if want_fp32 or out is not None:
        y = out if out is not None else empty_nhwc(B, N, Hf, Wf, dev)
    yp = F16Planes.empty(B, N, Hf, Wf, dev, min_slots=4 * ((B * H * W + 63) // 64) * ((N + 127) // 128)) if want_planes else None
    dims = (B, H, W, Cc, N, R)
    need = _WS_TCONV_BYTES.get(dims)
    if need is None:
        need = _WS_TCONV_BYTES[dims] = int(_lib.hip().stem_tconv2d_f16x3_workspace_bytes(*dims))
    ws_ptr = 0
    if need:
        slot = (dev, _stream())
        buf = _WS.get(slot)
        if buf is None or buf.numel() * 4 < need:
            buf = _WS[slot] = torch.zeros((need + 3) // 4, device=dev, dtype=torch.float32)       # zero head: arrival counters
        ws_ptr = buf.data_ptr()
    _chk(_lib.hip().stem_tconv2d_f16x3_fwd(xp.data_ptr(), xp.q_ptr(), xp.pix_bytes, wp.data_ptr(), _ptr(bias), epi, slope, _ptr(z),
                                            nhwc_ld(z) if z is not None else 0, _ptr(y), nhwc_ld(y) if y is not None else 0,
                                            yp.data.data_ptr() if yp is not None else None, yp.q_ptr() if yp is not None else None,
                                            B, H, W, Cc, N, R, Hf, Wf, ws_ptr, need, _stream()))
    return y, yp


def wgrad_f16x3_strided_plan(f_shape, K, R, S, stride, pad):
    """(splits, slab elements) of conv2d_wgrad_f16x3_strided; f_shape: the FINE-grid operand [B, C, H, W]"""
    B, Cc, H, W = f_shape
    splits = int(_lib.hip().stem_wgrad_f16x3_strided_splits(B, H, W, Cc, K, R, S, stride, pad))
    return splits, splits * R * S * K * Cc


def conv2d_wgrad_f16x3_strided(fp: F16Planes, gp: F16Planes, K, R, S, stride, pad, dwp, splits, bias_part=None):
    """packed weight-gradient slabs [splits][R*S][K][C] of a strided layer from planes operands: gp [B, K, OH, OW] on the coarse
    grid, fp [B, C, H, W] on the fine grid (nn.Conv2d: fp = input, gp = output gradient; nn.ConvTranspose2d: gp = the layer's
    input, fp = the gradient of its output -- the slabs then are [t][Cin][Cout]).  bias_part (splits * K floats): per-split column
    sums of gp (the Conv2d's bias gradient, first stage)."""
    B, Cc, H, W = fp.shape
    _chk(_lib.hip().stem_conv2d_wgrad_f16x3_strided(fp.data_ptr(), fp.q_ptr(), fp.pix_bytes, gp.data_ptr(), gp.q_ptr(), gp.pix_bytes, dwp.data_ptr(),
                                                     _ptr(bias_part), B, H, W, Cc, K, R, S, stride, pad, splits, _stream()))


def wgrad_f16x3_plan(x_shape, K, R, S, pad):
    """(splits, slab elements) of conv2d_wgrad_f16x3 for this geometry"""
    B, Cc, H, W = x_shape
    splits = int(_lib.hip().stem_wgrad_f16x3_splits(B, H, W, Cc, K, R, S, pad))
    return splits, splits * R * S * K * Cc


def conv2d_wgrad_f16x3_into(xp: F16Planes, dyp: F16Planes, K, R, S, pad, dw_out, db_out=None, accumulate=True):
    """weight (and bias) gradient of a stride-1 convolution from planes operands straight into the reference-layout
    buffers dw_out [K,C,R,S] / db_out [K] (accumulating: autograd's `.grad +=`); slabs from the per-geometry workspace"""
    B, Cc, H, W = xp.shape
    splits, elems = wgrad_f16x3_plan(xp.shape, K, R, S, pad)
    ws, _ = _wgrad_workspace(("f16x3", xp.data.device, tuple(xp.shape), K, R, S, pad), elems + splits * K, xp.data.device)
    dwp, bpart = ws[:elems], ws[elems:elems + splits * K]
    conv2d_wgrad_f16x3(xp, dyp, K, R, S, pad, dwp, splits, db=db_out, bias_part=bpart, accumulate_db=accumulate)
    _chk(_lib.hip().stem_unpack_wgrad(dwp.data_ptr(), dw_out.data_ptr(), K, Cc, R, S, splits, UNPACK_ACCUMULATE if accumulate else 0, _stream()))


def bias_grad_final_multi(descs):
    """descs: list of _lib.BiasFinalDesc -- every pending second stage of a module group's bias gradients with one launch"""
    for i in range(0, len(descs), 24):
        chunk = descs[i:i + 24]
        _chk(_lib.hip().stem_bias_grad_final_multi((_lib.BiasFinalDesc * len(chunk))(*chunk), len(chunk), _stream()))


def conv2d_wgrad_f16x3(xp: F16Planes, dyp: F16Planes, K, R, S, pad, dwp, splits, db=None, bias_part=None, accumulate_db=False, defer_bias=False):
    """packed weight-gradient slabs [splits][R*S][K][C] of a stride-1 convolution from planes operands (channel views allowed).
    With `db` (and a `bias_part` scratch of splits * K floats) the bias gradient comes out of the same pass: the kernel leaves
    per-split column sums of dy, a second tiny launch adds them into db."""
    B, Cc, H, W = xp.shape
    if db is not None and bias_part is None:
        bias_part = torch.empty(splits * K, device=dwp.device, dtype=torch.float32)
    _chk(_lib.hip().stem_conv2d_wgrad_f16x3(xp.data_ptr(), xp.q_ptr(), xp.pix_bytes, dyp.data_ptr(), dyp.q_ptr(), dyp.pix_bytes, dwp.data_ptr(),
                                             _ptr(bias_part) if db is not None else None,
                                             B, H, W, Cc, K, R, S, pad, splits, _stream()))
    if db is not None and defer_bias:           # the caller sums the per-split column sums later (bias_grad_final_multi)
        return _lib.BiasFinalDesc(bias_part.data_ptr(), db.data_ptr(), K, splits, int(accumulate_db), 0)
    if db is not None:
        _chk(_lib.hip().stem_bias_grad_final(bias_part.data_ptr(), K, splits, db.data_ptr(), int(accumulate_db), _stream()))
    return None


_BIAS_SCRATCH = {}


def bias_grad(dy, db, accumulate=False):
    """db (+)= column sums of the NHWC tensor dy (the Conv2d bias gradient), deterministic two-stage reduction"""
    B, K, H, W = dy.shape
    key = (dy.device, _stream(), B * H * W, K)
    sc = _BIAS_SCRATCH.get(key)
    if sc is None:
        sc = _BIAS_SCRATCH[key] = torch.empty(int(_lib.hip().stem_bias_grad_scratch_elems(B * H * W, K)), device=dy.device, dtype=torch.float32)
    _chk(_lib.hip().stem_bias_grad(dy.data_ptr(), nhwc_ld(dy), B * H * W, K, sc.data_ptr(), db.data_ptr(), int(accumulate), _stream()))
    return db


def _aligned16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def c4gdn_supported(K, R, S, inverse=False):
    """first layer + GDN on the fp16 kernel of csrc/c4gdn_f16x3.hip: N = 64 / 128 / 192 output channels, filters up to 25 taps;
    STEM_C4GDN_F16X3=0 keeps the fp32-MFMA kernel of igemm.hip (routing switch: both meet the 1e-4 gates; the fp16 form carries ~2^-21 relative per product, the fp32 instruction 2^-24)"""
    return (not inverse and _config.runtime().first_layer_f16x3 and bool(_lib.hip().stem_c4gdn_supported(K, R, S)))


def _c4gdn_fits(x4, K, R, S, stride, pad, ld=None, planes=False):
    """operands of the fp16 first-layer kernel are addressed through 2 GiB buffer views"""
    B, H, W, _ = x4.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    out_bytes = B * Ho * Wo * ((K // 32) * PLANES_SLAB_BYTES if planes else (ld or K) * 4)
    return B * H * W * 16 < 0x7FFFFF00 and out_bytes < 0x7FFF0000


def c4gdn_stream(wp_c4, gamma, K, R, S):
    """A-operand stream of conv2d_c4_gdn_f16x3: the C4-packed first-layer weight and the reparametrised gamma of the following
    GDN, split into fp16 planes in MFMA-fragment order (one small launch; cache it while the parameters do not change)."""
    out = torch.empty(int(_lib.hip().stem_c4gdn_stream_bytes(K, R, S)), device=wp_c4.device, dtype=torch.uint8)
    _chk(_lib.hip().stem_c4gdn_pack(wp_c4.data_ptr(), gamma.data_ptr(), out.data_ptr(), K, R, S, _stream()))
    return out


def conv2d_c4_gdn_f16x3(x4, astream, bias, beta, K, R, S, stride, pad, beta_min=1e-6, out=None, planes_out=False):
    """conv (3 -> K channels, x4 = [B,H,W,4] from nchw3_to_nhwc4) + GDN in one kernel, three fp16 MFMAs per fp32 product;
    result as an NHWC fp32 tensor or, with planes_out, pre-split for conv2d_f16x3_fwd"""
    B, H, W, _ = x4.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    if planes_out:
        res = F16Planes.empty(B, K, Ho, Wo, x4.device)
        y, ldy, yp, yq = None, 0, res.data.data_ptr(), res.q_ptr()
    else:
        res = out if out is not None else empty_nhwc(B, K, Ho, Wo, x4.device)
        y, ldy, yp, yq = res.data_ptr(), nhwc_ld(res), None, None
    _chk(_lib.hip().stem_conv2d_c4_gdn_f16x3(x4.data_ptr(), _nhwc4_record(x4).data_ptr(), astream.data_ptr(), _ptr(bias), beta.data_ptr(), beta_min,
                                              y, ldy, yp, yq, B, H, W, K, R, S, stride, pad, _stream()))
    return res


def conv2d_fwd_c4_gdn_planes(x4, wp, bias, beta, gamma, K, R, S, stride, pad, beta_min=1e-6, astream=None):
    """conv2d_fwd_c4_gdn whose result is handed to conv2d_f16x3_fwd: written pre-split (F16Planes), no fp32 copy.
    `astream`: a cached c4gdn_stream(wp, gamma, ...) (built per call otherwise)."""
    if c4gdn_supported(K, R, S) and _aligned16(bias, beta) and _c4gdn_fits(x4, K, R, S, stride, pad, planes=True):
        return conv2d_c4_gdn_f16x3(x4, astream if astream is not None else c4gdn_stream(wp, gamma, K, R, S), bias, beta, K, R, S,
                                    stride, pad, beta_min, planes_out=True)
    # channel counts the one-kernel form does not cover: the fp32-MFMA kernel, then a split pass
    return F16Planes.split(conv2d_fwd_c4_gdn(x4, wp, bias, beta, gamma, K, R, S, stride, pad, beta_min=beta_min))


def conv2d_fwd_c4_gdn(x4, wp, bias, beta, gamma, K, R, S, stride, pad, inverse=False, beta_min=1e-6, out=None, astream=None):
    if (c4gdn_supported(K, R, S, inverse) and _aligned16(bias, beta) and (out is None or _aligned16(out))
            and _c4gdn_fits(x4, K, R, S, stride, pad, ld=nhwc_ld(out) if out is not None else K)):
        return conv2d_c4_gdn_f16x3(x4, astream if astream is not None else c4gdn_stream(wp, gamma, K, R, S), bias, beta, K, R, S,
                                    stride, pad, beta_min, out=out)
    B, H, W, _ = x4.shape
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x4.device)
    _chk(_lib.hip().stem_conv2d_fwd_c4_gdn(x4.data_ptr(), wp.data_ptr(), _ptr(bias), beta.data_ptr(), gamma.data_ptr(), out.data_ptr(),
                                           nhwc_ld(out), B, H, W, K, R, S, stride, pad, int(inverse), beta_min, _stream()))
    return out


def deconv2d_gdn_fwd(x, wp, bias, beta, gamma, K, R, S, stride, pad, opad, inverse=True, beta_min=1e-6, out=None):
    B, Cc, H, W = x.shape
    Ho, Wo = deconv_out_hw(H, W, R, S, stride, pad, opad)
    if out is None:
        out = empty_nhwc(B, K, Ho, Wo, x.device)
    _chk(_lib.hip().stem_deconv2d_gdn_fwd(x.data_ptr(), nhwc_ld(x), wp.data_ptr(), _ptr(bias), beta.data_ptr(), gamma.data_ptr(),
                                          out.data_ptr(), nhwc_ld(out), B, H, W, Cc, K, R, S, stride, pad, opad, int(inverse),
                                          beta_min, _stream()))
    return out


def gdn_bwd(x, dy, beta, gamma, inverse=False, beta_min=1e-6):
    """-> (dx, dbeta, dgamma): gradients wrt the input and the STORED (reparametrised) parameters."""
    B, Cc, H, W = x.shape
    lib = _lib.hip()
    nbytes = int(lib.stem_gdn_bwd_workspace_bytes(B, H, W, Cc))
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    dx = empty_nhwc(B, Cc, H, W, x.device)
    dbeta, dgamma = torch.empty_like(beta), torch.empty_like(gamma)
    _chk(lib.stem_gdn_bwd(x.data_ptr(), nhwc_ld(x), dy.data_ptr(), nhwc_ld(dy), beta.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                          nhwc_ld(dx), dbeta.data_ptr(), dgamma.data_ptr(), B, H, W, Cc, int(inverse), beta_min, ws.data_ptr(),
                          nbytes, _stream()))
    return dx, dbeta, dgamma


def lrelu_bwd(yact, dy, slope=LRELU_SLOPE):
    assert nhwc_ld(yact) == yact.shape[1] and nhwc_ld(dy) == dy.shape[1]
    out = empty_nhwc(*yact.shape, yact.device)
    _chk(_lib.hip().stem_lrelu_bwd(yact.data_ptr(), dy.data_ptr(), out.data_ptr(), yact.numel(), slope, _stream()))
    return out


def lrelu_fwd(x, slope=LRELU_SLOPE):
    assert nhwc_ld(x) == x.shape[1]
    out = empty_nhwc(*x.shape, x.device)
    _chk(_lib.hip().stem_lrelu_fwd(x.data_ptr(), out.data_ptr(), x.numel(), slope, _stream()))
    return out


def sft_fwd(x, gamma, beta, slope=1.0):
    """act(x * (1 + gamma) + beta) on dense NHWC tensors (stem_utils.py:41); slope 1.0 = no activation."""
    assert nhwc_ld(x) == x.shape[1] and nhwc_ld(gamma) == x.shape[1] and nhwc_ld(beta) == x.shape[1]
    out = empty_nhwc(*x.shape, x.device)
    _chk(_lib.hip().stem_sft_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), x.numel(), slope, _stream()))
    return out


def sft_bwd(x, gamma, out, dout, slope=1.0):
    dx, dg, db = (empty_nhwc(*x.shape, x.device) for _ in range(3))
    _chk(_lib.hip().stem_sft_bwd(x.data_ptr(), gamma.data_ptr(), out.data_ptr(), dout.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                 db.data_ptr(), x.numel(), slope, _stream()))
    return dx, dg, db


def avgpool(x, Ho, Wo):
    """adaptive_avg_pool2d for integer ratios (quality map -> feature resolution)."""
    B, Cc, H, W = x.shape
    if (H, W) == (Ho, Wo):
        return x
    out = empty_nhwc(B, Cc, Ho, Wo, x.device)
    _chk(_lib.hip().stem_avgpool_fwd(x.data_ptr(), nhwc_ld(x), out.data_ptr(), Cc, B, H, W, Cc, Ho, Wo, _stream()))
    return out


def avgpool_bwd(dy, H, W):
    B, Cc, Ho, Wo = dy.shape
    if (H, W) == (Ho, Wo):
        return dy
    dx = empty_nhwc(B, Cc, H, W, dy.device)
    _chk(_lib.hip().stem_avgpool_bwd(dy.data_ptr(), nhwc_ld(dy), dx.data_ptr(), Cc, B, H, W, Cc, Ho, Wo, _stream()))
    return dx


def weighted_sqerr_sum(xhat, x, lam):
    """sum(lam * (xhat - x)^2) over contiguous NCHW images, lam [B,1,H,W] -> fp64 0-dim tensor."""
    B, Cc, H, W = xhat.shape
    assert xhat.is_contiguous() and x.is_contiguous() and lam.is_contiguous() and lam.shape == (B, 1, H, W) and x.shape == xhat.shape
    acc = torch.zeros(1, dtype=torch.float64, device=xhat.device)
    _chk(_lib.hip().stem_weighted_sqerr_sum(xhat.data_ptr(), x.data_ptr(), lam.data_ptr(), B, Cc, H * W, acc.data_ptr(), _stream()))
    return acc.reshape(())


def weighted_sqerr_bwd(xhat, x, lam, g, coef):
    B, Cc, H, W = xhat.shape
    g = g.to(torch.float64).contiguous()
    out = torch.empty_like(xhat)
    _chk(_lib.hip().stem_weighted_sqerr_bwd(xhat.data_ptr(), x.data_ptr(), lam.data_ptr(), out.data_ptr(), B, Cc, H * W, g.data_ptr(),
                                            float(coef), _stream()))
    return out


# ----------------------------------------------------------------------------- entropy models
EB_TENSORS = [f"_{k}{i}" for i in range(5) for k in (("matrix", "bias", "factor") if i < 4 else ("matrix", "bias"))]


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def eb_pack(tensors14):
    """14 EntropyBottleneck tensors in EB_TENSORS order -> [C,58] pack."""
    Cc = tensors14[0].shape[0]
    ts = [t.detach().contiguous() for t in tensors14]
    pack = torch.empty((Cc, EB_NPARAM), device=ts[0].device, dtype=torch.float32)
    _chk(_lib.hip().stem_eb_pack(_ptr_array(ts), pack.data_ptr(), Cc, _stream()))
    return pack


def eb_unpack_grads(dpack, grads14, accumulate=False):
    Cc = dpack.shape[0]
    _chk(_lib.hip().stem_eb_unpack_grads(dpack.data_ptr(), _ptr_array(grads14), Cc, 1 if accumulate else 0, _stream()))


def eb_forward(z, pack, medians=None, noise=None, bound=1e-9):
    """-> (z_hat, lik) dense NHWC tensors.  noise given -> training mode, else round around medians."""
    B, Cc, H, W = z.shape
    z_hat, lik = empty_nhwc(B, Cc, H, W, z.device), empty_nhwc(B, Cc, H, W, z.device)
    mode = 0 if noise is not None else 1
    if noise is not None:
        assert nhwc_ld(noise) == Cc
    _chk(_lib.hip().stem_eb_forward(z.data_ptr(), nhwc_ld(z), _ptr(noise), pack.data_ptr(), _ptr(medians), z_hat.data_ptr(),
                                    lik.data_ptr(), B, H, W, Cc, mode, bound, _stream()))
    return z_hat, lik


def eb_backward(z_hat, pack, dlik, dzhat_in=None, bound=1e-9, record=False):
    """record=True: -> (dz, dpack, q) with q the scale record of dz (one slot per channel) for F16Planes.split(dz, src_q=)"""
    B, Cc, H, W = z_hat.shape
    dz = empty_nhwc(B, Cc, H, W, z_hat.device)
    dpack = torch.empty_like(pack)
    if record:
        q = torch.empty(16 + Cc, device=z_hat.device, dtype=torch.float32)
        _chk(_lib.hip().stem_eb_backward_rec(z_hat.data_ptr(), pack.data_ptr(), dlik.data_ptr(), _ptr(dzhat_in), dz.data_ptr(),
                                             dpack.data_ptr(), B, H, W, Cc, bound, q.data_ptr(), _stream()))
        return dz, dpack, q
    _chk(_lib.hip().stem_eb_backward(z_hat.data_ptr(), pack.data_ptr(), dlik.data_ptr(), _ptr(dzhat_in), dz.data_ptr(),
                                     dpack.data_ptr(), B, H, W, Cc, bound, _stream()))
    return dz, dpack


def eb_aux_loss(quantiles, pack, target, need_grad=True):
    Cc = quantiles.shape[0]
    loss = torch.empty(1, device=quantiles.device, dtype=torch.float32)
    dq = torch.empty_like(quantiles) if need_grad else None
    _chk(_lib.hip().stem_eb_aux_loss(quantiles.detach().contiguous().data_ptr(), pack.data_ptr(), target.data_ptr(),
                                     loss.data_ptr(), _ptr(dq), Cc, _stream()))
    return loss, dq


def gc_forward(y, scales, means, noise=None, scale_bound=0.11, lik_bound=1e-9):
    """y dense NHWC; scales/means channel slices with a common pitch -> (out, lik)."""
    B, Cc, H, W = y.shape
    assert nhwc_ld(y) == Cc
    ldsm = nhwc_ld(scales)
    assert ldsm == nhwc_ld(means)
    out, lik = empty_nhwc(B, Cc, H, W, y.device), empty_nhwc(B, Cc, H, W, y.device)
    mode = 0 if noise is not None else 1
    _chk(_lib.hip().stem_gc_forward(y.data_ptr(), _ptr(noise), scales.data_ptr(), means.data_ptr(), ldsm, out.data_ptr(),
                                    lik.data_ptr(), B * H * W, Cc, mode, scale_bound, lik_bound, _stream()))
    return out, lik


def qrec_for(nelem, device):
    """an (uninitialised) scale record for an elementwise producer of `nelem` outputs, 256 per workgroup (stem_common.h)"""
    return torch.empty(16 + (nelem + 255) // 256, device=device, dtype=torch.float32)


def gc_backward(out, scales, means, dlik, dscales, dmeans, dy=None, scale_bound=0.11, lik_bound=1e-9, record=False):
    """record=True: also returns the scale record of (dscales | dmeans) for F16Planes.split(..., src_q=)"""
    B, Cc, H, W = out.shape
    ldd = nhwc_ld(dscales)
    assert ldd == nhwc_ld(dmeans)
    q = qrec_for(B * H * W * Cc, out.device) if record else None
    _chk(_lib.hip().stem_gc_backward(out.data_ptr(), scales.data_ptr(), means.data_ptr(), nhwc_ld(scales), dlik.data_ptr(),
                                     dscales.data_ptr(), dmeans.data_ptr(), ldd, _ptr(dy), B * H * W, Cc, scale_bound,
                                     lik_bound, _ptr(q), _stream()))
    return q


def log2_sum(lik, acc):
    """acc (float64[1]) += sum(log2(lik)); lik must be dense."""
    _chk(_lib.hip().stem_log2_sum(lik.data_ptr(), lik.numel(), acc.data_ptr(), _stream()))


def dlog(lik, coef):
    out = torch.empty_like(lik)
    _chk(_lib.hip().stem_dlog(lik.data_ptr(), out.data_ptr(), lik.numel(), coef, _stream()))
    return out


def _dense_like(a):
    B, Cc, H, W = a.shape
    return empty_nhwc(B, Cc, H, W, a.device)


def sub(a, b):
    assert nhwc_ld(a) == a.shape[1] and nhwc_ld(b) == b.shape[1]
    out = _dense_like(a)
    _chk(_lib.hip().stem_sub(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


def add(a, b):
    assert nhwc_ld(a) == a.shape[1] and nhwc_ld(b) == b.shape[1]
    out = _dense_like(a)
    _chk(_lib.hip().stem_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


def round_(a):
    assert nhwc_ld(a) == a.shape[1]
    out = _dense_like(a)
    _chk(_lib.hip().stem_round(a.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


NOISE_EPOCH_STRIDE = 1 << 40          # counter distance between two epochs of a device-counted noise stream


def uniform_noise_like(a, seed: int, offset: int, epoch=None):
    """U(-1/2, 1/2) from the Philox stream (seed, offset).  `epoch` (0-dim int64 device tensor) adds epoch * 2^40 to the
    counter on the device: inside a captured hipGraph the host-side `offset` is frozen, the epoch is what advances."""
    out = _dense_like(a)
    if epoch is None:
        _chk(_lib.hip().stem_uniform_noise(out.data_ptr(), out.numel(), seed & (2 ** 64 - 1), offset & (2 ** 64 - 1), _stream()))
    else:
        assert epoch.dtype == torch.int64 and epoch.is_cuda
        _chk(_lib.hip().stem_uniform_noise_epoch(out.data_ptr(), out.numel(), seed & (2 ** 64 - 1), offset & (2 ** 64 - 1),
                                                 epoch.data_ptr(), NOISE_EPOCH_STRIDE, _stream()))
    return out


def counter_add_(ctr, inc=1):
    """ctr (0-dim / 1-element int64 device tensor) += inc, as a kernel on the current stream (graph-capturable)"""
    assert ctr.dtype == torch.int64 and ctr.is_cuda
    _chk(_lib.hip().stem_counter_add(ctr.data_ptr(), int(inc), _stream()))
    return ctr


def build_indexes(scales, table, scale_bound=0.11):
    B, Cc, H, W = scales.shape
    idx = torch.empty((B, H, W, Cc), device=scales.device, dtype=torch.int32).permute(0, 3, 1, 2)
    _chk(_lib.hip().stem_build_indexes(scales.data_ptr(), nhwc_ld(scales), table.data_ptr(), table.numel(), idx.data_ptr(),
                                       B * H * W, Cc, scale_bound, _stream()))
    return idx


SUMSQ_SCRATCH = 2048          # include/stem_hip.h: `acc` holds 1 + SUMSQ_SCRATCH doubles, acc[0] is the running sum


def sumsq_accumulator(device):
    return torch.zeros(1 + SUMSQ_SCRATCH, dtype=torch.float64, device=device)


def sumsq(g, acc, overwrite=False):
    """acc[0] += sum(g^2) (or = with overwrite=True: no separately zeroed accumulator)"""
    assert acc.numel() >= 1 + SUMSQ_SCRATCH
    fn = _lib.hip().stem_sumsq_set if overwrite else _lib.hip().stem_sumsq
    _chk(fn(g.data_ptr(), g.numel(), acc.data_ptr(), _stream()))


def clip_scale(g, sumsq_acc, max_norm):
    _chk(_lib.hip().stem_clip_scale(g.data_ptr(), g.numel(), sumsq_acc.data_ptr(), float(max_norm), _stream()))


def axpy_(y, x, a):
    _chk(_lib.hip().stem_axpy(y.data_ptr(), x.data_ptr(), float(a), y.numel(), _stream()))
    return y


def adam_step(p, g, m, v, sumsq_acc, max_norm, gscale, lr, beta1, beta2, eps, step, zero_grad=False):
    fn = _lib.hip().stem_adam_step_zero if zero_grad else _lib.hip().stem_adam_step
    _chk(fn(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), _ptr(sumsq_acc),
            max_norm, gscale, lr, beta1, beta2, eps, step, _stream()))


def adam_step_bmax(p, g, m, v, sumsq_acc, max_norm, gscale, lr, beta1, beta2, eps, step, bmax, zero_grad=False):
    """adam_step that also leaves max |p_new| per chunk of stem_adam_chunk() parameters in `bmax` (the fp16 weight packing that
    follows takes its scales from them: engine.ensure_packed(block_max=...))"""
    _chk(_lib.hip().stem_adam_step_bmax(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), _ptr(sumsq_acc),
                                        max_norm, gscale, lr, beta1, beta2, eps, step, int(bool(zero_grad)), bmax.data_ptr(), _stream()))


def adam_chunk():
    return int(_lib.hip().stem_adam_chunk())


def adam_step_dev(p, g, m, v, sumsq_acc, max_norm, gscale, lr_dev, beta1, beta2, eps, step_dev, scal_dev):
    """adam_step with the step count (int64, incremented here) and learning rate (fp32) in device memory"""
    _chk(_lib.hip().stem_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), _ptr(sumsq_acc),
                                       max_norm, gscale, lr_dev.data_ptr(), beta1, beta2, eps, step_dev.data_ptr(),
                                       scal_dev.data_ptr(), _stream()))


# ----------------------------------------------------------------------------- fused training glue
_M64 = 2 ** 64 - 1


def _noise_args(noise, seed, offset, epoch):
    """(noise ptr | NULL, seed, offset, epoch ptr | NULL, epoch stride) of the fused kernels' noise source"""
    if noise is not None:
        assert noise.is_cuda and noise.dtype == torch.float32
        return noise.data_ptr(), 0, 0, None, 0
    return None, int(seed) & _M64, int(offset) & _M64, (None if epoch is None else epoch.data_ptr()), (0 if epoch is None else NOISE_EPOCH_STRIDE)


def prior_prologue(yc, yd, residual, training, with_t_hat, noise=None, seed=0, offset=0, epoch=None, records=None):
    """-> (he_in [B,2C,H,W], target, t_hat | None, y_hat | None), all NHWC.  One kernel for cat(y_cur, y_cond), the residual,
    its noisy / rounded version and y_hat (spatiotemporalpriors.py:846-856,863).  records: a dict that receives the scale
    records "in" (max over y_cur and y_cond: serves he_in and y_cond) and "t_hat" for F16Planes.split(..., src_q=)."""
    B, Cc, H, W = yc.shape
    dev = yc.device
    he_in, target = empty_nhwc(B, 2 * Cc, H, W, dev), empty_nhwc(B, Cc, H, W, dev)
    t_hat = empty_nhwc(B, Cc, H, W, dev) if with_t_hat else None
    y_hat = empty_nhwc(B, Cc, H, W, dev) if with_t_hat else None
    if noise is not None:
        assert nhwc_ld(noise) == Cc
    nptr, sd, off, ep, stride = _noise_args(noise, seed, offset, epoch)
    q_in = q_t = None
    if records is not None:
        q_in = records["in"] = qrec_for(B * H * W * Cc // 4, dev)
        if with_t_hat:
            q_t = records["t_hat"] = qrec_for(B * H * W * Cc // 4, dev)
    _chk(_lib.hip().stem_prior_prologue(yc.data_ptr(), nhwc_ld(yc), yd.data_ptr(), nhwc_ld(yd), he_in.data_ptr(), 2 * Cc, target.data_ptr(),
                                        _ptr(t_hat), _ptr(y_hat), nptr, sd, off, ep, stride, B * H * W, Cc, int(bool(residual)),
                                        int(bool(training)), _ptr(q_in), _ptr(q_t), _stream()))
    return he_in, target, t_hat, y_hat


def rate_partials(n):
    return int(_lib.hip().stem_rate_partials(n))


def eb_forward_train(z, pack, coef, noise=None, seed=0, offset=0, epoch=None, bound=1e-9, record=False):
    """-> (z_hat, lik, dlik, partials): training-mode EntropyBottleneck forward + dlik = coef / lik + log2 partial sums;
    record=True: -> (..., partials, q) with q the scale record of z_hat for F16Planes.split(z_hat, src_q=)"""
    B, Cc, H, W = z.shape
    z_hat, lik, dlik = (empty_nhwc(B, Cc, H, W, z.device) for _ in range(3))
    part = torch.empty(rate_partials(z.numel()), dtype=torch.float64, device=z.device)
    if noise is not None:
        assert nhwc_ld(noise) == Cc
    nptr, sd, off, ep, stride = _noise_args(noise, seed, offset, epoch)
    if record:
        q = qrec_for(z.numel(), z.device)
        _chk(_lib.hip().stem_eb_forward_train_rec(z.data_ptr(), nhwc_ld(z), pack.data_ptr(), nptr, sd, off, ep, stride, z_hat.data_ptr(),
                                                  lik.data_ptr(), dlik.data_ptr(), part.data_ptr(), B * H * W, Cc, bound, coef, q.data_ptr(),
                                                  _stream()))
        return z_hat, lik, dlik, part, q
    _chk(_lib.hip().stem_eb_forward_train(z.data_ptr(), nhwc_ld(z), pack.data_ptr(), nptr, sd, off, ep, stride, z_hat.data_ptr(),
                                          lik.data_ptr(), dlik.data_ptr(), part.data_ptr(), B * H * W, Cc, bound, coef, _stream()))
    return z_hat, lik, dlik, part


def gc_forward_train(y, scales, means, coef, noise=None, seed=0, offset=0, epoch=None, scale_bound=0.11, lik_bound=1e-9, backward=None,
                     record=False):
    """-> (out, lik, dlik, partials): training-mode GaussianConditional forward (y + noise; dense y) + dlik + log2 partials.
    backward=(dscales, dmeans): the gradients gc_backward would compute from dlik, in the same launch -> (..., partials, q) with q
    their scale record (record=True) or None"""
    B, Cc, H, W = y.shape
    assert nhwc_ld(y) == Cc and nhwc_ld(scales) == nhwc_ld(means)
    out, lik, dlik = (empty_nhwc(B, Cc, H, W, y.device) for _ in range(3))
    part = torch.empty(rate_partials(y.numel()), dtype=torch.float64, device=y.device)
    if noise is not None:
        assert nhwc_ld(noise) == Cc
    nptr, sd, off, ep, stride = _noise_args(noise, seed, offset, epoch)
    if backward is not None:
        dsc, dmu = backward
        assert nhwc_ld(dsc) == nhwc_ld(dmu)
        q = qrec_for(B * H * W * Cc, y.device) if record else None
        _chk(_lib.hip().stem_gc_forward_backward_train(y.data_ptr(), scales.data_ptr(), means.data_ptr(), nhwc_ld(scales), nptr, sd, off, ep,
                                                       stride, out.data_ptr(), lik.data_ptr(), dlik.data_ptr(), part.data_ptr(), B * H * W, Cc,
                                                       scale_bound, lik_bound, coef, dsc.data_ptr(), dmu.data_ptr(), nhwc_ld(dsc), _ptr(q),
                                                       _stream()))
        return out, lik, dlik, part, q
    _chk(_lib.hip().stem_gc_forward_train(y.data_ptr(), scales.data_ptr(), means.data_ptr(), nhwc_ld(scales), nptr, sd, off, ep, stride,
                                          out.data_ptr(), lik.data_ptr(), dlik.data_ptr(), part.data_ptr(), B * H * W, Cc, scale_bound,
                                          lik_bound, coef, _stream()))
    return out, lik, dlik, part


def em_loss_finalize(part_y, part_z, scale, out3=None):
    """-> fp64 [3] = (y_bpp, z_bpp, loss) from the partial log2 sums (EMLoss, utils.py:18-27)"""
    if out3 is None:
        out3 = torch.empty(3, dtype=torch.float64, device=part_y.device)
    _chk(_lib.hip().stem_em_loss_finalize(part_y.data_ptr(), part_y.numel(), part_z.data_ptr(), part_z.numel(), float(scale),
                                          out3.data_ptr(), _stream()))
    return out3


def eb_aux_loss_grad(quantiles, pack, target, dq_out, loss_out=None, accumulate=False):
    """EntropyBottleneck.loss and d loss / d quantiles in one single-workgroup kernel; dq_out is written (or added to)"""
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=quantiles.device)
    _chk(_lib.hip().stem_eb_aux_loss_grad(quantiles.data_ptr(), pack.data_ptr(), target.data_ptr(), loss_out.data_ptr(),
                                          _ptr(dq_out), quantiles.shape[0], int(bool(accumulate)), _stream()))
    return loss_out
