"""fp32 CPU port of the STEM hot path's dense ops on the host BLAS (numpy -> OpenBLAS SGEMM, all cores): the
`cpu_baseline` of bench.py.  TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/ -- nothing in the
product imports it.

Why it exists: oracle/stem_oracle.c is a checker (plain loops, double accumulation); timed, it is a strawman next to
what the reference actually runs on a CPU (torch -> MKL-DNN: im2col-style SGEMMs in fp32).  This module restates the
same arithmetic in that form -- im2col / col2im around one SGEMM per convolution, fp32 accumulation -- so that the CPU
number next to the MI355X number is the right order of magnitude for "the reference's CPU path on this box"
(kind = "port"; the reference's Python cannot travel to the GPU box).  It is pinned against the C oracle by
tests/test_oracle_vs_golden.py::test_blas_port_matches_oracle.

Same signatures as stem_oracle.{conv2d_fwd, conv2d_bwd, deconv2d_fwd, deconv2d_bwd, gdn_fwd}; `installed()` swaps them
into stem_oracle so that stem_oracle.g_a / stem_forward / stem_backward run on the BLAS:

    with stem_port_blas.installed():
        y = orc.g_a(isd, x); out = orc.stem_forward(...); g = orc.stem_backward(...)

Reference call sites restated: nn.Conv2d / nn.ConvTranspose2d forward + autograd (spatiotemporalpriors.py:807-838,
models/utils.py:112-130), GDN (layers/gdn.py:52-67, ops/parametrizers.py:27-45).
"""
import contextlib

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

_PED = np.float32(2.0 ** -36)


def _nhwc(x):
    return np.ascontiguousarray(np.transpose(np.asarray(x, np.float32), (0, 2, 3, 1)))


def _nchw(x):
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def _wmat(w):
    """Conv2d weight [K,C,R,S] -> [K, R*S*C]: the column order of _im2col (tap-major, channels contiguous)"""
    K = w.shape[0]
    return np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1)).reshape(K, -1))


def _im2col(xh, R, S, stride, pad, Ho, Wo):
    """xh [N,H,W,C] -> [N*Ho*Wo, R*S*C]: one strided block copy per tap, each moving contiguous runs of C floats"""
    N, _, _, C = xh.shape
    if R == 1 and S == 1 and pad == 0:
        return np.ascontiguousarray(xh[:, ::stride, ::stride, :][:, :Ho, :Wo]).reshape(-1, C)
    if pad:
        xh = np.pad(xh, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    col = np.empty((N, Ho, Wo, R * S, C), np.float32)
    for r in range(R):
        for s in range(S):
            col[:, :, :, r * S + s, :] = xh[:, r:r + stride * Ho:stride, s:s + stride * Wo:stride, :][:, :Ho, :Wo]
    return col.reshape(N * Ho * Wo, R * S * C)


def _col2im(col, N, H, W, C, R, S, stride, pad, Ho, Wo):
    """adjoint of _im2col: col [N*Ho*Wo, R*S*C] -> [N,H,W,C]"""
    if R == 1 and S == 1 and pad == 0 and stride == 1:
        return col.reshape(N, H, W, C)
    out = np.zeros((N, H + 2 * pad + stride + R, W + 2 * pad + stride + S, C), np.float32)
    col = col.reshape(N, Ho, Wo, R * S, C)
    for r in range(R):
        for s in range(S):
            out[:, r:r + stride * Ho:stride, s:s + stride * Wo:stride, :] += col[:, :, :, r * S + s, :]
    return out[:, pad:pad + H, pad:pad + W, :]


def _batches(N, rows_per_sample, cols, limit=1 << 28):
    """split the batch so that one im2col matrix stays below `limit` floats (1 GiB)"""
    step = max(1, int(limit // max(1, rows_per_sample * cols)))
    return [(i, min(N, i + step)) for i in range(0, N, step)]


def conv2d_fwd(x, w, b, stride, pad):
    xh, w = _nhwc(x), np.asarray(w, np.float32)
    N, H, W, C = xh.shape
    K, _, R, S = w.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    wm = np.ascontiguousarray(_wmat(w).T)
    y = np.empty((N, Ho, Wo, K), np.float32)
    for lo, hi in _batches(N, Ho * Wo, C * R * S):
        y[lo:hi] = (_im2col(xh[lo:hi], R, S, stride, pad, Ho, Wo) @ wm).reshape(hi - lo, Ho, Wo, K)
    if b is not None:
        y += np.asarray(b, np.float32)
    return _nchw(y)


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True):
    xh, dyh, w = _nhwc(x), _nhwc(dy), np.asarray(w, np.float32)
    N, H, W, C = xh.shape
    K, _, R, S = w.shape
    Ho, Wo = dyh.shape[1:3]
    wm = _wmat(w)
    dw = np.zeros((K, R * S * C), np.float32)
    dx = np.empty_like(xh) if need_dx else None
    for lo, hi in _batches(N, Ho * Wo, C * R * S):
        d2 = dyh[lo:hi].reshape(-1, K)
        dw += d2.T @ _im2col(xh[lo:hi], R, S, stride, pad, Ho, Wo)
        if need_dx:
            dx[lo:hi] = _col2im(d2 @ wm, hi - lo, H, W, C, R, S, stride, pad, Ho, Wo)
    dw = np.ascontiguousarray(np.transpose(dw.reshape(K, R, S, C), (0, 3, 1, 2)))
    return (None if dx is None else _nchw(dx)), dw, dyh.reshape(-1, K).sum(0, dtype=np.float32)


def deconv2d_fwd(x, w, b, stride, pad, opad):
    """ConvTranspose2d = the input-gradient map of Conv2d(w viewed as [C_in=K_conv ...]): y = col2im(x @ w[C, K*R*S])"""
    xh, w = _nhwc(x), np.asarray(w, np.float32)
    N, H, W, C = xh.shape
    _, K, R, S = w.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + R + opad, (W - 1) * stride - 2 * pad + S + opad
    wm = _wmat(w)                                                        # [C, R*S*K]
    y = np.empty((N, Ho, Wo, K), np.float32)
    for lo, hi in _batches(N, H * W, K * R * S):
        y[lo:hi] = _col2im(xh[lo:hi].reshape(-1, C) @ wm, hi - lo, Ho, Wo, K, R, S, stride, pad, H, W)
    if b is not None:
        y += np.asarray(b, np.float32)
    return _nchw(y)


def deconv2d_bwd(x, w, dy, stride, pad, opad, need_dx=True):
    xh, dyh, w = _nhwc(x), _nhwc(dy), np.asarray(w, np.float32)
    N, H, W, C = xh.shape
    _, K, R, S = w.shape
    wm = np.ascontiguousarray(_wmat(w).T)
    dw = np.zeros((C, R * S * K), np.float32)
    dx = np.empty_like(xh) if need_dx else None
    for lo, hi in _batches(N, H * W, K * R * S):
        col = _im2col(dyh[lo:hi], R, S, stride, pad, H, W)                 # [n*H*W, K*R*S]
        dw += xh[lo:hi].reshape(-1, C).T @ col
        if need_dx:
            dx[lo:hi] = (col @ wm).reshape(hi - lo, H, W, C)
    dw = np.ascontiguousarray(np.transpose(dw.reshape(C, R, S, K), (0, 3, 1, 2)))
    return (None if dx is None else _nchw(dx)), dw, dyh.reshape(-1, K).sum(0, dtype=np.float32)


def gdn_fwd(x, beta_p, gamma_p, inverse=False, beta_min=1e-6):
    xh = _nhwc(x)
    bound = np.float32(np.sqrt(beta_min + 2.0 ** -36))
    beta = np.maximum(np.asarray(beta_p, np.float32), bound) ** 2 - _PED
    gamma = np.maximum(np.asarray(gamma_p, np.float32), np.float32(2.0 ** -18)) ** 2 - _PED
    norm = (xh * xh).reshape(-1, xh.shape[3]) @ np.ascontiguousarray(gamma.T) + beta
    norm = np.sqrt(norm) if inverse else np.float32(1.0) / np.sqrt(norm)
    return _nchw(xh * norm.reshape(xh.shape))


@contextlib.contextmanager
def installed():
    """stem_oracle's dense ops -> this module's, for the duration of the block"""
    import stem_oracle as orc
    names = ("conv2d_fwd", "conv2d_bwd", "deconv2d_fwd", "deconv2d_bwd", "gdn_fwd")
    saved = {n: getattr(orc, n) for n in names}
    try:
        for n in names:
            setattr(orc, n, globals()[n])
        yield orc
    finally:
        for n, f in saved.items():
            setattr(orc, n, f)
