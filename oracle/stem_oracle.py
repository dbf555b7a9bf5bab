"""ctypes binding + model-level composition for the CPU ORACLE (test infrastructure).

NOT the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module.  The arithmetic lives in oracle/stem_oracle.c (plain C);
this file only marshals numpy arrays and chains the ops in the order the
reference's modules do (citations are paths under /root/reference):

* g_a / g_s of JointAutoregressiveHierarchicalPriors  compressai/models/priors.py:421-439,686-694,397-402
* SpatioTemporalPriorModel(_Res).forward              compressai/models/spatiotemporalpriors.py:561-585,845-868
* EMLoss                                              utils.py:18-27

Layout is the reference's NCHW fp32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
EB_NPARAM = 58
SCALE_BOUND = 0.11      # GaussianConditional scale_bound (entropy_models.py:484)
LIK_BOUND = 1e-9        # EntropyModel likelihood_bound (entropy_models.py:79)
LRELU = 0.01            # nn.LeakyReLU default slope


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libstem_oracle.so")
    src = os.path.join(_HERE, "stem_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libstem_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_rate_bpp.restype = C.c_double
        _LIB.orc_eb_aux_loss.restype = C.c_float
        _LIB.orc_rans_encode.restype = C.c_long
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- ops
def conv2d_fwd(x, w, b, stride, pad):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    N, Cc, H, W = x.shape
    K, _, R, S = w.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    y = np.empty((N, K, Ho, Wo), np.float32)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cc, H, W, K, R, S, stride, pad)
    return y


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True):
    x, w, dy = _f(x), _f(w), _f(dy)
    N, Cc, H, W = x.shape
    K, _, R, S = w.shape
    dx = np.empty_like(x) if need_dx else None
    if need_dx:
        lib().orc_conv2d_dgrad(_p(dy), _p(w), _p(dx), N, Cc, H, W, K, R, S, stride, pad)
    dw, db = np.empty_like(w), np.empty((K,), np.float32)
    lib().orc_conv2d_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, Cc, H, W, K, R, S, stride, pad)
    return dx, dw, db


def deconv2d_fwd(x, w, b, stride, pad, opad):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    N, Cc, H, W = x.shape
    _, K, R, S = w.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + R + opad, (W - 1) * stride - 2 * pad + S + opad
    y = np.empty((N, K, Ho, Wo), np.float32)
    lib().orc_deconv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cc, H, W, K, R, S, stride, pad, opad)
    return y


def deconv2d_bwd(x, w, dy, stride, pad, opad, need_dx=True):
    x, w, dy = _f(x), _f(w), _f(dy)
    N, Cc, H, W = x.shape
    _, K, R, S = w.shape
    dx = np.empty_like(x) if need_dx else None
    if need_dx:
        lib().orc_deconv2d_dgrad(_p(dy), _p(w), _p(dx), N, Cc, H, W, K, R, S, stride, pad, opad)
    dw, db = np.empty_like(w), np.empty((K,), np.float32)
    lib().orc_deconv2d_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, Cc, H, W, K, R, S, stride, pad, opad)
    return dx, dw, db


def lrelu_fwd(x, slope=LRELU):
    x = _f(x)
    y = np.empty_like(x)
    lib().orc_lrelu_fwd(_p(x), _p(y), C.c_size_t(x.size), C.c_float(slope))
    return y


def lrelu_bwd(y, dy, slope=LRELU):
    y, dy = _f(y), _f(dy)
    dx = np.empty_like(y)
    lib().orc_lrelu_bwd(_p(y), _p(dy), _p(dx), C.c_size_t(y.size), C.c_float(slope))
    return dx


def sft_fwd(x, gamma, beta, slope=1.0):
    x, gamma, beta = _f(x), _f(gamma), _f(beta)
    out = np.empty_like(x)
    lib().orc_sft_fwd(_p(x), _p(gamma), _p(beta), _p(out), C.c_size_t(x.size), C.c_float(slope))
    return out


def sft_bwd(x, gamma, out, dout, slope=1.0):
    x, gamma, out, dout = _f(x), _f(gamma), _f(out), _f(dout)
    dx, dg, db = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    lib().orc_sft_bwd(_p(x), _p(gamma), _p(out), _p(dout), _p(dx), _p(dg), _p(db), C.c_size_t(x.size), C.c_float(slope))
    return dx, dg, db


def avgpool_fwd(x, Ho, Wo):
    x = _f(x)
    N, Cc, H, W = x.shape
    y = np.empty((N, Cc, Ho, Wo), np.float32)
    lib().orc_adaptive_avgpool_fwd(_p(x), _p(y), N * Cc, H, W, Ho, Wo)
    return y


def avgpool_bwd(dy, H, W):
    dy = _f(dy)
    N, Cc, Ho, Wo = dy.shape
    dx = np.empty((N, Cc, H, W), np.float32)
    lib().orc_adaptive_avgpool_bwd(_p(dy), _p(dx), N * Cc, H, W, Ho, Wo)
    return dx


# ---- SFT / SFTResblk modules (compressai/models/stem_utils.py:24-63), forward and backward ----------------------
def sft_module_fwd(p, prefix, x, q, slope=1.0):
    """p: {name: ndarray} with the reference's parameter names under `prefix`; ks taken from the weights."""
    pad = p[prefix + "mlp_shared.0.weight"].shape[-1] // 2
    qp = avgpool_fwd(q, x.shape[2], x.shape[3])                                       # stem_utils.py:37
    actv = lrelu_fwd(conv2d_fwd(qp, p[prefix + "mlp_shared.0.weight"], p[prefix + "mlp_shared.0.bias"], 1, pad), 0.0)   # ReLU
    gamma = conv2d_fwd(actv, p[prefix + "mlp_gamma.weight"], p[prefix + "mlp_gamma.bias"], 1, pad)
    beta = conv2d_fwd(actv, p[prefix + "mlp_beta.weight"], p[prefix + "mlp_beta.bias"], 1, pad)
    out = sft_fwd(x, gamma, beta, slope)                                              # stem_utils.py:41
    return out, (x, q, qp, actv, gamma, out, slope, pad)


def sft_module_bwd(p, prefix, cache, dout):
    x, q, qp, actv, gamma, out, slope, pad = cache
    dx, dgamma, dbeta = sft_bwd(x, gamma, out, dout, slope)
    g = {}
    da1, g[prefix + "mlp_gamma.weight"], g[prefix + "mlp_gamma.bias"] = conv2d_bwd(actv, p[prefix + "mlp_gamma.weight"], dgamma, 1, pad)
    da2, g[prefix + "mlp_beta.weight"], g[prefix + "mlp_beta.bias"] = conv2d_bwd(actv, p[prefix + "mlp_beta.weight"], dbeta, 1, pad)
    dact = lrelu_bwd(actv, da1 + da2, 0.0)
    dqp, g[prefix + "mlp_shared.0.weight"], g[prefix + "mlp_shared.0.bias"] = conv2d_bwd(qp, p[prefix + "mlp_shared.0.weight"], dact, 1, pad)
    return dx, avgpool_bwd(dqp, q.shape[2], q.shape[3]), g


def sft_resblk_fwd(p, prefix, x, q):
    a0, c0 = sft_module_fwd(p, prefix + "norm_0.", x, q, 0.2)                         # stem_utils.py:56,62-63
    d0 = conv2d_fwd(a0, p[prefix + "conv_0.weight"], p[prefix + "conv_0.bias"], 1, 1)
    a1, c1 = sft_module_fwd(p, prefix + "norm_1.", d0, q, 0.2)
    d1 = conv2d_fwd(a1, p[prefix + "conv_1.weight"], p[prefix + "conv_1.bias"], 1, 1)
    return x + d1, (c0, a0, c1, a1)


def sft_resblk_bwd(p, prefix, cache, dout):
    c0, a0, c1, a1 = cache
    g = {}
    da1, g[prefix + "conv_1.weight"], g[prefix + "conv_1.bias"] = conv2d_bwd(a1, p[prefix + "conv_1.weight"], dout, 1, 1)
    dd0, dq1, g1 = sft_module_bwd(p, prefix + "norm_1.", c1, da1)
    da0, g[prefix + "conv_0.weight"], g[prefix + "conv_0.bias"] = conv2d_bwd(a0, p[prefix + "conv_0.weight"], dd0, 1, 1)
    dx, dq0, g0 = sft_module_bwd(p, prefix + "norm_0.", c0, da0)
    g.update(g0)
    g.update(g1)
    return dout + dx, dq0 + dq1, g


def gdn_fwd(x, beta_p, gamma_p, inverse=False, beta_min=1e-6):
    x, beta_p, gamma_p = _f(x), _f(beta_p), _f(gamma_p)
    N, Cc, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_gdn_fwd(_p(x), _p(beta_p), _p(gamma_p), _p(y), N, Cc, H, W, int(inverse), C.c_float(beta_min))
    return y


def gdn_bwd(x, dy, beta_p, gamma_p, inverse=False, beta_min=1e-6):
    x, dy, beta_p, gamma_p = _f(x), _f(dy), _f(beta_p), _f(gamma_p)
    N, Cc, H, W = x.shape
    dx, db, dg = np.empty_like(x), np.empty_like(beta_p), np.empty_like(gamma_p)
    lib().orc_gdn_bwd(_p(x), _p(dy), _p(beta_p), _p(gamma_p), _p(dx), _p(db), _p(dg), N, Cc, H, W, int(inverse), C.c_float(beta_min))
    return dx, db, dg


def eb_pack_params(sd, prefix="entropy_bottleneck."):
    """[C,58] pack in the order stem_oracle.c documents (matrix, bias, factor per layer)."""
    cols = []
    for i in range(5):
        cols.append(np.asarray(sd[f"{prefix}_matrix{i}"], np.float32).reshape(len(sd[f"{prefix}_matrix{i}"]), -1))
        cols.append(np.asarray(sd[f"{prefix}_bias{i}"], np.float32).reshape(cols[-1].shape[0], -1))
        if i < 4:
            cols.append(np.asarray(sd[f"{prefix}_factor{i}"], np.float32).reshape(cols[-1].shape[0], -1))
    out = np.concatenate(cols, axis=1)
    assert out.shape[1] == EB_NPARAM
    return np.ascontiguousarray(out)


def eb_unpack_grads(dpack, prefix="entropy_bottleneck."):
    shapes = [(3, 1), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (1, 3), (1, 1)]
    names = []
    for i in range(5):
        names += [f"_matrix{i}", f"_bias{i}"] + ([f"_factor{i}"] if i < 4 else [])
    out, o = {}, 0
    for n, s in zip(names, shapes):
        k = s[0] * s[1]
        out[prefix + n] = dpack[:, o:o + k].reshape(-1, *s).copy()
        o += k
    return out


def eb_likelihood_fwd(v_cl, pack):
    """v_cl: [C, L] channel-major values (already noised/rounded)."""
    v_cl, pack = _f(v_cl), _f(pack)
    lik = np.empty_like(v_cl)
    lib().orc_eb_likelihood_fwd(_p(v_cl), _p(pack), _p(lik), v_cl.shape[0], C.c_size_t(v_cl.shape[1]), C.c_float(LIK_BOUND))
    return lik


def eb_likelihood_bwd(v_cl, pack, dlik_cl):
    v_cl, pack, dlik_cl = _f(v_cl), _f(pack), _f(dlik_cl)
    dv, dp = np.empty_like(v_cl), np.empty_like(pack)
    lib().orc_eb_likelihood_bwd(_p(v_cl), _p(pack), _p(dlik_cl), _p(dv), _p(dp), v_cl.shape[0],
                                C.c_size_t(v_cl.shape[1]), C.c_float(LIK_BOUND))
    return dv, dp


def eb_aux_loss(quantiles, pack, target):
    q, pack, t = _f(quantiles).reshape(-1, 3), _f(pack), _f(target)
    dq = np.empty_like(q)
    loss = lib().orc_eb_aux_loss(_p(q), _p(pack), _p(t), _p(dq), q.shape[0])
    return float(loss), dq.reshape(-1, 1, 3)


def nchw_to_cl(x):
    """EntropyBottleneck.forward's permute(1,2,3,0).reshape(C,1,-1) (entropy_models.py:426-428) -> [C, H*W*N]"""
    return np.ascontiguousarray(np.transpose(x, (1, 2, 3, 0)).reshape(x.shape[1], -1))


def cl_to_nchw(v, shape):
    N, Cc, H, W = shape
    return np.ascontiguousarray(np.transpose(v.reshape(Cc, H, W, N), (3, 0, 1, 2)))


def gc_likelihood_fwd(y, scales, means):
    y, scales = _f(y), _f(scales)
    means = None if means is None else _f(means)
    lik = np.empty_like(y)
    lib().orc_gc_likelihood_fwd(_p(y), _p(scales), _p(means), _p(lik), C.c_size_t(y.size), C.c_float(SCALE_BOUND), C.c_float(LIK_BOUND))
    return lik


def gc_likelihood_bwd(y, scales, means, dlik):
    y, scales, means, dlik = _f(y), _f(scales), _f(means), _f(dlik)
    dy, ds, dm = np.empty_like(y), np.empty_like(y), np.empty_like(y)
    lib().orc_gc_likelihood_bwd(_p(y), _p(scales), _p(means), _p(dlik), _p(dy), _p(ds), _p(dm), C.c_size_t(y.size),
                                C.c_float(SCALE_BOUND), C.c_float(LIK_BOUND))
    return dy, ds, dm


def quantize_dequantize(x, means=None):
    x = _f(x)
    means = None if means is None else _f(np.broadcast_to(means, x.shape))
    out = np.empty_like(x)
    lib().orc_quantize_dequantize(_p(x), _p(means), _p(out), C.c_size_t(x.size))
    return out


def quantize_symbols(x, means=None):
    x = _f(x)
    means = None if means is None else _f(np.broadcast_to(means, x.shape))
    out = np.empty(x.shape, np.int32)
    lib().orc_quantize_symbols(_p(x), _p(means), _p(out), C.c_size_t(x.size))
    return out


def build_indexes(scales, table):
    scales, table = _f(scales), _f(table)
    idx = np.empty(scales.shape, np.int32)
    lib().orc_build_indexes(_p(scales), _p(table), len(table), _p(idx), C.c_size_t(scales.size), C.c_float(SCALE_BOUND))
    return idx


def rate_bpp(lik, num_pixels):
    lik = _f(lik)
    return float(lib().orc_rate_bpp(_p(lik), C.c_size_t(lik.size), C.c_double(num_pixels)))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = _f(pmf)
    cdf = np.empty(len(pmf) + 1, np.uint32)
    rc = lib().orc_pmf_to_quantized_cdf(_p(pmf), len(pmf), precision, _p(cdf))
    if rc != 0:
        raise ValueError(f"pmf_to_quantized_cdf failed ({rc})")
    return cdf


def rans_encode(symbols, indexes, cdfs, sizes, offsets) -> bytes:
    symbols, indexes = np.ascontiguousarray(symbols, np.int32), np.ascontiguousarray(indexes, np.int32)
    cdfs, sizes, offsets = (np.ascontiguousarray(a, np.int32) for a in (cdfs, sizes, offsets))
    cap = 8 * symbols.size + 64
    out = np.empty(cap, np.uint8)
    n = lib().orc_rans_encode(_p(symbols), _p(indexes), C.c_size_t(symbols.size), _p(cdfs), cdfs.shape[1], _p(sizes),
                              _p(offsets), _p(out), C.c_size_t(cap))
    if n < 0:
        raise RuntimeError("rans_encode: output buffer too small")
    return out[:n].tobytes()


def rans_decode(stream: bytes, indexes, cdfs, sizes, offsets):
    buf = np.frombuffer(stream, np.uint8).copy()
    indexes = np.ascontiguousarray(indexes, np.int32)
    cdfs, sizes, offsets = (np.ascontiguousarray(a, np.int32) for a in (cdfs, sizes, offsets))
    out = np.empty(indexes.shape, np.int32)
    lib().orc_rans_decode(_p(buf), C.c_size_t(buf.size), _p(indexes), C.c_size_t(indexes.size), _p(cdfs), cdfs.shape[1],
                          _p(sizes), _p(offsets), _p(out))
    return out


# ----------------------------------------------------------------------------- model composition
def g_a(sd, x, prefix="g_a."):
    """priors.py:421-429: 4x conv5x5 s2 with GDN after the first three."""
    h = x
    for i in range(4):
        h = conv2d_fwd(h, sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"], 2, 2)
        if i < 3:
            h = gdn_fwd(h, sd[f"{prefix}{2 * i + 1}.beta"], sd[f"{prefix}{2 * i + 1}.gamma"], inverse=False)
    return h


def g_s(sd, y, prefix="g_s."):
    """priors.py:431-439 + getX clamp (priors.py:397-402)."""
    h = y
    for i in range(4):
        h = deconv2d_fwd(h, sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"], 2, 2, 1)
        if i < 3:
            h = gdn_fwd(h, sd[f"{prefix}{2 * i + 1}.beta"], sd[f"{prefix}{2 * i + 1}.gamma"], inverse=True)
    return np.clip(h, 0.0, 1.0)


def masked_weight(w):
    """MaskedConv2d type-A mask (layers/layers.py:39-42)."""
    m = np.ones_like(w)
    _, _, h, ww = w.shape
    m[:, :, h // 2, ww // 2:] = 0
    m[:, :, h // 2 + 1:] = 0
    return w * m


def stem_forward(sd, y_cur, y_cond, residual: bool, training: bool, noise=None, keep=None):
    """SpatioTemporalPriorModel(_Res).forward.  `noise` = dict with 'z', 'q', 'lik' arrays (training).
    Returns dict(y_hat, lik_y, lik_z, scales, means); `keep` (dict) receives activations for backward."""
    k = {} if keep is None else keep
    sd = {n: np.asarray(v, np.float32) for n, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    cat = np.concatenate([y_cur, y_cond], 1)
    k["he_in"] = cat
    k["he0"] = lrelu_fwd(conv2d_fwd(cat, sd["HE.0.weight"], sd["HE.0.bias"], 1, 1))
    k["he2"] = lrelu_fwd(conv2d_fwd(k["he0"], sd["HE.2.weight"], sd["HE.2.bias"], 2, 2))
    z = conv2d_fwd(k["he2"], sd["HE.4.weight"], sd["HE.4.bias"], 2, 2)
    pack = eb_pack_params(sd)
    med = sd["entropy_bottleneck.quantiles"][:, 0, 1]
    zc = nchw_to_cl(z)
    if training:
        zq = zc + nchw_to_cl(noise["z"])
    else:
        zq = quantize_dequantize(zc, med[:, None])
    k["z_cl"], k["pack"] = zq, pack
    lik_z = cl_to_nchw(eb_likelihood_fwd(zq, pack), z.shape)
    z_hat = cl_to_nchw(zq, z.shape)
    k["z_hat"] = z_hat
    k["hd0"] = lrelu_fwd(deconv2d_fwd(z_hat, sd["HD.0.weight"], sd["HD.0.bias"], 2, 2, 1))
    k["hd2"] = lrelu_fwd(deconv2d_fwd(k["hd0"], sd["HD.2.weight"], sd["HD.2.bias"], 2, 2, 1))
    hp = conv2d_fwd(k["hd2"], sd["HD.4.weight"], sd["HD.4.bias"], 1, 1)
    # the ablations drop the temporal prior and / or the spatial (masked-conv) prior: spatiotemporalpriors.py:33-505
    has_tpm, has_spm = "TPM.0.weight" in sd, "context_prediction.weight" in sd
    k["has_tpm"], k["has_spm"] = has_tpm, has_spm
    priors = []
    if has_tpm:
        k["tp_in"] = y_cond
        k["tp0"] = lrelu_fwd(conv2d_fwd(y_cond, sd["TPM.0.weight"], sd["TPM.0.bias"], 1, 2))
        k["tp2"] = lrelu_fwd(conv2d_fwd(k["tp0"], sd["TPM.2.weight"], sd["TPM.2.bias"], 1, 2))
        priors.append(conv2d_fwd(k["tp2"], sd["TPM.4.weight"], sd["TPM.4.bias"], 1, 2))
    priors.append(hp)
    target = (y_cur - y_cond) if residual else y_cur
    t_hat = None
    if has_spm:
        if training:
            t_hat = target + noise["q"]
        else:
            t_hat = quantize_dequantize(target)
        k["t_hat"] = t_hat
        priors.append(conv2d_fwd(t_hat, masked_weight(sd["context_prediction.weight"]), sd["context_prediction.bias"], 1, 2))
    k["epm_in"] = np.concatenate(priors, 1)
    k["e0"] = lrelu_fwd(conv2d_fwd(k["epm_in"], sd["EPM.0.weight"], sd["EPM.0.bias"], 1, 0))
    k["e2"] = lrelu_fwd(conv2d_fwd(k["e0"], sd["EPM.2.weight"], sd["EPM.2.bias"], 1, 0))
    gp = conv2d_fwd(k["e2"], sd["EPM.4.weight"], sd["EPM.4.bias"], 1, 0)
    M = gp.shape[1] // 2
    scales, means = np.ascontiguousarray(gp[:, :M]), np.ascontiguousarray(gp[:, M:])
    if training:
        out = target + noise["lik"]
    else:
        out = quantize_dequantize(target, means)
    k["gc_in"], k["scales"], k["means"] = out, scales, means
    lik_y = gc_likelihood_fwd(out, scales, means)
    if has_spm:
        y_hat = (t_hat + y_cond) if residual else t_hat
    else:
        y_hat = out                                          # models without the spatial prior return the Gaussian's output
    return {"y_hat": y_hat, "lik_y": lik_y, "lik_z": lik_z, "scales": scales, "means": means}


def _lrelu_vec(v, slope=LRELU):
    return np.where(v >= 0, v, v * np.float32(slope)).astype(np.float32)


def reference_rans_decoder():
    """the reference's own stateful decoder (compressai/cpp_exts/rans/rans_interface.cpp compiled into oracle/_ref by the
    Makefile), or None when oracle/_ref is absent"""
    import os as _os
    import sys as _sys
    ref_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ref")
    if not _os.path.isdir(ref_dir) or not any(f.startswith("ans") for f in _os.listdir(ref_dir)):
        return None
    if ref_dir not in _sys.path:
        _sys.path.insert(0, ref_dir)
    import ans
    return ans.RansDecoder()


def stem_decoder_priors(sd, z_string, shape, y_cond, tables):
    """what the decoder knows before the raster-order loop (spatiotemporalpriors.py:976-980): z_hat off the bottleneck's string
    (channel c's table for every element of channel c, dequantised with the medians), hyper prior HD(z_hat), temporal prior
    TPM(y_conditioned) -> (hp, tp), each [1, 2M, H, W]"""
    sd = {n: np.asarray(v, np.float32) for n, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    y_cond = _f(y_cond)
    zc = sd["entropy_bottleneck.quantiles"].shape[0]
    zh, zw = int(shape[0]), int(shape[1])
    med = sd["entropy_bottleneck.quantiles"][:, 0, 1]
    zidx = np.repeat(np.arange(zc, dtype=np.int32), zh * zw)
    zsym = rans_decode(z_string, zidx, tables["eb_cdf"], tables["eb_cdf_length"], tables["eb_offset"])
    z_hat = (zsym.reshape(1, zc, zh, zw).astype(np.float32) + med.reshape(1, zc, 1, 1)).astype(np.float32)
    hd0 = lrelu_fwd(deconv2d_fwd(z_hat, sd["HD.0.weight"], sd["HD.0.bias"], 2, 2, 1))
    hd2 = lrelu_fwd(deconv2d_fwd(hd0, sd["HD.2.weight"], sd["HD.2.bias"], 2, 2, 1))
    hp = conv2d_fwd(hd2, sd["HD.4.weight"], sd["HD.4.bias"], 1, 1)
    tp0 = lrelu_fwd(conv2d_fwd(y_cond, sd["TPM.0.weight"], sd["TPM.0.bias"], 1, 2))
    tp2 = lrelu_fwd(conv2d_fwd(tp0, sd["TPM.2.weight"], sd["TPM.2.bias"], 1, 2))
    tp = conv2d_fwd(tp2, sd["TPM.4.weight"], sd["TPM.4.bias"], 1, 2)
    return hp, tp


def stem_decompress(sd, strings, shape, y_cond, tables, residual=True, max_positions=None, timing=None, decoder=None):
    """SpatioTemporalPriorModel_Res.decompress + _decompress_ar (spatiotemporalpriors.py:964-1054; residual=False: the plain model,
    :681-770): z_hat from the bottleneck's string, hyper / temporal prior through HD / TPM, then the raster-order loop -- per
    position a 5x5 crop of what has been decoded so far through the masked context convolution, the three 1x1 EPM layers on the
    concatenated (temporal, hyper, context) parameters, table indexes from the scales, `M` symbols off the rANS stream, dequantise
    with the means, write back.  One image (strings = [[y_string], [z_string]]).

    tables: {"eb_cdf", "eb_cdf_length", "eb_offset", "gc_cdf", "gc_cdf_length", "gc_offset", "gc_scale_table"} (the buffers a
    checkpoint carries).  max_positions: stop after that many positions (bounded CPU-baseline sample; the result is then partial).
    timing: a dict that receives {"positions", "loop_s"}.  decoder: a stateful symbol decoder with the reference's interface
    (set_stream / decode_stream: reference_rans_decoder()); without one the C restatement decodes prefixes of growing length
    (it keeps no state between calls: quadratic, for the small pinned cases only).  Returns y_hat [1, M, H, W]."""
    import time as _time
    sd = {n: np.asarray(v, np.float32) for n, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    y_cond = _f(y_cond)
    zh, zw = int(shape[0]), int(shape[1])
    hp, tp = stem_decoder_priors(sd, strings[1][0], shape, y_cond, tables)
    M = y_cond.shape[1]
    H, W = zh * 4, zw * 4
    pad = 2
    # (the decoder reads context_prediction.weight without the mask, :1028: the forward pass has multiplied the mask into the
    # parameter in place by then, layers.py:44-47; the taps at and behind the centre meet zeros of res_hat anyway)
    wctx = masked_weight(sd["context_prediction.weight"]).reshape(2 * M, -1)      # [2M, M*25]
    bctx = sd["context_prediction.bias"]
    w0, b0 = sd["EPM.0.weight"].reshape(sd["EPM.0.weight"].shape[0], -1), sd["EPM.0.bias"]
    w1, b1 = sd["EPM.2.weight"].reshape(sd["EPM.2.weight"].shape[0], -1), sd["EPM.2.bias"]
    w2, b2 = sd["EPM.4.weight"].reshape(sd["EPM.4.weight"].shape[0], -1), sd["EPM.4.bias"]
    table = _f(tables["gc_scale_table"])
    res = np.zeros((M, H + 2 * pad, W + 2 * pad), np.float32)
    # the stream is consumed position by position: one decoder state across calls (orc_rans_decode decodes a prefix of the
    # stream, so the loop decodes prefixes of growing length and keeps the newest M symbols -- quadratic, hence only for the
    # small pinned cases; the bounded timing sample uses its own per-position cost below)
    all_idx = []
    if decoder is not None:
        decoder.set_stream(strings[0][0])
        cdf_l = np.asarray(tables["gc_cdf"]).tolist()
        len_l, off_l = np.asarray(tables["gc_cdf_length"]).tolist(), np.asarray(tables["gc_offset"]).tolist()
    t0 = _time.perf_counter()
    npos = 0
    for h in range(H):
        for w in range(W):
            if max_positions is not None and npos >= max_positions:
                break
            crop = res[:, h:h + 5, w:w + 5].reshape(-1)
            ctx = wctx @ crop + bctx
            v = np.concatenate([tp[0, :, h, w], hp[0, :, h, w], ctx]).astype(np.float32)
            h1 = _lrelu_vec(w0 @ v + b0)
            h2 = _lrelu_vec(w1 @ h1 + b1)
            gp = (w2 @ h2 + b2).astype(np.float32)
            scales, means = gp[:M], gp[M:]
            idx = build_indexes(scales, table)
            if decoder is not None:
                sym = np.asarray(decoder.decode_stream(idx.tolist(), cdf_l, len_l, off_l), np.int32)
            else:
                all_idx.append(idx)
                cat = np.concatenate(all_idx)
                sym = rans_decode(strings[0][0], cat, tables["gc_cdf"], tables["gc_cdf_length"], tables["gc_offset"])[-M:]
            res[:, h + pad, w + pad] = sym.astype(np.float32) + means
            npos += 1
    if timing is not None:
        timing["positions"], timing["loop_s"] = npos, _time.perf_counter() - t0
    out = res[None, :, pad:pad + H, pad:pad + W]
    return (out + y_cond) if residual else out.copy()


def stem_backward(sd, keep, lik_y, lik_z, num_pixels):
    """Gradient of EMLoss = (sum log lik_y + sum log lik_z) / (-ln2 * num_pixels) wrt every STEM parameter
    (torch autograd of spatiotemporalpriors.py:845-868 with y_cur / y_cond detached as in stem/trainSTEM.py:208)."""
    sd = {n: np.asarray(v, np.float32) for n, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    k, g = keep, {}
    c = np.float32(1.0 / (-np.log(2.0) * num_pixels))
    dlik_y = (c / lik_y).astype(np.float32)
    dlik_z = (c / lik_z).astype(np.float32)
    _, dsc, dmu = gc_likelihood_bwd(k["gc_in"], k["scales"], k["means"], dlik_y)
    dgp = np.concatenate([dsc, dmu], 1)
    d, g["EPM.4.weight"], g["EPM.4.bias"] = conv2d_bwd(k["e2"], sd["EPM.4.weight"], dgp, 1, 0)
    d = lrelu_bwd(k["e2"], d)
    d, g["EPM.2.weight"], g["EPM.2.bias"] = conv2d_bwd(k["e0"], sd["EPM.2.weight"], d, 1, 0)
    d = lrelu_bwd(k["e0"], d)
    d, g["EPM.0.weight"], g["EPM.0.bias"] = conv2d_bwd(k["epm_in"], sd["EPM.0.weight"], d, 1, 0)
    has_tpm, has_spm = k.get("has_tpm", True), k.get("has_spm", True)
    P = k["epm_in"].shape[1] // (1 + int(has_tpm) + int(has_spm))
    o = 0
    dtp = dctx = None
    if has_tpm:
        dtp, o = d[:, o:o + P], o + P
    dhp, o = d[:, o:o + P], o + P
    if has_spm:
        dctx = d[:, o:o + P]
        # context_prediction: wgrad of all 25 taps (unmasked), no dgrad needed (input is detached data + noise)
        _, g["context_prediction.weight"], g["context_prediction.bias"] = conv2d_bwd(
            k["t_hat"], sd["context_prediction.weight"], np.ascontiguousarray(dctx), 1, 2, need_dx=False)
    if has_tpm:
        d, g["TPM.4.weight"], g["TPM.4.bias"] = conv2d_bwd(k["tp2"], sd["TPM.4.weight"], np.ascontiguousarray(dtp), 1, 2)
        d = lrelu_bwd(k["tp2"], d)
        d, g["TPM.2.weight"], g["TPM.2.bias"] = conv2d_bwd(k["tp0"], sd["TPM.2.weight"], d, 1, 2)
        d = lrelu_bwd(k["tp0"], d)
        _, g["TPM.0.weight"], g["TPM.0.bias"] = conv2d_bwd(k["tp_in"], sd["TPM.0.weight"], d, 1, 2, need_dx=False)
    # HD
    d, g["HD.4.weight"], g["HD.4.bias"] = conv2d_bwd(k["hd2"], sd["HD.4.weight"], np.ascontiguousarray(dhp), 1, 1)
    d = lrelu_bwd(k["hd2"], d)
    d, g["HD.2.weight"], g["HD.2.bias"] = deconv2d_bwd(k["hd0"], sd["HD.2.weight"], d, 2, 2, 1)
    d = lrelu_bwd(k["hd0"], d)
    dz_hat, g["HD.0.weight"], g["HD.0.bias"] = deconv2d_bwd(k["z_hat"], sd["HD.0.weight"], d, 2, 2, 1)
    # entropy bottleneck
    dv, dpack = eb_likelihood_bwd(k["z_cl"], k["pack"], nchw_to_cl(dlik_z))
    g.update(eb_unpack_grads(dpack))
    dz = dz_hat + cl_to_nchw(dv, dz_hat.shape)
    # HE
    d, g["HE.4.weight"], g["HE.4.bias"] = conv2d_bwd(k["he2"], sd["HE.4.weight"], dz, 2, 2)
    d = lrelu_bwd(k["he2"], d)
    d, g["HE.2.weight"], g["HE.2.bias"] = conv2d_bwd(k["he0"], sd["HE.2.weight"], d, 2, 2)
    d = lrelu_bwd(k["he0"], d)
    _, g["HE.0.weight"], g["HE.0.bias"] = conv2d_bwd(k["he_in"], sd["HE.0.weight"], d, 1, 1, need_dx=False)
    return g


# ---- variable-rate (ROI) pixel-domain models: forward pass (compressai/models/stem_roi.py:353-699, 1017-1325) -----------
def _seq_lrelu(sd, prefix, x, spec, slope):
    """nn.Sequential of (transposed) convolutions with LeakyReLU(slope) between them; spec = [(kind, stride, pad[, opad])]."""
    h = x
    for i, lay in enumerate(spec):
        w, b = sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"]
        h = deconv2d_fwd(h, w, b, lay[1], lay[2], lay[3]) if lay[0] == "T" else conv2d_fwd(h, w, b, lay[1], lay[2])
        if i < len(spec) - 1:
            h = lrelu_fwd(h, slope)
    return h


_QHEAD = [("C", 1, 1)] * 3                                   # qmap_feature_{ga1,ha1,gs0}: three 3x3 stride-1 convolutions
_QDOWN = [("C", 2, 1), ("C", 1, 0)]                          # qmap_feature_{ga2..4,ha2,ha3}: 3x3 stride 2, then 1x1
_QUP = [("T", 2, 1, 1), ("C", 1, 0)]                         # qmap_feature_gs{1..3}: 3x3 stride-2 transposed, then 1x1
_UP2 = [("T", 2, 2, 1), ("T", 2, 2, 1), ("C", 1, 1)]         # hs / wmap_generator
_CHAIN5 = [("C", 1, 2)] * 3                                  # TPM
_CHAIN1 = [("C", 1, 0)] * 3                                  # EPM


def _conv_gdn(sd, prefix, x, inverse):
    """nn.Sequential(conv | deconv (5x5, stride 2), GDN)"""
    w, b = sd[prefix + "0.weight"], sd[prefix + "0.bias"]
    h = deconv2d_fwd(x, w, b, 2, 2, 1) if inverse else conv2d_fwd(x, w, b, 2, 2)
    return gdn_fwd(h, sd[prefix + "1.beta"], sd[prefix + "1.gamma"], inverse=inverse)


def stem_roi_forward(sd, x_cur, x_cond, qmap, noise, temporal=True):
    """stem_roi.forward (temporal=True; :585-608) / stem_roi_i.forward (temporal=False) in training mode.
    noise = {"z": [B,256,h,w] (already in NCHW), "y": [B,192,H/16,W/16]}.  -> dict(x_hat, y_hat, lik_y, lik_z)."""
    sd = {n: np.asarray(v, np.float32) for n, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    sft = lambda pre, x, q, slope=1.0: sft_module_fwd(sd, pre + ".", x, q, slope)[0]
    res = lambda pre, x, q: sft_resblk_fwd(sd, pre + ".", x, q)[0]
    # PEncoder (:534-550)
    q = _seq_lrelu(sd, "qmap_feature_ga1.", np.concatenate([x_cur, qmap], 1), _QHEAD, 0.1)
    x = x_cur
    for i in (1, 2, 3):
        if i > 1:
            q = _seq_lrelu(sd, f"qmap_feature_ga{i}.", q, _QDOWN, 0.1)
        x = sft(f"ga{i}_SFT", _conv_gdn(sd, f"ga{i}.", x, False), q)
    q = _seq_lrelu(sd, "qmap_feature_ga4.", q, _QDOWN, 0.1)
    x = conv2d_fwd(x, sd["ga4.weight"], sd["ga4.bias"], 2, 2)
    y_cur = res("ga4_SFTResB2", res("ga4_SFTResB1", x, q), q)
    y_cond = None
    if temporal:                                             # ConditionEncoder (:503-511): the mbt2018 analysis ladder
        y_cond = g_a(sd, x_cond, prefix="ConditionEncoder.")
    # HE (:575-593)
    yy = np.concatenate([y_cur, y_cond], 1) if temporal else y_cur
    q = _seq_lrelu(sd, "qmap_feature_ha1.", np.concatenate([avgpool_fwd(qmap, yy.shape[2], yy.shape[3]), yy], 1), _QHEAD, 0.1)
    x = sft("ha1_SFT", conv2d_fwd(yy, sd["ha1.weight"], sd["ha1.bias"], 1, 1), q, 0.01)
    q = _seq_lrelu(sd, "qmap_feature_ha2.", q, _QDOWN, 0.1)
    x = sft("ha2_SFT", conv2d_fwd(x, sd["ha2.weight"], sd["ha2.bias"], 2, 2), q, 0.01)
    q = _seq_lrelu(sd, "qmap_feature_ha3.", q, _QDOWN, 0.1)
    x = conv2d_fwd(x, sd["ha3.weight"], sd["ha3.bias"], 2, 2)
    z = res("ha3_ResB2", res("ha3_ResB1", x, q), q)
    # entropy bottleneck in training mode (noise), hyper decoder, priors
    z_hat = z + noise["z"]
    lik_z = cl_to_nchw(eb_likelihood_fwd(nchw_to_cl(z_hat), eb_pack_params(sd)), z.shape)
    hyper = _seq_lrelu(sd, "hs.", z_hat, _UP2, 0.01)
    if temporal:
        gp = _seq_lrelu(sd, "EPM.", np.concatenate([_seq_lrelu(sd, "TPM.", y_cond, _CHAIN5, 0.01), hyper], 1), _CHAIN1, 0.01)
    else:
        gp = _seq_lrelu(sd, "EPM.", hyper, _CHAIN1, 0.01)
    M = y_cur.shape[1]
    scales, means = gp[:, :M], gp[:, M:]
    y_hat = y_cur + noise["y"]                               # quantize(y, "noise"): the means do not enter (:128-135)
    lik_y = gc_likelihood_fwd(y_hat, scales, means)
    # PDecoder (:552-573)
    w = _seq_lrelu(sd, "qmap_feature_gs0.", np.concatenate([_seq_lrelu(sd, "wmap_generator.", z_hat, _UP2, 0.01), y_hat], 1), _QHEAD, 0.1)
    x = res("gs0_SFTResB2", res("gs0_SFTResB1", y_hat, w), w)
    for i in (1, 2, 3):
        w = _seq_lrelu(sd, f"qmap_feature_gs{i}.", w, _QUP, 0.1)
        x = sft(f"gs{i}_SFT", _conv_gdn(sd, f"gs{i}.", x, True), w)
    x_hat = deconv2d_fwd(x, sd["gs4.weight"], sd["gs4.bias"], 2, 2, 1)
    return {"x_hat": x_hat, "y_hat": y_hat, "lik_y": lik_y, "lik_z": lik_z}
