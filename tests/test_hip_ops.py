"""GPU parity tests, op level: HIP kernels (through the C ABI) vs the CPU oracle and vs the golden
vectors captured from the reference.  Tolerance: 1e-4 relative fp32 (north_star); integer outputs exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_close

sys.path.insert(0, os.path.join(REPO, "oracle"))
import stem_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from spatiotemporalentropymodel_amd import functional
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return functional


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if t.dim() == 4 and t.dtype == torch.float32:
        return t.contiguous(memory_format=torch.channels_last)
    return t


def host(t):
    return t.detach().cpu().contiguous().numpy() if t.dim() != 4 else t.detach().permute(0, 1, 2, 3).cpu().contiguous().numpy()


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32)


# ------------------------------------------------------------------ golden op cases (odd sizes: scalar-path kernels)
@pytest.mark.parametrize("name", ["conv_k5s2", "conv_k5s1", "conv_k3s1", "conv_k1s1", "conv_c3"])
def test_conv_golden(F, golden, name):
    g = golden("ops_small.npz")
    N, C, H, W, K, R, st, pd = (int(v) for v in g[f"{name}:cfg"])
    x, w, b, dy = dev(g[f"{name}:x"]), dev(g[f"{name}:w"]), dev(g[f"{name}:b"]), dev(g[f"{name}:dy"])
    y = F.conv2d_fwd(x, F.pack_weight(w, F.PACK_CONV_FWD), b, K, R, R, st, pd)
    assert_close(host(y), g[f"{name}:y"], what=name + " y", floor=0.1)
    dx = F.conv2d_dgrad(dy, F.pack_weight(w, F.PACK_CONV_DGRAD), x.shape, K, R, R, st, pd)
    assert_close(host(dx), g[f"{name}:dx"], what=name + " dx", floor=0.1)
    dw, db = F.conv2d_wgrad(x, dy, K, R, R, st, pd)
    assert_close(host(dw), g[f"{name}:dw"], what=name + " dw", floor=0.1)
    assert_close(host(db), g[f"{name}:db"], what=name + " db", floor=0.1)


@pytest.mark.parametrize("name", ["deconv_k5s2", "deconv_c3"])
def test_deconv_golden(F, golden, name):
    g = golden("ops_small.npz")
    N, C, H, W, K, R, st, pd, op = (int(v) for v in g[f"{name}:cfg"])
    x, w, b, dy = dev(g[f"{name}:x"]), dev(g[f"{name}:w"]), dev(g[f"{name}:b"]), dev(g[f"{name}:dy"])
    y = F.deconv2d_fwd(x, F.pack_weight(w, F.PACK_DECONV_FWD), b, K, R, R, st, pd, op)
    assert_close(host(y), g[f"{name}:y"], what=name + " y", floor=0.1)
    dx = F.deconv2d_dgrad(dy, F.pack_weight(w, F.PACK_DECONV_DGRAD), x.shape, K, R, R, st, pd, op)
    assert_close(host(dx), g[f"{name}:dx"], what=name + " dx", floor=0.1)
    dw, db = F.deconv2d_wgrad(x, dy, K, R, R, st, pd, op)
    assert_close(host(dw), g[f"{name}:dw"], what=name + " dw", floor=0.1)
    assert_close(host(db), g[f"{name}:db"], what=name + " db", floor=0.1)


def test_masked_conv_golden(F, golden):
    g = golden("ops_small.npz")
    n = "masked_k5"
    x, w, b, dy = dev(g[f"{n}:x"]), dev(g[f"{n}:w_before"]), dev(g[f"{n}:b"]), dev(g[f"{n}:dy"])
    y = F.conv2d_fwd(x, F.pack_weight(w, F.PACK_CONV_FWD, masked=True), b, 10, 5, 5, 1, 2)
    assert_close(host(y), g[f"{n}:y"], what="masked y", floor=0.1)
    dx = F.conv2d_dgrad(dy, F.pack_weight(w, F.PACK_CONV_DGRAD, masked=True), x.shape, 10, 5, 5, 1, 2)
    assert_close(host(dx), g[f"{n}:dx"], what="masked dx", floor=0.1)
    dw, _ = F.conv2d_wgrad(x, dy, 10, 5, 5, 1, 2)
    assert_close(host(dw), g[f"{n}:dw"], what="masked dw: all 25 taps, unmasked (layers.py:44-47)", floor=0.1)


@pytest.mark.parametrize("mask_type", ["A", "B"])
def test_masked_conv_module_both_types_vs_oracle(mask_type):
    """MaskedConv2d as a module (layers.py:21-47): type A (the STEM context model) and type B (centre tap kept): forward and
    input gradient use the masked weight, the weight gradient covers all taps, and `weight` itself is zeroed in place."""
    from spatiotemporalentropymodel_amd.layers import MaskedConv2d
    torch.manual_seed(3)
    m = MaskedConv2d(8, 12, kernel_size=5, padding=2, stride=1, mask_type=mask_type).cuda()
    with torch.no_grad():
        m.weight.copy_(torch.randn_like(m.weight))
    w_before = m.weight.detach().cpu().numpy().copy()
    mask = m.mask.cpu().numpy()
    assert mask[0, 0].sum() == (12 if mask_type == "A" else 13) and mask[0, 0, 2, 2] == (0 if mask_type == "A" else 1)
    x = torch.randn(2, 8, 7, 6, device="cuda", requires_grad=True)
    y = m(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    wm = w_before * mask
    np.testing.assert_array_equal(m.weight.detach().cpu().numpy(), wm)            # `self.weight.data *= self.mask`
    xn, dyn, bn = host(x), host(dy), host(m.bias)
    assert_close(host(y), orc.conv2d_fwd(xn, wm, bn, 1, 2), what=f"masked {mask_type} forward", floor=0.1)
    rdx, rdw, rdb = orc.conv2d_bwd(xn, wm, dyn, 1, 2)
    assert_close(host(x.grad), rdx, what="dx through the masked weight", floor=0.1)
    assert_close(host(m.weight.grad), rdw, what="dw: every tap, unmasked", floor=0.1)
    assert float(np.abs(host(m.weight.grad) * (1 - mask)).max()) > 0


def test_gdn_golden(F, golden):
    g = golden("ops_small.npz")
    for n, inv in (("gdn", False), ("igdn", True)):
        y = F.gdn_fwd(dev(g[f"{n}:x"]), dev(g[f"{n}:beta"]), dev(g[f"{n}:gamma"]), inverse=inv)
        assert_close(host(y), g[f"{n}:y"], what=n, floor=0.1)
    x = g["gdn_init:x"]
    y = F.gdn_fwd(dev(x), dev(g["gdn_init:beta"]), dev(g["gdn_init:gamma"]))
    assert_close(host(y), x / np.sqrt(1 + 0.1 * x ** 2), what="GDN closed form (compressai_tests/test_layers.py:118-156)", floor=0.1)


# ------------------------------------------------------------------ production shapes vs the oracle (vector-path kernels)
CONV_SHAPES = [
    # B, C, H, W, K, R, stride, pad      (which layer)
    (2, 192, 16, 16, 256, 5, 1, 2),    # TPM.0
    (1, 320, 8, 8, 384, 5, 1, 2),      # TPM.4 (K tile tail: 384 = 3 x 128; C = 10 chunks)
    (2, 384, 8, 8, 256, 3, 1, 1),      # HE.0
    (2, 256, 16, 16, 256, 5, 2, 2),    # HE.2
    (2, 1152, 8, 8, 768, 1, 1, 0),     # EPM.0
    (1, 192, 32, 32, 192, 5, 2, 2),    # g_a.2 shape (north-star conv), reduced spatially
    (3, 96, 5, 7, 192, 5, 1, 2),       # small model ctx, ragged spatial size
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_vs_oracle(F, shape):
    B, C, H, W, K, R, st, pd = shape
    x, w, b = rnd((B, C, H, W), 1), rnd((K, C, R, R), 2, -0.05, 0.05), rnd((K,), 3)
    y_ref = orc.conv2d_fwd(x, w, b, st, pd)
    dy = rnd(y_ref.shape, 4)
    dx_ref, dw_ref, db_ref = orc.conv2d_bwd(x, w, dy, st, pd)
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y = F.conv2d_fwd(xd, F.pack_weight(wd, F.PACK_CONV_FWD), bd, K, R, R, st, pd)
    assert_close(host(y), y_ref, what="y", floor=0.1)
    ya = F.conv2d_fwd(xd, F.pack_weight(wd, F.PACK_CONV_FWD), bd, K, R, R, st, pd, act=F.ACT_LRELU)
    assert_close(host(ya), orc.lrelu_fwd(y_ref), what="fused LeakyReLU", floor=0.1)
    dx = F.conv2d_dgrad(dyd, F.pack_weight(wd, F.PACK_CONV_DGRAD), xd.shape, K, R, R, st, pd)
    assert_close(host(dx), dx_ref, what="dx", floor=0.1)
    xact = rnd((B, C, H, W), 5)
    dxa = F.conv2d_dgrad(dyd, F.pack_weight(wd, F.PACK_CONV_DGRAD), xd.shape, K, R, R, st, pd, xact=dev(xact))
    assert_close(host(dxa), orc.lrelu_bwd(xact, dx_ref), what="dx with fused LeakyReLU'", floor=0.1)
    dw, db = F.conv2d_wgrad(xd, dyd, K, R, R, st, pd)
    assert_close(host(dw), dw_ref, what="dw", floor=0.1)
    assert_close(host(db), db_ref, what="db", floor=0.1)


@pytest.mark.parametrize("shape", [(2, 256, 4, 4, 256, 5, 2, 2, 1), (2, 64, 3, 5, 256, 5, 2, 2, 1), (1, 192, 8, 8, 192, 5, 2, 2, 1)])
def test_deconv_vs_oracle(F, shape):
    B, C, H, W, K, R, st, pd, op = shape
    x, w, b = rnd((B, C, H, W), 11), rnd((C, K, R, R), 12, -0.05, 0.05), rnd((K,), 13)
    y_ref = orc.deconv2d_fwd(x, w, b, st, pd, op)
    dy = rnd(y_ref.shape, 14)
    dx_ref, dw_ref, db_ref = orc.deconv2d_bwd(x, w, dy, st, pd, op)
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y = F.deconv2d_fwd(xd, F.pack_weight(wd, F.PACK_DECONV_FWD), bd, K, R, R, st, pd, op, act=F.ACT_LRELU)
    assert_close(host(y), orc.lrelu_fwd(y_ref), what="y", floor=0.1)
    dx = F.deconv2d_dgrad(dyd, F.pack_weight(wd, F.PACK_DECONV_DGRAD), xd.shape, K, R, R, st, pd, op)
    assert_close(host(dx), dx_ref, what="dx", floor=0.1)
    dw, db = F.deconv2d_wgrad(xd, dyd, K, R, R, st, pd, op)
    assert_close(host(dw), dw_ref, what="dw", floor=0.1)
    assert_close(host(db), db_ref, what="db", floor=0.1)


def test_first_layer_c4(F):
    B, H, W, K = 2, 64, 48, 192
    x, w, b = rnd((B, 3, H, W), 21, 0, 1), rnd((K, 3, 5, 5), 22, -0.2, 0.2), rnd((K,), 23)
    y_ref = orc.conv2d_fwd(x, w, b, 2, 2)
    x4 = F.nchw3_to_nhwc4(torch.from_numpy(x).cuda())
    y = F.conv2d_fwd_c4(x4, F.pack_weight(dev(w), F.PACK_CONV_FWD_C4), dev(b), K, 5, 5, 2, 2)
    assert_close(host(y), y_ref, what="g_a.0 (3-channel input, NCHW -> NHWC fused)", floor=0.1)


def test_gdn_vs_oracle(F):
    B, C, H, W = 2, 192, 16, 12
    x = rnd((B, C, H, W), 31, -3, 3)
    beta = np.sqrt(1 + 0.2 * rnd((C,), 32) + 2.0 ** -36).astype(np.float32)
    gamma = np.sqrt(0.1 * np.eye(C) + 0.02 * np.abs(rnd((C, C), 33)) + 2.0 ** -36).astype(np.float32)
    gamma[0, :5] = 0.0          # below the reparametrisation bound -> clamped
    for inv in (False, True):
        y = F.gdn_fwd(dev(x), dev(beta), dev(gamma), inverse=inv)
        assert_close(host(y), orc.gdn_fwd(x, beta, gamma, inverse=inv), what=f"gdn inverse={inv}", floor=0.1)


def test_gdn_backward(F, golden):
    """dx, dbeta, dgamma (stored parameters, through the reparametrisation and its LowerBound rule)."""
    g = golden("ops_small.npz")
    for n, inv in (("gdn", False), ("igdn", True)):
        dx, db, dg = F.gdn_bwd(dev(g[f"{n}:x"]), dev(g[f"{n}:dy"]), dev(g[f"{n}:beta"]), dev(g[f"{n}:gamma"]), inverse=inv)
        assert_close(host(dx), g[f"{n}:dx"], what=n + " dx", floor=0.1)
        assert_close(host(db), g[f"{n}:dbeta"], what=n + " dbeta", floor=0.1)
        assert_close(host(dg), g[f"{n}:dgamma"], what=n + " dgamma", floor=0.1)
    B, C, H, W = 2, 192, 12, 10
    x, dy = rnd((B, C, H, W), 131, -3, 3), rnd((B, C, H, W), 132)
    beta = np.sqrt(1 + 0.2 * rnd((C,), 32) + 2.0 ** -36).astype(np.float32)
    gamma = np.sqrt(0.1 * np.eye(C) + 0.02 * np.abs(rnd((C, C), 33)) + 2.0 ** -36).astype(np.float32)
    gamma[3, :7] = 0.0                       # below the bound: gradient only passes when it pushes the value up
    beta[5] = 0.0
    for inv in (False, True):
        dx, db, dg = F.gdn_bwd(dev(x), dev(dy), dev(beta), dev(gamma), inverse=inv)
        rx, rb, rg = orc.gdn_bwd(x, dy, beta, gamma, inverse=inv)
        assert_close(host(dx), rx, what=f"dx inverse={inv}", floor=0.1)
        assert_close(host(db), rb, what=f"dbeta inverse={inv}", floor=0.1)
        assert_close(host(dg), rg, what=f"dgamma inverse={inv}", floor=0.1)


def test_fused_conv_gdn_vs_oracle(F):
    """conv -> GDN and deconv -> IGDN in one kernel (g_a / g_s inference path) == the two separate oracle ops."""
    C = 192
    beta = np.sqrt(1 + 0.2 * rnd((C,), 32) + 2.0 ** -36).astype(np.float32)
    gamma = np.sqrt(0.1 * np.eye(C) + 0.02 * np.abs(rnd((C, C), 33)) + 2.0 ** -36).astype(np.float32)
    x, w, b = rnd((2, 192, 20, 12), 34), rnd((C, 192, 5, 5), 35, -0.03, 0.03), rnd((C,), 36)
    ref = orc.gdn_fwd(orc.conv2d_fwd(x, w, b, 2, 2), beta, gamma, inverse=False)
    y = F.conv2d_gdn_fwd(dev(x), F.pack_weight(dev(w), F.PACK_CONV_FWD), dev(b), dev(beta), dev(gamma), C, 5, 5, 2, 2)
    assert_close(host(y), ref, what="conv+GDN", floor=0.1)
    x3, w3 = rnd((2, 3, 40, 24), 37, 0, 1), rnd((C, 3, 5, 5), 38, -0.2, 0.2)
    ref = orc.gdn_fwd(orc.conv2d_fwd(x3, w3, b, 2, 2), beta, gamma, inverse=False)
    y = F.conv2d_fwd_c4_gdn(F.nchw3_to_nhwc4(torch.from_numpy(x3).cuda()), F.pack_weight(dev(w3), F.PACK_CONV_FWD_C4), dev(b),
                            dev(beta), dev(gamma), C, 5, 5, 2, 2)
    assert_close(host(y), ref, what="first layer conv+GDN", floor=0.1)
    xd, wd = rnd((1, 192, 7, 9), 39), rnd((192, C, 5, 5), 40, -0.03, 0.03)
    ref = orc.gdn_fwd(orc.deconv2d_fwd(xd, wd, b, 2, 2, 1), beta, gamma, inverse=True)
    y = F.deconv2d_gdn_fwd(dev(xd), F.pack_weight(dev(wd), F.PACK_DECONV_FWD), dev(b), dev(beta), dev(gamma), C, 5, 5, 2, 2, 1, inverse=True)
    assert_close(host(y), ref, what="deconv+IGDN", floor=0.1)
    # fewer channels than the 192-wide tile (small model: N = 64)
    C2 = 64
    w2, b2 = rnd((C2, 64, 5, 5), 41, -0.05, 0.05), rnd((C2,), 42)
    x2 = rnd((2, 64, 10, 10), 43)
    ref = orc.gdn_fwd(orc.conv2d_fwd(x2, w2, b2, 2, 2), beta[:C2], gamma[:C2, :C2].copy(), inverse=False)
    y = F.conv2d_gdn_fwd(dev(x2), F.pack_weight(dev(w2), F.PACK_CONV_FWD), dev(b2), dev(beta[:C2].copy()), dev(gamma[:C2, :C2].copy()), C2, 5, 5, 2, 2)
    assert_close(host(y), ref, what="conv+GDN, N=64", floor=0.1)


def test_channel_slice_views(F):
    """torch.cat / chunk along channels are pitch views: conv reads a slice and writes into a slice."""
    B, H, W = 2, 8, 8
    x = rnd((B, 96, H, W), 41)
    w, b = rnd((64, 32, 3, 3), 42, -0.1, 0.1), rnd((64,), 43)
    xd = dev(x)
    buf = F.empty_nhwc(B, 160, H, W, xd.device)
    buf.zero_()
    out = F.channel_slice(buf, 96, 160)
    F.conv2d_fwd(F.channel_slice(xd, 64, 96), F.pack_weight(dev(w), F.PACK_CONV_FWD), dev(b), 64, 3, 3, 1, 1, out=out)
    ref = orc.conv2d_fwd(x[:, 64:96], w, b, 1, 1)
    assert_close(host(buf)[:, 96:], ref, what="slice in/out", floor=0.1)
    assert float(host(buf)[:, :96].__abs__().max()) == 0.0


def test_layout_roundtrip(F):
    x = rnd((3, 37, 9, 11), 51)
    xd = torch.from_numpy(x).cuda()
    y = F.to_nhwc(xd)
    assert F.nhwc_ld(y) == 37
    np.testing.assert_array_equal(host(y), x)
    np.testing.assert_array_equal(F.to_nchw(y).cpu().numpy(), x)
    np.testing.assert_array_equal(F.to_nchw(y, clamp01=True).cpu().numpy(), np.clip(x, 0, 1))


# ------------------------------------------------------------------ entropy models
def _eb_tensors(sd):
    from spatiotemporalentropymodel_amd.functional import EB_TENSORS
    return [dev(sd[n]) for n in EB_TENSORS]


def test_entropy_bottleneck_golden(F, golden):
    from spatiotemporalentropymodel_amd.weights import closed_form_input
    g = golden("ops_small.npz")
    sd = {k[len("eb:p:"):]: v for k, v in g.items() if k.startswith("eb:p:")}
    pack = F.eb_pack(_eb_tensors(sd))
    np.testing.assert_array_equal(pack.cpu().numpy(), orc.eb_pack_params(sd, prefix=""))
    x = g["eb:x"]
    noise_cl = closed_form_input("noise:eb:0", (4, 1, 2 * 3 * 5), -0.5, 0.5).numpy().reshape(4, -1)
    noise = orc.cl_to_nchw(noise_cl, x.shape)
    z_hat, lik = F.eb_forward(dev(x), pack, noise=dev(noise))
    assert_close(host(z_hat), g["eb:train_out"], 1e-6, what="eb noisy out", floor=0.1)
    assert_close(host(lik), g["eb:train_lik"], atol=1e-9, what="eb train lik", floor=0.1)
    dz, dpack = F.eb_backward(z_hat, pack, dev(g["eb:dlik"]))
    assert_close(host(dz), g["eb:dx"], what="eb dx", floor=0.1)
    for name, gr in orc.eb_unpack_grads(dpack.cpu().numpy(), prefix="").items():
        assert_close(gr, g[f"eb:g:{name}"], what="eb grad " + name, floor=0.1)
    med = dev(np.ascontiguousarray(sd["quantiles"][:, 0, 1]))
    z_hat, lik = F.eb_forward(dev(x), pack, medians=med)
    np.testing.assert_array_equal(host(z_hat), g["eb:eval_out"])
    assert_close(host(lik), g["eb:eval_lik"], atol=1e-9, what="eb eval lik", floor=0.1)
    target = dev(np.array([-np.log(2 / 1e-9 - 1), 0, np.log(2 / 1e-9 - 1)], np.float32))
    loss, dq = F.eb_aux_loss(dev(sd["quantiles"]), pack, target)
    assert_close(loss.cpu().numpy()[0], g["eb:aux"], what="aux loss", floor=0.1)
    assert_close(dq.cpu().numpy(), g["eb:aux_dquantiles"], what="aux dquantiles", floor=0.1)


def test_gaussian_conditional_golden(F, golden):
    from spatiotemporalentropymodel_amd.weights import closed_form_input
    g = golden("ops_small.npz")
    y, sc, mu = g["gc:y"], g["gc:scales"], g["gc:means"]
    B, C, H, W = y.shape
    gp = F.empty_nhwc(B, 2 * C, H, W, "cuda")          # scales | means as channel slices, like EPM's output
    gp[:, :C] = dev(sc)
    gp[:, C:] = dev(mu)
    noise = closed_form_input("noise:gc:0", y.shape, -0.5, 0.5).numpy()
    out, lik = F.gc_forward(dev(y), gp[:, :C], gp[:, C:], noise=dev(noise))
    assert_close(host(out), g["gc:train_out"], 1e-6, what="gc noisy", floor=0.1)
    assert_close(host(lik), g["gc:train_lik"], atol=1e-9, what="gc lik", floor=0.1)
    dgp = F.empty_nhwc(B, 2 * C, H, W, "cuda")
    dy = F.empty_nhwc(B, C, H, W, "cuda")
    F.gc_backward(out, gp[:, :C], gp[:, C:], dev(g["gc:dlik"]), dgp[:, :C], dgp[:, C:], dy=dy)
    assert_close(host(dy), g["gc:dy"], atol=1e-9, what="gc dy", floor=0.1)
    assert_close(host(dgp[:, :C]), g["gc:dscales"], atol=1e-9, what="gc dscales", floor=0.1)
    assert_close(host(dgp[:, C:]), g["gc:dmeans"], atol=1e-9, what="gc dmeans", floor=0.1)
    out, lik = F.gc_forward(dev(y), gp[:, :C], gp[:, C:])
    np.testing.assert_array_equal(host(out), g["gc:eval_out"])
    assert_close(host(lik), g["gc:eval_lik"], atol=1e-9, what="gc eval lik", floor=0.1)


def test_build_indexes_and_rate(F, golden):
    g = golden("codec.npz")
    table = g["gc:scale_table"]
    sc = rnd((2, 24, 5, 6), 61, -0.5, 300.0)
    idx = F.build_indexes(dev(sc), dev(table))
    np.testing.assert_array_equal(host(idx), orc.build_indexes(sc, table))
    lik = rnd((2, 24, 5, 6), 62, 1e-9, 1.0)
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    F.log2_sum(dev(lik), acc)
    ref = orc.rate_bpp(lik, 1.0) * -1.0          # = sum(log2 lik)
    assert abs(acc.item() - ref) < 1e-5 * abs(ref)
    assert_close(host(F.dlog(dev(lik), 0.25)), 0.25 / lik, 1e-6, what="dlog", floor=0.1)


def test_elementwise_and_noise(F):
    a, b = rnd((2, 8, 4, 4), 71, -5, 5), rnd((2, 8, 4, 4), 72, -5, 5)
    np.testing.assert_array_equal(host(F.sub(dev(a), dev(b))), a - b)
    np.testing.assert_array_equal(host(F.add(dev(a), dev(b))), a + b)
    h = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 7.0, -3.2], np.float32).reshape(1, 8, 1, 1)
    np.testing.assert_array_equal(host(F.round_(dev(h))).ravel(), np.rint(h).ravel())       # half to even
    big = F.empty_nhwc(4, 64, 32, 32, "cuda")
    n1 = F.uniform_noise_like(big, seed=1234, offset=0).cpu().numpy()
    n2 = F.uniform_noise_like(big, seed=1234, offset=0).cpu().numpy()
    n3 = F.uniform_noise_like(big, seed=1235, offset=0).cpu().numpy()
    np.testing.assert_array_equal(n1, n2)
    assert (n1 != n3).mean() > 0.99
    assert n1.min() >= -0.5 and n1.max() < 0.5
    assert abs(n1.mean()) < 2e-3 and abs(n1.var() - 1 / 12) < 2e-3


def test_fused_clip_adam_matches_torch(F):
    n = 100003
    p0, g0 = rnd((n,), 81), rnd((n,), 82, -3, 3)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref], lr=1e-4)
    p, m, v = dev(p0.copy()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in (1, 2, 3):
        gs = g0 * step
        ref.grad = torch.from_numpy(gs.copy())
        total = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        acc = F.sumsq_accumulator("cuda")
        gd = dev(gs)
        F.sumsq(gd, acc)
        acc2 = F.sumsq_accumulator("cuda")
        F.sumsq(gd, acc2)
        assert acc[0].item() == acc2[0].item()                        # two-stage reduction without atomics: bit-reproducible
        assert abs(np.sqrt(acc[0].item()) - float(total)) < 1e-5 * float(total)
        F.adam_step(p, gd, m, v, acc, 1.0, 1.0, 1e-4, 0.9, 0.999, 1e-8, step)
        assert_close(p.cpu().numpy(), ref.detach().numpy(), 1e-6, what=f"adam step {step}", floor=0.1)


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        k = int(rng.choice([1, 3, 5]))
        st = int(rng.choice([1, 2])) if k > 1 else 1
        up = bool(rng.integers(0, 2)) and st == 2
        cin = int(rng.choice([3, 4, 5, 8, 16, 36, 64, 100, 128, 160, 192]))
        cout = int(rng.choice([3, 4, 6, 8, 32, 64, 96, 128, 130, 192]))
        H, W = int(rng.integers(1, 20)), int(rng.integers(1, 24))
        if not up and (H + 2 * (k // 2) < k or W + 2 * (k // 2) < k):
            continue
        B = int(rng.integers(1, 4))
        cases.append((B, cin, cout, k, st, up, H, W))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(36, 20261002), ids=lambda c: "B%d_c%d_k%d_%dx%d_s%d_%s_%dx%d" % (c[0], c[1], c[2], c[3], c[3], c[4], "T" if c[5] else "C", c[6], c[7]))
def test_conv_layers_fuzz_vs_oracle(case):
    """Random layer geometries (odd sizes, channel counts off the vector path, 1-pixel images, batch 1..3) through the
    layer modules: forward, input gradient, weight gradient and bias gradient vs the oracle.  Guards the tile / split /
    folded-tap / fused-bias / 8-wavefront variants the planner may pick."""
    from spatiotemporalentropymodel_amd.layers import conv, deconv
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    B, cin, cout, k, st, up, H, W = case
    m = closed_form_fill_((deconv if up else conv)(cin, cout, k, st)).cuda()
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=gen)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    w, b = host(m.weight), host(m.bias)
    pad = k // 2
    ref = orc.deconv2d_fwd(x.numpy(), w, b, st, pad, st - 1) if up else orc.conv2d_fwd(x.numpy(), w, b, st, pad)
    assert_close(host(y), ref, what="forward", floor=0.1)
    dy = torch.randn(*ref.shape, generator=gen)
    y.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
    rdx, rdw, rdb = (orc.deconv2d_bwd(x.numpy(), w, dy.numpy(), st, pad, st - 1) if up else orc.conv2d_bwd(x.numpy(), w, dy.numpy(), st, pad))
    assert_close(host(xg.grad), rdx, what="input gradient", floor=0.1)
    assert_close(host(m.weight.grad), rdw, what="weight gradient", floor=0.1)
    assert_close(host(m.bias.grad), rdb, what="bias gradient", floor=0.1)


@pytest.mark.parametrize("case", _fuzz_cases(14, 77), ids=lambda c: "B%d_c%d_k%d_%dx%d_s%d_%s_%dx%d" % (c[0], c[1], c[2], c[3], c[3], c[4], "T" if c[5] else "C", c[6], c[7]))
def test_conv_layers_fuzz_flat_gradient_accumulation(case):
    """Same random geometries with the parameters living in optim.FlatParameters: two backward passes accumulate
    straight into the flat gradient buffer (unpack / column-sum kernels with the accumulate flag, on the weight-gradient
    side stream) and must equal the sum of the two oracle gradients."""
    from spatiotemporalentropymodel_amd.layers import conv, deconv, join_wgrad_stream
    from spatiotemporalentropymodel_amd.optim import FlatParameters
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    B, cin, cout, k, st, up, H, W = case
    m = closed_form_fill_((deconv if up else conv)(cin, cout, k, st)).cuda()
    flat = FlatParameters(sorted(m.named_parameters(), key=lambda t: t[0]))
    flat.zero_grad()
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31) + 1)
    w, b = host(m.weight), host(m.bias)
    pad = k // 2
    tot_w, tot_b = 0.0, 0.0
    for _ in range(2):
        x = torch.randn(B, cin, H, W, generator=gen)
        y = m(x.cuda())
        dy = torch.randn(*y.shape, generator=gen)
        y.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
        _, rdw, rdb = (orc.deconv2d_bwd(x.numpy(), w, dy.numpy(), st, pad, st - 1, need_dx=False) if up
                       else orc.conv2d_bwd(x.numpy(), w, dy.numpy(), st, pad, need_dx=False))
        tot_w, tot_b = tot_w + rdw.astype(np.float64), tot_b + rdb.astype(np.float64)
    join_wgrad_stream()
    assert m.weight.grad is m.weight._flat_grad_view and m.weight.grad.data_ptr() >= flat.grad.data_ptr()
    assert_close(host(m.weight.grad), tot_w, what="accumulated weight gradient", floor=0.1)
    assert_close(host(m.bias.grad), tot_b, what="accumulated bias gradient", floor=0.1)


def test_gaussian_conditional_without_means(F):
    """GaussianConditional.forward(inputs, scales) with means=None (entropy_models.py:570-596: quantisation without an offset, the
    likelihood of the values themselves): training (noise) and eval (rounding) outputs and likelihoods against the oracle with a
    zero mean, gradients flow to inputs and scales."""
    from spatiotemporalentropymodel_amd.entropy_models import GaussianConditional
    from spatiotemporalentropymodel_amd.weights import closed_form_input
    y, sc = rnd((2, 32, 6, 5), 81, -6, 6), rnd((2, 32, 6, 5), 82, 0.05, 4.0)
    gc = GaussianConditional(None).cuda()
    noise = closed_form_input("noise:gc_nomeans:0", y.shape, -0.5, 0.5)
    gc.noise_source = lambda shape, device: noise.to(device)
    gc.train()
    yt, st = dev(y).requires_grad_(), dev(sc).requires_grad_()
    out, lik = gc(yt, st)
    ref_lik = orc.gc_likelihood_fwd(y + noise.numpy(), sc, None)           # the oracle follows the reference's own means=None branch
    np.testing.assert_array_equal(host(out), (y + noise.numpy()).astype(np.float32))
    assert_close(host(lik), ref_lik, atol=1e-9, what="gc lik without means (training)", floor=0.1)
    lik.sum().backward()
    assert float(yt.grad.abs().max()) > 0 and float(st.grad.abs().max()) > 0
    gc.eval()
    with torch.no_grad():
        out, lik = gc(dev(y), dev(sc))
    np.testing.assert_array_equal(host(out), np.rint(y))
    assert_close(host(lik), orc.gc_likelihood_fwd(np.rint(y), sc, None), atol=1e-9, what="gc lik without means (eval)", floor=0.1)


@pytest.mark.gpu
def test_stream_flag_orders_a_stream_behind_a_later_write(F):
    """stem_stream_flag_*: a stream waits for `flag >= n` although nobody has issued the write yet (the data-parallel reducer's way
    back from its helper thread: distributed._CollectiveIssuer); writes come from another stream or, on the error path, the host."""
    import ctypes
    import time
    from spatiotemporalentropymodel_amd import _lib
    lib = _lib.hip()
    flag = ctypes.c_void_p()
    assert lib.stem_stream_flag_create(ctypes.byref(flag)) == 0 and flag.value
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    t = torch.ones(1 << 16, device="cuda")
    torch.cuda.synchronize()
    try:
        for n, host in ((1, False), (2, True)):
            t.fill_(1.0)
            torch.cuda.synchronize()
            assert lib.stem_stream_flag_wait_ge(flag, n, a.cuda_stream) == 0
            assert lib.stem_zero_bytes(t.data_ptr(), t.numel() * 4, a.cuda_stream) == 0
            time.sleep(0.05)
            assert not a.query()                                   # parked in front of the memset
            assert lib.stem_stream_flag_write(flag, n, ctypes.c_void_p(-1) if host else b.cuda_stream) == 0
            a.synchronize()
            assert float(t.abs().max()) == 0.0
        assert lib.stem_stream_flag_wait_ge(flag, 1, a.cuda_stream) == 0       # already past: no wait
        a.synchronize()
    finally:
        lib.stem_stream_flag_write(flag, 0xffffffff, ctypes.c_void_p(-1))     # whatever happened, nothing stays parked
        torch.cuda.synchronize()
        assert lib.stem_stream_flag_destroy(flag) == 0
