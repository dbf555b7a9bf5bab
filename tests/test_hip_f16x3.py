"""GPU parity tests of the analysis transform on the fp16 matrix cores (csrc/conv_f16x3.hip, through the C ABI): the frozen
g_a chain of the I-frame model (compressai/models/priors.py:613-621 under no_grad, stem/trainSTEM.py:128,171) with every fp32
operand pre-split into two fp16 numbers and three MFMAs per fp32 product.  Same bound as the fp32-MFMA kernels: 1e-4 relative
(north_star), against the CPU oracle and the golden vectors captured from the reference.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_close, close_ratio

sys.path.insert(0, os.path.join(REPO, "oracle"))
import stem_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from spatiotemporalentropymodel_amd import functional
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return functional


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().contiguous().numpy()


def planes_match(yp, y):
    """the planes copy of an output against its fp32 copy: two fp16 numbers per value, |err| <= 2^-22 |y| plus half a unit of the
    second plane's subnormal grid (2^-25 in scaled units = 2^-25 * inv); and the record's maximum is the measured one"""
    inv, rec_max = yp.record()
    err = (yp.merge().double() - y.double()).abs()
    ok = bool((err <= y.double().abs() * 2.0 ** -22 + inv * 2.0 ** -25).all())
    return ok and rec_max == float(y.abs().max()) and float(y.abs().max()) / inv < 2.0 ** 15


@pytest.mark.parametrize("mag", [1.0, 1e-6, 3e4, 1e30])
def test_split_is_a_scaled_fp16_pair(F, mag):
    """x * 2^e = p0 + p1 to 2^-22 |x| 2^e with p0 = rn16(x * 2^e); 2^e puts the tensor's maximum into [2^14, 2^15) whatever its
    magnitude (fp16 alone would overflow at 65504 and lose the second plane below 2^-3); the record holds 2^-e and the maximum."""
    x = (rnd((3, 64, 5, 7), 1, -4, 4) * np.float32(mag)).astype(np.float32)
    x.reshape(-1)[:6] = np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -20, -3.0], np.float32) * np.float32(mag)
    amax = float(np.abs(x).max())
    xp = F.F16Planes.split(dev(x))
    inv, rec_max = xp.record()
    assert rec_max == amax and np.log2(inv) == np.round(np.log2(inv)) and 2.0 ** 14 <= amax / inv < 2.0 ** 15
    ref = torch.from_numpy(x).contiguous(memory_format=torch.channels_last)
    err = (xp.merge().cpu().double() - ref.double()).abs()
    assert bool((err <= ref.double().abs() * 2.0 ** -22 + amax * 2.0 ** -39).all()), float(err.max())
    payload = 3 * 5 * 7 * 2 * 128
    raw = xp.data[:payload].view(torch.float16).view(3 * 5 * 7, 2, 2, 32).float().cpu()       # [pixel][slab][plane][32]
    nhwc = torch.from_numpy(x).permute(0, 2, 3, 1).reshape(3 * 5 * 7, 2, 32)
    assert torch.equal(raw[:, :, 0], (nhwc / inv).to(torch.float16).float())             # round-to-nearest-even leading plane
    assert bool(torch.isfinite(raw).all()) and float(raw[:, :, 0].abs().max()) <= 2.0 ** 15
    with pytest.raises(ValueError):
        F.F16Planes.empty(1, 48, 4, 4, torch.device("cuda:0"))


CASES = [  # B, C, H, W, K, R, stride
    (1, 64, 16, 16, 64, 3, 1),
    (2, 192, 20, 28, 192, 5, 2),
    (1, 96, 33, 47, 160, 5, 2),
    (3, 32, 9, 11, 100, 1, 1),        # K not a multiple of 32 (fp32 output only), 1x1
    (2, 128, 24, 24, 192, 3, 2),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile128", [False, True])
@pytest.mark.parametrize("loop", [(2, 32), (3, 32), (3, 16)], ids=["2stage", "3stage", "3stage-mfma16"])
@pytest.mark.parametrize("gdn", [False, True])
def test_conv_gdn_vs_oracle(F, case, tile128, loop, gdn):
    """conv (+ fused GDN) against the oracle, both workgroup tiles (64 pixels x 4 wavefronts, 128 pixels x 8 wavefronts), the
    main-loop forms (two / three LDS stages: a 1x1 case has fewer chunks than the three-stage prologue loads) and both MFMA shapes
    (32x32x16, 16x16x32 with its permuted accumulator layout through every epilogue), fp32 and planes output; ragged pixel
    counts and channel counts below the 192-wide tile."""
    B, C, H, W, K, R, st = case
    with F.tuning(fx3_tile=128 if tile128 else 64, fx3_depth=loop[0], fx3_mfma=loop[1]):
        _conv_gdn_vs_oracle(F, case, gdn)


def _conv_gdn_vs_oracle(F, case, gdn):
    B, C, H, W, K, R, st = case
    x, w, b = rnd((B, C, H, W), 11, -2, 2), (rnd((K, C, R, R), 12) / np.sqrt(C * R * R)).astype(np.float32), rnd((K,), 13, -0.1, 0.1)
    beta, gamma = rnd((K,), 14, 0.5, 1.5), (rnd((K, K), 15, 0.0, 0.1) + 0.1 * np.eye(K, dtype=np.float32)).astype(np.float32)
    ref = orc.conv2d_fwd(x, w, b, st, R // 2)
    if gdn:
        ref = orc.gdn_fwd(ref, beta, gamma)
    xp = F.F16Planes.split(dev(x))
    wp = F.pack_weight_f16x2(dev(w))
    kw = dict(beta=dev(beta), gamma=dev(gamma)) if gdn else {}
    y = F.conv2d_f16x3_fwd(xp, wp, dev(b), K, R, R, st, R // 2, **kw)
    assert_close(host(y), ref, what=f"f16x3 conv {case} gdn={gdn}", floor=0.1)
    if K % 32 == 0:
        yp = F.conv2d_f16x3_fwd(xp, wp, dev(b), K, R, R, st, R // 2, planes_out=True, **kw)
        assert planes_match(yp, y), "planes output != fp32 output"
    else:
        with pytest.raises(Exception):
            F.conv2d_f16x3_fwd(xp, wp, dev(b), K, R, R, st, R // 2, planes_out=True, **kw)


@pytest.mark.parametrize("route", ["f16x3", "fp32-mfma"])
def test_first_layer_writes_the_same_values_as_planes(F, monkeypatch, route):
    """The 3-channel first layer + GDN (csrc/c4gdn_f16x3.hip by default, igemm.hip's fp32-MFMA kernel with STEM_C4GDN_F16X3=0):
    its planes epilogue must hold exactly the fp32 result, and that result is the oracle's."""
    monkeypatch.setenv("STEM_C4GDN_F16X3", "1" if route == "f16x3" else "0")
    x, w, b = rnd((2, 3, 40, 56), 21, 0, 1), (rnd((192, 3, 5, 5), 22) / np.sqrt(75)).astype(np.float32), rnd((192,), 23, -0.1, 0.1)
    beta, gamma = rnd((192,), 24, 0.5, 1.5), rnd((192, 192), 25, 0.0, 0.1)
    x4 = F.nchw3_to_nhwc4(dev(x))
    wp = F.pack_weight(dev(w), F.PACK_CONV_FWD_C4)
    y = F.conv2d_fwd_c4_gdn(x4, wp, dev(b), dev(beta), dev(gamma), 192, 5, 5, 2, 2)
    yp = F.conv2d_fwd_c4_gdn_planes(x4, wp, dev(b), dev(beta), dev(gamma), 192, 5, 5, 2, 2)
    assert planes_match(yp, y)
    assert_close(host(y), orc.gdn_fwd(orc.conv2d_fwd(x, w, b, 2, 2), beta, gamma), what="g_a.0 + GDN", floor=0.1)


C4_CASES = [  # B, H, W, K, R, stride, pad
    (2, 40, 56, 192, 5, 2, 2),
    (1, 33, 47, 64, 5, 2, 2),        # ragged: 17 x 24 outputs, partial last workgroup, N = 64
    (2, 24, 24, 128, 3, 1, 1),       # 3x3 stride 1, N = 128
    (1, 9, 150, 192, 5, 2, 2),       # rows longer than a workgroup tile: tiles that straddle output rows
    (3, 16, 16, 192, 1, 1, 0),       # 1x1: a single conv k-step with one pair
]


@pytest.mark.parametrize("case", C4_CASES)
def test_first_layer_gdn_kernel_vs_oracle(F, case):
    """csrc/c4gdn_f16x3.hip on its own: conv (3 -> N) + GDN with the transposed contractions and the register hand-over of
    the squared outputs, fp32 and planes output, against the oracle (priors.py:421-423, gdn.py:52-67); image borders, ragged
    tiles, every supported N, bias present / absent."""
    B, H, W, K, R, st, pad = case
    x = rnd((B, 3, H, W), 71, 0, 1)
    w, b = (rnd((K, 3, R, R), 72) / np.sqrt(3 * R * R)).astype(np.float32), rnd((K,), 73, -0.1, 0.1)
    beta, gamma = rnd((K,), 74, 0.5, 1.5), (rnd((K, K), 75, 0.0, 0.1) + 0.1 * np.eye(K, dtype=np.float32)).astype(np.float32)
    gamma[0, :4] = [0.0, 1e-7, -0.5, 3.0e-6]                     # below the reparametrisation bound 2^-18
    assert F.c4gdn_supported(K, R, R)
    x4 = F.nchw3_to_nhwc4(dev(x))
    ast = F.c4gdn_stream(F.pack_weight(dev(w), F.PACK_CONV_FWD_C4), dev(gamma), K, R, R)
    assert torch.equal(ast, F.c4gdn_stream(F.pack_weight(dev(w), F.PACK_CONV_FWD_C4), dev(gamma), K, R, R))
    ref = orc.gdn_fwd(orc.conv2d_fwd(x, w, b, st, pad), beta, gamma)
    y = F.conv2d_c4_gdn_f16x3(x4, ast, dev(b), dev(beta), K, R, R, st, pad)
    assert_close(host(y), ref, what=f"c4gdn {case}", floor=0.1)
    yp = F.conv2d_c4_gdn_f16x3(x4, ast, dev(b), dev(beta), K, R, R, st, pad, planes_out=True)
    assert planes_match(yp, y), "planes output != fp32 output"
    y0 = F.conv2d_c4_gdn_f16x3(x4, ast, None, dev(beta), K, R, R, st, pad)
    assert_close(host(y0), orc.gdn_fwd(orc.conv2d_fwd(x, w, np.zeros(K, np.float32), st, pad), beta, gamma), what="no bias", floor=0.1)
    # into a channel slice of a wider NHWC buffer
    wide = torch.zeros(B, y.shape[2], y.shape[3], K + 64, device="cuda").permute(0, 3, 1, 2)
    F.conv2d_c4_gdn_f16x3(x4, ast, dev(b), dev(beta), K, R, R, st, pad, out=wide[:, 32:32 + K])
    assert torch.equal(wide[:, 32:32 + K], y) and float(wide[:, :32].abs().max()) == 0 and float(wide[:, 32 + K:].abs().max()) == 0


def test_analysis_transform_chain_vs_golden_and_fp32_kernels(F, golden, monkeypatch):
    """getY of the reference's I-frame model on the golden frames, forced through the fp16 chain (the golden batch is far below
    the size at which the chain is selected by itself): within 1e-4 of the reference's output, and next to the fp32-MFMA result."""
    import spatiotemporalentropymodel_amd.layers as L
    from spatiotemporalentropymodel_amd import selfcheck
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    g = golden("stem_small_forward.npz")               # BASELINE.json configs[0]: mbt2018(64,96) transforms, one septuplet
    d = torch.device("cuda:0")
    imodel, _ = selfcheck.build_models(64, 96, 64, 96, d, cls=SpatioTemporalPriorModel)
    frames = [f.to(d) for f in smooth_frames("septuplet0", 1, 7, 256)][:2]
    calls = []
    orig = F.conv2d_f16x3_fwd
    monkeypatch.setattr(F, "conv2d_f16x3_fwd", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    monkeypatch.setattr(L, "_F16X3_MIN_PIXELS", 0)
    monkeypatch.setenv("STEM_F16X3", "1")             # whatever the suite was started with
    with torch.no_grad():
        y0, _ = imodel.getY(frames[0])
        y1, _ = imodel.getY(frames[1])
        assert len(calls) == 6, calls                   # g_a.2, g_a.4, g_a.6 of both frames ran on the 192-wide fp16 kernel
        monkeypatch.setenv("STEM_F16X3", "0")
        y0_32, _ = imodel.getY(frames[0])
        assert len(calls) == 6
    assert_close(host(y0), g["y0"], what="g_a(frame 0), fp16 chain", floor=0.1)
    assert_close(host(y1), g["f1:y_cur"], what="g_a(frame 1), fp16 chain", floor=0.1)
    assert_close(host(y0), host(y0_32), what="fp16 chain vs fp32-MFMA kernels", floor=0.1)


def test_chain_is_selected_at_the_bench_size_and_not_under_autograd(F, monkeypatch):
    """B=16 x 256x256 (the bench workload): g_a.0 hands planes to g_a.2, g_a.2 to g_a.4, g_a.4 to the small last layer, which
    runs on the general split-K kernel; with autograd enabled (trainable transform) nothing is routed to the inference-only kernels."""
    from spatiotemporalentropymodel_amd.zoo import models
    torch.manual_seed(5)
    imodel = models["mbt2018"](quality=4).cuda().eval()
    x = torch.rand(16, 3, 256, 256, device="cuda")
    seen = []
    monkeypatch.setenv("STEM_F16X3", "1")             # whatever the suite was started with
    orig6, orig4 = F.conv2d_f16x3_fwd, F.conv2d_fwd_c4_gdn_planes
    monkeypatch.setattr(F, "conv2d_f16x3_fwd", lambda *a, **k: (seen.append(("fx3", a[0].shape, k.get("planes_out"))), orig6(*a, **k))[1])
    monkeypatch.setattr(F, "conv2d_fwd_c4_gdn_planes", lambda *a, **k: (seen.append(("c4",)), orig4(*a, **k))[1])
    origg = F.conv2d_f16x3_gen
    monkeypatch.setattr(F, "conv2d_f16x3_gen", lambda *a, **k: (seen.append(("gen", a[0].shape)), origg(*a, **k))[1])
    with torch.no_grad():
        y = imodel.g_a(x)
    assert seen == [("c4",), ("fx3", (16, 192, 128, 128), True), ("fx3", (16, 192, 64, 64), True), ("gen", (16, 192, 32, 32))], seen
    monkeypatch.setenv("STEM_F16X3", "0")
    with torch.no_grad():
        y32 = imodel.g_a(x)
    assert_close(host(y), host(y32), what="g_a at B=16, fp16 chain vs fp32-MFMA kernels", floor=0.1)
    assert float((y - y32).abs().max()) <= 1e-5 * float(y32.abs().max())          # measured: 2.7e-6 of the largest latent
    monkeypatch.setenv("STEM_F16X3", "1")
    n = len(seen)
    y_grad = imodel.g_a(x)
    assert len(seen) == n and y_grad.requires_grad


# ---------------------------------------------------------------------------------------------------------------------------
# General variant (training-time STEM layers): N tiles, split-K, activation epilogues, planes views
GEN_CASES = [  # B, C, H, W, K, R
    (2, 64, 9, 11, 96, 3),
    (1, 96, 13, 7, 160, 5),
    (4, 192, 16, 16, 256, 5),        # TPM.0 geometry at a quarter of the batch
    (2, 576, 8, 8, 384, 1),          # EPM.4-like 1x1
]


@pytest.mark.parametrize("case", GEN_CASES)
@pytest.mark.parametrize("split", [0, 1, 3])
@pytest.mark.parametrize("tile", [0, 64, 128])
@pytest.mark.parametrize("mfma", [32, 16])
def test_gen_forward_and_input_gradient_vs_oracle(F, case, split, tile, mfma):
    """forward + leaky ReLU and input-gradient x leaky-ReLU derivative (what autograd derives for conv(lrelu(u))) against the
    oracle, with the planner's split-K factor (0), unsplit (1) and a forced 3-way split, with the library's pixel tile (0),
    64-pixel (4 wavefronts) and 128-pixel (8 wavefronts) workgroups, and with both MFMA shapes (32x32x16; 16x16x32 with its
    permuted accumulator layout through the split-K slabs and the epilogue tile); planes output = fp32 output."""
    with F.tuning(fx3_split=split, fx3_gen_tile=tile, fx3_gen_mfma=mfma):
        _gen_forward_and_input_gradient(F, case, split)


@pytest.mark.parametrize("kx,kw", [(-70, 0), (0, -40), (45, 20), (-30, 50)])
def test_power_of_two_scaling_of_the_operands_is_exact(F, kx, kw):
    """What the scale records buy: conv(2^kx x, 2^kw w) == 2^(kx+kw) conv(x, w) BIT FOR BIT -- forward (both kernels, fp32 and
    planes output), input gradient and weight gradient -- for operands 21 decades below and 13 above the range fp16 itself
    covers; and an all-zero operand gives exact zeros (its record says max = 0, scale 1)."""
    B, C, H, W, K, R = 2, 64, 12, 20, 96, 3
    x, w, dy = rnd((B, C, H, W), 91, -2, 2), (rnd((K, C, R, R), 92) / np.sqrt(C * R * R)).astype(np.float32), rnd((B, K, H, W), 93)
    sx, sw = np.float32(2.0 ** kx), np.float32(2.0 ** kw)

    def run(xs, ws, dys):
        xp, dyp = F.F16Planes.split(dev(xs)), F.F16Planes.split(dev(dys))
        y, yp = F.conv2d_f16x3_gen(xp, F.pack_weight_f16x2_gen(dev(ws)), None, K, R, R, 1, 1, want_planes=True)
        y192 = F.conv2d_f16x3_fwd(xp, F.pack_weight_f16x2(dev(ws)), None, K, R, R, 1, 1)
        dx, _ = F.conv2d_f16x3_gen(dyp, F.pack_weight_f16x2_gen(dev(ws), flip=True), None, C, R, R, 1, 1)
        dw = torch.zeros(K, C, R, R, device="cuda")
        F.conv2d_wgrad_f16x3_into(xp, dyp, K, R, R, 1, dw, None, accumulate=False)
        return y, yp.merge(), y192, dx, dw

    base = run(x, w, dy)
    big = run(x * sx, w * sw, dy)
    for name, a, b, f in zip(("general kernel", "its planes output", "192-column kernel", "input gradient", "weight gradient"), base, big,
                             (sx * sw, sx * sw, sx * sw, sw, sx)):
        assert torch.equal(a * float(f), b), f"{name}: scaling the operands by 2^{kx}, 2^{kw} changed more than the exponent"
    z = run(np.zeros_like(x), w, dy)
    assert float(z[0].abs().max()) == 0.0 and float(z[2].abs().max()) == 0.0 and float(z[4].abs().max()) == 0.0
    assert F.F16Planes.split(dev(np.zeros_like(x))).record() == (1.0, 0.0)


def test_weight_scales_from_the_optimiser_pass(F):
    """stem_adam_step_bmax leaves max |p| per 4096-parameter chunk; stem_f16x2_pack_conv_weights_multi with those maxima (no maximum
    launch) must scale every image by an upper bound of its tensor's maximum that is at most the maximum of the chunks it touches
    (a neighbour can share a chunk), update the parameters exactly like stem_adam_step, zero the masked taps in place, and the
    convolutions with these images must match the oracle."""
    import ctypes as C
    from spatiotemporalentropymodel_amd import _lib
    ch = F.adam_chunk()
    shapes = [(96, 64, 3, 3), (7,), (64, 64, 5, 5), (128, 96, 1, 1)]             # conv, a bias in between, masked conv, 1x1
    sizes = [int(np.prod(sh)) for sh in shapes]
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    n = int(sum(sizes))
    rng = np.random.default_rng(5)
    p0 = (rng.standard_normal(n) * np.repeat([0.05, 30.0, 0.5, 2e-4], sizes)).astype(np.float32)     # very different magnitudes side by side
    g0 = rng.standard_normal(n).astype(np.float32)
    pa, pb = dev(p0), dev(p0)
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    ga, gb = dev(g0), dev(g0)
    bmax = torch.empty(4 * ((n + ch - 1) // ch), device="cuda")          # one value per wavefront of a chunk's workgroup
    F.adam_step(pa, ga, ma, va, None, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, zero_grad=True)
    F.adam_step_bmax(pb, gb, mb, vb, None, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, bmax, zero_grad=True)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and float(gb.abs().max()) == 0.0
    pn = host(pb)
    cmax = np.array([np.abs(pn[i:i + ch]).max() for i in range(0, n, ch)], np.float32)
    assert np.array_equal(host(bmax).reshape(-1, 4).max(1), cmax)
    descs, imgs = [], []
    for (sh, off, taps) in ((shapes[0], offs[0], 0), (shapes[2], offs[2], 12), (shapes[3], offs[3], 0)):
        K, Cc, R, S = sh
        img = torch.empty(F.f16x2_gen_weight_bytes(K, Cc, R, S), device="cuda", dtype=torch.uint8)
        b0 = int(off) // ch
        descs.append(_lib.F16PackDesc(pb.data_ptr() + 4 * int(off), img.data_ptr(), K, Cc, R, S, 0, taps, bmax.data_ptr(), b0,
                                      (int(off) + K * Cc * R * S - 1) // ch - b0 + 1, 0, 0))
        imgs.append(img)
    F.pack_weights_f16x2_multi((_lib.F16PackDesc * 3)(*descs))
    pz = host(pb)
    mask = np.ones(25, np.float32)
    mask[12:] = 0
    w2 = pn[offs[2]:offs[2] + sizes[2]].reshape(64, 64, 25) * mask
    assert np.array_equal(pz[offs[2]:offs[2] + sizes[2]].reshape(64, 64, 25), w2)          # masked taps zeroed in place, nothing else touched
    assert np.array_equal(np.delete(pz, np.s_[offs[2]:offs[2] + sizes[2]]), np.delete(pn, np.s_[offs[2]:offs[2] + sizes[2]]))
    for (sh, off, taps), img in zip(((shapes[0], offs[0], 0), (shapes[2], offs[2], 12), (shapes[3], offs[3], 0)), imgs):
        K, Cc, R, S = sh
        w = pz[off:off + K * Cc * R * S].reshape(sh)
        nrec = (img.numel() - 320) // 4
        rec = img.view(torch.float32)[nrec:nrec + 17].cpu()
        inv, bound = float(rec[1]), float(rec[16])
        lo, hi = int(off) // ch, (int(off) + w.size - 1) // ch
        assert np.abs(pn[off:off + w.size]).max() <= bound == float(cmax[lo:hi + 1].max()) and 2.0 ** 14 <= bound / inv < 2.0 ** 15
        x = rnd((2, Cc, 9, 11), 7 + K, -1, 1)
        y, _ = F.conv2d_f16x3_gen(F.F16Planes.split(dev(x)), img, None, K, R, S, 1, R // 2, taps=taps)
        assert_close(host(y), orc.conv2d_fwd(x, w, np.zeros(K, np.float32), 1, R // 2), what=f"conv with the optimiser-scaled image {sh}", floor=0.1)


@pytest.mark.parametrize("mask_type", ["A", "B"])
@pytest.mark.parametrize("R", [3, 5])
def test_masked_convolution_on_the_general_kernel(F, mask_type, R):
    """MaskedConv2d (layers.py:21-47) on the general f16x3 kernel: the weight image holds only the live taps (a prefix of the
    row-major order: 12 of 25 for the 5x5 type-A mask), the masked taps of the torch weight are zeroed IN PLACE by the pack (what
    the reference does at every forward), result against the oracle's convolution with the masked weight."""
    B, C, H, W, K = 2, 64, 11, 13, 96
    x, w, b = rnd((B, C, H, W), 81, -2, 2), (rnd((K, C, R, R), 82) / np.sqrt(C * R * R)).astype(np.float32), rnd((K,), 83, -0.1, 0.1)
    mask = np.ones((R, R), np.float32)
    mask[R // 2, R // 2 + (mask_type == "B"):] = 0
    mask[R // 2 + 1:] = 0
    taps = F.masked_live_taps(R, R, mask_type)
    assert taps == int(mask.sum()) and mask.reshape(-1)[:taps].all()
    wd = dev(w)
    wp = F.pack_weight_f16x2_gen(wd, taps=taps)
    assert np.array_equal(host(wd), w * mask)                        # zeroed in place, live taps untouched
    for tile, mfma in ((64, 32), (128, 32), (64, 16), (128, 16)):
        with F.tuning(fx3_gen_tile=tile, fx3_gen_mfma=mfma):
            y, _ = F.conv2d_f16x3_gen(F.F16Planes.split(dev(x)), wp, dev(b), K, R, R, 1, R // 2, taps=taps)
        assert_close(host(y), orc.conv2d_fwd(x, w * mask, b, 1, R // 2), what=f"masked {mask_type} {R}x{R} tile {tile} mfma {mfma}", floor=0.1)


def _gen_forward_and_input_gradient(F, case, split):
    B, C, H, W, K, R = case
    pad, sl = R // 2, 0.01
    x, w, b = rnd((B, C, H, W), 31, -2, 2), (rnd((K, C, R, R), 32) / np.sqrt(C * R * R)).astype(np.float32), rnd((K,), 33, -0.1, 0.1)
    dy = rnd((B, K, H, W), 34)
    ref = orc.conv2d_fwd(x, w, b, 1, pad)
    ref = np.where(ref > 0, ref, ref * sl).astype(np.float32)
    dx_ref = orc.conv2d_bwd(x, w, dy, 1, pad)[0]
    dx_ref = np.where(x > 0, dx_ref, dx_ref * sl).astype(np.float32)
    xd = dev(x).contiguous(memory_format=torch.channels_last)
    y, yp = F.conv2d_f16x3_gen(F.F16Planes.split(xd), F.pack_weight_f16x2_gen(dev(w)), dev(b), K, R, R, 1, pad,
                                epi=F.GEN_EPI_LRELU, slope=sl, want_planes=True)
    assert_close(host(y), ref, what=f"gen fwd {case} split={split}", floor=0.1)
    assert planes_match(yp, y)
    d, dp = F.conv2d_f16x3_gen(F.F16Planes.split(dev(dy)), F.pack_weight_f16x2_gen(dev(w), flip=True), None, C, R, R, 1, pad,
                                epi=F.GEN_EPI_DACT, slope=sl, z=xd, want_planes=True)
    assert_close(host(d), dx_ref, what=f"gen dgrad {case} split={split}", floor=0.1)
    assert planes_match(dp, d)


def test_gen_channel_views_strided_output_and_multi_pack(F):
    """A 32-aligned channel view of a wider planes tensor as input, the result written into a channel slice of a wider fp32
    buffer (the engine's epm_in), and the one-launch weight packing equal to the single-layer packing."""
    from spatiotemporalentropymodel_amd import _lib
    B, H, W = 2, 8, 8
    big = rnd((B, 160, H, W), 41, -2, 2)
    w = (rnd((96, 64, 3, 3), 42) / np.sqrt(64 * 9)).astype(np.float32)
    b = rnd((96,), 43, -0.1, 0.1)
    bigp = F.F16Planes.split(dev(big))
    wide = torch.zeros(B, H, W, 224, device="cuda").permute(0, 3, 1, 2)           # NHWC buffer, 224 channels
    out = wide[:, 64:160]
    wp = F.pack_weight_f16x2_gen(dev(w))
    y, _ = F.conv2d_f16x3_gen(bigp.channels(32, 96), wp, dev(b), 96, 3, 3, 1, 1, out=out)
    assert y.data_ptr() == out.data_ptr()
    assert_close(host(out), orc.conv2d_fwd(big[:, 32:96], w, b, 1, 1), what="view in, slice out", floor=0.1)
    assert float(wide[:, :64].abs().max()) == 0 and float(wide[:, 160:].abs().max()) == 0
    with pytest.raises(ValueError):
        bigp.channels(16, 80)
    # multi pack == single packs (forward and mirrored)
    w2 = (rnd((64, 96, 5, 5), 44) / 50).astype(np.float32)
    wd, w2d = dev(w), dev(w2)
    singles = [F.pack_weight_f16x2_gen(wd), F.pack_weight_f16x2_gen(wd, flip=True), F.pack_weight_f16x2_gen(w2d)]
    outs = [torch.empty_like(s) for s in singles]
    descs = (_lib.F16PackDesc * 3)(_lib.F16PackDesc(wd.data_ptr(), outs[0].data_ptr(), 96, 64, 3, 3, 0, 0),
                                    _lib.F16PackDesc(wd.data_ptr(), outs[1].data_ptr(), 64, 96, 3, 3, 1, 0),
                                    _lib.F16PackDesc(w2d.data_ptr(), outs[2].data_ptr(), 64, 96, 5, 5, 0, 0))
    F.pack_weights_f16x2_multi(descs)
    for a, s in zip(outs, singles):
        assert torch.equal(a, s)


def test_engine_schedule_with_and_without_bf16_layers(monkeypatch):
    """One training forward / backward of the full-size STEM model (B=2, 16x16 latents) through the explicit schedule with the
    stride-1 layers on the fp16 kernels and with every layer on the fp32-MFMA kernels, BOTH measured against the CPU oracle
    (double accumulation) on the same weights, inputs and noise (the two routes used to be compared only
    with each other, at 2e-4 of the tensor maximum).

    Metric per gradient tensor: max |err| / max(|ref|, rms(ref)) over ALL elements.  Measured on MI355X
    (tools/debug/route_vs_oracle.py, these inputs): fp32-MFMA route 5.8e-4 / fp16 route 4.2e-4 on the worst tensor (HE.4 / HD.2
    weights: hyper-path gradients are sums of a few dlik / lik terms over 4x4 latents, where the fp32 rounding of z decides
    ~1e-4 of the result on either side -- the reference's own fp32 run is that far from its float64 run, test_hip_models.py),
    1e-4 .. 3e-4 elsewhere.  So: (a) the fp16 route is held to the fp32-MFMA route's distance from the oracle, tensor by
    tensor (it is consistently the closer one); (b) both are held to an absolute 1e-3 in that strict metric; (c) the discrete
    decisions (leaky-ReLU sides, likelihood bound) of both runs and the oracle are compared, and with none flipped the two
    routes must also agree to 2e-4 of every tensor's maximum."""
    import oracle_pass as OP
    from spatiotemporalentropymodel_amd import engine as E
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    d = torch.device("cuda:0")
    # A flipped decision moves a pixel's back-propagated gradient outright and says nothing about rounding: the per-tensor gates
    # need a draw of the inputs on which neither route flips one against the oracle.  Up to four draws; the first clean one is
    # gated in full -- none of the four clean is a failure (a flip used to silence every gate of the run).
    seen = []
    for attempt in range(4):
        sfx = "" if attempt == 0 else str(attempt)
        y_cur = closed_form_input("eng:y" + sfx, (2, 192, 16, 16), -6, 6).to(d)
        y_cond = closed_form_input("eng:c" + sfx, (2, 192, 16, 16), -6, 6).to(d)
        runs = {}
        for tag, on in (("fp16", True), ("fp32", False)):
            monkeypatch.setattr(E.StemEngine, "use_fx3", on)
            m = closed_form_fill_(SpatioTemporalPriorModel_Res()).to(d).train()
            assert any(l.fx3 for l in m.engine().layers) == on
            runs[tag] = OP.hip_train_pass(m, y_cur, y_cond, "eng" + sfx)
        ref, rgrads, racts = OP.oracle_train_pass(m, y_cur, y_cond, runs["fp32"][4], residual=True)
        assert abs(runs["fp16"][1] - runs["fp32"][1]) <= 1e-5 * abs(runs["fp32"][1])
        assert runs["fp16"][2].keys() == runs["fp32"][2].keys() and len(rgrads) > 30
        dist = {tag: {n: OP.grad_distance(runs[tag][2][n], g) for n, g in rgrads.items() if n in runs[tag][2]} for tag in runs}
        flips = {tag: OP.decisions_flipped(runs[tag][3], racts, OP.host(runs[tag][0]["likelihoods"]["y"]), ref["lik_y"]) for tag in runs}
        flips["fp16 vs fp32"] = OP.decisions_flipped(runs["fp16"][3], runs["fp32"][3])
        for tag in runs:
            worst = sorted(((v, n) for n, v in dist[tag].items()), reverse=True)[:4]
            print(f"draw {attempt}, {tag} route vs oracle: decisions flipped {flips[tag] or 'none'}; worst gradients " + ", ".join(f"{n} {v:.1e}" for v, n in worst))
        assert sum(sum(f.values()) for f in flips.values()) <= 4, flips          # a handful of near-zero pre-activations, not a broken layer
        seen.append(flips)
        if any(flips.values()):
            continue
        ngated = 0
        for n in rgrads:
            if n not in dist["fp16"]:
                continue
            db, df = dist["fp16"][n], dist["fp32"][n]
            assert db <= 1.5 * df + 1e-4, f"{n}: fp16 route {db:.2e} from the oracle, fp32-MFMA route {df:.2e}"
            assert max(db, df) <= 1e-3, (n, db, df)
            assert_close(runs["fp16"][2][n], runs["fp32"][2][n], rtol=2e-4, what=f"grad {n}, route vs route", floor=1.0)
            ngated += 1
        assert ngated > 30
        return
    pytest.fail(f"no draw of the inputs without a flipped decision in four attempts: {seen}")


# ---------------------------------------------------------------------------------------------------------------------------
# Weight gradient (csrc/wgrad_f16x3.hip): contraction over pixels through the transposing LDS reads
WG_CASES = [  # B, C, H, W, K, R
    (2, 64, 9, 11, 96, 3),           # ragged pixel count (198 = 12 chunks of 16 + 6), one tile
    (1, 160, 13, 7, 96, 5),          # C = 160: second channel tile half empty
    (3, 32, 16, 16, 288, 1),         # 1x1, three k tiles
    (4, 192, 16, 16, 256, 5),        # TPM.0 geometry at a quarter of the batch (filter-row form: rows of 16 pixels)
    (2, 96, 8, 32, 160, 5),          # filter-row form, two chunks per image row; K = 160 / C = 96: half-empty k and c tiles
    (1, 96, 3, 48, 64, 5),           # ... three chunks per row, fewer image rows than filter rows
    (3, 64, 5, 16, 96, 3),           # ... 3 x 3 window
]


@pytest.mark.parametrize("case", WG_CASES)
@pytest.mark.parametrize("split", [0, 1, 3])
def test_weight_gradient_vs_oracle(F, case, split):
    with F.tuning(wg3_split=split):
        _weight_gradient_vs_oracle(F, case, split)


def _weight_gradient_vs_oracle(F, case, split):
    B, C, H, W, K, R = case
    pad = R // 2
    x, dy = rnd((B, C, H, W), 51, -2, 2), rnd((B, K, H, W), 52)
    w0 = np.zeros((K, C, R, R), np.float32)
    _, dw_ref, db_ref = orc.conv2d_bwd(x, w0, dy, 1, pad, need_dx=False)
    xd, dyd = dev(x), dev(dy)
    splits, elems = F.wgrad_f16x3_plan(xd.shape, K, R, R, pad)
    assert split == 0 or splits <= split             # clamped to >= 16 pixel chunks per split
    dwp = torch.empty(elems, device="cuda")
    dbf = torch.full((K,), float("nan"), device="cuda")
    F.conv2d_wgrad_f16x3(F.F16Planes.split(xd), F.F16Planes.split(dyd), K, R, R, pad, dwp, splits, db=dbf)
    dw = dwp.view(splits, R * R, K, C).sum(0).permute(1, 2, 0).reshape(K, C, R, R)
    assert_close(host(dw), dw_ref, what=f"wgrad {case} split={split}", floor=0.1)
    assert_close(host(dbf), db_ref, what="bias gradient from the weight-gradient pass", floor=0.1)
    dbacc = dbf.clone()
    F.conv2d_wgrad_f16x3(F.F16Planes.split(xd), F.F16Planes.split(dyd), K, R, R, pad, dwp, splits, db=dbacc, accumulate_db=True)
    assert_close(host(dbacc), 2 * db_ref, what="bias gradient from the weight-gradient pass, accumulated", floor=0.1)
    db = torch.zeros(K, device="cuda")
    F.bias_grad(dyd.contiguous(memory_format=torch.channels_last), db)
    assert_close(host(db), db_ref, what="bias gradient", floor=0.1)
    F.bias_grad(dyd.contiguous(memory_format=torch.channels_last), db, accumulate=True)
    assert_close(host(db), 2 * db_ref, what="bias gradient, accumulated", floor=0.1)


@pytest.mark.parametrize("case", [c for c in WG_CASES if c[3] % 16 == 0])
def test_weight_gradient_filter_row_form_vs_per_tap_form(F, case):
    """the two forms of csrc/wgrad_f16x3.hip on the geometries the filter-row form takes: each against the oracle (above, the
    row form by default) and against each other -- same products, sums over pixels in a different order"""
    B, C, H, W, K, R = case
    pad = R // 2
    xp, dyp = F.F16Planes.split(dev(rnd((B, C, H, W), 53, -2, 2))), F.F16Planes.split(dev(rnd((B, K, H, W), 54)))
    out = {}
    for form in (0, 1):
        with F.tuning(wg3_row=form):
            splits, elems = F.wgrad_f16x3_plan((B, C, H, W), K, R, R, pad)
            dwp, db = torch.empty(elems, device="cuda"), torch.zeros(K, device="cuda")
            F.conv2d_wgrad_f16x3(xp, dyp, K, R, R, pad, dwp, splits, db=db)
            out[form] = (host(dwp.view(splits, R * R, K, C).sum(0)), host(db))
    assert_close(out[0][0], out[1][0], what=f"filter-row form against per-tap form {case}", floor=0.1)
    assert_close(out[0][1], out[1][1], what="bias gradient of the two forms", floor=0.1)
    with F.tuning(wg3_row=1):
        per_tap_splits = F.wgrad_f16x3_plan((B, C, H, W), K, R, R, pad)[0]
    with F.tuning(wg3_row=0, wg3_split=2):      # a forced split is honoured by the row form's planner too
        assert F.wgrad_f16x3_plan((B, C, H, W), K, R, R, pad)[0] <= 2 and per_tap_splits >= 1


def test_weight_gradient_from_channel_views(F):
    """x and dy as 32-aligned channel views of wider planes tensors (how the engine feeds the prior branches' slices)"""
    B, H, W = 2, 8, 8
    xb, dyb = rnd((B, 160, H, W), 61, -2, 2), rnd((B, 224, H, W), 62)
    xp, dyp = F.F16Planes.split(dev(xb)).channels(32, 96), F.F16Planes.split(dev(dyb)).channels(64, 160)
    splits, elems = F.wgrad_f16x3_plan(xp.shape, 96, 3, 3, 1)
    dwp = torch.empty(elems, device="cuda")
    F.conv2d_wgrad_f16x3(xp, dyp, 96, 3, 3, 1, dwp, splits)
    dw = dwp.view(splits, 9, 96, 64).sum(0).permute(1, 2, 0).reshape(96, 64, 3, 3)
    ref = orc.conv2d_bwd(np.ascontiguousarray(xb[:, 32:96]), np.zeros((96, 64, 3, 3), np.float32), np.ascontiguousarray(dyb[:, 64:160]), 1, 1,
                         need_dx=False)[1]
    assert_close(host(dw), ref, what="wgrad from views", floor=0.1)
    # the same through the filter-row form (rows of 16 pixels): the views' pixel pitches are those of the wider tensors
    B, H, W = 2, 4, 16
    xb, dyb = rnd((B, 160, H, W), 63, -2, 2), rnd((B, 224, H, W), 64)
    xp, dyp = F.F16Planes.split(dev(xb)).channels(32, 96), F.F16Planes.split(dev(dyb)).channels(64, 160)
    splits, elems = F.wgrad_f16x3_plan(xp.shape, 96, 5, 5, 2)
    dwp = torch.empty(elems, device="cuda")
    F.conv2d_wgrad_f16x3(xp, dyp, 96, 5, 5, 2, dwp, splits)
    dw = dwp.view(splits, 25, 96, 64).sum(0).permute(1, 2, 0).reshape(96, 64, 5, 5)
    ref = orc.conv2d_bwd(np.ascontiguousarray(xb[:, 32:96]), np.zeros((96, 64, 5, 5), np.float32), np.ascontiguousarray(dyb[:, 64:160]), 1, 2,
                         need_dx=False)[1]
    assert_close(host(dw), ref, what="wgrad from views, filter-row form", floor=0.1)


def test_gen_kernel_at_eight_full_hd_sequences(F):
    """The entropy-parameter network's first layer shape when 8 full-HD sequences are coded side by side (65280 latent pixels,
    768 outputs): the split-K slabs must be sized by the split actually used (an earlier bound assumed 16 and refused the call);
    checked against the fp32-MFMA kernel."""
    B, C, H, W, K = 8, 512, 68, 120, 768
    torch.manual_seed(1)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 1, 1, device="cuda") / C ** 0.5
    b = torch.randn(K, device="cuda") * 0.1
    y, _ = F.conv2d_f16x3_gen(F.F16Planes.split(x), F.pack_weight_f16x2_gen(w), b, K, 1, 1, 1, 0, epi=F.GEN_EPI_LRELU, slope=0.01)
    y32 = F.conv2d_fwd(x, F.pack_weight(w, F.PACK_CONV_FWD), b, K, 1, 1, 1, 0, F.ACT_LRELU, slope=0.01)
    assert_close(host(y), host(y32), what="gen kernel, 65280 pixels x 768 channels", floor=0.1)


def test_random_shapes_through_all_three_kernels():
    """tools/debug/f16x3_fuzz.py: 40 random (batch, size, channels, kernel, stride) combinations through the analysis-transform
    kernel (+ GDN), the general kernel (forward, input gradient) and the weight-gradient kernel (+ bias gradient) against float64
    torch references; every planes output must equal its fp32 twin.  Called in-process (a GPU-initialised pytest
    process must not start GPU children on this pool)."""
    sys.path.insert(0, os.path.join(REPO, "tools", "debug"))
    import f16x3_fuzz
    worst = f16x3_fuzz.run(40, 7, verbose=False)
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("plan", [dict(fx3_tile=128, fx3_depth=3, fx3_mfma=16, fx3_gen_tile=128, fx3_gen_mfma=16), dict(fx3_tile=64, fx3_depth=3, fx3_mfma=16, fx3_gen_tile=64, fx3_gen_mfma=16),
                                  dict(fx3_tile=128, fx3_depth=3, fx3_mfma=32, fx3_gen_tile=64)],
                         ids=["tile128-3stage-mfma16", "tile64-3stage-mfma16", "tile128-3stage-mfma32"])
def test_random_shapes_under_forced_plans(F, plan):
    """the same fuzz with the plan selectors forced to the forms the library only picks for large launches (three LDS stages,
    the 16x16x32 MFMA shape and its permuted accumulator layout, 128-pixel tiles): ragged tiles, partial channel blocks, every
    epilogue"""
    sys.path.insert(0, os.path.join(REPO, "tools", "debug"))
    import f16x3_fuzz
    with F.tuning(**plan):
        worst = f16x3_fuzz.run(24, 11, verbose=False)
    assert worst <= 1e-5, worst


# ---- the transposed faces and the weight gradients of the stride-2 layers on the fp16 matrix cores (round 5) --------------------
TCONV_CASES = [  # B, C (coarse channels), H, W (coarse grid), N (outputs), R
    (2, 64, 4, 4, 96, 5),
    (16, 256, 4, 4, 256, 5),          # HD.0 / the input gradient of HE.4 at the bench geometry
    (16, 256, 8, 8, 256, 5),          # HD.2 / the input gradient of HE.2
    (3, 32, 5, 7, 100, 5),            # ragged coarse grid, N not a multiple of 32 (fp32 output only)
    (2, 96, 6, 9, 160, 3),            # 3 x 3: phases of 1 / 2 / 2 / 4 taps
]


@pytest.mark.parametrize("case", TCONV_CASES)
@pytest.mark.parametrize("split", [0, 1, 3])
def test_transposed_face_forward_vs_oracle(F, case, split):
    """nn.ConvTranspose2d(C, N, R, stride=2, padding=R//2, output_padding=1) forward (HD.0 / HD.2: spatiotemporalpriors.py:822-826)
    as ONE launch of the general fp16 kernel over the four sub-pixel phases (phase images from the multi-pack, flip = 2): fp32
    rows, leaky ReLU, planes of the fine tensor with one scale record -- against the oracle, with the planner's split, unsplit
    and a forced split."""
    B, C, H, W, N, R = case
    x = rnd((B, C, H, W), 301, -2, 2)
    w = (rnd((C, N, R, R), 302) / np.sqrt(C * R * R / 4)).astype(np.float32)
    b = rnd((N,), 303, -0.1, 0.1)
    ref = orc.deconv2d_fwd(x, w, b, 2, R // 2, 1)
    ref_act = np.where(ref > 0, ref, ref * np.float32(0.01)).astype(np.float32)
    xp = F.F16Planes.split(dev(x))
    wp = F.pack_weight_f16x2_tconv(dev(w))
    with F.tuning(fx3_split=split):
        y, _ = F.tconv2d_f16x3(xp, wp, dev(b), N, R)
        ya, yap = F.tconv2d_f16x3(xp, wp, dev(b), N, R, epi=F.GEN_EPI_LRELU, want_planes=N % 32 == 0)
    assert tuple(y.shape) == (B, N, 2 * H, 2 * W)
    assert_close(host(y), ref, what=f"tconv {case} split={split}", floor=0.1)
    assert_close(host(ya), ref_act, what=f"tconv + lrelu {case} split={split}", floor=0.1)
    if yap is not None:
        assert planes_match(yap, ya), "planes output != fp32 output"
    # into a channel slice of a wider buffer, no bias
    wide = torch.zeros(B, 2 * H, 2 * W, N + 32, device="cuda").permute(0, 3, 1, 2)
    with F.tuning(fx3_split=split):
        F.tconv2d_f16x3(xp, wp, None, N, R, out=wide[:, 16:16 + N])
    assert_close(host(wide[:, 16:16 + N]), orc.deconv2d_fwd(x, w, np.zeros(N, np.float32), 2, R // 2, 1), what="tconv slice", floor=0.1)
    assert float(wide[:, :16].abs().max()) == 0 and float(wide[:, 16 + N:].abs().max()) == 0


@pytest.mark.parametrize("case", TCONV_CASES[:3] + TCONV_CASES[4:])
def test_transposed_face_as_input_gradient_vs_oracle(F, case):
    """The input gradient of nn.Conv2d(N, C, R, stride=2, padding=R//2) on an even-sized input (HE.2 / HE.4, :814-818; torch
    autograd) is the same transposed face with the Conv2d weight [K = C_coarse][C_in = N] read as w[c][n][r][s]; the leaky-ReLU
    derivative of the layer's (activated) input is applied in the epilogue (DACT), planes of the result ride along."""
    B, C, H, W, N, R = case                                   # dy: [B, C, H, W] coarse; dx: [B, N, 2H, 2W]
    dy = rnd((B, C, H, W), 311, -1, 1)
    w = (rnd((C, N, R, R), 312) / np.sqrt(C * R * R / 4)).astype(np.float32)        # Conv2d weight [K][Cin][R][S] with K = C, Cin = N
    xact = rnd((B, N, 2 * H, 2 * W), 313, -1, 1)
    dx_ref, _, _ = orc.conv2d_bwd(np.zeros((B, N, 2 * H, 2 * W), np.float32), w, dy, 2, R // 2)
    ref = np.where(xact > 0, dx_ref, dx_ref * np.float32(0.01)).astype(np.float32)
    dyp = F.F16Planes.split(dev(dy))
    wp = F.pack_weight_f16x2_tconv(dev(w))
    z = F.to_nhwc(dev(xact))
    d, dp = F.tconv2d_f16x3(dyp, wp, None, N, R, epi=F.GEN_EPI_DACT, z=z, want_planes=N % 32 == 0)
    assert_close(host(d), ref, what=f"tconv dgrad {case}", floor=0.1)
    if dp is not None:
        assert planes_match(dp, d)


WGS_CASES = [  # B, C (fine-grid channels), H, W (fine grid), K (coarse-grid channels), R
    (2, 64, 8, 8, 96, 5),
    (16, 256, 8, 8, 256, 5),          # HE.4 (8x8 -> 4x4) / HD.0 with the roles swapped
    (16, 256, 16, 16, 256, 5),        # HE.2 / HD.2
    (3, 32, 10, 14, 64, 3),
]


@pytest.mark.parametrize("case", WGS_CASES)
def test_strided_weight_gradient_vs_oracle(F, case):
    """Weight (and bias) gradient of nn.Conv2d(C, K, R, stride=2, padding=R//2) from planes operands on the per-tap fp16 kernel
    (HE.2 / HE.4), and of nn.ConvTranspose2d(K, C, R, stride=2, padding=R//2, output_padding=1) with the operands' roles swapped
    (HD.0 / HD.2: its input on the coarse grid plays dy, the gradient of its output on the fine grid plays x), against the oracle."""
    B, C, H, W, K, R = case
    pad = R // 2
    x = rnd((B, C, H, W), 321, -1, 1)
    dy = rnd((B, K, H // 2, W // 2), 322, -1, 1)
    _, dw_ref, db_ref = orc.conv2d_bwd(x, np.zeros((K, C, R, R), np.float32), dy, 2, pad, need_dx=False)
    xp, dyp = F.F16Planes.split(dev(x)), F.F16Planes.split(dev(dy))
    splits, elems = F.wgrad_f16x3_strided_plan(xp.shape, K, R, R, 2, pad)
    dwp = torch.empty(elems, device="cuda")
    bpart = torch.empty(splits * K, device="cuda")
    F.conv2d_wgrad_f16x3_strided(xp, dyp, K, R, R, 2, pad, dwp, splits, bias_part=bpart)
    dw = dwp.view(splits, R * R, K, C).sum(0).permute(1, 2, 0).reshape(K, C, R, R)
    assert_close(host(dw), dw_ref, what=f"strided wgrad {case}", floor=0.1)
    assert_close(host(bpart.view(splits, K).sum(0)), db_ref, what="bias column sums", floor=0.1)
    # ConvTranspose2d(K -> C): weight [K][C][R][S]; x_t = its input [B, K, H/2, W/2] (coarse), dy_t = gradient of its output (fine)
    xt, dyt = dy, x
    _, dwt_ref, _ = orc.deconv2d_bwd(xt, np.zeros((K, C, R, R), np.float32), dyt, 2, pad, 1, need_dx=False)
    dwp.zero_()
    F.conv2d_wgrad_f16x3_strided(xp, dyp, K, R, R, 2, pad, dwp, splits)          # fp = planes(dy_t) = xp, gp = planes(x_t) = dyp
    dwt = dwp.view(splits, R * R, K, C).sum(0).permute(1, 2, 0).reshape(K, C, R, R)
    assert_close(host(dwt), dwt_ref, what=f"transposed layer's wgrad {case}", floor=0.1)


@pytest.mark.parametrize("odd", [(True, True), (True, False), (False, True)])
def test_transposed_face_odd_fine_grid_vs_oracle(F, odd):
    """a strided convolution of an odd-sized input (2H - 1 rows -> H rows: 1080p latents are 68 -> 34 -> 17) has an input gradient
    on a fine grid of 2H - 1 rows: the phases' pixels beyond it are not written"""
    B, C, H, W, N, R = 2, 64, 5, 6, 96, 5
    Hf, Wf = 2 * H - int(odd[0]), 2 * W - int(odd[1])
    dy = rnd((B, C, H, W), 331, -1, 1)
    w = (rnd((C, N, R, R), 332) / np.sqrt(C * R * R / 4)).astype(np.float32)
    xact = rnd((B, N, Hf, Wf), 333, -1, 1)
    dx_ref, _, _ = orc.conv2d_bwd(np.zeros((B, N, Hf, Wf), np.float32), w, dy, 2, R // 2)
    ref = np.where(xact > 0, dx_ref, dx_ref * np.float32(0.01)).astype(np.float32)
    d, dp = F.tconv2d_f16x3(F.F16Planes.split(dev(dy)), F.pack_weight_f16x2_tconv(dev(w)), None, N, R, epi=F.GEN_EPI_DACT,
                            z=F.to_nhwc(dev(xact)), want_planes=True, fine_hw=(Hf, Wf))
    assert tuple(d.shape) == (B, N, Hf, Wf)
    assert_close(host(d), ref, what=f"tconv dgrad odd {odd}", floor=0.1)
    assert planes_match(dp, d)


def test_pair_pack_writes_the_same_images_as_the_per_role_pack(F):
    """stem_f16x2_pack_conv_weights_pair_multi (both images of a layer from one read of its weights: 32 x 32 x RS tiles through
    LDS) against stem_f16x2_pack_conv_weights_multi called per role, byte for byte, with the scales taken from the optimiser
    pass's chunk maxima in both: stride-1 convolution (forward + mirrored input-gradient image), the masked context convolution
    (12 live taps, the others zeroed in place), a strided convolution (strided face + four phase images), a transposed
    convolution (phase images + strided face), 1x1 layers, row counts that are not multiples of 32 / 128."""
    from spatiotemporalentropymodel_amd import _lib
    ch = F.adam_chunk()
    # (A, B, R, role-0 (flip, taps) or None, role-1 flip or None)
    layers = [(96, 64, 3, (0, 0), 1), (384, 192, 5, (0, 12), None), (128, 160, 5, (0, 0), 2), (64, 96, 5, (2, 0), 0), (576, 384, 1, (0, 0), 1),
              (100, 64, 1, (0, 0), None), (320, 256, 5, (0, 0), 1)]
    sizes = [A * B * R * R for A, B, R, _, _ in layers]
    offs = np.concatenate([[0], np.cumsum([s + 37 for s in sizes])])[:-1]          # odd gaps: tensors share chunks with their neighbours
    offs = (offs + 3) // 4 * 4
    n = int(offs[-1] + sizes[-1])
    rng = np.random.default_rng(9)
    p0 = rng.standard_normal(n).astype(np.float32) * 0.05
    bufs = []
    for _ in range(2):
        p = dev(p0)
        g, m, v = dev(rng.standard_normal(n).astype(np.float32)), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        rng = np.random.default_rng(10)                      # the same gradient for both copies
        bmax = torch.empty(4 * ((n + ch - 1) // ch), device="cuda")
        bufs.append((p, g, m, v, bmax))
    bufs[1][1].copy_(bufs[0][1])
    for p, g, m, v, bmax in bufs:
        F.adam_step_bmax(p, g, m, v, None, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, bmax, zero_grad=True)
    assert torch.equal(bufs[0][0], bufs[1][0])

    def chunks(off, numel):
        b0 = int(off) // ch
        return b0, (int(off) + numel - 1) // ch - b0 + 1

    single, pair, imgs_s, imgs_p = [], [], [], []
    for (A, B, R, r0, r1), off, numel in zip(layers, offs, sizes):
        b0, nb = chunks(off, numel)
        roles_p = []
        for which, spec in ((0, r0), (1, (r1, 0) if r1 is not None else None)):
            if spec is None:
                roles_p += [None, 0, 0]
                continue
            flip, taps = spec
            N, Cc = (B, A) if flip else (A, B)
            nbytes = F.f16x2_gen_weight_bytes(N, Cc, R, R)
            a, b = torch.zeros(nbytes, device="cuda", dtype=torch.uint8), torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
            imgs_s.append(a)
            imgs_p.append(b)
            single.append(_lib.F16PackDesc(bufs[0][0].data_ptr() + 4 * int(off), a.data_ptr(), N, Cc, R, R, flip, taps, bufs[0][4].data_ptr(), b0, nb, 0, 0))
            roles_p += [b.data_ptr(), int(flip != 0) | (flip << 1), taps]
        pair.append(_lib.F16PairDesc(bufs[1][0].data_ptr() + 4 * int(off), A, B, R, R, *roles_p, bufs[1][4].data_ptr(), b0, nb))
    F.pack_weights_f16x2_multi((_lib.F16PackDesc * len(single))(*single))
    F.pack_weights_f16x2_pair_multi((_lib.F16PairDesc * len(pair))(*pair))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(imgs_s, imgs_p)):
        assert torch.equal(a, b), f"image {i}: {int((a != b).sum())} of {a.numel()} bytes differ"
    assert torch.equal(bufs[0][0], bufs[1][0])                  # the masked taps were zeroed in place by both
    wctx = bufs[1][0][int(offs[1]):int(offs[1]) + sizes[1]].view(384, 192, 25)
    assert float(wctx[:, :, 12:].abs().max()) == 0.0 and float(wctx[:, :, :12].abs().max()) > 0.0
