"""GPU parity tests of the analysis transform on the bf16 matrix cores (csrc/conv_bf16x6.hip, through the C ABI): the frozen
g_a chain of the I-frame model (compressai/models/priors.py:613-621 under no_grad, stem/trainSTEM.py:128,171) with every fp32
operand pre-split into three bf16 numbers and six MFMAs per fp32 product.  Same bound as the fp32-MFMA kernels: 1e-4 relative
(north_star), against the CPU oracle and the golden vectors captured from the reference.
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


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().contiguous().numpy()


def test_split_is_exact_and_planes_are_bf16(F):
    """x = p0 + p1 + p2 bit for bit (the planes hold the fp32 value, not an approximation of it), p0 = x rounded to bf16."""
    x = rnd((3, 64, 5, 7), 1, -4, 4)
    x.reshape(-1)[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -1.1754944e-38, 65504.0, 2.0 ** -100]
    xp = F.Bf16Planes.split(dev(x))
    assert torch.equal(xp.merge().cpu(), torch.from_numpy(x).contiguous(memory_format=torch.channels_last))
    raw = xp.data.view(torch.bfloat16).view(3 * 5 * 7, 2, 3, 32).float().cpu()          # [pixel][slab][plane][32]
    nhwc = torch.from_numpy(x).permute(0, 2, 3, 1).reshape(3 * 5 * 7, 2, 32)
    assert torch.equal(raw[:, :, 0], nhwc.to(torch.bfloat16).float())                  # round-to-nearest-even leading plane
    assert float((raw[:, :, 1].abs() - nhwc.abs() * 2.0 ** -8).max()) <= 0 and float((raw[:, :, 2].abs() - nhwc.abs() * 2.0 ** -16).max()) <= 0
    with pytest.raises(ValueError):
        F.Bf16Planes.empty(1, 48, 4, 4, torch.device("cuda:0"))


CASES = [  # B, C, H, W, K, R, stride
    (1, 64, 16, 16, 64, 3, 1),
    (2, 192, 20, 28, 192, 5, 2),
    (1, 96, 33, 47, 160, 5, 2),
    (3, 32, 9, 11, 100, 1, 1),        # K not a multiple of 32 (fp32 output only), 1x1
    (2, 128, 24, 24, 192, 3, 2),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile128", [False, True])
@pytest.mark.parametrize("gdn", [False, True])
def test_conv_gdn_vs_oracle(F, monkeypatch, case, tile128, gdn):
    """conv (+ fused GDN) against the oracle, both workgroup tiles (64 pixels x 4 wavefronts, 128 pixels x 8 wavefronts),
    fp32 and planes output; ragged pixel counts and channel counts below the 192-wide tile."""
    B, C, H, W, K, R, st = case
    if tile128:
        monkeypatch.setenv("STEM_BX6_EXPER", "2")
    x, w, b = rnd((B, C, H, W), 11, -2, 2), (rnd((K, C, R, R), 12) / np.sqrt(C * R * R)).astype(np.float32), rnd((K,), 13, -0.1, 0.1)
    beta, gamma = rnd((K,), 14, 0.5, 1.5), (rnd((K, K), 15, 0.0, 0.1) + 0.1 * np.eye(K, dtype=np.float32)).astype(np.float32)
    ref = orc.conv2d_fwd(x, w, b, st, R // 2)
    if gdn:
        ref = orc.gdn_fwd(ref, beta, gamma)
    xp = F.Bf16Planes.split(dev(x))
    wp = F.pack_weight_bf16x3(dev(w))
    kw = dict(beta=dev(beta), gamma=dev(gamma)) if gdn else {}
    y = F.conv2d_bf16x6_fwd(xp, wp, dev(b), K, R, R, st, R // 2, **kw)
    assert_close(host(y), ref, what=f"bf16x6 conv {case} gdn={gdn}", floor=0.1)
    if K % 32 == 0:
        yp = F.conv2d_bf16x6_fwd(xp, wp, dev(b), K, R, R, st, R // 2, planes_out=True, **kw)
        assert torch.equal(yp.merge(), y), "planes output != fp32 output"
    else:
        with pytest.raises(Exception):
            F.conv2d_bf16x6_fwd(xp, wp, dev(b), K, R, R, st, R // 2, planes_out=True, **kw)


def test_first_layer_writes_the_same_values_as_planes(F):
    """The 3-channel first layer stays on the fp32-MFMA kernel; its planes epilogue must hold exactly the fp32 result."""
    x, w, b = rnd((2, 3, 40, 56), 21, 0, 1), (rnd((192, 3, 5, 5), 22) / np.sqrt(75)).astype(np.float32), rnd((192,), 23, -0.1, 0.1)
    beta, gamma = rnd((192,), 24, 0.5, 1.5), rnd((192, 192), 25, 0.0, 0.1)
    x4 = F.nchw3_to_nhwc4(dev(x))
    wp = F.pack_weight(dev(w), F.PACK_CONV_FWD_C4)
    y = F.conv2d_fwd_c4_gdn(x4, wp, dev(b), dev(beta), dev(gamma), 192, 5, 5, 2, 2)
    yp = F.conv2d_fwd_c4_gdn_planes(x4, wp, dev(b), dev(beta), dev(gamma), 192, 5, 5, 2, 2)
    assert torch.equal(yp.merge(), y)
    assert_close(host(y), orc.gdn_fwd(orc.conv2d_fwd(x, w, b, 2, 2), beta, gamma), what="g_a.0 + GDN", floor=0.1)


def test_analysis_transform_chain_vs_golden_and_fp32_kernels(F, golden, monkeypatch):
    """getY of the reference's I-frame model on the golden frames, forced through the bf16 chain (the golden batch is far below
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
    orig = F.conv2d_bf16x6_fwd
    monkeypatch.setattr(F, "conv2d_bf16x6_fwd", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    monkeypatch.setattr(L, "_BF16X6_MIN_PIXELS", 0)
    with torch.no_grad():
        y0, _ = imodel.getY(frames[0])
        y1, _ = imodel.getY(frames[1])
        assert len(calls) == 6, calls                   # g_a.2, g_a.4, g_a.6 of both frames ran on the bf16 kernel
        monkeypatch.setenv("STEM_BF16X6", "0")
        y0_32, _ = imodel.getY(frames[0])
        assert len(calls) == 6
    assert_close(host(y0), g["y0"], what="g_a(frame 0), bf16 chain", floor=0.1)
    assert_close(host(y1), g["f1:y_cur"], what="g_a(frame 1), bf16 chain", floor=0.1)
    assert_close(host(y0), host(y0_32), what="bf16 chain vs fp32-MFMA kernels", floor=0.1)


def test_chain_is_selected_at_the_bench_size_and_not_under_autograd(F, monkeypatch):
    """B=16 x 256x256 (the bench workload): g_a.0 hands planes to g_a.2, g_a.2 to g_a.4, g_a.4 returns fp32 for the small last
    layer; with autograd enabled (trainable transform) nothing is routed to the inference-only kernels."""
    from spatiotemporalentropymodel_amd.zoo import models
    torch.manual_seed(5)
    imodel = models["mbt2018"](quality=4).cuda().eval()
    x = torch.rand(16, 3, 256, 256, device="cuda")
    seen = []
    orig6, orig4 = F.conv2d_bf16x6_fwd, F.conv2d_fwd_c4_gdn_planes
    monkeypatch.setattr(F, "conv2d_bf16x6_fwd", lambda *a, **k: (seen.append(("bx6", a[0].shape, k.get("planes_out"))), orig6(*a, **k))[1])
    monkeypatch.setattr(F, "conv2d_fwd_c4_gdn_planes", lambda *a, **k: (seen.append(("c4",)), orig4(*a, **k))[1])
    with torch.no_grad():
        y = imodel.g_a(x)
    assert seen == [("c4",), ("bx6", (16, 192, 128, 128), True), ("bx6", (16, 192, 64, 64), False)], seen
    monkeypatch.setenv("STEM_BF16X6", "0")
    with torch.no_grad():
        y32 = imodel.g_a(x)
    assert_close(host(y), host(y32), what="g_a at B=16, bf16 chain vs fp32-MFMA kernels", floor=0.1)
    assert float((y - y32).abs().max()) <= 1e-5 * float(y32.abs().max())          # measured: 2.7e-6 of the largest latent
    monkeypatch.delenv("STEM_BF16X6")
    n = len(seen)
    y_grad = imodel.g_a(x)
    assert len(seen) == n and y_grad.requires_grad
