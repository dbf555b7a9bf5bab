"""CPU tests of the host-side product code: module surface / state-dict compatibility, the host rANS
library (bit-exact vs the reference's bytes), CDF tables from update(), flat parameter buffers."""
import os
import types
import zlib

import numpy as np
import pytest
import torch

import spatiotemporalentropymodel_amd as pkg
from spatiotemporalentropymodel_amd import entropy_models as em
from spatiotemporalentropymodel_amd.models import (JointAutoregressiveHierarchicalPriors, SpatioTemporalPriorModel,
                                                   SpatioTemporalPriorModel_Res, SpatioTemporalPriorModelWithoutSPM,
                                                   SpatioTemporalPriorModelWithoutSPMTPM, SpatioTemporalPriorModelWithoutTPM,
                                                   get_scale_table)


def test_parameter_counts_match_survey():
    """SURVEY.md §2a / BASELINE.md: 18,083,712 (big), 10,642,944 (small), mbt2018(192,192) 14,130,467."""
    assert sum(p.numel() for p in SpatioTemporalPriorModel_Res().parameters()) == 18_083_712
    assert len(list(SpatioTemporalPriorModel_Res().parameters())) == 41
    assert sum(p.numel() for p in SpatioTemporalPriorModel_Res(64, 96).parameters()) == 10_642_944
    assert sum(p.numel() for p in JointAutoregressiveHierarchicalPriors(192, 192).parameters()) == 14_130_467


def test_state_dict_keys_match_reference(golden):
    g = golden("stem_train_small.npz")
    ref_params = [k[len("final:psum:"):] for k in g if k.startswith("final:psum:")]
    m = SpatioTemporalPriorModel_Res(64, 96)
    ours = [n for n, _ in m.named_parameters()]
    assert ours == ref_params                      # same names, same ORDER (optimizer state interchange)
    sd = m.state_dict()
    for extra in ("entropy_bottleneck._offset", "entropy_bottleneck._quantized_cdf", "entropy_bottleneck._cdf_length",
                  "entropy_bottleneck.target", "entropy_bottleneck.likelihood_lower_bound.bound", "context_prediction.mask",
                  "gaussian_conditional._offset", "gaussian_conditional._quantized_cdf", "gaussian_conditional._cdf_length",
                  "gaussian_conditional.scale_table", "gaussian_conditional.scale_bound",
                  "gaussian_conditional.likelihood_lower_bound.bound", "gaussian_conditional.lower_bound_scale.bound"):
        assert extra in sd, extra
    assert len(sd) == 54
    assert tuple(sd["HD.0.weight"].shape) == (64, 256, 5, 5) and tuple(sd["EPM.0.weight"].shape) == (768, 576, 1, 1)
    isd = JointAutoregressiveHierarchicalPriors(64, 96).state_dict()
    for k in ("g_a.0.weight", "g_a.1.beta", "g_a.1.gamma", "g_a.1.beta_reparam.pedestal", "g_a.1.beta_reparam.lower_bound.bound",
              "g_a.1.gamma_reparam.pedestal", "g_a.1.gamma_reparam.lower_bound.bound", "g_s.6.bias", "h_a.0.weight", "h_s.4.weight",
              "entropy_parameters.4.bias", "context_prediction.mask"):
        assert k in isd, k


@pytest.mark.parametrize("cls", ["stem_baseline", "stem_baselinev2", "stem_roi", "stem_roi_wo_gsc", "stem_roi_i"])
def test_pixel_domain_models_state_dict_matches_reference(golden, cls):
    """Key names, order and shapes of the five compressai/models/stem_roi.py classes (fixture: the reference's own state_dict())."""
    import spatiotemporalentropymodel_amd.models as M
    sd = getattr(M, cls)().state_dict()
    mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in sd.items()]
    assert mine == list(golden("stem_roi.npz")[f"keys:{cls}"])


def test_ablation_variants_structure():
    assert not hasattr(SpatioTemporalPriorModelWithoutSPMTPM(), "TPM")
    m = SpatioTemporalPriorModelWithoutSPM(64, 96)
    assert m.HE[4].out_channels == 256            # hard-coded width in the two SPM-less ablations (:44-58,150-164)
    assert m.EPM[0].in_channels == 96 * 4 and not hasattr(m, "context_prediction")
    m = SpatioTemporalPriorModelWithoutTPM(64, 96)
    assert m.HE[4].out_channels == 64 and m.EPM[0].in_channels == 96 * 4 and not hasattr(m, "TPM")
    assert SpatioTemporalPriorModel(64, 96).EPM[0].in_channels == 96 * 6


def test_state_dict_roundtrip_with_cdf_buffers():
    a = SpatioTemporalPriorModel_Res(64, 96)
    a.update(force=True)
    b = SpatioTemporalPriorModel_Res(64, 96)
    b.load_state_dict(a.state_dict())               # resizes the empty CDF buffers first (spatiotemporalpriors.py:1058-1066)
    assert tuple(b.gaussian_conditional._quantized_cdf.shape) == (64, 3133)
    assert torch.equal(b.entropy_bottleneck._quantized_cdf, a.entropy_bottleneck._quantized_cdf)


def test_compressai_alias_imports():
    pkg.install_compressai_alias()
    import compressai
    from compressai.ans import BufferedRansEncoder, RansDecoder  # noqa: F401
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel_Res as R
    from compressai.zoo import models
    assert R is SpatioTemporalPriorModel_Res
    from compressai.models.stem_roi import stem_roi, stem_roi_i  # noqa: F401  (stem_roi/train_stem_roi.py:9)
    from compressai.models.stem_utils import SFT, SFTResblk  # noqa: F401
    import spatiotemporalentropymodel_amd.models as mine
    assert stem_roi is mine.stem_roi and SFT is mine.SFT
    assert isinstance(models["mbt2018"](quality=4), JointAutoregressiveHierarchicalPriors)
    assert compressai.available_entropy_coders() == ["ans"]
    compressai.set_entropy_coder("ans")
    with pytest.raises(ValueError):
        compressai.set_entropy_coder("nope")


# ----------------------------------------------------------------------------- error behaviour (test_entropy_models.py:96-144)
def test_entropy_model_error_paths():
    gc = em.GaussianConditional(None)
    with pytest.raises(ValueError):
        gc.quantize(torch.zeros(1, 1, 1, 1), "bogus")
    with pytest.raises(ValueError):
        gc._check_cdf_size()
    with pytest.raises(ValueError):
        em.GaussianConditional(1)
    with pytest.raises(ValueError):
        em.GaussianConditional([])
    with pytest.raises(ValueError):
        em.GaussianConditional([2.0, 1.0])
    with pytest.raises(ValueError):
        em.GaussianConditional(None, scale_bound=-0.1)
    with pytest.raises(ValueError):
        em._EntropyCoder("rangecoder42")
    with pytest.raises(NotImplementedError):
        em.EntropyModel().forward()


# ----------------------------------------------------------------------------- host codec, bit exact
def test_gaussian_update_tables_match_reference(golden):
    g = golden("codec.npz")
    gc = em.GaussianConditional(None)
    assert gc.update_scale_table(get_scale_table()) is True
    assert gc.update_scale_table(get_scale_table()) is False
    np.testing.assert_array_equal(gc.scale_table.numpy(), g["gc:scale_table"])
    np.testing.assert_array_equal(gc._offset.numpy(), g["gc:offset"])
    np.testing.assert_array_equal(gc._cdf_length.numpy(), g["gc:cdf_length"])
    cdf = gc._quantized_cdf.numpy()
    assert tuple(cdf.shape) == tuple(g["gc:cdf_shape"])
    assert zlib.crc32(np.ascontiguousarray(cdf).tobytes()) == int(g["gc:cdf_crc32"][0])


def test_pmf_to_quantized_cdf_matches_reference(golden):
    g = golden("codec.npz")
    for i in range(4):
        np.testing.assert_array_equal(em.pmf_to_quantized_cdf(torch.from_numpy(g[f"pmf{i}"]), 16).numpy().astype(np.uint32), g[f"cdf{i}"])
    with pytest.raises(ValueError):
        em.pmf_to_quantized_cdf(torch.zeros(4), 16)


def _tables():
    gc = em.GaussianConditional(None)
    gc.update_scale_table(get_scale_table())
    return gc.host_tables(), gc


def test_rans_streams_are_byte_identical_to_reference(golden):
    g = golden("codec.npz")
    t, gc = _tables()
    for i in range(4):
        sym, idx = g[f"rans{i}:symbols"], g[f"rans{i}:indexes"]
        s = em.RansEncoder().encode_with_indexes(sym, idx, t)
        assert s == g[f"rans{i}:bytes"].tobytes()
        np.testing.assert_array_equal(em.RansDecoder().decode_with_indexes_np(s, idx, t), sym)
        # the reference's list-of-lists calling convention works too
        s2 = em.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), gc.quantized_cdf.tolist(),
                                                  gc.cdf_length.tolist(), gc.offset.tolist())
        assert s2 == s
    enc = em.BufferedRansEncoder()
    enc.encode_with_indexes(g["bufrans:sym_a"], g["bufrans:idx_a"], t)
    enc.encode_with_indexes(g["bufrans:sym_b"], g["bufrans:idx_b"], t)
    s = enc.flush()
    assert s == g["bufrans:bytes"].tobytes()
    dec = em.RansDecoder()
    dec.set_stream(s)
    assert dec.decode_stream(g["bufrans:idx_a"], t) == g["bufrans:sym_a"].tolist()
    assert dec.decode_stream(g["bufrans:idx_b"], t) == g["bufrans:sym_b"].tolist()


def test_rans_edge_cases():
    t, _ = _tables()
    # empty and 1-symbol streams (the reference's encoder overruns its own buffer here, rans_interface.cpp:170)
    s = em.RansEncoder().encode_with_indexes(np.zeros(0, np.int32), np.zeros(0, np.int32), t)
    assert len(s) == 8
    for v in (0, 5, -100000, 2 ** 20):
        s = em.RansEncoder().encode_with_indexes(np.array([v], np.int32), np.array([3], np.int32), t)
        assert em.RansDecoder().decode_with_indexes(s, np.array([3], np.int32), t) == [v]
    # long ragged stream with many escapes, maximum / minimum index
    rng = np.random.default_rng(3)
    idx = rng.choice([0, 63], size=20001).astype(np.int32)
    sym = rng.integers(-3000, 3000, size=20001).astype(np.int32)
    s = em.RansEncoder().encode_with_indexes(sym, idx, t)
    np.testing.assert_array_equal(em.RansDecoder().decode_with_indexes_np(s, idx, t), sym)
    # invalid input is an error, not UB
    with pytest.raises(RuntimeError):
        em.RansEncoder().encode_with_indexes(np.array([1], np.int32), np.array([64], np.int32), t)
    with pytest.raises(RuntimeError):
        em.RansDecoder().decode_with_indexes(b"\x00" * 5, np.array([0], np.int32), t)
    with pytest.raises(RuntimeError):
        em.RansDecoder().decode_stream(np.array([0], np.int32), t)          # set_stream not called
    with pytest.raises(RuntimeError):
        em.RansDecoder().decode_with_indexes(s[: len(s) // 2], idx, t)      # truncated stream


def test_rans_decoder_survives_corrupt_streams():
    """the escape path trusted the nibble count read from the stream (shift >= 32 = UB, attacker-controlled
    loop).  Corrupt streams must come back as an error or as (wrong) integers -- never crash, hang or read out of bounds."""
    t, _ = _tables()
    rng = np.random.default_rng(11)
    idx = rng.integers(0, 64, size=600).astype(np.int32)
    sym = rng.integers(-40000, 40000, size=600).astype(np.int32)            # almost every symbol takes the escape path
    good = em.RansEncoder().encode_with_indexes(sym, idx, t)
    np.testing.assert_array_equal(em.RansDecoder().decode_with_indexes_np(good, idx, t), sym)
    errors = 0
    for trial in range(300):
        b = bytearray(good)
        for _ in range(1 + trial % 6):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if trial % 7 == 0:
            b = b[: int(rng.integers(8, len(b)))] + bytes([0xFF] * 64)         # runs of 0xF nibbles: the old unbounded count
        try:
            out = em.RansDecoder().decode_with_indexes_np(bytes(b), idx, t)
            assert out.shape == sym.shape and out.dtype == np.int32
        except RuntimeError as e:
            errors += 1
            assert "rans decode" in str(e)
    assert errors > 0                                                          # the bound actually fires on some of them


def test_entropy_bottleneck_update_tables_match_reference(golden):
    g = golden("ops_small.npz")
    eb = em.EntropyBottleneck(4)
    with torch.no_grad():
        for n, p in eb.named_parameters():
            p.copy_(torch.from_numpy(g[f"eb:p:{n}"]))
    assert eb.update(force=True)
    np.testing.assert_array_equal(eb._offset.numpy(), g["eb:offset"])
    np.testing.assert_array_equal(eb._cdf_length.numpy(), g["eb:cdf_length"])
    np.testing.assert_array_equal(eb._quantized_cdf.numpy(), g["eb:cdf"])
    assert eb.update() is False


# ----------------------------------------------------------------------------- flat buffers
def test_flat_parameters_views_and_order():
    from spatiotemporalentropymodel_amd.optim import FlatParameters
    m = SpatioTemporalPriorModel_Res(64, 96)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    main = sorted([(n, p) for n, p in m.named_parameters() if not n.endswith(".quantiles")], key=lambda t: t[0])
    flat = FlatParameters(main)
    assert flat.numel >= sum(p.numel() for _, p in main) and flat.numel % 4 == 0
    for (n, p), o in zip(main, flat.offsets):
        assert torch.equal(p, before[n]) and o % 4 == 0
        assert p.data_ptr() == flat.data.data_ptr() + 4 * o          # a view, not a copy
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    flat.data.mul_(2.0)
    n0, p0 = main[0]
    assert torch.equal(p0, before[n0] * 2)


# ----------------------------------------------------------------------------- container (compressai_examples/codec.py:63-220)
def test_bitstream_container_matches_reference_bytes(golden):
    import io
    from spatiotemporalentropymodel_amd import bitstream as bs
    g = golden("container.npz")
    frames = []
    for i in range(3):
        model, metric, q, size, shape = str(g[f"case{i}:meta"][0]).split("|")
        size, shape = tuple(map(int, size.split(","))), tuple(map(int, shape.split(",")))
        strings = []
        j = 0
        while f"case{i}:s{j}" in g:
            strings.append([g[f"case{i}:s{j}"].tobytes()])
            j += 1
        f = io.BytesIO()
        bs.write_frame(f, bs.get_header(model, metric, int(q)), size, shape, strings)
        assert f.getvalue() == g[f"case{i}:bytes"].tobytes()
        hdr, osz, shp, back = bs.read_frame(io.BytesIO(f.getvalue()))
        assert hdr == (model, metric, int(q)) and osz == size and shp == shape and back == strings
        frames.append((bs.get_header(model, metric, int(q)), size, shape, strings))
    f = io.BytesIO()
    bs.write_sequence(f, frames)
    seq = bs.read_sequence(io.BytesIO(f.getvalue()))
    assert [fr[3] for fr in seq] == [fr[3] for fr in frames] and len(seq) == 3
    with pytest.raises(ValueError):
        bs.read_frame(io.BytesIO(f.getvalue()[4:40]))
    with pytest.raises(ValueError):
        bs.get_header("nope", "mse", 1)
    x = torch.from_numpy(g["pad:x"])
    xp = bs.pad(x, 64)
    np.testing.assert_array_equal(xp.numpy(), g["pad:xp"])
    np.testing.assert_array_equal(bs.crop(xp, (50, 75)).numpy(), g["pad:back"])


# ---------------------------------------------------------------------------------------------------------------
# round 2: optimiser / checkpoint interchange, root-utils mirror, CDF-buffer resizing
def _sorted_params(net):
    return sorted(net.named_parameters(), key=lambda t: t[0])


def test_fused_adam_is_a_torch_optimizer_and_speaks_adams_state_dict(tmp_path):
    """stem/trainSTEM.py:123 hands the optimiser to ReduceLROnPlateau and :286-297 checkpoints optimizer.state_dict():
    both must work with FusedClipAdam, in both directions with torch.optim.Adam."""
    from spatiotemporalentropymodel_amd.optim import FlatParameters, FusedClipAdam
    from spatiotemporalentropymodel_amd.utils import save_checkpoint
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    opt = FusedClipAdam(FlatParameters(_sorted_params(net)), 1e-4, max_norm=1.0)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", patience=0, factor=0.5)
    sched.step(1.0)
    sched.step(2.0)                                   # worse -> lr halves in param_groups[0], which step() reads
    assert opt.lr == pytest.approx(5e-5) and opt.param_groups[0]["lr"] == pytest.approx(5e-5)
    assert opt.state_dict()["state"] == {}            # like torch.optim.Adam before its first step
    # reference-side optimiser with two steps of state
    ref_net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    ref = torch.optim.Adam([p for _, p in _sorted_params(ref_net)], lr=3e-4)
    for _ in range(2):
        ref.zero_grad()
        ref_net(torch.randn(4, 3)).square().sum().backward()
        ref.step()
    rsd = ref.state_dict()
    opt.load_state_dict(rsd)
    assert opt.t == 2 and opt.lr == pytest.approx(3e-4)
    for i, (p, o) in enumerate(zip(opt.flat.params, opt.flat.offsets)):
        n = p.numel()
        assert torch.equal(opt.m[o:o + n].view(p.shape), rsd["state"][i]["exp_avg"])
        assert torch.equal(opt.v[o:o + n].view(p.shape), rsd["state"][i]["exp_avg_sq"])
    # ... and back, through a checkpoint file written by the utils mirror
    path = tmp_path / "ckpt.pth.tar"
    save_checkpoint({"epoch": 3, "optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict()}, str(path))
    back = torch.load(str(path), weights_only=False)
    assert back["optimizer"]["param_groups"][0].keys() == rsd["param_groups"][0].keys()
    ref2 = torch.optim.Adam([p for _, p in _sorted_params(ref_net)], lr=1.0)
    ref2.load_state_dict(back["optimizer"])
    for i in range(4):
        assert float(ref2.state_dict()["state"][i]["step"]) == 2.0
        assert torch.equal(ref2.state_dict()["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"])
    # torch-1.7 style integer step counts load as well; inconsistent counts are refused
    old = {"state": {i: dict(s, step=7) for i, s in rsd["state"].items()}, "param_groups": rsd["param_groups"]}
    opt.load_state_dict(old)
    assert opt.t == 7
    old["state"][2]["step"] = 8
    with pytest.raises(ValueError):
        opt.load_state_dict(old)
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [dict(rsd["param_groups"][0], params=[0, 1])]})


def test_root_utils_mirror():
    """reference utils.py:77-94 (MovingAverage), :97-101, :138-139 and the names its scripts import with `from utils import *`."""
    from spatiotemporalentropymodel_amd import utils as U
    for name in ("EMLoss", "RateDistortionLoss", "PixelwiseRateDistortionLoss", "MovingAverage", "quality2lambda",
                 "configure_optimizers", "save_checkpoint"):
        assert hasattr(U, name), name
    ma = U.MovingAverage(3)
    assert [ma.next(v) for v in (3, 6, 9, 12)] == [3.0, 4.5, 6.0, 9.0]
    assert len(ma.queue) == 3 and ma.Max_size == 3
    q = torch.tensor([0.0, 0.5, 1.0])
    assert torch.allclose(U.quality2lambda(q), 0.002 * torch.exp(3.4409 * q))


def test_update_registered_buffers_policies():
    """compressai/models/utils.py:27-110 behaviour: resize_if_empty / resize / register + the error types."""
    from spatiotemporalentropymodel_amd.models.utils import find_named_buffer, update_registered_buffers
    m = torch.nn.Module()
    m.register_buffer("_offset", torch.IntTensor())
    m.register_buffer("_cdf_length", torch.IntTensor([1, 2, 3]))
    sd = {"em._offset": torch.zeros(7, dtype=torch.int32), "em._cdf_length": torch.zeros(5, dtype=torch.int32),
          "em.fresh": torch.zeros(2, 4, dtype=torch.int32)}
    update_registered_buffers(m, "em", ["_offset", "_cdf_length"], sd)
    assert m._offset.shape == (7,) and m._cdf_length.shape == (3,)          # only the empty one took the new shape
    update_registered_buffers(m, "em", ["_cdf_length"], sd, policy="resize")
    assert m._cdf_length.shape == (5,)
    assert find_named_buffer(m, "_offset") is m._offset and find_named_buffer(m, "nope") is None
    with pytest.raises(ValueError):
        update_registered_buffers(m, "em", ["nope"], sd)
    with pytest.raises(ValueError):
        update_registered_buffers(m, "em", ["_offset"], sd, policy="bogus")
    with pytest.raises(RuntimeError):
        update_registered_buffers(m, "em", ["_offset"], sd, policy="register")


def test_entropy_bottleneck_init_values():
    """entropy_models.py:303-340 upstream: softplus(matrix_i) == 1/(s * fan_out), factors 0, quantiles (-10,0,10), target logits."""
    eb = em.EntropyBottleneck(6)
    s = 10 ** (1 / 5)
    for i, fo in enumerate((3, 3, 3, 3, 1)):
        mtx = getattr(eb, f"_matrix{i}")
        assert mtx.shape == (6, fo, (1, 3, 3, 3, 3)[i])
        assert torch.allclose(torch.nn.functional.softplus(mtx), torch.full_like(mtx, 1 / s / fo), rtol=1e-6)
        b = getattr(eb, f"_bias{i}")
        assert b.shape == (6, fo, 1) and float(b.detach().abs().max()) <= 0.5
        if i < 4:
            assert float(getattr(eb, f"_factor{i}").abs().max()) == 0.0
    assert torch.equal(eb.quantiles, torch.tensor([-10.0, 0.0, 10.0]).repeat(6, 1, 1))
    t = float(np.log(2 / 1e-9 - 1))
    assert torch.allclose(eb.target, torch.tensor([-t, 0.0, t]))
    # argument errors of EntropyModel.decompress (entropy_models.py:245-264 upstream) are ValueErrors
    idx = torch.zeros(1, 6, 2, 2)
    check = em.EntropyModel._check_decompress_args
    for bad in (("notalist", idx, None), ([b"", b""], idx, None), ([b""], torch.zeros(6, 2, 2), None),
                ([b""], idx, torch.zeros(2, 6, 2, 2)), ([b""], idx, torch.zeros(1, 6, 1, 2))):
        with pytest.raises(ValueError):
            check(*bad)
    check([b""], idx, torch.zeros(1, 6, 1, 1))
    check((b"",), idx, torch.zeros(1, 6, 2, 2))


def test_masked_conv_mask_layouts():
    """compressai_tests/test_layers.py:29-115: mask A zeroes the centre tap and everything after it in raster order, B keeps the centre"""
    from spatiotemporalentropymodel_amd.layers import MaskedConv2d
    a = MaskedConv2d(1, 1, 5, padding=2, mask_type="A").mask[0, 0]
    b = MaskedConv2d(1, 1, 5, padding=2, mask_type="B").mask[0, 0]
    exp_a = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]], dtype=torch.float32)
    exp_b = exp_a.clone()
    exp_b[2, 2] = 1
    assert torch.equal(a, exp_a) and torch.equal(b, exp_b)
    a3 = MaskedConv2d(1, 1, 3, padding=1, mask_type="A").mask[0, 0]
    assert torch.equal(a3, torch.tensor([[1, 1, 1], [1, 0, 0], [0, 0, 0]], dtype=torch.float32))
    with pytest.raises(ValueError):
        MaskedConv2d(1, 1, 3, mask_type="C")


def test_bf16_planes_views_and_engine_routing_table():
    """Host logic of the fp16 route (no device work): channel views of a planes tensor (32-aligned, same storage and pitch), and which
    layers of the full-size STEM model the engine sends to the fp16 kernels: forward / input gradient of the stride-1 unmasked
    convolutions and of the masked context convolution (forward, over its live taps), weight gradient of every stride-1
    convolution, nothing for the stride-2 and transposed layers."""
    import torch
    from spatiotemporalentropymodel_amd import functional as F
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    payload, total = F.F16Planes.nbytes(2 * 4 * 4, 160)
    data = torch.zeros(total, dtype=torch.uint8)
    p = F.F16Planes(data, (2, 160, 4, 4), payload)
    assert p.dense and p.pix_bytes == 5 * 128 and p.q_ptr() == data.data_ptr() + payload
    v = p.channels(32, 128)
    assert v.shape == (2, 96, 4, 4) and v.pix_bytes == p.pix_bytes and v.byte_offset == 128 and not v.dense
    assert v.channels(32, 64).byte_offset == 2 * 128 and v.data is data
    assert v.q_ptr() == p.q_ptr()                             # a channel view shares the scale record of the whole tensor
    for bad in ((16, 80), (0, 200), (64, 64)):
        with pytest.raises(ValueError):
            p.channels(*bad)
    eng = SpatioTemporalPriorModel_Res().engine()
    fwd = {n: l.fx3 for n, l in zip(("HE0", "HE2", "HE4", "HD0", "HD2", "HD4", "TPM0", "TPM2", "TPM4", "CTX", "EPM0", "EPM2", "EPM4"), eng.layers)}
    wg = {n: l.wg3 for n, l in zip(fwd, eng.layers)}
    assert fwd == {"HE0": True, "HE2": False, "HE4": False, "HD0": False, "HD2": False, "HD4": True, "TPM0": True, "TPM2": True, "TPM4": True,
                   "CTX": True, "EPM0": True, "EPM2": True, "EPM4": True}
    assert wg == fwd
    assert eng.CTX.taps == 12 and all(l.taps == 0 for l in eng.layers if l is not eng.CTX)     # 5x5 type-A mask: 12 live taps
    # the strided-convolution faces of the hyper path's stride-2 layers (HE.2 / HE.4 forward, HD.0 / HD.2 input gradient)
    assert {n: l.fx3s for n, l in zip(fwd, eng.layers)} == dict({n: False for n in fwd}, HE2=True, HE4=True, HD0=True, HD2=True)
    # ... and (round 5) their transposed faces + weight gradients: chain by chain
    assert {n: l.fx3t for n, l in zip(fwd, eng.layers)} == dict({n: False for n in fwd}, HE2=True, HE4=True, HD0=True, HD2=True)
    from spatiotemporalentropymodel_amd import config as _cfg
    with _cfg.override(engine_transposed_f16x3=False):
        assert not any(l.fx3t for l in SpatioTemporalPriorModel_Res().engine().layers)
    # Chains are routed as a whole: with latent channel counts that are not multiples of 16 some layers of a chain
    # are ineligible (C % 32), and a half-routed chain would call a kernel whose packed weights were never allocated
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel, SpatioTemporalPriorModelWithoutTPM
    names = list(fwd)
    for cls, cin in ((SpatioTemporalPriorModel_Res, 24), (SpatioTemporalPriorModel_Res, 48), (SpatioTemporalPriorModel, 24),
                     (SpatioTemporalPriorModelWithoutTPM, 24), (SpatioTemporalPriorModelWithoutTPM, 40)):
        e = cls(64, cin).engine()
        for group in (e.TPM, e.EPM):
            if group:
                assert len({l.fx3 for l in group}) == 1, (cls.__name__, cin, [l.fx3 for l in group])
        if e.EPM[0].fx3:
            assert (2 * cin) % 32 == 0                        # the EPM input gradient is read through 32-aligned channel views
        for l in e.layers:
            assert not l.fx3 or l.fx3_eligible() or l.fx3_masked_eligible()
            assert not l.fx3s or (l.fx3s_eligible() and not l.fx3)
            # every layer has exactly the packed weights its route needs once allocated (CPU: allocation only)
            l.alloc_packs(torch.device("cpu"))
            assert not l.fx3t or (l.fx3s and l.fx3t_eligible())
            if l.fx3t:      # round 5: both faces on the fp16 kernel (strided image + four phase images), no fp32 copy
                assert l.wp6_fwd is not None and (l.wp6_dgrad is not None) == l.need_dgrad and l.wp_dgrad is None and l.wp_fwd.numel() == 0
                flips = [d.flip for role in (0, 1) for d in l.role_descs(role)[1]]
                assert flips == ([0, 2] if l.kind == "conv" else [2, 0]) and not l.role_descs(0)[0] and not l.role_descs(1)[0]
            elif l.fx3s:    # one face on the fp16 kernel, the other on igemm.hip
                assert (l.wp6_fwd is not None) == (l.kind == "conv") and (l.wp6_dgrad is not None) == (l.kind == "deconv")
                assert l.wp_fwd is not None and (l.wp_dgrad is not None) == (l.kind == "conv" and l.need_dgrad)
            else:
                assert (l.wp6_fwd is not None) == l.fx3 and (l.wp_fwd is not None) == (not l.fx3)
            for role in (0, 1):     # descriptors can be built for both roles (every buffer they name exists)
                l.role_descs(role)
    e24 = SpatioTemporalPriorModel_Res(64, 24).engine()
    assert not any(l.fx3 for l in e24.TPM + e24.EPM)          # 2 * 24 = 48 channels: not a multiple of 32
    e48 = SpatioTemporalPriorModel_Res(64, 48).engine()
    assert all(l.fx3 for l in e48.EPM) and not any(l.fx3 for l in e48.TPM)     # TPM.0 contracts over Cin = 48 channels
    e64 = SpatioTemporalPriorModel_Res(64, 64).engine()
    assert all(l.fx3 for l in e64.EPM + e64.TPM)


def test_bf16_chain_is_not_selected_when_planes_exceed_a_buffer_view():
    """6 full-HD frames per call: their planes (two fp16 numbers per value, 2.4 GB) exceed the 2 GiB views the kernels address
    operands through: the chain must not start; 5 frames fit."""
    from spatiotemporalentropymodel_amd import layers as L
    from spatiotemporalentropymodel_amd.zoo import models
    conv1 = models["mbt2018"](quality=4).g_a[2]
    assert L._f16x3_shape_ok(conv1, (5, 192, 544, 960))
    assert not L._f16x3_shape_ok(conv1, (6, 192, 544, 960))
    assert L._planes_fit(5 * 544 * 960, 192) and not L._planes_fit(6 * 544 * 960, 192)
    assert not L._f16x3_shape_ok(conv1, (1, 192, 32, 32))          # too few output pixels for the 192-wide kernel


def test_planes_byte_count_matches_the_library():
    """F16Planes.empty sizes its storage in Python (hot path); the C ABI's stem_f16x2_planes_bytes is the definition."""
    from spatiotemporalentropymodel_amd import _lib
    from spatiotemporalentropymodel_amd import functional as F
    lib = _lib.hip()
    for npix, C in ((1, 32), (4096, 192), (65280, 1152), (7, 96), (64, 128), (65, 160), (16 * 128 * 128, 192)):
        payload, total = F.F16Planes.nbytes(npix, C)
        assert lib.stem_f16x2_planes_qrec_offset(npix, C) == payload == npix * (C // 32) * 128
        assert lib.stem_f16x2_planes_bytes(npix, C) == total
        # the scale record: 16 header words + one slot per 64-pixel x 128-channel producer tile, 16-byte granules
        assert total - payload >= (16 + -(-npix // 64) * -(-C // 128)) * 4 and (total - payload) % 16 == 0
    assert lib.stem_f16x2_planes_bytes(10, 48) == 0


def test_tuned_schedule_defaults_and_cu_mask_words(monkeypatch):
    """trainer.tuned_schedule installs the schedule bench.py measures unless the environment already decides, and the CU-mask
    words handed to hipExtStreamCreateWithCUMask cover exactly the requested CUs (CPU: parsing only, no stream is created)."""
    from spatiotemporalentropymodel_amd import functional as F
    from spatiotemporalentropymodel_amd import trainer
    assert trainer.SCHEDULE_DEFAULTS == {"STEM_STREAM_PRIO": "latents=0,side=-1,compute=-1", "STEM_STREAM_CUMASK": "latents=block:160"}
    monkeypatch.setenv("STEM_STREAM_CUMASK", "latents=block:160")
    words = F._cu_mask("latents")
    assert len(words) == 8 and sum(bin(w).count("1") for w in words) == 160 and words[:5] == [0xFFFFFFFF] * 5 and words[5:] == [0, 0, 0]
    assert F._cu_mask("side") is None
    monkeypatch.setenv("STEM_STREAM_CUMASK", "latents=mod8:3,side=block:32")
    assert all(w == 0x07070707 for w in F._cu_mask("latents")) and F._cu_mask("side") == [0xFFFFFFFF] + [0] * 7
    monkeypatch.setenv("STEM_STREAM_CUMASK", "")          # set and empty: no mask, and tuned_schedule must not overwrite it
    assert F._cu_mask("latents") is None
    monkeypatch.setenv("STEM_STREAM_PRIO", "")
    monkeypatch.setattr(F, "_STREAM_PRIO", None)
    monkeypatch.setattr(F, "make_stream", lambda device, role: (F.__dict__.__setitem__("_STREAM_PRIO", {}), None)[1])
    import contextlib
    assert isinstance(trainer.tuned_schedule(torch.device("cpu")), contextlib.nullcontext)
    assert os.environ["STEM_STREAM_CUMASK"] == "" and os.environ["STEM_STREAM_PRIO"] == ""


def test_host_codec_under_sanitizers(tmp_path):
    """`make sanitize` builds csrc/rans_host.cpp with -fsanitize=address,undefined (the reference's DEBUG_BUILD,
    setup.py:56-60); the rANS / pmf->CDF / table / container tests of this file -- including the corrupt-stream decoder test --
    then run against that library in a child process with the sanitizer runtime preloaded.  Any report fails the child
    (-fno-sanitize-recover, ASan's default abort)."""
    import shutil
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(REPO, "spatiotemporalentropymodel_amd", "csrc")
    cxx = shutil.which(os.environ.get("CXX", "g++"))
    if cxx is None:
        pytest.skip("no host C++ compiler")
    asan = subprocess.run([cxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("the compiler ships no AddressSanitizer runtime")
    subprocess.check_call(["make", "-C", csrc, "sanitize"], stdout=subprocess.DEVNULL)
    lib = os.path.join(REPO, "spatiotemporalentropymodel_amd", "libstem_rans_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ, STEM_RANS_LIBRARY=lib, LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
           "-k", "rans or pmf_to_quantized or update_tables or bitstream_container"]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    import re
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= 6, out[-2000:]            # the child really ran the codec tests on the sanitized library


def test_runtime_config_is_the_one_reader_of_the_stem_switches(monkeypatch):
    """config.StemRuntimeConfig: defaults, the environment's overrides (historical STEM_* spellings), override() blocks on top;
    no module of the package reads a STEM_* switch from os.environ on its own (the two library paths of _lib.py aside)"""
    import dataclasses
    import re
    from spatiotemporalentropymodel_amd import config
    for v in config._ENV.values():
        monkeypatch.delenv(v, raising=False)
    d = config.runtime()
    assert d == config.StemRuntimeConfig() and d.engine_f16x3 and d.dp_min_bytes == 8 << 20 and d.ar_persistent
    assert {f.name for f in dataclasses.fields(d)} == set(config._ENV)
    monkeypatch.setenv("STEM_ENGINE_F16X3", "0")
    monkeypatch.setenv("STEM_DP_MIN_BYTES", "4096")
    monkeypatch.setenv("STEM_STREAM_PRIO", "side=-1")
    c = config.runtime()
    assert not c.engine_f16x3 and c.dp_min_bytes == 4096 and c.stream_prio == "side=-1" and c.engine_wgrad_f16x3
    with config.override(engine_f16x3=True, ar_pipeline=True):
        assert config.runtime().engine_f16x3 and config.runtime().ar_pipeline
        monkeypatch.setenv("STEM_AR_STEPWISE", "1")                  # an environment change inside the block keeps the block's fields
        assert config.runtime().engine_f16x3 and config.runtime().ar_stepwise
        with config.override(engine_f16x3=False):
            assert not config.runtime().engine_f16x3
        assert config.runtime().engine_f16x3
    assert not config.runtime().engine_f16x3 and not config.runtime().ar_pipeline
    with pytest.raises(AttributeError):
        with config.override(no_such_field=1):
            pass
    assert config.from_env({"STEM_PIN_RANKS": "0", "STEM_DP_MIN_BYTES": "2"}).pin_ranks is False
    assert config.from_env({"STEM_DP_MIN_BYTES": "2"}).dp_min_bytes == 2
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spatiotemporalentropymodel_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py") and fn not in ("config.py", "_lib.py"):
                text = open(os.path.join(root, fn)).read()
                reads = re.findall(r'environ(?:\.get\(|\[)\s*"(STEM_[A-Z0-9_]+)"', text)
                assert not reads, (fn, reads)


def test_launch_tape_trampolines_on_host():
    """csrc/tape.hip re-issues recorded C calls through x86-64 SysV trampolines (integer-class arguments in registers / on the stack,
    floats and doubles as SSE patterns).  No GPU needed for that part: the recorded functions here are ctypes callbacks -- 1, 6, 7
    and 19 integer arguments with 0..6 SSE arguments in between, per-replay deltas, a patched pointer-sized argument, a failing call."""
    import ctypes as C
    import struct
    from spatiotemporalentropymodel_amd import _lib
    lib = _lib.hip()
    seen = []

    def make(nint, kinds):
        """callback taking the argument classes of `kinds` (0 int64, 1 float, 2 double), logging what it received"""
        ctys = [C.c_longlong if k == 0 else (C.c_float if k == 1 else C.c_double) for k in kinds]
        proto = C.CFUNCTYPE(C.c_int, *ctys)

        def body(*a):
            seen.append(tuple(a))
            return 0
        cb = proto(body)
        return cb, C.cast(cb, C.c_void_p).value

    def add(tape, addr, kinds, vals, deltas=None):
        n = len(kinds)
        k = (C.c_ubyte * n)(*[1 if x else 0 for x in kinds])
        iv = (C.c_longlong * n)(*[int(v) if kk == 0 else 0 for kk, v in zip(kinds, vals)])
        fbits = []
        for kk, v in zip(kinds, vals):
            if kk == 1:
                fbits.append(struct.unpack("<d", struct.pack("<fI", float(v), 0))[0])      # float bits in the low half of the register
            elif kk == 2:
                fbits.append(float(v))
            else:
                fbits.append(0.0)
        fv = (C.c_double * n)(*fbits)
        dl = (C.c_longlong * n)(*(deltas or [0] * n))
        idx = lib.stem_tape_add_call(tape, addr, n, k, iv, fv, dl)
        assert idx >= 0, lib.stem_last_error()
        return idx

    tape = lib.stem_tape_create()
    keep = []
    cases = [
        ([0], [41]),
        ([0, 0, 0, 0, 0, 0], [1, -2, 3, -4, 5, 1 << 40]),
        ([0, 1, 0, 2, 0, 0, 0, 0, 0], [7, 1.5, 8, -2.25, 9, 10, 11, 12, 13]),                      # seven integers: one on the stack
        ([0] * 10 + [2, 1, 1, 2, 1, 2] + [0] * 9, list(range(100, 110)) + [0.5, 1.25, -3.0, 4.75, 8.0, -0.125] + list(range(200, 209))),
    ]
    for kinds, vals in cases:
        cb, addr = make(sum(1 for k in kinds if k == 0), kinds)
        keep.append(cb)
        add(tape, addr, kinds, vals)
    # per-replay deltas and a patched argument
    cb, addr = make(3, [0, 0, 0])
    keep.append(cb)
    e_dyn = add(tape, addr, [0, 0, 0], [1000, 5, 77], deltas=[0, 3, 0])
    assert lib.stem_tape_length(tape) == 5
    assert lib.stem_tape_set_iarg(tape, e_dyn, 2, 0x7F0000001234) == 0
    assert lib.stem_tape_set_iarg(tape, e_dyn, 3, 1) != 0                                         # no such argument
    assert lib.stem_tape_replay(tape, 0, 5, 4) == 0
    assert seen[0] == (41,) and seen[1] == (1, -2, 3, -4, 5, 1 << 40)
    assert seen[2] == (7, 1.5, 8, -2.25, 9, 10, 11, 12, 13)
    assert seen[3] == tuple(list(range(100, 110)) + [0.5, 1.25, -3.0, 4.75, 8.0, -0.125] + list(range(200, 209)))
    assert seen[4] == (1000, 5 + 4 * 3, 0x7F0000001234)
    # a failing call stops the replay and reports its entry
    proto = C.CFUNCTYPE(C.c_int, C.c_longlong)
    bad = proto(lambda a: -7)
    keep.append(bad)
    e_bad = add(tape, C.cast(bad, C.c_void_p).value, [0], [1])
    n_before = len(seen)
    assert lib.stem_tape_replay(tape, 4, e_bad + 1, 1) == -(e_bad + 1)
    assert len(seen) == n_before + 1 and seen[-1] == (1000, 8, 0x7F0000001234)
    # float slots are patchable (the optimiser hyper-parameters a scheduler edits between replays): float and double patterns
    e_f = 2                                                                                       # kinds [0, 1, 0, 2, 0, ...]
    assert lib.stem_tape_set_farg(tape, e_f, 0, struct.unpack("<d", struct.pack("<fI", 0.375, 0))[0]) == 0
    assert lib.stem_tape_set_farg(tape, e_f, 1, 1e-5) == 0
    assert lib.stem_tape_set_farg(tape, e_f, 2, 0.0) != 0 and b"no float argument" in lib.stem_last_error()
    assert lib.stem_tape_set_farg(tape, 0, 0, 0.0) != 0                                           # entry 0 has no float argument
    assert lib.stem_tape_replay(tape, e_f, e_f + 1, 1) == 0
    assert seen[-1] == (7, 0.375, 8, 1e-5, 9, 10, 11, 12, 13)
    lib.stem_tape_destroy(tape)


def test_launch_tape_contract_is_checked_for_every_entry_point():
    """The trampolines' calling contract (scalar / pointer arguments only, <= 6 SSE-class, <= 40 integer-class) is a static_assert
    per int-returning entry point in csrc/tape.hip, driven by csrc/tape_entries.inc: the list must be what the header declares
    (tools/gen_tape_table.py), the library must know every listed function as recordable and nothing else, and the ctypes
    prototypes the recorder classifies arguments by must agree with the header on which arguments are float / double."""
    import ctypes as C
    import re
    import subprocess
    import sys
    from spatiotemporalentropymodel_amd import _lib
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([sys.executable, os.path.join(repo, "tools", "gen_tape_table.py"), "--check"]) == 0, \
        "csrc/tape_entries.inc is stale: run python tools/gen_tape_table.py"
    listed = re.findall(r"STEM_TAPE_ENTRY\((\w+)\)", open(os.path.join(repo, "spatiotemporalentropymodel_amd", "csrc", "tape_entries.inc")).read())
    lib, raw = _lib.hip(), C.CDLL(_lib.HIP_SO)
    assert len(listed) > 100
    for name in listed:
        assert lib.stem_tape_entry_recordable(C.cast(getattr(raw, name), C.c_void_p).value) == 1, name
    for name in ("stem_last_error", "stem_tape_create", "stem_tape_destroy", "stem_adam_chunk"):          # not int-returning
        assert lib.stem_tape_entry_recordable(C.cast(getattr(raw, name), C.c_void_p).value) == 0, name
    cb = C.CFUNCTYPE(C.c_int)(lambda: 0)
    assert lib.stem_tape_entry_recordable(C.cast(cb, C.c_void_p).value) == 0
    # header vs ctypes prototypes: same argument count, float / double exactly where the header says
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(repo, "include", "stem_hip.h")).read(), flags=re.S)
    protos = dict(re.findall(r"\bint\s+(stem_[a-z0-9_]+)\s*\(([^)]*)\)", src))
    for name in listed:
        params = [p.strip() for p in protos[name].split(",")] if protos[name].strip() not in ("", "void") else []
        sig = _lib._HIP_SIG[name]
        assert len(params) == len(sig), (name, params, sig)
        for p, ty in zip(params, sig):
            want = C.c_float if re.match(r"(const\s+)?float\s+\w+$", p) else (C.c_double if re.match(r"(const\s+)?double\s+\w+$", p) else None)
            if want is not None or ty in (C.c_float, C.c_double):
                assert ty is want, (name, p, ty)
            assert "[" not in p and not re.match(r"(const\s+)?struct\s+\w+\s+\w+$", p), (name, p)   # no aggregates by value


def test_engine_switches_follow_the_runtime_configuration(monkeypatch):
    """config.override(...) and changed STEM_* variables reach StemEngine's switches (they are descriptors over
    config.StemRuntimeConfig, not import-time copies); a value assigned on the class pins a switch."""
    from spatiotemporalentropymodel_amd import config
    from spatiotemporalentropymodel_amd.engine import StemEngine
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    config.runtime()                                              # an earlier test may have changed and restored the environment
    assert StemEngine.use_fx3 is True and StemEngine.overlap_wgrad is True
    with config.override(engine_f16x3=False, engine_overlap=False):
        assert StemEngine.use_fx3 is False and StemEngine.overlap_wgrad is False
        eng = SpatioTemporalPriorModel_Res(64, 96).engine()
        assert not any(l.fx3 for l in eng.layers)                 # routes chosen inside the block: fp32 MFMA everywhere
    assert StemEngine.use_fx3 is True
    eng = SpatioTemporalPriorModel_Res(64, 96).engine()
    assert any(l.fx3 for l in eng.layers) and eng.overlap_wgrad is True
    monkeypatch.setenv("STEM_ENGINE_F16X3", "0")
    eng2 = SpatioTemporalPriorModel_Res(64, 96).engine()          # building an engine parses the environment again
    assert eng2.use_fx3 is False and not any(l.fx3 for l in eng2.layers)
    monkeypatch.delenv("STEM_ENGINE_F16X3")
    config.runtime()
    monkeypatch.setattr(StemEngine, "use_wg3", False)             # pinned on the class
    with config.override(engine_wgrad_f16x3=True):
        assert StemEngine.use_wg3 is False


def test_rans_encoder_divides_like_the_oracle():
    """The host encoder replaces `x / freq` on its serial path by a multiplication with a tabulated reciprocal (csrc/rans_host.cpp:
    exact for every state below 2^63).  Random tables -- including frequencies of 1 and of almost 2^16 next to each other --, ragged
    lengths, escapes: the bytes equal those of the oracle's encoder, which divides (oracle/stem_oracle.c), and decode back."""
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc
    rng = np.random.default_rng(11)
    for trial in range(40):
        ncdf, stride = int(rng.integers(1, 6)), int(rng.integers(4, 48))
        cdfs = np.zeros((ncdf, stride), dtype=np.int32)
        sizes = np.zeros(ncdf, dtype=np.int32)
        offsets = rng.integers(-5, 3, ncdf).astype(np.int32)
        for c in range(ncdf):
            ln = int(rng.integers(3, stride + 1))
            sizes[c] = ln
            if trial % 3 == 0:
                w = np.ones(ln - 1)
                w[int(rng.integers(0, ln - 1))] = 1e9
            else:
                w = rng.random(ln - 1) ** int(rng.integers(1, 6)) + 1e-12
            f = np.maximum(1, np.floor(w / w.sum() * (65536 - (ln - 1)))).astype(np.int64)
            f[np.argmax(f)] += 65536 - f.sum()
            cdfs[c, 1:ln] = np.cumsum(f)
        n = int(rng.integers(1, 3000))
        idx = rng.integers(0, ncdf, n).astype(np.int32)
        sym = np.array([int(rng.integers(-3, sizes[i] + 2)) + int(offsets[i]) for i in idx], dtype=np.int32)
        if trial % 5 == 0:
            sym[rng.integers(0, n, 3)] = rng.integers(-70000, 70000, 3)
        ours = em.RansEncoder().encode_with_indexes(sym, idx, cdfs, sizes, offsets)
        assert ours == orc.rans_encode(sym, idx, cdfs, sizes, offsets), trial
        np.testing.assert_array_equal(np.asarray(em.RansDecoder().decode_with_indexes(ours, idx, cdfs, sizes, offsets)), sym)


def test_rans_decoder_lookup_search_equals_binary_search():
    """The host decoder finds a symbol through a per-row lookup table + a short walk once it has seen 2048 symbols of a stream
    (csrc/rans_host.cpp), with a binary search before that and for rows that are not strictly increasing 0 .. 2^16 sequences.  Long
    random streams (the switch happens in the middle of a stream decoded in pieces of 192 symbols, as the raster-order loop does),
    narrow and wide rows, escapes: the symbols equal the encoder's input and the oracle's decoder output."""
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stem_oracle as orc
    rng = np.random.default_rng(5)
    for trial in range(6):
        ncdf, stride = int(rng.integers(2, 9)), int(rng.choice([8, 64, 700, 3133]))
        cdfs = np.zeros((ncdf, stride), dtype=np.int32)
        sizes = np.zeros(ncdf, dtype=np.int32)
        offsets = rng.integers(-40, 3, ncdf).astype(np.int32)
        for c in range(ncdf):
            ln = int(rng.integers(3, stride + 1))
            sizes[c] = ln
            w = np.exp(-0.5 * ((np.arange(ln - 1) - (ln - 1) / 2) / max(1.0, (ln - 1) / rng.uniform(3, 30))) ** 2) + 1e-9
            f = np.maximum(1, np.floor(w / w.sum() * (65536 - (ln - 1)))).astype(np.int64)
            f[np.argmax(f)] += 65536 - f.sum()
            cdfs[c, 1:ln] = np.cumsum(f)
        n = 192 * int(rng.integers(20, 60))
        idx = rng.integers(0, ncdf, n).astype(np.int32)
        sym = np.array([int(rng.integers(-2, sizes[i])) + int(offsets[i]) for i in idx], dtype=np.int32)
        stream = em.RansEncoder().encode_with_indexes(sym, idx, cdfs, sizes, offsets)
        np.testing.assert_array_equal(np.asarray(orc.rans_decode(stream, idx, cdfs, sizes, offsets)), sym)
        np.testing.assert_array_equal(em.RansDecoder().decode_with_indexes_np(stream, idx, cdfs, sizes, offsets), sym)
        dec = em.RansDecoder()
        dec.set_stream(stream)
        t = em._as_tables(cdfs, sizes, offsets)
        got = np.concatenate([np.asarray(dec.decode_stream(idx[i:i + 192], t)) for i in range(0, n, 192)])
        np.testing.assert_array_equal(got, sym)


def test_rans_decoder_survives_tables_rewritten_behind_the_same_pointers():
    """ADVICE r5 (rans_host.cpp:250): the decoder's lookup table is keyed on the ADDRESSES of the CDF tables.  Tables updated in
    place inside one stream (an entropy model's `update()` writing into the same buffers) leave it stale: the walk from a stale
    start must stay inside the row (it runs under AddressSanitizer in test_host_codec_under_sanitizers: name contains "rans") and
    the symbols must still be the ones the CURRENT tables define -- the stale table is dropped and the binary search answers."""
    rng = np.random.default_rng(11)
    ncdf, stride = 4, 600

    def tables(spread):
        cdfs = np.zeros((ncdf, stride), dtype=np.int32)
        sizes = np.zeros(ncdf, dtype=np.int32)
        for c in range(ncdf):
            ln = int(rng.integers(40, stride + 1))
            sizes[c] = ln
            w = np.exp(-0.5 * ((np.arange(ln - 1) - (ln - 1) * rng.uniform(0.2, 0.8)) / spread) ** 2) + 1e-9
            f = np.maximum(1, np.floor(w / w.sum() * (65536 - (ln - 1)))).astype(np.int64)
            f[np.argmax(f)] += 65536 - f.sum()
            cdfs[c, 1:ln] = np.cumsum(f)
        return cdfs, sizes

    offsets = np.zeros(ncdf, dtype=np.int32)
    cdfs_a, sizes_a = tables(4.0)
    cdfs_b, sizes_b = tables(60.0)
    n = 192 * 30
    idx = rng.integers(0, ncdf, n).astype(np.int32)
    half = n // 2
    sym = np.empty(n, dtype=np.int32)
    sym[:half] = [int(rng.integers(0, sizes_a[i] - 1)) for i in idx[:half]]
    sym[half:] = [int(rng.integers(0, sizes_b[i] - 1)) for i in idx[half:]]
    enc = em.BufferedRansEncoder()                                   # one stream, two table sets
    enc.encode_with_indexes(sym[:half], idx[:half], cdfs_a, sizes_a, offsets)
    enc.encode_with_indexes(sym[half:], idx[half:], cdfs_b, sizes_b, offsets)
    stream = enc.flush()
    live_c, live_s = cdfs_a.copy(), sizes_a.copy()                   # the buffers the decoder sees: same addresses throughout
    t = em._as_tables(live_c, live_s, offsets)
    dec = em.RansDecoder()
    dec.set_stream(stream)
    got = [np.asarray(dec.decode_stream(idx[i:i + 192], t)) for i in range(0, half, 192)]            # builds the lookup table (> 2048 symbols)
    live_c[...] = cdfs_b                                             # rewritten IN PLACE
    live_s[...] = sizes_b
    got += [np.asarray(dec.decode_stream(idx[i:i + 192], t)) for i in range(half, n, 192)]
    np.testing.assert_array_equal(np.concatenate(got), sym)


def test_ms_ssim_against_an_independent_formulation():
    """evaluation.ms_ssim (the "ms-ssim" entry of stem/evalSTEM.py:81,147; the script takes it from the third-party pytorch_msssim,
    absent from the reference tree and from this image: no golden vector, see the module's header) against the same published
    algorithm written independently with scipy's 1-D correlations in float64, plus its defining properties."""
    import torch
    from scipy.ndimage import correlate1d
    from spatiotemporalentropymodel_amd.evaluation import ms_ssim
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:200, 0:232]
    img = np.stack([0.5 + 0.4 * np.sin(xx / (9.0 + c)) * np.cos(yy / (13.0 - c)) for c in range(3)])[None]
    noisy = np.clip(img + 0.05 * rng.standard_normal(img.shape), 0, 1)
    blurred = np.clip(0.5 * img + 0.5 * np.roll(img, 3, axis=3), 0, 1)

    def reference(a, b):
        c = np.arange(11) - 5
        g = np.exp(-(c ** 2) / (2 * 1.5 ** 2))
        g /= g.sum()

        def filt(v):                                    # 'valid' separable correlation along H and W
            v = correlate1d(correlate1d(v, g, axis=2, mode="constant"), g, axis=3, mode="constant")
            return v[:, :, 5:-5, 5:-5]

        def pool(v):
            ph, pw = v.shape[2] % 2, v.shape[3] % 2
            v = np.pad(v, ((0, 0), (0, 0), (ph, ph), (pw, pw)))            # avg_pool2d(padding=p) pads both sides with zeros, counted
            H2, W2 = v.shape[2] // 2, v.shape[3] // 2
            return v[:, :, :2 * H2, :2 * W2].reshape(v.shape[0], v.shape[1], H2, 2, W2, 2).mean(axis=(3, 5))

        C1, C2 = 0.01 ** 2, 0.03 ** 2
        w = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]
        out = np.ones(a.shape[:2])
        for lv in range(5):
            m1, m2 = filt(a), filt(b)
            s11, s22, s12 = filt(a * a) - m1 * m1, filt(b * b) - m2 * m2, filt(a * b) - m1 * m2
            cs = (2 * s12 + C2) / (s11 + s22 + C2)
            ss = (2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1) * cs
            t = cs.mean(axis=(2, 3)) if lv < 4 else ss.mean(axis=(2, 3))
            out *= np.maximum(t, 0) ** w[lv]
            if lv < 4:
                a, b = pool(a), pool(b)
        return float(out.mean())

    for other in (noisy, blurred):
        got = ms_ssim(torch.from_numpy(img).float(), torch.from_numpy(other).float())
        assert abs(got - reference(img.astype(np.float64), other.astype(np.float64))) < 2e-5, (got, reference(img, other))
    t = torch.from_numpy(img).float()
    assert abs(ms_ssim(t, t) - 1.0) < 1e-6
    a, b = ms_ssim(t, torch.from_numpy(noisy).float()), ms_ssim(torch.from_numpy(noisy).float(), t)
    assert abs(a - b) < 1e-6 and 0 < a < 1
    worse = np.clip(img + 0.15 * rng.standard_normal(img.shape), 0, 1)
    assert ms_ssim(t, torch.from_numpy(worse).float()) < a
    assert ms_ssim(t[:, :, :160, :160], t[:, :, :160, :160]) is None          # no fifth scale: reported as absent, not as a number
