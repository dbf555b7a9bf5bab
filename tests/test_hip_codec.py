"""GPU: compress()/decompress() of the STEM models (GPU probability model + host rANS) against the
reference's bitstreams (tests/golden/stem_codec_small.npz) and as encode->decode round trips."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().contiguous().numpy()


def _model(cls, dev):
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    m = cls(64, 96)
    closed_form_fill_(m)
    m = m.to(dev).eval()
    assert m.update(force=True) is True
    return m


@pytest.mark.parametrize("tag", ["res", "full"])
def test_bitstreams_match_reference(golden, tag):
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel, SpatioTemporalPriorModel_Res
    g = golden("stem_codec_small.npz")
    dev = torch.device("cuda:0")
    m = _model({"res": SpatioTemporalPriorModel_Res, "full": SpatioTemporalPriorModel}[tag], dev)
    # update() runs torch CPU transcendental kernels whose last ulp depends on the host CPU (the reference has the
    # same property); tables agree up to one count in a handful of entries across machines ...
    np.testing.assert_array_equal(host(m.entropy_bottleneck._offset), g[f"{tag}:eb_offset"])
    np.testing.assert_array_equal(host(m.entropy_bottleneck._cdf_length), g[f"{tag}:eb_cdf_length"])
    for ours, ref in ((host(m.entropy_bottleneck._quantized_cdf), g[f"{tag}:eb_cdf"]),
                      (host(m.gaussian_conditional._quantized_cdf), g["gc_cdf"])):
        diff = np.abs(ours.astype(np.int64) - ref)
        assert diff.max() <= 1 and (diff != 0).mean() < 5e-3
    # ... and, as with any checkpoint, the coder uses the tables stored in the state_dict
    sd = m.state_dict()
    for k, v in (("entropy_bottleneck._quantized_cdf", g[f"{tag}:eb_cdf"]), ("gaussian_conditional._quantized_cdf", g["gc_cdf"]),
                 ("gaussian_conditional._offset", g["gc_offset"]), ("gaussian_conditional._cdf_length", g["gc_cdf_length"])):
        sd[k] = torch.from_numpy(v).to(dev)
    m.load_state_dict(sd)
    y_cur, y_cond = torch.from_numpy(g["y_cur"]).to(dev), torch.from_numpy(g["y_cond"]).to(dev)
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        assert tuple(enc["shape"]) == tuple(g[f"{tag}:shape"])
        assert enc["strings"][1][0] == g[f"{tag}:z_string"].tobytes(), "hyper-latent bitstream differs"
        # symbols/indexes are integer decisions on fp32 quantities: identical unless a value sits within fp32
        # noise of a rounding / table threshold (none does in this fixture)
        assert enc["strings"][0][0] == g[f"{tag}:y_string"].tobytes(), "latent bitstream differs from the reference's"
        dec = m.decompress([[g[f"{tag}:y_string"].tobytes()], [g[f"{tag}:z_string"].tobytes()]], enc["shape"], y_cond)
        y_hat = dec["y_hat"] if tag == "res" else dec
        assert isinstance(dec, dict) == (tag == "res")          # upstream's inconsistent return types are kept
        assert_close(host(y_hat), g[f"{tag}:y_hat"], what="decoded y_hat vs reference", floor=0.1)
        fwd = m(y_cur, y_cond)
        assert_close(host(fwd["y_hat"]), g[f"{tag}:fwd_y_hat"], what="forward y_hat", floor=0.1)


@pytest.mark.parametrize("cls_name", ["SpatioTemporalPriorModelWithoutSPMTPM", "SpatioTemporalPriorModelWithoutSPM",
                                      "SpatioTemporalPriorModelWithoutTPM", "SpatioTemporalPriorModel_Res"])
def test_roundtrip_all_variants(cls_name):
    """encode -> decode reproduces the encoder's reconstruction for every model variant, batch of 2, ragged size."""
    import spatiotemporalentropymodel_amd.models as M
    dev = torch.device("cuda:0")
    cls = getattr(M, cls_name)
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    m = cls(256, 96) if "WithoutSPM" in cls_name else cls(64, 96)      # the SPM-less ablations hard-code 256 hyper channels
    closed_form_fill_(m)
    m = m.to(dev).eval()
    m.update(force=True)
    y_cur = (closed_form_input("rt:y", (2, 96, 8, 12), -6, 6)).to(dev)
    y_cond = (closed_form_input("rt:c", (2, 96, 8, 12), -6, 6)).to(dev)
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        assert len(enc["strings"][0]) == 2 and len(enc["strings"][1]) == 2
        dec = m.decompress(enc["strings"], enc["shape"], y_cond)
        y_hat = dec["y_hat"] if isinstance(dec, dict) else dec
        fwd = m(y_cur, y_cond)
    if m.HAS_SPM:
        # AR models: the decoder must land on exactly what the encoder wrote back (q + mean, position by position)
        res = host(y_hat) - (host(y_cond) if m.RESIDUAL else 0)
        assert np.isfinite(res).all()
        enc2 = m.compress(y_hat if not m.RESIDUAL else y_hat, y_cond)      # re-encoding the reconstruction is idempotent in size
        assert abs(len(enc2["strings"][0][0]) - len(enc["strings"][0][0])) <= max(8, len(enc["strings"][0][0]) // 10)
    else:
        # (the decoder's per-position products and the forward's layer kernels are different fp32-class routes: same symbols, the
        # means agree to the routes' rounding, ~1e-6 of the largest latent)
        assert_close(host(y_hat), host(fwd["y_hat"]), 2e-5, what="decode == eval forward reconstruction", floor=0.1)
    total_bits = 8 * sum(len(s) for s in enc["strings"][0] + enc["strings"][1])
    assert total_bits > 0


@pytest.mark.parametrize("cls_name", ["SpatioTemporalPriorModelWithoutTPM", "SpatioTemporalPriorModel_Res"])
def test_fused_decode_loop_equals_stepwise_entry_points(cls_name, monkeypatch):
    """stem_ar_decode_image (the raster loop inside the library, host decoder injected as a C pointer) against the same loop
    driven from Python through the single-step entry points (stem_gemv3_decode / stem_gemv3 / stem_ar_finish_decode)."""
    import spatiotemporalentropymodel_amd.models as M
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    dev = torch.device("cuda:0")
    m = closed_form_fill_(getattr(M, cls_name)(64, 96)).to(dev).eval()
    m.update(force=True)
    y_cur = closed_form_input("fd:y", (2, 96, 4, 12), -6, 6).to(dev)
    y_cond = closed_form_input("fd:c", (2, 96, 4, 12), -6, 6).to(dev)
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        fused = m.decompress(enc["strings"], enc["shape"], y_cond)
        monkeypatch.setenv("STEM_AR_STEPWISE", "1")
        step = m.decompress(enc["strings"], enc["shape"], y_cond)
        enc_step = m.compress(y_cur, y_cond)                      # encoder: per-step entry points vs stem_ar_encode_image
        assert enc_step["strings"] == enc["strings"]
    a = fused["y_hat"] if isinstance(fused, dict) else fused
    b = step["y_hat"] if isinstance(step, dict) else step
    np.testing.assert_array_equal(host(a), host(b))
    with pytest.raises(ValueError):          # 5x9 latents do not survive the hyper stages: loud error, not a corrupt stream
        m.compress(y_cur[:, :, :, :9].contiguous()[:, :, :3], y_cond[:, :, :, :9].contiguous()[:, :, :3])


def test_lockstep_batch_decode_equals_per_image_decode(monkeypatch):
    """decompress() of a batch: by default one persistent decoder per image, all five at once on an XCD each
    (codec._decode_concurrently); STEM_AR_CONCURRENT=0 / STEM_AR_FORCE_BATCH=1: the images advance in lockstep
    (stem_ar_decode_batch); STEM_AR_NO_BATCH=1: one image after the other.  Per image every route must reproduce what the
    per-position loop (stem_ar_decode_image, the reference's order) decodes, bit for bit -- 5 images, non-square."""
    import spatiotemporalentropymodel_amd.models as M
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    dev = torch.device("cuda:0")
    m = closed_form_fill_(M.SpatioTemporalPriorModel_Res(64, 96)).to(dev).eval()
    m.update(force=True)
    y_cur = closed_form_input("lb:y", (5, 96, 8, 12), -6, 6).to(dev)
    y_cond = closed_form_input("lb:c", (5, 96, 8, 12), -6, 6).to(dev)
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")                 # a kernel that gives up announces its fallback with a warning
            a = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        monkeypatch.setenv("STEM_AR_CONCURRENT", "0")
        a0 = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()        # lockstep
        monkeypatch.delenv("STEM_AR_CONCURRENT")
        monkeypatch.setenv("STEM_AR_PERSISTENT", "0")
        monkeypatch.setenv("STEM_AR_NO_BATCH", "1")
        ref = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()       # the per-position loop, image by image
        monkeypatch.delenv("STEM_AR_PERSISTENT")
        monkeypatch.delenv("STEM_AR_NO_BATCH")
        assert torch.equal(a, ref) and torch.equal(a0, ref)
        monkeypatch.setenv("STEM_AR_FORCE_BATCH", "1")     # the lockstep loop also for one image
        monkeypatch.setenv("STEM_AR_PIPELINE", "1")        # flag-polling variant: no stream synchronisation per position
        c = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        c1 = m.decompress([enc["strings"][0][:1], enc["strings"][1][:1]], enc["shape"], y_cond[:1])["y_hat"].clone()
        monkeypatch.delenv("STEM_AR_PIPELINE")
        c1s = m.decompress([enc["strings"][0][:1], enc["strings"][1][:1]], enc["shape"], y_cond[:1])["y_hat"].clone()
        monkeypatch.delenv("STEM_AR_FORCE_BATCH")
        monkeypatch.setenv("STEM_AR_NO_BATCH", "1")
        b = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        # and one image on its own equals its slot in the batch
        one = m.decompress([enc["strings"][0][3:4], enc["strings"][1][3:4]], enc["shape"], y_cond[3:4])["y_hat"]
    assert torch.equal(a, b)
    assert torch.equal(a, c) and torch.equal(c1, c1s)
    # (decoded alone, the hyper-prior convolutions run at batch 1 and pick another tile / split-K plan: the means move by
    # an ulp, the symbols do not)
    assert float((a[3:4] - one).abs().max()) <= 1e-4
    # within half a quantisation step of the input everywhere (y_hat = round(res - mu) + mu + y_cond)
    assert float((a - y_cur).abs().max()) <= 0.5 + 1e-4


@pytest.mark.parametrize("cls_name,widths,hw", [("SpatioTemporalPriorModelWithoutTPM", (64, 96), (8, 20)), ("SpatioTemporalPriorModel_Res", (64, 96), (8, 20)),
                                                ("SpatioTemporalPriorModel_Res", (256, 192), (4, 12)), ("SpatioTemporalPriorModel_Res", (64, 96), (12, 4))])
def test_persistent_decoder_equals_per_position_loop(cls_name, widths, hw, monkeypatch):
    """csrc/ar_persistent.hip, the default single-image decoder (one launch per image: 32 resident workgroups with the weights of
    their output rows in registers, tagged 8-byte words from product to product, the next position's known part accumulated
    while the host decodes, host symbol decoder behind a pinned mailbox) against stem_ar_decode_image (four launches + a
    synchronisation per position, STEM_AR_PERSISTENT=0) and the Python-driven single-step loop: identical reconstructions, bit for
    bit, no fallback taken, and the stream decodes to within half a quantisation step of the input.  Small and full-width model
    (M = 96: two 256-column steps per window row; M = 192: four, the last one partial), with and without the temporal prior, and the
    narrowest image the hyper path admits (4 columns: the window hangs over both borders at once).  spatiotemporalpriors.py:1015-1054."""
    import warnings
    import spatiotemporalentropymodel_amd.models as M
    from spatiotemporalentropymodel_amd import config
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    dev = torch.device("cuda:0")
    m = closed_form_fill_(getattr(M, cls_name)(*widths)).to(dev).eval()
    m.update(force=True)
    y_cur = closed_form_input("pd:y", (1, widths[1], *hw), -6, 6).to(dev)
    y_cond = closed_form_input("pd:c", (1, widths[1], *hw), -6, 6).to(dev)

    def y_hat(res):
        return (res["y_hat"] if isinstance(res, dict) else res).clone()

    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        monkeypatch.delenv("STEM_AR_PERSISTENT", raising=False)
        assert config.runtime().ar_persistent                                   # the default route
        with warnings.catch_warnings():
            warnings.simplefilter("error")                                      # the fallback announces itself with a warning
            a = y_hat(m.decompress(enc["strings"], enc["shape"], y_cond))
            a2 = y_hat(m.decompress(enc["strings"], enc["shape"], y_cond))      # the library's flags / mailboxes are reusable
        monkeypatch.setenv("STEM_AR_PERSISTENT", "0")
        b = y_hat(m.decompress(enc["strings"], enc["shape"], y_cond))
        monkeypatch.setenv("STEM_AR_STEPWISE", "1")
        c = y_hat(m.decompress(enc["strings"], enc["shape"], y_cond))
    assert torch.equal(a, b) and torch.equal(a, a2) and torch.equal(b, c)
    assert float((a - y_cur).abs().max()) <= 0.5 + 1e-4


def test_iframe_codec_matches_reference(golden):
    """JointAutoregressiveHierarchicalPriors ("mbt2018") as the evaluation loop's I-frame codec (stem/evalSTEM.py:54-59;
    compressai/models/priors.py:476-716): compress() reproduces the reference's two bitstreams byte for byte, decompress() of the
    reference's strings its latents and image, forward() its inference outputs -- small model, closed-form weights with the last
    analysis layer scaled x4 and the first synthesis layer x1/4 as in tests/golden/make_golden.py:gen_iframe_codec.  Both decoder routes (persistent kernel, loop)."""
    import os
    from spatiotemporalentropymodel_amd import config
    from spatiotemporalentropymodel_amd.models.priors import JointAutoregressiveHierarchicalPriors
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    g = golden("iframe_codec_small.npz")
    dev = torch.device("cuda:0")
    m = closed_form_fill_(JointAutoregressiveHierarchicalPriors(64, 96))
    with torch.no_grad():
        m.g_a[6].weight.mul_(4.0)
        m.g_a[6].bias.mul_(4.0)
        m.g_s[0].weight.mul_(0.25)
    m = m.to(dev).eval()
    assert m.update(force=True) is True
    # as with any checkpoint, the coder uses the tables stored in the state_dict (update() runs host transcendental kernels whose
    # last ulp depends on the CPU: test_bitstreams_match_reference)
    for ours, ref in ((host(m.entropy_bottleneck._quantized_cdf), g["eb_cdf"]), (host(m.gaussian_conditional._quantized_cdf), g["gc_cdf"])):
        diff = np.abs(ours.astype(np.int64) - ref)
        assert diff.max() <= 1 and (diff != 0).mean() < 5e-3
    sd = m.state_dict()
    for k, v in (("entropy_bottleneck._quantized_cdf", g["eb_cdf"]), ("entropy_bottleneck._offset", g["eb_offset"]),
                 ("entropy_bottleneck._cdf_length", g["eb_cdf_length"]), ("gaussian_conditional._quantized_cdf", g["gc_cdf"]),
                 ("gaussian_conditional._offset", g["gc_offset"]), ("gaussian_conditional._cdf_length", g["gc_cdf_length"])):
        sd[k] = torch.from_numpy(v).to(dev)
    m.load_state_dict(sd)
    x = torch.from_numpy(g["x"]).to(dev)
    with torch.no_grad():
        enc = m.compress(x)
        assert tuple(enc["shape"]) == tuple(g["shape"])
        assert enc["strings"][1][0] == g["z_string"].tobytes(), "hyper-latent bitstream differs"
        assert enc["strings"][0][0] == g["y_string"].tobytes(), "latent bitstream differs from the reference's"
        ref_strings = [[g["y_string"].tobytes()], [g["z_string"].tobytes()]]
        outs = []
        for persistent in ("1", "0"):
            os.environ["STEM_AR_PERSISTENT"] = persistent
            try:
                assert config.runtime().ar_persistent == (persistent == "1")
                outs.append(m.decompress(ref_strings, enc["shape"]))
            finally:
                os.environ.pop("STEM_AR_PERSISTENT")
        assert torch.equal(outs[0]["y_hat"], outs[1]["y_hat"]) and torch.equal(outs[0]["x_hat"], outs[1]["x_hat"])
        dec = outs[0]
        assert_close(host(dec["y_hat"]), g["y_hat"], what="decoded y_hat vs reference", floor=0.1)
        # the image is the synthesis transform's output clamped to [0, 1]: with these weights it spans +-|x|max before the clamp, and a
        # pixel next to a clamp edge carries the UNclamped tensor's rounding error -- 1e-4 of that tensor's scale
        pre = float(np.abs(g["fwd:x_hat"]).max())
        assert_close(host(dec["x_hat"]), g["x_hat"], atol=1e-4 * max(pre, 1.0), what="decoded image vs reference", floor=0.1)
        fwd = m(x)
    assert_close(host(fwd["y"]), g["fwd:y"], what="forward y", floor=0.1)
    assert_close(host(fwd["y_hat"]), g["fwd:y_hat"], what="forward y_hat", floor=0.1)
    assert_close(host(fwd["x_hat"]), g["fwd:x_hat"], what="forward x_hat", floor=0.1)
    assert_close(host(fwd["entropy_params"]["scales_hat"]), g["fwd:scales"], what="scales", floor=0.1)
    assert_close(host(fwd["entropy_params"]["means_hat"]), g["fwd:means"], what="means", floor=0.1)
    assert_close(host(fwd["likelihoods"]["z"]), g["fwd:lik_z"], what="lik_z", floor=0.1, atol=1e-9)
    assert_close(host(fwd["likelihoods"]["y"]), g["fwd:lik_y"], rtol=2e-4, what="lik_y", floor=0.1, atol=1e-9)


@pytest.mark.parametrize("batch,giveup_at", [(1, 1), (1, 57), (1, 160), (3, 40)])
def test_persistent_decoder_give_up_falls_back_to_the_loop(batch, giveup_at, monkeypatch):
    """The persistent decoder's bounded waits may run out (codec.py: "persistent decoder gave up"): the host raises the abort word,
    the kernel's workgroups leave their polls, the call returns an error with the latent buffer half written -- and the image is
    then decoded again by the per-position loop from a FRESH buffer and a FRESH rANS decoder.  Forced here at the first position,
    in the middle of the image and at its last position (stem_tuning_set("arp_giveup_at", k): the host gives up at position k - 1
    exactly as a timed-out wait does), for a single image and for a batch decoded concurrently (each image's own kernel gives up):
    the warning fires, the reconstruction equals the loop's bit for bit, and the next call -- knob cleared -- runs persistently
    again without a warning (the library's per-thread flags and mailboxes survive an aborted image).
    spatiotemporalpriors.py:1015-1054."""
    import warnings
    import spatiotemporalentropymodel_amd.models as M
    from spatiotemporalentropymodel_amd import _lib
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    dev = torch.device("cuda:0")
    m = closed_form_fill_(M.SpatioTemporalPriorModel_Res(64, 96)).to(dev).eval()
    m.update(force=True)
    y_cur = closed_form_input("gu:y", (batch, 96, 8, 20), -6, 6).to(dev)
    y_cond = closed_form_input("gu:c", (batch, 96, 8, 20), -6, 6).to(dev)
    lib = _lib.hip()
    with torch.no_grad():
        enc = m.compress(y_cur, y_cond)
        monkeypatch.setenv("STEM_AR_PERSISTENT", "0")
        monkeypatch.setenv("STEM_AR_NO_BATCH", "1")
        ref = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()          # the loop
        monkeypatch.delenv("STEM_AR_PERSISTENT")
        monkeypatch.delenv("STEM_AR_NO_BATCH")
        m.decompress(enc["strings"], enc["shape"], y_cond)             # (the process's one-time self-check of the persistent route runs undisturbed)
        assert lib.stem_tuning_set(b"arp_giveup_at", giveup_at) == 0
        try:
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                got = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        finally:
            assert lib.stem_tuning_set(b"arp_giveup_at", 0) == 0
        gave_up = [w for w in seen if "persistent decoder gave up" in str(w.message)]
        assert len(gave_up) >= batch, [str(w.message) for w in seen]      # a batch retries each image alone before it takes the loop
        assert f"position {giveup_at - 1}" in str(gave_up[0].message)
        assert torch.equal(got, ref)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            again = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        assert torch.equal(again, ref)


def _eval_gop_models(g, dev):
    """the two models of tests/golden/make_golden.py:eval_gop_models, with the reference's CDF tables loaded as a checkpoint's
    state_dict would bring them (update() runs host transcendental kernels whose last ulp depends on the CPU)"""
    from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res
    from spatiotemporalentropymodel_amd.models.priors import JointAutoregressiveHierarchicalPriors
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_
    imodel = closed_form_fill_(JointAutoregressiveHierarchicalPriors(64, 96))
    with torch.no_grad():
        imodel.g_a[6].weight.mul_(4.0)
        imodel.g_a[6].bias.mul_(4.0)
        imodel.g_s[0].weight.mul_(0.25)
    stem = closed_form_fill_(SpatioTemporalPriorModel_Res(64, 96))
    out = []
    for tag, m in (("i", imodel), ("p", stem)):
        m = m.to(dev).eval()
        assert m.update(force=True) is True
        for ours, ref in ((host(m.entropy_bottleneck._quantized_cdf), g[f"{tag}:eb_cdf"]), (host(m.gaussian_conditional._quantized_cdf), g[f"{tag}:gc_cdf"])):
            diff = np.abs(ours.astype(np.int64) - ref)
            assert diff.max() <= 1 and (diff != 0).mean() < 5e-3
        sd = m.state_dict()
        for k, v in (("entropy_bottleneck._quantized_cdf", g[f"{tag}:eb_cdf"]), ("entropy_bottleneck._offset", g[f"{tag}:eb_offset"]),
                     ("entropy_bottleneck._cdf_length", g[f"{tag}:eb_cdf_length"]), ("gaussian_conditional._quantized_cdf", g[f"{tag}:gc_cdf"]),
                     ("gaussian_conditional._offset", g[f"{tag}:gc_offset"]), ("gaussian_conditional._cdf_length", g[f"{tag}:gc_cdf_length"])):
            sd[k] = torch.from_numpy(v).to(dev)
        m.load_state_dict(sd)
        out.append(m)
    return out


@pytest.mark.parametrize("persistent", ["1", "0"])
def test_eval_gop_chain_matches_reference(golden, persistent, monkeypatch):
    """BASELINE configs[3] end to end, as a CHAIN: evaluation.eval_gop (I frame through mbt2018's compress / decompress, two P frames
    through getY -> forward -> compress -> decompress -> getX, each conditioned on the previous frame's DECODED latents; pad to 64,
    crop) against what the reference's own inferenceI_DVR / inferenceP_DVR (stem/evalSTEM.py:34-153) produced on the same 120 x 104
    frames (tests/golden/eval_gop.npz): every string byte for byte -- so the actual bpp is EQUAL --, the rate estimate and the PSNR
    within 1e-4 relative, the conditioning latents and the last reconstruction within 1e-4.  Both decoder routes."""
    from spatiotemporalentropymodel_amd import config, evaluation
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    g = golden("eval_gop.npz")
    dev = torch.device("cuda:0")
    imodel, stem = _eval_gop_models(g, dev)
    h, w = (int(v) for v in g["size"])
    n = int(g["nframes"][0])
    frames = [f[0, :, 4:4 + h, 12:12 + w].contiguous().to(dev) for f in smooth_frames("evalgop", 1, n, 128)]
    monkeypatch.setenv("STEM_AR_PERSISTENT", persistent)
    assert config.runtime().ar_persistent == (persistent == "1")
    res = evaluation.eval_gop(imodel, stem, frames, gop=12)
    assert [f["type"] for f in res["frames"]] == ["I"] + ["P"] * (n - 1)
    for t, f in enumerate(res["frames"]):
        assert tuple(f["shape"]) == tuple(g[f"f{t}:shape"])
        assert f["strings"][1][0] == g[f"f{t}:z_string"].tobytes(), f"frame {t}: hyper-latent bitstream differs"
        assert f["strings"][0][0] == g[f"f{t}:y_string"].tobytes(), f"frame {t}: latent bitstream differs from the reference's"
        bpp, est, ps = g[f"f{t}:scalars"]
        assert f["bpp"] == bpp
        assert abs(f["estimate_bpp"] - est) <= 1e-4 * abs(est), (t, f["estimate_bpp"], est)
        assert abs(f["psnr"] - ps) <= 1e-4 * abs(ps), (t, f["psnr"], ps)
        assert_close(host(f["y_conditioned"]), g[f"f{t}:y_conditioned"], what=f"frame {t}: decoded latents", floor=0.1)
    assert_close(host(res["frames"][-1]["x_hat"]), g["last:x_hat"], atol=1e-4, what="last frame's reconstruction", floor=0.1)
    assert abs(res["bpp_ave"] - np.mean([g[f"f{t}:scalars"][0] for t in range(n)])) < 1e-12


def test_eval_gop_restarts_the_chain_at_every_i_frame(golden):
    """the frame loop of stem/evalSTEM.py:186-209: frame k (1-based) is an I frame when k % GOP == 1.  With GOP = 2 the five-frame
    sequence is I P I P I; every I frame's result does not depend on what came before it (same bytes as coding it alone), and a P
    frame's strings change with its conditioning latents."""
    from spatiotemporalentropymodel_amd import evaluation
    from spatiotemporalentropymodel_amd.weights import smooth_frames
    g = golden("eval_gop.npz")
    dev = torch.device("cuda:0")
    imodel, stem = _eval_gop_models(g, dev)
    frames = [f[0, :, :64, :64].contiguous().to(dev) for f in smooth_frames("evalgop2", 1, 5, 64)]
    res = evaluation.eval_gop(imodel, stem, frames, gop=2)
    assert [f["type"] for f in res["frames"]] == ["I", "P", "I", "P", "I"]
    alone = evaluation.inference_iframe(imodel, frames[2])
    assert alone["strings"] == res["frames"][2]["strings"] and alone["bpp"] == res["frames"][2]["bpp"]
    other = evaluation.inference_pframe(imodel, stem, frames[3], res["frames"][0]["y_conditioned"])
    assert other["strings"][0] != res["frames"][3]["strings"][0]
    allintra = evaluation.eval_gop(imodel, stem, frames[:2], gop=12, all_intra=True)
    assert [f["type"] for f in allintra["frames"]] == ["I", "I"]


def test_persistent_route_is_self_checked_once_per_process(monkeypatch):
    """codec._persistent_trusted: the first decode of a model geometry in a process encodes a 4 x 6 synthetic image with the model's
    own weights and decodes it by the persistent kernel AND by the per-position loop; the persistent kernel is used from then on
    only if the two agree bit for bit.  Here: a fresh geometry passes its check (and is not checked again); a check that sees the
    kernel give up (forced with the give-up knob) switches that geometry to the loop for the rest of the process, with a warning,
    and decoding stays correct."""
    import warnings
    import spatiotemporalentropymodel_amd.models as M
    from spatiotemporalentropymodel_amd import _lib, codec
    from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input
    dev = torch.device("cuda:0")
    lib = _lib.hip()
    saved = dict(codec._ARP_TRUSTED)
    try:
        codec._ARP_TRUSTED.clear()
        m = closed_form_fill_(M.SpatioTemporalPriorModel_Res(64, 96)).to(dev).eval()
        m.update(force=True)
        y_cur = closed_form_input("sc:y", (1, 96, 4, 8), -6, 6).to(dev)
        y_cond = closed_form_input("sc:c", (1, 96, 4, 8), -6, 6).to(dev)
        with torch.no_grad():
            enc = m.compress(y_cur, y_cond)
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                a = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
            assert list(codec._ARP_TRUSTED.values()) == [True]
            codec._ARP_TRUSTED.clear()
            assert lib.stem_tuning_set(b"arp_giveup_at", 3) == 0
            try:
                with warnings.catch_warnings(record=True) as seen:
                    warnings.simplefilter("always")
                    b = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
            finally:
                assert lib.stem_tuning_set(b"arp_giveup_at", 0) == 0
            assert list(codec._ARP_TRUSTED.values()) == [False]
            assert any("did not reproduce the per-position loop" in str(w.message) for w in seen)
            with warnings.catch_warnings():
                warnings.simplefilter("error")                  # the loop from here on: no persistent launch, nothing to give up
                c = m.decompress(enc["strings"], enc["shape"], y_cond)["y_hat"].clone()
        assert torch.equal(a, b) and torch.equal(a, c)
    finally:
        codec._ARP_TRUSTED.clear()
        codec._ARP_TRUSTED.update(saved)
