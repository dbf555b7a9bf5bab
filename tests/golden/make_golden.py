#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  Follows the import
recipe of SURVEY.md §8(c): copy the reference's `compressai` package to a
scratch directory (never into this repo), drop the Windows stubs, compile the
two pybind11 host extensions from the sources where they lie, stub the two
torchvision names the STEM modules import but never call, then import it and
run the reference code on closed-form weights/inputs
(`spatiotemporalentropymodel_amd.weights`).  Only inputs/outputs are saved.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import importlib.util
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("STEM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
from spatiotemporalentropymodel_amd.weights import (  # noqa: E402
    closed_form_fill_,
    closed_form_fill_spread_,
    closed_form_fill_scaled_,
    closed_form_input,
    smooth_frames,
)


def import_reference():
    scratch = tempfile.mkdtemp(prefix="stem_ref_")
    shutil.copytree(os.path.join(REF, "compressai"), os.path.join(scratch, "compressai"))
    pkg = os.path.join(scratch, "compressai")
    for f in ("ans.py", "_CXX.py"):
        os.remove(os.path.join(pkg, f))
    for f in os.listdir(pkg):
        if f.endswith(".pyd"):
            os.remove(os.path.join(pkg, f))
    open(os.path.join(pkg, "models", "gain.py"), "w").close()
    os.makedirs(os.path.join(scratch, "torchvision"))
    open(os.path.join(scratch, "torchvision", "__init__.py"), "w").close()
    with open(os.path.join(scratch, "torchvision", "utils.py"), "w") as f:
        f.write("def make_grid(*a, **k):\n    raise NotImplementedError\n"
                "def save_image(*a, **k):\n    raise NotImplementedError\n")
    os.makedirs(os.path.join(scratch, "torchvision", "transforms"))
    with open(os.path.join(scratch, "torchvision", "transforms", "__init__.py"), "w") as f:      # names compressai_examples/codec.py imports
        f.write("class ToPILImage:\n    pass\nclass ToTensor:\n    pass\n")
    with open(os.path.join(scratch, "torchvision", "transforms", "functional.py"), "w") as f:
        # stand-in for torchvision's to_tensor (uint8 HWC -> float CHW / 255), used ONLY by gen_roi_dataset, which records
        # integers decoded from the pixels (crop offsets, frame order) and the quality map -- nothing that depends on it
        f.write("import numpy as np, torch\n"
                "def to_tensor(pic):\n"
                "    return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)\n")
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = subprocess.check_output([sys.executable, "-m", "pybind11", "--includes"]).decode().split()
    base = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", *inc]
    subprocess.check_call(base + [f"-I{REF}/third_party/ryg_rans", f"-I{pkg}/cpp_exts/rans",
                                  f"{pkg}/cpp_exts/rans/rans_interface.cpp", "-o", f"{pkg}/ans{ext}"])
    subprocess.check_call(base + [f"{pkg}/cpp_exts/ops/ops.cpp", "-o", f"{pkg}/_CXX{ext}"])
    sys.path.insert(0, scratch)
    spec = importlib.util.spec_from_file_location("ref_root_utils", os.path.join(REF, "utils.py"))
    root_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(root_utils)
    return scratch, root_utils


class NoiseFeed:
    """Replaces EntropyModel._get_noise_cached (entropy_models.py:112-120) on one
    instance: the k-th draw of role R is closed_form_input(f"noise:{R}:{k}")."""

    def __init__(self, role, log):
        self.role, self.k, self.log = role, 0, log

    def __call__(self, x):
        name = f"noise:{self.role}:{self.k}"
        self.k += 1
        self.log.append((name, tuple(x.shape)))
        return closed_form_input(name, tuple(x.shape), -0.5, 0.5).to(x.dtype)


def t2n(t):
    return np.array(t.detach().cpu().numpy())        # a copy: views of .grad would alias later accumulation


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


# ---------------------------------------------------------------------------
def gen_ops(ref):
    import torch.nn as nn
    import torch.nn.functional as F
    from compressai.entropy_models import EntropyBottleneck, GaussianConditional
    from compressai.layers import GDN, MaskedConv2d
    from compressai.ops import LowerBound

    d = {}
    # --- convolutions (nn.Conv2d as built by models/utils.py:112-120 and spatiotemporalpriors.py:807-838)
    conv_cases = {
        "conv_k5s2": (2, 6, 11, 9, 8, 5, 2, 2),
        "conv_k5s1": (1, 5, 7, 8, 6, 5, 1, 2),
        "conv_k3s1": (2, 7, 6, 5, 9, 3, 1, 1),
        "conv_k1s1": (2, 12, 4, 5, 10, 1, 1, 0),
        "conv_c3": (1, 3, 16, 12, 8, 5, 2, 2),
    }
    for name, (N, C, H, W, K, R, st, pd) in conv_cases.items():
        m = nn.Conv2d(C, K, R, stride=st, padding=pd)
        closed_form_fill_wrapped(m, name)
        x = closed_form_input(name + ":x", (N, C, H, W), -1, 1).requires_grad_()
        y = m(x)
        dy = closed_form_input(name + ":dy", tuple(y.shape), -1, 1)
        y.backward(dy)
        d.update({f"{name}:cfg": np.array([N, C, H, W, K, R, st, pd]), f"{name}:x": t2n(x), f"{name}:w": t2n(m.weight),
                  f"{name}:b": t2n(m.bias), f"{name}:y": t2n(y), f"{name}:dy": t2n(dy), f"{name}:dx": t2n(x.grad),
                  f"{name}:dw": t2n(m.weight.grad), f"{name}:db": t2n(m.bias.grad)})
    # --- transposed convolution (models/utils.py:122-130; spatiotemporalpriors.py:821-826)
    for name, (N, C, H, W, K, R, st, pd, op) in {"deconv_k5s2": (2, 6, 5, 4, 8, 5, 2, 2, 1),
                                                 "deconv_c3": (1, 8, 6, 5, 3, 5, 2, 2, 1)}.items():
        m = nn.ConvTranspose2d(C, K, R, stride=st, padding=pd, output_padding=op)
        closed_form_fill_wrapped(m, name)
        x = closed_form_input(name + ":x", (N, C, H, W), -1, 1).requires_grad_()
        y = m(x)
        dy = closed_form_input(name + ":dy", tuple(y.shape), -1, 1)
        y.backward(dy)
        d.update({f"{name}:cfg": np.array([N, C, H, W, K, R, st, pd, op]), f"{name}:x": t2n(x), f"{name}:w": t2n(m.weight),
                  f"{name}:b": t2n(m.bias), f"{name}:y": t2n(y), f"{name}:dy": t2n(dy), f"{name}:dx": t2n(x.grad),
                  f"{name}:dw": t2n(m.weight.grad), f"{name}:db": t2n(m.bias.grad)})
    # --- MaskedConv2d (layers/layers.py:21-47): forward masks weight.data in place, wgrad is NOT masked
    name = "masked_k5"
    m = MaskedConv2d(6, 10, kernel_size=5, padding=2, stride=1)
    closed_form_fill_wrapped(m, name)
    w_before = t2n(m.weight).copy()
    x = closed_form_input(name + ":x", (2, 6, 6, 7), -1, 1).requires_grad_()
    y = m(x)
    dy = closed_form_input(name + ":dy", tuple(y.shape), -1, 1)
    y.backward(dy)
    d.update({f"{name}:x": t2n(x), f"{name}:w_before": w_before, f"{name}:w_after": t2n(m.weight), f"{name}:mask": t2n(m.mask),
              f"{name}:b": t2n(m.bias), f"{name}:y": t2n(y), f"{name}:dy": t2n(dy), f"{name}:dx": t2n(x.grad),
              f"{name}:dw": t2n(m.weight.grad), f"{name}:db": t2n(m.bias.grad)})
    # --- GDN / IGDN (layers/gdn.py:22-67)
    for name, inv in (("gdn", False), ("igdn", True)):
        m = GDN(8, inverse=inv)
        closed_form_fill_wrapped(m, name)
        x = closed_form_input(name + ":x", (2, 8, 5, 6), -2, 2).requires_grad_()
        y = m(x)
        dy = closed_form_input(name + ":dy", tuple(y.shape), -1, 1)
        y.backward(dy)
        d.update({f"{name}:x": t2n(x), f"{name}:beta": t2n(m.beta), f"{name}:gamma": t2n(m.gamma), f"{name}:y": t2n(y),
                  f"{name}:dy": t2n(dy), f"{name}:dx": t2n(x.grad), f"{name}:dbeta": t2n(m.beta.grad),
                  f"{name}:dgamma": t2n(m.gamma.grad)})
    # GDN known-answer at init (compressai_tests/test_layers.py:118-156): y = x / sqrt(1 + .1 x^2)
    m = GDN(4)
    x = closed_form_input("gdn_init:x", (1, 4, 3, 3), -2, 2)
    d["gdn_init:x"], d["gdn_init:y"] = t2n(x), t2n(m(x))
    d["gdn_init:beta"], d["gdn_init:gamma"] = t2n(m.beta), t2n(m.gamma)
    # --- LowerBound (ops/bound_ops.py:19-53; compressai_tests/test_ops.py:33-55)
    lb = LowerBound(0.3)
    x = closed_form_input("lb:x", (40,), -1, 1).requires_grad_()
    y = lb(x)
    dy = closed_form_input("lb:dy", (40,), -1, 1)
    y.backward(dy)
    d.update({"lb:x": t2n(x), "lb:y": t2n(y), "lb:dy": t2n(dy), "lb:dx": t2n(x.grad)})
    # --- EntropyBottleneck (entropy_models.py:282-470)
    eb = EntropyBottleneck(4)
    closed_form_fill_wrapped(eb, "eb")
    log = []
    eb._get_noise_cached = NoiseFeed("eb", log)
    x = closed_form_input("eb:x", (2, 4, 3, 5), -6, 6).requires_grad_()
    eb.train()
    out, lik = eb(x)
    dlik = closed_form_input("eb:dlik", tuple(lik.shape), -1, 1)
    lik.backward(dlik)
    d.update({"eb:x": t2n(x), "eb:train_out": t2n(out), "eb:train_lik": t2n(lik), "eb:dlik": t2n(dlik), "eb:dx": t2n(x.grad)})
    for n_, p in eb.named_parameters():
        d[f"eb:p:{n_}"] = t2n(p)
        if p.grad is not None:
            d[f"eb:g:{n_}"] = t2n(p.grad)
    eb.zero_grad()
    aux = eb.loss()
    aux.backward()
    d["eb:aux"], d["eb:aux_dquantiles"] = t2n(aux), t2n(eb.quantiles.grad)
    eb.eval()
    out, lik = eb(x.detach())
    d["eb:eval_out"], d["eb:eval_lik"] = t2n(out), t2n(lik)
    eb.update(force=True)
    d["eb:offset"], d["eb:cdf"], d["eb:cdf_length"] = t2n(eb._offset), t2n(eb._quantized_cdf), t2n(eb._cdf_length)
    strings = eb.compress(x.detach())
    d["eb:string0"] = np.frombuffer(strings[0], dtype=np.uint8)
    d["eb:string1"] = np.frombuffer(strings[1], dtype=np.uint8)
    d["eb:decompressed"] = t2n(eb.decompress(strings, (3, 5)))
    # --- GaussianConditional (entropy_models.py:473-604)
    gc = GaussianConditional(None)
    gc._get_noise_cached = NoiseFeed("gc", log)
    y = closed_form_input("gc:y", (2, 6, 4, 5), -8, 8).requires_grad_()
    sc = closed_form_input("gc:scales", (2, 6, 4, 5), -0.2, 3.0).requires_grad_()   # some below the 0.11 bound
    mu = closed_form_input("gc:means", (2, 6, 4, 5), -6, 6).requires_grad_()
    gc.train()
    out, lik = gc(y, sc, means=mu)
    dlik = closed_form_input("gc:dlik", tuple(lik.shape), -1, 1)
    lik.backward(dlik)
    d.update({"gc:y": t2n(y), "gc:scales": t2n(sc), "gc:means": t2n(mu), "gc:train_out": t2n(out), "gc:train_lik": t2n(lik),
              "gc:dlik": t2n(dlik), "gc:dy": t2n(y.grad), "gc:dscales": t2n(sc.grad), "gc:dmeans": t2n(mu.grad)})
    gc.eval()
    out, lik = gc(y.detach(), sc.detach(), means=mu.detach())
    d["gc:eval_out"], d["gc:eval_lik"] = t2n(out), t2n(lik)
    d["noise_log"] = np.array([f"{n}|{','.join(map(str, s))}" for n, s in log])
    save("ops_small.npz", d)


def closed_form_fill_wrapped(m, prefix):
    """closed_form_fill_ keyed by '<prefix>.<param name>' so each case gets its own stream."""
    from spatiotemporalentropymodel_amd.weights import closed_form_tensor
    with torch.no_grad():
        for n_, p in m.named_parameters():
            t = closed_form_tensor(f"{prefix}.{n_}", p.shape, p)
            if t is not None:
                p.copy_(t)


# ---------------------------------------------------------------------------
def gen_codec(ref):
    from compressai import ans
    from compressai._CXX import pmf_to_quantized_cdf
    from compressai.entropy_models import GaussianConditional
    from compressai.models.spatiotemporalpriors import get_scale_table

    d = {}
    rng = np.random.default_rng(20261002)
    # pmf_to_quantized_cdf (cpp_exts/ops/ops.cpp:24-81)
    for i, n in enumerate((3, 17, 64, 255)):
        pmf = rng.random(n).astype(np.float32) ** (1 + i)
        pmf[rng.integers(0, n, size=max(1, n // 6))] = 0.0        # force the steal-from-smallest branch
        pmf = (pmf / pmf.sum()).astype(np.float32)
        d[f"pmf{i}"] = pmf
        d[f"cdf{i}"] = np.array(pmf_to_quantized_cdf(pmf.tolist(), 16), dtype=np.uint32)
    # GaussianConditional.update tables (entropy_models.py:532-568)
    gc = GaussianConditional(None)
    gc.update_scale_table(get_scale_table())
    cdf = t2n(gc._quantized_cdf)
    d["gc:scale_table"] = t2n(gc.scale_table)
    d["gc:offset"], d["gc:cdf_length"] = t2n(gc._offset), t2n(gc._cdf_length)
    d["gc:cdf_shape"] = np.array(cdf.shape)
    d["gc:cdf_crc32"] = np.array([zlib.crc32(np.ascontiguousarray(cdf).tobytes())], dtype=np.uint64)
    for r in (0, 1, 17, 31, 48, 63):
        d[f"gc:cdf_row{r}"] = cdf[r]
    # rANS streams (cpp_exts/rans/rans_interface.cpp:99-350) against the Gaussian tables,
    # symbols far outside the table ranges force the 4-bit bypass escapes.
    sizes, offsets = t2n(gc._cdf_length).astype(np.int32), t2n(gc._offset).astype(np.int32)
    cdf_list, sizes_list, offsets_list = cdf.tolist(), sizes.tolist(), offsets.tolist()
    for i, n in enumerate((5, 7, 192, 4096)):  # n=1 overruns the reference encoder's own buffer (UB, rans_interface.cpp:170)
        idx = rng.integers(0, 64, size=n).astype(np.int32)
        scale = t2n(gc.scale_table)[idx]
        sym = np.rint(rng.normal(size=n) * scale).astype(np.int32)
        out = rng.random(n) < 0.05
        sym[out] += (rng.integers(-1, 2, size=out.sum()) * rng.integers(1, 70000, size=out.sum())).astype(np.int32)
        s = ans.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf_list, sizes_list, offsets_list)
        dec = ans.RansDecoder().decode_with_indexes(s, idx.tolist(), cdf_list, sizes_list, offsets_list)
        assert dec == sym.tolist()
        d[f"rans{i}:symbols"], d[f"rans{i}:indexes"] = sym, idx
        d[f"rans{i}:bytes"] = np.frombuffer(s, dtype=np.uint8)
    # BufferedRansEncoder with two pushes + flush, decoded piecewise with decode_stream
    enc = ans.BufferedRansEncoder()
    idx_a, idx_b = rng.integers(0, 64, size=50).astype(np.int32), rng.integers(0, 64, size=30).astype(np.int32)
    sym_a = np.rint(rng.normal(size=50) * t2n(gc.scale_table)[idx_a]).astype(np.int32)
    sym_b = np.rint(rng.normal(size=30) * t2n(gc.scale_table)[idx_b]).astype(np.int32)
    enc.encode_with_indexes(sym_a.tolist(), idx_a.tolist(), cdf_list, sizes_list, offsets_list)
    enc.encode_with_indexes(sym_b.tolist(), idx_b.tolist(), cdf_list, sizes_list, offsets_list)
    s = enc.flush()
    dec = ans.RansDecoder()
    dec.set_stream(s)
    assert dec.decode_stream(idx_a.tolist(), cdf_list, sizes_list, offsets_list) == sym_a.tolist()
    assert dec.decode_stream(idx_b.tolist(), cdf_list, sizes_list, offsets_list) == sym_b.tolist()
    d["bufrans:sym_a"], d["bufrans:idx_a"], d["bufrans:sym_b"], d["bufrans:idx_b"] = sym_a, idx_a, sym_b, idx_b
    d["bufrans:bytes"] = np.frombuffer(s, dtype=np.uint8)
    save("codec.npz", d)


# ---------------------------------------------------------------------------
def gen_stem_small_forward(ref, dtype=torch.float32, write=True):
    """BASELINE.json configs[0]: one 7x256x256 septuplet, I-frame mbt2018(N=64,M=96) transforms +
    SpatioTemporalPriorModel(ebc=64,in=96) forward in eval mode, bpp / MSE plumbing."""
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel

    log = []
    imodel = JointAutoregressiveHierarchicalPriors(64, 96).eval()
    closed_form_fill_(imodel)
    imodel.gaussian_conditional._get_noise_cached = NoiseFeed("iframe_gc", log)
    stem = SpatioTemporalPriorModel(64, 96).eval()
    closed_form_fill_(stem)
    imodel, stem = imodel.to(dtype), stem.to(dtype)
    cap = {}
    stem.EPM.register_forward_hook(lambda m, i, o: cap.__setitem__("gp", o.detach()))
    frames = [f.to(dtype) for f in smooth_frames("septuplet0", 1, 7, 256)]
    d = {"frame0_crop": t2n(frames[0][:, :, :32, :32])}
    bpp_y, bpp_z, mse = [], [], []
    with torch.no_grad():
        y0, y_cond = imodel.getY(frames[0])
        d["y0"] = t2n(y0)
        for t in range(1, 7):
            y_cur, _ = imodel.getY(frames[t])
            out = stem(y_cur, y_cond)
            x_hat = imodel.getX(out["y_hat"])
            npix = 256 * 256
            bpp_y.append(float(torch.log(out["likelihoods"]["y"]).double().sum() / (-np.log(2) * npix)))
            bpp_z.append(float(torch.log(out["likelihoods"]["z"]).double().sum() / (-np.log(2) * npix)))
            mse.append(float(((x_hat - frames[t]).double() ** 2).mean()))
            if t == 1:
                sc, mu = cap["gp"].chunk(2, 1)
                d.update({"f1:y_cur": t2n(y_cur), "f1:y_cond": t2n(y_cond), "f1:y_hat": t2n(out["y_hat"]),
                          "f1:lik_y": t2n(out["likelihoods"]["y"]), "f1:lik_z": t2n(out["likelihoods"]["z"]),
                          "f1:scales": t2n(sc), "f1:means": t2n(mu), "f1:x_hat_crop": t2n(x_hat[:, :, 100:132, 60:92])})
            if t == 6:
                d["f6:y_hat"] = t2n(out["y_hat"])
            y_cond = out["y_hat"]
    d["bpp_y"], d["bpp_z"], d["mse"] = np.array(bpp_y), np.array(bpp_z), np.array(mse)
    d["noise_log"] = np.array([f"{n}|{','.join(map(str, s))}" for n, s in log])
    if write:
        save("stem_small_forward.npz", d)
    return d


def _train_case(ref, ebc, cin, N, M, batch, size, tag, steps=2, dtype=torch.float32, write=True):
    """The per-P-frame loop of stem/trainSTEM.py:194-218 (getY -> stem fwd -> EMLoss -> backward ->
    clip_grad_norm_ -> optimizer.step -> aux_loss.backward -> aux_optimizer.step), reference code."""
    import types
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel_Res

    log = []
    imodel = JointAutoregressiveHierarchicalPriors(N, M).eval()      # trainSTEM.py:128 IFrameCompressor.eval()
    closed_form_fill_(imodel)
    imodel.gaussian_conditional._get_noise_cached = NoiseFeed("iframe_gc", log)
    stem = SpatioTemporalPriorModel_Res(ebc, cin).train()
    closed_form_fill_(stem)
    imodel, stem = imodel.to(dtype), stem.to(dtype)          # float64: the "exact" run the fp32 results are measured against
    stem.entropy_bottleneck._get_noise_cached = NoiseFeed("stem_eb", log)
    stem.gaussian_conditional._get_noise_cached = NoiseFeed("stem_gc", log)
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    optimizer, aux_optimizer = ref.configure_optimizers(stem, args)
    criterion = ref.EMLoss()
    frames = [f.to(dtype) for f in smooth_frames("train:" + tag, batch, steps + 1, size)]
    d = {}
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
    for t in range(1, steps + 1):
        optimizer.zero_grad()
        aux_optimizer.zero_grad()
        y_cur, _ = imodel.getY(frames[t])
        out = stem(y_cur.detach(), y_cond.detach())
        y_cond = out["y_hat"]
        oc = criterion(out, frames[t])
        oc["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(stem.parameters(), 1.0)
        if t == 1:
            d["s1:y_cur"], d["s1:y_hat"] = t2n(y_cur), t2n(out["y_hat"])
            d["s1:lik_y"], d["s1:lik_z"] = t2n(out["likelihoods"]["y"]), t2n(out["likelihoods"]["z"])
            for n_, p in stem.named_parameters():
                if p.grad is None:
                    continue
                g = p.grad.double()          # after clipping
                d[f"s1:gsum:{n_}"] = np.array([float(g.sum()), float(g.abs().sum()), float((g * g).sum()), float(p.numel())])
                d[f"s1:gslice:{n_}"] = t2n(p.grad.reshape(-1)[:: max(1, p.numel() // 64)][:64])
        optimizer.step()
        aux = stem.aux_loss()
        aux.backward()
        aux_optimizer.step()
        d[f"s{t}:scalars"] = np.array([float(oc["loss"]), float(oc["y_bpp_loss"]), float(oc["z_bpp_loss"]), float(aux), float(gn)])
        if t == 1:
            d["s1:dquantiles"] = t2n(stem.entropy_bottleneck.quantiles.grad)
    for n_, p in stem.named_parameters():
        pd_ = p.detach().double()
        d[f"final:psum:{n_}"] = np.array([float(pd_.sum()), float(pd_.abs().sum())])
        d[f"final:pslice:{n_}"] = t2n(p.reshape(-1)[:: max(1, p.numel() // 64)][:64])
    d["noise_log"] = np.array([f"{n}|{','.join(map(str, s))}" for n, s in log])
    d["cfg"] = np.array([ebc, cin, N, M, batch, size, steps])
    if write:
        save(f"stem_train_{tag}.npz", d)
    return d


def gen_stem_codec(ref):
    """compress()/decompress() with the host rANS (spatiotemporalpriors.py:871-1054 and 588-770)."""
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel, SpatioTemporalPriorModel_Res

    log = []
    imodel = JointAutoregressiveHierarchicalPriors(64, 96).eval()
    closed_form_fill_(imodel)
    imodel.gaussian_conditional._get_noise_cached = NoiseFeed("iframe_gc", log)
    frames = smooth_frames("codec", 1, 2, 128)
    d = {}
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
        y_cur, _ = imodel.getY(frames[1])
        y_cur = y_cur * 4.0          # random-init latents are tiny; scale so that symbols are not all zero
        y_cond = y_cond * 4.0
        d["y_cur"], d["y_cond"] = t2n(y_cur), t2n(y_cond)
        for tag, cls in (("res", SpatioTemporalPriorModel_Res), ("full", SpatioTemporalPriorModel)):
            stem = cls(64, 96).eval()
            closed_form_fill_(stem)
            stem.update(force=True)
            enc = stem.compress(y_cur, y_cond)
            dec = stem.decompress(enc["strings"], enc["shape"], y_cond)
            y_hat = dec["y_hat"] if isinstance(dec, dict) else dec
            fwd = stem(y_cur, y_cond)
            d[f"{tag}:y_string"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
            d[f"{tag}:z_string"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
            d[f"{tag}:shape"] = np.array(enc["shape"])
            d[f"{tag}:y_hat"] = t2n(y_hat)
            d[f"{tag}:fwd_y_hat"] = t2n(fwd["y_hat"])
            d[f"{tag}:eb_cdf"], d[f"{tag}:eb_offset"] = t2n(stem.entropy_bottleneck._quantized_cdf), t2n(stem.entropy_bottleneck._offset)
            d[f"{tag}:eb_cdf_length"] = t2n(stem.entropy_bottleneck._cdf_length)
            # the Gaussian tables travel with a checkpoint's state_dict; committed so that the bitstream test does not
            # depend on the libm of the machine that runs update() (torch CPU erfc/tanh differ by an ulp across CPUs)
            d["gc_cdf"] = t2n(stem.gaussian_conditional._quantized_cdf).astype(np.int32)
            d["gc_offset"] = t2n(stem.gaussian_conditional._offset)
            d["gc_cdf_length"] = t2n(stem.gaussian_conditional._cdf_length)
            d["gc_scale_table"] = t2n(stem.gaussian_conditional.scale_table)
    save("stem_codec_small.npz", d)


def gen_iframe_codec(ref):
    """mbt2018 = JointAutoregressiveHierarchicalPriors as the I-frame codec of the evaluation loop (stem/evalSTEM.py:54-59 calls
    its compress / decompress; compressai/models/priors.py:476-676): forward (eval), compress, decompress of one 128 x 128 image
    through the small model; the last analysis layer is scaled x4 after the closed-form fill so that the symbols are not all zero,
    and the first synthesis layer's weight by 1/4 so that the inverse-GDN chain sees the magnitudes it was filled for."""
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors

    imodel = JointAutoregressiveHierarchicalPriors(64, 96).eval()
    closed_form_fill_(imodel)
    with torch.no_grad():
        imodel.g_a[6].weight.mul_(4.0)
        imodel.g_a[6].bias.mul_(4.0)
        imodel.g_s[0].weight.mul_(0.25)
    imodel.update(force=True)
    x = smooth_frames("icodec", 1, 1, 128)[0]
    d = {"x": t2n(x)}
    with torch.no_grad():
        enc = imodel.compress(x)
        dec = imodel.decompress(enc["strings"], enc["shape"])
        fwd = imodel(x)
    d["y_string"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
    d["z_string"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
    d["shape"] = np.array(enc["shape"])
    d["x_hat"], d["y_hat"] = t2n(dec["x_hat"]), t2n(dec["y_hat"])
    for k in ("y", "y_hat", "x_hat"):
        d[f"fwd:{k}"] = t2n(fwd[k])
    d["fwd:lik_y"], d["fwd:lik_z"] = t2n(fwd["likelihoods"]["y"]), t2n(fwd["likelihoods"]["z"])
    d["fwd:scales"], d["fwd:means"] = t2n(fwd["entropy_params"]["scales_hat"]), t2n(fwd["entropy_params"]["means_hat"])
    d["eb_cdf"], d["eb_offset"] = t2n(imodel.entropy_bottleneck._quantized_cdf), t2n(imodel.entropy_bottleneck._offset)
    d["eb_cdf_length"] = t2n(imodel.entropy_bottleneck._cdf_length)
    d["gc_cdf"] = t2n(imodel.gaussian_conditional._quantized_cdf).astype(np.int32)
    d["gc_offset"], d["gc_cdf_length"] = t2n(imodel.gaussian_conditional._offset), t2n(imodel.gaussian_conditional._cdf_length)
    d["gc_scale_table"] = t2n(imodel.gaussian_conditional.scale_table)
    nz = int((np.frombuffer(enc["strings"][0][0], dtype=np.uint8) != 0).sum())
    print(f"iframe codec: y string {len(enc['strings'][0][0])} bytes ({nz} non-zero), z string {len(enc['strings'][1][0])} bytes, "
          f"|y| max {float(fwd['y'].abs().max()):.2f}, |x_hat| max before the clamp {float(fwd['x_hat'].abs().max()):.2f}, "
          f"decoded pixels strictly inside (0, 1): {float(((dec['x_hat'] > 0) & (dec['x_hat'] < 1)).float().mean()):.2f}, y_hat non-zero {float((dec['y_hat'] != 0).float().mean()):.2f}")
    save("iframe_codec_small.npz", d)


def _import_reference_eval_script(scratch):
    """stem/evalSTEM.py imported as a module (its `inferenceI_DVR` / `inferenceP_DVR` are what gen_eval_gop runs).  The script
    imports three packages this image does not have and that the two functions use only for a number that is NOT recorded
    (`ms_ssim`) or not at all (`torchvision.transforms`, `torchvision.utils`): stand-ins for the names are put on the scratch path
    (torchvision's by import_reference already) -- ms_ssim returns a zero tensor, the "ms-ssim" entries are dropped."""
    os.makedirs(os.path.join(scratch, "pytorch_msssim"), exist_ok=True)
    with open(os.path.join(scratch, "pytorch_msssim", "__init__.py"), "w") as f:
        f.write("import torch\ndef ms_ssim(*a, **k):\n    return torch.zeros(())\n")
    spec = importlib.util.spec_from_file_location("ref_evalSTEM", os.path.join(REF, "stem", "evalSTEM.py"))
    mod = importlib.util.module_from_spec(spec)
    keep = os.environ.get("CUDA_VISIBLE_DEVICES")
    spec.loader.exec_module(mod)                      # (sets CUDA_VISIBLE_DEVICES=0 at import: restored)
    if keep is None:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    else:
        os.environ["CUDA_VISIBLE_DEVICES"] = keep
    return mod


EVAL_GOP_SIZE = (120, 104)          # not multiples of 64: padded to 128 x 128 (4 rows top / bottom, 12 columns left / right)


def eval_gop_models(JA, STEM):
    """the two models of the evaluation chain on closed-form weights: small I-frame model with its last analysis layer x4 / first
    synthesis layer x1/4 (as gen_iframe_codec, so that symbols are not all zero), small SpatioTemporalPriorModel_Res"""
    imodel = closed_form_fill_(JA(64, 96)).eval()
    with torch.no_grad():
        imodel.g_a[6].weight.mul_(4.0)
        imodel.g_a[6].bias.mul_(4.0)
        imodel.g_s[0].weight.mul_(0.25)
    stem = closed_form_fill_(STEM(64, 96)).eval()
    return imodel, stem


def gen_eval_gop(ref, scratch, nframes=3):
    """BASELINE configs[3] as a CHAIN: the reference's own evaluation functions (stem/evalSTEM.py:34-89 inferenceI_DVR, :92-153
    inferenceP_DVR) run on frame 0 (I frame through mbt2018.compress / decompress) and frames 1, 2 (P frames: getY -> forward ->
    compress -> decompress -> getX) with the `y_conditioned` feedback of evalDataset (:199, :209), on 120 x 104 frames (pad / crop
    exercised).  Recorded per frame: both strings, shape, bpp, estimate_bpp, PSNR, the decoded latent that conditions the next
    frame; the last frame's reconstruction; the CDF tables of both models (they travel with a checkpoint's state_dict)."""
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel_Res
    ev = _import_reference_eval_script(scratch)
    imodel, stem = eval_gop_models(JointAutoregressiveHierarchicalPriors, SpatioTemporalPriorModel_Res)
    imodel.update(force=True)
    stem.update(force=True)
    h, w = EVAL_GOP_SIZE
    frames = [f[0, :, 4:4 + h, 12:12 + w].contiguous() for f in smooth_frames("evalgop", 1, nframes, 128)]
    d = {"size": np.array([h, w]), "nframes": np.array([nframes])}
    captured = {}

    def tap(model, name):                                     # the strings never leave the reference's functions: taken at compress()
        real = model.compress

        def compress(*a, **k):
            out = real(*a, **k)
            captured[name] = out
            return out
        model.compress = compress

    tap(imodel, "i")
    tap(stem, "p")
    # As shipped, inferenceP_DVR ends with out_dec["entropy_params"] (stem/evalSTEM.py:152), a key SpatioTemporalPriorModel_Res
    # .decompress does not return (spatiotemporalpriors.py:1012): KeyError after everything has been computed.  The key is added,
    # holding None, to the dictionary the model returns; no recorded number passes through it.
    real_decompress = stem.decompress
    stem.decompress = lambda *a, **k: dict(real_decompress(*a, **k), entropy_params=None)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):           # the functions print their bpp lines
        out = ev.inferenceI_DVR(imodel, frames[0])
    y_cond = out["y_conditioned"]
    for t in range(nframes):
        if t > 0:
            with contextlib.redirect_stdout(io.StringIO()):
                out = ev.inferenceP_DVR(imodel, stem, frames[t], y_cond)
            y_cond = out["y_conditioned"]
        enc = captured["i" if t == 0 else "p"]
        d[f"f{t}:y_string"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
        d[f"f{t}:z_string"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
        d[f"f{t}:shape"] = np.array(enc["shape"])
        d[f"f{t}:scalars"] = np.array([out["bpp"], out["estimate_bpp"], out["psnr"]], dtype=np.float64)
        d[f"f{t}:y_conditioned"] = t2n(y_cond)
        print(f"eval gop frame {t}: y {len(enc['strings'][0][0])} B, z {len(enc['strings'][1][0])} B, bpp {out['bpp']:.4f} "
              f"(estimate {out['estimate_bpp']:.4f}), PSNR {out['psnr']:.3f} dB")
    with torch.no_grad():                                     # the last frame's image as the reference's decoder leaves it
        h_, w_ = frames[-1].shape[-2:]
        x_hat = imodel.getX(y_cond)
        d["last:x_hat"] = t2n(x_hat[:, :, 4:4 + h_, 12:12 + w_])
    for tag, m in (("i", imodel), ("p", stem)):
        d[f"{tag}:eb_cdf"], d[f"{tag}:eb_offset"] = t2n(m.entropy_bottleneck._quantized_cdf), t2n(m.entropy_bottleneck._offset)
        d[f"{tag}:eb_cdf_length"] = t2n(m.entropy_bottleneck._cdf_length)
        d[f"{tag}:gc_cdf"] = t2n(m.gaussian_conditional._quantized_cdf).astype(np.int32)
        d[f"{tag}:gc_offset"], d[f"{tag}:gc_cdf_length"] = t2n(m.gaussian_conditional._offset), t2n(m.gaussian_conditional._cdf_length)
    save("eval_gop.npz", d)


ROI_CONV_SCALE = 0.7


def gen_roi_ops(ref):
    """SFT / SFTResblk (compressai/models/stem_utils.py:24-63) and adaptive_avg_pool2d forward + autograd gradients."""
    import torch.nn.functional as F
    from compressai.models.stem_utils import SFT, SFTResblk
    d = {}
    for tag, mod, xs, qs in (("sft", SFT(x_nc=8, prior_nc=6, ks=3, nhidden=16), (2, 8, 6, 5), (2, 6, 12, 10)),
                             ("resblk", SFTResblk(8, 6, ks=3), (2, 8, 6, 5), (2, 6, 6, 5))):
        closed_form_fill_wrapped(mod, "roiops_" + tag)
        x = closed_form_input(f"roiops:{tag}:x", xs, -1.0, 1.0).requires_grad_(True)
        q = closed_form_input(f"roiops:{tag}:q", qs, -1.0, 1.0).requires_grad_(True)
        out = mod(x, q)
        dout = closed_form_input(f"roiops:{tag}:dout", tuple(out.shape), -1.0, 1.0)
        out.backward(dout)
        d[f"{tag}:x"], d[f"{tag}:q"], d[f"{tag}:out"], d[f"{tag}:dout"] = t2n(x), t2n(q), t2n(out), t2n(dout)
        d[f"{tag}:dx"], d[f"{tag}:dq"] = t2n(x.grad), t2n(q.grad)
        for n_, p in mod.named_parameters():
            d[f"{tag}:p:{n_}"], d[f"{tag}:g:{n_}"] = t2n(p), t2n(p.grad)
    for i, (shape, osz) in enumerate((((2, 3, 16, 32), (1, 2)), ((1, 4, 10, 7), (4, 3)), ((2, 1, 64, 64), (4, 4)))):
        x = closed_form_input(f"roiops:pool{i}", shape, -1.0, 1.0).requires_grad_(True)
        y = F.adaptive_avg_pool2d(x, osz)
        dy = closed_form_input(f"roiops:pool{i}:dy", tuple(y.shape), -1.0, 1.0)
        y.backward(dy)
        d[f"pool{i}:x"], d[f"pool{i}:y"], d[f"pool{i}:dy"], d[f"pool{i}:dx"] = t2n(x), t2n(y), t2n(dy), t2n(x.grad)
    save("roi_ops.npz", d)


def _grad_digest(d, tag, module):
    for n_, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.double()
        d[f"{tag}:gsum:{n_}"] = np.array([float(g.sum()), float(g.abs().sum()), float((g * g).sum()), float(p.numel())])
        d[f"{tag}:gslice:{n_}"] = t2n(p.grad.reshape(-1)[:: max(1, p.numel() // 64)][:64])


def gen_stem_roi(ref, batch=1, size=128):
    """BASELINE.json configs[4]: the I-frame + first P-frame of stem_roi/train_stem_roi.py:515-566
    with the reference's stem_roi_i / stem_roi, PixelwiseRateDistortionLoss and quality2lambda; then the eval-mode
    compress / decompress of both models (stem_roi/eval_stem_roi.py)."""
    from compressai.models.stem_roi import stem_roi, stem_roi_i

    log, d = [], {}
    imodel, pmodel = stem_roi_i().train(), stem_roi().train()
    for tag, m in (("roi_i", imodel), ("roi_p", pmodel)):
        closed_form_fill_scaled_(m, tag, ROI_CONV_SCALE)
        m.entropy_bottleneck._get_noise_cached = NoiseFeed(tag + "_eb", log)
        m.gaussian_conditional._get_noise_cached = NoiseFeed(tag + "_gc", log)
    frames = smooth_frames("roi", batch, 2, size)
    qmap = closed_form_input("roi:qmap", (batch, 1, size, size), 0.0, 1.0)
    criterion = ref.PixelwiseRateDistortionLoss()
    lmbdamap = ref.quality2lambda(qmap)
    d["qmap"], d["lmbdamap"] = t2n(qmap), t2n(lmbdamap)

    out_i = imodel(frames[0], qmap)
    oc_i = criterion(out_i, frames[0], lmbdamap)
    oc_i["loss"].backward(retain_graph=True)
    _grad_digest(d, "i", imodel)
    caught = []
    out_i["x_hat"].register_hook(lambda g: caught.append(g.detach().clone()))   # dL_p / dx_conditioned alone
    out_p = pmodel(frames[1], out_i["x_hat"], qmap)
    oc_p = criterion(out_p, frames[1], lmbdamap)
    oc_p["loss"].backward()
    d["p:dx_conditioned"] = t2n(caught[0])
    _grad_digest(d, "p", pmodel)
    _grad_digest(d, "ip", imodel)               # I-frame model gradients accumulated through x_conditioned
    for tag, out, oc in (("i", out_i, oc_i), ("p", out_p, oc_p)):
        d[f"{tag}:x_hat"], d[f"{tag}:y_hat"] = t2n(out["x_hat"]), t2n(out["y_hat"])
        d[f"{tag}:lik_y"], d[f"{tag}:lik_z"] = t2n(out["likelihoods"]["y"]), t2n(out["likelihoods"]["z"])
        d[f"{tag}:scalars"] = np.array([float(oc["loss"]), float(oc["mse_loss"]), float(oc["bpp_loss"])])
    d["aux"] = np.array([float(imodel.aux_loss()), float(pmodel.aux_loss())])

    imodel.eval(), pmodel.eval()
    imodel.update(force=True), pmodel.update(force=True)
    with torch.no_grad():
        enc_i = imodel.compress(frames[0], qmap)
        dec_i = imodel.decompress(enc_i["strings"], enc_i["shape"])
        enc_p = pmodel.compress(frames[1], dec_i["x_hat"], qmap)
        dec_p = pmodel.decompress(enc_p["strings"], enc_p["shape"], dec_i["x_hat"])
        ev_p = pmodel(frames[1], dec_i["x_hat"], qmap)
    for tag, enc, dec in (("ci", enc_i, dec_i), ("cp", enc_p, dec_p)):
        d[f"{tag}:nbytes"] = np.array([[len(s) for s in enc["strings"][0]], [len(s) for s in enc["strings"][1]]])
        d[f"{tag}:x_hat"], d[f"{tag}:y_hat"] = t2n(dec["x_hat"]), t2n(dec["y_hat"])
    d["evp:x_hat"], d["evp:lik_y"] = t2n(ev_p["x_hat"]), t2n(ev_p["likelihoods"]["y"])
    for tag, m in (("roi_i", imodel), ("roi_p", pmodel)):
        sd = m.state_dict()
        for k in ("entropy_bottleneck._quantized_cdf", "entropy_bottleneck._offset", "entropy_bottleneck._cdf_length",
                  "gaussian_conditional._quantized_cdf", "gaussian_conditional._offset", "gaussian_conditional._cdf_length"):
            d[f"{tag}:{k}"] = t2n(sd[k])
    d["noise_log"] = np.array([f"{n}|{','.join(map(str, s))}" for n, s in log])
    d["param_names_i"] = np.array([n for n, _ in imodel.named_parameters()])
    d["param_names_p"] = np.array([n for n, _ in pmodel.named_parameters()])
    d["cfg"] = np.array([batch, size])
    import compressai.models.stem_roi as ref_roi
    for cls in ("stem_baseline", "stem_baselinev2", "stem_roi", "stem_roi_wo_gsc", "stem_roi_i"):
        sd = getattr(ref_roi, cls)().state_dict()
        d[f"keys:{cls}"] = np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in sd.items()])
    save("stem_roi.npz", d)


def gen_stem_roi_gop(ref, batch=1, size=64, nframes=3):
    """One GOP iteration of stem_roi/train_stem_roi.py:509-631 with the reference's models, criterion, clip_grad_norm_ and
    torch Adam optimisers: gradients accumulate over the frames (graph retained, x_conditioned NOT detached), the running
    gradient is clipped after every frame, the four optimisers step once at the end."""
    import types
    from compressai.models.stem_roi import stem_roi, stem_roi_i

    log, d = [], {}
    imodel, pmodel = stem_roi_i().train(), stem_roi().train()
    for tag, m in (("gop_i", imodel), ("gop_p", pmodel)):
        closed_form_fill_scaled_(m, tag, ROI_CONV_SCALE)
        m.entropy_bottleneck._get_noise_cached = NoiseFeed(tag + "_eb", log)
        m.gaussian_conditional._get_noise_cached = NoiseFeed(tag + "_gc", log)
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3, clip_max_norm=1.0)
    opt_i, aux_i = ref.configure_optimizers(imodel, args)
    opt_p, aux_p = ref.configure_optimizers(pmodel, args)
    frames = smooth_frames("roigop", batch, nframes, size)
    qmap = closed_form_input("roigop:qmap", (batch, 1, size, size), 0.0, 1.0)
    criterion = ref.PixelwiseRateDistortionLoss()
    for o in (opt_i, aux_i, opt_p, aux_p):
        o.zero_grad()
    lmbdamap = ref.quality2lambda(qmap)
    scal = []
    for idx in range(nframes):
        if idx == 0:
            out = imodel(frames[0], qmap)
            model = imodel
        else:
            out = pmodel(frames[idx], x_cond, qmap)
            model = pmodel
        x_cond = out["x_hat"]
        oc = criterion(out, frames[idx], lmbdamap)
        oc["loss"].backward(retain_graph=True)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm)
        aux = model.aux_loss()
        aux.backward()
        scal.append([float(oc["loss"]), float(oc["mse_loss"]), float(oc["bpp_loss"]), float(gn), float(aux)])
    d["scalars"] = np.array(scal)
    _grad_digest(d, "i", imodel)
    _grad_digest(d, "p", pmodel)
    for o in (opt_i, aux_i, opt_p, aux_p):
        o.step()
    for tag, m in (("i", imodel), ("p", pmodel)):
        for n_, p in m.named_parameters():
            pd_ = p.detach().double()
            d[f"{tag}:psum:{n_}"] = np.array([float(pd_.sum()), float(pd_.abs().sum())])
            d[f"{tag}:pslice:{n_}"] = t2n(p.reshape(-1)[:: max(1, p.numel() // 64)][:64])
    d["qmap"] = t2n(qmap)
    d["cfg"] = np.array([batch, size, nframes])
    save("stem_roi_gop.npz", d)


def gen_stem_variants(ref, batch=1, size=64):
    """The three remaining classes of compressai/models/stem_roi.py (stem_baseline, stem_baselinev2, stem_roi_wo_gsc):
    training-mode forward + loss.backward() with the reference, closed-form weights (scaled like the ROI goldens)."""
    import compressai.models.stem_roi as ref_roi
    d = {}
    frames = smooth_frames("variants", batch, 2, size)
    qmap = closed_form_input("variants:qmap", (batch, 1, size, size), 0.0, 1.0)
    d["qmap"] = t2n(qmap)
    for cls in ("stem_baseline", "stem_baselinev2", "stem_roi_wo_gsc"):
        log = []
        m = getattr(ref_roi, cls)().train()
        closed_form_fill_scaled_(m, cls, ROI_CONV_SCALE)
        m.entropy_bottleneck._get_noise_cached = NoiseFeed(cls + "_eb", log)
        m.gaussian_conditional._get_noise_cached = NoiseFeed(cls + "_gc", log)
        if cls == "stem_roi_wo_gsc":
            out = m(frames[1], frames[0], qmap)
            oc = ref.PixelwiseRateDistortionLoss()(out, frames[1], ref.quality2lambda(qmap))
        else:
            out = m(frames[1], frames[0])
            oc = ref.RateDistortionLoss(lmbda=0.01)(out, frames[1])
        oc["loss"].backward()
        _grad_digest(d, cls, m)
        d[f"{cls}:x_hat"], d[f"{cls}:y_hat"] = t2n(out["x_hat"]), t2n(out["y_hat"])
        d[f"{cls}:lik_y"], d[f"{cls}:lik_z"] = t2n(out["likelihoods"]["y"]), t2n(out["likelihoods"]["z"])
        d[f"{cls}:scalars"] = np.array([float(oc["loss"]), float(oc["mse_loss"]), float(oc["bpp_loss"])])
    d["cfg"] = np.array([batch, size])
    save("stem_variants.npz", d)


def gen_stem_ablations(ref, batch=2, ls=8, cin=96, dtype=torch.float32, write=True):
    """Training forward + EMLoss backward of the four non-residual STEM variants (spatiotemporalpriors.py:33-788) on
    closed-form latents: outputs, loss and every parameter gradient."""
    import compressai.models.spatiotemporalpriors as sp
    d = {}
    y_cur = closed_form_input("abl:y", (batch, cin, ls, ls), -5.0, 5.0).to(dtype)
    y_cond = closed_form_input("abl:c", (batch, cin, ls, ls), -5.0, 5.0).to(dtype)
    target = torch.zeros(batch, 3, ls * 16, ls * 16)
    for cls, ebc in (("SpatioTemporalPriorModelWithoutSPMTPM", 256), ("SpatioTemporalPriorModelWithoutSPM", 256),
                     ("SpatioTemporalPriorModelWithoutTPM", 64), ("SpatioTemporalPriorModel", 64)):
        log = []
        m = getattr(sp, cls)(ebc, cin).train()
        closed_form_fill_wrapped(m, cls)
        m = m.to(dtype)
        m.entropy_bottleneck._get_noise_cached = NoiseFeed(cls + "_eb", log)
        m.gaussian_conditional._get_noise_cached = NoiseFeed(cls + "_gc", log)
        out = m(y_cur, y_cond)
        oc = ref.EMLoss()(out, target)
        oc["loss"].backward()
        _grad_digest(d, cls, m)
        d[f"{cls}:y_hat"], d[f"{cls}:lik_y"], d[f"{cls}:lik_z"] = t2n(out["y_hat"]), t2n(out["likelihoods"]["y"]), t2n(out["likelihoods"]["z"])
        d[f"{cls}:scalars"] = np.array([float(oc["loss"]), float(oc["y_bpp_loss"]), float(oc["z_bpp_loss"])])
        d[f"{cls}:noise_log"] = np.array([f"{n}|{','.join(map(str, s_))}" for n, s_ in log])
    d["cfg"] = np.array([batch, ls, cin])
    if write:
        save("stem_ablations.npz", d)
    return d


def _close_ratio(a, b, floor=0.1, atol=0.0):
    """max over elements of |a-b| / (atol/rtol-free part of tests/conftest.py:assert_close): err / max(|b|, floor*max|b|)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float((np.maximum(np.abs(a - b) - atol, 0.0) / np.maximum(np.abs(b), floor * scale)).max())


def gen_f64(ref):
    """The same reference code run in float64 (module.double(), identical closed-form weights / inputs / noise): the
    exact values both fp32 implementations -- the reference's torch-CPU path and the HIP path -- approximate.  Saved:
    the float64 results the widened parity gates compare against, and `ref32:*` = how far the REFERENCE'S OWN fp32 run
    is from them in the tests' metrics (e.g. torch's fp32 clip_grad_norm_ is 6e-5..1e-4 off the exact norm)."""
    d = {}
    for cfg in ((64, 96, 64, 96, 2, 128, "small"), (256, 192, 192, 192, 2, 64, "big")):
        tag = cfg[-1]
        a = _train_case(ref, *cfg[:6], tag=tag, dtype=torch.float32, write=False)
        b = _train_case(ref, *cfg[:6], tag=tag, dtype=torch.float64, write=False)
        for k in ("s1:lik_y", "s1:lik_z", "s1:scalars", "s2:scalars", "s1:dquantiles"):
            d[f"{tag}:{k}"] = b[k]
        d[f"{tag}:ref32:lik_y"] = np.array([_close_ratio(a["s1:lik_y"], b["s1:lik_y"], atol=1e-9)])
        d[f"{tag}:ref32:lik_z"] = np.array([_close_ratio(a["s1:lik_z"], b["s1:lik_z"], atol=1e-9)])
        for t in (1, 2):
            d[f"{tag}:ref32:s{t}:scalars"] = np.abs(a[f"s{t}:scalars"] - b[f"s{t}:scalars"]) / np.abs(b[f"s{t}:scalars"])
        worst_slice, worst_sum = 0.0, 0.0
        for k in b:
            if k.startswith("s1:gsum:"):
                n_ = k[len("s1:gsum:"):]
                d[f"{tag}:{k}"], d[f"{tag}:s1:gslice:{n_}"] = b[k], b[f"s1:gslice:{n_}"]
                rms = float(np.sqrt(b[k][2] / b[k][3]))
                sl_err = np.abs(a[f"s1:gslice:{n_}"].astype(np.float64) - b[f"s1:gslice:{n_}"])
                worst_slice = max(worst_slice, float((sl_err / np.maximum(np.abs(b[f"s1:gslice:{n_}"]), rms)).max()))
                worst_sum = max(worst_sum, abs(a[k][0] - b[k][0]) / b[k][1], abs(a[k][1] - b[k][1]) / b[k][1])
        d[f"{tag}:ref32:grad_slice"], d[f"{tag}:ref32:grad_sums"] = np.array([worst_slice]), np.array([worst_sum])
    a = gen_stem_ablations(ref, dtype=torch.float32, write=False)
    b = gen_stem_ablations(ref, dtype=torch.float64, write=False)
    for k in b:
        if k.endswith(":noise_log") or k == "cfg":
            continue
        d["abl:" + k] = b[k]
    for cls in sorted({k.split(":")[0] for k in b if ":" in k}):
        d[f"abl:{cls}:ref32:lik_y"] = np.array([_close_ratio(a[f"{cls}:lik_y"], b[f"{cls}:lik_y"], atol=1e-9)])
        worst = 0.0
        for k in b:
            if k.startswith(f"{cls}:gslice:"):
                n_ = k[len(cls) + 8:]
                gs = b[f"{cls}:gsum:{n_}"]
                rms = float(np.sqrt(gs[2] / gs[3]))
                e = np.abs(a[k].astype(np.float64) - b[k]) / np.maximum(np.abs(b[k]), rms)
                worst = max(worst, float(e.max()))
        d[f"abl:{cls}:ref32:grad_slice"] = np.array([worst])
    # config-1 forward: the float64 STEM evaluated on the SAME (fp32) latents the fp32 golden run saw, as the test feeds them
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel
    a = gen_stem_small_forward(ref, dtype=torch.float32, write=False)
    stem64 = closed_form_fill_(SpatioTemporalPriorModel(64, 96).eval()).double()
    with torch.no_grad():
        o64 = stem64(torch.from_numpy(a["f1:y_cur"]).double(), torch.from_numpy(a["f1:y_cond"]).double())
    d["fwd:f1:lik_y"], d["fwd:f1:lik_z"] = t2n(o64["likelihoods"]["y"]), t2n(o64["likelihoods"]["z"])
    d["fwd:ref32:lik_y"] = np.array([_close_ratio(a["f1:lik_y"], d["fwd:f1:lik_y"], atol=1e-9)])
    save("stem_f64.npz", d)
    for k in sorted(d):
        if "ref32" in k:
            print(f"  {k}: {d[k]}")


def _per_channel_ratio(a, b, axis=1):
    """max over channels (index along `axis`) of max|a - b| over the channel / max|b| over the channel: every channel against
    ITS OWN maximum (floor 0), the metric of tests/test_hip_spread.py"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    red = tuple(i for i in range(b.ndim) if i != axis)
    cmax = np.abs(b).max(axis=red)
    return float((np.abs(a - b).max(axis=red) / np.maximum(cmax, 1e-300)).max())


SPREAD_DECADES = 3.0
SPREAD_ROWS = 8              # sampled elements per output-channel row of a weight gradient


def _spread_train_step(ref, dtype):
    """step 1 of the training loop (stem/trainSTEM.py:194-218, small configuration, B = 2, 128 x 128) on INHOMOGENEOUS weights:
    closed_form_fill_spread_ -- every convolution's output channels log-uniform over three decades -- for the frozen I-frame
    model and the STEM model.  Returns tensors of the forward and, for every 4-D weight, SPREAD_ROWS sampled gradient elements of
    every output-channel row (gradients BEFORE clipping: row k of dW scales with channel k's upstream gradient)."""
    from compressai.models.priors import JointAutoregressiveHierarchicalPriors
    from compressai.models.spatiotemporalpriors import SpatioTemporalPriorModel_Res
    log = []
    imodel = closed_form_fill_spread_(JointAutoregressiveHierarchicalPriors(64, 96).eval(), decades=SPREAD_DECADES)
    imodel.gaussian_conditional._get_noise_cached = NoiseFeed("iframe_gc", log)
    stem = closed_form_fill_spread_(SpatioTemporalPriorModel_Res(64, 96).train(), decades=SPREAD_DECADES)
    imodel, stem = imodel.to(dtype), stem.to(dtype)
    stem.entropy_bottleneck._get_noise_cached = NoiseFeed("stem_eb", log)
    stem.gaussian_conditional._get_noise_cached = NoiseFeed("stem_gc", log)
    frames = [f.to(dtype) for f in smooth_frames("spread", 2, 2, 128)]
    d = {}
    with torch.no_grad():
        _, y_cond = imodel.getY(frames[0])
        y_cur, _ = imodel.getY(frames[1])
    out = stem(y_cur.detach(), y_cond.detach())
    oc = ref.EMLoss()(out, frames[1])
    oc["loss"].backward()
    d["y_cur"], d["y_cond"], d["y_hat"] = t2n(y_cur), t2n(y_cond), t2n(out["y_hat"])
    d["lik_y"], d["lik_z"] = t2n(out["likelihoods"]["y"]), t2n(out["likelihoods"]["z"])
    d["scalars"] = np.array([float(oc["loss"]), float(oc["y_bpp_loss"]), float(oc["z_bpp_loss"])])
    for n_, p in stem.named_parameters():
        if p.grad is None or p.dim() != 4:
            continue
        tr = isinstance(dict(stem.named_modules())[n_.rsplit(".", 1)[0]], torch.nn.ConvTranspose2d)
        g = p.grad.transpose(0, 1) if tr else p.grad                  # rows = output channels
        g = g.reshape(g.shape[0], -1)
        cols = np.linspace(0, g.shape[1] - 1, SPREAD_ROWS).astype(np.int64)
        d[f"grow:{n_}"] = t2n(g[:, cols])
        d[f"growmax:{n_}"] = t2n(g.abs().amax(dim=1))
    return d


def gen_spread(ref):
    """Model-level parity on inhomogeneous weights: float64 and float32 runs of the reference on weights whose
    output channels span three decades per layer; the float64 tensors are the gate values, `ref32:*` the reference's own fp32
    distance from them, channel by channel against the channel's own maximum.  Second family: the variable-rate P-frame model
    (stem_roi) conditioned on the variable-rate I-frame model's reconstruction, training forward at 64 x 64."""
    a, b = _spread_train_step(ref, torch.float32), _spread_train_step(ref, torch.float64)
    d = {"decades": np.array([SPREAD_DECADES])}
    for k, v in b.items():
        d["stem:" + k] = v
    for k in ("y_cur", "y_hat", "lik_y", "lik_z"):
        d[f"stem:ref32:{k}"] = np.array([_per_channel_ratio(a[k], b[k])])
    worst = 0.0
    for k in b:
        if k.startswith("grow:"):
            # per layer: the reference's OWN fp32 distance from the exact rows, for the sampled elements and for the row maxima.  A row
            # whose gradient is the small remainder of large cancelling terms is not representable to 1e-4 of itself in ANY fp32
            # evaluation (HE.2 / EPM.0 below: 1.3-1.5e-4 for the reference's fp32 run): those rows are gated at this yardstick
            rowmax = b["growmax:" + k[5:]].astype(np.float64)
            e = np.abs(a[k].astype(np.float64) - b[k]).max(axis=1) / np.maximum(rowmax, 1e-300)
            em = np.abs(a["growmax:" + k[5:]].astype(np.float64) - rowmax) / np.maximum(rowmax, 1e-300)
            d["stem:ref32:grow:" + k[5:]] = np.array([float(e.max()), float(em.max())])
            worst = max(worst, float(e.max()))
    d["stem:ref32:grad_rows"] = np.array([worst])
    d["stem:ref32:scalars"] = np.abs(a["scalars"] - b["scalars"]) / np.abs(b["scalars"])
    # variable-rate family
    from compressai.models.stem_roi import stem_roi, stem_roi_i
    runs = {}
    for dtype in (torch.float32, torch.float64):
        log = []
        imodel, pmodel = stem_roi_i().train(), stem_roi().train()
        for tag, m in (("roi_i", imodel), ("roi_p", pmodel)):
            closed_form_fill_spread_(m, tag, decades=SPREAD_DECADES, conv_scale=ROI_CONV_SCALE)
            m.to(dtype)
            m.entropy_bottleneck._get_noise_cached = NoiseFeed("spread_" + tag + "_eb", log)
            m.gaussian_conditional._get_noise_cached = NoiseFeed("spread_" + tag + "_gc", log)
        frames = [f.to(dtype) for f in smooth_frames("spread:roi", 1, 2, 64)]
        qmap = closed_form_input("spread:qmap", (1, 1, 64, 64), 0.0, 1.0).to(dtype)
        with torch.no_grad():
            out_i = imodel(frames[0], qmap)
            out_p = pmodel(frames[1], out_i["x_hat"], qmap)
        runs[dtype] = {"i:x_hat": t2n(out_i["x_hat"]), "i:lik_y": t2n(out_i["likelihoods"]["y"]),
                       "p:x_hat": t2n(out_p["x_hat"]), "p:lik_y": t2n(out_p["likelihoods"]["y"]), "p:lik_z": t2n(out_p["likelihoods"]["z"])}
    for k, v in runs[torch.float64].items():
        d["roi:" + k] = v
        d["roi:ref32:" + k] = np.array([_per_channel_ratio(runs[torch.float32][k], v)])
    save("spread_f64.npz", d)
    for k in sorted(d):
        if "ref32" in k:
            print(f"  {k}: {d[k]}")
    for k in ("stem:y_cur", "stem:lik_y", "stem:lik_z", "roi:p:lik_y", "roi:p:x_hat", "roi:i:x_hat"):
        v = np.abs(d[k])
        red = tuple(i for i in range(v.ndim) if i != 1)
        cm = v.max(axis=red)
        print(f"  {k}: channel maxima {cm.min():.3e} .. {cm.max():.3e}")


def gen_roi_f64(ref, batch=1, size=128, vsize=64):
    """float64 runs of the variable-rate models' training forward (the cases of gen_stem_roi / gen_stem_variants, same closed-form
    weights / inputs / noise): the exact likelihoods the fp32 implementations approximate, and how far the reference's own fp32
    run is from them (`ref32:*`) -- the ROI tests gate lik_y at north_star's 1e-4 of the EXACT value with these."""
    import compressai.models.stem_roi as ref_roi
    from compressai.models.stem_roi import stem_roi, stem_roi_i
    d = {}
    runs = {}
    for dtype in (torch.float32, torch.float64):
        log = []
        imodel, pmodel = stem_roi_i().train(), stem_roi().train()
        for tag, m in (("roi_i", imodel), ("roi_p", pmodel)):
            closed_form_fill_scaled_(m, tag, ROI_CONV_SCALE)
            m.to(dtype)
            m.entropy_bottleneck._get_noise_cached = NoiseFeed(tag + "_eb", log)
            m.gaussian_conditional._get_noise_cached = NoiseFeed(tag + "_gc", log)
        frames = [f.to(dtype) for f in smooth_frames("roi", batch, 2, size)]
        qmap = closed_form_input("roi:qmap", (batch, 1, size, size), 0.0, 1.0).to(dtype)
        with torch.no_grad():
            out_i = imodel(frames[0], qmap)
            out_p = pmodel(frames[1], out_i["x_hat"], qmap)
        r = {"i:lik_y": t2n(out_i["likelihoods"]["y"]), "p:lik_y": t2n(out_p["likelihoods"]["y"])}
        vframes = [f.to(dtype) for f in smooth_frames("variants", batch, 2, vsize)]
        vq = closed_form_input("variants:qmap", (batch, 1, vsize, vsize), 0.0, 1.0).to(dtype)
        for cls in ("stem_baseline", "stem_baselinev2", "stem_roi_wo_gsc"):
            m = getattr(ref_roi, cls)().train()
            closed_form_fill_scaled_(m, cls, ROI_CONV_SCALE)
            m.to(dtype)
            m.entropy_bottleneck._get_noise_cached = NoiseFeed(cls + "_eb", log)
            m.gaussian_conditional._get_noise_cached = NoiseFeed(cls + "_gc", log)
            with torch.no_grad():
                out = m(vframes[1], vframes[0], vq) if cls == "stem_roi_wo_gsc" else m(vframes[1], vframes[0])
            r[f"{cls}:lik_y"] = t2n(out["likelihoods"]["y"])
        runs[dtype] = r
    for k, v in runs[torch.float64].items():
        d[k] = v
        d["ref32:" + k] = np.array([_close_ratio(runs[torch.float32][k], v, atol=1e-9)])
    save("stem_roi_f64.npz", d)
    for k in sorted(d):
        if k.startswith("ref32"):
            print(f"  {k}: {d[k]}")


def gen_container(ref):
    """Byte layout of compressai_examples/codec.py's container (:63-119,178-187) from the reference's own writers."""
    import io
    spec = importlib.util.spec_from_file_location("ref_codec_tool", os.path.join(REF, "compressai_examples", "codec.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    d = {}
    cases = [("mbt2018", "mse", 4, (1080, 1920), (17, 30), [[bytes(range(37))], [b"\x00\xffabc"]]),
             ("bmshj2018-factorized", "mse", 1, (64, 64), (4, 4), [[b"x"]]),
             ("cheng2020-attn", "mse", 8, (768, 512), (12, 8), [[b""], [bytes(1000)], [b"\x01\x02\x03"]])]
    for i, (model, metric, q, size, shape, strings) in enumerate(cases):
        f = io.BytesIO()
        tool.write_uchars(f, tool.get_header(model, metric, q))
        tool.write_uints(f, size)
        tool.write_uints(f, (shape[0], shape[1], len(strings)))
        for s_ in strings:
            tool.write_uints(f, (len(s_[0]),))
            tool.write_bytes(f, s_[0])
        d[f"case{i}:bytes"] = np.frombuffer(f.getvalue(), dtype=np.uint8).copy()
        d[f"case{i}:meta"] = np.array([f"{model}|{metric}|{q}|{size[0]},{size[1]}|{shape[0]},{shape[1]}"])
        for j, s_ in enumerate(strings):
            d[f"case{i}:s{j}"] = np.frombuffer(s_[0], dtype=np.uint8).copy()
    x = closed_form_input("container:x", (1, 3, 50, 75), 0.0, 1.0)
    xp = tool.pad(x, 64)
    d["pad:x"], d["pad:xp"], d["pad:back"] = t2n(x), t2n(xp), t2n(tool.crop(xp, (50, 75)))
    save("container.npz", d)


def _write_coordinate_septuplets(root, names, H=256, W=448):
    """PNG septuplets whose pixels encode (x, y, frame): R = x & 255, G = y & 255, B = (x >> 8) | (y >> 8) << 2 | frame << 4."""
    from PIL import Image
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for name in names:
        d_ = os.path.join(root, "sequences", name)
        os.makedirs(d_, exist_ok=True)
        for fr in range(1, 8):
            img = np.stack([xx & 255, yy & 255, (xx >> 8) | ((yy >> 8) << 2) | (fr << 4)], -1).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(d_, f"f00{fr}.png"))
    for lst in ("vimeo_sep_trainlist_all.txt", "sep_trainlist.txt", "sep_testlist.txt"):
        with open(os.path.join(root, lst), "w") as f:
            f.write("\n".join(names) + "\n")


def gen_roi_dataset(ref, cropsize=64, nseeds=400):
    """stem_roi/stem_roi_dataset.py: VimeoSepTuplet_QMap.__getitem__ of the reference on coordinate-coded PNGs, one call
    per `random.seed(s)`: crop offsets and frame order (decoded from the pixels), and the quality map.  Seeds are chosen
    so that every branch of the map synthesis is present."""
    import random
    spec = importlib.util.spec_from_file_location("ref_roi_dataset", os.path.join(REF, "stem_roi", "stem_roi_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    root = tempfile.mkdtemp(prefix="stem_vimeo_")
    try:
        _write_coordinate_septuplets(root, ["00001/0001"])
        d = {}
        for training in (True, False):
            ds = mod.VimeoSepTuplet_QMap(root, is_training=training, cropsize=cropsize, level=37)
            want = {"zero": 1, "hi": 2, "uni": 2, "grad": 3, "gradT": 3, "gauss": 5} if training else {"test": 2}
            kept = []
            for seed in range(nseeds):
                random.seed(seed)
                images, qmap = ds[0]
                q = qmap.numpy()[0]
                px = (images[0] * 255).round().long()
                left = int(px[0, 0, 0]) | ((int(px[2, 0, 0]) & 3) << 8)
                top = int(px[1, 0, 0]) | (((int(px[2, 0, 0]) >> 2) & 3) << 8)
                order = [int(im[2, 0, 0] * 255 + 0.5) >> 4 for im in images]
                if not training:
                    tag = "test"
                elif q.max() == q.min():
                    # replay the draws to tell the three uniform branches apart
                    random.seed(seed)
                    random.randint(0, 256 - cropsize), random.randint(0, 448 - cropsize), random.random(), random.random()
                    tmp = random.random()
                    tag = "zero" if tmp < 0.01 else "hi" if tmp < 0.2 else "uni"
                elif np.all(q == q[:1]):
                    tag = "grad"
                elif np.all(q == q[:, :1]):
                    tag = "gradT"
                else:
                    tag = "gauss"
                if want.get(tag, 0) > 0:
                    want[tag] -= 1
                    kept.append((seed, tag, top, left, order, q))
            assert not any(want.values()), want
            key = "train" if training else "test"
            d[f"{key}:seeds"] = np.array([k[0] for k in kept])
            d[f"{key}:tags"] = np.array([k[1] for k in kept])
            d[f"{key}:top_left"] = np.array([[k[2], k[3]] for k in kept])
            d[f"{key}:order"] = np.array([k[4] for k in kept])
            d[f"{key}:qmap"] = np.stack([k[5] for k in kept]).astype(np.float32)
        d["cfg"] = np.array([cropsize, 256, 448, 37])
        save("roi_dataset.npz", d)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    scratch, ref_utils = import_reference()
    try:
        which = sys.argv[1:] or ["ops", "codec", "fwd", "train_small", "train_big", "stemcodec", "iframecodec", "roi", "roiops", "roigop", "container", "dataset", "variants", "ablations", "f64", "evalgop"]
        if "ops" in which:
            gen_ops(ref_utils)
        if "codec" in which:
            gen_codec(ref_utils)
        if "fwd" in which:
            gen_stem_small_forward(ref_utils)
        if "train_small" in which:
            _train_case(ref_utils, 64, 96, 64, 96, batch=2, size=128, tag="small")
        if "train_big" in which:
            _train_case(ref_utils, 256, 192, 192, 192, batch=2, size=64, tag="big")
        if "stemcodec" in which:
            gen_stem_codec(ref_utils)
        if "iframecodec" in which:
            gen_iframe_codec(ref_utils)
        if "roi" in which:
            gen_stem_roi(ref_utils)
        if "roiops" in which:
            gen_roi_ops(ref_utils)
        if "roigop" in which:
            gen_stem_roi_gop(ref_utils)
        if "container" in which:
            gen_container(ref_utils)
        if "variants" in which:
            gen_stem_variants(ref_utils)
        if "ablations" in which:
            gen_stem_ablations(ref_utils)
        if "dataset" in which:
            gen_roi_dataset(ref_utils)
        if "f64" in which:
            gen_f64(ref_utils)
        if "roif64" in which:
            gen_roi_f64(ref_utils)
        if "spread" in which:
            gen_spread(ref_utils)
        if "evalgop" in which:
            gen_eval_gop(ref_utils, scratch)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
