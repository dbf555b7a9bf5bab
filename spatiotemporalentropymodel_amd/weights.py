"""Closed-form deterministic parameter fill.

Real STEM checkpoints are 43-72 MB and cannot be committed as fixtures
(SURVEY.md §8(c) "Golden-vector plan").  Instead every parameter tensor is
filled from an integer hash of (tensor name, flat index), with the same
variance the reference's initialisers give (kaiming_normal_ for conv weights,
`compressai/models/priors.py:67-72`; GDN `compressai/layers/gdn.py:42-50`;
EntropyBottleneck `compressai/entropy_models/entropy_models.py:310-335`) plus a
perturbation so that biases, tanh factors, off-diagonal gammas and non-zero
medians are all exercised.  The golden generator applies this to the imported
reference model, the tests apply it to ours: identical weights, no blobs.

Pure numpy/torch on CPU; no dependency on the HIP extension.
"""
from __future__ import annotations

import re
import zlib

import numpy as np
import torch

_PEDESTAL = 2.0 ** -36  # NonNegativeParametrizer.reparam_offset**2 (parametrizers.py:29-35)


def hash_uniform(name: str, n: int, salt: int = 0) -> np.ndarray:
    """float64 array of n values in [-1, 1), a pure function of (name, index, salt)."""
    seed = np.uint64(zlib.crc32(name.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 / float(1 << 53)) - 1.0


def closed_form_tensor(name: str, shape, like: torch.Tensor | None = None) -> torch.Tensor | None:
    """Value for one state-dict entry, or None when the entry is left untouched."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.rsplit(".", 1)[-1]
    u = hash_uniform(name, n).reshape(shape)

    if leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        v = u * np.sqrt(3.0) * np.sqrt(2.0 / fan_in)
    elif leaf == "bias" and len(shape) == 1:
        v = 0.05 * u
    elif leaf == "beta" and len(shape) == 1:
        v = np.sqrt(np.maximum(1.0 + 0.2 * u + _PEDESTAL, _PEDESTAL))
    elif leaf == "gamma" and len(shape) == 2:
        v = 0.1 * np.eye(shape[0]) + 0.02 * np.abs(u)
        v = np.sqrt(np.maximum(v + _PEDESTAL, _PEDESTAL))
    elif re.fullmatch(r"_matrix\d", leaf):
        # entropy_models.py:312-319: init = log(expm1(1/scale/filters[i+1]))
        scale = 10.0 ** (1.0 / 5.0)
        init = np.log(np.expm1(1.0 / scale / shape[1]))
        v = init + 0.2 * u
    elif re.fullmatch(r"_bias\d", leaf):
        v = 0.5 * u
    elif re.fullmatch(r"_factor\d", leaf):
        v = 0.3 * u
    elif leaf == "quantiles" and len(shape) == 3:
        v = np.empty(shape)
        v[:, 0, 0] = -10.0 + u[:, 0, 0]
        v[:, 0, 1] = 0.4 * u[:, 0, 1]
        v[:, 0, 2] = 10.0 + u[:, 0, 2]
    else:
        return None
    t = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    if like is not None:
        t = t.to(device=like.device, dtype=like.dtype)
    return t


@torch.no_grad()
def closed_form_fill_(module: torch.nn.Module) -> torch.nn.Module:
    """Fill every parameter of `module` in place (works on the reference's and on our modules)."""
    for name, p in module.named_parameters():
        t = closed_form_tensor(name, p.shape, p)
        if t is not None:
            p.copy_(t)
    return module


@torch.no_grad()
def closed_form_fill_scaled_(module: torch.nn.Module, prefix: str, conv_scale: float) -> torch.nn.Module:
    """closed_form_fill_ keyed by '<prefix>.<name>', with every 4-D (convolution) weight multiplied by `conv_scale`.
    The variable-rate models stack ~60 convolutions with multiplicative SFT stages: the variance-preserving scale of
    closed_form_tensor overflows there, 0.7 keeps latents in +-2 and reconstructions in +-2 (tests/golden/make_golden.py)."""
    for name, p in module.named_parameters():
        t = closed_form_tensor(f"{prefix}.{name}", p.shape, p)
        if t is not None:
            p.copy_(t * conv_scale if t.dim() == 4 else t)
    return module


def channel_spread(name: str, K: int, decades: float = 3.0) -> np.ndarray:
    """per-output-channel scale factors of one layer: log-uniform over `decades` decades (a pure function of the layer's name and
    the channel index), normalised to unit mean square so that a stack of layers keeps its signal level"""
    u = (hash_uniform(name + ":spread", K, salt=7) + 1.0) * 0.5              # [0, 1)
    s = 10.0 ** (-decades * u)
    return s / np.sqrt(np.mean(s * s))


@torch.no_grad()
def closed_form_fill_spread_(module: torch.nn.Module, prefix: str = "", decades: float = 3.0, conv_scale: float = 1.0) -> torch.nn.Module:
    """closed_form_fill_ with INHOMOGENEOUS layers: output channel k of every convolution (4-D weight and its bias) is multiplied
    by channel_spread(name)[k] -- three decades between the loudest and the quietest channel of each layer, as trained
    entropy-parameter heads and GDN-normalised transforms show, instead of the one variance per layer of the plain fill.  The
    model-level parity tests on these weights judge every output channel against ITS OWN maximum (tests/test_hip_spread.py).
    For a transposed convolution (weight [in, out, kh, kw]) the output channels are dimension 1."""
    transposed = {n for n, m in module.named_modules() if type(m).__name__ == "ConvTranspose2d"}      # torch's and layers.ConvTranspose2d
    for name, p in module.named_parameters():
        key = f"{prefix}.{name}" if prefix else name
        t = closed_form_tensor(key, p.shape, p)
        if t is None:
            continue
        owner, leaf = name.rsplit(".", 1) if "." in name else ("", name)
        if leaf == "weight" and t.dim() == 4:
            tr = owner in transposed
            K = t.shape[1] if tr else t.shape[0]
            sc = torch.from_numpy(channel_spread(key[: -len(".weight")], K, decades).astype(np.float32)).to(t.device)
            t = t * conv_scale * (sc.view(1, K, 1, 1) if tr else sc.view(K, 1, 1, 1))
        elif leaf == "bias" and t.dim() == 1:
            w = dict(module.named_parameters()).get(owner + ".weight")
            if w is not None and w.dim() == 4 and t.shape[0] in (w.shape[0], w.shape[1]):
                sc = torch.from_numpy(channel_spread(key[: -len(".bias")], t.shape[0], decades).astype(np.float32)).to(t.device)
                t = t * sc
        p.copy_(t)
    return module


def closed_form_input(name: str, shape, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    """Deterministic input/noise tensor in [lo, hi)."""
    n = int(np.prod(shape))
    u = (hash_uniform(name, n, salt=1).reshape(shape) + 1.0) * 0.5
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32))


def smooth_frames(name: str, batch: int, frames: int, size: int) -> list[torch.Tensor]:
    """Synthetic septuplet in the shape contract of stem/dataset_vidseq.py:57-88:
    list of `frames` tensors [B,3,size,size] in [0,1]; a sum of low-frequency
    sinusoids translated by (2t, t) pixels per frame."""
    rng = hash_uniform(name, batch * 3 * 8 * 4).reshape(batch, 3, 8, 4)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    out = []
    for t in range(frames):
        img = np.full((batch, 3, size, size), 0.5)
        for b in range(batch):
            for c in range(3):
                for k in range(8):
                    fy, fx, ph, amp = rng[b, c, k]
                    img[b, c] += 0.08 * (1 + amp) * np.sin(
                        2 * np.pi * ((1 + 3 * (fy + 1)) * (yy + t) / size + (1 + 3 * (fx + 1)) * (xx + 2 * t) / size) + np.pi * ph
                    )
        out.append(torch.from_numpy(np.clip(img, 0.0, 1.0).astype(np.float32)))
    return out
