"""The evaluation loop of stem/evalSTEM.py (BASELINE configs[3]) as package functions.

    inference_iframe(imodel, x)                    stem/evalSTEM.py:34-89   (inferenceI_DVR)
    inference_pframe(imodel, stem, x, y_cond)      stem/evalSTEM.py:92-153  (inferenceP_DVR)
    eval_gop(imodel, stem, frames, gop=12)         the frame loop of evalDataset, stem/evalSTEM.py:180-216: frame index % GOP == 1
                                                   is an I frame coded by the image model's own compress / decompress, every other
                                                   frame a P frame conditioned on the previous frame's DECODED latents

Same order of operations, same returned keys, same arithmetic for bpp / PSNR as the script.  "ms-ssim": the script calls
`pytorch_msssim.ms_ssim(x, x_hat, data_range=1.0)` (:81, :147), a third-party package that is not in the reference tree (nor in this
image; unpinned in the reference's requirements): `ms_ssim` below restates the published algorithm (Wang, Simoncelli, Bovik 2003)
with that package's conventions.  PARITY UNPINNED for this one number -- no golden vector exists; tests/test_host_api.py checks it
against an independent scipy formulation and its defining properties.  It is a reporting metric computed on the HOST, outside the
timed encode / decode regions; frames smaller than 161 pixels on a side have no five-scale MS-SSIM and report None.  One deliberate difference: the script's I frame runs
on the CPU and moves `y_conditioned` to the GPU for the P frames (:196-207); here everything stays on the models' device.
The script's last line reads out_dec["entropy_params"], a key the reference model's decompress() does not return
(spatiotemporalpriors.py:1012 -> KeyError at :152 as shipped); the key is returned holding None.

Pinned by tests/golden/eval_gop.npz (tests/golden/make_golden.py:gen_eval_gop runs the reference's two functions themselves on
an I + 2 P chain) through tests/test_hip_codec.py::test_eval_gop_chain_matches_reference.
"""
from __future__ import annotations

import math
import time

import torch

from . import bitstream


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """stem/evalSTEM.py:29-31 (peak 1.0)"""
    mse = float(torch.mean((a.float() - b.float()) ** 2))
    return -10 * math.log10(mse)


_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size=11, sigma=1.5):
    c = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).float()


def _gauss_filter(x, win):
    """separable, 'valid' (no padding), one filter per channel"""
    C = x.shape[1]
    k = win.to(x.dtype)
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)


def ms_ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0):
    """Multi-scale structural similarity of two image batches [B,C,H,W] (host tensors; device tensors are copied): five scales,
    11-tap Gaussian window (sigma 1.5), K = (0.01, 0.03), the contrast-structure terms of scales 1-4 and the full SSIM of scale 5,
    each clamped at 0, raised to the published exponents and multiplied; 2x2 average pooling (odd sizes padded by one) between
    scales; mean over channels and batch.  None when the smaller side is <= 160 pixels (the fifth scale would be empty)."""
    x, y = x.detach().float().cpu(), y.detach().float().cpu()
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        return None
    win = _gauss_window()
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    terms = []
    for level in range(5):
        mu1, mu2 = _gauss_filter(x, win), _gauss_filter(y, win)
        s11 = _gauss_filter(x * x, win) - mu1 * mu1
        s22 = _gauss_filter(y * y, win) - mu2 * mu2
        s12 = _gauss_filter(x * y, win) - mu1 * mu2
        cs_map = (2 * s12 + C2) / (s11 + s22 + C2)
        ssim_map = (2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * cs_map
        if level < 4:
            terms.append(torch.relu(cs_map.flatten(2).mean(-1)))
            pad = [s % 2 for s in x.shape[2:]]
            x = torch.nn.functional.avg_pool2d(x, kernel_size=2, padding=pad)
            y = torch.nn.functional.avg_pool2d(y, kernel_size=2, padding=pad)
        else:
            terms.append(torch.relu(ssim_map.flatten(2).mean(-1)))
    w = torch.tensor(_MS_WEIGHTS).view(-1, 1, 1)
    return float(torch.prod(torch.stack(terms) ** w, dim=0).mean())


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def _bpp_terms(out_enc, out_forward, num_pixels):
    bpp = sum(len(s[0]) for s in out_enc["strings"]) * 8.0 / num_pixels
    est = {k: float(torch.log(v.float()).sum() / (-math.log(2) * num_pixels)) for k, v in out_forward["likelihoods"].items()}
    return bpp, est


@torch.no_grad()
def inference_iframe(model, x, with_msssim=True):
    """x: one image [3,h,w] in [0,1].  Pad to multiples of 64 (centred), compress + forward (the rate estimate), decompress, crop.
    `y_conditioned` is the DECODED latent of the padded image: what the next P frame is conditioned on."""
    x = x.unsqueeze(0)
    h, w = x.size(2), x.size(3)
    x_padded = bitstream.pad(x, 64)
    _sync(x)
    start = time.time()
    out_enc = model.compress(x_padded)
    out_forward = model(x_padded)
    _sync(x)
    enc_time = time.time() - start
    start = time.time()
    out_dec = model.decompress(out_enc["strings"], out_enc["shape"])
    _sync(x)
    dec_time = time.time() - start
    x_hat = bitstream.crop(out_dec["x_hat"], (h, w))
    num_pixels = x.size(0) * h * w
    bpp, est = _bpp_terms(out_enc, out_forward, num_pixels)
    return {"y_conditioned": out_dec["y_hat"], "psnr": psnr(x, x_hat), "ms-ssim": ms_ssim(x, x_hat, data_range=1.0) if with_msssim else None, "bpp": bpp,
            "estimate_bpp": sum(est.values()),
            "estimate_y_bpp": est.get("y"), "estimate_z_bpp": est.get("z"), "y_bpp": len(out_enc["strings"][0][0]) * 8.0 / num_pixels,
            "z_bpp": len(out_enc["strings"][1][0]) * 8.0 / num_pixels, "encoding_time": enc_time, "decoding_time": dec_time,
            "out_forward": out_forward, "strings": out_enc["strings"], "shape": tuple(out_enc["shape"]), "x_hat": x_hat}


@torch.no_grad()
def inference_pframe(imodel, stem, x, y_conditioned, with_msssim=True):
    """x: one frame [3,h,w]; y_conditioned: the previous frame's decoded latents.  encode = getY + forward + compress, decode =
    decompress + getX, timed as the script times them."""
    x = x.unsqueeze(0)
    h, w = x.size(2), x.size(3)
    x_padded = bitstream.pad(x, 64)
    _sync(x)
    start = time.time()
    y_cur, _ = imodel.getY(x_padded)
    out_forward = stem(y_cur, y_conditioned)
    out_enc = stem.compress(y_cur, y_conditioned)
    _sync(x)
    enc_time = time.time() - start
    start = time.time()
    out_dec = stem.decompress(out_enc["strings"], out_enc["shape"], y_conditioned)
    y_hat = out_dec["y_hat"] if isinstance(out_dec, dict) else out_dec
    x_hat = imodel.getX(y_hat)
    _sync(x)
    dec_time = time.time() - start
    x_hat = bitstream.crop(x_hat, (h, w))
    num_pixels = x.size(0) * h * w
    bpp, est = _bpp_terms(out_enc, out_forward, num_pixels)
    return {"y_conditioned": y_hat, "psnr": psnr(x, x_hat), "ms-ssim": ms_ssim(x, x_hat, data_range=1.0) if with_msssim else None, "bpp": bpp,
            "estimate_bpp": sum(est.values()),
            "estimate_y_bpp": est.get("y"), "estimate_z_bpp": est.get("z"), "y_bpp": len(out_enc["strings"][0][0]) * 8.0 / num_pixels,
            "z_bpp": len(out_enc["strings"][1][0]) * 8.0 / num_pixels, "encoding_time": enc_time, "decoding_time": dec_time,
            "entropy_params": out_dec.get("entropy_params") if isinstance(out_dec, dict) else None,
            "strings": out_enc["strings"], "shape": tuple(out_enc["shape"]), "x_hat": x_hat}


@torch.no_grad()
def eval_gop(imodel, stem, frames, gop=12, all_intra=False, with_msssim=True):
    """frames: iterable of [3,h,w] images of ONE sequence, in display order (the script's f001.png, f002.png, ...).  Frame k
    (1-based) with k % gop == 1 is an I frame, every other one a P frame conditioned on the previous frame's decoded latents
    (stem/evalSTEM.py:186-209; gop = 12 for UVG, 10 for the HEVC classes).  Returns the per-frame dictionaries of the two
    inference functions (plus "type") and the sequence averages the script logs (:217-224).  with_msssim=False leaves the host-side
    MS-SSIM out (a reporting metric next to the codec path, ~0.3 s per 1080p frame on the host)."""
    per_frame, y_cond = [], None
    for index, x in enumerate(frames, start=1):
        if all_intra or index % gop == 1 or y_cond is None:
            out = inference_iframe(imodel, x, with_msssim)
            out["type"] = "I"
        else:
            out = inference_pframe(imodel, stem, x, y_cond, with_msssim)
            out["type"] = "P"
        y_cond = out["y_conditioned"]
        per_frame.append(out)
    n = max(1, len(per_frame))
    ms = [f["ms-ssim"] for f in per_frame if f["ms-ssim"] is not None]
    return {"frames": per_frame, "psnr_ave": sum(f["psnr"] for f in per_frame) / n, "bpp_ave": sum(f["bpp"] for f in per_frame) / n,
            "msssim_ave": (sum(ms) / len(ms)) if ms else None,
            "estimate_bpp_ave": sum(f["estimate_bpp"] for f in per_frame) / n}
