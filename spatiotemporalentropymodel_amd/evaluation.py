"""The evaluation loop of stem/evalSTEM.py (BASELINE configs[3]) as package functions.

    inference_iframe(imodel, x)                    stem/evalSTEM.py:34-89   (inferenceI_DVR)
    inference_pframe(imodel, stem, x, y_cond)      stem/evalSTEM.py:92-153  (inferenceP_DVR)
    eval_gop(imodel, stem, frames, gop=12)         the frame loop of evalDataset, stem/evalSTEM.py:180-216: frame index % GOP == 1
                                                   is an I frame coded by the image model's own compress / decompress, every other
                                                   frame a P frame conditioned on the previous frame's DECODED latents

Same order of operations, same returned keys, same arithmetic for bpp / PSNR as the script (it reads "ms-ssim" from
pytorch_msssim, which is not part of this path: the key is absent here).  One deliberate difference: the script's I frame runs
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


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def _bpp_terms(out_enc, out_forward, num_pixels):
    bpp = sum(len(s[0]) for s in out_enc["strings"]) * 8.0 / num_pixels
    est = {k: float(torch.log(v.float()).sum() / (-math.log(2) * num_pixels)) for k, v in out_forward["likelihoods"].items()}
    return bpp, est


@torch.no_grad()
def inference_iframe(model, x):
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
    return {"y_conditioned": out_dec["y_hat"], "psnr": psnr(x, x_hat), "bpp": bpp, "estimate_bpp": sum(est.values()),
            "estimate_y_bpp": est.get("y"), "estimate_z_bpp": est.get("z"), "y_bpp": len(out_enc["strings"][0][0]) * 8.0 / num_pixels,
            "z_bpp": len(out_enc["strings"][1][0]) * 8.0 / num_pixels, "encoding_time": enc_time, "decoding_time": dec_time,
            "out_forward": out_forward, "strings": out_enc["strings"], "shape": tuple(out_enc["shape"]), "x_hat": x_hat}


@torch.no_grad()
def inference_pframe(imodel, stem, x, y_conditioned):
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
    return {"y_conditioned": y_hat, "psnr": psnr(x, x_hat), "bpp": bpp, "estimate_bpp": sum(est.values()),
            "estimate_y_bpp": est.get("y"), "estimate_z_bpp": est.get("z"), "y_bpp": len(out_enc["strings"][0][0]) * 8.0 / num_pixels,
            "z_bpp": len(out_enc["strings"][1][0]) * 8.0 / num_pixels, "encoding_time": enc_time, "decoding_time": dec_time,
            "entropy_params": out_dec.get("entropy_params") if isinstance(out_dec, dict) else None,
            "strings": out_enc["strings"], "shape": tuple(out_enc["shape"]), "x_hat": x_hat}


@torch.no_grad()
def eval_gop(imodel, stem, frames, gop=12, all_intra=False):
    """frames: iterable of [3,h,w] images of ONE sequence, in display order (the script's f001.png, f002.png, ...).  Frame k
    (1-based) with k % gop == 1 is an I frame, every other one a P frame conditioned on the previous frame's decoded latents
    (stem/evalSTEM.py:186-209; gop = 12 for UVG, 10 for the HEVC classes).  Returns the per-frame dictionaries of the two
    inference functions (plus "type") and the sequence averages the script logs (:217-224)."""
    per_frame, y_cond = [], None
    for index, x in enumerate(frames, start=1):
        if all_intra or index % gop == 1 or y_cond is None:
            out = inference_iframe(imodel, x)
            out["type"] = "I"
        else:
            out = inference_pframe(imodel, stem, x, y_cond)
            out["type"] = "P"
        y_cond = out["y_conditioned"]
        per_frame.append(out)
    n = max(1, len(per_frame))
    return {"frames": per_frame, "psnr_ave": sum(f["psnr"] for f in per_frame) / n, "bpp_ave": sum(f["bpp"] for f in per_frame) / n,
            "estimate_bpp_ave": sum(f["estimate_bpp"] for f in per_frame) / n}
