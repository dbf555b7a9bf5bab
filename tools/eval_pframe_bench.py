#!/usr/bin/env python3
"""BASELINE.json configs[3]: the P-frame step of stem/evalSTEM.py:93-153 (inferenceP_DVR) on a synthetic 1920x1080 GOP:
pad to multiples of 64 -> getY -> STEM forward (rate estimate) -> compress -> decompress -> getX -> crop -> PSNR / bpp.
Times are per frame as the reference measures them (encode = getY + forward + compress, decode = decompress + getX).

    python tools/eval_pframe_bench.py [--frames 4] [--height 1080] [--width 1920] [--sequences G]

--sequences G > 1 codes G independent sequences side by side (batch dimension of compress / decompress): the decoder's
raster loop is latency-bound per position, and G images share that latency (csrc/ar.hip: stem_ar_decode_batch), so the
per-frame decode time drops ~G-fold; times are reported per frame (wall time of the batched call / G).
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporalentropymodel_amd import bitstream  # noqa: E402
from spatiotemporalentropymodel_amd.models import SpatioTemporalPriorModel_Res  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_  # noqa: E402
from spatiotemporalentropymodel_amd.zoo import models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--sequences", type=int, default=1)
    ap.add_argument("--workers", type=int, default=0, help="ignored since round 5 (the persistent decoder's row map fixes 32 workgroups)")
    a = ap.parse_args()
    if a.workers:
        from spatiotemporalentropymodel_amd import _lib
        _lib.check(_lib.hip().stem_tuning_set(b"arp_workers", a.workers))
    G = a.sequences
    dev = torch.device("cuda:0")
    imodel = closed_form_fill_(models["mbt2018"](quality=4)).to(dev).eval()
    imodel.update(force=True)
    stem = closed_form_fill_(SpatioTemporalPriorModel_Res()).to(dev).eval()
    stem.update(force=True)
    yy, xx = torch.meshgrid(torch.arange(a.height, device=dev), torch.arange(a.width, device=dev), indexing="ij")
    frames = [torch.stack([torch.stack([0.5 + 0.4 * torch.sin((xx + 3 * t + 17 * g) / (40.0 + 10 * c + g)) * torch.cos((yy + t) / (55.0 - 5 * c))
                                        for c in range(3)]) for g in range(G)]) for t in range(a.frames + 1)]
    enc_t, dec_t = [], []

    def getY(x):        # the transforms take <= 4 full-HD frames per call (2 GiB buffer-descriptor range of the conv kernels)
        return torch.cat([imodel.getY(x[i:i + 4])[0] for i in range(0, x.shape[0], 4)])

    def getX(y):
        return torch.cat([imodel.getX(y[i:i + 4].contiguous(memory_format=torch.channels_last)) for i in range(0, y.shape[0], 4)])

    with torch.no_grad():
        # the I frame through mbt2018's own codec, as inferenceI_DVR does (stem/evalSTEM.py:54-59); its decoded latent conditions frame 1
        x0 = bitstream.pad(frames[0], 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc_i = [imodel.compress(x0[i:i + 1]) for i in range(G)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dec_i = [imodel.decompress(e["strings"], e["shape"]) for e in enc_i]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        y_cond = torch.cat([d["y_hat"] for d in dec_i])
        bpp_i = sum(len(b) for e in enc_i for s_ in e["strings"] for b in s_) * 8.0 / (a.height * a.width * G)
        print(f"I frame x {G} sequences: encode {(t1 - t0) / G:.3f} s, decode {(t2 - t1) / G:.3f} s per frame, bpp {bpp_i:.4f} (mbt2018.compress / decompress)")
        for t in range(1, a.frames + 1):
            x = frames[t]
            xp = bitstream.pad(x, 64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y_cur = getY(xp)
            out_forward = stem(y_cur, y_cond)
            enc = stem.compress(y_cur, y_cond)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec = stem.decompress(enc["strings"], enc["shape"], y_cond)
            x_hat = bitstream.crop(getX(dec["y_hat"]), (a.height, a.width))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            y_cond = dec["y_hat"]
            npix = a.height * a.width
            bpp = sum(len(b) for s in enc["strings"] for b in s) * 8.0 / (npix * G)
            est = sum(float(torch.log(l).sum()) / (-math.log(2) * npix * G) for l in out_forward["likelihoods"].values())
            mse = float(((x - x_hat) ** 2).mean())
            enc_t.append((t1 - t0) / G)
            dec_t.append((t2 - t1) / G)
            print(f"P frame {t} x {G} sequences: encode {(t1 - t0) / G:.3f} s, decode {(t2 - t1) / G:.3f} s per frame, bpp {bpp:.4f} "
                  f"(estimate {est:.4f}), PSNR {10 * math.log10(1.0 / mse):.2f} dB (untrained closed-form weights)")
    print(f"mean over {a.frames} P frames x {G} sequences: encode {sum(enc_t) / len(enc_t):.3f} s, decode {sum(dec_t) / len(dec_t):.3f} s per frame")
    print("reference (SURVEY.md 3.2, torch CPU in the survey container): 13 s encode + 39 s decode per 1080p P frame")


if __name__ == "__main__":
    main()
