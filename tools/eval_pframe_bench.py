#!/usr/bin/env python3
"""BASELINE.json configs[3]: the P-frame step of stem/evalSTEM.py:93-153 (inferenceP_DVR) on a synthetic 1920x1080 GOP:
pad to multiples of 64 -> getY -> STEM forward (rate estimate) -> compress -> decompress -> getX -> crop -> PSNR / bpp.
Times are per frame as the reference measures them (encode = getY + forward + compress, decode = decompress + getX).

    python tools/eval_pframe_bench.py [--frames 4] [--height 1080] [--width 1920]
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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    imodel = closed_form_fill_(models["mbt2018"](quality=4)).to(dev).eval()
    stem = closed_form_fill_(SpatioTemporalPriorModel_Res()).to(dev).eval()
    stem.update(force=True)
    yy, xx = torch.meshgrid(torch.arange(a.height, device=dev), torch.arange(a.width, device=dev), indexing="ij")
    frames = [torch.stack([0.5 + 0.4 * torch.sin((xx + 3 * t) / (40.0 + 10 * c)) * torch.cos((yy + t) / (55.0 - 5 * c)) for c in range(3)]).unsqueeze(0)
              for t in range(a.frames + 1)]
    with torch.no_grad():
        y_cond, _ = imodel.getY(bitstream.pad(frames[0], 64))        # stands in for the decoded I frame's latent
        y_cond = torch.round(y_cond)
        for t in range(1, a.frames + 1):
            x = frames[t]
            xp = bitstream.pad(x, 64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y_cur, _ = imodel.getY(xp)
            out_forward = stem(y_cur, y_cond)
            enc = stem.compress(y_cur, y_cond)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dec = stem.decompress(enc["strings"], enc["shape"], y_cond)
            x_hat = bitstream.crop(imodel.getX(dec["y_hat"]), (a.height, a.width))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            y_cond = dec["y_hat"]
            npix = a.height * a.width
            bpp = sum(len(s[0]) for s in enc["strings"]) * 8.0 / npix
            est = sum(float(torch.log(l).sum()) / (-math.log(2) * npix) for l in out_forward["likelihoods"].values())
            mse = float(((x - x_hat) ** 2).mean())
            print(f"P frame {t}: encode {t1 - t0:.3f} s, decode {t2 - t1:.3f} s, bpp {bpp:.4f} (estimate {est:.4f}), "
                  f"PSNR {10 * math.log10(1.0 / mse):.2f} dB (untrained closed-form weights)")
    print("reference (SURVEY.md 3.2, torch CPU in the survey container): 13 s encode + 39 s decode per 1080p P frame")


if __name__ == "__main__":
    main()
