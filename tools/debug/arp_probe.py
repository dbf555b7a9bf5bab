#!/usr/bin/env python3
"""Persistent decoder (csrc/ar_persistent.hip) against the per-position loop, position by position: the loop runs with the real rANS
decoder behind a recording callback; the persistent kernel then gets the recorded symbols whatever indexes it posts, and its
indexes are compared with the loop's.  usage: arp_probe.py [ModelClass] [H] [W]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spatiotemporalentropymodel_amd.models as Mo  # noqa: E402
from spatiotemporalentropymodel_amd import _lib, codec, functional as F  # noqa: E402
from spatiotemporalentropymodel_amd.weights import closed_form_fill_, closed_form_input  # noqa: E402

cls = sys.argv[1] if len(sys.argv) > 1 else "SpatioTemporalPriorModelWithoutTPM"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
big = len(sys.argv) > 4
dev = torch.device("cuda:0")
m = closed_form_fill_(getattr(Mo, cls)(*((256, 192) if big else (64, 96)))).to(dev).eval()
m.update(force=True)
Mch = 192 if big else 96
y_cur = closed_form_input("pd:y", (1, Mch, H, W), -6, 6).to(dev)
y_cond = closed_form_input("pd:c", (1, Mch, H, W), -6, 6).to(dev)
with torch.no_grad():
    enc = m.compress(y_cur, y_cond)
    gc = m.gaussian_conditional
    _, _, hp, tp = codec._hyper(m, None, y_cond, strings_z=enc["strings"][1], shape=enc["shape"])
B, P, H, W = hp.shape
M = P // 2
ar = codec._ARContext(m, dev)
tables = gc.host_tables()
lib = _lib.hip()
real = _lib.rans().stem_rans_decoder_decode
FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32))
real_fn = FN(C.cast(real, C.c_void_p).value)
rec_idx, rec_sym = [], []


def recording(dec, idx, n, cdfs, ncdf, stride, sizes, offsets, sym):
    rc = real_fn(dec, idx, n, cdfs, ncdf, stride, sizes, offsets, sym)
    rec_idx.append(np.ctypeslib.as_array(idx, (n,)).copy())
    rec_sym.append(np.ctypeslib.as_array(sym, (n,)).copy())
    return rc


got_idx = []


def replay(dec, idx, n, cdfs, ncdf, stride, sizes, offsets, sym):
    p = len(got_idx)
    got_idx.append(np.ctypeslib.as_array(idx, (n,)).copy())
    np.ctypeslib.as_array(sym, (n,))[:] = rec_sym[p]
    return 0


def common(buf, tp_b, hp_b):
    return (ar.w_ctx.data_ptr(), 12 * M, ar.b_ctx.data_ptr(), ar.w0.data_ptr(), ar.w0.shape[1], ar.b0.data_ptr(), ar.w0.shape[0],
            ar.w1.data_ptr(), ar.w1.shape[1], ar.b1.data_ptr(), ar.w1.shape[0], ar.w2.data_ptr(), ar.w2.shape[1], ar.b2.data_ptr(),
            buf.data_ptr(), H, W, M, codec._P, tp_b, hp_b, ar.ctx.data_ptr(), ar.h1.data_ptr(), ar.h2.data_ptr(), ar.gp.data_ptr(),
            ar.table.data_ptr(), ar.table.numel(), ar.bound, F.LRELU_SLOPE)


hp_b = hp.data_ptr()
tp_b = tp.data_ptr() if tp is not None else 0
idx_host = torch.empty(M, dtype=torch.int32).pin_memory()
sym_host = torch.empty(M, dtype=torch.int32).pin_memory()
buf1 = codec._padded(None, H, W, M, dev)
dec = codec.RansDecoder()
dec.set_stream(enc["strings"][0][0])
cb1 = FN(recording)
F._chk(lib.stem_ar_decode_image(*common(buf1, tp_b, hp_b), idx_host.data_ptr(), sym_host.data_ptr(), C.cast(cb1, C.c_void_p).value, dec._h,
                                *tables.args(), F._stream()))
torch.cuda.synchronize()
print(f"loop: {len(rec_idx)} positions")
buf2 = codec._padded(None, H, W, M, dev)
cb2 = FN(replay)
n0, n1 = ar.w0.shape[0], ar.w1.shape[0]
dbg = None
if hasattr(lib, "stem_exper_arp_debug"):
    dbg_all = torch.zeros(H * W * (4 * M + n0 + n1) + 256 * (256 + 64 * 3), device=dev)
    dbg = dbg_all[:H * W * (4 * M + n0 + n1)].view(H * W, 4 * M + n0 + n1)
    lib.stem_exper_arp_debug.argtypes = [C.c_void_p]
    lib.stem_exper_arp_debug.restype = None
    lib.stem_exper_arp_debug(dbg_all.data_ptr())
rc = lib.stem_ar_decode_image_persistent(*common(buf2, tp_b, hp_b), C.cast(cb2, C.c_void_p).value, dec._h, *tables.args(), F._stream())
torch.cuda.synchronize()
print("persistent rc", rc, (lib.stem_last_error() or b"").decode() if rc else "")
bad = [(p, int((a != b).sum())) for p, (a, b) in enumerate(zip(got_idx, rec_idx)) if (a != b).any()]
print(f"{len(got_idx)} positions answered; positions with wrong indexes: {len(bad)}; first: {bad[:10]}")
if bad:
    p = bad[0][0]
    ch = np.nonzero(got_idx[p] != rec_idx[p])[0]
    print(f" position {p} (h={p // W}, w={p % W}): channels {ch[:16]} got {got_idx[p][ch[:16]]} want {rec_idx[p][ch[:16]]}")
print("buffers equal:", bool(torch.equal(buf1, buf2)), " max |diff|", float((buf1 - buf2).abs().max()))

if dbg is not None:
    # experiments build: the four products' outputs per position (dumped by workgroup 1) against torch on the loop's final buffer,
    # and what every wavefront multiplied in EPM.0 at position 0 (its ctx columns, its look-ahead lane partials)
    lib.stem_exper_arp_debug(None)
    Wp = W + 4
    b1 = buf1.view(H + 4, Wp, M)
    hpv = hp[0].permute(1, 2, 0).reshape(H * W, 2 * M)
    tpv = None if tp is None else tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)
    for p in (0, 1, 2, W, W + 1, H * W - 1):
        h, w = divmod(p, W)
        win = torch.cat([b1[h, w:w + 5].reshape(-1), b1[h + 1, w:w + 5].reshape(-1), b1[h + 2, w:w + 2].reshape(-1)])
        ctx = ar.b_ctx + ar.w_ctx @ win
        parts = ([] if tpv is None else [tpv[p]]) + [hpv[p], ctx]
        h1 = torch.nn.functional.leaky_relu(ar.b0 + ar.w0 @ torch.cat(parts), F.LRELU_SLOPE)
        h2 = torch.nn.functional.leaky_relu(ar.b1 + ar.w1 @ h1, F.LRELU_SLOPE)
        gp = ar.b2 + ar.w2 @ h2
        d = dbg[p]
        e = [float((d[:2 * M] - ctx).abs().max()), float((d[2 * M:2 * M + n0] - h1).abs().max()), float((d[2 * M + n0:2 * M + n0 + n1] - h2).abs().max()),
             float((d[2 * M + n0 + n1:] - gp).abs().max())]
        print(f" position {p}: max |kernel - torch| ctx {e[0]:.3e} h1 {e[1]:.3e} h2 {e[2]:.3e} gp {e[3]:.3e}   (|ctx| max {float(ctx.abs().max()):.2f})")
    ex = dbg_all[H * W * (4 * M + n0 + n1):].view(256, 256 + 64 * 3)
    ctx0 = dbg[0][:2 * M]
    seen = ex[:, :2 * M]
    print("EPM.0 at position 0: wavefronts whose ctx columns differ from ctx:", int(((seen - ctx0.unsqueeze(0)).abs().max(1).values > 0).sum()), "of 256")
    segs = ([] if tpv is None else [tpv[0]]) + [hpv[0]]
    xcat = torch.cat(segs)
    worst = 0.0
    for gw in (0, 1, 100, 255):
        for r in range(3):
            n = gw + 256 * r
            if n >= n0:
                continue
            wrow = ar.w0[n, :xcat.numel()]
            ref = torch.zeros(64, device=dev)
            for si in range(len(segs)):
                for k in range(0, 2 * M, 256):
                    for l in range(64):
                        c = k + 4 * l
                        if c < 2 * M:
                            o = si * 2 * M + c
                            ref[l] += (xcat[o:o + 4] * wrow[o:o + 4]).sum()
            worst = max(worst, float((ex[gw, 256 + 64 * r:256 + 64 * r + 64] - ref).abs().max()))
    print("look-ahead lane partials of EPM.0 (4 wavefronts x 3 rows): max |kernel - reference|", worst)
