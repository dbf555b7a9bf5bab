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
    lib.stem_exper_arp_debug(None)
    Wp = W + 4
    b1 = buf1.view(H + 4, Wp, M)
    hpv = hp[0].permute(1, 2, 0).reshape(H * W, 2 * M)
    for p in (0, 1, 2, W, W + 1):
        h, w = divmod(p, W)
        win = torch.cat([b1[h, w:w + 5].reshape(-1), b1[h + 1, w:w + 5].reshape(-1), b1[h + 2, w:w + 2].reshape(-1)])
        ctx = ar.b_ctx + ar.w_ctx @ win
        parts = ([] if tp is None else [tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[p]]) + [hpv[p], ctx]
        h1 = torch.nn.functional.leaky_relu(ar.b0 + ar.w0 @ torch.cat(parts), F.LRELU_SLOPE)
        h2 = torch.nn.functional.leaky_relu(ar.b1 + ar.w1 @ h1, F.LRELU_SLOPE)
        gp = ar.b2 + ar.w2 @ h2
        d = dbg[p]
        e = [float((d[:2 * M] - ctx).abs().max()), float((d[2 * M:2 * M + n0] - h1).abs().max()), float((d[2 * M + n0:2 * M + n0 + n1] - h2).abs().max()),
             float((d[2 * M + n0 + n1:] - gp).abs().max())]
        print(f" position {p}: max |kernel - torch| ctx {e[0]:.3e} h1 {e[1]:.3e} h2 {e[2]:.3e} gp {e[3]:.3e}   (|ctx| max {float(ctx.abs().max()):.2f})")
        if p == 0:
            hk = d[2 * M:2 * M + n0]
            pre = lambda hpx, cx, b: torch.nn.functional.leaky_relu(b + ar.w0 @ torch.cat(([] if tp is None else [tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[p]]) + [hpx, cx]), F.LRELU_SLOPE)
            for name, alt in (("bias = 0", pre(hpv[p], ctx, 0 * ar.b0)), ("hp = 0", pre(0 * hpv[p], ctx, ar.b0)), ("ctx = 0", pre(hpv[p], 0 * ctx, ar.b0)),
                              ("ctx = 2 ctx", pre(hpv[p], 2 * ctx, ar.b0)), ("hp of p+1", pre(hpv[p + 1], ctx, ar.b0))):
                print(f"   h1(0) if {name}: max |kernel - alt| {float((hk - alt).abs().max()):.3e}")
            inv = torch.where(hk > 0, hk, hk / F.LRELU_SLOPE)
            o = 0 if tp is None else 2 * M
            comps = [ar.b0, ar.w0[:, o:o + 2 * M] @ hpv[p], ar.w0[:, o + 2 * M:o + 4 * M] @ ctx] + ([] if tp is None else [ar.w0[:, :2 * M] @ tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[p]])
            kc = inv - comps[0] - comps[1] - (comps[3] if len(comps) > 3 else 0)
            print("   kernel's ctx part (rows 0..7):", [round(float(v), 4) for v in kc[:8]], " true:", [round(float(v), 4) for v in comps[2][:8]])
            kh = inv - comps[0] - comps[2] - (comps[3] if len(comps) > 3 else 0)
            print("   kernel's hp part (rows 0..7):", [round(float(v), 4) for v in kh[:8]], " true:", [round(float(v), 4) for v in comps[1][:8]])
            for lanes in (8, 16, 24, 32, 40, 48):
                part = ar.w0[:, o + 2 * M:o + 2 * M + 4 * lanes] @ ctx[:4 * lanes]
                print(f"   W_ctx ctx over the first {lanes} lanes' columns: max |kernel ctx part - it| {float((kc - part).abs().max()):.3e}")
            A = torch.stack(comps, 1).double().cpu()
            sol = torch.linalg.lstsq(A, inv.double().cpu().unsqueeze(1)).solution.flatten()
            print("   least-squares coefficients of [bias, W_hp hp, W_ctx ctx, (W_tp tp)] in the kernel's pre-activation:", [round(float(v), 4) for v in sol],
                  " residual", float((A @ sol.unsqueeze(1) - inv.double().cpu().unsqueeze(1)).abs().max()))
            for nm, a_, b_ in (("h1", d[2 * M:2 * M + n0], h1), ("h2", d[2 * M + n0:2 * M + n0 + n1], h2), ("gp", d[2 * M + n0 + n1:], gp)):
                bad = ((a_ - b_).abs() > 1e-4).nonzero().flatten()
                print(f"   {nm}: {bad.numel()} wrong rows: {bad[:32].tolist()}")
                good = ((a_ - b_).abs() <= 1e-4).nonzero().flatten()
                print(f"   {nm}: right rows: {good.tolist()[:120]}")
                print(f"   {nm}: errors of rows 0..15: {[round(float(v), 4) for v in (a_ - b_)[:16]]}")
        if p >= 1:
            hk = d[2 * M:2 * M + n0]
            for name, hpx, cx in (("hp of p-1", hpv[p - 1], ctx), ("ctx of p-1", hpv[p], dbg[p - 1][:2 * M]), ("both of p-1", hpv[p - 1], dbg[p - 1][:2 * M]),
                                  ("hp = 0", torch.zeros_like(hpv[p]), ctx), ("ctx = 0", hpv[p], torch.zeros_like(ctx))):
                pr = ([] if tp is None else [tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[p]]) + [hpx, cx]
                alt = torch.nn.functional.leaky_relu(ar.b0 + ar.w0 @ torch.cat(pr), F.LRELU_SLOPE)
                print(f"   h1 if {name}: max |kernel - alt| {float((hk - alt).abs().max()):.3e}")
        if p == 1:
            bad = (d[:2 * M] - ctx).abs() > 1e-3
            print("   wrong ctx rows:", bad.nonzero().flatten()[:24].tolist(), "of", int(bad.sum()))

if dbg is not None:
    ex = dbg_all[H * W * (4 * M + n0 + n1):].view(256, 256 + 64 * 3)
    ctx0 = dbg[0][:2 * M]
    seen = ex[:, :2 * M]
    print("E0 at position 0: waves whose ctx columns differ from ctx:", int(((seen - ctx0.unsqueeze(0)).abs().max(1).values > 0).sum()), "of 256; max diff",
          float((seen - ctx0.unsqueeze(0)).abs().max()))
    # lane partials over tp | hp: lane l sums columns 4 l + 256 t of each segment, in order
    segs = ([] if tp is None else [tp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[0]]) + [hp[0].permute(1, 2, 0).reshape(H * W, 2 * M)[0]]
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
    print("look-ahead partials of EPM.0 (4 waves x 3 rows): max |kernel - reference|", worst)
if dbg is not None:
    print("ctx[:12]      ", [round(float(v), 4) for v in ctx0[:12]])
    print("wave 0 sees   ", [round(float(v), 4) for v in seen[0][:12]])
    print("wave 7 sees   ", [round(float(v), 4) for v in seen[7][:12]])
    print("b_ctx[:12]    ", [round(float(v), 4) for v in ar.b_ctx[:12]])
    # is what they see a permutation of ctx?
    a_, b_ = torch.sort(seen[0])[0], torch.sort(ctx0)[0]
    print("sorted equal:", bool(torch.equal(a_, b_)), " max |sorted diff|", float((a_ - b_).abs().max()))
