"""TEST / BASELINE INFRASTRUCTURE -- not part of the product path (nothing under spatiotemporalentropymodel_amd/ imports this).

The hot path as the REFERENCE executes it on a CPU: torch CPU operators (torch.nn.functional.conv2d / conv_transpose2d through
MKL-DNN, autograd, torch.nn.utils.clip_grad_norm_, torch.optim.Adam), written here from the reference's description -- not
its files -- so that `bench.py`'s `cpu_baseline` times what a user of the reference gets on the host cores:

  g_a                      priors.py:421-429 (4x conv5x5 s2, GDN gdn.py:52-67 with the reparametrisation parametrizers.py:42-45)
  STEM forward (training)  spatiotemporalpriors.py:845-868 (_Res) / :561-585: HE, EntropyBottleneck (entropy_models.py:388-452),
                           HD, TPM, masked context convolution (layers.py:21-47), EPM, GaussianConditional (entropy_models.py:570-596)
  EMLoss                   utils.py:18-27
  optimisation step        stem/trainSTEM.py:203-218: backward, clip_grad_norm_(1.0), Adam(1e-4), aux loss + Adam(1e-3) on .quantiles

State is a dict of tensors with the state-dict names of the reference (the oracle's `sd`); noise is injected (dict 'z', 'q', 'lik')
exactly as oracle/stem_oracle.py:stem_forward takes it, which is how tests/test_oracle_vs_golden.py pins this file against the C
oracle (forward tensors and every parameter gradient)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as Fn

LRELU = 0.01
SCALE_BOUND, LIK_BOUND = 0.11, 1e-9
PEDESTAL = 2.0 ** -36


class _LowerBound(torch.autograd.Function):
    """max(x, bound) whose gradient passes where x >= bound or where it pushes x up (bound_ops.py:19-53)"""

    @staticmethod
    def forward(ctx, x, bound):
        b = torch.full((), float(bound), dtype=x.dtype)
        ctx.save_for_backward(x, b)
        return torch.max(x, b)

    @staticmethod
    def backward(ctx, g):
        x, b = ctx.saved_tensors
        return ((x >= b) | (g < 0)).to(g.dtype) * g, None


def lower_bound(x, bound):
    return _LowerBound.apply(x, bound)


def reparam(p, minimum):
    """NonNegativeParametrizer.forward: max(p, sqrt(minimum + pedestal))^2 - pedestal"""
    return lower_bound(p, math.sqrt(minimum + PEDESTAL)) ** 2 - PEDESTAL


def gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    C = x.shape[1]
    norm = Fn.conv2d(x * x, reparam(gamma, 0.0).reshape(C, C, 1, 1), reparam(beta, beta_min))
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


def g_a(sd, x, prefix="g_a."):
    h = x
    for i in range(4):
        h = Fn.conv2d(h, sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"], stride=2, padding=2)
        if i < 3:
            h = gdn(h, sd[f"{prefix}{2 * i + 1}.beta"], sd[f"{prefix}{2 * i + 1}.gamma"])
    return h


def _eb_logits(sd, v, stop_gradient=False, prefix="entropy_bottleneck."):
    """cumulative logits of the per-channel MLP 1 -> 3 -> 3 -> 3 -> 3 -> 1 at v [C,1,n]"""
    logits = v
    for i in range(5):
        m, b = sd[f"{prefix}_matrix{i}"], sd[f"{prefix}_bias{i}"]
        if stop_gradient:
            m, b = m.detach(), b.detach()
        logits = torch.matmul(Fn.softplus(m), logits) + b
        if i < 4:
            f = sd[f"{prefix}_factor{i}"]
            if stop_gradient:
                f = f.detach()
            logits = logits + torch.tanh(f) * torch.tanh(logits)
    return logits


def eb_likelihood(sd, v):
    lower, upper = _eb_logits(sd, v - 0.5), _eb_logits(sd, v + 0.5)
    sign = -torch.sign(lower + upper).detach()
    return lower_bound(torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower)), LIK_BOUND)


def eb_aux_loss(sd, target, prefix="entropy_bottleneck."):
    """EntropyBottleneck.loss: sum |logits_cumulative(quantiles) - target| with the MLP parameters held fixed"""
    return torch.abs(_eb_logits(sd, sd[f"{prefix}quantiles"], stop_gradient=True) - target).sum()


def gc_likelihood(y, scales, means):
    s = lower_bound(scales, SCALE_BOUND)
    v = torch.abs(y - means)
    c = 2.0 ** -0.5
    upper = 0.5 * torch.erfc(-c * (0.5 - v) / s)
    lower = 0.5 * torch.erfc(-c * (-0.5 - v) / s)
    return lower_bound(upper - lower, LIK_BOUND)


def _lrelu(x):
    return Fn.leaky_relu(x, LRELU)


def stem_forward(sd, y_cur, y_cond, residual: bool, training: bool, noise=None):
    """SpatioTemporalPriorModel(_Res).forward on torch CPU operators; same contract as oracle/stem_oracle.py:stem_forward"""
    he = torch.cat([y_cur, y_cond], 1)
    he = _lrelu(Fn.conv2d(he, sd["HE.0.weight"], sd["HE.0.bias"], padding=1))
    he = _lrelu(Fn.conv2d(he, sd["HE.2.weight"], sd["HE.2.bias"], stride=2, padding=2))
    z = Fn.conv2d(he, sd["HE.4.weight"], sd["HE.4.bias"], stride=2, padding=2)
    B, Cz, hz, wz = z.shape
    zc = z.permute(1, 0, 2, 3).reshape(Cz, 1, -1)
    if training:
        zq = zc + noise["z"].permute(1, 0, 2, 3).reshape(Cz, 1, -1)
    else:
        med = sd["entropy_bottleneck.quantiles"][:, :, 1:2]
        zq = torch.round(zc - med) + med
    lik_z = eb_likelihood(sd, zq).reshape(Cz, B, hz, wz).permute(1, 0, 2, 3)
    z_hat = zq.reshape(Cz, B, hz, wz).permute(1, 0, 2, 3)
    hd = _lrelu(Fn.conv_transpose2d(z_hat, sd["HD.0.weight"], sd["HD.0.bias"], stride=2, padding=2, output_padding=1))
    hd = _lrelu(Fn.conv_transpose2d(hd, sd["HD.2.weight"], sd["HD.2.bias"], stride=2, padding=2, output_padding=1))
    hp = Fn.conv2d(hd, sd["HD.4.weight"], sd["HD.4.bias"], padding=1)
    tp = _lrelu(Fn.conv2d(y_cond, sd["TPM.0.weight"], sd["TPM.0.bias"], padding=2))
    tp = _lrelu(Fn.conv2d(tp, sd["TPM.2.weight"], sd["TPM.2.bias"], padding=2))
    tp = Fn.conv2d(tp, sd["TPM.4.weight"], sd["TPM.4.bias"], padding=2)
    target = (y_cur - y_cond) if residual else y_cur
    t_hat = target + noise["q"] if training else torch.round(target)
    w = sd["context_prediction.weight"]
    kh, kw = w.shape[2:]
    mask = torch.ones_like(w)
    mask[:, :, kh // 2, kw // 2:] = 0
    mask[:, :, kh // 2 + 1:] = 0
    w.data *= mask                                      # in place on the data, as the reference does: the gradient is NOT masked
    ctx = Fn.conv2d(t_hat, w, sd["context_prediction.bias"], padding=2)
    e = torch.cat([tp, hp, ctx], 1)
    e = _lrelu(Fn.conv2d(e, sd["EPM.0.weight"], sd["EPM.0.bias"]))
    e = _lrelu(Fn.conv2d(e, sd["EPM.2.weight"], sd["EPM.2.bias"]))
    scales, means = Fn.conv2d(e, sd["EPM.4.weight"], sd["EPM.4.bias"]).chunk(2, 1)
    out = target + noise["lik"] if training else torch.round(target - means) + means
    lik_y = gc_likelihood(out, scales, means)
    y_hat = (t_hat + y_cond) if residual else t_hat
    return {"y_hat": y_hat, "lik_y": lik_y, "lik_z": lik_z, "scales": scales, "means": means}


def em_loss(lik_y, lik_z, num_pixels):
    c = -math.log(2.0) * num_pixels
    return torch.log(lik_y).sum() / c + torch.log(lik_z).sum() / c


class PFrameTrainer:
    """One P-frame optimisation step of stem/trainSTEM.py:203-218 on the CPU (the timed body of bench.py's cpu_baseline)."""

    def __init__(self, isd, ssd, target=None):
        self.isd = {k: torch.as_tensor(v) for k, v in isd.items()}
        self.ssd = {k: torch.as_tensor(v).clone().requires_grad_(True) for k, v in ssd.items()}
        q = "entropy_bottleneck.quantiles"
        self.main = [v for k, v in sorted(self.ssd.items()) if k != q]
        self.opt = torch.optim.Adam(self.main, lr=1e-4)
        self.aux_opt = torch.optim.Adam([self.ssd[q]], lr=1e-3)
        C = self.ssd[q].shape[0]
        t = math.log(2.0 / 1e-9 - 1.0)                  # EntropyBottleneck.target for tail_mass 1e-9
        self.target = torch.tensor([-t, 0.0, t]).reshape(1, 1, 3).expand(C, 1, 3) if target is None else torch.as_tensor(target)

    def step(self, x, noise, y_noise):
        """x [B,3,H,W] in [0,1]; noise as stem_forward; y_noise: the U(-1/2,1/2) of getY (priors.py:691).  Returns (loss, t_g_a)."""
        import time
        t0 = time.perf_counter()
        with torch.no_grad():
            y = g_a(self.isd, x)
        t_ga = time.perf_counter() - t0
        y_cond = y + y_noise
        out = stem_forward(self.ssd, y, y_cond, residual=True, training=True, noise=noise)
        loss = em_loss(out["lik_y"], out["lik_z"], x.shape[0] * x.shape[2] * x.shape[3])
        self.opt.zero_grad()
        self.aux_opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.main, 1.0)
        self.opt.step()
        aux = eb_aux_loss(self.ssd, self.target)
        aux.backward()
        self.aux_opt.step()
        return float(loss.detach()), t_ga


@torch.no_grad()
def decode_positions(ssd, y_cond, hp, tp, string, tables, decoder, max_positions=None):
    """The raster-order decoding loop of SpatioTemporalPriorModel_Res._decompress_ar (spatiotemporalpriors.py:1015-1054) as the
    reference executes it on a CPU: per position a 5x5 crop through torch's conv2d with the context weights, the three 1x1
    EPM convolutions, table indexes from the scales, Python lists into the symbol decoder, dequantise, write back.  `decoder`:
    the reference's own RansDecoder (oracle/_ref, stem_oracle.reference_rans_decoder()).  hp / tp: the hyper / temporal prior
    tensors [1, 2M, H, W].  Stops after max_positions (the bounded timing sample of bench.py --config eval).
    -> (res_hat [1, M, H, W] as far as decoded, positions, seconds in the loop)"""
    import time
    f = lambda k: torch.as_tensor(ssd[k], dtype=torch.float32)      # noqa: E731
    M = y_cond.shape[1]
    H, W = hp.shape[-2:]
    wc = f("context_prediction.weight").clone()
    wc[:, :, 2, 2:] = 0                                              # MaskedConv2d type A (layers.py:39-42), multiplied in place by forward
    wc[:, :, 3:] = 0
    bc = f("context_prediction.bias")
    epm = [(f(f"EPM.{i}.weight"), f(f"EPM.{i}.bias")) for i in (0, 2, 4)]
    table = torch.as_tensor(tables["gc_scale_table"], dtype=torch.float32)
    cdf = [list(map(int, r)) for r in tables["gc_cdf"]]
    lens, offs = [int(v) for v in tables["gc_cdf_length"]], [int(v) for v in tables["gc_offset"]]
    decoder.set_stream(string)
    hp, tp = torch.as_tensor(hp, dtype=torch.float32), torch.as_tensor(tp, dtype=torch.float32)
    res = torch.zeros(1, M, H + 4, W + 4)
    n, t0 = 0, time.perf_counter()
    for h in range(H):
        for w in range(W):
            if max_positions is not None and n >= max_positions:
                return res[:, :, 2:2 + H, 2:2 + W], n, time.perf_counter() - t0
            crop = res[:, :, h:h + 5, w:w + 5]
            ctx = Fn.conv2d(crop, wc, bc)
            g = torch.cat((tp[:, :, h:h + 1, w:w + 1], hp[:, :, h:h + 1, w:w + 1], ctx), dim=1)
            g = _lrelu(Fn.conv2d(g, *epm[0]))
            g = _lrelu(Fn.conv2d(g, *epm[1]))
            g = Fn.conv2d(g, *epm[2])
            scales, means = g.chunk(2, 1)
            # GaussianConditional.build_indexes (entropy_models.py:598-604): count the table entries below the bounded scale
            s = torch.clamp(scales, min=SCALE_BOUND)
            idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
            for v in table[:-1]:
                idx -= (s <= v).int()
            rv = decoder.decode_stream(idx.squeeze().tolist(), cdf, lens, offs)
            rv = torch.Tensor(rv).reshape(1, -1, 1, 1) + means
            res[:, :, h + 2:h + 3, w + 2:w + 3] = rv
            n += 1
    return res[:, :, 2:2 + H, 2:2 + W], n, time.perf_counter() - t0
