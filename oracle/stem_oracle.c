/*
 * stem_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the arithmetic on the STEM hot path of
 * mmSir/SpatioTemporalEntropyModel (a CompressAI 1.1.1 fork).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product path (libstem_hip.so) never links or calls it.
 *
 * Pinning: every function here is checked in tests/test_oracle_vs_golden.py
 * against golden vectors captured from the imported reference itself
 * (tests/golden/make_golden.py, run in the build container where
 * /root/reference exists).
 *
 * Layout: NCHW fp32, exactly the reference's tensors.  Sums are accumulated
 * in double so that the oracle is the *most accurate* fp32-rounded answer; the
 * reference (torch CPU / MKL-DNN) and the HIP kernels both differ from it only
 * by fp32 summation-order noise (<<1e-4 relative, the north_star tolerance).
 *
 * Reference citations are `path:line` relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* nn.Conv2d forward (cross-correlation, zero padding).                      */
/* Call sites: compressai/models/utils.py:112-120 (conv factory, k5 s2 p2),   */
/* compressai/models/spatiotemporalpriors.py:807-838 (TPM/HE/HD.4/EPM).       */
/* x[N,C,H,W] w[K,C,R,S] b[K] (may be NULL) -> y[N,K,Ho,Wo]                   */
/* ------------------------------------------------------------------------- */
ORC_API void orc_conv2d_fwd(const float *x, const float *w, const float *b, float *y,
                            int N, int C, int H, int W, int K, int R, int S,
                            int stride, int pad)
{
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)Ho * Wo);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const double bias = b ? (double)b[k] : 0.0;
                for (int i = 0; i < Ho * Wo; ++i) acc[i] = bias;
                for (int c = 0; c < C; ++c) {
                    const float *xp = x + ((size_t)n * C + c) * H * W;
                    const float *wp = w + ((size_t)k * C + c) * R * S;
                    for (int r = 0; r < R; ++r)
                        for (int s = 0; s < S; ++s) {
                            const double wv = wp[r * S + s];
                            for (int oy = 0; oy < Ho; ++oy) {
                                const int iy = oy * stride - pad + r;
                                if (iy < 0 || iy >= H) continue;
                                /* ox range with 0 <= ox*stride - pad + s < W */
                                int ox0 = 0;
                                while (ox0 < Wo && ox0 * stride - pad + s < 0) ++ox0;
                                int ox1 = Wo;
                                while (ox1 > ox0 && (ox1 - 1) * stride - pad + s >= W) --ox1;
                                const float *xr = xp + (size_t)iy * W - pad + s;
                                double *ar = acc + (size_t)oy * Wo;
                                for (int ox = ox0; ox < ox1; ++ox)
                                    ar[ox] += wv * (double)xr[ox * stride];
                            }
                        }
                }
                float *yp = y + ((size_t)n * K + k) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) yp[i] = (float)acc[i];
            }
        free(acc);
    }
}

/* dX of nn.Conv2d (torch autograd of F.conv2d).  dy[N,K,Ho,Wo] -> dx[N,C,H,W] */
ORC_API void orc_conv2d_dgrad(const float *dy, const float *w, float *dx,
                              int N, int C, int H, int W, int K, int R, int S,
                              int stride, int pad)
{
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)H * W);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                for (int i = 0; i < H * W; ++i) acc[i] = 0.0;
                for (int k = 0; k < K; ++k) {
                    const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                    const float *wp = w + ((size_t)k * C + c) * R * S;
                    for (int r = 0; r < R; ++r)
                        for (int s = 0; s < S; ++s) {
                            const double wv = wp[r * S + s];
                            for (int oy = 0; oy < Ho; ++oy) {
                                const int iy = oy * stride - pad + r;
                                if (iy < 0 || iy >= H) continue;
                                for (int ox = 0; ox < Wo; ++ox) {
                                    const int ix = ox * stride - pad + s;
                                    if (ix < 0 || ix >= W) continue;
                                    acc[(size_t)iy * W + ix] += wv * (double)dyp[(size_t)oy * Wo + ox];
                                }
                            }
                        }
                }
                float *dxp = dx + ((size_t)n * C + c) * H * W;
                for (int i = 0; i < H * W; ++i) dxp[i] = (float)acc[i];
            }
        free(acc);
    }
}

/* dW, db of nn.Conv2d.  dw[K,C,R,S], db[K] (db may be NULL).                 */
/* All R*S taps are computed, also for MaskedConv2d: the reference masks     */
/* weight.data in forward only (compressai/layers/layers.py:44-47), autograd */
/* still produces gradients for masked taps.                                 */
ORC_API void orc_conv2d_wgrad(const float *x, const float *dy, float *dw, float *db,
                              int N, int C, int H, int W, int K, int R, int S,
                              int stride, int pad)
{
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
            for (int r = 0; r < R; ++r)
                for (int s = 0; s < S; ++s) {
                    double a = 0.0;
                    for (int n = 0; n < N; ++n) {
                        const float *xp = x + ((size_t)n * C + c) * H * W;
                        const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride - pad + r;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < Wo; ++ox) {
                                const int ix = ox * stride - pad + s;
                                if (ix < 0 || ix >= W) continue;
                                a += (double)xp[(size_t)iy * W + ix] * (double)dyp[(size_t)oy * Wo + ox];
                            }
                        }
                    }
                    dw[(((size_t)k * C + c) * R + r) * S + s] = (float)a;
                }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int n = 0; n < N; ++n) {
                const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) a += dyp[i];
            }
            db[k] = (float)a;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* nn.ConvTranspose2d forward.  w[C,K,R,S] (in x out), output_padding op.     */
/* compressai/models/utils.py:122-130 (deconv factory k5 s2 p2 op1),          */
/* compressai/models/spatiotemporalpriors.py:821-826 (HD.0, HD.2).            */
/* x[N,C,H,W] -> y[N,K,Ho,Wo], Ho=(H-1)*stride-2*pad+R+op                      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_deconv2d_fwd(const float *x, const float *w, const float *b, float *y,
                              int N, int C, int H, int W, int K, int R, int S,
                              int stride, int pad, int opad)
{
    const int Ho = (H - 1) * stride - 2 * pad + R + opad;
    const int Wo = (W - 1) * stride - 2 * pad + S + opad;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)Ho * Wo);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const double bias = b ? (double)b[k] : 0.0;
                for (int i = 0; i < Ho * Wo; ++i) acc[i] = bias;
                for (int c = 0; c < C; ++c) {
                    const float *xp = x + ((size_t)n * C + c) * H * W;
                    const float *wp = w + ((size_t)c * K + k) * R * S;
                    for (int r = 0; r < R; ++r)
                        for (int s = 0; s < S; ++s) {
                            const double wv = wp[r * S + s];
                            for (int iy = 0; iy < H; ++iy) {
                                const int oy = iy * stride - pad + r;
                                if (oy < 0 || oy >= Ho) continue;
                                for (int ix = 0; ix < W; ++ix) {
                                    const int ox = ix * stride - pad + s;
                                    if (ox < 0 || ox >= Wo) continue;
                                    acc[(size_t)oy * Wo + ox] += wv * (double)xp[(size_t)iy * W + ix];
                                }
                            }
                        }
                }
                float *yp = y + ((size_t)n * K + k) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) yp[i] = (float)acc[i];
            }
        free(acc);
    }
}

/* dX of ConvTranspose2d: dx[n,c,iy,ix] = sum_{k,r,s} dy[n,k,iy*st-p+r, ix*st-p+s] w[c,k,r,s] */
ORC_API void orc_deconv2d_dgrad(const float *dy, const float *w, float *dx,
                                int N, int C, int H, int W, int K, int R, int S,
                                int stride, int pad, int opad)
{
    const int Ho = (H - 1) * stride - 2 * pad + R + opad;
    const int Wo = (W - 1) * stride - 2 * pad + S + opad;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double a = 0.0;
                    for (int k = 0; k < K; ++k) {
                        const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                        const float *wp = w + ((size_t)c * K + k) * R * S;
                        for (int r = 0; r < R; ++r) {
                            const int oy = iy * stride - pad + r;
                            if (oy < 0 || oy >= Ho) continue;
                            for (int s = 0; s < S; ++s) {
                                const int ox = ix * stride - pad + s;
                                if (ox < 0 || ox >= Wo) continue;
                                a += (double)dyp[(size_t)oy * Wo + ox] * (double)wp[r * S + s];
                            }
                        }
                    }
                    dx[(((size_t)n * C + c) * H + iy) * W + ix] = (float)a;
                }
}

/* dW[C,K,R,S], db[K] of ConvTranspose2d */
ORC_API void orc_deconv2d_wgrad(const float *x, const float *dy, float *dw, float *db,
                                int N, int C, int H, int W, int K, int R, int S,
                                int stride, int pad, int opad)
{
    const int Ho = (H - 1) * stride - 2 * pad + R + opad;
    const int Wo = (W - 1) * stride - 2 * pad + S + opad;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k)
            for (int r = 0; r < R; ++r)
                for (int s = 0; s < S; ++s) {
                    double a = 0.0;
                    for (int n = 0; n < N; ++n) {
                        const float *xp = x + ((size_t)n * C + c) * H * W;
                        const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                        for (int iy = 0; iy < H; ++iy) {
                            const int oy = iy * stride - pad + r;
                            if (oy < 0 || oy >= Ho) continue;
                            for (int ix = 0; ix < W; ++ix) {
                                const int ox = ix * stride - pad + s;
                                if (ox < 0 || ox >= Wo) continue;
                                a += (double)xp[(size_t)iy * W + ix] * (double)dyp[(size_t)oy * Wo + ox];
                            }
                        }
                    }
                    dw[(((size_t)c * K + k) * R + r) * S + s] = (float)a;
                }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int n = 0; n < N; ++n) {
                const float *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) a += dyp[i];
            }
            db[k] = (float)a;
        }
    }
}

/* nn.LeakyReLU (default negative_slope 0.01: spatiotemporalpriors.py:809 etc.) */
ORC_API void orc_lrelu_fwd(const float *x, float *y, size_t n, float slope)
{
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}
/* grad wrt input given the *output* y (sign(y)==sign(x) because slope>0) */
ORC_API void orc_lrelu_bwd(const float *y, const float *dy, float *dx, size_t n, float slope)
{
    for (size_t i = 0; i < n; ++i) dx[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

/* ------------------------------------------------------------------------- */
/* Spatial feature transform of the variable-rate models,                    */
/* compressai/models/stem_utils.py:36-43: out = x * (1 + gamma) + beta, with  */
/* the leaky-ReLU its callers apply next (stem_utils.py:56-57,62-63;          */
/* stem_roi.py:566-573) folded in; slope 1 = no activation.                   */
ORC_API void orc_sft_fwd(const float *x, const float *g, const float *b, float *o, size_t n, float slope)
{
    for (size_t i = 0; i < n; ++i) {
        const float v = x[i] * (1.f + g[i]) + b[i];
        o[i] = v > 0.f ? v : v * slope;
    }
}
ORC_API void orc_sft_bwd(const float *x, const float *g, const float *o, const float *dout, float *dx, float *dg, float *db,
                         size_t n, float slope)
{
    for (size_t i = 0; i < n; ++i) {
        const float d = o[i] > 0.f ? dout[i] : dout[i] * slope;
        dx[i] = d * (1.f + g[i]);
        dg[i] = d * x[i];
        db[i] = d;
    }
}
/* F.adaptive_avg_pool2d(x, (Ho, Wo)) (stem_utils.py:37, stem_roi.py:563), general window rule of              */
/* torch: rows floor(o*H/Ho) .. ceil((o+1)*H/Ho); NCHW.                                                        */
ORC_API void orc_adaptive_avgpool_fwd(const float *x, float *y, int NC, int H, int W, int Ho, int Wo)
{
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho;
                const int x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
                float s = 0.f;
                for (int iy = y0; iy < y1; ++iy)
                    for (int ix = x0; ix < x1; ++ix) s += x[((size_t)p * H + iy) * W + ix];
                y[((size_t)p * Ho + oy) * Wo + ox] = s / (float)((y1 - y0) * (x1 - x0));
            }
}
ORC_API void orc_adaptive_avgpool_bwd(const float *dy, float *dx, int NC, int H, int W, int Ho, int Wo)
{
    memset(dx, 0, sizeof(float) * (size_t)NC * H * W);
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho;
                const int x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
                const float d = dy[((size_t)p * Ho + oy) * Wo + ox] / (float)((y1 - y0) * (x1 - x0));
                for (int iy = y0; iy < y1; ++iy)
                    for (int ix = x0; ix < x1; ++ix) dx[((size_t)p * H + iy) * W + ix] += d;
            }
}

/* ------------------------------------------------------------------------- */
/* GDN / IGDN forward.  compressai/layers/gdn.py:52-67 with the               */
/* NonNegativeParametrizer (compressai/ops/parametrizers.py:27-45):           */
/*   pedestal = 2^-36 ; bound = sqrt(minimum + pedestal)                      */
/*   p' = max(p, bound)^2 - pedestal                                          */
/*   norm[n,i,h,w] = beta'[i] + sum_j gamma'[i,j] x[n,j,h,w]^2                */
/*   y = x * rsqrt(norm)   (inverse: x * sqrt(norm))                          */
/* beta_p[C], gamma_p[C,C] are the stored (reparametrised) parameters.        */
/* ------------------------------------------------------------------------- */
ORC_API void orc_gdn_fwd(const float *x, const float *beta_p, const float *gamma_p, float *y,
                         int N, int C, int H, int W, int inverse, float beta_min)
{
    const float pedestal = (float)ldexp(1.0, -36);
    const float bbound = (float)sqrt((double)beta_min + ldexp(1.0, -36));
    const float gbound = (float)sqrt(0.0 + ldexp(1.0, -36));
    float *beta = (float *)malloc(sizeof(float) * C);
    float *gamma = (float *)malloc(sizeof(float) * (size_t)C * C);
    for (int i = 0; i < C; ++i) {
        float v = beta_p[i] > bbound ? beta_p[i] : bbound;
        beta[i] = v * v - pedestal;
    }
    for (size_t i = 0; i < (size_t)C * C; ++i) {
        float v = gamma_p[i] > gbound ? gamma_p[i] : gbound;
        gamma[i] = v * v - pedestal;
    }
    const size_t HW = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < C; ++i) {
            const float *xn = x + (size_t)n * C * HW;
            for (size_t p = 0; p < HW; ++p) {
                double a = beta[i];
                for (int j = 0; j < C; ++j) {
                    const double xv = xn[(size_t)j * HW + p];
                    a += (double)gamma[(size_t)i * C + j] * (double)(float)(xv * xv);
                }
                const float norm = (float)a;
                const float xv = xn[(size_t)i * HW + p];
                y[((size_t)n * C + i) * HW + p] = inverse ? xv * sqrtf(norm) : xv / sqrtf(norm);
            }
        }
    free(beta);
    free(gamma);
}

/* GDN / IGDN backward = torch autograd of gdn.py:52-67 including the NonNegativeParametrizer chain
 * (parametrizers.py:42-45: out = LowerBound(p)^2 - pedestal; LowerBound passes the gradient iff
 * p >= bound or the incoming gradient is negative, bound_ops.py:28-31).
 *   n_i = beta'_i + sum_j gamma'_ij x_j^2 ;  GDN: y = x n^-1/2 ;  IGDN: y = x n^1/2
 *   g_i := dL/dn_i = dy_i x_i * (-1/2 n_i^-3/2 | +1/2 n_i^-1/2)
 *   dx_k = dy_k n_k^(-1/2|+1/2) + 2 x_k sum_i gamma'_ik g_i ; dbeta'_i = sum_pix g_i ; dgamma'_ij = sum_pix g_i x_j^2 */
ORC_API void orc_gdn_bwd(const float *x, const float *dy, const float *beta_p, const float *gamma_p, float *dx,
                         float *dbeta, float *dgamma, int N, int C, int H, int W, int inverse, float beta_min)
{
    const double pedestal = ldexp(1.0, -36);
    const float bbound = (float)sqrt((double)beta_min + pedestal), gbound = (float)sqrt(pedestal);
    const size_t HW = (size_t)H * W;
    double *beta = (double *)malloc(sizeof(double) * C), *gamma = (double *)malloc(sizeof(double) * (size_t)C * C);
    double *db = (double *)calloc(C, sizeof(double)), *dg = (double *)calloc((size_t)C * C, sizeof(double));
    for (int i = 0; i < C; ++i) {
        const double v = beta_p[i] > bbound ? beta_p[i] : bbound;
        beta[i] = v * v - pedestal;
    }
    for (size_t i = 0; i < (size_t)C * C; ++i) {
        const double v = gamma_p[i] > gbound ? gamma_p[i] : gbound;
        gamma[i] = v * v - pedestal;
    }
    double *g = (double *)malloc(sizeof(double) * C), *nn = (double *)malloc(sizeof(double) * C);
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < HW; ++p) {
            const float *xp = x + (size_t)n * C * HW + p;
            const float *dyp = dy + (size_t)n * C * HW + p;
            for (int i = 0; i < C; ++i) {
                double a = beta[i];
                for (int j = 0; j < C; ++j) a += gamma[(size_t)i * C + j] * (double)xp[j * HW] * (double)xp[j * HW];
                nn[i] = a;
                const double xi = xp[i * HW], dyi = dyp[i * HW];
                g[i] = inverse ? 0.5 * dyi * xi / sqrt(a) : -0.5 * dyi * xi / (a * sqrt(a));
                db[i] += g[i];
                for (int j = 0; j < C; ++j) dg[(size_t)i * C + j] += g[i] * (double)xp[j * HW] * (double)xp[j * HW];
            }
            for (int k = 0; k < C; ++k) {
                double a = 0.0;
                for (int i = 0; i < C; ++i) a += gamma[(size_t)i * C + k] * g[i];
                const double u = inverse ? dyp[k * HW] * sqrt(nn[k]) : dyp[k * HW] / sqrt(nn[k]);
                dx[(size_t)n * C * HW + k * HW + p] = (float)(u + 2.0 * xp[k * HW] * a);
            }
        }
    for (int i = 0; i < C; ++i) {
        const double lb = beta_p[i] > bbound ? beta_p[i] : bbound;
        const double gl = 2.0 * lb * db[i];
        dbeta[i] = (beta_p[i] >= bbound || gl < 0) ? (float)gl : 0.f;
    }
    for (size_t i = 0; i < (size_t)C * C; ++i) {
        const double lb = gamma_p[i] > gbound ? gamma_p[i] : gbound;
        const double gl = 2.0 * lb * dg[i];
        dgamma[i] = (gamma_p[i] >= gbound || gl < 0) ? (float)gl : 0.f;
    }
    free(beta); free(gamma); free(db); free(dg); free(g); free(nn);
}

/* ------------------------------------------------------------------------- */
/* EntropyBottleneck factorised density.                                      */
/* compressai/entropy_models/entropy_models.py:388-422.                       */
/* Per channel c: logits = v; for i in 0..4:                                   */
/*   logits = softplus(M_i[c]) @ logits + b_i[c];                              */
/*   if i<4: logits += tanh(f_i[c]) * tanh(logits)                             */
/* filters (1,3,3,3,3,1).  Parameter pack per channel (58 floats):            */
/*   M0[3x1] b0[3] f0[3] M1[3x3] b1[3] f1[3] M2[3x3] b2[3] f2[3]                */
/*   M3[3x3] b3[3] f3[3] M4[1x3] b4[1]                                          */
/* ------------------------------------------------------------------------- */
#define EB_NPARAM 58
static const int eb_in[5] = {1, 3, 3, 3, 3};
static const int eb_out[5] = {3, 3, 3, 3, 1};

static double softplus_d(double x) { return x > 20.0 ? x : log1p(exp(x)); }
static double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

/* forward of the cumulative-logits MLP; if `st` != NULL stores pre/post
 * activations for the backward: st[i*6 + 0..2] = pre-tanh logits (after bias),
 * st[i*6+3..5] = layer input. */
static double eb_logits(const float *p, double v, double *st)
{
    double in[3] = {v, 0, 0}, out[3];
    const float *q = p;
    for (int i = 0; i < 5; ++i) {
        const int ni = eb_in[i], no = eb_out[i];
        const float *M = q;
        q += no * ni;
        const float *b = q;
        q += no;
        const float *f = NULL;
        if (i < 4) {
            f = q;
            q += no;
        }
        for (int o = 0; o < no; ++o) {
            double a = 0.0;
            for (int j = 0; j < ni; ++j) a += softplus_d(M[o * ni + j]) * in[j];
            a += b[o];
            if (st) {
                st[i * 6 + o] = a;
            }
            if (f) a += tanh((double)f[o]) * tanh(a);
            out[o] = a;
        }
        if (st)
            for (int j = 0; j < 3; ++j) st[i * 6 + 3 + j] = j < ni ? in[j] : 0.0;
        for (int o = 0; o < 3; ++o) in[o] = o < no ? out[o] : 0.0;
    }
    return in[0];
}

/* backward of eb_logits: given dL/dlogit `g`, accumulate into dp[58] and return dL/dv */
static double eb_logits_bwd(const float *p, const double *st, double g, double *dp)
{
    /* offsets of each layer's block in the 58-pack */
    int off[5];
    {
        int o = 0;
        for (int i = 0; i < 5; ++i) {
            off[i] = o;
            o += eb_out[i] * eb_in[i] + eb_out[i] + (i < 4 ? eb_out[i] : 0);
        }
    }
    double gout[3] = {g, 0, 0};
    for (int i = 4; i >= 0; --i) {
        const int ni = eb_in[i], no = eb_out[i];
        const float *M = p + off[i];
        const float *f = i < 4 ? p + off[i] + no * ni + no : NULL;
        double *dM = dp ? dp + off[i] : NULL;
        double *db = dp ? dp + off[i] + no * ni : NULL;
        double *df = (dp && i < 4) ? dp + off[i] + no * ni + no : NULL;
        double gin[3] = {0, 0, 0};
        for (int o = 0; o < no; ++o) {
            const double a = st[i * 6 + o];
            double ga = gout[o];
            if (f) {
                const double tf = tanh((double)f[o]), ta = tanh(a);
                if (df) df[o] += gout[o] * ta * (1.0 - tf * tf);
                ga += gout[o] * tf * (1.0 - ta * ta);
            }
            if (db) db[o] += ga;
            for (int j = 0; j < ni; ++j) {
                const double m = M[o * ni + j];
                if (dM) dM[o * ni + j] += ga * st[i * 6 + 3 + j] * sigmoid_d(m);
                gin[j] += ga * softplus_d(m);
            }
        }
        for (int j = 0; j < 3; ++j) gout[j] = gin[j];
    }
    return gout[0];
}

/* likelihood of values v (already noised / rounded) for channel-major input.
 * v[C][L] -> lik[C][L], floor `bound` (LowerBound, bound_ops.py:19-31).       */
ORC_API void orc_eb_likelihood_fwd(const float *v, const float *params, float *lik,
                                   int C, size_t L, float bound)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float *p = params + (size_t)c * EB_NPARAM;
        for (size_t i = 0; i < L; ++i) {
            const double x = v[(size_t)c * L + i];
            const double lo = eb_logits(p, x - 0.5, NULL);
            const double up = eb_logits(p, x + 0.5, NULL);
            const double sum = lo + up;
            const double sg = sum > 0 ? -1.0 : (sum < 0 ? 1.0 : 0.0);
            double l = fabs(sigmoid_d(sg * up) - sigmoid_d(sg * lo));
            float lf = (float)l;
            lik[(size_t)c * L + i] = lf > bound ? lf : bound;
        }
    }
}

/* backward: dlik[C][L] -> dv[C][L], dparams[C][58] (accumulated over L).
 * LowerBound rule: pass iff lik_raw >= bound or grad < 0.                     */
ORC_API void orc_eb_likelihood_bwd(const float *v, const float *params, const float *dlik,
                                   float *dv, float *dparams, int C, size_t L, float bound)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float *p = params + (size_t)c * EB_NPARAM;
        double dp[EB_NPARAM];
        for (int k = 0; k < EB_NPARAM; ++k) dp[k] = 0.0;
        for (size_t i = 0; i < L; ++i) {
            const double x = v[(size_t)c * L + i];
            double st_lo[30], st_up[30];
            const double lo = eb_logits(p, x - 0.5, st_lo);
            const double up = eb_logits(p, x + 0.5, st_up);
            const double sum = lo + up;
            const double sg = sum > 0 ? -1.0 : (sum < 0 ? 1.0 : 0.0);
            const double su = sigmoid_d(sg * up), sl = sigmoid_d(sg * lo);
            const double diff = su - sl;
            const float lraw = (float)fabs(diff);
            double g = dlik[(size_t)c * L + i];
            if (!(lraw >= bound || g < 0)) g = 0.0;
            const double gd = g * (diff > 0 ? 1.0 : (diff < 0 ? -1.0 : 0.0));
            const double gup = gd * su * (1.0 - su) * sg;
            const double glo = -gd * sl * (1.0 - sl) * sg;
            double gx = 0.0;
            gx += eb_logits_bwd(p, st_up, gup, dparams ? dp : NULL);
            gx += eb_logits_bwd(p, st_lo, glo, dparams ? dp : NULL);
            if (dv) dv[(size_t)c * L + i] = (float)gx;
        }
        if (dparams)
            for (int k = 0; k < EB_NPARAM; ++k) dparams[(size_t)c * EB_NPARAM + k] = (float)dp[k];
    }
}

/* EntropyBottleneck.loss (entropy_models.py:383-386): sum |logits(quantiles) - target|
 * with stop-gradient on the MLP -> gradient only wrt quantiles[C][3].          */
ORC_API float orc_eb_aux_loss(const float *quantiles, const float *params, const float *target3,
                              float *dquant, int C)
{
    double loss = 0.0;
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < 3; ++k) {
            double st[30];
            const double lg = eb_logits(params + (size_t)c * EB_NPARAM, quantiles[c * 3 + k], st);
            const double d = lg - target3[k];
            loss += fabs(d);
            if (dquant) {
                const double s = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
                dquant[c * 3 + k] = (float)eb_logits_bwd(params + (size_t)c * EB_NPARAM, st, s, NULL);
            }
        }
    return (float)loss;
}

/* ------------------------------------------------------------------------- */
/* GaussianConditional._likelihood + LowerBounds.                              */
/* compressai/entropy_models/entropy_models.py:521-526,570-596.                */
/*   values = |y - mu| ; s = max(sigma, scale_bound)                           */
/*   lik = 0.5 erfc(-(0.5-values)/(s sqrt2)) - 0.5 erfc(-(-0.5-values)/(s sqrt2)) */
/*   lik = max(lik, lik_bound)                                                 */
/* ------------------------------------------------------------------------- */
ORC_API void orc_gc_likelihood_fwd(const float *y, const float *scales, const float *means,
                                   float *lik, size_t n, float scale_bound, float lik_bound)
{
    const double c = -0.70710678118654752440;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float vf = fabsf(y[i] - (means ? means[i] : 0.f));
        const float sf = scales[i] > scale_bound ? scales[i] : scale_bound;
        const float a = (0.5f - vf) / sf, b = (-0.5f - vf) / sf;
        const float up = (float)(0.5 * erfc(c * (double)a));
        const float lo = (float)(0.5 * erfc(c * (double)b));
        const float l = up - lo;
        lik[i] = l > lik_bound ? l : lik_bound;
    }
}

/* backward wrt y, scales, means (any may be NULL) */
ORC_API void orc_gc_likelihood_bwd(const float *y, const float *scales, const float *means,
                                   const float *dlik, float *dy, float *dscales, float *dmeans,
                                   size_t n, float scale_bound, float lik_bound)
{
    const double c = -0.70710678118654752440;
    const double inv_sqrt_2pi = 0.39894228040143267794;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const double d = (double)y[i] - (means ? (double)means[i] : 0.0);
        const double v = fabs(d);
        const double s = scales[i] > scale_bound ? scales[i] : scale_bound;
        const double a = (0.5 - v) / s, b = (-0.5 - v) / s;
        const double up = 0.5 * erfc(c * a), lo = 0.5 * erfc(c * b);
        const float lraw = (float)up - (float)lo;
        double g = dlik[i];
        if (!(lraw >= lik_bound || g < 0)) g = 0.0;
        const double pa = inv_sqrt_2pi * exp(-0.5 * a * a), pb = inv_sqrt_2pi * exp(-0.5 * b * b);
        const double dv = g * (-(pa - pb) / s);
        double ds = g * (-(pa * a - pb * b) / s);
        const double sgn = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
        if (dy) dy[i] = (float)(dv * sgn);
        if (dmeans) dmeans[i] = (float)(-dv * sgn);
        if (dscales) {
            if (!(scales[i] >= scale_bound || ds < 0)) ds = 0.0;
            dscales[i] = (float)ds;
        }
    }
}

/* EntropyModel.quantize "dequantize"/"symbols" (entropy_models.py:137-150):
 * torch.round is round-half-to-even.                                           */
ORC_API void orc_quantize_dequantize(const float *x, const float *means, float *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const float m = means ? means[i] : 0.f;
        out[i] = nearbyintf(x[i] - m) + m;
    }
}
ORC_API void orc_quantize_symbols(const float *x, const float *means, int32_t *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = (int32_t)nearbyintf(x[i] - (means ? means[i] : 0.f));
}

/* GaussianConditional.build_indexes (entropy_models.py:598-604):
 * idx = (T-1) - #{ t in table[:-1] : max(sigma,bound) <= t }                    */
ORC_API void orc_build_indexes(const float *scales, const float *table, int T, int32_t *idx,
                               size_t n, float scale_bound)
{
    for (size_t i = 0; i < n; ++i) {
        const float s = scales[i] > scale_bound ? scales[i] : scale_bound;
        int k = T - 1;
        for (int t = 0; t < T - 1; ++t) k -= (s <= table[t]);
        idx[i] = k;
    }
}

/* EMLoss rate term (utils.py:18-27): sum(log(lik)) / (-ln2 * num_pixels), fp64 */
ORC_API double orc_rate_bpp(const float *lik, size_t n, double num_pixels)
{
    double a = 0.0;
    for (size_t i = 0; i < n; ++i) a += log((double)lik[i]);
    return a / (-log(2.0) * num_pixels);
}

/* ------------------------------------------------------------------------- */
/* pmf_to_quantized_cdf (compressai/cpp_exts/ops/ops.cpp:24-81), integer-exact. */
/* cdf must hold n+1 uint32.                                                   */
/* ------------------------------------------------------------------------- */
ORC_API int orc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf)
{
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -1;
    for (int i = 0; i <= n; ++i)
        cdf[i] = (uint32_t)((((uint64_t)1 << precision) * (uint64_t)cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t f = cdf[j + 1] - cdf[j];
                if (f > 1 && f < best_freq) {
                    best_freq = f;
                    best = j;
                }
            }
            if (best < 0) return -2;
            if (best < i)
                for (int j = best + 1; j <= i; ++j) cdf[j]--;
            else
                for (int j = i + 1; j <= best; ++j) cdf[j]++;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* rANS (third_party/ryg_rans/rans64.h:59-141; 64-bit state, 32-bit renorm,    */
/* L = 2^31) with CompressAI's indexed coding + 4-bit bypass escapes            */
/* (compressai/cpp_exts/rans/rans_interface.cpp:60-275).                         */
/* ------------------------------------------------------------------------- */
#define RANS_L (1ull << 31)
#define RANS_PREC 16
#define BYPASS_PREC 4
#define BYPASS_MAX 15

typedef struct {
    uint16_t start, range;
    uint8_t bypass;
} orc_sym;

/* Encode n symbols.  cdfs is a dense [ncdf][cdf_stride] int32 table.  Returns
 * the number of bytes written to out (capacity cap), or -1 when out is too
 * small.  Bytes are produced exactly as BufferedRansEncoder::flush does.       */
ORC_API long orc_rans_encode(const int32_t *symbols, const int32_t *indexes, size_t n,
                             const int32_t *cdfs, int cdf_stride, const int32_t *sizes,
                             const int32_t *offsets, uint8_t *out, size_t cap)
{
    size_t cap_syms = n * 2 + 16, ns = 0;
    orc_sym *syms = (orc_sym *)malloc(sizeof(orc_sym) * cap_syms);
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = sizes[ci] - 2;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (value < 0) {
            raw = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        if (ns + 24 > cap_syms) {
            cap_syms *= 2;
            syms = (orc_sym *)realloc(syms, sizeof(orc_sym) * cap_syms);
        }
        syms[ns++] = (orc_sym){(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0};
        if (value == max_value) {
            int32_t nb = 0;
            while ((raw >> (nb * BYPASS_PREC)) != 0) ++nb;
            int32_t val = nb;
            while (val >= BYPASS_MAX) {
                syms[ns++] = (orc_sym){BYPASS_MAX, BYPASS_MAX + 1, 1};
                val -= BYPASS_MAX;
                if (ns + 24 > cap_syms) {
                    cap_syms *= 2;
                    syms = (orc_sym *)realloc(syms, sizeof(orc_sym) * cap_syms);
                }
            }
            syms[ns++] = (orc_sym){(uint16_t)val, (uint16_t)(val + 1), 1};
            for (int32_t j = 0; j < nb; ++j) {
                const int32_t v = (raw >> (j * BYPASS_PREC)) & BYPASS_MAX;
                syms[ns++] = (orc_sym){(uint16_t)v, (uint16_t)(v + 1), 1};
            }
        }
    }
    /* flush: encode in reverse into a word buffer that fills from the end */
    const size_t nwords = ns + 4;
    uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * nwords);
    uint32_t *ptr = buf + nwords;
    uint64_t x = RANS_L;
    for (size_t k = ns; k-- > 0;) {
        const orc_sym s = syms[k];
        if (!s.bypass) {
            const uint64_t x_max = ((RANS_L >> RANS_PREC) << 32) * (uint64_t)s.range;
            if (x >= x_max) {
                *--ptr = (uint32_t)x;
                x >>= 32;
            }
            x = ((x / s.range) << RANS_PREC) + (x % s.range) + s.start;
        } else {
            const uint32_t freq = 1u << (16 - BYPASS_PREC);
            const uint64_t x_max = ((RANS_L >> 16) << 32) * (uint64_t)freq;
            if (x >= x_max) {
                *--ptr = (uint32_t)x;
                x >>= 32;
            }
            x = (x << BYPASS_PREC) | s.start;
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    const size_t nbytes = (size_t)(buf + nwords - ptr) * 4;
    long ret = -1;
    if (nbytes <= cap) {
        memcpy(out, ptr, nbytes);
        ret = (long)nbytes;
    }
    free(buf);
    free(syms);
    return ret;
}

static inline uint32_t rans_get_bits(uint64_t *x, const uint32_t **pp, uint32_t nbits)
{
    const uint32_t val = (uint32_t)(*x & ((1u << nbits) - 1));
    *x >>= nbits;
    if (*x < RANS_L) {
        *x = (*x << 32) | **pp;
        *pp += 1;
    }
    return val;
}

/* Decode n symbols (RansDecoder::decode_with_indexes, rans_interface.cpp:206-275). */
ORC_API int orc_rans_decode(const uint8_t *stream, size_t nbytes, const int32_t *indexes, size_t n,
                            const int32_t *cdfs, int cdf_stride, const int32_t *sizes,
                            const int32_t *offsets, int32_t *out)
{
    (void)nbytes;
    const uint32_t *ptr = (const uint32_t *)stream;
    uint64_t x = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
    ptr += 2;
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
        const int32_t max_value = sizes[ci] - 2;
        const uint32_t cum = (uint32_t)(x & ((1u << RANS_PREC) - 1));
        int32_t s = 0;
        while (s + 1 < sizes[ci] && (uint32_t)cdf[s + 1] <= cum) ++s;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = (uint64_t)freq * (x >> RANS_PREC) + (x & ((1ull << RANS_PREC) - 1)) - start;
        if (x < RANS_L) {
            x = (x << 32) | *ptr;
            ptr += 1;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)rans_get_bits(&x, &ptr, BYPASS_PREC);
            int32_t nb = val;
            while (val == BYPASS_MAX) {
                val = (int32_t)rans_get_bits(&x, &ptr, BYPASS_PREC);
                nb += val;
            }
            int32_t raw = 0;
            for (int j = 0; j < nb; ++j) {
                val = (int32_t)rans_get_bits(&x, &ptr, BYPASS_PREC);
                raw |= val << (j * BYPASS_PREC);
            }
            value = raw >> 1;
            if (raw & 1)
                value = -value - 1;
            else
                value += max_value;
        }
        out[i] = value + offsets[ci];
    }
    return 0;
}
