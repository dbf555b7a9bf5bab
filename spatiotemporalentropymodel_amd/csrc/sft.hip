// Elementwise / pooling kernels of the variable-rate (ROI) STEM models, compressai/models/stem_utils.py:24-63:
//   SFT:        out = x * (1 + gamma) + beta                     (stem_utils.py:41)
//   SFTResblk:  leaky_relu(SFT(x), 0.2) feeding a 3x3 conv        (stem_utils.py:55-63) -> fused as one pass
//   adaptive_avg_pool2d of the quality map to a feature resolution (stem_utils.py:37, stem_roi.py:563)
// All HBM-bound: 16 B/element forward (x, gamma, beta in; out), vectorised 16-byte accesses.
#include "stem_common.h"

namespace {

__global__ __launch_bounds__(256) void sft_fwd_kernel(const f32x4 *x, const f32x4 *g, const f32x4 *b, f32x4 *o, size_t n4, float slope)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 xv = x[i], gv = g[i], bv = b[i];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = xv[e] * (1.f + gv[e]) + bv[e];
        r[e] = v > 0.f ? v : v * slope;
    }
    o[i] = r;
}

// d = dout * act'(out) ; dx = d (1 + gamma) ; dgamma = d x ; dbeta = d
__global__ __launch_bounds__(256) void sft_bwd_kernel(const f32x4 *x, const f32x4 *g, const f32x4 *o, const f32x4 *dout, f32x4 *dx,
                                                      f32x4 *dg, f32x4 *db, size_t n4, float slope)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 xv = x[i], gv = g[i], ov = o[i], dv = dout[i];
    f32x4 rx, rg, rb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d = ov[e] > 0.f ? dv[e] : dv[e] * slope;
        rx[e] = d * (1.f + gv[e]);
        rg[e] = d * xv[e];
        rb[e] = d;
    }
    dx[i] = rx;
    dg[i] = rg;
    db[i] = rb;
}

__global__ void lrelu_fwd_kernel(const float *x, float *y, size_t n, float slope)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

// x[B,H,W,C] -> y[B,H/f,W/f,C], mean over f x f windows (adaptive_avg_pool2d when H,W are multiples of the target)
__global__ void avgpool_kernel(const float *x, int ldx, float *y, int ldy, int B, int Ho, int Wo, int C, int fy, int fx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t p = i / C;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho), b = (int)(p / Ho);
    const int W = Wo * fx, H = Ho * fy;
    float s = 0.f;
    for (int r = 0; r < fy; ++r)
        for (int q = 0; q < fx; ++q) s += x[((size_t)(b * H + oy * fy + r) * W + ox * fx + q) * ldx + c];
    y[((size_t)(b * Ho + oy) * Wo + ox) * ldy + c] = s / (float)(fy * fx);
}

// dx[B,H,W,C] = dy[B,H/f,W/f,C] / (fy*fx) broadcast over each window
__global__ void avgpool_bwd_kernel(const float *dy, int ldy, float *dx, int ldx, int B, int H, int W, int C, int fy, int fx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t p = i / C;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H), b = (int)(p / H);
    const int Ho = H / fy, Wo = W / fx;
    dx[((size_t)(b * H + iy) * W + ix) * ldx + c] = dy[((size_t)(b * Ho + iy / fy) * Wo + ix / fx) * ldy + c] / (float)(fy * fx);
}

// sum over [B,C,H,W] of lambda[b,h,w] * (xhat - x)^2, NCHW images (root utils.py:69-71), fp64 accumulation
__global__ __launch_bounds__(256) void wsqerr_sum_kernel(const float *xhat, const float *x, const float *lam, size_t n, int C, size_t HW,
                                                         double *acc)
{
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / (HW * C), p = i % HW;
        const float d = xhat[i] - x[i];
        s += (double)(lam[b * HW + p] * (d * d));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}
// dxhat = g * coef * 2 lambda (xhat - x); g is the upstream scalar gradient, read from device memory
__global__ void wsqerr_bwd_kernel(const float *xhat, const float *x, const float *lam, float *dxhat, size_t n, int C, size_t HW,
                                  const double *g, float coef)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t b = i / (HW * C), p = i % HW;
    dxhat[i] = (float)(*g) * coef * 2.f * lam[b * HW + p] * (xhat[i] - x[i]);
}

inline unsigned nb(size_t n) { return (unsigned)cdivz(n, 256); }
inline bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

}   // namespace

STEM_EXPORT int stem_sft_fwd(const float *x, const float *gamma, const float *beta, float *out, size_t n, float slope, void *stream)
{
    STEM_CHECK_ARG(x && gamma && beta && out, "stem_sft_fwd: null pointer");
    STEM_CHECK_ARG(n % 4 == 0 && al16(x) && al16(gamma) && al16(beta) && al16(out), "stem_sft_fwd: dense 16-byte aligned tensors expected");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sft_fwd_kernel, dim3(nb(n / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)gamma,
                       (const f32x4 *)beta, (f32x4 *)out, n / 4, slope);
    STEM_LAUNCH_CHECK("sft_fwd");
    return 0;
}

STEM_EXPORT int stem_sft_bwd(const float *x, const float *gamma, const float *out, const float *dout, float *dx, float *dgamma,
                             float *dbeta, size_t n, float slope, void *stream)
{
    STEM_CHECK_ARG(x && gamma && out && dout && dx && dgamma && dbeta, "stem_sft_bwd: null pointer");
    STEM_CHECK_ARG(n % 4 == 0 && al16(x) && al16(gamma) && al16(out) && al16(dout) && al16(dx) && al16(dgamma) && al16(dbeta),
                   "stem_sft_bwd: dense 16-byte aligned tensors expected");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sft_bwd_kernel, dim3(nb(n / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)gamma,
                       (const f32x4 *)out, (const f32x4 *)dout, (f32x4 *)dx, (f32x4 *)dgamma, (f32x4 *)dbeta, n / 4, slope);
    STEM_LAUNCH_CHECK("sft_bwd");
    return 0;
}

STEM_EXPORT int stem_lrelu_fwd(const float *x, float *y, size_t n, float slope, void *stream)
{
    STEM_CHECK_ARG(x && y, "stem_lrelu_fwd: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, slope);
    STEM_LAUNCH_CHECK("lrelu_fwd");
    return 0;
}

STEM_EXPORT int stem_avgpool_fwd(const float *x, int ldx, float *y, int ldy, int B, int H, int W, int C, int Ho, int Wo, void *stream)
{
    STEM_CHECK_ARG(x && y && Ho > 0 && Wo > 0, "stem_avgpool_fwd: bad arguments");
    STEM_CHECK_ARG(H % Ho == 0 && W % Wo == 0, "stem_avgpool_fwd: %dx%d is not an integer multiple of %dx%d", H, W, Ho, Wo);
    hipLaunchKernelGGL(avgpool_kernel, dim3(nb((size_t)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, Ho, Wo, C,
                       H / Ho, W / Wo);
    STEM_LAUNCH_CHECK("avgpool");
    return 0;
}

STEM_EXPORT int stem_avgpool_bwd(const float *dy, int ldy, float *dx, int ldx, int B, int H, int W, int C, int Ho, int Wo, void *stream)
{
    STEM_CHECK_ARG(dy && dx && Ho > 0 && Wo > 0, "stem_avgpool_bwd: bad arguments");
    STEM_CHECK_ARG(H % Ho == 0 && W % Wo == 0, "stem_avgpool_bwd: %dx%d is not an integer multiple of %dx%d", H, W, Ho, Wo);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(nb((size_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, ldy, dx, ldx, B, H, W, C,
                       H / Ho, W / Wo);
    STEM_LAUNCH_CHECK("avgpool_bwd");
    return 0;
}

STEM_EXPORT int stem_weighted_sqerr_sum(const float *xhat, const float *x, const float *lambda, int B, int C, size_t HW, double *acc,
                                        void *stream)
{
    STEM_CHECK_ARG(xhat && x && lambda && acc, "stem_weighted_sqerr_sum: null pointer");
    const size_t n = (size_t)B * C * HW;
    if (n == 0) return 0;
    unsigned blocks = nb(n);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(wsqerr_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xhat, x, lambda, n, C, HW, acc);
    STEM_LAUNCH_CHECK("weighted_sqerr_sum");
    return 0;
}

STEM_EXPORT int stem_weighted_sqerr_bwd(const float *xhat, const float *x, const float *lambda, float *dxhat, int B, int C, size_t HW,
                                        const double *g, float coef, void *stream)
{
    STEM_CHECK_ARG(xhat && x && lambda && dxhat && g, "stem_weighted_sqerr_bwd: null pointer");
    const size_t n = (size_t)B * C * HW;
    if (n == 0) return 0;
    hipLaunchKernelGGL(wsqerr_bwd_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, xhat, x, lambda, dxhat, n, C, HW, g, coef);
    STEM_LAUNCH_CHECK("weighted_sqerr_bwd");
    return 0;
}
