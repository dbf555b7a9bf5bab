// Kernels for the autoregressive (raster-order) coding loop of the STEM models with a spatial prior
// (compressai/models/spatiotemporalpriors.py:916-961 encode, :1015-1054 decode).
//
// One spatial position is a chain of four matrix-vector products on a single pixel
//   ctx[2M]   = b_c + W_c[:, 12 live taps x M] . window        (masked 5x5 conv restricted to one output pixel)
//   h1[768]   = lrelu(b_0 + W_0 . (tp | hp | ctx))             (EPM.0, 1x1)
//   h2[576]   = lrelu(b_1 + W_1 . h1)                          (EPM.2)
//   gp[2M]    = b_2 + W_2 . h2                                 (EPM.4)  -> scales | means
// followed by index lookup, quantisation and write-back into the running latent buffer.  Each product
// is HBM/L2-bound (weights are read once per position): one wavefront per output row, 16-byte loads,
// reduction across the 64 lanes with wavefront shuffles.  The input vector is given as up to three
// contiguous segments so neither the 5x5 window nor cat(tp, hp, ctx) is ever materialised.
#include "stem_common.h"

namespace {

struct Seg {
    const float *x;      // segment start
    int len;             // floats (multiple of 4)
    int woff;            // column offset of this segment inside a weight row
};

__global__ __launch_bounds__(256) void gemv3_kernel(const float *W, int ldw, const float *bias, Seg s0, Seg s1, Seg s2,
                                                    float *y, int N, int act, float slope)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = W + (size_t)n * ldw;
    float acc = 0.f;
    const Seg segs[3] = {s0, s1, s2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const Seg s = segs[q];
        for (int k = lane * 4; k < s.len; k += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(s.x + k);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + s.woff + k);
            acc += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        float v = acc + (bias ? bias[n] : 0.f);
        if (act == STEM_ACT_LRELU) v = v > 0.f ? v : v * slope;
        y[n] = v;
    }
}

// encode side: index = build_indexes(scale), symbol = round(target - mean), buffer <- symbol + mean
__global__ void ar_finish_encode_kernel(const float *gp, const float *table, int T, float scale_bound, float *pix,
                                        int32_t *sym, int32_t *idx, int M)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const float s = fmaxf(gp[c], scale_bound), mu = gp[M + c];
    int k = T - 1;
    for (int t = 0; t < T - 1; ++t) k -= (s <= table[t]) ? 1 : 0;
    const float q = rintf(pix[c] - mu);
    pix[c] = q + mu;
    sym[c] = (int32_t)q;
    idx[c] = k;
}
__global__ void ar_index_kernel(const float *gp, const float *table, int T, float scale_bound, int32_t *idx, int M)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const float s = fmaxf(gp[c], scale_bound);
    int k = T - 1;
    for (int t = 0; t < T - 1; ++t) k -= (s <= table[t]) ? 1 : 0;
    idx[c] = k;
}
// decode side: buffer <- symbol + mean  (EntropyModel.dequantize, entropy_models.py:156-163)
__global__ void ar_finish_decode_kernel(const float *gp, const int32_t *sym, float *pix, int M)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    pix[c] = (float)sym[c] + gp[M + c];
}

// masked conv weight [2M][M][5][5] -> [2M][12 live taps][M]  (live taps of the type-A mask: rows 0,1 full, row 2 cols 0,1)
__global__ void pack_ctx_gemv_kernel(const float *w, float *out, int K, int C)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * 12 * C) return;
    const int c = (int)(i % C);
    const int t = (int)((i / C) % 12);
    const int k = (int)(i / ((size_t)12 * C));
    out[i] = w[((size_t)k * C + c) * 25 + t];      // live taps are exactly the first 12 of the 25 in raster order
}

int seg_ok(const Seg &s) { return s.len == 0 || (s.x && (s.len % 4 == 0) && (s.woff % 4 == 0) && (((uintptr_t)s.x & 15) == 0)); }

}   // namespace

STEM_EXPORT int stem_gemv3(const float *W, int ldw, const float *bias, const float *x0, int len0, int woff0,
                           const float *x1, int len1, int woff1, const float *x2, int len2, int woff2, float *y, int N,
                           int act, float slope, void *stream)
{
    STEM_CHECK_ARG(W && y && N > 0 && ldw % 4 == 0 && (((uintptr_t)W & 15) == 0), "stem_gemv3: bad arguments");
    Seg s0{x0, len0, woff0}, s1{x1, len1, woff1}, s2{x2, len2, woff2};
    STEM_CHECK_ARG(seg_ok(s0) && seg_ok(s1) && seg_ok(s2), "stem_gemv3: segments must be 16-byte aligned multiples of 4 floats");
    hipLaunchKernelGGL(gemv3_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, W, ldw, bias, s0, s1, s2, y, N, act, slope);
    STEM_LAUNCH_CHECK("gemv3");
    return 0;
}

STEM_EXPORT int stem_pack_ctx_gemv(const float *w, float *out, int K, int C, void *stream)
{
    STEM_CHECK_ARG(w && out && K > 0 && C > 0, "stem_pack_ctx_gemv: bad arguments");
    const size_t n = (size_t)K * 12 * C;
    hipLaunchKernelGGL(pack_ctx_gemv_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, (hipStream_t)stream, w, out, K, C);
    STEM_LAUNCH_CHECK("pack_ctx_gemv");
    return 0;
}

STEM_EXPORT int stem_ar_finish_encode(const float *gp, const float *table, int T, float scale_bound, float *pix,
                                      int32_t *sym, int32_t *idx, int M, void *stream)
{
    STEM_CHECK_ARG(gp && table && pix && sym && idx && M > 0 && T >= 1, "stem_ar_finish_encode: bad arguments");
    hipLaunchKernelGGL(ar_finish_encode_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, gp, table, T, scale_bound,
                       pix, sym, idx, M);
    STEM_LAUNCH_CHECK("ar_finish_encode");
    return 0;
}

STEM_EXPORT int stem_ar_index(const float *gp, const float *table, int T, float scale_bound, int32_t *idx, int M, void *stream)
{
    STEM_CHECK_ARG(gp && table && idx && M > 0 && T >= 1, "stem_ar_index: bad arguments");
    hipLaunchKernelGGL(ar_index_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, gp, table, T, scale_bound, idx, M);
    STEM_LAUNCH_CHECK("ar_index");
    return 0;
}

STEM_EXPORT int stem_ar_finish_decode(const float *gp, const int32_t *sym, float *pix, int M, void *stream)
{
    STEM_CHECK_ARG(gp && sym && pix && M > 0, "stem_ar_finish_decode: bad arguments");
    hipLaunchKernelGGL(ar_finish_decode_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, gp, sym, pix, M);
    STEM_LAUNCH_CHECK("ar_finish_decode");
    return 0;
}
