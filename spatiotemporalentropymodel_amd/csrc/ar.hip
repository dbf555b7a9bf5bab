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

// Decoder variants of the product (two launches fewer per position):
//   head: the previous position's y_hat = symbol + mean is written back to the latent buffer by workgroup 0 and, when that
//         pixel is the left neighbour inside this window, every wavefront substitutes it on the fly for the (not yet
//         visible) buffer contents -- segment 2 is [pixel w-2 | pixel w-1];
//   tail: the last product also turns each scale into its CDF index (written to the pinned host mailbox).
struct DecodeExtra {
    const int32_t *sym_prev;   // null: plain product
    const float *mean_prev;    // gp + M of the previous position
    float *pix_prev;           // where its y_hat goes in the latent buffer
    int M, prev_is_left;
    const float *table;        // null: no index epilogue
    int T;
    float bound;
    int32_t *idx;
};

__global__ __launch_bounds__(256) void gemv3_decode_kernel(const float *W, int ldw, const float *bias, Seg s0, Seg s1, Seg s2,
                                                           float *y, int N, int act, float slope, DecodeExtra e)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e.sym_prev && blockIdx.x == 0)
        for (int c = threadIdx.x; c < e.M; c += 256) e.pix_prev[c] = (float)e.sym_prev[c] + e.mean_prev[c];
    if (n >= N) return;
    const float *wr = W + (size_t)n * ldw;
    float acc = 0.f;
    const Seg segs[3] = {s0, s1, s2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const Seg s = segs[q];
        const bool subst = q == 2 && e.sym_prev && e.prev_is_left;
        for (int k = lane * 4; k < s.len; k += 256) {
            f32x4 xv;
            if (subst && k >= e.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = (float)e.sym_prev[k - e.M + j] + e.mean_prev[k - e.M + j];
            } else {
                xv = *reinterpret_cast<const f32x4 *>(s.x + k);
            }
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
        if (e.table && n < e.M) {
            const float sc = fmaxf(v, e.bound);
            int k = e.T - 1;
            for (int t = 0; t < e.T - 1; ++t) k -= (sc <= e.table[t]) ? 1 : 0;
            e.idx[n] = k;
        }
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

STEM_EXPORT int stem_gemv3_decode(const float *W, int ldw, const float *bias, const float *x0, int len0, int woff0,
                                  const float *x1, int len1, int woff1, const float *x2, int len2, int woff2, float *y, int N,
                                  int act, float slope, const int32_t *sym_prev, const float *mean_prev, float *pix_prev, int M,
                                  int prev_is_left, const float *table, int T, float scale_bound, int32_t *idx, void *stream)
{
    STEM_CHECK_ARG(W && y && N > 0 && ldw % 4 == 0 && (((uintptr_t)W & 15) == 0) && M % 4 == 0, "stem_gemv3_decode: bad arguments");
    Seg s0{x0, len0, woff0}, s1{x1, len1, woff1}, s2{x2, len2, woff2};
    STEM_CHECK_ARG(seg_ok(s0) && seg_ok(s1) && seg_ok(s2), "stem_gemv3_decode: segments must be 16-byte aligned multiples of 4 floats");
    STEM_CHECK_ARG(!sym_prev || (mean_prev && pix_prev && (!prev_is_left || len2 == 2 * M)), "stem_gemv3_decode: inconsistent write-back arguments");
    STEM_CHECK_ARG(!table || (idx && T >= 1 && M <= N), "stem_gemv3_decode: inconsistent index arguments");
    DecodeExtra e{sym_prev, mean_prev, pix_prev, M, prev_is_left, table, T, scale_bound, idx};
    hipLaunchKernelGGL(gemv3_decode_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, W, ldw, bias, s0, s1, s2, y, N, act, slope, e);
    STEM_LAUNCH_CHECK("gemv3_decode");
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

// ================================================================================================
// Wavefront-parallel ENCODER.  With a 5x5 type-A mask and raster order, position (h, w) depends on
// (h, w-1..w-2) and on rows h-1, h-2 at columns w-2..w+2, so all positions with the same t = w + 3h are
// mutually independent: a frame of H x W latents needs W + 3(H-1) steps (321 for 68 x 120) instead of H*W
// (8160), each step a *batch* of up to min(H, W/3) pixels.  Symbols and indexes land in raster order, so the
// host coder produces the same bytes as the sequential loop.  (The decoder cannot do this: the rANS stream
// itself is sequential in raster order.)
//
// Batched matrix-vector product: one wavefront per output row n, looping over the step's positions so the
// weight row stays in L1; the input of position p is up to three segments at offsets sh*h + sw*w + sp*p.
namespace {

struct WSeg {
    const float *x;
    int len, woff;
    long sh, sw, sp;
};

__device__ __forceinline__ void wave_range(int t, int H, int Wd, int &h0, int &np)
{
    int lo = t - (Wd - 1);
    lo = lo > 0 ? (lo + 2) / 3 : 0;
    int hi = t / 3;
    if (hi > H - 1) hi = H - 1;
    h0 = lo;
    np = hi - lo + 1;
}

constexpr int WAVE_ROWS = 24;     // workgroup rows over which the positions of one wavefront step are spread (more waves in flight)

__global__ __launch_bounds__(256) void gemv3_wave_kernel(const float *W, int ldw, const float *bias, WSeg s0, WSeg s1, WSeg s2,
                                                         float *y, int ldy, int N, int act, float slope, int t, int H, int Wd)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    int h0, np;
    wave_range(t, H, Wd, h0, np);
    const float *wr = W + (size_t)n * ldw;
    const float b = bias ? bias[n] : 0.f;
    const WSeg segs[3] = {s0, s1, s2};
    for (int p = blockIdx.y; p < np; p += gridDim.y) {      // positions of the step are spread over gridDim.y workgroup rows
        const int h = h0 + p, w = t - 3 * h;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const WSeg s = segs[q];
            const float *xp = s.x + s.sh * h + s.sw * w + s.sp * p;
            for (int k = lane * 4; k < s.len; k += 256) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xp + k);
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + s.woff + k);
                acc += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) {
            float v = acc + b;
            if (act == STEM_ACT_LRELU) v = v > 0.f ? v : v * slope;
            y[(size_t)p * ldy + n] = v;
        }
    }
}

__global__ void ar_finish_encode_wave_kernel(const float *gp, const float *table, int T, float scale_bound, float *buf,
                                             int32_t *sym, int32_t *idx, int M, int t, int H, int Wd, int Wp, int pad)
{
    int h0, np;
    wave_range(t, H, Wd, h0, np);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np * M) return;
    const int p = i / M, c = i - p * M;
    const int h = h0 + p, w = t - 3 * h;
    const float *g = gp + (size_t)p * 2 * M;
    const float s = fmaxf(g[c], scale_bound), mu = g[M + c];
    int k = T - 1;
    for (int q = 0; q < T - 1; ++q) k -= (s <= table[q]) ? 1 : 0;
    float *pix = buf + ((size_t)(h + pad) * Wp + (w + pad)) * M;
    const float qv = rintf(pix[c] - mu);
    pix[c] = qv + mu;
    const size_t o = ((size_t)h * Wd + w) * M + c;
    sym[o] = (int32_t)qv;
    idx[o] = k;
}

}   // namespace

STEM_EXPORT int stem_gemv3_wave(const float *W, int ldw, const float *bias, const stem_wave_seg *segs, float *y, int ldy, int N,
                                int act, float slope, int t, int H, int Wd, void *stream)
{
    STEM_CHECK_ARG(W && segs && y && N > 0 && ldw % 4 == 0, "stem_gemv3_wave: bad arguments");
    WSeg s[3];
    for (int i = 0; i < 3; ++i) {
        s[i].x = segs[i].x; s[i].len = segs[i].len; s[i].woff = segs[i].woff;
        s[i].sh = segs[i].sh; s[i].sw = segs[i].sw; s[i].sp = segs[i].sp;
        STEM_CHECK_ARG(s[i].len == 0 || (s[i].x && s[i].len % 4 == 0 && s[i].woff % 4 == 0 && s[i].sh % 4 == 0 && s[i].sw % 4 == 0 && s[i].sp % 4 == 0),
                       "stem_gemv3_wave: segment %d is not 16-byte granular", i);
    }
    const int maxp = H < (Wd + 2) / 3 ? H : (Wd + 2) / 3;
    hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(N, 4), maxp < WAVE_ROWS ? maxp : WAVE_ROWS), dim3(256), 0, (hipStream_t)stream, W, ldw, bias,
                       s[0], s[1], s[2], y, ldy, N, act, slope, t, H, Wd);
    STEM_LAUNCH_CHECK("gemv3_wave");
    return 0;
}

STEM_EXPORT int stem_ar_finish_encode_wave(const float *gp, const float *table, int T, float scale_bound, float *buf,
                                           int32_t *sym, int32_t *idx, int M, int t, int H, int Wd, int Wp, int pad, void *stream)
{
    STEM_CHECK_ARG(gp && table && buf && sym && idx && M > 0 && T >= 1, "stem_ar_finish_encode_wave: bad arguments");
    const int maxp = H < (Wd + 2) / 3 ? H : (Wd + 2) / 3;
    hipLaunchKernelGGL(ar_finish_encode_wave_kernel, dim3(cdiv(maxp * M, 256)), dim3(256), 0, (hipStream_t)stream, gp, table, T,
                       scale_bound, buf, sym, idx, M, t, H, Wd, Wp, pad);
    STEM_LAUNCH_CHECK("ar_finish_encode_wave");
    return 0;
}

// ================================================================================================
// DECODER, whole image: the raster-order loop of spatiotemporalpriors.py:1015-1054 as ONE C-ABI call.  Per position: four
// launches (context + write-back of the previous pixel, EPM.0, EPM.2, EPM.4 + CDF indexes), one stream synchronisation,
// and one call of the host symbol decoder (a C function pointer -- stem_rans_decoder_decode of libstem_rans -- so the two
// libraries stay independent) through the pinned mailbox.  No interpreter in the loop: ~30 us per position instead of ~45.
STEM_EXPORT int stem_ar_decode_image(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                                     const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                                     float *buf, int H, int W, int M, int pad, const float *tp, const float *hp,
                                     float *ctx, float *h1, float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                                     int32_t *idx_host, int32_t *sym_host, stem_symbol_decoder_fn decode, void *dec,
                                     const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, void *stream)
{
    STEM_CHECK_ARG(w_ctx && b_ctx && w0 && b0 && w1 && b1 && w2 && b2 && buf && hp && ctx && h1 && h2 && gp && table && idx_host && sym_host && decode,
                   "stem_ar_decode_image: null pointer");
    STEM_CHECK_ARG(H > 0 && W > 0 && M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && ld_ctx % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 &&
                   ld2 % 4 == 0 && T >= 1 && pad == 2, "stem_ar_decode_image: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, Wp = W + 2 * pad;
    float *pix_prev = nullptr;
    const DecodeExtra none{nullptr, nullptr, nullptr, M, 0, nullptr, 0, 0.f, nullptr};
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            const size_t pos = (size_t)h * W + w;
            const float *r0 = buf + ((size_t)h * Wp + w) * M, *r1 = r0 + (size_t)Wp * M, *r2 = r1 + (size_t)Wp * M;
            DecodeExtra head = none;
            if (pix_prev) {
                head.sym_prev = sym_host; head.mean_prev = gp + M; head.pix_prev = pix_prev; head.prev_is_left = w > 0 ? 1 : 0;
            }
            hipLaunchKernelGGL(gemv3_decode_kernel, dim3(cdiv(P, 4)), dim3(256), 0, st, w_ctx, ld_ctx, b_ctx, Seg{r0, 5 * M, 0}, Seg{r1, 5 * M, 5 * M},
                               Seg{r2, 2 * M, 10 * M}, ctx, P, 0, 0.f, head);
            const float *hp_pix = hp + pos * P;
            if (tp)
                hipLaunchKernelGGL(gemv3_kernel, dim3(cdiv(n0, 4)), dim3(256), 0, st, w0, ld0, b0, Seg{tp + pos * P, P, 0}, Seg{hp_pix, P, P},
                                   Seg{ctx, P, 2 * P}, h1, n0, (int)STEM_ACT_LRELU, slope);
            else
                hipLaunchKernelGGL(gemv3_kernel, dim3(cdiv(n0, 4)), dim3(256), 0, st, w0, ld0, b0, Seg{hp_pix, P, 0}, Seg{ctx, P, P},
                                   Seg{nullptr, 0, 0}, h1, n0, (int)STEM_ACT_LRELU, slope);
            hipLaunchKernelGGL(gemv3_kernel, dim3(cdiv(n1, 4)), dim3(256), 0, st, w1, ld1, b1, Seg{h1, n0, 0}, Seg{nullptr, 0, 0},
                               Seg{nullptr, 0, 0}, h2, n1, (int)STEM_ACT_LRELU, slope);
            DecodeExtra tail = none;
            tail.table = table; tail.T = T; tail.bound = scale_bound; tail.idx = idx_host;
            hipLaunchKernelGGL(gemv3_decode_kernel, dim3(cdiv(P, 4)), dim3(256), 0, st, w2, ld2, b2, Seg{h2, n1, 0}, Seg{nullptr, 0, 0},
                               Seg{nullptr, 0, 0}, gp, P, 0, 0.f, tail);
            if (hipStreamSynchronize(st) != hipSuccess) {
                stem_set_error("stem_ar_decode_image: device error at position (%d, %d): %s", h, w, hipGetErrorString(hipGetLastError()));
                return -2;
            }
            if (int rc = decode(dec, idx_host, (size_t)M, cdfs, ncdf, cdf_stride, sizes, offsets, sym_host)) {
                stem_set_error("stem_ar_decode_image: host symbol decoder failed (%d) at position (%d, %d)", rc, h, w);
                return -3;
            }
            pix_prev = buf + ((size_t)(h + pad) * Wp + (w + pad)) * M;
        }
    hipLaunchKernelGGL(ar_finish_decode_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, gp, sym_host, pix_prev, M);
    STEM_LAUNCH_CHECK("ar_decode_image");
    return 0;
}

// ENCODER, whole image: the W + 3(H-1) wavefront steps of the section above queued by one C-ABI call (5 launches per step,
// no synchronisation): the host then makes a single rANS call on the raster-ordered symbols / indexes.
STEM_EXPORT int stem_ar_encode_image(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                                     const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                                     float *buf, int H, int W, int M, int pad, const float *tp, const float *hp,
                                     float *wctx, float *wh1, float *wh2, float *wgp, const float *table, int T, float scale_bound,
                                     float slope, int32_t *sym, int32_t *idx, void *stream)
{
    STEM_CHECK_ARG(w_ctx && b_ctx && w0 && b0 && w1 && b1 && w2 && b2 && buf && hp && wctx && wh1 && wh2 && wgp && table && sym && idx,
                   "stem_ar_encode_image: null pointer");
    STEM_CHECK_ARG(H > 0 && W > 0 && M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && pad == 2 && T >= 1, "stem_ar_encode_image: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, Wp = W + 2 * pad;
    const long row = (long)Wp * M;
    const int maxp = H < (W + 2) / 3 ? H : (W + 2) / 3;
    const WSeg none{nullptr, 0, 0, 0, 0, 0};
    // context window of position (h, w): rows h, h+1 (5 pixels) and h+2 (2 pixels) of the padded buffer, starting at column w
    const WSeg c0{buf, 5 * M, 0, row, M, 0}, c1{buf + row, 5 * M, 5 * M, row, M, 0}, c2{buf + 2 * row, 2 * M, 10 * M, row, M, 0};
    const WSeg sctx{wctx, P, tp ? 2 * P : P, 0, 0, P};
    const WSeg stp{tp, P, 0, (long)W * P, P, 0}, shp{hp, P, tp ? P : 0, (long)W * P, P, 0};
    const WSeg sh1{wh1, n0, 0, 0, 0, n0}, sh2{wh2, n1, 0, 0, 0, n1};
    const int gy = maxp < WAVE_ROWS ? maxp : WAVE_ROWS;
    for (int t = 0; t < W + 3 * (H - 1); ++t) {
        hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(P, 4), gy), dim3(256), 0, st, w_ctx, ld_ctx, b_ctx, c0, c1, c2, wctx, P, P, 0, 0.f, t, H, W);
        if (tp)
            hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(n0, 4), gy), dim3(256), 0, st, w0, ld0, b0, stp, shp, sctx, wh1, n0, n0,
                               (int)STEM_ACT_LRELU, slope, t, H, W);
        else
            hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(n0, 4), gy), dim3(256), 0, st, w0, ld0, b0, shp, sctx, none, wh1, n0, n0,
                               (int)STEM_ACT_LRELU, slope, t, H, W);
        hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(n1, 4), gy), dim3(256), 0, st, w1, ld1, b1, sh1, none, none, wh2, n1, n1,
                           (int)STEM_ACT_LRELU, slope, t, H, W);
        hipLaunchKernelGGL(gemv3_wave_kernel, dim3(cdiv(P, 4), gy), dim3(256), 0, st, w2, ld2, b2, sh2, none, none, wgp, P, P, 0, 0.f, t, H, W);
        hipLaunchKernelGGL(ar_finish_encode_wave_kernel, dim3(cdiv(maxp * M, 256)), dim3(256), 0, st, wgp, table, T, scale_bound, buf, sym, idx,
                           M, t, H, W, Wp, pad);
    }
    STEM_LAUNCH_CHECK("ar_encode_image");
    return 0;
}
