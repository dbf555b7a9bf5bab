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
#include <chrono>

#include "stem_common.h"

// Encoder and decoder must produce the SAME floats for every entropy parameter (a mean that differs in the last bit
// shifts y_hat, which feeds later contexts; a scale on the other side of a table entry desynchronises the coder), and the
// encoder's wavefront kernels, the one-image decoder and the lockstep decoder are different kernels.  Their dot products
// are therefore written once (dot4) and compiled without FMA contraction: products rounded, then added left to right.
#pragma clang fp contract(off)

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

// ------------------------------------------------------------------------------------------------------------------
// DECODER, G images in lockstep.  The raster-order chain (the context of a position needs the previous position's symbols,
// which needed that position's entropy parameters ...) makes ONE image latency-bound: ~35 us per position for four
// dependent dispatches + a host round trip, whatever the arithmetic costs.  Independent images (the batch elements of
// decompress(), i.e. different sequences / GOPs: spatiotemporalpriors.py:1015-1054 loops over them one after the other)
// share that latency: the same four launches advance all G images by one position -- every wavefront still owns one
// output row, loads its weight row ONCE and accumulates G dot products in the single-image order, so every image's floats
// (hence symbols, indexes and bytes) are exactly those of stem_ar_decode_image.
namespace {

constexpr int GMAX = 8;
struct SegB {
    const float *x;      // image 0
    int len, woff;
    long stride;         // floats between consecutive images
};
struct DecodeExtraB {
    const int32_t *sym_prev;   // [G][M] (pinned host), null: plain product
    const float *mean_prev;    // gp + M of image 0, image stride gp_stride
    float *pix_prev;           // image 0, image stride buf_stride
    long gp_stride, buf_stride;
    int M, prev_is_left;
    const float *table;
    int T;
    float bound;
    int32_t *idx;              // [G][M] (pinned host)
    // pipelined loop (stem_ar_decode_batch with STEM_AR_PIPELINE): the launch that writes `idx` also tells the host, through
    // a flag in pinned memory, that this position's indexes are complete; every launch returns at once after an abort
    int *cnt;                  // device arrival counter of this launch (zero before and after)
    int *flag_idx;             // pinned host word, null: no signalling
    int flag_val;
    const int *abort_dev;      // device word, non-zero: a wait timed out or the host gave up
};

// One wavefront per output row (the products are latency-bound: what counts is how many independent loads are in flight,
// row-blocked variants with fewer wavefronts were 2x slower).  G is a template parameter and each segment is walked in up to
// MAXS fully unrolled 256-column steps, so that the weight loads of a segment, then each image's x loads, are issued back to
// back instead of one exposed latency per step.  Every (row, image) sum accumulates its steps in ascending k: the order --
// and therefore the bits -- of gemv3_decode_kernel.
constexpr int MAXS = 4;          // segments of up to 1024 floats (5 M = 960 for M = 192, n0 = 768)
template <int G>
__global__ __launch_bounds__(256) void gemv3b_decode_kernel(const float *W, int ldw, const float *bias, SegB s0, SegB s1, SegB s2,
                                                            float *y, long ystride, int N, int act, float slope, DecodeExtraB e)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e.abort_dev && *reinterpret_cast<const volatile int *>(e.abort_dev)) return;
    if (n >= N) return;
    const float *wr = W + (size_t)n * ldw;
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
    const SegB segs[3] = {s0, s1, s2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const SegB s = segs[q];
        if (s.len == 0) continue;
        f32x4 wv[MAXS];
        bool ok[MAXS];
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            const int k = lane * 4 + 256 * t;
            ok[t] = k < s.len;
            wv[t] = *reinterpret_cast<const f32x4 *>(wr + s.woff + (ok[t] ? k : 0));
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 xv[MAXS];
#pragma unroll
            for (int t = 0; t < MAXS; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(s.x + g * s.stride + (ok[t] ? lane * 4 + 256 * t : 0));
#pragma unroll
            for (int t = 0; t < MAXS; ++t) {
                const float d = xv[t][0] * wv[t][0] + xv[t][1] * wv[t][1] + xv[t][2] * wv[t][2] + xv[t][3] * wv[t][3];
                acc[g] = ok[t] ? acc[g] + d : acc[g];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float a = acc[g];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) {
            float v = a + (bias ? bias[n] : 0.f);
            if (act == STEM_ACT_LRELU) v = v > 0.f ? v : v * slope;
            y[g * ystride + n] = v;
            if (e.table && n < e.M) {
                const float sc = fmaxf(v, e.bound);
                int k = e.T - 1;
                for (int t = 0; t < e.T - 1; ++t) k -= (sc <= e.table[t]) ? 1 : 0;
                if (e.flag_idx) __hip_atomic_store(e.idx + g * e.M + n, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // write-through
                else e.idx[g * e.M + n] = k;
            }
        }
    }
    if (e.flag_idx) {      // N is a multiple of 4 here (checked by the host): whole workgroups reach the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's index words have left for the host (no L2-wide fence)
        __syncthreads();
        if (threadIdx.x == 0) {
            const int ticket = atomicAdd(e.cnt, 1);
            if (ticket == (int)gridDim.x - 1) {
                __hip_atomic_store(e.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(e.flag_idx, e.flag_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Pipelined loop: y_hat of the previous position = symbol + mean, as soon as the host has posted the symbols.  One workgroup;
// thread 0 polls the host's flag in pinned memory (bounded: ~1 s, then the abort word stops every later launch).
__global__ __launch_bounds__(256) void ar_commit_wait_kernel(const float *gp, long gp_stride, const int32_t *sym, float *pix, long buf_stride,
                                                             int M, int G, const int *flag_sym, int need, int *abort_dev)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = *reinterpret_cast<volatile int *>(abort_dev) ? 0 : 1;
        long spins = 0;
        while (good) {
            const int v = __hip_atomic_load(flag_sym, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v >= need) break;
            if (v < 0 || ++spins > 2000000) {          // host gave up / ~1 s without an answer
                good = 0;
                __hip_atomic_store(abort_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    for (int i = threadIdx.x; i < M * G; i += 256) {
        const int g = i / M, c = i - g * M;
        const int sv = __hip_atomic_load(sym + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        pix[g * buf_stride + c] = (float)sv + gp[g * gp_stride + M + c];
    }
}

template <int G>
void launch_gemv3b(hipStream_t st, int N, const float *W, int ldw, const float *bias, SegB a, SegB b, SegB c, float *y, long ystride, int act,
                   float slope, const DecodeExtraB &e)
{
    hipLaunchKernelGGL((gemv3b_decode_kernel<G>), dim3(cdiv(N, 4)), dim3(256), 0, st, W, ldw, bias, a, b, c, y, ystride, N, act, slope, e);
}
void launch_gemv3b_g(int G, hipStream_t st, int N, const float *W, int ldw, const float *bias, SegB a, SegB b, SegB c, float *y, long ystride,
                     int act, float slope, const DecodeExtraB &e)
{
    switch (G) {
    case 1: launch_gemv3b<1>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 2: launch_gemv3b<2>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 3: launch_gemv3b<3>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 4: launch_gemv3b<4>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 5: launch_gemv3b<5>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 6: launch_gemv3b<6>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    case 7: launch_gemv3b<7>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    default: launch_gemv3b<8>(st, N, W, ldw, bias, a, b, c, y, ystride, act, slope, e); break;
    }
}

__global__ void ar_finish_decode_batch_kernel(const float *gp, long gp_stride, const int32_t *sym, float *pix, long buf_stride, int M, int G)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * G) return;
    const int g = i / M, c = i - g * M;
    pix[g * buf_stride + c] = (float)sym[g * M + c] + gp[g * gp_stride + M + c];
}

}   // namespace

STEM_EXPORT int stem_ar_decode_batch(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                                     const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                                     float *buf, int G, int H, int W, int M, int pad, const float *tp, const float *hp,
                                     float *ctx, float *h1, float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                                     int32_t *idx_host, int32_t *sym_host, stem_symbol_decoder_fn decode, void *const *decs,
                                     const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, void *stream)
{
    STEM_CHECK_ARG(w_ctx && b_ctx && w0 && b0 && w1 && b1 && w2 && b2 && buf && hp && ctx && h1 && h2 && gp && table && idx_host && sym_host && decode && decs,
                   "stem_ar_decode_batch: null pointer");
    STEM_CHECK_ARG(G >= 1 && G <= GMAX, "stem_ar_decode_batch: 1..%d images per call, got %d", GMAX, G);
    STEM_CHECK_ARG(5 * M <= 256 * MAXS && n0 <= 256 * MAXS && n1 <= 256 * MAXS, "stem_ar_decode_batch: segments longer than %d floats (M=%d n0=%d n1=%d)",
                   256 * MAXS, M, n0, n1);
    STEM_CHECK_ARG(H > 0 && W > 0 && M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && ld_ctx % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 &&
                   ld2 % 4 == 0 && T >= 1 && pad == 2, "stem_ar_decode_batch: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, Wp = W + 2 * pad;
    const long bufs = (long)(H + 2 * pad) * Wp * M, pris = (long)H * W * P;        // image strides of buf and of tp / hp
    float *pix_prev = nullptr;
    static const bool prof = getenv("STEM_AR_PROFILE") != nullptr;         // where a position's time goes (launch / wait / host coder)
    double t_launch = 0, t_wait = 0, t_host = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    DecodeExtraB none;
    memset(&none, 0, sizeof(none));
    none.M = M; none.gp_stride = P; none.buf_stride = bufs;
    const SegB nil{nullptr, 0, 0, 0};
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            const size_t pos = (size_t)h * W + w;
            const double ta = prof ? now() : 0;
            const float *r0 = buf + ((size_t)h * Wp + w) * M, *r1 = r0 + (size_t)Wp * M, *r2 = r1 + (size_t)Wp * M;
            // The previous position's symbols arrive in the pinned host mailbox.  The one-image loop lets every wavefront of
            // the context product substitute them on the fly (saving a launch); with G images that is 2M x G floats fetched
            // over PCIe by each of the 2M rows (measured: 90 of 110 us per step at G = 8), so here a one-workgroup kernel
            // commits y_hat = symbol + mean to the latent buffers first and the product reads device memory only.
            const DecodeExtraB head = none;
            if (pix_prev)
                hipLaunchKernelGGL(ar_finish_decode_batch_kernel, dim3(cdiv(M * G, 256)), dim3(256), 0, st, gp, (long)P, sym_host, pix_prev, bufs, M, G);
            launch_gemv3b_g(G, st, P, w_ctx, ld_ctx, b_ctx, SegB{r0, 5 * M, 0, bufs}, SegB{r1, 5 * M, 5 * M, bufs}, SegB{r2, 2 * M, 10 * M, bufs},
                            ctx, (long)P, 0, 0.f, head);
            const float *hp_pix = hp + pos * P;
            if (tp)
                launch_gemv3b_g(G, st, n0, w0, ld0, b0, SegB{tp + pos * P, P, 0, pris}, SegB{hp_pix, P, P, pris}, SegB{ctx, P, 2 * P, (long)P},
                                h1, (long)n0, (int)STEM_ACT_LRELU, slope, none);
            else
                launch_gemv3b_g(G, st, n0, w0, ld0, b0, SegB{hp_pix, P, 0, pris}, SegB{ctx, P, P, (long)P}, nil, h1, (long)n0,
                                (int)STEM_ACT_LRELU, slope, none);
            launch_gemv3b_g(G, st, n1, w1, ld1, b1, SegB{h1, n0, 0, (long)n0}, nil, nil, h2, (long)n1, (int)STEM_ACT_LRELU, slope, none);
            DecodeExtraB tail = none;
            tail.table = table; tail.T = T; tail.bound = scale_bound; tail.idx = idx_host;
            launch_gemv3b_g(G, st, P, w2, ld2, b2, SegB{h2, n1, 0, (long)n1}, nil, nil, gp, (long)P, 0, 0.f, tail);
            const double tb = prof ? now() : 0;
            if (hipStreamSynchronize(st) != hipSuccess) {
                stem_set_error("stem_ar_decode_batch: device error at position (%d, %d): %s", h, w, hipGetErrorString(hipGetLastError()));
                return -2;
            }
            const double tc = prof ? now() : 0;
            for (int g = 0; g < G; ++g)
                if (int rc = decode(decs[g], idx_host + (size_t)g * M, (size_t)M, cdfs, ncdf, cdf_stride, sizes, offsets, sym_host + (size_t)g * M)) {
                    stem_set_error("stem_ar_decode_batch: host symbol decoder failed (%d) for image %d at position (%d, %d)", rc, g, h, w);
                    return -3;
                }
            pix_prev = buf + ((size_t)(h + pad) * Wp + (w + pad)) * M;
            if (prof) {
                const double td = now();
                t_launch += tb - ta; t_wait += tc - tb; t_host += td - tc;
            }
        }
    if (prof)
        fprintf(stderr, "[ar decode batch] G=%d positions=%d: launch %.1f us, wait %.1f us, host coder %.1f us per position step\n", G, H * W,
                t_launch / (H * W), t_wait / (H * W), t_host / (H * W));
    hipLaunchKernelGGL(ar_finish_decode_batch_kernel, dim3(cdiv(M * G, 256)), dim3(256), 0, st, gp, (long)P, sym_host, pix_prev, bufs, M, G);
    STEM_LAUNCH_CHECK("ar_decode_batch");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The lockstep decoder without a stream synchronisation per position.  stem_ar_decode_batch pays, per position step, the
// launches (13 us of host time), a hipStreamSynchronize and the host coder one after the other.  Here
//   * the GPU tells the host that a position's indexes are complete by writing a flag into pinned memory (last workgroup of
//     the launch that produces them), the host answers the same way after decoding, and the kernel that commits the symbols
//     (first launch of the next position) polls that flag: no synchronisation call, and the launches of position p + 2 are
//     issued while the GPU works on p + 1;
//   * the images are split into two groups that alternate on the stream, so that the host decodes group A's symbols while
//     the GPU advances group B.
// Mailboxes are double-buffered per group (slot = position & 1).  Every wait is bounded: a GPU-side poll gives up after ~1 s
// and raises a device abort word that makes all later launches return at once; the host gives up after 5 s and posts a
// negative flag.  Arithmetic per image is unchanged (same kernels), so are the decoded symbols.
namespace {

struct PipeState {
    int *pinned = nullptr;       // [0..1] flag_idx per group, [2..3] flag_sym per group, then idx / sym mailboxes
    int *dev = nullptr;          // [0..3] arrival counters (group x slot), [4] abort word
    size_t pinned_ints = 0;
};
thread_local PipeState g_pipe;

int pipe_reserve(size_t mailbox_ints)
{
    const size_t need = 16 + 2 * 2 * 2 * mailbox_ints;          // flags + {idx, sym} x 2 groups x 2 slots
    if (g_pipe.pinned_ints < need) {
        if (g_pipe.pinned) (void)hipHostFree(g_pipe.pinned);
        g_pipe.pinned = nullptr;
        if (hipHostMalloc((void **)&g_pipe.pinned, need * sizeof(int), hipHostMallocDefault) != hipSuccess) return -1;
        g_pipe.pinned_ints = need;
    }
    if (!g_pipe.dev && hipMalloc((void **)&g_pipe.dev, 8 * sizeof(int)) != hipSuccess) return -1;
    return 0;
}

}   // namespace

STEM_EXPORT int stem_ar_decode_batch_pipelined(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0,
                                               int n0, const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2,
                                               const float *b2, float *buf, int G, int H, int W, int M, int pad, const float *tp,
                                               const float *hp, float *ctx, float *h1, float *h2, float *gp, const float *table, int T,
                                               float scale_bound, float slope, stem_symbol_decoder_fn decode, void *const *decs,
                                               const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets,
                                               void *stream)
{
    STEM_CHECK_ARG(w_ctx && b_ctx && w0 && b0 && w1 && b1 && w2 && b2 && buf && hp && ctx && h1 && h2 && gp && table && decode && decs,
                   "stem_ar_decode_batch_pipelined: null pointer");
    STEM_CHECK_ARG(G >= 1 && G <= GMAX, "stem_ar_decode_batch_pipelined: 1..%d images per call, got %d", GMAX, G);
    STEM_CHECK_ARG(5 * M <= 256 * MAXS && n0 <= 256 * MAXS && n1 <= 256 * MAXS, "stem_ar_decode_batch_pipelined: segments longer than %d floats",
                   256 * MAXS);
    STEM_CHECK_ARG(H > 0 && W > 0 && M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && ld_ctx % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 &&
                   ld2 % 4 == 0 && T >= 1 && pad == 2, "stem_ar_decode_batch_pipelined: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, Wp = W + 2 * pad, N = H * W;
    const long bufs = (long)(H + 2 * pad) * Wp * M, pris = (long)H * W * P;
    const int ng = G >= 2 ? 2 : 1;
    const int gsz[2] = {ng == 2 ? (G + 1) / 2 : G, ng == 2 ? G / 2 : 0}, gbeg[2] = {0, gsz[0]};
    const size_t mb = (size_t)GMAX * M;                          // ints per mailbox
    if (pipe_reserve(mb)) {
        stem_set_error("stem_ar_decode_batch_pipelined: cannot allocate the pinned mailboxes");
        return -2;
    }
    int *pin = g_pipe.pinned, *dev = g_pipe.dev;
    auto mbox = [&](int kind, int k, int slot) { return pin + 16 + ((size_t)(kind * 2 + k) * 2 + slot) * mb; };      // kind 0 idx, 1 sym
    for (int i = 0; i < 4; ++i) pin[i] = 0;
    if (hipMemsetAsync(dev, 0, 8 * sizeof(int), st) != hipSuccess) return -2;
    int *abort_dev = dev + 4;

    DecodeExtraB none;
    memset(&none, 0, sizeof(none));
    none.M = M; none.gp_stride = P; none.buf_stride = bufs; none.abort_dev = abort_dev;
    const SegB nil{nullptr, 0, 0, 0};
    auto pix_of = [&](int k, int p) { return buf + (size_t)gbeg[k] * bufs + ((size_t)(p / W + pad) * Wp + (p % W + pad)) * M; };
    auto commit = [&](int k, int p) {      // y_hat of position p of group k, once the host has posted its symbols (flag_sym >= p + 1)
        hipLaunchKernelGGL(ar_commit_wait_kernel, dim3(1), dim3(256), 0, st, gp + (size_t)gbeg[k] * P, (long)P, mbox(1, k, p & 1), pix_of(k, p),
                           bufs, M, gsz[k], (const int *)(pin + 2 + k), p + 1, abort_dev);
    };
    auto chain = [&](int k, int p) {
        const int Gk = gsz[k], g0 = gbeg[k], h = p / W, w = p % W;
        if (p > 0) commit(k, p - 1);
        const float *r0 = buf + (size_t)g0 * bufs + ((size_t)h * Wp + w) * M, *r1 = r0 + (size_t)Wp * M, *r2 = r1 + (size_t)Wp * M;
        float *ctx_k = ctx + (size_t)g0 * P, *h1_k = h1 + (size_t)g0 * n0, *h2_k = h2 + (size_t)g0 * n1, *gp_k = gp + (size_t)g0 * P;
        launch_gemv3b_g(Gk, st, P, w_ctx, ld_ctx, b_ctx, SegB{r0, 5 * M, 0, bufs}, SegB{r1, 5 * M, 5 * M, bufs}, SegB{r2, 2 * M, 10 * M, bufs},
                        ctx_k, (long)P, 0, 0.f, none);
        const float *hp_pix = hp + (size_t)g0 * pris + (size_t)p * P;
        if (tp)
            launch_gemv3b_g(Gk, st, n0, w0, ld0, b0, SegB{tp + (size_t)g0 * pris + (size_t)p * P, P, 0, pris}, SegB{hp_pix, P, P, pris},
                            SegB{ctx_k, P, 2 * P, (long)P}, h1_k, (long)n0, (int)STEM_ACT_LRELU, slope, none);
        else
            launch_gemv3b_g(Gk, st, n0, w0, ld0, b0, SegB{hp_pix, P, 0, pris}, SegB{ctx_k, P, P, (long)P}, nil, h1_k, (long)n0,
                            (int)STEM_ACT_LRELU, slope, none);
        launch_gemv3b_g(Gk, st, n1, w1, ld1, b1, SegB{h1_k, n0, 0, (long)n0}, nil, nil, h2_k, (long)n1, (int)STEM_ACT_LRELU, slope, none);
        DecodeExtraB tail = none;
        tail.table = table; tail.T = T; tail.bound = scale_bound; tail.idx = mbox(0, k, p & 1);
        tail.cnt = dev + k * 2 + (p & 1); tail.flag_idx = pin + k; tail.flag_val = p + 1;
        launch_gemv3b_g(Gk, st, P, w2, ld2, b2, SegB{h2_k, n1, 0, (long)n1}, nil, nil, gp_k, (long)P, 0, 0.f, tail);
    };
    auto give_up = [&](const char *what, int k, int p) {
        for (int q = 0; q < ng; ++q) __atomic_store_n(pin + 2 + q, -1, __ATOMIC_RELEASE);       // polling kernels stop, later ones return
        (void)hipStreamSynchronize(st);
        stem_set_error("stem_ar_decode_batch_pipelined: %s (group %d, position %d)", what, k, p);
    };

    for (int p = 0; p < 2 && p < N; ++p)
        for (int k = 0; k < ng; ++k) chain(k, p);
    if (hipGetLastError() != hipSuccess) {
        give_up("launch failed", 0, 0);
        return -2;
    }
    for (int p = 0; p < N; ++p)
        for (int k = 0; k < ng; ++k) {
            const auto t0 = std::chrono::steady_clock::now();
            long spins = 0;
            while (__atomic_load_n(pin + k, __ATOMIC_ACQUIRE) < p + 1) {
                if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    give_up("timed out waiting for the device", k, p);
                    return -2;
                }
                __builtin_ia32_pause();
            }
            const int32_t *idx = mbox(0, k, p & 1);
            int32_t *sym = mbox(1, k, p & 1);
            for (int g = 0; g < gsz[k]; ++g)
                if (int rc = decode(decs[gbeg[k] + g], idx + (size_t)g * M, (size_t)M, cdfs, ncdf, cdf_stride, sizes, offsets, sym + (size_t)g * M)) {
                    give_up("host symbol decoder failed", k, p);
                    return rc < 0 ? -3 : -3;
                }
            __atomic_store_n(pin + 2 + k, p + 1, __ATOMIC_RELEASE);
            if (p + 2 < N) chain(k, p + 2);
        }
    for (int k = 0; k < ng; ++k) commit(k, N - 1);
    if (hipStreamSynchronize(st) != hipSuccess) {
        stem_set_error("stem_ar_decode_batch_pipelined: device error: %s", hipGetErrorString(hipGetLastError()));
        return -2;
    }
    int aborted = 0;
    if (hipMemcpy(&aborted, abort_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || aborted) {
        stem_set_error("stem_ar_decode_batch_pipelined: a device-side wait timed out");
        return -2;
    }
    return 0;
}
