// Entropy-model kernels: EntropyBottleneck / GaussianConditional likelihoods (forward + backward),
// quantisation, rate reduction, counter-based noise.  All HBM/latency-bound elementwise work; every
// multi-pass torch expression of the reference (entropy_models.py:388-452, 570-596) is one kernel.
#include "stem_common.h"

namespace {

constexpr int NP = STEM_EB_NPARAM;   // 58 floats per channel: M0 b0 f0 | M1 b1 f1 | M2 b2 f2 | M3 b3 f3 | M4 b4

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// per-channel transformed parameters: softplus(matrix), bias, tanh(factor)
struct EbPrep {
    float sp0[3], b0[3], tf0[3];
    float sp[3][9], b[3][3], tf[3][3];
    float sp4[3], b4;
};

__device__ void eb_prepare(const float *p, EbPrep &e)
{
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        e.sp0[o] = softplus_f(p[o]);
        e.b0[o] = p[3 + o];
        e.tf0[o] = tanhf(p[6 + o]);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float *q = p + 9 + 15 * l;
#pragma unroll
        for (int k = 0; k < 9; ++k) e.sp[l][k] = softplus_f(q[k]);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            e.b[l][o] = q[9 + o];
            e.tf[l][o] = tanhf(q[12 + o]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) e.sp4[k] = softplus_f(p[54 + k]);
    e.b4 = p[57];
}

// eb_prepare for a whole workgroup that works on ONE channel: threads 0 .. 57 transform one parameter each (softplus for the
// matrices, identity for the biases, tanh for the factors -- most of the arithmetic of the per-element kernels when every thread
// repeats it), every thread then reads the 58 results from LDS.  Two barriers; `prep` holds NP floats.
__device__ inline void eb_prepare_shared(const float *p, float *prep, EbPrep &e)
{
    __syncthreads();
    if (threadIdx.x < NP) {
        const int k = threadIdx.x;
        const float pv = p[k];
        const int r = k < 9 ? k / 3 : (k < 54 ? ((k - 9) % 15 < 9 ? 0 : ((k - 9) % 15 < 12 ? 1 : 2)) : (k < 57 ? 0 : 1));
        prep[k] = r == 0 ? softplus_f(pv) : (r == 1 ? pv : tanhf(pv));
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        e.sp0[o] = prep[o];
        e.b0[o] = prep[3 + o];
        e.tf0[o] = prep[6 + o];
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int k = 0; k < 9; ++k) e.sp[l][k] = prep[9 + 15 * l + k];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            e.b[l][o] = prep[9 + 15 * l + 9 + o];
            e.tf[l][o] = prep[9 + 15 * l + 12 + o];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) e.sp4[k] = prep[54 + k];
    e.b4 = prep[57];
}

// _logits_cumulative (entropy_models.py:388-407).  pre[l][o] = value before the tanh gate of layer l,
// in[l][o] = input of layer l (l=1..4); kept for the backward when KEEP.
template <bool KEEP>
__device__ __forceinline__ float eb_logits(const EbPrep &e, float v, float pre[4][3], float inp[4][3])
{
    float h[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float a = e.sp0[o] * v + e.b0[o];
        if (KEEP) pre[0][o] = a;
        h[o] = a + e.tf0[o] * tanhf(a);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        float g[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (KEEP) inp[l][o] = h[o];
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float a = e.sp[l][o * 3 + 0] * h[0] + e.sp[l][o * 3 + 1] * h[1] + e.sp[l][o * 3 + 2] * h[2] + e.b[l][o];
            if (KEEP) pre[l + 1][o] = a;
            g[o] = a + e.tf[l][o] * tanhf(a);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) h[o] = g[o];
    }
    if (KEEP) {
#pragma unroll
        for (int o = 0; o < 3; ++o) inp[3][o] = h[o];
    }
    return e.sp4[0] * h[0] + e.sp4[1] * h[1] + e.sp4[2] * h[2] + e.b4;
}

// reverse pass; accumulates d/d(raw params) into dp[58] and returns d/dv
__device__ float eb_logits_bwd(const float *p, const EbPrep &e, float v, const float pre[4][3], const float inp[4][3],
                               float gl, float *dp)
{
    float gh[3];
    // layer 4: out = sp4 . h + b4
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dp[54 + k] += gl * inp[3][k] * sigmoid_f(p[54 + k]);
        gh[k] = gl * e.sp4[k];
    }
    dp[57] += gl;
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        const float *q = p + 9 + 15 * l;
        float *dq = dp + 9 + 15 * l;
        float gin[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float a = pre[l + 1][o], ta = tanhf(a), tf = e.tf[l][o];
            dq[12 + o] += gh[o] * ta * (1.f - tf * tf);
            const float ga = gh[o] * (1.f + tf * (1.f - ta * ta));
            dq[9 + o] += ga;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dq[o * 3 + k] += ga * inp[l][k] * sigmoid_f(q[o * 3 + k]);
                gin[k] += ga * e.sp[l][o * 3 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) gh[k] = gin[k];
    }
    float gv = 0.f;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float a = pre[0][o], ta = tanhf(a), tf = e.tf0[o];
        dp[6 + o] += gh[o] * ta * (1.f - tf * tf);
        const float ga = gh[o] * (1.f + tf * (1.f - ta * ta));
        dp[3 + o] += ga;
        dp[o] += ga * v * sigmoid_f(p[o]);
        gv += ga * e.sp0[o];
    }
    return gv;
}

__global__ __launch_bounds__(256) void eb_forward_kernel(const float *z, int ldz, const float *noise, const float *pack,
                                                         const float *med, float *zhat, float *lik, size_t npix, int C,
                                                         int mode, float bound)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * C) return;
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    EbPrep e;
    eb_prepare(pack + (size_t)c * NP, e);
    float v = z[pix * ldz + c];
    if (mode == 0) {
        v += noise[i];
    } else {
        const float m = med[c];
        v = rintf(v - m) + m;
    }
    zhat[i] = v;
    const float lo = eb_logits<false>(e, v - 0.5f, nullptr, nullptr);
    const float up = eb_logits<false>(e, v + 0.5f, nullptr, nullptr);
    const float s = lo + up;
    const float sg = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
    const float l = fabsf(sigmoid_f(sg * up) - sigmoid_f(sg * lo));
    lik[i] = fmaxf(l, bound);
}

// eb_forward_kernel with one workgroup per channel (threads over pixels, eb_prepare_shared): the form for tensors with few pixels
// per channel (the hyper-latents); identical arithmetic per element to eb_forward_train_cm_kernel
__global__ __launch_bounds__(256) void eb_forward_cm_kernel(const float *z, int ldz, const float *noise, const float *pack,
                                                            const float *med, float *zhat, float *lik, size_t npix, int C,
                                                            int mode, float bound)
{
    __shared__ float prep[NP];
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        EbPrep e;
        eb_prepare_shared(pack + (size_t)c * NP, prep, e);
        for (size_t pix = threadIdx.x; pix < npix; pix += 256) {
            const size_t i = pix * C + c;
            float v = z[pix * ldz + c];
            if (mode == 0) {
                v += noise[i];
            } else {
                const float m = med[c];
                v = rintf(v - m) + m;
            }
            zhat[i] = v;
            const float lo = eb_logits<false>(e, v - 0.5f, nullptr, nullptr);
            const float up = eb_logits<false>(e, v + 0.5f, nullptr, nullptr);
            const float s = lo + up;
            const float sg = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
            const float l = fabsf(sigmoid_f(sg * up) - sigmoid_f(sg * lo));
            lik[i] = fmaxf(l, bound);
        }
    }
}

// one workgroup (EB_BWD_NT threads) per channel; reduces the 58 parameter gradients over pixels: wavefront shuffles, then a
// fixed-order sum of the wavefronts' partials through LDS (no atomics)
constexpr int EB_BWD_NT = 256;
__global__ __launch_bounds__(EB_BWD_NT) void eb_backward_kernel(const float *zhat, const float *pack, const float *dlik,
                                                                const float *dzin, float *dz, float *dpack, size_t npix, int C,
                                                                float bound, float *q)
{
    __shared__ float red[EB_BWD_NT / 64][NP];
    float dzmax = 0.f;
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *p = pack + (size_t)c * NP;
    __shared__ float prep[NP];
    EbPrep e;
    eb_prepare_shared(p, prep, e);
    float dp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) dp[k] = 0.f;
    for (size_t pix = threadIdx.x; pix < npix; pix += EB_BWD_NT) {
        const size_t i = pix * C + c;
        const float v = zhat[i];
        float pre_lo[4][3], in_lo[4][3], pre_up[4][3], in_up[4][3];
        const float lo = eb_logits<true>(e, v - 0.5f, pre_lo, in_lo);
        const float up = eb_logits<true>(e, v + 0.5f, pre_up, in_up);
        const float s = lo + up;
        const float sg = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
        const float su = sigmoid_f(sg * up), sl = sigmoid_f(sg * lo);
        const float diff = su - sl;
        float g = dlik[i];
        if (!(fabsf(diff) >= bound || g < 0.f)) g = 0.f;     // LowerBound rule (bound_ops.py:28-31)
        const float gd = g * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
        const float gup = gd * su * (1.f - su) * sg;
        const float glo = -gd * sl * (1.f - sl) * sg;
        float gv = eb_logits_bwd(p, e, v + 0.5f, pre_up, in_up, gup, dp);
        gv += eb_logits_bwd(p, e, v - 0.5f, pre_lo, in_lo, glo, dp);
        if (dz) {
            const float o = gv + (dzin ? dzin[i] : 0.f);
            dz[i] = o;
            dzmax = fmaxf(dzmax, fabsf(o));
        }
    }
    if (q) record_block_max(q, dzmax);         // max |dz| of this channel: the split in front of the hyper encoder's backward needs no maximum pass
    if (dpack) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float s = dp[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < NP) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < EB_BWD_NT / 64; ++w) s += red[w][threadIdx.x];
            dpack[(size_t)c * NP + threadIdx.x] = s;
        }
    }
}

// EntropyBottleneck.loss: sum |logits(quantiles) - target|, gradient wrt quantiles only
__global__ __launch_bounds__(256) void eb_aux_kernel(const float *quant, const float *pack, const float *target, float *loss,
                                                     float *dq, int C)
{
    __shared__ float red[256];
    float local = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < C * 3; i += gridDim.x * 256) {
        const int c = i / 3, k = i - c * 3;
        EbPrep e;
        eb_prepare(pack + (size_t)c * NP, e);
        float pre[4][3], inp[4][3], dp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) dp[q] = 0.f;
        const float v = quant[i];
        const float d = eb_logits<true>(e, v, pre, inp) - target[k];
        local += fabsf(d);
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (dq) dq[i] = eb_logits_bwd(pack + (size_t)c * NP, e, v, pre, inp, sgn, dp);
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, red[0]);
}

struct Ptr14 {
    const float *p[14];
};
struct MPtr14 {
    float *p[14];
};
__constant__ const int kOff[14] = {0, 3, 6, 9, 18, 21, 24, 33, 36, 39, 48, 51, 54, 57};
__constant__ const int kLen[14] = {3, 3, 3, 9, 3, 3, 9, 3, 3, 9, 3, 3, 3, 1};

__global__ void eb_pack_kernel(Ptr14 t, float *pack, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * NP) return;
    const int c = i / NP, k = i - c * NP;
    int s = 0;
#pragma unroll
    for (int q = 1; q < 14; ++q)
        if (k >= kOff[q]) s = q;
    pack[i] = t.p[s][c * kLen[s] + (k - kOff[s])];
}
__global__ void eb_unpack_kernel(const float *dpack, MPtr14 t, int C, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * NP) return;
    const int c = i / NP, k = i - c * NP;
    int s = 0;
#pragma unroll
    for (int q = 1; q < 14; ++q)
        if (k >= kOff[q]) s = q;
    float *dst = &t.p[s][c * kLen[s] + (k - kOff[s])];
    *dst = accumulate ? *dst + dpack[i] : dpack[i];
}

// ---- GaussianConditional -----------------------------------------------------------------------
__device__ __forceinline__ float std_cum(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

__global__ __launch_bounds__(256) void gc_forward_kernel(const float *y, const float *noise, const float *scales,
                                                         const float *means, int ldsm, float *out, float *lik, size_t npix,
                                                         int C, int mode, float sb, float lb)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * C) return;
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    const float mu = means[pix * ldsm + c], sc = scales[pix * ldsm + c];
    float o = y[i];
    if (mode == 0)
        o += noise[i];
    else
        o = rintf(o - mu) + mu;
    out[i] = o;
    const float v = fabsf(o - mu);
    const float s = fmaxf(sc, sb);
    const float l = std_cum((0.5f - v) / s) - std_cum((-0.5f - v) / s);
    lik[i] = fmaxf(l, lb);
}

__global__ __launch_bounds__(256) void gc_backward_kernel(const float *out, const float *scales, const float *means, int ldsm,
                                                          const float *dlik, float *dsc, float *dmu, int ldd, float *dy,
                                                          size_t npix, int C, float sb, float lb, float *q)
{
    // q (optional): scale record of (dscales | dmeans): max |value| per workgroup, for the fp16 split of the EPM's output gradient
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (i < npix * C) {
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    const float mu = means[pix * ldsm + c], sc = scales[pix * ldsm + c];
    const float d = out[i] - mu, v = fabsf(d);
    const float s = fmaxf(sc, sb);
    const float a = (0.5f - v) / s, b = (-0.5f - v) / s;
    const float lraw = std_cum(a) - std_cum(b);
    float g = dlik[i];
    if (!(lraw >= lb || g < 0.f)) g = 0.f;
    const float k = 0.39894228040143267794f;
    const float pa = k * expf(-0.5f * a * a), pb = k * expf(-0.5f * b * b);
    const float dv = g * (-(pa - pb) / s);
    float ds = g * (-(pa * a - pb * b) / s);
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (!(sc >= sb || ds < 0.f)) ds = 0.f;
    if (dsc) dsc[pix * ldd + c] = ds;
    if (dmu) dmu[pix * ldd + c] = -dv * sgn;
    if (dy) dy[i] = dv * sgn;
    m = fmaxf(fabsf(ds), fabsf(dv));
    }
    if (q) record_block_max(q, m);
}

__global__ __launch_bounds__(256) void log2_sum_kernel(const float *lik, size_t n, double *acc)
{
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)log2f(lik[i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}

__global__ void dlog_kernel(const float *lik, float *dlik, size_t n, float coef)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dlik[i] = coef / lik[i];
}
template <int OP>
__global__ void ew_kernel(const float *a, const float *b, float *o, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (OP == 0) o[i] = a[i] - b[i];
    if (OP == 1) o[i] = a[i] + b[i];
    if (OP == 2) o[i] = rintf(a[i]);     // torch.round: half to even
}
__global__ void lrelu_bwd_kernel(const float *yact, const float *dy, float *dx, size_t n, float slope)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = yact[i] > 0.f ? dy[i] : dy[i] * slope;
}

// Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// epoch != nullptr: the counter additionally advances by epoch[0] * epoch_stride, a device-resident draw count, so that a
// captured hipGraph produces fresh noise on every replay (the host-side offset is frozen into the graph at capture).
__global__ void noise_kernel(float *out, size_t n, uint64_t seed, uint64_t offset, const long long *epoch, uint64_t epoch_stride)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    const uint64_t ctr = offset + q + (epoch ? (uint64_t)epoch[0] * epoch_stride : 0);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t r4[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q * 4 + e < n) out[q * 4 + e] = (float)(r4[e] >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

__global__ void build_indexes_kernel(const float *scales, int lds, const float *table, int T, int32_t *idx, size_t npix, int C,
                                     float sb)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    const float s = fmaxf(scales[pix * lds + c], sb);
    int k = T - 1;
    for (int t = 0; t < T - 1; ++t) k -= (s <= table[t]) ? 1 : 0;
    idx[i] = k;
}

inline unsigned nblk(size_t n) { return (unsigned)cdivz(n, 256); }

}   // namespace

STEM_EXPORT int stem_eb_pack(const float *const *tensors14, float *pack, int C, void *stream)
{
    STEM_CHECK_ARG(tensors14 && pack && C > 0, "stem_eb_pack: bad arguments");
    Ptr14 t;
    for (int i = 0; i < 14; ++i) t.p[i] = tensors14[i];
    hipLaunchKernelGGL(eb_pack_kernel, dim3(nblk((size_t)C * NP)), dim3(256), 0, (hipStream_t)stream, t, pack, C);
    STEM_LAUNCH_CHECK("eb_pack");
    return 0;
}
STEM_EXPORT int stem_eb_unpack_grads(const float *dpack, float *const *tensors14, int C, int accumulate, void *stream)
{
    STEM_CHECK_ARG(tensors14 && dpack && C > 0, "stem_eb_unpack_grads: bad arguments");
    MPtr14 t;
    for (int i = 0; i < 14; ++i) t.p[i] = tensors14[i];
    hipLaunchKernelGGL(eb_unpack_kernel, dim3(nblk((size_t)C * NP)), dim3(256), 0, (hipStream_t)stream, dpack, t, C, accumulate);
    STEM_LAUNCH_CHECK("eb_unpack");
    return 0;
}

STEM_EXPORT int stem_eb_forward(const float *z, int ldz, const float *noise, const float *pack, const float *medians,
                                float *z_hat, float *lik, int B, int H, int W, int C, int mode, float bound, void *stream)
{
    STEM_CHECK_ARG(z && pack && z_hat && lik, "stem_eb_forward: null pointer");
    STEM_CHECK_ARG((mode == 0 && noise) || (mode == 1 && medians), "stem_eb_forward: mode %d needs %s", mode, mode ? "medians" : "noise");
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(npix <= 4096 ? eb_forward_cm_kernel : eb_forward_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, z, ldz, noise, pack, medians,
                       z_hat, lik, npix, C, mode, bound);
    STEM_LAUNCH_CHECK("eb_forward");
    return 0;
}

STEM_EXPORT int stem_eb_backward_rec(const float *z_hat, const float *pack, const float *dlik, const float *dzhat_in,
                                     float *dz, float *dpack, int B, int H, int W, int C, float bound, float *dz_rec, void *stream)
{
    STEM_CHECK_ARG(z_hat && pack && dlik && (dz || !dz_rec), "stem_eb_backward: null pointer");
    hipLaunchKernelGGL(eb_backward_kernel, dim3(C), dim3(EB_BWD_NT), 0, (hipStream_t)stream, z_hat, pack, dlik, dzhat_in, dz, dpack,
                       (size_t)B * H * W, C, bound, dz_rec);
    STEM_LAUNCH_CHECK("eb_backward");
    return 0;
}

STEM_EXPORT int stem_eb_backward(const float *z_hat, const float *pack, const float *dlik, const float *dzhat_in,
                                 float *dz, float *dpack, int B, int H, int W, int C, float bound, void *stream)
{
    return stem_eb_backward_rec(z_hat, pack, dlik, dzhat_in, dz, dpack, B, H, W, C, bound, nullptr, stream);
}

STEM_EXPORT int stem_eb_aux_loss(const float *quantiles, const float *pack, const float *target3, float *loss,
                                 float *dquantiles, int C, void *stream)
{
    STEM_CHECK_ARG(quantiles && pack && target3 && loss, "stem_eb_aux_loss: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) {
        stem_set_error("stem_eb_aux_loss: memset failed");
        return -2;
    }
    hipLaunchKernelGGL(eb_aux_kernel, dim3(cdiv(C * 3, 256)), dim3(256), 0, st, quantiles, pack, target3, loss, dquantiles, C);
    STEM_LAUNCH_CHECK("eb_aux");
    return 0;
}

STEM_EXPORT int stem_gc_forward(const float *y, const float *noise, const float *scales, const float *means, int ldsm,
                                float *out, float *lik, size_t npix, int C, int mode, float scale_bound, float lik_bound,
                                void *stream)
{
    STEM_CHECK_ARG(y && scales && means && out && lik, "stem_gc_forward: null pointer");
    STEM_CHECK_ARG(mode == 1 || noise, "stem_gc_forward: noise mode without a noise tensor");
    hipLaunchKernelGGL(gc_forward_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, y, noise, scales, means, ldsm,
                       out, lik, npix, C, mode, scale_bound, lik_bound);
    STEM_LAUNCH_CHECK("gc_forward");
    return 0;
}

STEM_EXPORT int stem_gc_backward(const float *out, const float *scales, const float *means, int ldsm, const float *dlik,
                                 float *dscales, float *dmeans, int lddsm, float *dy, size_t npix, int C,
                                 float scale_bound, float lik_bound, float *q, void *stream)
{
    STEM_CHECK_ARG(out && scales && means && dlik, "stem_gc_backward: null pointer");
    hipLaunchKernelGGL(gc_backward_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, out, scales, means, ldsm,
                       dlik, dscales, dmeans, lddsm, dy, npix, C, scale_bound, lik_bound, q);
    STEM_LAUNCH_CHECK("gc_backward");
    return 0;
}

STEM_EXPORT int stem_log2_sum(const float *lik, size_t n, double *acc, void *stream)
{
    STEM_CHECK_ARG(lik && acc, "stem_log2_sum: null pointer");
    unsigned nb = nblk(n);
    if (nb > 1024) nb = 1024;
    if (nb == 0) return 0;
    hipLaunchKernelGGL(log2_sum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, lik, n, acc);
    STEM_LAUNCH_CHECK("log2_sum");
    return 0;
}

STEM_EXPORT int stem_dlog(const float *lik, float *dlik, size_t n, float coef, void *stream)
{
    STEM_CHECK_ARG(lik && dlik, "stem_dlog: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dlog_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, lik, dlik, n, coef);
    STEM_LAUNCH_CHECK("dlog");
    return 0;
}

#define STEM_EW(name, OP)                                                                                    \
    STEM_EXPORT int name(const float *a, const float *b, float *out, size_t n, void *stream)                 \
    {                                                                                                        \
        STEM_CHECK_ARG(a && b && out, #name ": null pointer");                                               \
        if (n == 0) return 0;                                                                                \
        hipLaunchKernelGGL((ew_kernel<OP>), dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n); \
        STEM_LAUNCH_CHECK(#name);                                                                            \
        return 0;                                                                                            \
    }
STEM_EW(stem_sub, 0)
STEM_EW(stem_add, 1)

STEM_EXPORT int stem_round(const float *a, float *out, size_t n, void *stream)
{
    STEM_CHECK_ARG(a && out, "stem_round: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL((ew_kernel<2>), dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, a, out, n);
    STEM_LAUNCH_CHECK("stem_round");
    return 0;
}

STEM_EXPORT int stem_lrelu_bwd(const float *yact, const float *dy, float *dx, size_t n, float slope, void *stream)
{
    STEM_CHECK_ARG(yact && dy && dx, "stem_lrelu_bwd: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, yact, dy, dx, n, slope);
    STEM_LAUNCH_CHECK("lrelu_bwd");
    return 0;
}

STEM_EXPORT int stem_uniform_noise(float *out, size_t n, uint64_t seed, uint64_t offset, void *stream)
{
    STEM_CHECK_ARG(out, "stem_uniform_noise: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(noise_kernel, dim3(nblk(cdivz(n, 4))), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset,
                       (const long long *)nullptr, (uint64_t)0);
    STEM_LAUNCH_CHECK("noise");
    return 0;
}

STEM_EXPORT int stem_uniform_noise_epoch(float *out, size_t n, uint64_t seed, uint64_t offset, const long long *epoch_dev,
                                         uint64_t epoch_stride, void *stream)
{
    STEM_CHECK_ARG(out && epoch_dev, "stem_uniform_noise_epoch: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(noise_kernel, dim3(nblk(cdivz(n, 4))), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, epoch_dev,
                       epoch_stride);
    STEM_LAUNCH_CHECK("noise_epoch");
    return 0;
}

STEM_EXPORT int stem_build_indexes(const float *scales, int lds, const float *table, int T, int32_t *idx, size_t npix, int C,
                                   float scale_bound, void *stream)
{
    STEM_CHECK_ARG(scales && table && idx && T >= 1, "stem_build_indexes: bad arguments");
    hipLaunchKernelGGL(build_indexes_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, scales, lds, table, T, idx,
                       npix, C, scale_bound);
    STEM_LAUNCH_CHECK("build_indexes");
    return 0;
}

// =================================================================================================================
// Fused training glue.  Between the convolutions, one P-frame optimisation step (stem/trainSTEM.py:203-218) issues ~50
// elementwise / reduction kernels of a few microseconds each (concat copies, residual, three noise draws, quantisation,
// likelihoods, log-sums, their autograd mirror images, norm / Adam bookkeeping).  On MI355X each dependent dispatch costs
// ~10 us of queue latency on top of its run time (tools/timeline.py: 4.8 ms of a 32 ms bench step), so the same
// arithmetic is regrouped into four kernels: prologue, bottleneck (+rate, +d rate), Gaussian (+rate, +d rate), finalise.
// EMLoss is a sum of logs (utils.py:18-27), so d loss / d likelihood = coef / likelihood is known in the forward pass.
namespace {

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, float r[4])
{
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t v[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (float)(v[e] >> 8) * (1.0f / 16777216.0f) - 0.5f;      // == noise_kernel
}

struct NoiseSrc {            // either an explicit tensor (parity tests) or the Philox stream (seed, offset [+ epoch * stride])
    const float *ptr;
    uint64_t seed, offset, stride;
    const long long *epoch;
};
__device__ __forceinline__ uint64_t noise_base(const NoiseSrc &s) { return s.offset + (s.epoch ? (uint64_t)s.epoch[0] * s.stride : 0); }

// he_in = [y_cur | y_cond]; target = y_cur - y_cond (residual) or y_cur; t_hat = target + U(-1/2,1/2) (training) or
// round(target); y_hat = t_hat + y_cond (residual) or t_hat.     spatiotemporalpriors.py:846-856,863
// q_in / q_t (optional): scale records (stem_common.h) of he_in (= of y_cur and y_cond) and of t_hat: one slot of max |value| per
// workgroup, so that the fp16 splits of these tensors need no maximum pass of their own
__device__ inline float max4(const f32x4 v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }
__device__ inline void prior_prologue_body(const float *ycur, int ldc, const float *ycond, int ldd, float *he_in, int ldh, float *target,
                                           float *t_hat, float *y_hat, const NoiseSrc &nq, size_t i4, int C, int residual, int training,
                                           float &m_in, float &m_t);
__global__ __launch_bounds__(256) void prior_prologue_kernel(const float *ycur, int ldc, const float *ycond, int ldd, float *he_in,
                                                             int ldh, float *target, float *t_hat, float *y_hat, NoiseSrc nq,
                                                             size_t npix, int C, int residual, int training, float *q_in, float *q_t)
{
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    float m_in = 0.f, m_t = 0.f;
    if (i4 < npix * (C >> 2))
        prior_prologue_body(ycur, ldc, ycond, ldd, he_in, ldh, target, t_hat, y_hat, nq, i4, C, residual, training, m_in, m_t);
    if (q_in) record_block_max(q_in, m_in);
    if (q_t) record_block_max(q_t, m_t);
}
__device__ inline void prior_prologue_body(const float *ycur, int ldc, const float *ycond, int ldd, float *he_in, int ldh, float *target,
                                           float *t_hat, float *y_hat, const NoiseSrc &nq, size_t i4, int C, int residual, int training,
                                           float &m_in, float &m_t)
{
    const int c4n = C >> 2;
    const size_t pix = i4 / c4n;
    const int c = (int)(i4 - pix * c4n) * 4;
    const f32x4 yc = *reinterpret_cast<const f32x4 *>(ycur + pix * ldc + c);
    const f32x4 yd = *reinterpret_cast<const f32x4 *>(ycond + pix * ldd + c);
    *reinterpret_cast<f32x4 *>(he_in + pix * ldh + c) = yc;
    *reinterpret_cast<f32x4 *>(he_in + pix * ldh + C + c) = yd;
    m_in = fmaxf(max4(yc), max4(yd));
    f32x4 tg = residual ? yc - yd : yc;
    *reinterpret_cast<f32x4 *>(target + pix * C + c) = tg;
    if (!t_hat) return;
    f32x4 th;
    if (training) {
        float r[4];
        if (nq.ptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = nq.ptr[pix * C + c + e];
        } else {
            philox4(nq.seed, noise_base(nq) + i4, r);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) th[e] = tg[e] + r[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) th[e] = rintf(tg[e]);
    }
    *reinterpret_cast<f32x4 *>(t_hat + pix * C + c) = th;
    m_t = max4(th);
    if (y_hat) *reinterpret_cast<f32x4 *>(y_hat + pix * C + c) = residual ? th + yd : th;
}

// block-level sum of log2(lik) in double -> part[blockIdx.x] (fixed order: deterministic, no atomics)
__device__ __forceinline__ void block_log2_partial(double s, double *part)
{
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// EntropyBottleneck.forward in training mode (+noise) with its rate term and d rate / d likelihood
__global__ __launch_bounds__(256) void eb_forward_train_kernel(const float *z, int ldz, NoiseSrc nz, const float *pack, float *zhat,
                                                               float *lik, float *dlik, double *part, size_t npix, int C,
                                                               float bound, float coef, float *q)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double lg = 0.0;
    float zm = 0.f;
    if (i < npix * C) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        EbPrep e;
        eb_prepare(pack + (size_t)c * NP, e);
        float v = z[pix * ldz + c];
        if (nz.ptr) {
            v += nz.ptr[i];
        } else {
            float r[4];
            philox4(nz.seed, noise_base(nz) + (i >> 2), r);
            v += r[i & 3];
        }
        zhat[i] = v;
        zm = fmaxf(zm, fabsf(v));
        const float lo = eb_logits<false>(e, v - 0.5f, nullptr, nullptr);
        const float up = eb_logits<false>(e, v + 0.5f, nullptr, nullptr);
        const float s = lo + up;
        const float sg = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
        const float l = fmaxf(fabsf(sigmoid_f(sg * up) - sigmoid_f(sg * lo)), bound);
        lik[i] = l;
        dlik[i] = coef / l;
        lg = (double)log2f(l);
    }
    block_log2_partial(lg, part);
    if (q) record_block_max(q, zm);
}

// The same with one workgroup per CHANNEL (threads over pixels): the 48 softplus / tanh values of a channel's parameters
// (eb_prepare: most of the kernel's arithmetic when every thread repeats it) are computed once per workgroup and shared through
// LDS.  For the hyper-latents of the STEM model (a few hundred pixels per channel) this is the form used: the accesses along a
// channel are strided, which only matters for large tensors.  Same arithmetic per element, same Philox counters.
__global__ __launch_bounds__(256) void eb_forward_train_cm_kernel(const float *z, int ldz, NoiseSrc nz, const float *pack, float *zhat,
                                                                  float *lik, float *dlik, double *part, size_t npix, int C,
                                                                  float bound, float coef, float *q)
{
    __shared__ float prep[NP];
    double lg = 0.0;
    float zm = 0.f;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        EbPrep e;
        eb_prepare_shared(pack + (size_t)c * NP, prep, e);
        for (size_t pix = threadIdx.x; pix < npix; pix += 256) {
            const size_t i = pix * C + c;
            float v = z[pix * ldz + c];
            if (nz.ptr) {
                v += nz.ptr[i];
            } else {
                float r[4];
                philox4(nz.seed, noise_base(nz) + (i >> 2), r);
                v += r[i & 3];
            }
            zhat[i] = v;
            zm = fmaxf(zm, fabsf(v));
            const float lo = eb_logits<false>(e, v - 0.5f, nullptr, nullptr);
            const float up = eb_logits<false>(e, v + 0.5f, nullptr, nullptr);
            const float s = lo + up;
            const float sg = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
            const float l = fmaxf(fabsf(sigmoid_f(sg * up) - sigmoid_f(sg * lo)), bound);
            lik[i] = l;
            dlik[i] = coef / l;
            lg += (double)log2f(l);
        }
    }
    block_log2_partial(lg, part);
    if (q) record_block_max(q, zm);
}

// GaussianConditional.forward in training mode (+noise; means are ignored by the noise quantiser, entropy_models.py:128-135)
// dsc / dmu (optional, both or none): the backward of the same element in the same pass -- gc_backward_kernel's expressions on the
// values this thread already holds (d loss / d likelihood is known here: coef / lik) -- and the scale record q of (dscales | dmeans)
__global__ __launch_bounds__(256) void gc_forward_train_kernel(const float *y, NoiseSrc nl, const float *scales, const float *means,
                                                               int ldsm, float *out, float *lik, float *dlik, double *part,
                                                               size_t npix, int C, float sb, float lb, float coef, float *dsc, float *dmu,
                                                               int ldd, float *q)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double lg = 0.0;
    float gm = 0.f;
    if (i < npix * C) {
        const size_t pix = i / C;
        const int c = (int)(i - pix * C);
        const float mu = means[pix * ldsm + c], sc = scales[pix * ldsm + c];
        float o = y[i];
        if (nl.ptr) {
            o += nl.ptr[i];
        } else {
            float r[4];
            philox4(nl.seed, noise_base(nl) + (i >> 2), r);
            o += r[i & 3];
        }
        out[i] = o;
        const float v = fabsf(o - mu);
        const float s = fmaxf(sc, sb);
        const float l = fmaxf(std_cum((0.5f - v) / s) - std_cum((-0.5f - v) / s), lb);
        lik[i] = l;
        dlik[i] = coef / l;
        lg = (double)log2f(l);
        if (dsc) {
            const float d = o - mu;
            const float a = (0.5f - v) / s, b = (-0.5f - v) / s;
            const float lraw = std_cum(a) - std_cum(b);
            float g = coef / l;
            if (!(lraw >= lb || g < 0.f)) g = 0.f;
            const float k = 0.39894228040143267794f;
            const float pa = k * expf(-0.5f * a * a), pb = k * expf(-0.5f * b * b);
            const float dv = g * (-(pa - pb) / s);
            float ds = g * (-(pa * a - pb * b) / s);
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            if (!(sc >= sb || ds < 0.f)) ds = 0.f;
            dsc[pix * ldd + c] = ds;
            dmu[pix * ldd + c] = -dv * sgn;
            gm = fmaxf(fabsf(ds), fabsf(dv));
        }
    }
    block_log2_partial(lg, part);
    if (q) record_block_max(q, gm);
}

// out[0] = y_bpp, out[1] = z_bpp, out[2] = loss   (EMLoss, utils.py:18-27): sums of the per-block partials in index order
__global__ __launch_bounds__(256) void em_loss_finalize_kernel(const double *py, int ny, const double *pz, int nz, double scale, double *out)
{
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < ny; i += 256) a += py[i];
    for (int i = threadIdx.x; i < nz; i += 256) b += pz[i];
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) {
            red[0][threadIdx.x] += red[0][threadIdx.x + k];
            red[1][threadIdx.x] += red[1][threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0][0] * scale;
        out[1] = red[1][0] * scale;
        out[2] = out[0] + out[1];
    }
}

// EntropyBottleneck.loss with its quantile gradient (entropy_models.py:383-386), one workgroup, no pre-zeroed accumulator:
// loss[0] = sum |logits - target|, dq = d loss / d quantiles (written, or added when accumulate).
//   phase 1: the C x 58 parameter transforms (softplus / identity / tanh: most of the arithmetic) once each, spread over the
//            workgroup, into LDS -- round 5's kernel repeated a channel's 58 transforms for each of its three quantiles;
//   phase 2: one (channel, quantile) pair per thread, AUX_T pairs per pass (3 C = 768 for the training model: one pass): the
//            forward keeps the gates' tanh values, the reverse pass walks only the input-gradient chain (no parameter gradients)
//            and re-uses them.
// Same numbers as before: the transforms, the forward and the d/dv chain are the operations of eb_prepare / eb_logits /
// eb_logits_bwd in their order, and the |d| terms are summed in the order of the 256-thread kernel (item t, t + 256, t + 512 per
// slot, then the same tree).  86 -> ~15 us per launch.
constexpr int AUX_T = 768;
__global__ __launch_bounds__(AUX_T) void eb_aux_block_kernel(const float *quant, const float *pack, const float *target, float *loss,
                                                             float *dq, int C, int accumulate)
{
    extern __shared__ float prep[];                    // [C][NP] transformed parameters
    __shared__ float red[256];
    __shared__ float vals[AUX_T];
    for (int i = threadIdx.x; i < C * NP; i += AUX_T) {
        const int k = i % NP;
        const float pv = pack[i];
        const int r = k < 9 ? k / 3 : (k < 54 ? ((k - 9) % 15 < 9 ? 0 : ((k - 9) % 15 < 12 ? 1 : 2)) : (k < 57 ? 0 : 1));
        prep[i] = r == 0 ? softplus_f(pv) : (r == 1 ? pv : tanhf(pv));
    }
    __syncthreads();
    const int n = C * 3;
    float slot = 0.f;                                  // threads 0 .. 255: the running sum of items t, t + 256, t + 512, ...
    for (int base = 0; base < n; base += AUX_T) {
        const int i = base + threadIdx.x;
        float ad = 0.f;
        if (i < n) {
            const int c = i / 3, k = i - c * 3;
            const float *e = prep + (size_t)c * NP;    // sp0[3] b0[3] tf0[3] | 3 x (sp[9] b[3] tf[3]) | sp4[3] b4
            const float v = quant[i];
            float h[3], ta[4][3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float a = e[o] * v + e[3 + o];
                ta[0][o] = tanhf(a);
                h[o] = a + e[6 + o] * ta[0][o];
            }
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float *q = e + 9 + 15 * l;
                float gg[3];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float a = q[o * 3 + 0] * h[0] + q[o * 3 + 1] * h[1] + q[o * 3 + 2] * h[2] + q[9 + o];
                    ta[l + 1][o] = tanhf(a);
                    gg[o] = a + q[12 + o] * ta[l + 1][o];
                }
#pragma unroll
                for (int o = 0; o < 3; ++o) h[o] = gg[o];
            }
            const float d = (e[54] * h[0] + e[55] * h[1] + e[56] * h[2] + e[57]) - target[k];
            ad = fabsf(d);
            const float gl = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float gh[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) gh[kk] = gl * e[54 + kk];
#pragma unroll
            for (int l = 2; l >= 0; --l) {
                const float *q = e + 9 + 15 * l;
                float gin[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float t = ta[l + 1][o], tf = q[12 + o];
                    const float ga = gh[o] * (1.f + tf * (1.f - t * t));
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) gin[kk] += ga * q[o * 3 + kk];
                }
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) gh[kk] = gin[kk];
            }
            float g = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float t = ta[0][o], tf = e[6 + o];
                const float ga = gh[o] * (1.f + tf * (1.f - t * t));
                g += ga * e[o];
            }
            if (dq) dq[i] = accumulate ? dq[i] + g : g;
        }
        vals[threadIdx.x] = ad;
        __syncthreads();
        if (threadIdx.x < 256) {
#pragma unroll
            for (int j = 0; j < AUX_T / 256; ++j) slot += vals[threadIdx.x + 256 * j];
        }
        __syncthreads();
    }
    if (threadIdx.x < 256) red[threadIdx.x] = slot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0];
}

NoiseSrc make_noise(const float *ptr, uint64_t seed, uint64_t offset, const long long *epoch, uint64_t stride)
{
    NoiseSrc s;
    s.ptr = ptr; s.seed = seed; s.offset = offset; s.epoch = epoch; s.stride = stride;
    return s;
}

}   // namespace

STEM_EXPORT int stem_prior_prologue(const float *y_cur, int ldc, const float *y_cond, int ldd, float *he_in, int ldh, float *target,
                                    float *t_hat, float *y_hat, const float *noise, uint64_t seed, uint64_t offset,
                                    const long long *epoch_dev, uint64_t epoch_stride, size_t npix, int C, int residual,
                                    int training, float *q_in, float *q_t, void *stream)
{
    STEM_CHECK_ARG(y_cur && y_cond && he_in && target, "stem_prior_prologue: null pointer");
    STEM_CHECK_ARG(C % 4 == 0 && ldc % 4 == 0 && ldd % 4 == 0 && ldh % 4 == 0 && ldh >= 2 * C,
                   "stem_prior_prologue: channel counts / pitches must be multiples of 4 (C=%d ldc=%d ldd=%d ldh=%d)", C, ldc, ldd, ldh);
    STEM_CHECK_ARG(((((uintptr_t)y_cur) | ((uintptr_t)y_cond) | ((uintptr_t)he_in) | ((uintptr_t)target) | ((uintptr_t)t_hat) |
                     ((uintptr_t)y_hat)) & 15) == 0, "stem_prior_prologue: pointers must be 16-byte aligned");
    if (npix == 0) return 0;
    hipLaunchKernelGGL(prior_prologue_kernel, dim3(nblk(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, y_cur, ldc, y_cond, ldd,
                       he_in, ldh, target, t_hat, y_hat, make_noise(noise, seed, offset, epoch_dev, epoch_stride), npix, C, residual,
                       training, q_in, t_hat ? q_t : nullptr);
    STEM_LAUNCH_CHECK("prior_prologue");
    return 0;
}

STEM_EXPORT int stem_rate_partials(size_t n) { return (int)nblk(n); }

STEM_EXPORT int stem_eb_forward_train_rec(const float *z, int ldz, const float *pack, const float *noise, uint64_t seed, uint64_t offset,
                                          const long long *epoch_dev, uint64_t epoch_stride, float *z_hat, float *lik, float *dlik,
                                          double *partials, size_t npix, int C, float bound, float coef, float *zhat_rec, void *stream)
{
    STEM_CHECK_ARG(z && pack && z_hat && lik && dlik && partials, "stem_eb_forward_train: null pointer");
    if (npix == 0) return 0;
    if (npix <= 4096)        // few pixels per channel (hyper-latents): one workgroup per channel shares the prepared parameters
        hipLaunchKernelGGL(eb_forward_train_cm_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, z, ldz,
                           make_noise(noise, seed, offset, epoch_dev, epoch_stride), pack, z_hat, lik, dlik, partials, npix, C, bound, coef, zhat_rec);
    else
        hipLaunchKernelGGL(eb_forward_train_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, z, ldz,
                           make_noise(noise, seed, offset, epoch_dev, epoch_stride), pack, z_hat, lik, dlik, partials, npix, C, bound, coef, zhat_rec);
    STEM_LAUNCH_CHECK("eb_forward_train");
    return 0;
}

STEM_EXPORT int stem_eb_forward_train(const float *z, int ldz, const float *pack, const float *noise, uint64_t seed, uint64_t offset,
                                      const long long *epoch_dev, uint64_t epoch_stride, float *z_hat, float *lik, float *dlik,
                                      double *partials, size_t npix, int C, float bound, float coef, void *stream)
{
    return stem_eb_forward_train_rec(z, ldz, pack, noise, seed, offset, epoch_dev, epoch_stride, z_hat, lik, dlik, partials, npix, C, bound, coef,
                                     nullptr, stream);
}

STEM_EXPORT int stem_gc_forward_train(const float *y, const float *scales, const float *means, int ldsm, const float *noise,
                                      uint64_t seed, uint64_t offset, const long long *epoch_dev, uint64_t epoch_stride, float *out,
                                      float *lik, float *dlik, double *partials, size_t npix, int C, float scale_bound,
                                      float lik_bound, float coef, void *stream)
{
    STEM_CHECK_ARG(y && scales && means && out && lik && dlik && partials, "stem_gc_forward_train: null pointer");
    if (npix == 0) return 0;
    hipLaunchKernelGGL(gc_forward_train_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, y,
                       make_noise(noise, seed, offset, epoch_dev, epoch_stride), scales, means, ldsm, out, lik, dlik, partials, npix, C,
                       scale_bound, lik_bound, coef, nullptr, nullptr, 0, nullptr);
    STEM_LAUNCH_CHECK("gc_forward_train");
    return 0;
}

STEM_EXPORT int stem_gc_forward_backward_train(const float *y, const float *scales, const float *means, int ldsm, const float *noise,
                                               uint64_t seed, uint64_t offset, const long long *epoch_dev, uint64_t epoch_stride, float *out,
                                               float *lik, float *dlik, double *partials, size_t npix, int C, float scale_bound,
                                               float lik_bound, float coef, float *dscales, float *dmeans, int lddsm, float *q, void *stream)
{
    STEM_CHECK_ARG(y && scales && means && out && lik && dlik && partials && dscales && dmeans, "stem_gc_forward_backward_train: null pointer");
    if (npix == 0) return 0;
    hipLaunchKernelGGL(gc_forward_train_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, y,
                       make_noise(noise, seed, offset, epoch_dev, epoch_stride), scales, means, ldsm, out, lik, dlik, partials, npix, C,
                       scale_bound, lik_bound, coef, dscales, dmeans, lddsm, q);
    STEM_LAUNCH_CHECK("gc_forward_backward_train");
    return 0;
}

STEM_EXPORT int stem_em_loss_finalize(const double *partials_y, int ny, const double *partials_z, int nz, double scale, double *out3,
                                      void *stream)
{
    STEM_CHECK_ARG(partials_y && partials_z && out3 && ny >= 0 && nz >= 0, "stem_em_loss_finalize: bad arguments");
    hipLaunchKernelGGL(em_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials_y, ny, partials_z, nz, scale, out3);
    STEM_LAUNCH_CHECK("em_loss_finalize");
    return 0;
}

STEM_EXPORT int stem_eb_aux_loss_grad(const float *quantiles, const float *pack, const float *target3, float *loss, float *dquantiles,
                                      int C, int accumulate, void *stream)
{
    STEM_CHECK_ARG(quantiles && pack && target3 && loss, "stem_eb_aux_loss_grad: null pointer");
    const size_t lds = (size_t)C * NP * sizeof(float);
    STEM_CHECK_ARG(C >= 1 && lds + (256 + AUX_T) * sizeof(float) <= 160 * 1024, "stem_eb_aux_loss_grad: C = %d channels do not fit one workgroup's LDS", C);
    static int lds_set = 0;
    if (lds > (size_t)lds_set) {
        (void)hipFuncSetAttribute((const void *)eb_aux_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL(eb_aux_block_kernel, dim3(1), dim3(AUX_T), lds, (hipStream_t)stream, quantiles, pack, target3, loss, dquantiles, C,
                       accumulate);
    STEM_LAUNCH_CHECK("eb_aux_block");
    return 0;
}
