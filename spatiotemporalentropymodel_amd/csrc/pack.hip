// Weight (un)packing and NCHW <-> NHWC layout kernels.  All HBM-bound byte movers: every global
// access is a contiguous run (taps of one filter on the reference side, a channel row on the
// packed side); the permutation happens in LDS.
#include <stdarg.h>

#include "stem_common.h"

static thread_local char g_err[512] = "";
void stem_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
STEM_EXPORT const char *stem_last_error(void) { return g_err; }
STEM_EXPORT int stem_abi_version(void) { return 5; }      // 5 (round 5): + stem_tconv2d_f16x3_*, stem_conv2d_wgrad_f16x3_strided, stem_f16x2_pack_conv_weights_pair_multi, stem_tape_set_farg, stem_tape_entry_recordable, stem_zero_bytes, stem_stream_flag_*; stem_f16x2_pack_desc.flip = 2
STEM_EXPORT int stem_built_with_experiments(void)
{
#ifdef STEM_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

static int g_tuning[STEM_TUNE_COUNT] = {0};
static const char *const kTuningNames[STEM_TUNE_COUNT] = {"fx3_tile", "fx3_split", "wg3_split", "arp_workers", "fx3_depth", "fx3_gen_tile", "fx3_mfma", "fx3_gen_mfma", "fx3_gen_img", "fx3_img_w", "wg3_row", "wg3_minch", "tconv_cps", "arp_giveup_at", "unpack_mb"};
int stem_tuning(int id) { return g_tuning[id]; }
STEM_EXPORT int stem_tuning_set(const char *name, int value)
{
    STEM_CHECK_ARG(name && value >= 0, "stem_tuning_set: null name or negative value");
    for (int i = 0; i < STEM_TUNE_COUNT; ++i)
        if (!strcmp(name, kTuningNames[i])) {
            STEM_CHECK_ARG(i != STEM_TUNE_FX3_TILE || value == 0 || value == 64 || value == 128, "stem_tuning_set: fx3_tile is 0 (automatic), 64 or 128");
            STEM_CHECK_ARG(i != STEM_TUNE_FX3_GEN_TILE || value == 0 || value == 64 || value == 128, "stem_tuning_set: fx3_gen_tile is 0 (automatic), 64 or 128");
            STEM_CHECK_ARG(i != STEM_TUNE_FX3_DEPTH || value == 0 || value == 2 || value == 3, "stem_tuning_set: fx3_depth is 0 (automatic), 2 or 3 (LDS stages of the split-operand main loops)");
            STEM_CHECK_ARG(i != STEM_TUNE_FX3_GEN_MFMA || value == 0 || value == 16 || value == 32, "stem_tuning_set: fx3_gen_mfma is 0 (automatic), 16 or 32 (rows of the general kernel's MFMA shape)");
            STEM_CHECK_ARG(i != STEM_TUNE_UNPACK_MB || value == 0 || (value % 32 == 0 && value >= 32 && value <= 192), "stem_tuning_set: unpack_mb is 0 (automatic) or a multiple of 32 up to 192");
            STEM_CHECK_ARG(i != STEM_TUNE_FX3_MFMA || value == 0 || value == 16 || value == 32, "stem_tuning_set: fx3_mfma is 0 (automatic), 16 or 32 (rows of the MFMA shape)");
            g_tuning[i] = value;
            return 0;
        }
    stem_set_error("stem_tuning_set: unknown selector '%s' (fx3_tile, fx3_split, wg3_split, arp_workers, fx3_depth, fx3_gen_tile, fx3_mfma, fx3_gen_mfma, fx3_gen_img, fx3_img_w, wg3_row, wg3_minch, tconv_cps, arp_giveup_at, unpack_mb)", name);
    return -1;
}
STEM_EXPORT int stem_tuning_get(const char *name)
{
    for (int i = 0; name && i < STEM_TUNE_COUNT; ++i)
        if (!strcmp(name, kTuningNames[i])) return g_tuning[i];
    return -1;
}

namespace {

// out[t][a][b] = w[a*sa + b*sb + t]  (taps are the innermost axis of OIHW and IOHW).
// One block per `a`: gather the Bd x T slab into LDS reading runs of T floats, write rows of Bd.
__global__ __launch_bounds__(256) void pack_kernel(float *w, float *out, int A, int Bd, int T, long sa, long sb,
                                                   int R, int S, int masked)
{
    extern __shared__ float tile[];   // [Bd][T]
    const int a = blockIdx.x;
    const int n = Bd * T;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / T, t = i - b * T;
        float v = w[a * sa + b * sb + t];
        if (masked & 3) {   // MaskedConv2d (layers.py:39-42): row > R/2, or row == R/2 and col >= S/2 (type A) / col > S/2 (type B, bit 2)
            const int r = t / S, s = t - r * S;
            if (r > R / 2 || (r == R / 2 && s >= S / 2 + ((masked >> 2) & 1))) {
                v = 0.f;
                if ((masked & 3) == 2) w[a * sa + b * sb + t] = 0.f;   // `self.weight.data *= self.mask`, in place
            }
        }
        tile[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / Bd, b = i - t * Bd;
        out[((size_t)t * A + a) * Bd + b] = tile[b * T + t];
    }
}

// ---- multi-tensor variants: all layers of a model in ONE launch (26 packs / 13 unpacks per optimiser step
// would otherwise be 39 latency-bound launches of 10-30 us each) ------------------------------------------
constexpr int MAXD = 32;
struct PackD {
    float *w;
    float *out;
    int A, Bd, T, R, S, masked, block0;
    long sa, sb;
};
struct PackTable {
    PackD d[MAXD];
    int n;
};
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackTable tb)
{
    extern __shared__ float tile[];
    int i = 0;
    while (i + 1 < tb.n && (int)blockIdx.x >= tb.d[i + 1].block0) ++i;
    const PackD d = tb.d[i];
    const int a = blockIdx.x - d.block0;
    const int n = d.Bd * d.T;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int b = k / d.T, t = k - b * d.T;
        float v = d.w[a * d.sa + b * d.sb + t];
        if (d.masked & 3) {
            const int r = t / d.S, s = t - r * d.S;
            if (r > d.R / 2 || (r == d.R / 2 && s >= d.S / 2 + ((d.masked >> 2) & 1))) {
                v = 0.f;
                if ((d.masked & 3) == 2) d.w[a * d.sa + b * d.sb + t] = 0.f;
            }
        }
        tile[k] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int t = k / d.Bd, b = k - t * d.Bd;
        d.out[((size_t)t * d.A + a) * d.Bd + b] = tile[b * d.T + t];
    }
}

struct UnpackD {
    const float *dwp;
    float *dw;
    int A, Bd, T, splits, block0, accumulate, mb;
};
struct UnpackTable {
    UnpackD d[MAXD];
    int n;
};
// one workgroup per (row a, mb-wide range of b; mb = 64 .. 192 in steps of 32): reads T coalesced runs of mb floats per split, transposes the [t][b]
// tile through LDS and writes one contiguous mb*T run.  (The first version used one workgroup per row a with a serial loop
// over all Bd*T elements x splits: ~1000 workgroups of dependent loads ran at 0.5 TB/s, 2 ms per bench step.)  Round 6: ranges of up
// to 192 floats (768-byte runs) where they divide the row -- the pass gathers its input in runs, and longer runs are what
// HBM delivers faster; the sums and their order are unchanged.
constexpr int UNPACK_MB = 64, UNPACK_MB_MAX = 192;
__global__ __launch_bounds__(256) void unpack_multi_kernel(const UnpackTable tb)
{
    extern __shared__ float tile[];   // [mb][T+1]
    int i = 0;
    while (i + 1 < tb.n && (int)blockIdx.x >= tb.d[i + 1].block0) ++i;
    const UnpackD d = tb.d[i];
    const int mb = d.mb;
    const int chunks = (d.Bd + mb - 1) / mb;
    const int blk = blockIdx.x - d.block0;
    const int a = blk / chunks, b0 = (blk - a * chunks) * mb;
    const int nb = d.Bd - b0 < mb ? d.Bd - b0 : mb;
    const int n = nb * d.T;
    const size_t slab = (size_t)d.T * d.A * d.Bd;
    // Round 5: a full range whose rows start on 16-byte boundaries is read as float4 (four consecutive b per thread) with
    // the slabs of a piece in flight together, and written back as float4: the same sums in the same order (even slabs into one
    // accumulator, odd into the other), a third fewer instructions per byte -- the pass is HBM-bound and sits at the tail of backward
    const bool vec = nb == mb && (d.Bd & 3) == 0 && ((reinterpret_cast<uintptr_t>(d.dwp) | reinterpret_cast<uintptr_t>(d.dw)) & 15) == 0 &&
                     ((size_t)d.Bd * d.T) % 4 == 0 && (slab & 3) == 0;
    if (vec) {
        const int q = mb / 4, n4 = d.T * q;                      // q = 16 or 32 float4 pieces per run
        for (int k = threadIdx.x; k < n4; k += 256) {
            const int t = k / q, b4 = k - t * q;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(d.dwp + ((size_t)t * d.A + a) * d.Bd + b0) + b4;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            int sp = 0;
            for (; sp + 3 < d.splits; sp += 4) {
                const f32x4 x0 = src[(size_t)sp * (slab / 4)], x1 = src[(size_t)(sp + 1) * (slab / 4)];
                const f32x4 x2 = src[(size_t)(sp + 2) * (slab / 4)], x3 = src[(size_t)(sp + 3) * (slab / 4)];
                v0 += x0; v1 += x1; v0 += x2; v1 += x3;
            }
            for (; sp + 1 < d.splits; sp += 2) {
                const f32x4 x0 = src[(size_t)sp * (slab / 4)], x1 = src[(size_t)(sp + 1) * (slab / 4)];
                v0 += x0; v1 += x1;
            }
            if (sp < d.splits) v0 += src[(size_t)sp * (slab / 4)];
            const f32x4 v = v0 + v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[(b4 * 4 + j) * (d.T + 1) + t] = v[j];
        }
        __syncthreads();
        float *dst = d.dw + ((size_t)a * d.Bd + b0) * d.T;        // mb * T consecutive floats, 16-byte aligned (Bd * T % 4 == 0, b0 % 64 == 0)
        for (int k4 = threadIdx.x; k4 < n / 4; k4 += 256) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * k4 + j, b = k / d.T, t = k - b * d.T;
                v[j] = tile[b * (d.T + 1) + t];
            }
            f32x4 *o = reinterpret_cast<f32x4 *>(dst) + k4;
            *o = d.accumulate ? *o + v : v;
        }
        return;
    }
    for (int k = threadIdx.x; k < n; k += 256) {
        const int t = k / nb, b = k - t * nb;
        const float *src = d.dwp + ((size_t)t * d.A + a) * d.Bd + b0 + b;
        float v0 = 0.f, v1 = 0.f;
        int s = 0;
        for (; s + 1 < d.splits; s += 2) {
            v0 += src[s * slab];
            v1 += src[(s + 1) * slab];
        }
        if (s < d.splits) v0 += src[s * slab];
        tile[b * (d.T + 1) + t] = v0 + v1;
    }
    __syncthreads();
    float *dst = d.dw + ((size_t)a * d.Bd + b0) * d.T;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int b = k / d.T, t = k - b * d.T;
        const float v = tile[b * (d.T + 1) + t];
        dst[k] = d.accumulate ? dst[k] + v : v;
    }
}

// first layer: w[K][C<=4][T] -> out[K][32][4], zero padded
__global__ void pack_c4_kernel(const float *w, float *out, int K, int C, int T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * 128) return;
    const int k = i >> 7, r = i & 127, t = r >> 2, c = r & 3;
    out[i] = (t < T && c < C) ? w[((size_t)k * C + c) * T + t] : 0.f;
}

// dw[(a*Bd + b)*T + t] (+)= sum_s dwp[s][t][a][b]; one workgroup per (a, 32-wide range of b): A * Bd/32 workgroups
constexpr int UNPACK_BT = 32;
__global__ __launch_bounds__(256) void unpack_kernel(const float *dwp, float *dw, int A, int Bd, int T, int splits, int accumulate)
{
    extern __shared__ float tile[];   // [UNPACK_BT][T+1]
    const int a = blockIdx.x, b0 = blockIdx.y * UNPACK_BT;
    const int nb = Bd - b0 < UNPACK_BT ? Bd - b0 : UNPACK_BT;
    const int n = nb * T;
    const size_t slab = (size_t)T * A * Bd;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / nb, b = i - t * nb;
        const float *src = dwp + ((size_t)t * A + a) * Bd + b0 + b;
        float v0 = 0.f, v1 = 0.f;
        int s = 0;
        for (; s + 1 < splits; s += 2) {
            v0 += src[s * slab];
            v1 += src[(s + 1) * slab];
        }
        if (s < splits) v0 += src[s * slab];
        tile[b * (T + 1) + t] = v0 + v1;
    }
    __syncthreads();
    float *dst = dw + ((size_t)a * Bd + b0) * T;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / T, t = i - b * T;
        const float v = tile[b * (T + 1) + t];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

// [B][C][HW] -> [B][HW][ld]  (32x32 LDS tiles)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *x, float *y, int ldy, int C, int HW)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) y[((size_t)b * HW + p) * ldy + c] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *x, int ldx, float *y, int C, int HW, int clamp01)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((size_t)b * HW + p) * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) {
            float v = tile[tx][r];
            if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
            y[((size_t)b * C + c) * HW + p] = v;
        }
    }
}

// dst[p*ldd + c] = src[p*lds + c]   (channel-slice copy: the only thing left of torch.cat)
__global__ void copy2d_kernel(const float *src, int lds, float *dst, int ldd, size_t npix, int C)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t p = i / C;
    const int c = (int)(i - p * C);
    dst[p * ldd + c] = src[p * lds + c];
}

// 1024 pixels per workgroup; q (optional): scale record of the image, one slot of max |x| per workgroup (stem_common.h) --
// what the first-layer kernel c4gdn_f16x3.hip scales its in-register fp16 split of the patches by
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float *x, f32x4 *y, size_t HW, size_t total, float *q)
{
    __shared__ float qred[16];
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = ((size_t)blockIdx.x * 4 + k) * 256 + threadIdx.x;
        if (i >= total) break;
        const size_t b = i / HW, p = i - b * HW;
        const float *src = x + b * 3 * HW + p;
        f32x4 v = {src[0], src[HW], src[2 * HW], 0.f};
        y[i] = v;
        m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2])));
    }
    if (!q) return;
    m = block_max(m, qred);
    if (threadIdx.x == 0) {
        q[QREC_HDR + blockIdx.x] = m;
        if (blockIdx.x == 0) {
            q_header(q, gridDim.x);
            q[1] = 1.f;
        }
    }
}

}   // namespace

STEM_EXPORT size_t stem_packed_weight_elems(int K, int C, int R, int S, int role)
{
    if (role == STEM_PACK_CONV_FWD_C4) return (size_t)K * 128;
    return (size_t)K * C * R * S;
}

STEM_EXPORT int stem_pack_weight(const float *w, float *wp, int K, int C, int R, int S, int role, int masked, void *stream)
{
    STEM_CHECK_ARG(w && wp, "stem_pack_weight: null pointer");
    STEM_CHECK_ARG(K > 0 && C > 0 && R > 0 && S > 0 && R * S <= 25, "stem_pack_weight: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const int T = R * S;
    int A, Bd;
    long sa, sb;
    switch (role) {
    case STEM_PACK_CONV_FWD:     A = K; Bd = C; sa = (long)C * T; sb = T; break;            // w[K][C][T] -> [t][K][C]
    case STEM_PACK_CONV_DGRAD:   A = C; Bd = K; sa = T; sb = (long)C * T; break;            // w[K][C][T] -> [t][C][K]
    case STEM_PACK_DECONV_FWD:   A = K; Bd = C; sa = T; sb = (long)K * T; break;            // w[C][K][T] -> [t][K][C]
    case STEM_PACK_DECONV_DGRAD: A = C; Bd = K; sa = (long)K * T; sb = T; break;            // w[C][K][T] -> [t][C][K]
    case STEM_PACK_CONV_FWD_C4:
        STEM_CHECK_ARG(C <= 4 && T <= 32, "stem_pack_weight: C4 role needs C<=4, taps<=32");
        hipLaunchKernelGGL(pack_c4_kernel, dim3(cdiv(K * 128, 256)), dim3(256), 0, st, w, wp, K, C, T);
        STEM_LAUNCH_CHECK("pack_c4");
        return 0;
    default:
        stem_set_error("stem_pack_weight: unknown role %d", role);
        return -1;
    }
    const size_t lds = (size_t)Bd * T * sizeof(float);
    STEM_CHECK_ARG(lds <= 160 * 1024, "stem_pack_weight: slab of %zu B exceeds LDS", lds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pack_kernel, dim3(A), dim3(256), lds, st, const_cast<float *>(w), wp, A, Bd, T, sa, sb, R, S, masked);
    STEM_LAUNCH_CHECK("pack");
    return 0;
}

static int role_geometry(int role, int K, int C, int T, int *A, int *Bd, long *sa, long *sb)
{
    switch (role) {
    case STEM_PACK_CONV_FWD:     *A = K; *Bd = C; *sa = (long)C * T; *sb = T; return 0;
    case STEM_PACK_CONV_DGRAD:   *A = C; *Bd = K; *sa = T; *sb = (long)C * T; return 0;
    case STEM_PACK_DECONV_FWD:   *A = K; *Bd = C; *sa = T; *sb = (long)K * T; return 0;
    case STEM_PACK_DECONV_DGRAD: *A = C; *Bd = K; *sa = (long)K * T; *sb = T; return 0;
    default: return -1;
    }
}

STEM_EXPORT int stem_pack_weights_multi(const stem_pack_desc *descs, int n, void *stream)
{
    STEM_CHECK_ARG(descs && n >= 0, "stem_pack_weights_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += MAXD) {
        PackTable tb;
        tb.n = n - base < MAXD ? n - base : MAXD;
        int blocks = 0;
        size_t lds = 0;
        for (int i = 0; i < tb.n; ++i) {
            const stem_pack_desc &q = descs[base + i];
            PackD &d = tb.d[i];
            const int T = q.R * q.S;
            STEM_CHECK_ARG(q.w && q.wp && T >= 1 && T <= 25, "stem_pack_weights_multi: bad descriptor %d", base + i);
            STEM_CHECK_ARG(role_geometry(q.role, q.K, q.C, T, &d.A, &d.Bd, &d.sa, &d.sb) == 0,
                           "stem_pack_weights_multi: role %d not supported in the multi-tensor path", q.role);
            d.w = const_cast<float *>(q.w);
            d.out = q.wp;
            d.T = T; d.R = q.R; d.S = q.S; d.masked = q.masked;
            d.block0 = blocks;
            blocks += d.A;
            const size_t need = (size_t)d.Bd * T * sizeof(float);
            if (need > lds) lds = need;
        }
        STEM_CHECK_ARG(lds <= 160 * 1024, "stem_pack_weights_multi: slab of %zu B exceeds LDS", lds);
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)pack_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (blocks) hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks), dim3(256), lds, st, tb);
        STEM_LAUNCH_CHECK("pack_multi");
    }
    return 0;
}

STEM_EXPORT int stem_unpack_wgrads_multi(const stem_unpack_desc *descs, int n, void *stream)
{
    STEM_CHECK_ARG(descs && n >= 0, "stem_unpack_wgrads_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += MAXD) {
        UnpackTable tb;
        tb.n = n - base < MAXD ? n - base : MAXD;
        int blocks = 0;
        size_t lds = 0;
        for (int i = 0; i < tb.n; ++i) {
            const stem_unpack_desc &q = descs[base + i];
            UnpackD &d = tb.d[i];
            STEM_CHECK_ARG(q.dwp && q.dw && q.splits >= 1, "stem_unpack_wgrads_multi: bad descriptor %d", base + i);
            d.dwp = q.dwp; d.dw = q.dw;
            d.T = q.R * q.S;
            const int deconv = q.flags & STEM_UNPACK_DECONV;
            d.accumulate = (q.flags & STEM_UNPACK_ACCUMULATE) ? 1 : 0;
            d.A = deconv ? q.C : q.K;
            d.Bd = deconv ? q.K : q.C;
            d.splits = q.splits;
            d.block0 = blocks;
            // 128-wide ranges where they divide the row and leave enough workgroups to fill the chip (sweeps: stem_tuning_set("unpack_mb", 64 | 128))
            const int forced_mb = stem_tuning(STEM_TUNE_UNPACK_MB);
            d.mb = UNPACK_MB;
            if (forced_mb) {
                d.mb = forced_mb;
            } else {
                for (int mb = UNPACK_MB_MAX; mb > UNPACK_MB; mb -= 32)           // the longest run that divides the row and still fills the chip twice
                    if (d.Bd % mb == 0 && (long)d.A * (d.Bd / mb) >= 512) {
                        d.mb = mb;
                        break;
                    }
            }
            blocks += d.A * cdiv(d.Bd, d.mb);
            const size_t need = (size_t)d.mb * (d.T + 1) * sizeof(float);
            if (need > lds) lds = need;
        }
        if (blocks) hipLaunchKernelGGL(unpack_multi_kernel, dim3(blocks), dim3(256), lds, st, tb);
        STEM_LAUNCH_CHECK("unpack_multi");
    }
    return 0;
}

STEM_EXPORT int stem_unpack_wgrad(const float *dwp, float *dw, int K, int C, int R, int S, int splits, int flags,
                                  void *stream)
{
    STEM_CHECK_ARG(dwp && dw && splits >= 1, "stem_unpack_wgrad: bad arguments");
    const int T = R * S;
    const int deconv = flags & STEM_UNPACK_DECONV;
    // Conv2d: slabs [t][K][C] -> dw[K][C][T].  ConvTranspose2d: slabs [t][C][K] -> dw[C][K][T].
    const int A = deconv ? C : K, Bd = deconv ? K : C;
    const size_t lds = (size_t)UNPACK_BT * (T + 1) * sizeof(float);
    STEM_CHECK_ARG(lds <= 64 * 1024, "stem_unpack_wgrad: %d taps exceed the staging tile", T);
    hipLaunchKernelGGL(unpack_kernel, dim3(A, cdiv(Bd, UNPACK_BT)), dim3(256), lds, (hipStream_t)stream, dwp, dw, A, Bd, T, splits,
                       (flags & STEM_UNPACK_ACCUMULATE) ? 1 : 0);
    STEM_LAUNCH_CHECK("unpack");
    return 0;
}

STEM_EXPORT int stem_nchw_to_nhwc(const float *x, float *y, int ldy, int B, int C, int H, int W, void *stream)
{
    STEM_CHECK_ARG(x && y && ldy >= C, "stem_nchw_to_nhwc: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, x, y, ldy, C, HW);
    STEM_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}

STEM_EXPORT int stem_nhwc_to_nchw(const float *x, int ldx, float *y, int B, int C, int H, int W, int clamp01, void *stream)
{
    STEM_CHECK_ARG(x && y && ldx >= C, "stem_nhwc_to_nchw: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, x, ldx, y, C, HW, clamp01);
    STEM_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}

STEM_EXPORT size_t stem_nhwc4_qrec_floats(int B, int H, int W) { return QREC_HDR + cdivz((size_t)B * H * W, 1024); }

STEM_EXPORT int stem_nchw3_to_nhwc4(const float *x, float *y, int B, int H, int W, float *q, void *stream)
{
    STEM_CHECK_ARG(x && y, "stem_nchw3_to_nhwc4: null pointer");
    const size_t HW = (size_t)H * W, total = HW * B;
    if (total == 0) return 0;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3((unsigned)cdivz(total, 1024)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<f32x4 *>(y), HW, total, q);
    STEM_LAUNCH_CHECK("nchw3_to_nhwc4");
    return 0;
}

STEM_EXPORT int stem_copy_channels(const float *src, int lds, float *dst, int ldd, size_t npix, int C, void *stream)
{
    STEM_CHECK_ARG(src && dst && lds >= C && ldd >= C, "stem_copy_channels: bad arguments");
    if (npix == 0) return 0;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)cdivz(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, npix, C);
    STEM_LAUNCH_CHECK("copy_channels");
    return 0;
}
