// Weight-gradient kernels (fp32 MFMA) for nn.Conv2d / nn.ConvTranspose2d on NHWC activations.
//
//   out[split][t][i][j] = sum_{m in split}  P[m][i] * G[g(m,t)][j]
// P is the tensor that lives on the loop grid (dY for Conv2d, x for ConvTranspose2d),
// G the tensor gathered at (py*stride - pad + r, px*stride - pad + s) (x for Conv2d, dY for
// ConvTranspose2d).  The reduction runs over pixels, which is the slow axis of both NHWC
// operands, so both tiles are staged "pixel-major" ([32 px][channels]) and the MFMA operand
// fetch is a conflict-free ds_read_b32 (consecutive lanes = consecutive channels).
// Split-K over pixel ranges writes separate slabs that stem_unpack_wgrad sums in a fixed order
// (bit-reproducible, no float atomics).
#include <stdlib.h>

#include <type_traits>

#include "stem_common.h"

namespace {

constexpr int KP = 32;   // pixels per chunk
#ifndef WGRAD_PAD
#define WGRAD_PAD 0
#endif

struct WgradArgs {
    const float *p, *g;
    float *out;
    int ldp, ldg, CP, CG;
    int B, PH, PW, GH, GW;
    int stride, pad, R, S;
    int splits, chunks_per_split, nchunks;
    const int *ptab;     // [M][2]: byte offset of the gathered pixel for tap (0,0), validity mask of the R*S taps
    int pbytes, gbytes, tbytes;
    int gsq;             // square the gathered operand while staging (GDN: dgamma = sum g (x) x^2)
    float *dbp;          // [splits][CP] partial column sums of P (the Conv2d bias gradient) written by the tap-0 / first-column
                         // workgroups from the registers they stage anyway; null = not wanted (vector path only)
};

// one entry per loop-grid pixel; removes the per-chunk div/mod and bounds tests from the gather
__global__ void wgrad_pixtab_kernel(int *ptab, int Mtot, int PH, int PW, int GH, int GW, int ldg, int stride, int pad, int R, int S)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mtot) return;
    const int phw = PH * PW;
    const int b = m / phw, rem = m - b * phw;
    const int py = rem / PW, px = rem - py * PW;
    const int gy0 = py * stride - pad, gx0 = px * stride - pad;
    unsigned mask = 0;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s)
            if (gy0 + r >= 0 && gy0 + r < GH && gx0 + s >= 0 && gx0 + s < GW) mask |= 1u << (r * S + s);
    ptab[2 * m] = ((b * GH + gy0) * GW + gx0) * ldg * 4;
    ptab[2 * m + 1] = (int)mask;
}

// second launch-bound = waves per SIMD the register allocation must leave room for (the LDS footprint allows 2 / 3 / 3 / 4
// workgroups per CU for the four tiles; without it the 128x128 tile takes 284 VGPRs = one workgroup per CU)
//
// C4: the gathered operand has 4 channels at pitch 4 (image + quality map, or an image gradient padded to 4): the taps
// are folded into the tile's N axis -- column j = (t mod BN/4) * 4 + c, BN/4 taps per tile -- instead of one workgroup
// row per tap with a 4-of-BN filled tile.
template <int BM, int BN, int WM, int WN, bool VEC, bool C4 = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, ((BM / WM) * (BN / WN) == 8 ? 4 : BM * BN >= 128 * 128 ? 2 : BM * BN >= 128 * 64 ? 3 : 5))
void wgrad_kernel(const WgradArgs a)
{
    constexpr int NT = (BM / WM) * (BN / WN) * 64;          // 4 wavefronts, or 8 for the 128x128 tile with 32x64 wave tiles
    static_assert(!C4 || VEC, "the folded-tap mode uses the vector path");
    constexpr int TPT = BN / 4;                             // taps per tile (C4)
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WCOLS = BN / WN;
    // LDS pitches = tile widths, no padding: the operand fetch is ds_read_b32 (lane groups are the two 32-lane halves, which
    // read different rows and never share a cycle) and the staging writes are 16-byte runs of 8 consecutive lanes, so both
    // are conflict-free at any pitch; unpadded, the 64x64 tile takes exactly 32 KiB = five workgroups per CU.
    constexpr int PA = BM + WGRAD_PAD, PB = BN + WGRAD_PAD;
    constexpr int F4A = BM / 4, F4B = BN / 4;               // float4 per pixel row
    constexpr int RPA = NT / F4A, RPB = NT / F4B;           // pixel rows per pass
    constexpr int NPA = KP / RPA, NPB = KP / RPB;           // passes
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                    // [2][KP][PA]
    float *Gs = smem + 2 * KP * PA;      // [2][KP][PB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_j = C4 ? (a.R * a.S + TPT - 1) / TPT : (a.CG + BN - 1) / BN;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (private L2 each).  All tiles x taps of one
    // pixel range (split) re-read the same dY / x rows, so consecutive *virtual* ids -- split slowest -- are handed to
    // the same XCD: the rows are fetched from HBM into one L2 instead of eight.  Any placement is correct.
    int vid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    {
        const int nb = gridDim.x * gridDim.y * gridDim.z, qq = nb >> 3, rr = nb & 7, xcd = vid & 7, idx = vid >> 3;
        if (nb >= 16) vid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tile = vid % (int)gridDim.x, vt = vid / (int)gridDim.x;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * BM, j0 = tj * BN;
    const int split = vt / (int)gridDim.y;
    const int colB_ = ((int)threadIdx.x % (BN / 4)) * 4;
    const int t = C4 ? tj * TPT + colB_ / 4 : vt % (int)gridDim.y;      // C4: this thread's own tap
    const int tr = t / a.S, ts = t - tr * a.S;
    const int Mtot = a.B * a.PH * a.PW, phw = a.PH * a.PW;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    const int rowA = tid / F4A, colA = (tid - rowA * F4A) * 4;
    const int rowB = tid / F4B, colB = (tid - rowB * F4B) * 4;
    f32x4 ra[NPA], rb[NPB];

    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.p), 0, a.pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.g), 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.ptab), 0, a.tbytes, 0x00020000);
    const int tapoff = (tr * a.GW + ts) * a.ldg * 4;          // byte offset of this workgroup's tap
    const bool colA_ok = i0 + colA < a.CP, colB_ok = C4 ? t < a.R * a.S : j0 + colB < a.CG;
    const bool do_bias = VEC && a.dbp != nullptr && tj == 0 && (C4 || t == 0);      // workgroup-uniform
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    auto gload = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int m = chunk * KP + rowA + q * RPA;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < Mtot) {
                const float *src = a.p + (size_t)m * a.ldp + i0 + colA;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + colA + e < a.CP) v[e] = src[e];
            }
            ra[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const int m = chunk * KP + rowB + q * RPB;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < Mtot) {
                const int b = m / phw, rem = m - b * phw;
                const int py = rem / a.PW, px = rem - py * a.PW;
                const int gy = py * a.stride - a.pad + tr, gx = px * a.stride - a.pad + ts;
                if (gy >= 0 && gy < a.GH && gx >= 0 && gx < a.GW) {
                    const float *src = a.g + (size_t)((b * a.GH + gy) * a.GW + gx) * a.ldg + j0 + colB;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j0 + colB + e < a.CG) v[e] = src[e];
                }
            }
            rb[q] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NPA; ++q)
            *reinterpret_cast<f32x4 *>(&Ps[(buf * KP + rowA + q * RPA) * PA + colA]) = ra[q];
#pragma unroll
        for (int q = 0; q < NPB; ++q)
            *reinterpret_cast<f32x4 *>(&Gs[(buf * KP + rowB + q * RPB) * PB + colB]) = rb[q];
    };

    const int wm0 = (wave / WCOLS) * WM, wn0 = (wave % WCOLS) * WN;
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (VEC) {
        // Woven pipeline (same scheme as igemm.hip): two register sets run two chunks ahead, their gather-table entries
        // one step further; LDS writes and global loads are placed between the MFMA groups of the current chunk, all in
        // one basic block (masks instead of branches), so the scheduler can hide them under the 64-cycle MFMAs.
        f32x4 raB[NPA], rbB[NPB];
        i32x2 ptA[NPB], ptB[NPB];
        auto tl = [&](int chunk, i32x2 (&p)[NPB]) {
#pragma unroll
            for (int q = 0; q < NPB; ++q)
                p[q] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rt, (chunk * KP + rowB + q * RPB) * 8, 0, 0));
        };
        auto glA = [&](int chunk, f32x4 (&r)[NPA]) {
#pragma unroll
            for (int q = 0; q < NPA; ++q) {
                const int m = chunk * KP + rowA + q * RPA;
                const int mk = -(int)(m < Mtot && colA_ok);
                const int off = (((m * a.ldp + i0 + colA) * 4) & mk) | (0x7FFFFF00 & ~mk);
                r[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, 0));
            }
        };
        auto glB = [&](f32x4 (&r)[NPB], const i32x2 (&p)[NPB]) {
#pragma unroll
            for (int q = 0; q < NPB; ++q) {
                const int mk = -(int)(((unsigned)p[q][1] >> (t & 31)) & 1u) & -(int)colB_ok;
                const int off = ((p[q][0] + tapoff + (C4 ? 0 : (j0 + colB) * 4)) & mk) | (0x7FFFFF00 & ~mk);
                r[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
            }
        };
        auto stA = [&](int buf, const f32x4 (&r)[NPA]) {
#pragma unroll
            for (int q = 0; q < NPA; ++q) *reinterpret_cast<f32x4 *>(&Ps[(buf * KP + rowA + q * RPA) * PA + colA]) = r[q];
        };
        // `gsq` (GDN backward only) and `do_bias` (the tap-0 / first-column workgroups only) are workgroup-uniform, but as
        // run-time conditions inside the loop they were if-converted into 64 VALU selects / multiplies / adds per pair of
        // chunks for EVERY workgroup (3.2 VALU instructions per MFMA; each costs ~4 issue cycles next to a 64-cycle MFMA).
        // The loop is therefore instantiated per (GSQ, BIAS) and chosen by two uniform branches.
        auto main_loop = [&](auto gsq_c, auto bias_c) {
            constexpr bool GSQ = decltype(gsq_c)::value, BIAS = decltype(bias_c)::value;
            auto stB = [&](int buf, const f32x4 (&r)[NPB]) {
#pragma unroll
                for (int q = 0; q < NPB; ++q) {
                    if constexpr (GSQ)
                        *reinterpret_cast<f32x4 *>(&Gs[(buf * KP + rowB + q * RPB) * PB + colB]) = r[q] * r[q];
                    else
                        *reinterpret_cast<f32x4 *>(&Gs[(buf * KP + rowB + q * RPB) * PB + colB]) = r[q];
                }
            };
            auto step = [&](int cur, f32x4 (&sa)[NPA], f32x4 (&sb)[NPB], i32x2 (&p)[NPB], int cn) {
                const float *Ab = Ps + (cur * KP + lh) * PA + wm0 + lr;
                const float *Bb = Gs + (cur * KP + lh) * PB + wn0 + lr;
#pragma unroll
                for (int ks = 0; ks < KP / 2; ++ks) {
                    float af[TM], bf[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = Ab[ks * 2 * PA + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = Bb[ks * 2 * PB + j * 32];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                    if (ks == 2) {
                        stA(cur ^ 1, sa);
                        if constexpr (BIAS) {
                            if (cn - 2 < c_end) {      // `sa` holds chunk cn - 2: count it once, only if it is ours
#pragma unroll
                                for (int q = 0; q < NPA; ++q) bsum += sa[q];
                            }
                        }
                    }
                    if (ks == 5) stB(cur ^ 1, sb);
                    if (ks == 9) glA(cn, sa);
                    if (ks == 12) glB(sb, p);
                    if (ks == 14) tl(cn + 2, p);
                }
            };
            if (c_begin < c_end) {
                tl(c_begin, ptA);
                glA(c_begin, ra);
                glB(rb, ptA);
                stA(0, ra);
                if constexpr (BIAS) {
#pragma unroll
                    for (int q = 0; q < NPA; ++q) bsum += ra[q];
                }
                stB(0, rb);
                tl(c_begin + 1, ptA);
                tl(c_begin + 2, ptB);
                glA(c_begin + 1, ra);
                glB(rb, ptA);
                glA(c_begin + 2, raB);
                glB(rbB, ptB);
                tl(c_begin + 3, ptA);
                tl(c_begin + 4, ptB);
            }
            __syncthreads();
            int c = c_begin;
            for (; c + 1 < c_end; c += 2) {
                step(0, ra, rb, ptA, c + 3);          // chunk c; stage c+1 -> buffer 1; prefetch c+3 (table c+5)
                __syncthreads();
                step(1, raB, rbB, ptB, c + 4);        // chunk c+1; stage c+2 -> buffer 0; prefetch c+4 (table c+6)
                __syncthreads();
            }
            if (c < c_end) step(0, ra, rb, ptA, c + 3);      // odd tail (its prefetches are masked out of range)
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (a.gsq) {
            if (do_bias) main_loop(T_{}, T_{}); else main_loop(T_{}, F_{});
        } else {
            if (do_bias) main_loop(F_{}, T_{}); else main_loop(F_{}, F_{});
        }
    } else if (c_begin < c_end) {
        gload(c_begin);
        sstore(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            if (c + 1 < c_end) gload(c + 1);
            const float *Ab = Ps + (cur * KP + lh) * PA + wm0 + lr;
            const float *Bb = Gs + (cur * KP + lh) * PB + wn0 + lr;
#pragma unroll
            for (int ks = 0; ks < KP / 2; ++ks) {
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = Ab[ks * 2 * PA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = Bb[ks * 2 * PB + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            if (c + 1 < c_end) sstore(cur ^ 1);
            __syncthreads();
        }
    }

    if (do_bias) {      // column sums of this workgroup's P rows: RPA partial sums per column -> one, written to the split's slab
        __syncthreads();
        float *red = smem;                                    // [RPA][BM]
        *reinterpret_cast<f32x4 *>(&red[rowA * BM + colA]) = bsum;
        __syncthreads();
        if (tid < BM && i0 + tid < a.CP) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RPA; ++r) v += red[r * BM + tid];
            a.dbp[(size_t)split * a.CP + i0 + tid] = v;
        }
    }
    if (C4) {
        float *outs = a.out + (size_t)split * a.R * a.S * a.CP * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int jl = wn0 + j * 32 + lr, tt = tj * TPT + (jl >> 2);
            if (tt >= a.R * a.S) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ii = i0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (ii < a.CP) outs[((size_t)tt * a.CP + ii) * 4 + (jl & 3)] = acc[i][j][r];
                }
        }
        return;
    }
    float *outp = a.out + ((size_t)split * a.R * a.S + t) * a.CP * a.CG;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jj = j0 + wn0 + j * 32 + lr;
        if (jj >= a.CG) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ii = i0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ii < a.CP) outp[(size_t)ii * a.CG + jj] = acc[i][j][r];
            }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const WgradArgs &a, bool vec, hipStream_t st)
{
    dim3 grid(cdiv(a.CP, BM) * cdiv(a.CG, BN), a.R * a.S, a.splits), block((BM / WM) * (BN / WN) * 64);
    const size_t lds = (size_t)2 * KP * (BM + BN + 2 * WGRAD_PAD) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)wgrad_kernel<BM, BN, WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)wgrad_kernel<BM, BN, WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (vec)
        hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, true>), grid, block, lds, st, a);
    else
        hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, false>), grid, block, lds, st, a);
    STEM_LAUNCH_CHECK("wgrad");
    return 0;
}

int launch_c4(const WgradArgs &a, hipStream_t st)
{
    constexpr int BM = 64, BN = 64;
    dim3 grid(cdiv(a.CP, BM) * cdiv(a.R * a.S, BN / 4), 1, a.splits), block(256);
    const size_t lds = (size_t)2 * KP * (BM + BN + 2 * WGRAD_PAD) * sizeof(float);
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, 32, 32, true, true>), grid, block, lds, st, a);
    STEM_LAUNCH_CHECK("wgrad_c4");
    return 0;
}

struct WTile {
    int bm, bn;
    float eff;
};
constexpr int NCFG = 5;
// relative MFMA efficiencies measured on exactly fitting problems (tools/wgrad_sweep.py): the tiles differ by < 10 %, what
// decides is padding, split count and resident wavefronts.  [4] = 128x128 with 8 wavefronts (32x64 wave tiles).
constexpr WTile kWT[NCFG] = {{128, 128, 0.97f}, {128, 64, 0.93f}, {64, 128, 0.92f}, {64, 64, 0.92f}, {128, 128, 1.00f}};

// tile + split choice: minimise padded MFMA work / efficiency, split the pixel reduction to fill the chip
void plan(int CP, int CG, int T, int nchunks, int *cfg, int *splits)
{
    double best = 1e300;
    *cfg = 0;
    *splits = 1;
    if (CG == 4) {      // folded-tap mode (launch_c4): 64-row tiles x 16-tap tiles; split the pixels to ~1024 workgroups
        const int tiles = cdiv(CP, 64) * cdiv(T, 16);
        int s = 1024 / tiles, max_s = nchunks >= 8 ? nchunks / 8 : 1;
        if (s > max_s) s = max_s;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
        const int cps = cdiv(nchunks, s);
        *cfg = 3;
        *splits = cdiv(nchunks, cps);
        return;
    }
    const int slots[NCFG] = {512, 768, 768, 1280, 512};      // co-resident workgroups (LDS: 64 / 48 / 48 / 32 / 64 KiB)
    const int wps[NCFG] = {1, 1, 1, 1, 2};                   // wavefronts per SIMD contributed by one workgroup
    static const float eff4 = getenv("STEM_WGRAD_EFF4") ? (float)atof(getenv("STEM_WGRAD_EFF4")) : kWT[4].eff;     // tuning aid
    static const int forced = getenv("STEM_WGRAD_CFG") ? atoi(getenv("STEM_WGRAD_CFG")) : -1;      // tuning aid
    for (int c = 0; c < NCFG; ++c) {
        if (forced >= 0 && c != forced) continue;
        const float eff = c == 4 ? eff4 : kWT[c].eff;
        const long ti = cdiv(CP, kWT[c].bm), tj = cdiv(CG, kWT[c].bn);
        const long tiles = ti * tj * T;
        const int max_s = nchunks >= 8 ? (nchunks / 8 > 64 ? 64 : nchunks / 8) : 1;      // >= 8 chunks (256 px) per split
        for (int s = 1; s <= max_s; ++s) {
            const int cps = cdiv(nchunks, s);
            if (s > 1 && cdiv(nchunks, cps) != s) continue;
            const long blocks = tiles * s;
            // workgroups are dealt round-robin to the 256 CUs and share the CU's MFMA pipes: time ~ (blocks per CU) x
            // (MFMA work per block); fewer co-resident workgroups hide less latency (measured on the igemm twin kernel)
            const long per_cu = cdiv((int)blocks, 256);
            const int maxblk = slots[c] / 256;
            // one round = up to maxblk co-resident workgroups per CU sharing its MFMA pipes (n of them cost n / occf(n));
            // what does not fit runs in further rounds, each priced as a FULL round (the dispatcher packs the leftovers onto
            // the CUs that drain first; see the igemm planner for the counter evidence)
            double units;
            if (per_cu <= maxblk) {
                const long resident = per_cu * wps[c];                                    // wavefronts per SIMD
                const double occf = resident >= 4 ? 1.0 : resident == 3 ? 0.96 : resident == 2 ? 0.91 : 0.75;
                units = (double)per_cu / occf;
            } else {
                units = (double)cdiv((int)per_cu, maxblk) * maxblk;
            }
            double cost = units * (cps + 3) * kWT[c].bm * kWT[c].bn / eff;
            cost += 0.1 * (double)s * CP * CG * T * 32.0 / 256.0;                    // slab write + unpack read
            if (cost < best) {
                best = cost;
                *cfg = c;
                *splits = s;
            }
        }
    }
}

bool wgrad_vec_ok(const float *p, int ldp, int CP, const float *g, int ldg, int CG)
{
    return (CP % 4 == 0) && (CG % 4 == 0) && (ldp % 4 == 0) && (ldg % 4 == 0) && (((uintptr_t)p & 15) == 0) && (((uintptr_t)g & 15) == 0);
}

int run(const float *p, int ldp, int CP, const float *g, int ldg, int CG, float *out, int *ptab, int B, int PH, int PW,
        int GH, int GW, int R, int S, int stride, int pad, int splits, int flags, hipStream_t st, float *dbp = nullptr)
{
    WgradArgs a;
    a.p = p; a.g = g; a.out = out; a.dbp = dbp;
    a.ldp = ldp; a.ldg = ldg; a.CP = CP; a.CG = CG;
    a.B = B; a.PH = PH; a.PW = PW; a.GH = GH; a.GW = GW;
    a.stride = stride; a.pad = pad; a.R = R; a.S = S;
    a.nchunks = cdiv(B * PH * PW, KP);
    a.splits = splits;
    a.chunks_per_split = cdiv(a.nchunks, splits);
    const int Mtot = B * PH * PW;
    const long pb = (((long)Mtot - 1) * ldp + CP) * 4, gb = (((long)B * GH * GW - 1) * ldg + CG) * 4;
    if (pb >= 0x7FFFFF00L || gb >= 0x7FFFFF00L) {
        stem_set_error("wgrad: tensor view of %ld / %ld bytes exceeds the 2 GiB buffer-descriptor range", pb, gb);
        return -1;
    }
    a.pbytes = (int)pb;
    a.gbytes = (int)gb;
    a.tbytes = Mtot * 8;
    a.ptab = ptab;
    a.gsq = (flags & STEM_WGRAD_SQUARE_G) ? 1 : 0;
    if (a.gsq && !((CP % 4 == 0) && (CG % 4 == 0) && (ldp % 4 == 0) && (ldg % 4 == 0))) {
        stem_set_error("wgrad: STEM_WGRAD_SQUARE_G needs 16-byte aligned channel counts");
        return -1;
    }
    if (!(flags & STEM_WGRAD_TABLE_VALID)) {      // the table depends on geometry only: callers that keep `dwp` reuse it
        hipLaunchKernelGGL(wgrad_pixtab_kernel, dim3(cdiv(Mtot, 256)), dim3(256), 0, st, ptab, Mtot, PH, PW, GH, GW, ldg, stride, pad, R, S);
        STEM_LAUNCH_CHECK("wgrad_pixtab");
    }
    const bool vec = wgrad_vec_ok(p, ldp, CP, g, ldg, CG);
    int cfg, s_unused;
    plan(CP, CG, R * S, a.nchunks, &cfg, &s_unused);
    if (vec && CG == 4 && ldg == 4 && !a.gsq && R * S <= 32) return launch_c4(a, st);
    switch (cfg) {
    case 0: return launch_cfg<128, 128, 64, 64>(a, vec, st);
    case 1: return launch_cfg<128, 64, 64, 32>(a, vec, st);
    case 2: return launch_cfg<64, 128, 32, 64>(a, vec, st);
    case 4: return launch_cfg<128, 128, 32, 64>(a, vec, st);
    default: return launch_cfg<64, 64, 32, 32>(a, vec, st);
    }
}

// Bias gradient db[k] = sum_m dy[m][k] in two deterministic stages (no float atomics): `parts` pixel ranges x 64-channel
// tiles write partial column sums, then the partials are added in a fixed order.  HBM-bound (dy is read once, 16-byte
// loads, 4 rows in flight per thread); parts is sized so that the launch has ~2048 workgroups.
constexpr int CS_MAX_PARTS = 512;

inline int colsum_parts(size_t npix, int K)
{
    const int kt = cdiv(K, 64);
    long parts = 1536 / kt;
    const long by_rows = (long)((npix + 127) / 128);        // >= 128 rows per part
    if (parts > by_rows) parts = by_rows;
    if (parts > CS_MAX_PARTS) parts = CS_MAX_PARTS;
    return parts < 1 ? 1 : (int)parts;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *dy, int ld, size_t npix, int K, int parts, float *part)
{
    __shared__ float red[16][64 + 4];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;          // 16 float4 column groups x 16 row groups
    const int c = blockIdx.x * 64 + cg * 4;
    const size_t per = (npix + parts - 1) / parts;
    const size_t m0 = blockIdx.y * per, m1 = m0 + per < npix ? m0 + per : npix;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (VEC4) {
        if (c < K) {
            const float *base = dy + c;
            size_t m = m0 + rg;
            for (; m + 48 < m1; m += 64) {
                s0 += *reinterpret_cast<const f32x4 *>(base + m * ld);
                s1 += *reinterpret_cast<const f32x4 *>(base + (m + 16) * ld);
                s2 += *reinterpret_cast<const f32x4 *>(base + (m + 32) * ld);
                s3 += *reinterpret_cast<const f32x4 *>(base + (m + 48) * ld);
            }
            for (; m < m1; m += 16) s0 += *reinterpret_cast<const f32x4 *>(base + m * ld);
        }
    } else {
        for (size_t m = m0 + rg; m < m1; m += 16)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < K) s0[e] += dy[m * ld + c + e];
    }
    const f32x4 s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][cg * 4 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < K) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        part[(size_t)blockIdx.y * K + blockIdx.x * 64 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float *part, int K, int parts, float *out, int accumulate)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;          // 16 row groups x 64 channels
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < K)
        for (int p = r; p < parts; p += 16) s += part[(size_t)p * K + c];
    red[r][lane] = s;
    __syncthreads();
    if (r == 0 && c < K) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][lane];
        out[c] = accumulate ? out[c] + v : v;
    }
}

// final = false: only the first stage (the caller sums the parts later: stem_bias_grad_final / _final_multi)
int colsum(const float *dy, int ld, size_t npix, int K, float *scratch, float *db, int accumulate, hipStream_t st, bool final = true)
{
    const int parts = colsum_parts(npix, K);
    const bool vec = (K % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)dy) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((colsum_partial_kernel<true>), dim3(cdiv(K, 64), parts), dim3(256), 0, st, dy, ld, npix, K, parts, scratch);
    else
        hipLaunchKernelGGL((colsum_partial_kernel<false>), dim3(cdiv(K, 64), parts), dim3(256), 0, st, dy, ld, npix, K, parts, scratch);
    STEM_LAUNCH_CHECK("colsum_partial");
    if (!final) return 0;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(K, 64)), dim3(1024), 0, st, scratch, K, parts, db, accumulate);
    STEM_LAUNCH_CHECK("colsum_final");
    return 0;
}

}   // namespace

// bias gradient on its own (for the weight-gradient kernel of wgrad_f16x3.hip, which does not touch the fp32 dy)
STEM_EXPORT size_t stem_bias_grad_scratch_elems(long npix, int K) { return (size_t)colsum_parts((size_t)npix, K) * K; }

/* second stage only: db (+)= sum over `parts` rows of part[parts][K] (first stage done by stem_conv2d_wgrad_f16x3) */
STEM_EXPORT int stem_bias_grad_final(const float *part, int K, int parts, float *db, int accumulate, void *stream)
{
    STEM_CHECK_ARG(part && db && K >= 1 && parts >= 1, "stem_bias_grad_final: bad arguments");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(K, 64)), dim3(1024), 0, (hipStream_t)stream, part, K, parts, db, accumulate);
    STEM_LAUNCH_CHECK("stem_bias_grad_final");
    return 0;
}

// every pending bias gradient of a module group with ONE launch: blockIdx.y = descriptor (the second stage used to be one
// 5-us launch per layer, 13 per P-frame step, each with its dispatch gap on the weight-gradient stream)
namespace {
constexpr int MAXBIASFIN = 24;
struct BiasFinTable {
    stem_bias_final_desc d[MAXBIASFIN];
};
__global__ __launch_bounds__(1024) void colsum_final_multi_kernel(const BiasFinTable tb)
{
    __shared__ float red[16][64];
    const stem_bias_final_desc d = tb.d[blockIdx.y];
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= d.K) return;                 // workgroup-uniform: this tensor has fewer 64-column blocks than the widest
    float s = 0.f;
    if (c < d.K)
        for (int p = r; p < d.parts; p += 16) s += d.part[(size_t)p * d.K + c];
    red[r][lane] = s;
    __syncthreads();
    if (r == 0 && c < d.K) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][lane];           // the summation order of colsum_final_kernel: same bits
        d.db[c] = d.accumulate ? d.db[c] + v : v;
    }
}
}   // namespace

STEM_EXPORT int stem_bias_grad_final_multi(const stem_bias_final_desc *descs, int n, void *stream)
{
    STEM_CHECK_ARG(descs && n >= 1 && n <= MAXBIASFIN, "stem_bias_grad_final_multi: 1..%d descriptors per call, got %d", MAXBIASFIN, n);
    BiasFinTable tb;
    memset(&tb, 0, sizeof(tb));
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        STEM_CHECK_ARG(descs[i].part && descs[i].db && descs[i].K >= 1 && descs[i].parts >= 1, "stem_bias_grad_final_multi: descriptor %d", i);
        tb.d[i] = descs[i];
        if (descs[i].K > kmax) kmax = descs[i].K;
    }
    hipLaunchKernelGGL(colsum_final_multi_kernel, dim3(cdiv(kmax, 64), n), dim3(1024), 0, (hipStream_t)stream, tb);
    STEM_LAUNCH_CHECK("stem_bias_grad_final_multi");
    return 0;
}

STEM_EXPORT int stem_bias_grad(const float *dy, int lddy, long npix, int K, float *scratch, float *db, int accumulate, void *stream)
{
    STEM_CHECK_ARG(dy && scratch && db && npix >= 1 && K >= 1 && lddy >= K, "stem_bias_grad: bad arguments");
    return colsum(dy, lddy, (size_t)npix, K, scratch, db, accumulate, (hipStream_t)stream);
}

STEM_EXPORT size_t stem_wgrad_workspace_elems(int splits, int C, int K, int R, int S, int npix)
{
    // slabs | bias-gradient partial sums | per-pixel gather table (offset, tap mask)
    return (size_t)splits * R * S * K * C + (size_t)CS_MAX_PARTS * K + (size_t)2 * npix + 4;
}

STEM_EXPORT int stem_wgrad_splits(int B, int Ho, int Wo, int C, int K, int R, int S)
{
    int cfg, splits;
    plan(K, C, R * S, cdiv(B * Ho * Wo, KP), &cfg, &splits);
    return splits;
}

// number of per-part column sums a STEM_WGRAD_DEFER_DB call leaves at dwp + splits * R * S * K * C (each K floats): the splits of the
// weight-gradient kernel when it sums dy on its way (convolutions on the vector path), else the column-sum pass's own parts
STEM_EXPORT int stem_wgrad_bias_parts(const float *x, int ldx, const float *dy, int lddy, long npix_dy, int C, int K, int splits, int flags,
                                      int deconv)
{
    const bool fused_db = !deconv && splits <= CS_MAX_PARTS && wgrad_vec_ok(dy, lddy, K, x, ldx, C) && !(flags & STEM_WGRAD_SQUARE_G);
    return fused_db ? splits : colsum_parts((size_t)npix_dy, K);
}

STEM_EXPORT int stem_conv2d_wgrad(const float *x, int ldx, const float *dy, int lddy, float *dwp, float *db,
                                  int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                  int splits, int flags, void *stream)
{
    STEM_CHECK_ARG(x && dy && dwp, "stem_conv2d_wgrad: null pointer");
    STEM_CHECK_ARG(splits >= 1, "stem_conv2d_wgrad: splits must be >= 1");
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    float *scratch = dwp + (size_t)splits * R * S * K * C;
    const int acc_db = (flags & STEM_WGRAD_ACCUMULATE_DB) ? 1 : 0;
    // The bias gradient is the column sum of dY, the operand the weight-gradient kernel stages on its loop grid: on the vector
    // path its tap-0 workgroups sum the rows they load (one partial per split) and only the tiny second stage is launched.
    const bool fused_db = db && splits <= CS_MAX_PARTS && wgrad_vec_ok(dy, lddy, K, x, ldx, C) && !(flags & STEM_WGRAD_SQUARE_G);
    const bool defer = (flags & STEM_WGRAD_DEFER_DB) != 0;      // the parts stay in the scratch behind the slabs: stem_wgrad_bias_parts of them
    if (db && !fused_db && colsum(dy, lddy, (size_t)B * Ho * Wo, K, scratch, db, acc_db, st, !defer)) return -2;
    // P = dY on the output grid (K channels), G = x gathered at oy*stride - pad + r  ->  [t][K][C]
    int *ptab = reinterpret_cast<int *>(scratch + (size_t)CS_MAX_PARTS * K);
    if (int rc = run(dy, lddy, K, x, ldx, C, dwp, ptab, B, Ho, Wo, H, W, R, S, stride, pad, splits, flags, st, fused_db ? scratch : nullptr))
        return rc;
    if (fused_db && !defer) {
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(K, 64)), dim3(1024), 0, st, scratch, K, splits, db, acc_db);
        STEM_LAUNCH_CHECK("colsum_final");
    }
    return 0;
}

STEM_EXPORT int stem_deconv2d_wgrad(const float *x, int ldx, const float *dy, int lddy, float *dwp, float *db,
                                    int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                                    int splits, int flags, void *stream)
{
    STEM_CHECK_ARG(x && dy && dwp, "stem_deconv2d_wgrad: null pointer");
    STEM_CHECK_ARG(splits >= 1, "stem_deconv2d_wgrad: splits must be >= 1");
    const int Ho = (H - 1) * stride - 2 * pad + R + opad, Wo = (W - 1) * stride - 2 * pad + S + opad;
    hipStream_t st = (hipStream_t)stream;
    if (db && colsum(dy, lddy, (size_t)B * Ho * Wo, K, dwp + (size_t)splits * R * S * K * C, db, (flags & STEM_WGRAD_ACCUMULATE_DB) ? 1 : 0, st,
                     !(flags & STEM_WGRAD_DEFER_DB)))
        return -2;
    // P = x on the input grid (C channels), G = dY gathered at iy*stride - pad + r  ->  [t][C][K]
    int *ptab = reinterpret_cast<int *>(dwp + (size_t)splits * R * S * K * C + (size_t)CS_MAX_PARTS * K);
    return run(x, ldx, C, dy, lddy, K, dwp, ptab, B, H, W, Ho, Wo, R, S, stride, pad, splits, flags, st);
}
