// Implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chain, 157 TFLOP/s peak).
//
// One kernel covers every dense contraction on the STEM hot path:
//   out[b, qy*osy+ooy, qx*osx+oox, n] = epi( bias[n] + sum_t sum_c  X[b, qy*isy+dy_t, qx*isx+dx_t, c] * Wp[wt_t][n][c] )
// with a per-phase tap table (dy,dx,wt).  Direct convolutions use one phase with isy=stride;
// transposed convolutions and the input-gradients of strided convolutions are decomposed into
// stride*stride sub-pixel phases (blockIdx.z) so no zero-inserted tensor ever exists.
//
// Tiling: 256 threads = 4 wavefronts of 64 lanes; block tile BM pixels x BN channels, K chunk
// of 32 (one tap x 32 input channels, or 8 taps x 4 channels for the 3-channel first layer).
// Operands are staged global -> VGPR -> LDS ([row][32+4] fp32, rows 144 B so ds_write_b128 /
// ds_read_b128 are aligned and bank-conflict free), double buffered, one barrier per chunk.
// Each lane reads 4 consecutive k of its row with one ds_read_b128; lane half h supplies
// k = 4h+s to MFMA step s (any bijection of k is valid as long as A and B agree), so one
// LDS read per operand tile feeds four MFMAs.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <vector>

#include "stem_common.h"

namespace {

constexpr int KC = 32;        // K chunk
constexpr int PITCH = 36;     // LDS row pitch in floats (32 + 4 pad)
constexpr int MAXTAP = 28;

enum { EPI_BIAS = 0, EPI_LRELU = 1, EPI_DACT = 2, EPI_GDN = 3, EPI_IGDN = 4, EPI_NORM = 5 /* GDN denominator only */ };

struct TapPhase {
    int ooy, oox, qh, qw, ntaps;
    signed char dy[MAXTAP];
    signed char dx[MAXTAP];
    unsigned char wt[MAXTAP];
};

struct IgemmArgs {
    const float *x, *w, *bias, *z, *beta;
    float *y;
    int ldx, ldw, ldy, ldz;
    int B, H, W, C, N, OH, OW;
    int isy, isx, osy, osx;
    int nphase, epi, asquare, breparam;
    float slope, beta_bound;
    // split-K: blockIdx.z = phase * nsplit + split; each split reduces chunks [split*cps, (split+1)*cps) and writes its raw
    // fp32 partial tile to ws[split][out pixel][n]; the LAST workgroup to arrive at a tile (arrival counter) adds the
    // partials in split order 0..nsplit-1 -- the result does not depend on who arrives last -- and applies the epilogue.
    float *ws;
    int *cnt;                // one arrival counter per output tile (zero before and after every launch)
    int nsplit, cps;
    long slab;
    int xbytes, wbytes;      // extents of the x / w views for the range-checked buffer loads
    const float *gamma;      // fused conv + GDN/IGDN (FUSE kernels): stored gamma [N][N], beta in `beta`
    int fuse, gmbytes;       // 0 none, 1 GDN, 2 IGDN
    int exper;               // tuning experiments (STEM_IGEMM_EXPER), 0 in production
    int ident;               // output pixel index == m (stride-1, single phase): no div/mod in the epilogue
    TapPhase ph[4];
};

template <int BM, int BN, int WM, int WN, bool VEC, bool C4, bool GDNOP, bool FUSE = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_kernel(const IgemmArgs a)
{
    constexpr int NT = (BM / WM) * (BN / WN) * 64;       // 256 (4 wavefronts) or 512 (8 wavefronts)
    constexpr int RPP = NT / 8;                          // tile rows staged per pass (8 float4 per 32-float row)
    constexpr int TM = WM / 32, TN = WN / 32, AR = BM / RPP, BR = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    constexpr int WCOLS = BN / WN;
    static_assert(NT == 256 || NT == 512, "4 or 8 waves per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [2][BM][PITCH]
    float *Bs = smem + 2 * BM * PITCH;      // [2][BN][PITCH]
    int *tapi = reinterpret_cast<int *>(smem + 2 * (BM + BN) * PITCH);   // [32][4]: dy, dx, wt, valid

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zphase = blockIdx.z / a.nsplit, zsplit = blockIdx.z - zphase * a.nsplit;
    const TapPhase &ph = a.ph[zphase];
    const int Mtot = a.B * ph.qh * ph.qw;
    // XCD-aware pixel-tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so
    // tile t -> (t % 8) gets a contiguous eighth of the pixel tiles: neighbouring tiles share halo rows / the same
    // weights in ONE L2 instead of fetching them once per XCD.  Pure speed heuristic; any placement is correct.
    int tile_m = blockIdx.x;
    {
        const int nb = gridDim.x, qq = nb >> 3, rr = nb & 7, xcd = tile_m & 7, idx = tile_m >> 3;
        if (nb >= 16) tile_m = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int bm0 = tile_m * BM, bn0 = blockIdx.y * BN;
    if (bm0 >= Mtot) return;

    const int qhw = ph.qh * ph.qw;
    if (tid < 32) {
        const bool v = tid < ph.ntaps;
        const int dy = v ? ph.dy[tid] : 0, dx = v ? ph.dx[tid] : 0, wt = v ? ph.wt[tid] : 0;
        tapi[tid * 4 + 0] = VEC ? (dy * a.W + dx) * a.ldx * 4 : dy;      // VEC: byte offset of the tap inside x
        tapi[tid * 4 + 1] = VEC ? wt * a.N * a.ldw * 4 : dx;             // VEC: byte offset of the tap's weight slab
        tapi[tid * 4 + 2] = wt;
        tapi[tid * 4 + 3] = v ? 1 : 0;
    }

    // ---- staging assignment: thread -> (row srow+32j, float4 column c4) -------------------------
    const int srow = tid >> 3, c4 = tid & 7;
    int p_by[AR], p_bx[AR], p_base[AR];
    unsigned p_mask[AR];          // bit t: tap t of this row reads inside the image (computed once, not per chunk)
    bool p_ok[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = bm0 + srow + RPP * j;
        p_ok[j] = m < Mtot;
        const int mm = p_ok[j] ? m : 0;
        const int b = mm / qhw, rem = mm - b * qhw;
        const int qy = rem / ph.qw, qx = rem - qy * ph.qw;
        p_by[j] = qy * a.isy;
        p_bx[j] = qx * a.isx;
        p_base[j] = VEC ? ((b * a.H + p_by[j]) * a.W + p_bx[j]) * a.ldx * 4 : b * a.H * a.W;
        unsigned mk = 0;
        for (int t = 0; t < ph.ntaps; ++t) {
            const int iy = p_by[j] + ph.dy[t], ix = p_bx[j] + ph.dx[t];
            if (p_ok[j] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) mk |= 1u << t;
        }
        p_mask[j] = mk;
    }
    int n_off[BR];                // VEC: byte offset of weight row n, or an out-of-range offset when n >= N
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = bn0 + srow + RPP * j;
        n_off[j] = n < a.N ? n * a.ldw * 4 : 0x40000000;      // + tap/k offsets (< 2^30, checked on the host) stays out of range
    }
    const int nkc = C4 ? 1 : (a.C + KC - 1) / KC;
    const int nchunks_all = C4 ? (ph.ntaps + 7) / 8 : ph.ntaps * nkc;
    const int q_begin = zsplit * a.cps;
    const int q_end = a.nsplit > 1 ? (q_begin + a.cps < nchunks_all ? q_begin + a.cps : nchunks_all) : nchunks_all;
    const int q_last_u = q_end - 1;
    __syncthreads();

    // Hardware range-checked buffer loads: an offset past the end of the view returns 0, so padding pixels,
    // tile tails and K tails need no branches; the loop body is one basic block the scheduler can interleave
    // with the MFMAs.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, a.wbytes, 0x00020000);

    // two register sets: loads run TWO chunks ahead of the MFMAs that consume them (L2/MALL latency under load
    // exceeds one chunk of matrix work); named sets + a 2x unrolled loop keep every index static.
    f32x4 raA[AR], rbA[BR], raB[AR], rbB[BR];

    auto gload = [&](int q, f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        int t, k0;
        if (C4) {
            t = q * 8 + c4;
            k0 = 0;
        } else {
            // channel chunk outer, taps inner: the 25 taps of one 32-channel slab re-touch the same input lines
            // back to back (L1/L2 hits) instead of once per full sweep of all channels (L2-capacity misses)
            const int kc = q / ph.ntaps;
            t = q - kc * ph.ntaps;
            k0 = kc * KC + 4 * c4;
        }
        const int t0 = tapi[t * 4 + 0], t1 = tapi[t * 4 + 1], wt = tapi[t * 4 + 2];
        const bool tv = tapi[t * 4 + 3] != 0;
        if (VEC) {
            const int kmask = -(int)(k0 < a.C), tvmask = -(int)tv;
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const int mk = -(int)((p_mask[j] >> t) & 1u) & kmask;
                const int off = ((p_base[j] + t0 + k0 * 4) & mk) | (0x7FFFFF00 & ~mk);
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int mk = C4 ? -1 : (tvmask & kmask);
                const int o = C4 ? n_off[j] + (q * KC + 4 * c4) * 4 : n_off[j] + t1 + k0 * 4;
                const int off = (o & mk) | (0x7FFFFF00 & ~mk);
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
            }
            return;
        }
        const int dy = t0, dx = t1;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int iy = p_by[j] + dy, ix = p_bx[j] + dx;
            const bool ok = p_ok[j] && tv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float *src = a.x + (size_t)(p_base[j] + iy * a.W + ix) * a.ldx + k0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + e < a.C) v[e] = src[e];
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int n = bn0 + srow + RPP * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < a.N && tv) {
                const float *src = a.w + ((size_t)wt * a.N + n) * a.ldw + k0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + e < a.C) v[e] = src[e];
            }
            rb[j] = v;
        }
    };
    // Loaded values are first touched here, AFTER the MFMA block of the current chunk, so the global loads of
    // chunk q+1 stay in flight under the matrix work of chunk q (a use inside gload would force vmcnt(0) there).
    auto sstore = [&](int buf, f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            f32x4 v = ra[j];
            if (GDNOP) v = v * v;                                   // GDN: x^2 (gdn.py:58)
            *reinterpret_cast<f32x4 *>(&As[(buf * BM + srow + RPP * j) * PITCH + 4 * c4]) = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            f32x4 v = rb[j];
            if (GDNOP) {   // NonNegativeParametrizer on gamma (parametrizers.py:42-45); padded lanes (0) map to 0
                const float bound = 3.814697265625e-06f, ped = 1.4551915228366852e-11f;   // 2^-18, 2^-36
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = fmaxf(v[e], bound);
                    v[e] = g * g - ped;
                }
            }
            *reinterpret_cast<f32x4 *>(&Bs[(buf * BN + srow + RPP * j) * PITCH + 4 * c4]) = v;
        }
    };

    // ---- wave tile -------------------------------------------------------------------------------
    const int wm0 = (wave / WCOLS) * WM, wn0 = (wave % WCOLS) * WN;
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One pipeline step = the MFMAs of the chunk in LDS buffer `cur`, with the staging of later chunks woven into
    // the same basic block so that the scheduler can slot it between MFMAs (each MFMA occupies the matrix pipe for
    // 64 cycles but the issue port only briefly): after k-group 0/1 the register set (chunk +1) is written to the
    // other LDS buffer, after k-group 2/3 the same set is refilled with chunk `qn` (two chunks ahead).
    auto step = [&](int cur, f32x4 (&ra)[AR], f32x4 (&rb)[BR], int qn) {
        const float *Ab = As + (cur * BM + wm0 + lr) * PITCH + 4 * lh;
        const float *Bb = Bs + (cur * BN + wn0 + lr) * PITCH + 4 * lh;
        int t, k0;
        if (C4) {
            t = qn * 8 + c4;
            k0 = 0;
        } else {
            const int kc = qn / ph.ntaps;
            t = qn - kc * ph.ntaps;
            k0 = kc * KC + 4 * c4;
        }
        const int t0 = tapi[t * 4 + 0], t1 = tapi[t * 4 + 1];
        const int tvmask = -tapi[t * 4 + 3];                 // 0 / -1
        const int kmask = -(int)(k0 < a.C);
#pragma unroll
        for (int k8 = 0; k8 < KC / 8; ++k8) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            if (k8 == 0) {
#pragma unroll
                for (int j = 0; j < AR; ++j) {
                    f32x4 v = ra[j];
                    if (GDNOP) v = v * v;
                    *reinterpret_cast<f32x4 *>(&As[((cur ^ 1) * BM + srow + RPP * j) * PITCH + 4 * c4]) = v;
                }
            } else if (k8 == 1) {
#pragma unroll
                for (int j = 0; j < BR; ++j) {
                    f32x4 v = rb[j];
                    if (GDNOP) {
                        const float bound = 3.814697265625e-06f, ped = 1.4551915228366852e-11f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = fmaxf(v[e], bound);
                            v[e] = g * g - ped;
                        }
                    }
                    *reinterpret_cast<f32x4 *>(&Bs[((cur ^ 1) * BN + srow + RPP * j) * PITCH + 4 * c4]) = v;
                }
            } else if (k8 == 2) {
#pragma unroll
                for (int j = 0; j < AR; ++j) {
                    // mask arithmetic instead of ?: -- a select feeding the load gets lowered to two loads under
                    // divergent branches, which also breaks the compiler's vmcnt accounting (it then waits for ALL loads)
                    const int mk = -(int)((p_mask[j] >> t) & 1u) & kmask;
                    const int off = ((p_base[j] + t0 + k0 * 4) & mk) | (0x7FFFFF00 & ~mk);
                    ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                }
            } else {
#pragma unroll
                for (int j = 0; j < BR; ++j) {
                    const int mk = C4 ? -1 : (tvmask & kmask);
                    const int o = C4 ? n_off[j] + (qn * KC + 4 * c4) * 4 : n_off[j] + t1 + k0 * 4;
                    const int off = (o & mk) | (0x7FFFFF00 & ~mk);
                    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
                }
            }
        }
    };
    auto compute = [&](int cur) {
        const float *Ab = As + (cur * BM + wm0 + lr) * PITCH + 4 * lh;
        const float *Bb = Bs + (cur * BN + wn0 + lr) * PITCH + 4 * lh;
#pragma unroll
        for (int k8 = 0; k8 < KC / 8; ++k8) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
    };

    // ---- lean staging for the common case (16-byte operands, channel count a multiple of the K chunk) --------------------
    // The generic step() spends ~2.7 VALU instructions per MFMA on addresses and masks (two integer divisions per pair of
    // chunks among them); next to a 64-cycle MFMA each costs ~4 issue cycles on the same port, and the 64x64 tile has
    // only 16 MFMAs per chunk to hide them behind (SQ counters: 60-75 % matrix-pipe utilisation vs 83 % for the 128x192
    // tile).  Here the chunk coordinates (tap, channel slab) of the prefetch target are advanced incrementally as wave-
    // uniform values, the slab part of both addresses (and the weights' tap offset) goes into the buffer instruction's scalar
    // offset, the per-lane part (pixel base + 16 B column) is loop invariant, and an activation row costs one add, one
    // bit-field extract (tap validity) and one bit-field insert; weight rows cost nothing.
    int pb16[AR], nb16[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) pb16[j] = p_base[j] + 16 * c4;
#pragma unroll
    for (int j = 0; j < BR; ++j) nb16[j] = n_off[j] + 16 * c4;
    int pf_q = 0, pf_t = 0, pf_kc = 0;          // prefetch target of the NEXT step: chunk index, tap, channel slab
    auto step_fast = [&](int cur, f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        const float *Ab = As + (cur * BM + wm0 + lr) * PITCH + 4 * lh;
        const float *Bb = Bs + (cur * BN + wn0 + lr) * PITCH + 4 * lh;
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc);
        // tap offsets of the activations can be negative (taps above / left of the pixel) and the scalar offset of a buffer
        // instruction is unsigned: they are added per lane (one VALU add with a scalar operand); the channel-slab part and
        // the (non-negative) weight-slab offset go through the scalar offset
        const int tA = __builtin_amdgcn_readfirstlane(tapi[t * 4 + 0]), sA = kc * (KC * 4);
        const int sB = __builtin_amdgcn_readfirstlane(tapi[t * 4 + 1]) + kc * (KC * 4);
#pragma unroll
        for (int k8 = 0; k8 < KC / 8; ++k8) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * PITCH + k8 * 8);
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][ss], bf[j][ss], acc[i][j], 0, 0, 0);
            if (k8 == 0) {
#pragma unroll
                for (int j = 0; j < AR; ++j)
                    *reinterpret_cast<f32x4 *>(&As[((cur ^ 1) * BM + srow + RPP * j) * PITCH + 4 * c4]) = ra[j];
            } else if (k8 == 1) {
#pragma unroll
                for (int j = 0; j < BR; ++j)
                    *reinterpret_cast<f32x4 *>(&Bs[((cur ^ 1) * BN + srow + RPP * j) * PITCH + 4 * c4]) = rb[j];
            } else if (k8 == 2) {
#pragma unroll
                for (int j = 0; j < AR; ++j) {
                    const int mk = __builtin_amdgcn_sbfe((int)p_mask[j], (unsigned)t, 1u);          // 0 / -1: tap t reads inside the image
                    const int off = ((pb16[j] + tA) & mk) | (0x7FFFFF00 & ~mk);
                    ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, sA, 0));
                }
            } else {
#pragma unroll
                for (int j = 0; j < BR; ++j)
                    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, nb16[j], sB, 0));
            }
        }
        // advance the prefetch target (it stays on the last chunk once reached: loaded again, never used)
        if (pf_q < q_last_u) {
            ++pf_q;
            if (++pf_t == ph.ntaps) {
                pf_t = 0;
                ++pf_kc;
            }
        }
    };

    // chunk q lives in LDS buffer (q - q_begin) & 1; set A holds chunk q+1, set B chunk q+2 (odd/even alternate).
    // Prefetch targets past the last chunk are clamped to it (loaded, never used): no branches in the loop body.
    const int q_last = q_end - 1;
    auto clampq = [&](int q) { return q < q_last ? q : q_last; };
    if (q_begin < q_end) {
        gload(q_begin, raA, rbA);
        sstore(0, raA, rbA);
        gload(clampq(q_begin + 1), raA, rbA);
        gload(clampq(q_begin + 2), raB, rbB);
    }
    __syncthreads();
    if (VEC && !C4 && !GDNOP && (a.C % KC) == 0 && q_begin < q_end) {
        // lean staging (see step_fast): the first prefetch target is chunk min(q_begin + 3, q_last)
        pf_q = q_begin + 3 < q_last ? q_begin + 3 : q_last;
        pf_kc = pf_q / ph.ntaps;
        pf_t = pf_q - pf_kc * ph.ntaps;
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step_fast(0, raA, rbA);
            __syncthreads();
            step_fast(1, raB, rbB);
            __syncthreads();
        }
        if (q < q_end) {
            step_fast(0, raA, rbA);
            __syncthreads();
        }
    } else if (VEC) {
        // break-free two-step body (the odd tail runs after the loop): with a mid-loop exit the compiler merged the wait
        // counters of both exits into vmcnt(0) at the top of the body and copied the accumulators between the steps
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step(0, raA, rbA, clampq(q + 3));                  // chunk q; stage q+1 -> buffer 1; prefetch q+3
            __syncthreads();
            step(1, raB, rbB, clampq(q + 4));                  // chunk q+1; stage q+2 -> buffer 0; prefetch q+4
            __syncthreads();
        }
        if (q < q_end) {
            step(0, raA, rbA, clampq(q + 3));
            __syncthreads();
        }
    } else {
        for (int q = q_begin; q < q_end; q += 2) {
            compute(0);
            if (q + 1 < q_end) sstore(1, raA, rbA);
            if (q + 3 < q_end) gload(q + 3, raA, rbA);
            __syncthreads();
            if (q + 1 >= q_end) break;
            compute(1);
            if (q + 2 < q_end) sstore(0, raB, rbB);
            if (q + 4 < q_end) gload(q + 4, raB, rbB);
            __syncthreads();
        }
    }

    if (a.nsplit > 1) {   // raw partial sums to the workspace; the last arriver of this tile reduces them
        float *wsp = a.ws + (size_t)zsplit * a.slab;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = bn0 + wn0 + j * 32 + lr;
            if (n >= a.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = bm0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= Mtot) continue;
                    size_t opix = m;
                    if (!a.ident) {
                        const int b = m / qhw, rem = m - b * qhw;
                        const int qy = rem / ph.qw, qx = rem - qy * ph.qw;
                        opix = (size_t)(b * a.OH + qy * a.osy + ph.ooy) * a.OW + qx * a.osx + ph.oox;
                    }
                    // agent-scope (sc1) store: written through this XCD's L2 to the device-coherent level
                    __hip_atomic_store(&wsp[opix * a.N + n], acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        // Arrival protocol.  The splits of one tile run on different XCDs, whose L2s are not coherent with each other.  A
        // device-scope fence (__threadfence) would write back / invalidate the WHOLE L2 per wave -- measured: +25 % on the
        // bench step, it also evicts the concurrent weight-gradient kernel's working set.  Instead the partials themselves
        // are moved with agent-scope accesses (sc1: write-through stores, L2-bypassing loads), every thread waits for its
        // own stores to be acknowledged (vmcnt 0), the workgroup meets at a barrier and ONE thread takes a ticket.  The
        // workgroup that draws the last ticket knows all nsplit partial tiles are complete; it re-zeroes the counter for
        // the next launch.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *flag = tapi;                     // LDS scratch, free after the main loop
        if (tid == 0) {
            int *c = a.cnt + ((size_t)zphase * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            flag[0] = splitk_last_arriver(c, a.nsplit);
        }
        __syncthreads();
        if (!flag[0]) return;
        // y[pix][n] = epi(bias[n] + sum_s ws[s][pix][n]): rows of the tile, 4 channels per thread, fixed split order.  The
        // partials are fetched with sc1 (agent-scope) buffer loads, 4 splits in flight per thread.
        constexpr int NQ = BN / 4;
        constexpr int SC1 = 16;               // cache-policy bit 4 of the buffer instructions = sc1 on gfx94x/gfx950
        const bool v4 = (a.N % 4 == 0) && (a.ldy % 4 == 0) && (a.epi != EPI_DACT || a.ldz % 4 == 0);
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)(a.slab * a.nsplit * 4), 0x00020000);
        for (int e = tid; e < BM * NQ; e += NT) {
            const int row = e / NQ, n = bn0 + (e - row * NQ) * 4;
            const int m = bm0 + row;
            if (m >= Mtot || n >= a.N) continue;
            size_t opix = m;
            if (!a.ident) {
                const int b = m / qhw, rem = m - b * qhw;
                const int qy = rem / ph.qw, qx = rem - qy * ph.qw;
                opix = (size_t)(b * a.OH + qy * a.osy + ph.ooy) * a.OW + qx * a.osx + ph.oox;
            }
            const int nn = a.N - n < 4 ? a.N - n : 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < nn; ++q) v[q] = a.bias ? a.bias[n + q] : 0.f;
            const int off0 = (int)((opix * a.N + n) * 4), sstep = (int)(a.slab * 4);
            if (v4) {
                int sp = 0;
                for (; sp + 4 <= a.nsplit; sp += 4) {
                    f32x4 t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        t[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, off0 + (sp + u) * sstep, 0, SC1));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[0] += t[u][0]; v[1] += t[u][1]; v[2] += t[u][2]; v[3] += t[u][3];
                    }
                }
                for (; sp < a.nsplit; ++sp) {
                    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, off0 + sp * sstep, 0, SC1));
                    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
                }
            } else {
                for (int sp = 0; sp < a.nsplit; ++sp)
                    for (int q = 0; q < nn; ++q)
                        v[q] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rws, off0 + sp * sstep + q * 4, 0, SC1));
            }
            for (int q = 0; q < nn; ++q) {
                if (a.epi == EPI_LRELU) {
                    v[q] = v[q] > 0.f ? v[q] : v[q] * a.slope;
                } else if (a.epi == EPI_DACT) {
                    v[q] = a.z[opix * a.ldz + n + q] > 0.f ? v[q] : v[q] * a.slope;
                }
            }
            if (v4) {
                const f32x4 o = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4 *>(a.y + opix * a.ldy + n) = o;
            } else {
                for (int q = 0; q < nn; ++q) a.y[opix * a.ldy + n + q] = v[q];
            }
        }
        return;
    }

    // ---- epilogue: lane holds column n = ..+lr, rows (r&3)+8*(r>>2)+4*lh of each 32x32 tile -------
    // ---- fused GDN / IGDN (gdn.py:52-67) on the tile still held in registers: the workgroup owns ALL N channels
    // of its pixels (BN >= N), so norm[px][i] = beta'[i] + sum_j gamma'[i][j] * v[px][j]^2 is a second small GEMM
    // (K = N) whose A operand is the squared conv output parked in LDS and whose B operand (gamma, reparametrised
    // on the fly) is streamed 32 columns at a time.  The conv output never makes the HBM round trip.
    if (FUSE) {
        constexpr int XP = BN + 4;                     // row pitch of the x^2 tile (conflict-free ds_read_b128)
        float *X2 = smem;                              // [BM][XP]
        float *Gs = smem + BM * XP;                    // [BN][PITCH]
        __syncthreads();                               // every wave is done with the main-loop buffers
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = bn0 + wn0 + j * 32 + lr;
            const float bias = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] + bias;
                    acc[i][j][r] = v;                  // keep v for the final multiply
                    X2[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * XP + wn0 + j * 32 + lr] = v * v;
                }
        }
        f32x16 nrm[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[i][j][r] = 0.f;
        const __amdgpu_buffer_rsrc_t rgm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gamma), 0, a.gmbytes, 0x00020000);
        const int nkg = (a.N + KC - 1) / KC;
        // gamma chunk kc + 1 is fetched into registers while the MFMAs of chunk kc run (the first layer has only 4 conv
        // chunks per tile: this second contraction IS its main loop, and an exposed L2 latency per chunk cost 20 % of it)
        auto gload_gamma = [&](int kc, f32x4 (&g4)[BR]) {
            const int kcol = kc * KC + 4 * c4;
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int n = srow + RPP * j;
                const int mk = -(int)(n < a.N && kcol < a.N);
                const int off = (((n * a.N + kcol) * 4) & mk) | (0x7FFFFF00 & ~mk);
                g4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgm, off, 0, 0));
            }
        };
        f32x4 g4[BR], g4n[BR];
        gload_gamma(0, g4);
        for (int kc = 0; kc < nkg; ++kc) {
            __syncthreads();                           // previous chunk's Gs (and, first time, X2) settled
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                f32x4 v = g4[j];
                const float bound = 3.814697265625e-06f, ped = 1.4551915228366852e-11f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = fmaxf(v[e], bound);
                    v[e] = g * g - ped;
                }
                *reinterpret_cast<f32x4 *>(&Gs[(srow + RPP * j) * PITCH + 4 * c4]) = v;
            }
            __syncthreads();
            gload_gamma(kc + 1 < nkg ? kc + 1 : kc, g4n);      // in flight during the MFMAs below (clamped: last one unused)
            const float *Ab = X2 + (wm0 + lr) * XP + kc * KC + 4 * lh;
            const float *Bb = Gs + (wn0 + lr) * PITCH + 4 * lh;
#pragma unroll
            for (int k8 = 0; k8 < KC / 8; ++k8) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * XP + k8 * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * PITCH + k8 * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            nrm[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], nrm[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < BR; ++j) g4[j] = g4n[j];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = bn0 + wn0 + j * 32 + lr;
            const bool nok = n < a.N;
            float bt = 0.f;
            if (nok) {
                const float bb = fmaxf(a.beta[n], a.beta_bound);
                bt = bb * bb - 1.4551915228366852e-11f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = bm0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= Mtot || !nok) continue;
                    size_t opix = m;
                    if (!a.ident) {
                        const int b = m / qhw, rem = m - b * qhw;
                        const int qy = rem / ph.qw, qx = rem - qy * ph.qw;
                        opix = (size_t)(b * a.OH + qy * a.osy + ph.ooy) * a.OW + qx * a.osx + ph.oox;
                    }
                    const float nv = nrm[i][j][r] + bt;
                    const float o = acc[i][j][r] * (a.fuse == 2 ? __builtin_amdgcn_sqrtf(nv) : __builtin_amdgcn_rsqf(nv));
                    a.y[opix * a.ldy + n] = o;
                }
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn0 + wn0 + j * 32 + lr;
        const bool nok = n < a.N;
        float bias = (a.bias && nok) ? a.bias[n] : 0.f;
        if ((a.epi == EPI_GDN || a.epi == EPI_IGDN || a.epi == EPI_NORM) && nok) {   // beta' = max(beta, bound)^2 - 2^-36
            const float bb = fmaxf(a.beta[n], a.beta_bound);
            bias = bb * bb - 1.4551915228366852e-11f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= Mtot || !nok) continue;
                size_t opix = m;
                if (!a.ident) {
                    const int b = m / qhw, rem = m - b * qhw;
                    const int qy = rem / ph.qw, qx = rem - qy * ph.qw;
                    opix = (size_t)(b * a.OH + qy * a.osy + ph.ooy) * a.OW + qx * a.osx + ph.oox;
                }
                float v = acc[i][j][r] + bias;
                if (a.epi == EPI_LRELU) {
                    v = v > 0.f ? v : v * a.slope;
                } else if (a.epi == EPI_DACT) {
                    const float zz = a.z[opix * a.ldz + n];
                    v = zz > 0.f ? v : v * a.slope;
                } else if (a.epi == EPI_GDN) {
                    v = a.z[opix * a.ldz + n] * __builtin_amdgcn_rsqf(v);      // v_rsq_f32, 1 ulp (torch.rsqrt, gdn.py:63)
                } else if (a.epi == EPI_IGDN) {
                    v = a.z[opix * a.ldz + n] * __builtin_amdgcn_sqrtf(v);     // v_sqrt_f32, 1 ulp (gdn.py:61)
                }
                a.y[opix * a.ldy + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
void build_direct(IgemmArgs &g, int R, int S, int stride, int pad, int OH, int OW)
{
    g.nphase = 1;
    g.isy = g.isx = stride;
    g.osy = g.osx = 1;
    TapPhase &p = g.ph[0];
    p.ooy = p.oox = 0;
    p.qh = OH;
    p.qw = OW;
    p.ntaps = R * S;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            p.dy[r * S + s] = (signed char)(r - pad);
            p.dx[r * S + s] = (signed char)(s - pad);
            p.wt[r * S + s] = (unsigned char)(r * S + s);
        }
}

// out[oy] gathers in[(oy + pad - r)/stride] for taps with (oy + pad - r) % stride == 0
void build_transposed(IgemmArgs &g, int R, int S, int stride, int pad, int OH, int OW)
{
    g.nphase = stride * stride;
    g.isy = g.isx = 1;
    g.osy = g.osx = stride;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            TapPhase &p = g.ph[py * stride + px];
            p.ooy = py;
            p.oox = px;
            p.qh = (OH - py + stride - 1) / stride;
            p.qw = (OW - px + stride - 1) / stride;
            if (p.qh < 0) p.qh = 0;
            if (p.qw < 0) p.qw = 0;
            int nt = 0;
            for (int r = 0; r < R; ++r) {
                if (((py + pad - r) % stride + stride) % stride != 0) continue;
                for (int s = 0; s < S; ++s) {
                    if (((px + pad - s) % stride + stride) % stride != 0) continue;
                    // floor division for negative numerators is exact here (remainder is 0)
                    p.dy[nt] = (signed char)((py + pad - r) / stride);
                    p.dx[nt] = (signed char)((px + pad - s) / stride);
                    p.wt[nt] = (unsigned char)(r * S + s);
                    ++nt;
                }
            }
            p.ntaps = nt;
        }
}

// Tile configurations (all 4 wavefronts / 256 threads): {BM, BN, WM, WN, relative efficiency}.
// 128x128 has the most flop per staged byte; 64x192 and 128x64 exist so that N = 192 / 320 / 576 do not
// pad to a multiple of 128 (25 % / 20 % / 11 % of the MFMAs would multiply zeros).
struct TileCfg {
    int bm, bn;
    float eff;
};
// relative efficiencies measured on MI355X with STEM_IGEMM_CFG sweeps of tools/kernel_bench.py (padding-free shapes)
constexpr int NCFG = 6;
constexpr TileCfg kCfg[NCFG] = {{128, 128, 0.88f}, {64, 192, 0.95f}, {128, 64, 0.85f}, {64, 64, 0.90f},
                                {128, 192, 1.00f} /* 8 waves */, {128, 128, 0.97f} /* 8 waves */};

struct Plan {
    int cfg;
    int nsplit, cps;
    size_t ws_bytes;
};
// head of the split-K workspace: one int per output tile of the launch (split layers have few tiles by construction)
constexpr size_t kCntBytes = 64 * 1024;

// Tile / split-K choice: minimise padded MFMA work / efficiency, then split the reduction (taps x channels)
// over blockIdx.z when the output is too small to fill 256 CUs.
Plan make_plan(const IgemmArgs &g, bool c4, int only_cfg = -1)
{
    int maxM = 0, maxchunks = 0;
    const int nkc = c4 ? 1 : cdiv(g.C, KC);
    for (int p = 0; p < g.nphase; ++p) {
        const int m = g.B * g.ph[p].qh * g.ph[p].qw;
        if (m > maxM) maxM = m;
        const int nc = c4 ? cdiv(g.ph[p].ntaps, 8) : g.ph[p].ntaps * nkc;
        if (nc > maxchunks) maxchunks = nc;
    }
    Plan pl{0, 1, maxchunks, 0};
    if (maxM == 0) return pl;
    const bool can_split = g.epi != EPI_GDN && g.epi != EPI_IGDN && g.epi != EPI_NORM && !c4 && !g.fuse;
    // MFMA-bound model: time ~ (workgroups on the most loaded CU) x (chunks + fixed prologue/epilogue) x tile / efficiency
    double best = 1e300;
    static const int forced = getenv("STEM_IGEMM_CFG") ? atoi(getenv("STEM_IGEMM_CFG")) : -1;     // tuning aid
    for (int c = 0; c < NCFG; ++c) {
        if (only_cfg >= 0 && c != only_cfg) continue;
        if (forced >= 0 && only_cfg < 0 && c != forced && !g.fuse) continue;
        if (g.fuse && !((c == 1 || c == 4) && kCfg[c].bn >= g.N)) continue;      // fused GDN: all channels in one 192-wide tile
        static const int fuse_cfg = getenv("STEM_IGEMM_FUSE_CFG") ? atoi(getenv("STEM_IGEMM_FUSE_CFG")) : -1;     // tuning aid
        if (g.fuse && fuse_cfg >= 0 && c != fuse_cfg) continue;
        const long tm = cdiv(maxM, kCfg[c].bm), tn = cdiv(g.N, kCfg[c].bn);
        const long tiles = tm * tn * g.nphase;
        const int min_cps = kCfg[c].bm * kCfg[c].bn >= 128 * 96 ? 8 : 4;
        const int max_split = can_split ? (maxchunks / min_cps > 64 ? 64 : (maxchunks / min_cps < 1 ? 1 : maxchunks / min_cps)) : 1;
        for (int split = 1; split <= max_split; ++split) {
            const int cps = cdiv(maxchunks, split);
            if (split > 1 && cdiv(maxchunks, cps) != split) continue;          // same schedule as a smaller split
            const long blocks = tiles * split;
            const long per_cu = cdiv((int)blocks, 256);
            // resident wavefronts per SIMD decide how well staging / barriers hide under the other waves' MFMAs
            // (measured with padded LDS: 1 wave 0.75, 2 waves 0.91, 4 waves 1.0 of the same tile's throughput)
            static const int maxblk[NCFG] = {2, 2, 2, 4, 1, 2}, wpb[NCFG] = {1, 1, 1, 1, 2, 2};
            static const double occf[5] = {0.0, 0.75, 0.91, 0.96, 1.0};
            const int resmax = maxblk[c] * wpb[c];
            // Time in units of "one workgroup's chunk with the CU to itself".  n co-resident workgroups share the CU's MFMA
            // pipes: a round of n costs n / occf(n).  Only maxblk fit (LDS); the rest runs in further rounds, and a partly
            // filled later round costs a FULL one: when the first round drains, the dispatcher hands the leftovers to the
            // first CUs that free up, 4 at a time, instead of spreading them (SQ counters on TPM.2: 1280 workgroups = 1.25
            // rounds ran 424 k cycles with the MFMA pipes busy for 256 k; the old model priced that as 5 units instead of 8).
            double units;
            if (per_cu <= maxblk[c]) {
                const int res = (int)per_cu * wpb[c];
                units = (double)per_cu / (occf[res > 4 ? 4 : res] / occf[resmax > 4 ? 4 : resmax]);
            } else {
                units = (double)cdiv((int)per_cu, maxblk[c]) * maxblk[c];
            }
            double cost = units * (cps + 3) * kCfg[c].bm * kCfg[c].bn / kCfg[c].eff;
            if (split > 1) cost += 0.15 * (double)split * maxM * g.N * 32.0 / 256.0;   // slab write + reduce pass
            if (cost < best) {
                best = cost;
                pl.cfg = c;
                pl.nsplit = split;
            }
        }
    }
    static const int forced_split = getenv("STEM_IGEMM_SPLIT") ? atoi(getenv("STEM_IGEMM_SPLIT")) : 0;      // tuning aid
    if (forced_split > 0 && can_split) pl.nsplit = forced_split < maxchunks ? forced_split : maxchunks;
    pl.cps = cdiv(maxchunks, pl.nsplit);
    pl.nsplit = cdiv(maxchunks, pl.cps);           // drop empty trailing splits
    static const bool verbose = getenv("STEM_IGEMM_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[igemm plan] M=%d N=%d C=%d phases=%d chunks=%d -> tile %dx%d split %d (cps %d)\n", maxM, g.N, g.C,
                g.nphase, maxchunks, kCfg[pl.cfg].bm, kCfg[pl.cfg].bn, pl.nsplit, pl.cps);
    if (pl.nsplit > 1) {
        const long tiles = (long)cdiv(maxM, kCfg[pl.cfg].bm) * cdiv(g.N, kCfg[pl.cfg].bn) * g.nphase;
        if (tiles * (long)sizeof(int) > (long)kCntBytes) {         // cannot happen for the shapes that split; stay correct anyway
            pl.nsplit = 1;
            pl.cps = maxchunks;
        } else {
            pl.ws_bytes = kCntBytes + (size_t)pl.nsplit * g.B * g.OH * g.OW * g.N * sizeof(float);
        }
    }
    return pl;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const IgemmArgs &g, bool vec, bool c4, hipStream_t st)
{
    int maxM = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const int m = g.B * g.ph[p].qh * g.ph[p].qw;
        if (m > maxM) maxM = m;
    }
    if (maxM == 0 || g.N == 0) return 0;
    dim3 grid(cdiv(maxM, BM), cdiv(g.N, BN), g.nphase * g.nsplit), block((BM / WM) * (BN / WN) * 64);
    static const size_t extra_lds = getenv("STEM_IGEMM_EXTRA_LDS") ? atoi(getenv("STEM_IGEMM_EXTRA_LDS")) : 0;   // occupancy experiments
    size_t lds = (size_t)2 * (BM + BN) * PITCH * sizeof(float) + 32 * 4 * sizeof(int) + extra_lds;
    constexpr bool can_fuse = BN == 192;
    const size_t lds_fuse = ((size_t)BM * (BN + 4) + (size_t)BN * PITCH) * sizeof(float);
    if (g.fuse && lds_fuse > lds) lds = lds_fuse;
    static bool attr_done = false;     // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel
    if (!attr_done) {
        const int mx = (int)(lds_fuse > lds ? lds_fuse : lds) + 1024;
        (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if constexpr (can_fuse) {
            (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
            (void)hipFuncSetAttribute((const void *)igemm_kernel<BM, BN, WM, WN, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        }
        attr_done = true;
    }
    const bool gdn = g.epi == EPI_GDN || g.epi == EPI_IGDN || g.epi == EPI_NORM;
    if (g.fuse) {
        if constexpr (can_fuse) {
            if (!vec) {
                stem_set_error("igemm: fused GDN needs 16-byte aligned channel counts / pitches");
                return -1;
            }
            if (c4)
                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, true, false, true>), grid, block, lds, st, g);
            else
                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, false, false, true>), grid, block, lds, st, g);
        } else {
            stem_set_error("igemm: fused GDN needs a 192-wide tile");
            return -1;
        }
    } else if (c4)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, true, false>), grid, block, lds, st, g);
    else if (gdn && vec)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, false, true>), grid, block, lds, st, g);
    else if (gdn)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, false, true>), grid, block, lds, st, g);
    else if (vec)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, false, false>), grid, block, lds, st, g);
    else
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, false, false>), grid, block, lds, st, g);
    STEM_LAUNCH_CHECK("igemm");
    return 0;
}

// run one planned launch (+ the split-K reduction); g must already carry pointers, pitches and byte ranges
int run_plan(IgemmArgs &g, const Plan &pl, bool vec, bool c4, void *ws, hipStream_t st)
{
    g.nsplit = pl.nsplit;
    g.cps = pl.cps;
    g.cnt = (int *)ws;                                        // [kCntBytes] arrival counters, then the partial slabs
    g.ws = (float *)((char *)ws + kCntBytes);
    g.slab = (long)g.B * g.OH * g.OW * g.N;
    int rc;
    switch (pl.cfg) {
    case 0: rc = launch_cfg<128, 128, 64, 64>(g, vec, c4, st); break;
    case 1: rc = launch_cfg<64, 192, 32, 96>(g, vec, c4, st); break;
    case 2: rc = launch_cfg<128, 64, 64, 32>(g, vec, c4, st); break;
    case 4: rc = launch_cfg<128, 192, 32, 96>(g, vec, c4, st); break;
    case 5: rc = launch_cfg<128, 128, 32, 64>(g, vec, c4, st); break;
    default: rc = launch_cfg<64, 64, 32, 32>(g, vec, c4, st); break;
    }
    return rc;
}

// Per-geometry choice from measurements (STEM_IGEMM_AUTOTUNE=1): the first launch of a geometry times the model's best
// split for every tile (plus half / double that split) on the caller's own operands -- the kernels are pure functions of
// their inputs, so re-running them is harmless -- and the fastest plan is cached for the life of the process.  Costs one
// stream synchronisation per new geometry; never changes results beyond the summation order split-K already varies.
struct TuneKey {
    int maxM, N, C, chunks, nphase, flags;
    bool operator<(const TuneKey &o) const { return memcmp(this, &o, sizeof(TuneKey)) < 0; }
};
std::map<TuneKey, Plan> g_tuned;
std::mutex g_tuned_mu;

int launch(IgemmArgs &g, bool c4, void *ws, size_t ws_bytes, hipStream_t st)
{
    const bool vec = c4 || ((g.C % 4 == 0) && (g.ldx % 4 == 0) && (g.ldw % 4 == 0) &&
                            (((uintptr_t)g.x & 15) == 0) && (((uintptr_t)g.w & 15) == 0));
    Plan pl = make_plan(g, c4);
    if (pl.nsplit > 1 && (ws == nullptr || ws_bytes < pl.ws_bytes)) {      // no workspace: run unsplit (slower, same result)
        pl.nsplit = 1;
        pl.ws_bytes = 0;
        int maxchunks = 0;
        for (int p = 0; p < g.nphase; ++p) {
            const int nc = c4 ? cdiv(g.ph[p].ntaps, 8) : g.ph[p].ntaps * cdiv(g.C, KC);
            if (nc > maxchunks) maxchunks = nc;
        }
        pl.cps = maxchunks;
    }
    {
        const long xb = (((long)g.B * g.H * g.W - 1) * g.ldx + g.C) * 4;
        int maxwt = 0;
        for (int p = 0; p < g.nphase; ++p)
            for (int t = 0; t < g.ph[p].ntaps; ++t)
                if (g.ph[p].wt[t] > maxwt) maxwt = g.ph[p].wt[t];
        const long wb = c4 ? (long)g.N * g.ldw * 4 : ((((long)maxwt + 1) * g.N - 1) * g.ldw + g.C) * 4;
        if (xb >= 0x7FFFFF00L || wb >= 0x40000000L) {
            stem_set_error("igemm: tensor view of %ld / %ld bytes exceeds the 2 GiB buffer-descriptor range", xb, wb);
            return -1;
        }
        g.xbytes = (int)xb;
        g.wbytes = (int)wb;
    }
    static const int exper = STEM_EXPER_ENV("STEM_IGEMM_EXPER") ? atoi(STEM_EXPER_ENV("STEM_IGEMM_EXPER")) : 0;     // ablations: -DSTEM_EXPERIMENTS builds only
    g.exper = exper;
    g.ident = (g.nphase == 1 && g.osy == 1 && g.osx == 1 && g.ph[0].ooy == 0 && g.ph[0].oox == 0 &&
               g.ph[0].qh == g.OH && g.ph[0].qw == g.OW) ? 1 : 0;
    static const bool autotune = getenv("STEM_IGEMM_AUTOTUNE") && atoi(getenv("STEM_IGEMM_AUTOTUNE")) != 0;
    if (!autotune || c4 || g.fuse || !vec) return run_plan(g, pl, vec, c4, ws, st);

    int maxM = 0, maxchunks = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const int m = g.B * g.ph[p].qh * g.ph[p].qw;
        if (m > maxM) maxM = m;
        const int nc = g.ph[p].ntaps * cdiv(g.C, KC);
        if (nc > maxchunks) maxchunks = nc;
    }
    const TuneKey key{maxM, g.N, g.C, maxchunks, g.nphase, g.epi * 4 + (ws ? 1 : 0) + (g.ident ? 2 : 0)};
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            Plan tp = it->second;
            if (tp.nsplit > 1 && (ws == nullptr || ws_bytes < tp.ws_bytes)) tp = pl;
            return run_plan(g, tp, vec, c4, ws, st);
        }
    }
    // candidates: for every tile, the model's split and its neighbours, as far as the caller's workspace reaches
    std::vector<Plan> cands;
    const bool can_split = g.epi != EPI_GDN && g.epi != EPI_IGDN && g.epi != EPI_NORM;
    const size_t out_bytes = (size_t)g.B * g.OH * g.OW * g.N * sizeof(float);
    for (int c = 0; c < NCFG; ++c) {
        const Plan base = make_plan(g, c4, c);
        const int trial[3] = {base.nsplit, base.nsplit / 2, base.nsplit * 2};
        for (int q = 0; q < 3; ++q) {
            int sp = trial[q] < 1 ? 1 : trial[q];
            if (!can_split) sp = 1;
            if (sp > maxchunks) sp = maxchunks;
            Plan cnd{c, sp, cdiv(maxchunks, sp), 0};
            cnd.nsplit = cdiv(maxchunks, cnd.cps);
            cnd.ws_bytes = cnd.nsplit > 1 ? kCntBytes + cnd.nsplit * out_bytes : 0;
            if (cnd.nsplit > 1 && (long)cdiv(maxM, kCfg[c].bm) * cdiv(g.N, kCfg[c].bn) * g.nphase * (long)sizeof(int) > (long)kCntBytes) continue;
            if (cnd.nsplit > 1 && (ws == nullptr || cnd.ws_bytes > ws_bytes)) continue;
            bool dup = false;
            for (const Plan &o : cands) dup |= (o.cfg == cnd.cfg && o.nsplit == cnd.nsplit);
            if (!dup) cands.push_back(cnd);
        }
    }
    hipEvent_t e0, e1;
    if (cands.size() < 2 || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return run_plan(g, pl, vec, c4, ws, st);
    Plan best = pl;
    float best_ms = 1e30f;
    for (const Plan &cnd : cands) {
        IgemmArgs t = g;
        if (run_plan(t, cnd, vec, c4, ws, st)) continue;               // warm (code objects, caches)
        (void)hipEventRecord(e0, st);
        t = g;
        (void)run_plan(t, cnd, vec, c4, ws, st);
        t = g;
        (void)run_plan(t, cnd, vec, c4, ws, st);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best_ms) {
            best_ms = ms;
            best = cnd;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    static const bool verbose = getenv("STEM_IGEMM_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[igemm tune] M=%d N=%d C=%d chunks=%d: model %dx%d/%d -> measured %dx%d/%d (%.1f us)\n", maxM, g.N, g.C, maxchunks,
                kCfg[pl.cfg].bm, kCfg[pl.cfg].bn, pl.nsplit, kCfg[best.cfg].bm, kCfg[best.cfg].bn, best.nsplit, best_ms * 500.f);
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        g_tuned[key] = best;
    }
    return run_plan(g, best, vec, c4, ws, st);           // the timed runs already produced the output; one more keeps it simple
}

int check_common(const char *name, const void *x, const void *w, const void *y, int B, int H, int W, int C, int K,
                 int R, int S, int stride)
{
    STEM_CHECK_ARG(x && w && y, "%s: null pointer", name);
    STEM_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && K > 0, "%s: bad dims B=%d H=%d W=%d C=%d K=%d", name, B, H, W, C, K);
    STEM_CHECK_ARG(R >= 1 && S >= 1 && R * S <= 25, "%s: kernel %dx%d unsupported (max 25 taps)", name, R, S);
    STEM_CHECK_ARG(stride == 1 || stride == 2, "%s: stride %d unsupported", name, stride);
    return 0;
}

}   // namespace

// =================================================================================================
namespace {

enum { KIND_CONV_FWD = 0, KIND_CONV_DGRAD = 1, KIND_DECONV_FWD = 2, KIND_DECONV_DGRAD = 3 };

// geometry of the four convolution-shaped ops in terms of the generic kernel (B,H,W,C,K = the LAYER's dims)
void fill_geometry(IgemmArgs &g, int kind, int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad)
{
    memset(&g, 0, sizeof(g));
    g.B = B;
    const int Hc = (H + 2 * pad - R) / stride + 1, Wc = (W + 2 * pad - S) / stride + 1;              // Conv2d output
    const int Hd = (H - 1) * stride - 2 * pad + R + opad, Wd = (W - 1) * stride - 2 * pad + S + opad;  // ConvTranspose2d output
    switch (kind) {
    case KIND_CONV_FWD:        // x[B,H,W,C] -> y[B,Hc,Wc,K]
        g.H = H; g.W = W; g.C = C; g.N = K; g.OH = Hc; g.OW = Wc; g.ldw = C;
        build_direct(g, R, S, stride, pad, Hc, Wc);
        break;
    case KIND_CONV_DGRAD:      // dy[B,Hc,Wc,K] -> dx[B,H,W,C]
        g.H = Hc; g.W = Wc; g.C = K; g.N = C; g.OH = H; g.OW = W; g.ldw = K;
        build_transposed(g, R, S, stride, pad, H, W);
        break;
    case KIND_DECONV_FWD:      // x[B,H,W,C] -> y[B,Hd,Wd,K]
        g.H = H; g.W = W; g.C = C; g.N = K; g.OH = Hd; g.OW = Wd; g.ldw = C;
        build_transposed(g, R, S, stride, pad, Hd, Wd);
        break;
    default:                   // dy[B,Hd,Wd,K] -> dx[B,H,W,C]
        g.H = Hd; g.W = Wd; g.C = K; g.N = C; g.OH = H; g.OW = W; g.ldw = K;
        build_direct(g, R, S, stride, pad, H, W);
        break;
    }
}

}   // namespace

STEM_EXPORT size_t stem_conv_workspace_bytes(int kind, int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad)
{
    const bool masked = (kind & STEM_CONV_MASKED_A) != 0;
    kind &= ~STEM_CONV_MASKED_A;
    if (kind < 0 || kind > 3 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R * S > 25 || R < 1 || S < 1) return 0;
    IgemmArgs g;
    fill_geometry(g, kind, B, H, W, C, K, R, S, stride, pad, opad);
    if (g.OH <= 0 || g.OW <= 0) return 0;
    if (masked && kind == KIND_CONV_FWD) g.ph[0].ntaps = (R / 2) * S + S / 2;
    static const bool autotune = getenv("STEM_IGEMM_AUTOTUNE") && atoi(getenv("STEM_IGEMM_AUTOTUNE")) != 0;
    const Plan pl = make_plan(g, false);
    if (!autotune || pl.nsplit <= 1) return pl.ws_bytes;
    // room for the measured choice to split twice as far as the model would (layers the model leaves unsplit stay so)
    return kCntBytes + (size_t)pl.nsplit * 2 * g.B * g.OH * g.OW * g.N * sizeof(float);
}

STEM_EXPORT int stem_conv2d_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, int ldy,
                                int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                int act, float slope, void *ws, size_t ws_bytes, void *stream)
{
    if (check_common("stem_conv2d_fwd", x, wp, y, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_FWD, B, H, W, C, K, R, S, stride, pad, 0);
    STEM_CHECK_ARG(g.OH > 0 && g.OW > 0, "stem_conv2d_fwd: empty output");
    if (act & STEM_CONV_MASKED_A) {      // type-A mask: the live taps are the first (R/2)*S + S/2 in raster order
        g.ph[0].ntaps = (R / 2) * S + S / 2;
        STEM_CHECK_ARG(g.ph[0].ntaps >= 1, "stem_conv2d_fwd: a masked 1x1 convolution has no live tap");
        act &= ~STEM_CONV_MASKED_A;
    }
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = ldx; g.ldy = ldy;
    g.epi = act == STEM_ACT_LRELU ? EPI_LRELU : EPI_BIAS;
    g.slope = slope;
    return launch(g, false, ws, ws_bytes, (hipStream_t)stream);
}

STEM_EXPORT int stem_conv2d_fwd_c4(const float *x4, const float *wp, const float *bias, float *y, int ldy,
                                   int B, int H, int W, int K, int R, int S, int stride, int pad, void *stream)
{
    if (check_common("stem_conv2d_fwd_c4", x4, wp, y, B, H, W, 4, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_FWD, B, H, W, 4, K, R, S, stride, pad, 0);
    g.x = x4; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = 4; g.ldw = 128; g.ldy = ldy;
    g.epi = EPI_BIAS;
    return launch(g, true, nullptr, 0, (hipStream_t)stream);
}

STEM_EXPORT int stem_conv2d_dgrad(const float *dy, int lddy, const float *wp, float *dx, int lddx,
                                  const float *xact, int ldxact, float slope,
                                  int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                  void *ws, size_t ws_bytes, void *stream)
{
    if (check_common("stem_conv2d_dgrad", dy, wp, dx, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_DGRAD, B, H, W, C, K, R, S, stride, pad, 0);
    g.x = dy; g.w = wp; g.y = dx; g.z = xact;
    g.ldx = lddy; g.ldy = lddx; g.ldz = ldxact;
    g.epi = xact ? EPI_DACT : EPI_BIAS;
    g.slope = slope;
    return launch(g, false, ws, ws_bytes, (hipStream_t)stream);
}

STEM_EXPORT int stem_deconv2d_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, int ldy,
                                  int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                                  int act, float slope, void *ws, size_t ws_bytes, void *stream)
{
    if (check_common("stem_deconv2d_fwd", x, wp, y, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_DECONV_FWD, B, H, W, C, K, R, S, stride, pad, opad);
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = ldx; g.ldy = ldy;
    g.epi = act == STEM_ACT_LRELU ? EPI_LRELU : EPI_BIAS;
    g.slope = slope;
    return launch(g, false, ws, ws_bytes, (hipStream_t)stream);
}

STEM_EXPORT int stem_deconv2d_dgrad(const float *dy, int lddy, const float *wp, float *dx, int lddx,
                                    const float *xact, int ldxact, float slope,
                                    int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                                    void *ws, size_t ws_bytes, void *stream)
{
    if (check_common("stem_deconv2d_dgrad", dy, wp, dx, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_DECONV_DGRAD, B, H, W, C, K, R, S, stride, pad, opad);
    g.x = dy; g.w = wp; g.y = dx; g.z = xact;
    g.ldx = lddy; g.ldy = lddx; g.ldz = ldxact;
    g.epi = xact ? EPI_DACT : EPI_BIAS;
    g.slope = slope;
    return launch(g, false, ws, ws_bytes, (hipStream_t)stream);
}

namespace {
int set_fuse(IgemmArgs &g, const float *beta, const float *gamma, int inverse, float beta_min, const char *name)
{
    STEM_CHECK_ARG(beta && gamma, "%s: null beta/gamma", name);
    STEM_CHECK_ARG(g.N <= 192 && g.N % 4 == 0, "%s: fused GDN supports up to 192 channels (multiple of 4), got %d", name, g.N);
    g.beta = beta;
    g.gamma = gamma;
    g.fuse = inverse ? 2 : 1;
    g.gmbytes = g.N * g.N * 4;
    g.beta_bound = (float)sqrt((double)beta_min + 1.4551915228366852e-11);
    g.epi = EPI_BIAS;
    return 0;
}
}   // namespace

STEM_EXPORT int stem_conv2d_gdn_fwd(const float *x, int ldx, const float *wp, const float *bias, const float *beta,
                                    const float *gamma, float *y, int ldy, int B, int H, int W, int C, int K, int R, int S,
                                    int stride, int pad, int inverse, float beta_min, void *stream)
{
    if (check_common("stem_conv2d_gdn_fwd", x, wp, y, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_FWD, B, H, W, C, K, R, S, stride, pad, 0);
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = ldx; g.ldy = ldy;
    if (set_fuse(g, beta, gamma, inverse, beta_min, "stem_conv2d_gdn_fwd")) return -1;
    return launch(g, false, nullptr, 0, (hipStream_t)stream);
}

STEM_EXPORT int stem_conv2d_fwd_c4_gdn(const float *x4, const float *wp, const float *bias, const float *beta,
                                       const float *gamma, float *y, int ldy, int B, int H, int W, int K, int R, int S,
                                       int stride, int pad, int inverse, float beta_min, void *stream)
{
    if (check_common("stem_conv2d_fwd_c4_gdn", x4, wp, y, B, H, W, 4, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_FWD, B, H, W, 4, K, R, S, stride, pad, 0);
    g.x = x4; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = 4; g.ldw = 128; g.ldy = ldy;
    if (set_fuse(g, beta, gamma, inverse, beta_min, "stem_conv2d_fwd_c4_gdn")) return -1;
    return launch(g, true, nullptr, 0, (hipStream_t)stream);
}

STEM_EXPORT int stem_deconv2d_gdn_fwd(const float *x, int ldx, const float *wp, const float *bias, const float *beta,
                                      const float *gamma, float *y, int ldy, int B, int H, int W, int C, int K, int R, int S,
                                      int stride, int pad, int opad, int inverse, float beta_min, void *stream)
{
    if (check_common("stem_deconv2d_gdn_fwd", x, wp, y, B, H, W, C, K, R, S, stride)) return -1;
    IgemmArgs g;
    fill_geometry(g, KIND_DECONV_FWD, B, H, W, C, K, R, S, stride, pad, opad);
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.ldx = ldx; g.ldy = ldy;
    if (set_fuse(g, beta, gamma, inverse, beta_min, "stem_deconv2d_gdn_fwd")) return -1;
    return launch(g, false, nullptr, 0, (hipStream_t)stream);
}

STEM_EXPORT int stem_gdn_fwd(const float *x, int ldx, const float *beta, const float *gamma, float *y, int ldy,
                             int B, int H, int W, int C, int inverse, float beta_min, void *stream)
{
    if (check_common("stem_gdn_fwd", x, gamma, y, B, H, W, C, C, 1, 1, 1)) return -1;
    STEM_CHECK_ARG(beta, "stem_gdn_fwd: null beta");
    IgemmArgs g;
    fill_geometry(g, KIND_CONV_FWD, B, H, W, C, C, 1, 1, 1, 0, 0);
    g.x = x; g.w = gamma; g.y = y; g.z = x; g.beta = beta;
    g.ldx = ldx; g.ldy = ldy; g.ldz = ldx;
    g.epi = inverse == 2 ? EPI_NORM : (inverse ? EPI_IGDN : EPI_GDN);      // inverse == 2: write the denominator n (backward)
    g.asquare = 1;
    g.breparam = 1;
    g.beta_bound = (float)sqrt((double)beta_min + 1.4551915228366852e-11);
    return launch(g, false, nullptr, 0, (hipStream_t)stream);
}
