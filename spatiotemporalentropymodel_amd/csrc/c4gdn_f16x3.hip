// First analysis layer g_a.0 + GDN g_a.1 (compressai/models/priors.py:421-423, gdn.py:52-67) as ONE kernel on the fp16 matrix
// cores with fp32-class products: a 3-channel image -> N = 64 / 128 / 192 channels (5x5 stride 2 in the reference), divided by
// sqrt(beta + gamma . x^2), written pre-split as two fp16 planes for conv_f16x3.hip (and / or as fp32 NHWC).
//
// The fp32-MFMA kernel this replaces (igemm.hip, C4 + FUSE) ran at 0.29 of its pipe: 72 % of its flop is the K = N GDN
// contraction on the 64-flop/clk instruction, its 128 x 192 tile with the squared values parked in LDS left one workgroup per
// CU, and nothing overlapped the planes write.  Here
//   * both contractions are computed TRANSPOSED, D[channel][pixel] = A[channel][k] . B[k][pixel]: the accumulator of
//     v_mfma_f32_32x32x16_f16 keeps one PIXEL per lane (column) and 16 channels in its registers -- exactly the B-operand
//     shape of the next MFMA that sums over channels.  The squared conv outputs therefore go from the accumulators into the
//     GDN contraction as registers (square, scale, split into two fp16 planes, pack): no LDS tile, no cross-lane movement;
//   * a wavefront owns 32 pixels x ALL N channels (x: N/32 accumulators), so a workgroup needs LDS only for the A-operand
//     stream (conv weights, then gamma), which is the same for every workgroup: it is pre-split into fp16 planes and stored in
//     fragment order by c4gdn_pack_kernel, and every 12 KiB chunk of it is copied global -> LDS by global_load_lds (no
//     registers) one chunk ahead of its use; 24 KiB of LDS per workgroup, several workgroups per CU, so that the epilogue
//     stores of one overlap the MFMAs of another;
//   * three fp16 MFMAs per fp32 product as in conv_f16x3.hip (operands as two fp16 numbers of a power-of-two-scaled copy,
//     the products a0.b0, a0.b1, a1.b0): fp32-class accuracy, fp32 accumulation; the image is scaled by its measured maximum,
//     the squared conv outputs and the result by upper bounds derived from it (scale records, stem_common.h);
//   * the image patch of a pixel is read straight from the NHWC4 image: a conv k-step is FIVE taps of one filter row x 3 image
//     channels = 15 K slots + one zero slot (round 6: the 75 real products of the 5x5 filter in 80 slots / 5 k-steps; until round 5
//     tap pairs x 4 channels = 128 slots / 8 k-steps, a quarter of them multiplying the NHWC4 padding channel).  Lane half h reads
//     the three pixels at columns 5 u + 2 h + {0, 1, 2} of the row (three 16-byte loads, zero-filled outside the image) and picks
//     its eight slots from them.
//
// Channel order inside a 32-row MFMA tile: row r of the A operand holds channel 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3), which
// makes the 16 accumulator registers of lane (pixel p, half h) the CONTIGUOUS channels 16 h .. 16 h + 15 of the tile: k-step s
// of the GDN contraction takes registers 8 s .. 8 s + 7 (channels 16 h + 8 s + j, what the gamma fragments are packed for), and
// the epilogue stores 16 consecutive channels per lane (32 bytes per plane, 64 bytes of fp32).
#include <math.h>

#include "stem_common.h"

namespace {

typedef hp8 h16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK = 12288;                    // bytes of one A-operand chunk: 6 tiles x 2 planes x 64 lanes x 16 B
constexpr int OOR = 0x7FFFFF00;                 // voffset that every buffer view rejects (returns 0)
constexpr int OOR_ST = 0x7FFF0000;              // the same with room for the stores' constant offsets (views are < 0x7FFF0000 bytes)
constexpr int WG_PIX = 128, NTHREADS = 256;     // 4 wavefronts x 32 pixels

__host__ __device__ inline int rho(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }


struct C4gArgs {
    const float *x4;               // [B][H][W][4] fp32 (stem_nchw3_to_nhwc4)
    const float *xq;               // scale record of the image: slots = max |x| (stem_nchw3_to_nhwc4 / stem_amax_nhwc)
    const unsigned char *astream;  // c4gdn_pack_kernel's output
    const float *wq, *gq;          // scale records of the two parts of the stream (conv weights, gamma'), behind its chunks
    float *yq;                     // scale record of the output
    const float *bias, *beta;
    float *y;
    void *yp;
    int ldy;
    int B, H, W, N, OH, OW, R, S, stride, pad;
    int ksc;                       // conv k-steps: R * ceil(S / 5)
    int xbytes;
    float beta_bound;
    int ablate;                    // libstem_hip_exper.so only (WRONG results, timing ablations): bit 1 = the ring's wait + barrier only at every second
                                   // step (bit 0 was "5 instead of 8 conv k-steps" before the kernel did that itself: profiles/r06_c4gdn_ablation.log)
};
#ifdef STEM_EXPERIMENTS
int g_c4g_ablate = 0;
#endif

// ---- A-operand stream -------------------------------------------------------------------------------------------------------
// chunk t < ksc (conv k-step t = filter row t / SPR, column group u = t % SPR, SPR = ceil(S / 5)): fragment (nb, plane) at
//   ((nb * 2 + plane) * 64 + lane) * 16; lane (r, h) element j is the weight of channel nb * 32 + rho(r) for K slot 8 h + j of the
//   step: slot q < 15 = (tap column 5 u + q / 3, image channel q % 3), slot 15 = 0; columns beyond the filter hold 0;
// chunk ksc + 2 kb + s (GDN k-step s of k-block kb): fragment (nb, plane) at the same place; lane (r, h) element j =
//   gamma'[nb * 32 + rho(r)][kb * 32 + 16 h + 8 s + j], gamma' = max(gamma, 2^-18)^2 - 2^-36 (parametrizers.py:42-45).
__global__ __launch_bounds__(256) void c4gdn_pack_kernel(const float *wp_c4, const float *gamma, unsigned char *out, int N, int R, int S, int ksc,
                                                         float *wq, float *gq)
{
    const int NB = N / 32, SPR = (S + 4) / 5;
    const int nchunks = ksc + 2 * NB;
    // scales: the weights by their maximum, gamma' = max(gamma, 2^-18)^2 - 2^-36 by the bound max(|gamma|max, 2^-18)^2
    __shared__ float qred[16];
    const int we = q_exp(q_amax(wq, qred));
    const float gm = fmaxf(q_amax(gq, qred), 3.814697265625e-06f);
    const int ge = q_exp(gm * gm);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        wq[1] = q_pow2(-we);
        gq[1] = q_pow2(-ge);
    }
    const int e = blockIdx.x * 256 + threadIdx.x;           // (chunk, fragment position without the plane, lane)
    const int lane = e & 63, fp = (e >> 6) % 6, c = (e >> 6) / 6;
    if (c >= nchunks) return;
    const int r = lane & 31, h = lane >> 5;
    float v[8];
    const int nb = fp;
    const bool used = nb < NB;
    if (c < ksc) {
        const int ch = nb * 32 + rho(r), prow = c / SPR, u = c - prow * SPR;
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * h + j, s = 5 * u + q / 3, cc = q % 3;
            v[j] = (used && q < 15 && prow < R && s < S) ? wp_c4[(ch * 32 + prow * S + s) * 4 + cc] : 0.f;
        }
    } else {
        const int gi = c - ksc, kb = gi >> 1, s = gi & 1;
        const int n = nb * 32 + rho(r), k0 = kb * 32 + 16 * h + 8 * s;
        for (int j = 0; j < 8; ++j) {
            float gv = 0.f;
            if (used) {
                gv = fmaxf(gamma[n * N + k0 + j], 3.814697265625e-06f);
                gv = gv * gv - 1.4551915228366852e-11f;
            }
            v[j] = gv;
        }
    }
    if (!used) return;
    h16x8 p0, p1;
    const float sc = q_pow2(c < ksc ? we : ge);
    for (int j = 0; j < 8; ++j) {
        hp_t a, b;
        q_split(v[j], sc, a, b);
        p0[j] = a; p1[j] = b;
    }
    unsigned char *dst = out + (size_t)c * CHUNK + ((size_t)(fp * 2) * 64 + lane) * 16;
    *reinterpret_cast<h16x8 *>(dst) = p0;
    *reinterpret_cast<h16x8 *>(dst + 1024) = p1;
}

// the three products of one fp32 product, smallest terms first (as conv_f16x3.hip)
__device__ inline f32x16 mfma3(const h16x8 (&a)[2], const h16x8 (&b)[2], f32x16 acc)
{
    acc = STEM_MFMA16(a[1], b[0], acc);
    acc = STEM_MFMA16(a[0], b[1], acc);
    acc = STEM_MFMA16(a[0], b[0], acc);
    return acc;
}

// the two planes of one A fragment (consecutive 1 KiB pieces of the chunk, lane-linear: conflict-free ds_read_b128)
__device__ inline void lda(const unsigned char *p, h16x8 (&af)[2])
{
    af[0] = *reinterpret_cast<const h16x8 *>(p);
    af[1] = *reinterpret_cast<const h16x8 *>(p + 1024);
}

template <int NB>
__global__ __launch_bounds__(NTHREADS, 2) void c4gdn_f16x3_kernel(const C4gArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // [2][CHUNK]: the A-operand ring
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int Mtot = a.B * a.OH * a.OW;
    const int m = blockIdx.x * WG_PIX + wave * 32 + p;
    const bool ok = m < Mtot;
    int by, bx, pix0;
    {
        const int mm = ok ? m : 0, ohw = a.OH * a.OW, b = mm / ohw, rem = mm - b * ohw, qy = rem / a.OW, qx = rem - qy * a.OW;
        by = qy * a.stride - a.pad;
        bx = qx * a.stride - a.pad;
        pix0 = (b * a.H + by) * a.W + bx;              // pixel index of tap (0, 0); may be negative at the border (masked below)
    }
    const int nchunks = a.ksc + 2 * NB;
    const int SPR = (a.S + 4) / 5;

    // ---- A-operand ring: chunk c -> buffer c & 1, 3 KiB per wavefront as three 1 KiB pieces (wavefront w copies bytes
    // [3072 w, 3072 w + 3072); buffer form: one VGPR of lane offset, the chunk offset in an SGPR, the piece offsets as immediates)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(a.astream), 0, nchunks * CHUNK, 0x00020000);
    const int ring_v = wave * 3072 + lane * 16;
    auto ring_issue = [&](int c) {
        if (c >= nchunks) return;
        const int co = c * CHUNK;
        unsigned char *dst = smem + (c & 1) * CHUNK;
        auto *ls = (__attribute__((address_space(3))) void *)(dst + wave * 3072);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ls, 16, ring_v, co, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ls, 16, ring_v, co, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ls, 16, ring_v, co, 2048, 0);
    };
    // every wavefront waits for its own pieces, the barrier publishes all of them (and retires the buffer read last step)
    auto ring_wait = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- image patch of this lane's pixel for conv k-step t: filter row t / SPR, tap columns 5 u + 2 h + {0, 1, 2} -------------------
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x4), 0, a.xbytes, 0x00020000);
    auto patch_load = [&](int t, f32x4 (&v)[3]) {
        const int prow = t / SPR, s0 = 5 * (t - prow * SPR) + 2 * h;
        const int iy = by + prow, ix = bx + s0;
        const bool rowok = ok && t < a.ksc && iy >= 0 && iy < a.H;
        const int off = (pix0 + prow * a.W + s0) * 16;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int o = (rowok && ix + i >= 0 && ix + i < a.W && s0 + i < a.S) ? off + 16 * i : OOR;
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
        }
    };

    // ---- scales (stem_common.h): the image by its measured maximum; the squares and the output by upper bounds derived from it --
    __shared__ float qred[16];
    float xscale, cfac, sqscale, nfac, oscale;
    {
        const float xmax = q_amax(a.xq, qred), wmax = q_amax(a.wq, qred);
        float bm = 0.f, btm = 3.0e38f;
        if (tid < a.N) {
            if (a.bias) bm = fabsf(a.bias[tid]);
            const float bb = fmaxf(a.beta[tid], a.beta_bound);
            btm = bb * bb - 1.4551915228366852e-11f;
        }
        bm = block_max(bm, qred);
        btm = -block_max(-btm, qred);
        const float vb = (float)(3 * a.R * a.S) * xmax * wmax + bm;        // |conv output| <= K |x|max |w|max + |bias|max
        const int xe = q_exp(xmax), se = q_exp(vb * vb), oe = q_exp(vb * __builtin_amdgcn_rsqf(fmaxf(btm, 1e-30f)));
        xscale = q_pow2(xe);
        cfac = q_pow2(-xe) * q_inv(a.wq);
        sqscale = q_pow2(se);
        nfac = q_pow2(-se) * q_inv(a.gq);
        oscale = q_pow2(oe);
        if (a.yq && blockIdx.x == 0 && tid == 0) {
            q_header(a.yq, gridDim.x);
            a.yq[1] = a.yp ? q_pow2(-oe) : 1.f;
        }
    }
    const float binit = 1.f / cfac;                      // the accumulators hold conv * 2^(ex + ew): the bias enters scaled alike

    // ---- x[nb][i] = conv output of channel nb * 32 + 16 h + i at this lane's pixel, starting from the bias ---------------------
    f32x16 x[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (a.bias) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + nb * 32 + 16 * h + 4 * i4);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[nb][4 * i4 + e] = bv[e] * binit;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[nb][i] = 0.f;
        }
    }

    ring_issue(0);
    f32x4 pv[3];
    patch_load(0, pv);

    // ---- convolution: ksc chunks of one k-step x NB tiles x 3 products -----------------------------------------------------------
#ifdef STEM_EXPERIMENTS
    const int ksc_run = a.ksc;            // (ablation bit 0 of round 6 -- 5 instead of 8 conv k-steps -- is what the kernel does now)
    int step_no = 0;
#define C4G_RING_WAIT() do { if (!((a.ablate & 2) && (step_no++ & 1))) ring_wait(); } while (0)
#else
    const int ksc_run = a.ksc;
#define C4G_RING_WAIT() ring_wait()
#endif
#pragma unroll 1
    for (int t = 0; t < ksc_run; ++t) {
        C4G_RING_WAIT();
        ring_issue(t + 1);
        h16x8 b[2];
        {
            // the lane's eight K slots out of its three pixels: h = 0: slots 0..7 = p0.xyz p1.xyz p2.xy; h = 1: slots 8..15 = p0.z p1.xyz
            // p2.xyz 0 (its p0 is the tap column 5 u + 2, whose x and y belong to the other half)
            const float e[8] = {h ? pv[0][2] : pv[0][0], h ? pv[1][0] : pv[0][1], h ? pv[1][1] : pv[0][2], h ? pv[1][2] : pv[1][0],
                                h ? pv[2][0] : pv[1][1], h ? pv[2][1] : pv[1][2], h ? pv[2][2] : pv[2][0], h ? 0.f : pv[2][1]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hp_t h0, h1;
                q_split(e[j], xscale, h0, h1);
                b[0][j] = h0; b[1][j] = h1;
            }
        }
        patch_load(t + 1 < a.ksc ? t + 1 : t, pv);
        const unsigned char *buf = smem + (t & 1) * CHUNK + lane * 16;
        h16x8 af[2][2];
        lda(buf, af[0]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {          // fragments of tile nb + 1 are fetched behind the MFMAs of tile nb
            if (nb + 1 < NB) lda(buf + (nb + 1) * 2048, af[(nb + 1) & 1]);
            x[nb] = mfma3(af[nb & 1], b, x[nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- GDN: 2 NB chunks of one k-step (16 channels of x^2) x NB output tiles; B operand = the squared accumulators ----------------
    // outputs through buffer stores: one VGPR of pixel offset each (out of range for the pixels beyond M: dropped by the hardware)
    const int opix = NB * 128;
    const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc(a.yp, 0, a.yp ? Mtot * opix : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryf = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y ? (Mtot - 1) * a.ldy * 4 + a.N * 4 : 0, 0x00020000);
    const int vyp = ok ? m * opix + 32 * h : OOR_ST, vyf = ok ? m * a.ldy * 4 + 64 * h : OOR_ST;
    f32x16 nrm[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            nrm[nb][i] = 0.f;
            x[nb][i] *= cfac;                            // back to true units (exact: a power of two)
        }
    int c = a.ksc;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s, ++c) {
            C4G_RING_WAIT();
            ring_issue(c + 1);
            const unsigned char *buf = smem + (c & 1) * CHUNK + lane * 16;
            h16x8 b[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = x[kb][8 * s + j];
                hp_t h0, h1;
                q_split(v * v, sqscale, h0, h1);
                b[0][j] = h0; b[1][j] = h1;
            }
            h16x8 af[2][2];
            lda(buf, af[0]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (nb + 1 < NB) lda(buf + (nb + 1) * 2048, af[(nb + 1) & 1]);
                nrm[nb] = mfma3(af[nb & 1], b, nrm[nb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float omax = 0.f;
    {
        // epilogue: y = x * rsqrt(beta' + norm), 16 consecutive channels per lane and tile
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float yv[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 bt = *reinterpret_cast<const f32x4 *>(a.beta + nb * 32 + 16 * h + 4 * i4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bb = fmaxf(bt[e], a.beta_bound);
                    yv[4 * i4 + e] = x[nb][4 * i4 + e] * __builtin_amdgcn_rsqf(nrm[nb][4 * i4 + e] * nfac + (bb * bb - 1.4551915228366852e-11f));
                }
            }
            if (ok) {
#pragma unroll
                for (int i = 0; i < 16; ++i) omax = fmaxf(omax, fabsf(yv[i]));
            }
            if (a.y) {
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = yv[4 * i4 + e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ryf, vyf + nb * 128 + 16 * i4, 0, 0);
                }
            }
            if (a.yp) {
                h16x8 q0[2], q1[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    hp_t h0, h1;
                    q_split(yv[i], oscale, h0, h1);
                    q0[i >> 3][i & 7] = h0; q1[i >> 3][i & 7] = h1;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0[e]), ryp, vyp + nb * 128 + 16 * e, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1[e]), ryp, vyp + nb * 128 + 64 + 16 * e, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // one tile's epilogue at a time: 56 live registers instead of 3 x 56
        }
    }
    if (a.yq) {
        omax = block_max(omax, qred);
        if (tid == 0) a.yq[QREC_HDR + blockIdx.x] = omax;
    }
}

int conv_ksteps(int R, int S) { return R * ((S + 4) / 5); }
constexpr int QSLOTS = 16;
constexpr size_t QREC_BYTES = (QREC_HDR + QSLOTS) * sizeof(float);

__global__ __launch_bounds__(256) void c4gdn_amax_kernel(const float *w, long nw, const float *g, long ng, float *wq, float *gq)
{
    __shared__ float qred[16];
    const float *src = blockIdx.y ? g : w;
    const long n = blockIdx.y ? ng : nw;
    float m = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) m = fmaxf(m, fabsf(src[e]));
    m = block_max(m, qred);
    if (threadIdx.x == 0) {
        float *q = blockIdx.y ? gq : wq;
        q[QREC_HDR + blockIdx.x] = m;
        if (blockIdx.x == 0) q_header(q, gridDim.x);
    }
}

}   // namespace

#ifdef STEM_EXPERIMENTS
// tools/debug/c4gdn_time.py <mask>: timing ablations of the first-layer kernel (results are WRONG under them)
STEM_EXPORT void stem_exper_c4gdn_ablate(int mask) { g_c4g_ablate = mask; }
#endif

// ---- C ABI --------------------------------------------------------------------------------------------------------------------
STEM_EXPORT int stem_c4gdn_supported(int N, int R, int S)
{
    return (N == 64 || N == 128 || N == 192) && R >= 1 && S >= 1 && R * S <= 25 && conv_ksteps(R, S) >= 1;
}

STEM_EXPORT size_t stem_c4gdn_stream_bytes(int N, int R, int S)
{
    if (!stem_c4gdn_supported(N, R, S)) return 0;
    return (size_t)(conv_ksteps(R, S) + 2 * (N / 32)) * CHUNK + 2 * QREC_BYTES;       // + the scale records of weights and gamma'
}

STEM_EXPORT int stem_c4gdn_pack(const float *wp_c4, const float *gamma, void *astream, int N, int R, int S, void *stream)
{
    STEM_CHECK_ARG(wp_c4 && gamma && astream, "stem_c4gdn_pack: null pointer");
    STEM_CHECK_ARG(stem_c4gdn_supported(N, R, S), "stem_c4gdn_pack: N must be 64, 128 or 192 and R*S <= 25 (N=%d R=%d S=%d)", N, R, S);
    const int ksc = conv_ksteps(R, S), nchunks = ksc + 2 * (N / 32);
    (void)hipMemsetAsync(astream, 0, (size_t)nchunks * CHUNK, (hipStream_t)stream);
    float *wq = reinterpret_cast<float *>(static_cast<unsigned char *>(astream) + (size_t)nchunks * CHUNK);
    float *gq = wq + QREC_HDR + QSLOTS;
    hipLaunchKernelGGL(c4gdn_amax_kernel, dim3(QSLOTS, 2), dim3(256), 0, (hipStream_t)stream, wp_c4, (long)N * 32 * 4, gamma, (long)N * N, wq, gq);
    const int n = nchunks * 6 * 64;
    hipLaunchKernelGGL(c4gdn_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, wp_c4, gamma,
                       static_cast<unsigned char *>(astream), N, R, S, ksc, wq, gq);
    STEM_LAUNCH_CHECK("stem_c4gdn_pack");
    return 0;
}

STEM_EXPORT int stem_conv2d_c4_gdn_f16x3(const float *x4, const float *xq, const void *astream, const float *bias, const float *beta, float beta_min,
                                          float *y, int ldy, void *yp, float *yq, int B, int H, int W, int N, int R, int S, int stride, int pad,
                                          void *stream)
{
    STEM_CHECK_ARG(x4 && xq && astream && beta && (y || yp) && (yq || !yp), "stem_conv2d_c4_gdn_f16x3: null pointer (planes come with their scale record)");
    STEM_CHECK_ARG(stem_c4gdn_supported(N, R, S), "stem_conv2d_c4_gdn_f16x3: N must be 64, 128 or 192 and R*S <= 25 (N=%d R=%d S=%d)", N, R, S);
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && stride >= 1 && pad >= 0, "stem_conv2d_c4_gdn_f16x3: bad geometry");
    STEM_CHECK_ARG(!y || (ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0), "stem_conv2d_c4_gdn_f16x3: y rows must be 16-byte aligned, ldy >= N");
    STEM_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0, "stem_conv2d_c4_gdn_f16x3: bias must be 16-byte aligned");
    STEM_CHECK_ARG(((uintptr_t)beta & 15) == 0 && ((uintptr_t)astream & 15) == 0, "stem_conv2d_c4_gdn_f16x3: beta / stream must be 16-byte aligned");
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    STEM_CHECK_ARG(OH >= 1 && OW >= 1, "stem_conv2d_c4_gdn_f16x3: empty output");
    const size_t xb = (size_t)B * H * W * 16;
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && (size_t)B * OH * OW < 0x7FFFFFFFull, "stem_conv2d_c4_gdn_f16x3: image batch must stay below 2 GiB");
    STEM_CHECK_ARG((!yp || (size_t)B * OH * OW * (N / 32) * 128 < 0x7FFF0000ull) && (!y || (size_t)B * OH * OW * ldy * 4 < 0x7FFF0000ull),
                   "stem_conv2d_c4_gdn_f16x3: outputs are addressed through 2 GiB buffer views (split the batch)");
    C4gArgs a;
    memset(&a, 0, sizeof(a));
    a.x4 = x4; a.xq = xq; a.yq = yq; a.astream = static_cast<const unsigned char *>(astream);
    a.wq = reinterpret_cast<const float *>(a.astream + (size_t)(conv_ksteps(R, S) + 2 * (N / 32)) * CHUNK);
    a.gq = a.wq + QREC_HDR + QSLOTS; a.bias = bias; a.beta = beta; a.y = y; a.yp = yp; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.N = N; a.OH = OH; a.OW = OW; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.ksc = conv_ksteps(R, S);
    a.xbytes = (int)xb;
    a.beta_bound = (float)sqrt((double)beta_min + 1.4551915228366852e-11);
#ifdef STEM_EXPERIMENTS
    a.ablate = g_c4g_ablate;
#endif
    const int M = B * OH * OW;
    const dim3 grid(cdiv(M, WG_PIX)), block(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
    if (N == 192)
        hipLaunchKernelGGL(c4gdn_f16x3_kernel<6>, grid, block, 2 * CHUNK, st, a);
    else if (N == 128)
        hipLaunchKernelGGL(c4gdn_f16x3_kernel<4>, grid, block, 2 * CHUNK, st, a);
    else
        hipLaunchKernelGGL(c4gdn_f16x3_kernel<2>, grid, block, 2 * CHUNK, st, a);
    STEM_LAUNCH_CHECK("stem_conv2d_c4_gdn_f16x3");
    return 0;
}
