// Image-tile form of the general split-operand convolution (conv_f16x3.hip, second part) for the stride-1 "same" layers of
// the STEM network at training time (TPM spatiotemporalpriors.py:807-813, HE.0 / HD.4 :814-829, the masked context layer
// layers.py:21-47 over its live taps, EPM :832-838) and their input gradients.
//
// Why a second form.  conv_f16x3_gen_kernel fetches, per K chunk (one tap x 32 channels), a 128-pixel x 64-byte x 2-plane
// activation tile AND a 128-row weight tile from L2: 32 KB per 3.1 MF of executed fp16 MFMA.  At the 16x16 latents of a
// 256x256 crop that is ~51 GB/s per CU at the chip's sustained matrix rate against the 66-73 GB/s the L2 delivers to a CU,
// so operand delivery paces the kernel (0.33 matrix-pipe utilisation on TPM.4, profiles/r03i_pmc_gen_tpm4.csv).  Here a
// workgroup owns a 16x16 block of output pixels of ONE image and all of it happens im2col-free inside LDS:
//   * the input HALO of the block, (16 + KS - 1)^2 pixels of one 32-channel slab, is staged ONCE per slab (51 KB at 5x5)
//     and every tap reads it at a shifted row address -- the 25 taps of a 5x5 layer re-use it 25 times;
//   * the weight tile of a chunk (128 output channels x 32 input channels x 2 planes = 16 KB, stored by the pack kernels as
//     the exact LDS image) arrives by LDS-DMA (buffer_load ... lds) through a three-slot ring, two chunks in flight, no
//     registers and no ds_write on its way;
//   * 8 wavefronts as 4 (M) x 2 (N), each 64 pixels (4 image rows) x 64 channels = four 32x32 accumulators: 24 MFMAs per
//     16 ds_read_b128 per chunk.
// L2 -> LDS traffic per executed flop: 451 KB per 157 MF for a 5x5 slab against 800 KB per 79 MF in the 128x128 form (7x
// less); the 1x1 layers (no halo to re-use) still halve the weight traffic per flop.
//
// LDS images.  Halo: [plane][halo pixel h = hy * HP + hx][64 B], 16-byte pieces XOR-swizzled by (h >> 2) & 3.  The MFMA row of
// lane lr (0..31) is pixel (row 2 mi + (lr >> 4), column lr < 16 ? lr : (lr - HP) & 15) of the wavefront's four image rows:
// the rotation of the odd rows makes h == lr (mod 16) for every lane whatever the halo pitch HP and whatever the tap's shift,
// which is the condition under which the 16-lane groups of ds_read_b128 touch 16 different bank columns (the argument of
// conv_f16x3.hip's 64-byte-row image).  Weights: [plane][128 rows][64 B] with the same swizzle, as packed.
//
// Loop: chunks q = slab * T + tap of this workgroup's split [q_begin, q_end); per chunk one barrier.  Weight chunk c + 2 is
// issued at the top of chunk c into the slot chunk c - 1 has just released; the halo of slab s + 2 is fetched to registers in
// the last chunk of slab s and stored in the first chunk of slab s + 1 into the buffer slab s has released (so a 1x1 layer,
// whose slabs last one chunk, still has a chunk of latency cover).  vmcnt is counted by hand: LDS-DMA and register loads
// retire in issue order, and the two DMA instructions of chunk c + 1 are always the youngest at the top of chunk c.
//
// Epilogue, split-K (sc1 partial tiles, ticket, last arriver sums in split order), scale records: conv_f16x3_gen_kernel's.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "stem_common.h"

namespace {

typedef hp8 h16x8;

constexpr int KC = 32, NPL = 2, SLAB = NPL * 64;
constexpr int IBN = 128;                                   // output channels per workgroup
constexpr int IB_PLANE = IBN * 64, IB_BUF = NPL * IB_PLANE; // 16384 B per weight chunk
constexpr int TS = 16, TPX = TS * TS;                      // 16 x 16 output pixels per workgroup
constexpr int HPMAX = TS + 4, NHMAX = HPMAX * HPMAX;       // halo at 5x5
constexpr int IA_PLANE = NHMAX * 64, IA_BUF = NPL * IA_PLANE;       // 25600 / 51200
constexpr int NRING = 3;
constexpr int ITP = IBN + 4;                               // fp32 pitch of the epilogue tile
constexpr int ILDS_MAIN = 2 * IA_BUF + NRING * IB_BUF;     // 151552
static_assert(TPX * ITP * 4 <= ILDS_MAIN, "the epilogue tile re-uses the main loop's LDS");
constexpr int ILDS = ILDS_MAIN + 32 * 4;                   // + reduction scratch and the last-arriver flag
constexpr int NTHR = 512;                                  // threads
constexpr int OOR = 0x7FFFFF00;
enum { EPI_BIAS = 0, EPI_LRELU = 1, EPI_DACT = 2 };

struct ImgArgs {
    const void *xp, *wp;
    const float *xq, *wq;
    float *yq;
    const float *bias;
    float *y;
    void *yp;
    int ldy;
    int B, H, W, C, N, ntaps;
    int xbytes, wbytes;
    const float *z;
    int ldz, epi;
    float slope;
    float *ws;
    int *cnt;
    int nsplit, cps;
    int xpix;
    int tiles_x, tiles_y;
    unsigned long long *stamps;     // phase stamps of every workgroup (libstem_hip_exper.so only; null otherwise)
    int ny;                         // N tiles of the launch (cdiv(N, IBN))
    int ablate;                     // libstem_hip_exper.so only (WRONG results: timing ablations): 1 = every second chunk's barrier left out, 2 = return after the
                                    // main loop, 3 = one-dimensional launch with the splits of a tile on one XCD and partial tiles without sc1
};

#ifndef STEM_IMG_ZFLIP
#define STEM_IMG_ZFLIP 0        // experiments: the K splits dealt to the workgroups in reverse order
#endif
#ifndef STEM_IMG_LAZY_LAST
#define STEM_IMG_LAZY_LAST 0    // 1: the last ARRIVING split keeps its tile in LDS instead of writing and re-reading it (see the split-K hand-over).
#endif                          // Built, bit-identical, and measured SLOWER in the step (round 6, profiles/r06_ab_img_lazy_last.log): off
#ifdef STEM_EXPERIMENTS
unsigned long long *g_img_stamps = nullptr;
int g_img_ablate = 0;
__device__ unsigned long long g_img_waits[4];      // sums over wavefronts: cycles waiting for vmcnt, at the barrier, in the loop
#define IMG_STAMP(i)                                                                                                              \
    do {                                                                                                                          \
        if (a.stamps && threadIdx.x == 0)                                                                                         \
            a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define IMG_STAMP(i) do { } while (0)
#endif

// block_max on a caller-provided LDS scratch (the kernel keeps ALL its LDS in the one dynamic array: a second __shared__ object
// beside an LDS-DMA ring makes hipcc drain the ring before every LDS read)
__device__ inline float block_max512(float v, float *red)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NTHR / 64; ++w) m = fmaxf(m, red[w]);
    return m;
}
__device__ inline float q_amax512(const float *q, float *red)
{
    int ns = q_nslots(q);
    ns = ns < 0 ? 0 : (ns > (1 << 20) ? (1 << 20) : ns);
    float m = 0.f;
    for (int i = threadIdx.x; i < ns; i += NTHR) m = fmaxf(m, q[QREC_HDR + i]);
    return block_max512(m, red);
}

template <int KS>
__global__ __launch_bounds__(NTHR) void conv_f16x3_img_kernel(const ImgArgs a)
{
    static_assert(KS == 1 || KS == 3 || KS == 5, "square odd windows up to 5x5");
    constexpr int HP = TS + KS - 1, NH = HP * HP, PAD = KS / 2;
    constexpr int NAI = (NH * 4 + NTHR - 1) / NTHR;              // 16-byte pieces of one halo plane per thread: 2 / 3 / 4
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *red = reinterpret_cast<float *>(smem + ILDS_MAIN);      // 24 floats
    int *flag = reinterpret_cast<int *>(smem + ILDS_MAIN + 96);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (tile, N tile, K split) of this workgroup: the launch's three grid dimensions
    int BX = blockIdx.x, BY = blockIdx.y, BZ = blockIdx.z, GX = gridDim.x, GY = gridDim.y, GZ = gridDim.z;
#ifdef STEM_EXPERIMENTS
    if (a.ablate == 3) {
        // ablation (round 6, timing only): a one-dimensional launch in which the splits of one tile run on ONE XCD (workgroup i is
        // dispatched to XCD i mod 8): linear id L -> XCD L & 7, slot L >> 3 -> tile (slot / nsplit) * 8 + XCD, split slot % nsplit;
        // the partial tiles are then stored and read WITHOUT sc1 (they stay in that XCD's L2).  Results are not guaranteed.
        const int L = blockIdx.x, slot = L >> 3, T = (slot / a.nsplit) * 8 + (L & 7);
        GX = a.B * a.tiles_x * a.tiles_y; GY = a.ny; GZ = a.nsplit;
        if (T >= GX * GY) return;
        BX = T % GX; BY = T / GX; BZ = slot % a.nsplit;
    }
#endif
    IMG_STAMP(0);
    const int tpi = a.tiles_x * a.tiles_y;
    const int bimg = BX / tpi, trem = BX - bimg * tpi, tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int y0 = tyi * TS, x0 = txi * TS;
    const int bn0 = BY * IBN, zsplit = STEM_IMG_ZFLIP ? GZ - 1 - BZ : BZ;
    const int nslab = a.C / KC, T = a.ntaps, nchunks = T * nslab;
    const int q_begin = zsplit * a.cps;
    const int q_end = q_begin + a.cps < nchunks ? q_begin + a.cps : nchunks;
    const int wbase = BY * nchunks;
    const int Mtot = a.B * a.H * a.W;

    // ---- halo staging: piece idx = tid + i * 512 -> (halo pixel h, 16-byte piece p), both planes -----------------------------
    int aoff[NAI], adst[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int idx = tid + i * NTHR, h = idx >> 2, p = idx & 3;
        const bool valid = h < NH;
        const int hy = h / HP, hx = h - hy * HP, iy = y0 - PAD + hy, ix = x0 - PAD + hx;
        const bool ok = valid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        aoff[i] = ok ? ((bimg * a.H + iy) * a.W + ix) * a.xpix + p * 16 : OOR;
        adst[i] = valid ? h * 64 + ((p ^ ((h >> 2) & 3)) << 4) : -1;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.xp), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wp), 0, a.wbytes, 0x00020000);
    f32x4 ra[NAI][NPL];
    auto gloadA = [&](int slab) {
        const int so = slab * SLAB;
#pragma unroll
        for (int i = 0; i < NAI; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ra[i][pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, aoff[i] + pl * 64, so, 0));
    };
    auto storeA = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NAI; ++i)
            if (adst[i] >= 0) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<f32x4 *>(smem + buf * IA_BUF + pl * IA_PLANE + adst[i]) = ra[i][pl];
            }
    };
    // weight chunk q -> ring slot: wavefront w copies bytes [2048 w, 2048 w + 2048) as two 1 KiB LDS-DMA instructions
    const int ring_v = wave * 2048 + lane * 16;
    auto dmaB = [&](int q, int slot) {
        auto *ls = (__attribute__((address_space(3))) void *)(smem + 2 * IA_BUF + slot * IB_BUF + wave * 2048);
        const int so = (wbase + q) * IB_BUF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, ls, 16, ring_v, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, ls, 16, ring_v, so, 1024, 0);
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int xl = lr < 16 ? lr : ((lr - HP) & 15);
    const int hb0 = (4 * wm + (lr >> 4)) * HP + xl;                   // halo pixel of tap (0, 0) for MFMA tile mi = 0; mi = 1: + 2 HP
    const int swB = (lr >> 2) & 3;
    const int rdB = 2 * IA_BUF + (wn * 64 + lr) * 64;
    const int pkB0 = ((0 + lh) ^ swB) << 4;                             // k-step 1: this address ^ 32
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
    // a wavefront whose 64 columns lie beyond N only stages and synchronises (wave-uniform: a scalar branch, not an exec mask)
    const bool dead = bn0 + __builtin_amdgcn_readfirstlane(wn) * 64 >= a.N;

    // Fragment addresses of a chunk's first 16-channel step: halo rows of the two MFMA tiles (buffer ab, tap offset toff) and the
    // weight rows (ring slot); the second step's addresses are these ^ 32 (the k piece index 2 ks + lh enters through a XOR)
    auto addrA = [&](int ab, int toff, int &v0, int &v1) {
        const int h0 = hb0 + toff, h1 = h0 + 2 * HP;
        v0 = ab * IA_BUF + h0 * 64 + ((lh ^ ((h0 >> 2) & 3)) << 4);
        v1 = ab * IA_BUF + h1 * 64 + ((lh ^ ((h1 >> 2) & 3)) << 4);
    };
    auto addrB = [&](int slot) { return rdB + slot * IB_BUF + pkB0; };
    auto rdfrag = [&](int vA0, int vA1, int vB, h16x8 (&af)[2][NPL], h16x8 (&bf)[2][NPL]) {        // 8 ds_read_b128
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            af[0][pl] = *reinterpret_cast<const h16x8 *>(smem + vA0 + pl * IA_PLANE);
            af[1][pl] = *reinterpret_cast<const h16x8 *>(smem + vA1 + pl * IA_PLANE);
        }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[nj][pl] = *reinterpret_cast<const h16x8 *>(smem + vB + pl * IB_PLANE + nj * 32 * 64);
    };
    auto mma = [&](const h16x8 (&af)[2][NPL], const h16x8 (&bf)[2][NPL]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {          // the three products of an fp32 product, smallest terms first
                acc[mi][nj] = STEM_MFMA16(af[mi][1], bf[nj][0], acc[mi][nj]);
                acc[mi][nj] = STEM_MFMA16(af[mi][0], bf[nj][1], acc[mi][nj]);
                acc[mi][nj] = STEM_MFMA16(af[mi][0], bf[nj][0], acc[mi][nj]);
            }
    };

    // ---- prologue: first halo, first two weight chunks, the next slab's halo into registers ----------------------------------
    int c = q_begin;
    const int s0 = q_begin / T;
    int slab = s0, t = q_begin - slab * T, ts = t % KS, toff = (t / KS) * HP + ts;
    const int s_last = q_begin < q_end ? (q_end - 1) / T : slab;
    // `pend`: the slab whose halo sits in the staging registers (-1: none); it is stored into buffer (pend - s0) & 1 as soon as
    // slab pend - 2, the previous user of that buffer, is finished
    int pend = -1;
    if (q_begin < q_end) {
        gloadA(slab);
        dmaB(c, 0);
        if (c + 1 < q_end) dmaB(c + 1, 1);
    }
    IMG_STAMP(1);
    // scales of the epilogue, computed while those loads are in flight (conv_f16x3_gen_kernel); the thread's four output
    // columns are the same in every pass of the epilogue (512 threads = 16 rows x 32 column groups): its bias values are loaded here
    const float fac = q_inv(a.xq) * q_inv(a.wq);
    const int ecol = bn0 + (tid & 31) * 4;
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && ecol < a.N) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ebias[cc] = a.bias[ecol + cc];          // parameters may sit at any 4-byte offset of a flat buffer
    }
    float oscale = 1.f;
    if (a.yp) {
        // max |x| and max |w| over the slots of the operands' records and max |bias|: the loads of all three go out together,
        // ONE workgroup reduction carries the three maxima (three back-to-back reductions cost 3 us of every launch's start)
        int nsx = q_nslots(a.xq), nsw = q_nslots(a.wq);
        nsx = nsx < 0 ? 0 : (nsx > (1 << 20) ? (1 << 20) : nsx);
        nsw = nsw < 0 ? 0 : (nsw > (1 << 20) ? (1 << 20) : nsw);
        float xmax = 0.f, wmax = 0.f, bm = 0.f;
        for (int i = tid; i < nsx; i += NTHR) xmax = fmaxf(xmax, a.xq[QREC_HDR + i]);
        for (int i = tid; i < nsw; i += NTHR) wmax = fmaxf(wmax, a.wq[QREC_HDR + i]);
        if (a.bias)
            for (int n = tid; n < a.N; n += NTHR) bm = fmaxf(bm, fabsf(a.bias[n]));
        xmax = wave_max(xmax); wmax = wave_max(wmax); bm = wave_max(bm);
        __syncthreads();
        if (lane == 0) {
            red[wave] = xmax; red[8 + wave] = wmax; red[16 + wave] = bm;
        }
        __syncthreads();
        xmax = red[0]; wmax = red[8]; bm = red[16];
#pragma unroll
        for (int w8 = 1; w8 < NTHR / 64; ++w8) {
            xmax = fmaxf(xmax, red[w8]); wmax = fmaxf(wmax, red[8 + w8]); bm = fmaxf(bm, red[16 + w8]);
        }
        const int oe = q_exp((float)(a.C * a.ntaps) * xmax * wmax + bm);
        oscale = q_pow2(oe);
        if (BX == 0 && BY == 0 && tid == 0) a.yq[1] = q_pow2(-oe);
    }
    if (q_begin < q_end) {
        storeA(0);
        if (slab + 1 <= s_last) {
            gloadA(slab + 1);
            pend = slab + 1;
            // the loop stores a pending halo BEHIND a chunk's barrier and reads the next chunk's first fragments in that same chunk:
            // when the very first chunk already ends its slab (a 1x1 layer; a split that starts on a slab's last tap), the second
            // halo has to be in place before the loop's first barrier
            if (t == T - 1) {
                storeA(1);
                if (slab + 2 <= s_last) {
                    gloadA(slab + 2);
                    pend = slab + 2;
                } else {
                    pend = -1;
                }
            }
        }
    }

    // ---- main loop ---------------------------------------------------------------------------------------------------------------
    // Software-pipelined over the 16-channel steps with two fragment sets, and woven: the 8 fragment reads of the NEXT step, the
    // weight DMA and the address arithmetic are issued between the 12 MFMAs of the current step (sched_group_barrier pins the
    // order), so that only the barrier itself sits between two MFMA blocks.  Per chunk c:
    //   X: MFMAs (c, 0)  +  reads (c, 1), tap walk and fragment addresses of chunk c + 1
    //      wait (this wavefront's reads of chunk c done; chunk c + 1's weights landed) + barrier
    //      [rarely: halo store / next halo loads -- once per slab]
    //   Y: MFMAs (c, 1)  +  weight DMA of chunk c + 3 into the slot chunk c released, reads (c + 1, 0)
    // The DMA, the barrier and the look-ahead reads are unconditional (beyond the last chunk they re-fetch the last chunk into a
    // free slot and read fragments nobody multiplies): the two blocks stay single basic blocks and vmcnt(2) always means
    // "everything but the youngest chunk's DMA".
    IMG_STAMP(2);
    // The loop exists twice: for wavefronts with live columns and for the dead ones of a half-empty last N tile (staging and
    // barriers only) -- chosen once, so that the live loop body has no branch between its reads and its MFMAs.
    const int c0_ = c, slab0_ = slab, t0_ = t, ts0_ = ts, toff0_ = toff, pend0_ = pend;
    auto run = [&](auto live_tag) {
        constexpr bool LIVE = decltype(live_tag)::value;
        int c = c0_, slab = slab0_, t = t0_, ts = ts0_, toff = toff0_, pend = pend0_;      // loop state private to this instance
#ifdef STEM_EXPERIMENTS
        unsigned long long t_vm_ = 0, t_bar_ = 0;
        const unsigned long long t_loop0_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        int ab = 0, slot = 0;
        h16x8 fa0[2][NPL], fb0[2][NPL], fa1[2][NPL], fb1[2][NPL];
        int a0 = 0, a1 = 0, b0 = 0;
        if (c >= q_end) return;
        // chunk q_begin visible; chunk q_begin + 2 on its way
        if (c + 1 < q_end)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dmaB(c + 2 < q_end ? c + 2 : q_end - 1, 2);
        if constexpr (LIVE) {
            addrA(0, toff, a0, a1);
            b0 = addrB(0);
            rdfrag(a0, a1, b0, fa0, fb0);
        }
        // nothing of the prologue (scalar argument loads, LDS stores, these reads) stays pending into the loop: with a clean
        // state on both edges of the loop header the compiler counts the loop's own LDS reads exactly
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (; c < q_end; ++c) {
            const bool last = t == T - 1;
            // ---- X ----
            // tap walk to chunk c + 1 (scalar)
            const int nslot = slot == 2 ? 0 : slot + 1;
            const bool rowend = ts + 1 == KS;                      // selects, not branches: the walk sits between two MFMA blocks
            const int nt = last ? 0 : t + 1, nts = (last || rowend) ? 0 : ts + 1;
            const int ntoff = last ? 0 : toff + (rowend ? HP - (KS - 1) : 1);
            const int nslab = slab + (last ? 1 : 0), nab = ab ^ (last ? 1 : 0);
            int na0 = 0, na1 = 0, nb0 = 0;
            if constexpr (LIVE) {
                rdfrag(a0 ^ 32, a1 ^ 32, b0 ^ 32, fa1, fb1);
                addrA(nab, ntoff, na0, na1);
                nb0 = addrB(nslot);
                mma(fa0, fb0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            // this wavefront's reads of chunk c are complete (issued behind the first eight MFMAs of X): its slot and, at a slab's
            // end, its halo may be overwritten behind the barrier -- as a builtin, so that the compiler knows the second fragment set
            // is ready.  Chunk c + 1's weights have landed once all but the youngest two vector-memory operations are done.
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
#ifdef STEM_EXPERIMENTS
            const unsigned long long ts0_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            const unsigned long long ts1_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            // ablation (round 6): what a barrier every TWO chunks could buy at most -- every second barrier left out (the weight ring
            // is then read while it is overwritten: results are wrong, the instruction stream and its timing are the loop's)
            if (!(a.ablate == 1 && (c & 1))) asm volatile("s_barrier" ::: "memory");
            if (a.stamps) {
                const unsigned long long ts2_ = __builtin_amdgcn_s_memtime();
                t_vm_ += ts1_ - ts0_;
                t_bar_ += ts2_ - ts1_;
            }
#else
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
#endif
            if (pend >= 0 && pend - 2 <= (last ? slab : slab - 1)) {
                storeA((pend - s0) & 1);
                if (pend + 1 <= s_last) {
                    gloadA(pend + 1);
                    ++pend;
                } else {
                    pend = -1;
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);     // the halo stores are done before Y's reads are counted
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- Y ----
            dmaB(c + 3 < q_end ? c + 3 : q_end - 1, slot);          // into the slot chunk c has just released
            if constexpr (LIVE) {
                rdfrag(na0, na1, nb0, fa0, fb0);
                mma(fa1, fb1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                // the reads of step (c + 1, 0) went out behind MFMAs 3..10 of Y: landed by now; said so that the compiler's
                // wait-count pass does not drain the NEXT step's reads in front of X's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot; t = nt; ts = nts; toff = ntoff; slab = nslab; ab = nab;
            a0 = na0; a1 = na1; b0 = nb0;
        }
#ifdef STEM_EXPERIMENTS
        if (a.stamps && lane == 0) {        // per wavefront: shader cycles in the loop, waiting for the weight DMA, waiting at the barrier
            unsigned long long *w = a.stamps + ((size_t)(BZ * GY + BY) * GX + BX) * 8;
            if (wave == 0) {                  // where the workgroup ran: XCC_ID [63:60], HW_ID (se / sh / cu) [59:44], shader cycles in the loop [43:0]
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                w[7] = ((unsigned long long)(xcc & 0xF) << 60) | ((unsigned long long)(hw & 0xFFFF) << 44) | ((__builtin_amdgcn_s_memtime() - t_loop0_) & 0xFFFFFFFFFFFull);
            }
            atomicAdd(&g_img_waits[0], t_vm_);
            atomicAdd(&g_img_waits[1], t_bar_);
            atomicAdd(&g_img_waits[2], __builtin_amdgcn_s_memtime() - t_loop0_);
        }
#endif
    };
    if (dead)
        run(std::false_type{});
    else
        run(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the look-ahead DMA of the last chunks still writes LDS
    __syncthreads();            // every wavefront is done with the operand images: the epilogue tile takes their place
    IMG_STAMP(3);
#ifdef STEM_EXPERIMENTS
    if (a.ablate == 2) return;  // ablation: nothing after the main loop (no partial tiles, ticket, slab read, epilogue): what the tail costs at most
#endif

    // ---- sums of this workgroup -> the epilogue tile Tt (directly, or through the split-K workspace) --------------------------
    // accumulator register r of MFMA tile mi: tile row rho = (r & 3) + 8 (r >> 2) + 4 lh -> pixel (4 wm + 2 mi + (rho >> 4), rotated column)
    auto PIXL = [&](int mi, int r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int py = 4 * wm + 2 * mi + (rho >> 4), px = rho < 16 ? rho : ((rho - HP) & 15);
        return py * TS + px;
    };
    auto MOF = [&](int pix) {       // global output row of tile pixel pix, or -1 outside the image
        const int oy = y0 + (pix >> 4), ox = x0 + (pix & 15);
        return (oy < a.H && ox < a.W) ? (bimg * a.H + oy) * a.W + ox : -1;
    };
    float *Tt = reinterpret_cast<float *>(smem);                   // [256][ITP]
    const int Npad = GY * IBN;
    // every pass below gives a thread the same pieces: column group ec4 = tid & 31 (4 floats), tile rows (tid >> 5) + 16 k
    constexpr int ER = TPX / (NTHR / 32);                          // 16 pieces per thread
    const int ec4 = tid & 31, er0 = tid >> 5;
    f32x4 ev[ER];
    int em[ER];                                                    // global output row of each piece, -1 outside the image
#pragma unroll
    for (int k = 0; k < ER; ++k) em[k] = MOF(er0 + 16 * k);
    if (a.nsplit > 1) {
        // partial tile -> workspace, one 4-byte agent-scope store per accumulator element (each register: two 128-byte row
        // segments), then the arrival protocol of igemm.hip: stores acknowledged, one ticket per workgroup, the last arriver owns
        // the tile and re-zeroes the counter.  NOT 16-byte `buffer_store ... sc1` stores of the tile staged through LDS: with
        // those the ticket overtook the data on the first launch after the workspace was (re)allocated -- the last arriver
        // summed the allocation's zeros for a few pieces (tools/debug/repro_fwd.py: 40 of 40 fresh runs; this form and a
        // plain-store + agent-release form: 0 of 40)
        //
        // Round 6 (STEM_IMG_LAZY_LAST): the workgroup that ARRIVES last does not write and re-read its own tile.  The counter holds two
        // numbers: arrivals (low 16 bits: taken right behind the main loop, before anything is written) and written tiles (high 16
        // bits: added behind the stores' acknowledgement, as the old ticket was).  The last arriver keeps its tile in LDS, waits
        // until the nsplit - 1 others -- which arrived earlier, i.e. are already writing -- have all signalled "written", and sums
        // the tiles IN SPLIT ORDER with its own at its place in that order: the same additions on the same values as before (a
        // store / load round trip is exact), so results do not depend on who arrived last, nor on this switch.  What it saves is
        // the 128-KB write-through + its acknowledgement on the one workgroup every tile waits for (17 us of a 107-us TPM.4 launch
        // lie behind the main loop: profiles/r06_img_tail_ablation.log).  MEASURED (alternating in-step passes, same results bit for
        // bit): 11.46 / 11.54 / 11.52 ms with it against 11.30 / 11.28 / 11.31 without -- the extra agent-scope atomic + barrier
        // right behind the loop is paid by EVERY workgroup, and inside the step the splits of a tile arrive closer together than
        // the 8 us a write takes, so the last arriver waits for the others' writes instead of doing its own.  Default off.
        constexpr int SC1 = 16;
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)((size_t)a.nsplit * Mtot * Npad * 4), 0x00020000);
        const int sstep = Mtot * Npad * 4;
        int eoff[ER];
#pragma unroll
        for (int k = 0; k < ER; ++k) eoff[k] = em[k] >= 0 ? (em[k] * Npad + bn0 + ec4 * 4) * 4 : OOR;
        float *wsp = a.ws + (size_t)zsplit * Mtot * Npad;
        int *cn = a.cnt + BY * GX + BX;
#if STEM_IMG_LAZY_LAST
        if (tid == 0) flag[0] = (__hip_atomic_fetch_add(cn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFF) == a.nsplit - 1;
        __syncthreads();
        const bool lazy = flag[0] != 0;
#else
        const bool lazy = false;
#endif
        if (!lazy) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = MOF(PIXL(mi, r));
                    if (m >= 0) {
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
                        {
#ifdef STEM_EXPERIMENTS
                            if (a.ablate == 3) {
                                wsp[(size_t)m * Npad + bn0 + wn * 64 + nj * 32 + lr] = acc[mi][nj][r];
                                continue;
                            }
#endif
                            __hip_atomic_store(&wsp[(size_t)m * Npad + bn0 + wn * 64 + nj * 32 + lr], acc[mi][nj][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#if STEM_IMG_LAZY_LAST
            if (tid == 0) __hip_atomic_fetch_add(cn, 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // "written"
            IMG_STAMP(4);
            return;
#else
            if (tid == 0) flag[0] = splitk_last_arriver(cn, a.nsplit);
            __syncthreads();
            IMG_STAMP(4);
            if (!flag[0]) return;
#endif
        } else {
            // own tile -> LDS (the layout of the unsplit path); the others' tiles are complete once "written" has reached nsplit - 1.
            // They arrived before this workgroup and need nothing from it: the wait cannot deadlock; it is bounded all the same
            // (~1 s: a launch is over in 100 us) so that a lost workgroup shows up as a wrong result in a test, not as a hung GPU
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = PIXL(mi, r);
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) Tt[pix * ITP + wn * 64 + nj * 32 + lr] = acc[mi][nj][r];
                }
            if (tid == 0) {
                long spins = 0;
                while ((__hip_atomic_load(cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) != a.nsplit - 1 && ++spins < (1L << 22)) __builtin_amdgcn_s_sleep(8);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(cn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            IMG_STAMP(4);
        }
        // the owner sums the slabs in split order (own tile included at its place: one fixed order whoever arrives last), 8 rows x 4
        // splits = 32 sc1 loads in flight per thread: the read is latency-bound (cross-XCD, ~1 us per dependent round)
#pragma unroll
        for (int k = 0; k < ER; ++k) ev[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < ER; kb += 8) {
            for (int sp = 0; sp < a.nsplit; sp += 4) {
                f32x4 tt[8][4];
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool other = sp + u < a.nsplit && !(lazy && sp + u == zsplit);
                        const int so = (other ? sp + u : 0) * sstep;
#ifdef STEM_EXPERIMENTS
                        if (a.ablate == 3) {
                            tt[k][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, other ? eoff[kb + k] : OOR, so, 1));      // sc0: past the L1, from this XCD's L2
                            continue;
                        }
#endif
                        tt[k][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, other ? eoff[kb + k] : OOR, so, SC1));
                    }
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (lazy && sp + u == zsplit)      // this workgroup's own tile, from LDS, at its place in the order
                            ev[kb + k] += *reinterpret_cast<const f32x4 *>(&Tt[(er0 + 16 * (kb + k)) * ITP + ec4 * 4]);
                        else
                            ev[kb + k] += tt[k][u];          // beyond nsplit: zeros (out-of-range loads)
                    }
            }
        }
        IMG_STAMP(5);
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = PIXL(mi, r);
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) Tt[pix * ITP + wn * 64 + nj * 32 + lr] = acc[mi][nj][r];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ER; ++k) ev[k] = *reinterpret_cast<const f32x4 *>(&Tt[(er0 + 16 * k) * ITP + ec4 * 4]);
    }
    // ---- bias / activation, fp32 rows (16 bytes per thread), activated values into Tt for the planes pass -------------------------
    float omax = 0.f;
    const bool colok = ecol < a.N;                                 // N % 4 == 0 (host check)
    f32x4 ez[ER];
    if (a.epi == EPI_DACT) {
#pragma unroll
        for (int k = 0; k < ER; ++k) ez[k] = (colok && em[k] >= 0) ? *reinterpret_cast<const f32x4 *>(a.z + (size_t)em[k] * a.ldz + ecol) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int k = 0; k < ER; ++k) {
        f32x4 v = ev[k] * fac + ebias;
        if (a.epi == EPI_LRELU) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v[cc] = v[cc] > 0.f ? v[cc] : v[cc] * a.slope;
        } else if (a.epi == EPI_DACT) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v[cc] = ez[k][cc] > 0.f ? v[cc] : v[cc] * a.slope;
        }
        if (colok && em[k] >= 0) {
            if (a.y) *reinterpret_cast<f32x4 *>(a.y + (size_t)em[k] * a.ldy + ecol) = v;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        if (a.yp) *reinterpret_cast<f32x4 *>(&Tt[(er0 + 16 * k) * ITP + ec4 * 4]) = v;
    }
    if (a.yq) {
        omax = block_max512(omax, red);
        if (tid == 0) {
            a.yq[QREC_HDR + BY * GX + BX] = omax;
            if (BX == 0 && BY == 0) {
                q_header(a.yq, GX * GY);
                if (!a.yp) a.yq[1] = 1.f;
            }
        }
    }
    if (a.yp) {
        __syncthreads();
        const int oslab = a.N / KC, opix = oslab * SLAB;
        unsigned char *yp = static_cast<unsigned char *>(a.yp);
        const int c8 = tid & 15, pn = bn0 + c8 * 8;                // N % 32 == 0 for planes (host check)
        if (pn < a.N) {
#pragma unroll 4
            for (int k = 0; k < TPX / (NTHR / 16); ++k) {
                const int row = (tid >> 4) + 32 * k, m = MOF(row);
                if (m < 0) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&Tt[row * ITP + c8 * 8]);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&Tt[row * ITP + c8 * 8 + 4]);
                h16x8 h0, h1;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    hp_t x0h, x1h;
                    q_split(v0[cc], oscale, x0h, x1h);
                    h0[cc] = x0h; h1[cc] = x1h;
                    q_split(v1[cc], oscale, x0h, x1h);
                    h0[4 + cc] = x0h; h1[4 + cc] = x1h;
                }
                unsigned char *dst = yp + (size_t)m * opix + (pn >> 5) * SLAB + ((pn >> 3) & 3) * 16;
                *reinterpret_cast<h16x8 *>(dst) = h0;
                *reinterpret_cast<h16x8 *>(dst + 64) = h1;
            }
        }
    }
    IMG_STAMP(6);
}

}   // namespace

#ifdef STEM_EXPERIMENTS
// tools/debug/f16x3_img_phases.py: a device buffer of 8 x uint64 per workgroup receives s_memrealtime (100 MHz) at the phase borders
STEM_EXPORT void stem_exper_img_stamps(void *p) { g_img_stamps = static_cast<unsigned long long *>(p); }
STEM_EXPORT void stem_exper_img_ablate(int mode) { g_img_ablate = mode; }
STEM_EXPORT void stem_exper_img_waits(unsigned long long *out4, int reset)
{
    unsigned long long z[4] = {0, 0, 0, 0};
    if (out4) (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_img_waits), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_img_waits), z, sizeof(z));
}
#endif

// ---- host side: called from stem_conv2d_f16x3_gen_fwd (conv_f16x3.hip) -------------------------------------------------------------
bool stem_fx3_img_eligible(int B, int H, int W, int N, int R, int S, int stride, int pad)
{
    // stem_tuning_set("fx3_gen_img", v): 0 default; 1: the 128-pixel form everywhere; 2: this form for the 1x1 layers too (it runs
    // them correctly, but with no halo to re-use the 128-pixel form's two workgroups per CU are faster: EPM.0 / .2 / .4 forward
    // 42 / 32 / 29 us against 47 / 36 / 30 alone, training step 15.14 against 15.27 ms on one box)
    const int sel = stem_tuning(STEM_TUNE_FX3_GEN_IMG);
    if (sel == 1) return false;
    if (stem_tuning(STEM_TUNE_FX3_GEN_TILE)) return false;         // a forced pixel tile of the 128-pixel form asks for that form
    if (R == 1 && sel != 2) return false;
    if (stride != 1 || R != S || (R != 1 && R != 3 && R != 5) || pad != R / 2) return false;
    const long tiles = (long)cdiv(H, TS) * cdiv(W, TS);
    // at least half of every tile's pixels exist on average (also what keeps the scale record's slots within planes_slots())
    if ((long)H * W * 2 < tiles * TPX) return false;
    if ((long)B * tiles * cdiv(N, IBN) > 16384) return false;       // arrival counters
    // This form exists for launches that cannot fill the chip by pixels (the 16 x 16 latents of the training step: 32 pixel tiles
    // of 128): one workgroup per CU, the halo staged once per channel slab, deep split-K.  A launch whose 128-pixel form already
    // has two workgroups for every CU is left to that form (the variable-rate models' 3 x 3 layers at 64 x 64 .. 256 x 256).  Measured
    // warm on configs[4], B = 16: 992 / 994 ms per GOP iteration with this rule, 994 / 999 without (sel == 3) -- neutral there
    // (profiles/r05_roi_rule_ab_warm.log); sel == 2 (sweeps) skips it.
    if (sel != 2 && sel != 3 && (long)cdiv(B * H * W, 128) * cdiv(N, 128) >= 2 * 256) return false;      // sel == 3: round 4's eligibility (A/B of this rule)
    return true;
}

// Split factor: one workgroup per CU (150 KB of LDS), so a launch runs in rounds of 256 workgroups, each as long as one
// workgroup's chunks plus its prologue / epilogue (~10 chunk times: the halo fill, a 256 x 128 tile through LDS, the slab round trip)
int stem_fx3_img_split(int tiles, int nchunks)
{
    const int forced = stem_tuning(STEM_TUNE_FX3_SPLIT);
    if (forced > 0) return forced < nchunks ? forced : nchunks;
    // stem_tuning_set("fx3_img_w", tenths of a chunk time charged per split): sweeps of the planner inside the training step
    const double w = stem_tuning(STEM_TUNE_FX3_IMG_W) > 0 ? 0.1 * stem_tuning(STEM_TUNE_FX3_IMG_W) : 16.0;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 32 && s * 4 <= nchunks; ++s) {
        const int cps = cdiv(nchunks, s), ns = cdiv(nchunks, cps), rounds = cdiv(tiles * ns, 256);
        const double cost = rounds * (cps + 10.0) + (ns > 1 ? w * ns : 0.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = ns;
        }
    }
    return best;
}

int stem_fx3_img_tiles(int B, int H, int W, int N) { return B * cdiv(H, TS) * cdiv(W, TS) * cdiv(N, IBN); }

int stem_fx3_img_launch(const void *xp, const float *xq, int xpix, int xbytes, const void *wp, const float *wq, int wbytes, const float *bias, int epi,
                        float slope, const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int KS,
                        int T, int split, float *ws, int *cnt, void *stream)
{
    ImgArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.wp = wp; a.xq = xq; a.wq = wq; a.yq = yq; a.bias = bias; a.y = y; a.yp = yp; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.ntaps = T;
    a.xbytes = xbytes; a.wbytes = wbytes; a.z = z; a.ldz = ldz; a.epi = epi; a.slope = slope;
    a.xpix = xpix;
#ifdef STEM_EXPERIMENTS
    a.stamps = g_img_stamps;
    a.ablate = g_img_ablate;
#endif
    a.tiles_x = cdiv(W, TS); a.tiles_y = cdiv(H, TS);
    const int nchunks = (C / 32) * T;
    a.cps = cdiv(nchunks, split);
    a.nsplit = cdiv(nchunks, a.cps);
    if (a.nsplit > 1) {
        a.ws = ws;
        a.cnt = cnt;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_f16x3_img_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ILDS);
        (void)hipFuncSetAttribute((const void *)conv_f16x3_img_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ILDS);
        (void)hipFuncSetAttribute((const void *)conv_f16x3_img_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, ILDS);
        attr_done = true;
    }
    a.ny = cdiv(N, IBN);
    dim3 grid(B * a.tiles_x * a.tiles_y, cdiv(N, IBN), a.nsplit);
#ifdef STEM_EXPERIMENTS
    if (a.ablate == 3 && a.nsplit > 1) grid = dim3(cdiv(B * a.tiles_x * a.tiles_y * a.ny, 8) * 8 * a.nsplit);
    else if (a.ablate == 3) a.ablate = 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (KS == 1)
        hipLaunchKernelGGL((conv_f16x3_img_kernel<1>), grid, dim3(NTHR), ILDS, st, a);
    else if (KS == 3)
        hipLaunchKernelGGL((conv_f16x3_img_kernel<3>), grid, dim3(NTHR), ILDS, st, a);
    else
        hipLaunchKernelGGL((conv_f16x3_img_kernel<5>), grid, dim3(NTHR), ILDS, st, a);
    STEM_LAUNCH_CHECK("stem_conv2d_f16x3_gen_fwd (image-tile form)");
    return 0;
}
