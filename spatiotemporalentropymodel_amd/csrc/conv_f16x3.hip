// fp32-accurate convolutions on the fp16 matrix cores of gfx950 (MI355X): the FROZEN analysis transform (first part of this
// file) and the general kernel for the training-time STEM layers (second part); weight gradients: wgrad_f16x3.hip.
//
// v_mfma_f32_32x32x2_f32 (igemm.hip) is the only MFMA that multiplies fp32 operands, at 64 flop/clk/SIMD; the 16-bit forms
// (v_mfma_f32_32x32x16_f16 / _bf16) run at 1024.  Every fp32 number a is stored as TWO fp16 numbers of a scaled copy,
//     a 2^e = a0 + a1 + r,   a0 = rn16(a 2^e), a1 = rn16(a 2^e - a0),   |r| <= 2^-22 |a| 2^e   (two 11-bit significands)
// so a.b 2^(ea+eb) = a0.b0 + a0.b1 + a1.b0 + O(2^-21 |a.b|): every product of two fp16 numbers is exact in the MFMA's fp32
// accumulator input (11 x 11 bit significands), the dropped a1.b1 is <= 2^-22 |a.b|.  Three fp16 MFMAs (K = 16 each, 32
// cycles) replace eight fp32 MFMAs (K = 2 each, 64 cycles): 96 instead of 512 matrix-pipe cycles per 16 input channels, with
// fp32 accumulation throughout (tests: the 1e-4 gates of the fp32 path; measured 1-2.5e-6 of max|ref| per layer, the fp32-MFMA
// kernels' own distance from fp64).  Round 2 used three bf16 planes and six products (no scaling needed, twice the MFMAs).
// fp16's range is handled by the power-of-two scale 2^e per tensor (scale records, stem_common.h): a split takes e from the
// measured max |a|, a convolution epilogue from the bound K max|x| max|w| + max|bias| (one layer's worth of over-estimate:
// the inputs' maxima are measured), placing the bound in [2^14, 2^15); stored values keep an absolute floor of 2^-25 2^-e
// (fp16 subnormals, kept by the MFMA), i.e. 2^-39 of the bound.
//
// Operands live in HBM pre-split ("planes" layout): per pixel and per 32-channel slab two consecutive rows of 32 fp16,
//     x_planes[pixel][slab][plane 0..1][32]        (128 B per pixel and slab = the fp32 bytes)
// written by the producing kernel's epilogue (or by stem_f16x2_split_nhwc for the first input); the weights are frozen
// (stem/trainSTEM.py:128), split once, and stored chunk by chunk as the exact LDS image the kernel wants (swizzled), so
// that their staging is a straight 24 KiB copy.
//
// Tile: 128 pixels x 192 channels per workgroup, 8 wavefronts as 4 (M) x 2 (N), each 32 x 96 = three 32x32 accumulators;
// K chunk = one tap x 32 input channels = 2 k-steps x 3 tiles x 3 products = 18 MFMAs per wavefront.  LDS rows are 64 B
// (32 fp16) per plane with the 16-byte piece index XOR-swizzled by (row >> 2) & 3: ds_read_b128 / ds_write_b128 are
// conflict-free without padding.  Staging is register-based, two chunks ahead, woven between the MFMA groups as in igemm.hip;
// with three LDS stages (DEPTH 3) the next k-step's fragments are read and the stores issued between the MFMAs of the current one.
// The same products are also instantiated on v_mfma_f32_16x16x32_f16 (MS 16: one k-step per chunk, twice the instructions of
// half the size, a permuted accumulator layout that keeps ds_read_b128 conflict-free on the same LDS image): equal or slower
// alone, faster on a loaded, power-limited chip -- which is where these kernels run.
// The GDN that follows every analysis convolution (gdn.py:52-67) is fused: second contraction over the squared outputs, which
// are scaled by the tile's own maximum, split and parked in LDS as A-operand images; gamma' streams in as a packed 1x1 weight
// image; the same fp16 instruction.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "stem_common.h"

namespace {

typedef hp8 h16x8;

constexpr int BN = 192, KC = 32;
constexpr int NPL = 2, SLAB = NPL * 64;                            // planes per value; bytes per pixel and 32-channel slab
constexpr int B_PLANE = BN * 64, B_BUF = NPL * B_PLANE;            // bytes of one plane of a weight chunk; 24576 per chunk
constexpr int XP = BN + 4;                                         // fp32 pitch of the tile parked for the planes pass
// LDS of a BM-pixel tile: the largest of the main loop (2 x NPL planes x (BM + BN) rows x 64 B), the fused GDN (the squares of six
// 32-channel slabs as A-operand images + one chunk of gamma') and the planes pass (BM x XP floats); then the tap table
constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int lds_stages(int depth) { return depth == 3 ? 3 : 2; }
constexpr int lds_taps(int bm, int depth) { return imax(imax(lds_stages(depth) * NPL * (bm + BN) * 64, 6 * NPL * bm * 64 + NPL * BN * 64), bm * XP * 4); }
constexpr int lds_total(int bm, int depth) { return lds_taps(bm, depth) + 32 * 4; }
constexpr int MAXTAP = 25;
constexpr int OOR = 0x7FFFFF00;                                    // voffset that every buffer view rejects (returns 0)

struct Fx3Phase {
    int ntaps, S;                  // taps of this phase's window (row-major, S per row)
    int dy0, dx0;                  // offset of its first tap on the coarse grid; tap t reads (qy + dy0 + t / S, qx + dx0 + t % S)
    int wofs;                      // byte offset of its weight image inside wp
    int cps, nsplit;               // split-K plan of this phase (blocks with blockIdx.z >= nsplit leave at once)
    int wsofs;                     // float offset of its partial-tile slabs inside ws
    int ooy, oox;                  // py, px
};

struct Fx3Args {
    const void *xp, *wp;
    const float *xq, *wq;          // scale records of the two operands (stem_common.h)
    float *yq;                     // ... of the output: slots always, the scale when planes are written (may be null without planes)
    const float *bias, *beta;
    const void *gp;                // fused GDN: gamma' as the packed weight image of a 1x1 convolution (N x ceil32(N)), its record behind it
    const float *gq;
    float *y;
    void *yp;
    int ldy;
    int B, H, W, C, N, OH, OW, stride, ntaps;
    int S;                         // taps per filter row: tap t = (t / S, t % S) of the R x S window (ntaps may be a prefix)
    int xbytes, wbytes, gbytes;
    float beta_bound;
    int fuse;                      // 0: bias only, 1: GDN
    // general variant (conv_f16x3_gen_kernel): activation epilogue, N tiles, split-K
    const float *z;                // EPI_DACT: the activation output the slope is selected by (z > 0 ? 1 : slope)
    int ldz, epi;                  // epi: 0 bias, 1 bias + leaky ReLU, 2 times d(leaky ReLU)(z)
    float slope;
    float *ws;                     // split-K partial tiles [nsplit][M][gridDim.y * BN]
    int *cnt;                      // one arrival counter per output tile, zero before and after every launch
    int nsplit, cps;               // blockIdx.z = split, chunks [split * cps, (split + 1) * cps)
    int xpix;                      // bytes per pixel of the planes buffer x lives in (a 32-channel-aligned slice of a wider tensor)
    signed char dy[MAXTAP], dx[MAXTAP];
    // transposed face of a stride-2 layer as ONE launch over its four sub-pixel phases (conv_f16x3_gen_kernel<.., .., true>):
    // blockIdx.x = phase * ptiles + pixel tile of the COARSE grid (OH x OW = H x W here); phase (py, px) is a stride-1 convolution
    // of the coarse grid with its own regular window of taps and weight image, written to the fine pixels (2 qy + py, 2 qx + px)
    // of the OHf x OWf output
    int ptiles, btaps, OHf, OWf;   // btaps: the tap count of the output bound (the largest phase's: one scale for all phases)
    Fx3Phase ph[4];
};

__device__ inline int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// Byte offset of tap t = (t / S, t % S) relative to a pixel's own position in the planes.  The kernels walk it in scalar
// registers (TAP_WALK_NEXT): a table lookup in LDS would drain the wavefront's LDS queue in the middle of the woven block.
#define TAP_OFFSET(t) (tw_first + ((t) / tw_S) * tw_line + ((t) % tw_S) * tw_col)
#define TAP_WALK_NEXT()                                                                         \
    do {                                                                                        \
        const bool wrap_ = pf_t + 1 == tw_T, rowend_ = pf_ts + 1 == tw_S;                       \
        ++pf_q;                                                                                 \
        pf_to = wrap_ ? tw_first : pf_to + (rowend_ ? tw_line - (tw_S - 1) * tw_col : tw_col);  \
        pf_ts = (wrap_ || rowend_) ? 0 : pf_ts + 1;                                             \
        pf_t = wrap_ ? 0 : pf_t + 1;                                                            \
        pf_kc += wrap_ ? 1 : 0;                                                                 \
    } while (0)


// BM = pixels per workgroup: 128 (8 wavefronts) or 64 (4 wavefronts, for layers with too few 128-pixel tiles to fill the chip).
// DEPTH 2: two LDS stages, two chunks in flight in registers; the fragments of a 16-channel step are read right before its MFMAs.
// DEPTH 3: three LDS stages -- chunk c + 2 is written while chunk c is multiplied, so the fragments of the NEXT 16-channel step
//          (the first one of chunk c + 1 included) are read and the LDS stores issued between the MFMAs of the current one; only
//          the barrier itself is left between two chunks.  120 KB of LDS with 128-pixel tiles (one workgroup per CU either way).
// MS = rows of the MFMA shape: 32 (v_mfma_f32_32x32x16_f16) or 16 (v_mfma_f32_16x16x32_f16, three-stage loop only): the same
//      operands, LDS traffic and flop in twice as many instructions of half the size, which the chip runs at a higher clock where a
//      launch fills it and sits at the power limit (g_a.2: -6 %); smaller launches lose with it (more issue slots), so it is a
//      second instantiation.  MFMA row (column) m of a 16-block reads image row 4 PI[m >> 2] + (m & 3), PI = 0,2,3,1 -- with the
//      64-byte-row image and its XOR swizzle that makes the 16-lane groups of ds_read_b128 conflict-free -- so lane l holds
//      channels 16 nj + 4 PI[(l & 15) >> 2] + (l & 3) and pixel rows 16 mi + 4 PI[l >> 4] + i of the wavefront's 32 x 96 tile.
__device__ inline int pi4(int x) { return (0x78 >> (2 * x)) & 3; }

template <int BM, int DEPTH, int MS>
__global__ __launch_bounds__(BM * 4) void conv_f16x3_kernel(const Fx3Args a)
{
    static_assert(DEPTH == 2 || DEPTH == 3, "two or three LDS stages");
    static_assert(MS == 32 || (MS == 16 && DEPTH == 3), "the 16-row MFMA shape exists for the three-stage loop");
    constexpr int NSET = 2;                             // register sets (chunks on their way to LDS)
    constexpr int NT = BM * 4;
    constexpr int A_PLANE = BM * 64, A_BUF = NPL * A_PLANE;
    constexpr int PL = NPL;
    constexpr int WPIECES = PL * BN * 4;                // 16-byte pieces of the weight chunk that are read
    constexpr int BP = (WPIECES + NT - 1) / NT;         // ... per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NST = lds_stages(DEPTH);
    unsigned char *As = smem;                       // [NST][NPL][BM][64 B]
    unsigned char *Bs = smem + NST * A_BUF;         // [NST][NPL][BN][64 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Mtot = a.B * a.OH * a.OW;
    int tile_m = blockIdx.x;
    {   // XCD-aware tile order (see igemm.hip): neighbouring pixel tiles share one L2
        const int nb = gridDim.x, qq = nb >> 3, rr = nb & 7, xcd = tile_m & 7, idx = tile_m >> 3;
        if (nb >= 16) tile_m = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int bm0 = tile_m * BM;
    const int nslab = a.C / KC, pixbytes = nslab * SLAB;

    // ---- staging assignment.  Activations: thread -> (row = tid / 4, 16-byte column = tid % 4) of all three planes ---------
    const int srow = tid >> 2, scol = tid & 3;
    int pb;
    unsigned pmask = 0;
    {
        const int m = bm0 + srow;
        const bool ok = m < Mtot;
        const int mm = ok ? m : 0;
        const int ohw = a.OH * a.OW, b = mm / ohw, rem = mm - b * ohw, qy = rem / a.OW, qx = rem - qy * a.OW;
        const int by = qy * a.stride, bx = qx * a.stride;
        pb = ((b * a.H + by) * a.W + bx) * pixbytes + scol * 16;
        for (int t = 0; t < a.ntaps; ++t) {
            const int iy = by + a.dy[t], ix = bx + a.dx[t];
            if (ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) pmask |= 1u << t;
        }
    }
    const int a_st = srow * 64 + ((scol ^ ((srow >> 2) & 3)) << 4);          // LDS byte offset inside a plane
    // weights: the chunk image is copied linearly, 16 bytes per thread and pass; the last pass may be partial
    constexpr int WFULL = WPIECES / NT;                 // full passes
    const int wv0 = tid * 16, wvl = WFULL * NT + tid < WPIECES ? (WFULL * NT + tid) * 16 : OOR;
    const int nchunks = a.ntaps * nslab;
    const int q_last = nchunks - 1;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.xp), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wp), 0, a.wbytes, 0x00020000);

    f32x4 rsa[NSET][PL], rsb[NSET][BP];         // register set s holds a chunk on its way to LDS
    auto gload = [&](int t, int tA, int kc, int q, f32x4 (&ra)[PL], f32x4 (&rb)[BP]) {
        const int sA = kc * SLAB, sB = q * B_BUF;
        const int mk = __builtin_amdgcn_sbfe((int)pmask, (unsigned)t, 1u);                 // 0 / -1: tap t inside the image
        const int off = ((pb + tA) & mk) | (OOR & ~mk);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) ra[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off + pl * 64, sA, 0));
#pragma unroll
        for (int j = 0; j < WFULL; ++j)
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wv0 + j * (NT * 16), sB, 0));
        if (BP > WFULL) rb[BP - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvl, sB, 0));
    };
    auto sstore = [&](int buf, f32x4 (&ra)[PL], f32x4 (&rb)[BP]) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<f32x4 *>(As + buf * A_BUF + pl * A_PLANE + a_st) = ra[pl];
#pragma unroll
        for (int j = 0; j < WFULL; ++j) *reinterpret_cast<f32x4 *>(Bs + buf * B_BUF + j * (NT * 16) + tid * 16) = rb[j];
        if (BP > WFULL && WFULL * NT + tid < WPIECES) *reinterpret_cast<f32x4 *>(Bs + buf * B_BUF + (WFULL * NT + tid) * 16) = rb[BP - 1];
    };

    // ---- wave tile -----------------------------------------------------------------------------------------------------
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 96;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 2) & 3;                                   // (row >> 2) & 3 of every row this lane reads
    const int rdA = (wm0 + lr) * 64, rdB = (wn0 + lr) * 64;
    const int pk0 = ((0 + lh) ^ sw) << 4, pk1 = ((2 + lh) ^ sw) << 4;
    // MS == 16 (see the kernel's header)
    const int l16 = lane & 15, lq = lane >> 4;
    const int cm = 4 * pi4(l16 >> 2) + (l16 & 3), rq = 4 * pi4(lq);
    const int rdA16 = (wm0 + cm) * 64 + ((lq ^ pi4(l16 >> 2)) << 4), rdB16 = (wn0 + cm) * 64 + ((lq ^ pi4(l16 >> 2)) << 4);
    // element (j, r) of this lane's accumulators: local pixel row / local channel inside the workgroup's tile
    auto ROWL = [&](int r) { return MS == 32 ? wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh : wm0 + ((r >> 2) & 1) * 16 + rq + (r & 3); };
    auto COLL = [&](int j, int r) { return MS == 32 ? wn0 + j * 32 + lr : wn0 + j * 32 + (r >> 3) * 16 + cm; };
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // prefetch target of the next step (wave-uniform).  The byte offset of tap t inside the planes is carried along in scalar
    // registers (pf_ts = t % S): a table lookup in LDS would drain the wavefront's LDS queue in the middle of the woven block.
    int pf_q = 0, pf_t = 0, pf_kc = 0, pf_ts = 0, pf_to = 0;
    const int tw_T = a.ntaps, tw_S = a.S, tw_col = pixbytes, tw_line = a.W * pixbytes, tw_first = (a.dy[0] * a.W + a.dx[0]) * pixbytes;
    auto step = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[BP]) {
        const unsigned char *Ab = As + cur * A_BUF + rdA, *Bb = Bs + cur * B_BUF + rdB;
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pk = ks ? pk1 : pk0;
            h16x8 af[PL], bf[PL][3];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) af[pl] = *reinterpret_cast<const h16x8 *>(Ab + pl * A_PLANE + pk);
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int j = 0; j < 3; ++j) bf[pl][j] = *reinterpret_cast<const h16x8 *>(Bb + pl * B_PLANE + j * 32 * 64 + pk);
#pragma unroll
            for (int j = 0; j < 3; ++j) {      // smallest terms first
                acc[j] = STEM_MFMA16(af[1], bf[0][j], acc[j]);
                acc[j] = STEM_MFMA16(af[0], bf[1][j], acc[j]);
                acc[j] = STEM_MFMA16(af[0], bf[0][j], acc[j]);
            }
            if (ks == 0)
                sstore(cur ^ 1, ra, rb);           // the register set holds the next chunk
            else
                gload(t, to, kc, q, ra, rb);       // refill it two chunks ahead
        }
        if (pf_q < q_last) TAP_WALK_NEXT();
    };
    // DEPTH 3: F0 holds the fragments of (stage rd, channels 0..15) on entry and of (stage nx, channels 0..15) on exit
    h16x8 f0a[PL], f0b[PL][3], f1a[PL], f1b[PL][3];
    auto lfrag = [&](int stage, int pk, h16x8 (&af)[PL], h16x8 (&bf)[PL][3]) {       // in the order the MFMAs consume them
        const unsigned char *Ab = As + stage * A_BUF + rdA + pk, *Bb = Bs + stage * B_BUF + rdB + pk;
        af[1] = *reinterpret_cast<const h16x8 *>(Ab + A_PLANE);
        bf[0][0] = *reinterpret_cast<const h16x8 *>(Bb);
        af[0] = *reinterpret_cast<const h16x8 *>(Ab);
        bf[1][0] = *reinterpret_cast<const h16x8 *>(Bb + B_PLANE);
#pragma unroll
        for (int j = 1; j < 3; ++j) {
            bf[0][j] = *reinterpret_cast<const h16x8 *>(Bb + j * 32 * 64);
            bf[1][j] = *reinterpret_cast<const h16x8 *>(Bb + B_PLANE + j * 32 * 64);
        }
    };
    auto mfma9 = [&](h16x8 (&af)[PL], h16x8 (&bf)[PL][3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc[j] = STEM_MFMA16(af[1], bf[0][j], acc[j]);
            acc[j] = STEM_MFMA16(af[0], bf[1][j], acc[j]);
            acc[j] = STEM_MFMA16(af[0], bf[0][j], acc[j]);
        }
    };
    auto step3 = [&](int rd, int nx, int wr, f32x4 (&ra)[PL], f32x4 (&rb)[BP]) {
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
        lfrag(rd, pk1, f1a, f1b);
        mfma9(f0a, f0b);
        sstore(wr, ra, rb);
        lfrag(nx, pk0, f0a, f0b);
        mfma9(f1a, f1b);
        gload(t, to, kc, q, ra, rb);
        // issue order: one LDS read behind each MFMA of the first half; the stores, then the reads, behind those of the second
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int NW = PL + BP;                                  // LDS stores per thread and chunk
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 3) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 2) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 1) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, NW / 4, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, PL + BP, 0);
        if (pf_q < q_last) TAP_WALK_NEXT();
    };

    // MS == 16: a chunk is ONE k-step of the 16x16x32 instruction; its two halves are the channel blocks 0..2 / 3..5 of the
    // wavefront's 96 columns.  ga[s] = pixel fragments of the chunk in flight (two sets: the next chunk's are read during the
    // second half), gb0 / gb1 = weight fragments of the two halves.
    h16x8 ga[2][2][PL], gb0[3][PL], gb1[3][PL];
    f32x4 a16[2][6];
    if constexpr (MS == 16) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 6; ++nj) a16[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto lfragA16 = [&](const unsigned char *base, h16x8 (&A)[2][PL]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) A[mi][pl] = *reinterpret_cast<const h16x8 *>(base + rdA16 + mi * 16 * 64 + pl * A_PLANE);
    };
    auto lfragB16 = [&](const unsigned char *base, int half, h16x8 (&B)[3][PL]) {
#pragma unroll
        for (int n3 = 0; n3 < 3; ++n3)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) B[n3][pl] = *reinterpret_cast<const h16x8 *>(base + rdB16 + (half * 3 + n3) * 16 * 64 + pl * B_PLANE);
    };
    auto mma18 = [&](h16x8 (&A)[2][PL], h16x8 (&B)[3][PL], int half, f32x4 (&d)[2][6]) {
        // smallest terms first; the six accumulators of a product follow each other (independent instructions back to back)
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int n3 = 0; n3 < 3; ++n3)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    d[mi][half * 3 + n3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[mi][prod == 0 ? 1 : 0], B[n3][prod == 1 ? 1 : 0],
                                                                                   d[mi][half * 3 + n3], 0, 0, 0);
    };
    auto step3_16 = [&](int rd, int nx, int wr, f32x4 (&ra)[PL], f32x4 (&rb)[BP], h16x8 (&Acur)[2][PL], h16x8 (&Anext)[2][PL]) {
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
        lfragB16(Bs + rd * B_BUF, 1, gb1);
        mma18(Acur, gb0, 0, a16);
        sstore(wr, ra, rb);
        lfragA16(As + nx * A_BUF, Anext);
        lfragB16(Bs + nx * B_BUF, 0, gb0);
        mma18(Acur, gb1, 1, a16);
        gload(t, to, kc, q, ra, rb);
        // issue order: 18 MFMAs + 6 LDS reads, then 18 MFMAs + the stores + 10 reads, the global loads last
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        constexpr int NW = PL + BP;
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 3) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 2) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (NW + 1) / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, NW / 4, 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, PL + BP, 0);
        if (pf_q < q_last) TAP_WALK_NEXT();
    };

    // chunk q = kc * ntaps + t (channel slab outer, taps inner: the taps of one slab re-touch the same input lines)
    auto chunk_of = [&](int q, int &t, int &kc) {
        q = q < q_last ? q : q_last;
        kc = q / a.ntaps;
        t = q - kc * a.ntaps;
        return q;
    };
    {
        int t, kc, q;
        constexpr int PRE = NST - 1;               // chunks that are in LDS before the loop starts
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            q = chunk_of(c, t, kc);
            gload(t, TAP_OFFSET(t), kc, q, rsa[0], rsb[0]);
            sstore(c, rsa[0], rsb[0]);
        }
#pragma unroll
        for (int s = 0; s < NSET; ++s) {           // set s <- chunk PRE + s
            q = chunk_of(PRE + s, t, kc);
            gload(t, TAP_OFFSET(t), kc, q, rsa[s], rsb[s]);
        }
        pf_q = chunk_of(PRE + NSET, pf_t, pf_kc);
        pf_ts = pf_t % tw_S;
        pf_to = TAP_OFFSET(pf_t);
    }
    // ---- scales of the epilogue, computed HERE: their slot / bias / beta loads and block reductions run while the first chunks
    // are in flight instead of after the last MFMA (one workgroup per CU: nothing else would hide them there)
    __shared__ float qred[16];
    const float fac = q_inv(a.xq) * q_inv(a.wq);                   // the operands were stored times 2^ex, 2^ew
    float oscale = 1.f;                                            // 2^e of the planes output
    if (a.yp) {
        // upper bound of |output| from the operands' maxima: K terms of at most xmax * wmax each, plus the bias; GDN divides by
        // at least sqrt(min beta').  One layer's worth of over-estimation (the inputs' maxima are measured, not bounded).
        const float xmax = q_amax(a.xq, qred), wmax = q_amax(a.wq, qred);
        float bm = 0.f, btm = 3.0e38f;
        for (int n = tid; n < a.N; n += NT) {
            if (a.bias) bm = fmaxf(bm, fabsf(a.bias[n]));
            if (a.fuse) {
                const float bb = fmaxf(a.beta[n], a.beta_bound);
                btm = fminf(btm, bb * bb - 1.4551915228366852e-11f);
            }
        }
        bm = block_max(bm, qred);
        float ob = (float)(a.C * a.ntaps) * xmax * wmax + bm;
        if (a.fuse) ob *= __builtin_amdgcn_rsqf(fmaxf(-block_max(-btm, qred), 1e-30f));
        const int oe = q_exp(ob);
        oscale = q_pow2(oe);
        if (blockIdx.x == 0 && tid == 0) a.yq[1] = q_pow2(-oe);
    }
    __syncthreads();
    if constexpr (MS == 16) {
        int rd = 0, nx = 1, wr = 2;
        lfragA16(As, ga[0]);
        lfragB16(Bs, 0, gb0);
        int q = 0;
        for (; q + 1 < nchunks; q += 2) {
            step3_16(rd, nx, wr, rsa[0], rsb[0], ga[0], ga[1]);
            __syncthreads();
            step3_16(nx, wr, rd, rsa[1], rsb[1], ga[1], ga[0]);
            __syncthreads();
            const int o = rd;
            rd = wr; wr = nx; nx = o;
        }
        if (q < nchunks) {
            step3_16(rd, nx, wr, rsa[0], rsb[0], ga[0], ga[1]);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = a16[(r >> 2) & 1][2 * j + (r >> 3)][r & 3];
    } else if constexpr (DEPTH == 3) {
        int rd = 0, nx = 1, wr = 2;
        lfrag(0, pk0, f0a, f0b);
        int q = 0;
        for (; q + 1 < nchunks; q += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                step3(rd, nx, wr, rsa[s], rsb[s]);
                __syncthreads();
                const int o = rd;
                rd = nx; nx = wr; wr = o;
            }
        }
        if (q < nchunks) {
            step3(rd, nx, wr, rsa[0], rsb[0]);
            __syncthreads();
        }
    } else {
        int q = 0;
        for (; q + 1 < nchunks; q += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                step(s, rsa[s], rsb[s]);
                __syncthreads();
            }
        }
        if (q < nchunks) {
            step(0, rsa[0], rsb[0]);
            __syncthreads();
        }
    }

    // ---- epilogue: element (j, r) of the lane is local pixel row ROWL(r), local channel COLL(j, r) (MS == 32: column
    // wn0 + 32 j + lr, rows (r & 3) + 8 (r >> 2) + 4 lh of each 32x32 tile; MS == 16: two channels per j, selected by r >> 3) ---
    float *X2 = reinterpret_cast<float *>(smem);                   // [BM][XP]
    float omax = 0.f;                                              // max |output| of this thread
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float bias2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = COLL(j, h * 8);
            bias2[h] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * fac + bias2[r >> 3];
        if (a.epi == 1) {          // leaky ReLU (slope 0 = ReLU) of a conv + activation pair (layer-wise models)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] > 0.f ? acc[j][r] : acc[j][r] * a.slope;
        }
    }
    if (a.fuse) {
        // norm[px][i] = beta'[i] + sum_j gamma'[i][j] v[px][j]^2 (gdn.py:52-67): a second contraction, K = N, on the same fp16
        // instruction.  The squares are this workgroup's own data and the contraction runs inside one pixel, so they get a
        // scale of their own from the TILE's maximum (measured, no over-estimate): t = v 2^ev with |t| < 2^7, t^2 < 2^14, split
        // into two fp16 planes and parked in LDS in the A-operand layout of the main loop (one [plane][BM][64 B] image per
        // 32-channel slab); gamma' arrives pre-split as the weight image of a 1x1 convolution (stem_f16x2_pack_conv_weight of the
        // reparametrised matrix, parametrizers.py:42-45; frozen: packed once) and is streamed through one LDS buffer.
        float vm = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) vm = fmaxf(vm, fabsf(acc[j][r]));
        vm = block_max(vm, qred);                          // (its barriers also retire the main loop's last LDS reads)
        const int ev = q_exp(vm) - 8;
        const float vs = q_pow2(ev), vsi = q_pow2(-ev);
        const int nkg = (a.N + KC - 1) / KC;
        unsigned char *Sq = smem;                          // [6][NPL][BM][64 B]
        unsigned char *Bg = smem + 6 * NPL * A_PLANE;      // [NPL][BN][64 B]: one chunk of gamma'
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int slab = (wn0 >> 5) + j;
            if (slab >= nkg) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = ROWL(r), c = COLL(j, r) & 31;                 // channel inside the 32-channel slab
                const float t = acc[j][r] * vs;
                hp_t h0, h1;
                q_split(t * t, 1.f, h0, h1);
                unsigned char *d = Sq + slab * (NPL * A_PLANE) + m * 64 + ((((c >> 3) ^ ((m >> 2) & 3)) << 4) | ((c & 7) << 1));
                *reinterpret_cast<hp_t *>(d) = h0;
                *reinterpret_cast<hp_t *>(d + A_PLANE) = h1;
            }
        }
        f32x16 nrm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[j][r] = 0.f;
        f32x4 n16[2][6];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 6; ++nj) n16[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.gp), 0, a.gbytes, 0x00020000);
        f32x4 gb[BP];
        auto gload_g = [&](int kc) {
#pragma unroll
            for (int u = 0; u < WFULL; ++u) gb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, wv0 + u * (NT * 16), kc * B_BUF, 0));
            if (BP > WFULL) gb[BP - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, wvl, kc * B_BUF, 0));
        };
        gload_g(0);
        for (int kc = 0; kc < nkg; ++kc) {
            __syncthreads();                               // squares written (kc = 0) / previous chunk's fragments read
#pragma unroll
            for (int u = 0; u < WFULL; ++u) *reinterpret_cast<f32x4 *>(Bg + u * (NT * 16) + tid * 16) = gb[u];
            if (BP > WFULL && WFULL * NT + tid < WPIECES) *reinterpret_cast<f32x4 *>(Bg + (WFULL * NT + tid) * 16) = gb[BP - 1];
            __syncthreads();
            if (kc + 1 < nkg) gload_g(kc + 1);
            if constexpr (MS == 16) {
                h16x8 A[2][PL], B0[3][PL], B1[3][PL];
                lfragA16(Sq + kc * (NPL * A_PLANE), A);
                lfragB16(Bg, 0, B0);
                lfragB16(Bg, 1, B1);
                mma18(A, B0, 0, n16);
                mma18(A, B1, 1, n16);
                continue;
            }
            const unsigned char *Ab = Sq + kc * (NPL * A_PLANE) + rdA, *Bb = Bg + rdB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pk = ks ? pk1 : pk0;
                h16x8 af[NPL], bf[NPL][3];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) af[pl] = *reinterpret_cast<const h16x8 *>(Ab + pl * A_PLANE + pk);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int j = 0; j < 3; ++j) bf[pl][j] = *reinterpret_cast<const h16x8 *>(Bb + pl * B_PLANE + j * 32 * 64 + pk);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    nrm[j] = STEM_MFMA16(af[1], bf[0][j], nrm[j]);
                    nrm[j] = STEM_MFMA16(af[0], bf[1][j], nrm[j]);
                    nrm[j] = STEM_MFMA16(af[0], bf[0][j], nrm[j]);
                }
            }
        }
        if constexpr (MS == 16) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[j][r] = n16[(r >> 2) & 1][2 * j + (r >> 3)][r & 3];
        }
        const float gfac = q_inv(a.gq);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float bt2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = COLL(j, h * 8);
                bt2[h] = 1.f;
                if (n < a.N) {
                    const float bb = fmaxf(a.beta[n], a.beta_bound);
                    bt2[h] = bb * bb - 1.4551915228366852e-11f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] *= __builtin_amdgcn_rsqf(nrm[j][r] * gfac * vsi * vsi + bt2[r >> 3]);
        }
        __syncthreads();                                   // the parked squares are free (the planes pass reuses the front of LDS)
    }
    if (a.yq) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (COLL(j, r) < a.N && bm0 + ROWL(r) < Mtot) omax = fmaxf(omax, fabsf(acc[j][r]));
        omax = block_max(omax, qred);
        if (tid == 0) {
            a.yq[QREC_HDR + blockIdx.x] = omax;
            if (blockIdx.x == 0) {
                q_header(a.yq, gridDim.x);
                if (!a.yp) a.yq[1] = 1.f;
            }
        }
    }

    if (a.y) {      // fp32 NHWC output (last layer of the transform)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm0 + ROWL(r), n = COLL(j, r);
                if (m < Mtot && n < a.N) a.y[(size_t)m * a.ldy + n] = acc[j][r];
            }
    }
    if (a.yp) {     // planes output for the next convolution: through LDS so that every thread stores whole 16-byte pieces
        __syncthreads();                                   // the parked tile / the main-loop buffers are free
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) X2[ROWL(r) * XP + COLL(j, r)] = acc[j][r];
        __syncthreads();
        const int oslab = a.N / KC, opix = oslab * SLAB;
        unsigned char *yp = static_cast<unsigned char *>(a.yp);
        for (int e = tid; e < BM * oslab * 4; e += NT) {
            const int row = e / (oslab * 4), rem = e - row * (oslab * 4), sl = rem >> 2, p = rem & 3;
            const int m = bm0 + row;
            if (m >= Mtot) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&X2[row * XP + sl * 32 + p * 8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&X2[row * XP + sl * 32 + p * 8 + 4]);
            h16x8 h0, h1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hp_t x0, x1;
                q_split(v0[c], oscale, x0, x1);
                h0[c] = x0; h1[c] = x1;
                q_split(v1[c], oscale, x0, x1);
                h0[4 + c] = x0; h1[4 + c] = x1;
            }
            unsigned char *dst = yp + (size_t)m * opix + sl * SLAB + p * 16;
            *reinterpret_cast<h16x8 *>(dst) = h0;
            *reinterpret_cast<h16x8 *>(dst + 64) = h1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// General variant for the small-M layers of the STEM network at training time (16x16 latents: 4096 pixels per batch of 16):
// 64 pixels x 128 channels per workgroup (4 wavefronts as 2 x 2, each 32 x 64), N tiles over blockIdx.y, split-K over
// blockIdx.z with the in-kernel last-arriver reduction of igemm.hip (agent-scope partials, integer ticket, fixed summation
// order), 72 KiB of LDS so that two workgroups share a CU.  Epilogue through an LDS tile: bias, leaky ReLU or its derivative
// (dgrad of a conv whose input was activated), fp32 rows with a pitch (the result may be a channel slice of a wider buffer)
// and / or planes for the next layer.  Weight image per (N tile, chunk): [3][128][64 B], swizzled like the 192-row one.
enum { GEN_EPI_BIAS = 0, GEN_EPI_LRELU = 1, GEN_EPI_DACT = 2 };

// ---- the transposed face of a stride-2 layer as four sub-pixel phases (forward of nn.ConvTranspose2d, input gradient of a strided
// nn.Conv2d: spatiotemporalpriors.py:814-829) ---------------------------------------------------------------------------------------
//   out[b, n, oy, ox] = sum over ch, r, s of in[b, ch, iy, ix] * w[ch][n][r][s]   with   oy = 2 iy - pad + r,  ox = 2 ix - pad + s
// An output pixel of parity (py, px) = (oy & 1, ox & 1) only meets the taps r = py + pad (mod 2), s = px + pad (mod 2): on the COARSE
// grid (qy, qx) = (oy >> 1, ox >> 1) phase (py, px) is a stride-1 convolution whose window holds those taps, tap r reading the
// coarse row qy + (py + pad - r) / 2.  Windows are kept in ascending offset order (descending r): position jy = (rmax - r) / 2.
// For 5 x 5, pad 2 the four windows are 3x3, 3x2, 2x3, 2x2 = all 25 taps once: no multiplication by structural zeros.
struct TAxis {
    int cnt, d0, rmax;             // taps of this parity, offset of the first (smallest offset), largest r
};
__host__ __device__ inline TAxis tconv_axis(int R, int pad, int par)
{
    TAxis t;
    const int rmin = (par + pad) & 1;
    t.cnt = rmin < R ? (R - 1 - rmin) / 2 + 1 : 0;
    t.rmax = rmin + 2 * (t.cnt - 1);
    t.d0 = (par + pad - t.rmax) / 2;               // even numerator: exact
    return t;
}
// position of tap (r, s) of the R x S window in the phase-ordered tap sequence [phase 0 | phase 1 | phase 2 | phase 3], phase = 2 py + px
__host__ __device__ inline int tconv_slot(int R, int S, int pad, int r, int s)
{
    const int py = (r + pad) & 1, px = (s + pad) & 1;
    int base = 0;
    for (int p = 0; p < 2 * py + px; ++p) base += tconv_axis(R, pad, p >> 1).cnt * tconv_axis(S, pad, p & 1).cnt;
    const TAxis ay = tconv_axis(R, pad, py), ax = tconv_axis(S, pad, px);
    return base + ((ay.rmax - r) / 2) * ax.cnt + (ax.rmax - s) / 2;
}
constexpr int GBN = 128;
constexpr int GB_PLANE = GBN * 64, GB_BUF = NPL * GB_PLANE;          // 16384
constexpr int GTP = GBN + 4;                                         // fp32 pitch of the epilogue tile
// LDS of a GBM-pixel tile: main loop 2 x (GBM + 128) rows x 64 B x NPL planes, reused by the GBM x 132 float epilogue tile; + tap table
// 16x16x32 by default: alone the STEM layers are 0-8 % slower with it (more issue slots), inside the training step -- a loaded,
// power-limited chip -- the step is 0.2 ms faster
constexpr bool GEN_DEFAULT_MFMA16 = true;
constexpr int glds_main(int gbm) { return 2 * NPL * (gbm + GBN) * 64 > gbm * GTP * 4 ? 2 * NPL * (gbm + GBN) * 64 : gbm * GTP * 4; }
constexpr int glds(int gbm) { return glds_main(gbm) + 32 * 4; }          // 49280 (64 pixels: three workgroups per CU) / 67712 (128: two)

// GBM = pixels per workgroup: 64 (4 wavefronts as 2 x 2) or 128 (8 wavefronts as 4 x 2).  The weight tile of a chunk (16 KB) is
// fetched once per workgroup: with three products per fp32 product the 64-pixel form is bound by that L2 -> LDS traffic
// (24 KB per chunk for 1.6 MF), the 128-pixel form moves 32 KB for twice the work.
// MS = rows of the MFMA shape (see conv_f16x3_kernel): 32, or 16 = v_mfma_f32_16x16x32_f16 with the permuted accumulator layout
// PH: the launch covers the four sub-pixel phases of a transposed face (Fx3Args::ph; host: stem_tconv2d_f16x3_fwd)
template <int GBM, int MS, bool PH = false>
__global__ __launch_bounds__(GBM * 4, 2) void conv_f16x3_gen_kernel(const Fx3Args a)
{
    static_assert(MS == 32 || MS == 16, "MFMA shape");
    int bxt = blockIdx.x, phase = 0;
    if constexpr (PH) {
        phase = blockIdx.x / a.ptiles;
        bxt = blockIdx.x - phase * a.ptiles;
    }
    const int ntaps = PH ? a.ph[phase].ntaps : a.ntaps, tapS = PH ? a.ph[phase].S : a.S;
    const int cps = PH ? a.ph[phase].cps : a.cps, nsplit = PH ? a.ph[phase].nsplit : a.nsplit;
    const int pdy0 = PH ? a.ph[phase].dy0 : (int)a.dy[0], pdx0 = PH ? a.ph[phase].dx0 : (int)a.dx[0];
    const int wofs = PH ? a.ph[phase].wofs : 0;
    if constexpr (PH) {
        if ((int)blockIdx.z >= nsplit) return;        // the whole workgroup: this phase has fewer splits than the launch's z extent
    }
    float *const wsb = PH ? a.ws + a.ph[phase].wsofs : a.ws;
    constexpr int GNT = GBM * 4, GA_PLANE = GBM * 64, GA_BUF = NPL * GA_PLANE;
    constexpr int PL = NPL, BPC = NPL * GBN * 4 / GNT;         // planes; 16-byte weight pieces per thread and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *As = smem;                        // [2][NPL][GBM][64 B]
    unsigned char *Bs = smem + 2 * GA_BUF;           // [2][NPL][128][64 B]
    int *tapi = reinterpret_cast<int *>(smem + glds_main(GBM));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Mtot = a.B * a.OH * a.OW;
    const int bm0 = bxt * GBM, bn0 = blockIdx.y * GBN, zsplit = blockIdx.z;
    const int nslab = a.C / KC, pixbytes = a.xpix;

    const int srow = tid >> 2, scol = tid & 3;
    int pb;
    unsigned pmask = 0;
    {
        const int m = bm0 + srow;
        const bool ok = m < Mtot;
        const int mm = ok ? m : 0;
        const int ohw = a.OH * a.OW, b = mm / ohw, rem = mm - b * ohw, qy = rem / a.OW, qx = rem - qy * a.OW;
        const int by = qy * a.stride, bx = qx * a.stride;
        pb = ((b * a.H + by) * a.W + bx) * pixbytes + scol * 16;
        for (int t = 0; t < ntaps; ++t) {
            const int iy = by + (PH ? pdy0 + t / tapS : (int)a.dy[t]), ix = bx + (PH ? pdx0 + t % tapS : (int)a.dx[t]);
            if (ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) pmask |= 1u << t;
        }
    }
    const int a_st = srow * 64 + ((scol ^ ((srow >> 2) & 3)) << 4);
    const int nchunks = ntaps * nslab;
    const int q_begin = zsplit * cps;
    const int q_end = q_begin + cps < nchunks ? q_begin + cps : nchunks;
    const int q_last = q_end - 1;
    const int wbase = blockIdx.y * nchunks;          // this N tile's chunk sequence inside the packed weights
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.xp), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wp), 0, a.wbytes, 0x00020000);

    f32x4 raA[PL], rbA[BPC], raB[PL], rbB[BPC];
    auto gload = [&](int t, int tA, int kc, int q, f32x4 (&ra)[PL], f32x4 (&rb)[BPC]) {
        const int sA = kc * SLAB, sB = wofs + (wbase + q) * GB_BUF;
        const int mk = __builtin_amdgcn_sbfe((int)pmask, (unsigned)t, 1u);
        const int off = ((pb + tA) & mk) | (OOR & ~mk);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) ra[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off + pl * 64, sA, 0));
#pragma unroll
        for (int j = 0; j < BPC; ++j) rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, tid * 16 + j * (GNT * 16), sB, 0));
    };
    auto sstore = [&](int buf, f32x4 (&ra)[PL], f32x4 (&rb)[BPC]) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<f32x4 *>(As + buf * GA_BUF + pl * GA_PLANE + a_st) = ra[pl];
#pragma unroll
        for (int j = 0; j < BPC; ++j) *reinterpret_cast<f32x4 *>(Bs + buf * GB_BUF + j * (GNT * 16) + tid * 16) = rb[j];
    };

    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 2) & 3;
    const int rdA = (wm0 + lr) * 64, rdB = (wn0 + lr) * 64;
    const int pk0 = ((0 + lh) ^ sw) << 4, pk1 = ((2 + lh) ^ sw) << 4;
    // MS == 16: lane l holds channels 16 nj + cm and pixel rows 16 mi + rq + i of the wavefront's 32 x 64 tile
    const int l16 = lane & 15, lq = lane >> 4;
    const int cm = 4 * pi4(l16 >> 2) + (l16 & 3), rq = 4 * pi4(lq);
    const int rdA16 = (wm0 + cm) * 64 + ((lq ^ pi4(l16 >> 2)) << 4), rdB16 = (wn0 + cm) * 64 + ((lq ^ pi4(l16 >> 2)) << 4);
    auto ROWL = [&](int r) { return MS == 32 ? wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh : wm0 + ((r >> 2) & 1) * 16 + rq + (r & 3); };
    auto COLL = [&](int j, int r) { return MS == 32 ? wn0 + j * 32 + lr : wn0 + j * 32 + (r >> 3) * 16 + cm; };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    int pf_q = 0, pf_t = 0, pf_kc = 0, pf_ts = 0, pf_to = 0;          // see conv_f16x3_kernel
    const int tw_T = ntaps, tw_S = tapS, tw_col = pixbytes, tw_line = a.W * pixbytes, tw_first = (pdy0 * a.W + pdx0) * pixbytes;
    auto step = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[BPC]) {
        const unsigned char *Ab = As + cur * GA_BUF + rdA, *Bb = Bs + cur * GB_BUF + rdB;
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pk = ks ? pk1 : pk0;
            h16x8 af[PL], bf[PL][2];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) af[pl] = *reinterpret_cast<const h16x8 *>(Ab + pl * GA_PLANE + pk);
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[pl][j] = *reinterpret_cast<const h16x8 *>(Bb + pl * GB_PLANE + j * 32 * 64 + pk);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = STEM_MFMA16(af[1], bf[0][j], acc[j]);
                acc[j] = STEM_MFMA16(af[0], bf[1][j], acc[j]);
                acc[j] = STEM_MFMA16(af[0], bf[0][j], acc[j]);
            }
            if (ks == 0)
                sstore(cur ^ 1, ra, rb);
            else
                gload(t, to, kc, q, ra, rb);
        }
        if (pf_q < q_last) TAP_WALK_NEXT();
    };
    // MS == 16: a chunk is one k-step of the 16x16x32 instruction; 2 x 4 accumulators of four registers, the eight independent
    // accumulators of a product back to back
    f32x4 a16[2][4];
    if constexpr (MS == 16) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) a16[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto step16 = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[BPC]) {
        const unsigned char *Ab = As + cur * GA_BUF + rdA16, *Bb = Bs + cur * GB_BUF + rdB16;
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
        h16x8 A[2][PL], B[4][PL];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) A[mi][pl] = *reinterpret_cast<const h16x8 *>(Ab + mi * 16 * 64 + pl * GA_PLANE);
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) B[nj][pl] = *reinterpret_cast<const h16x8 *>(Bb + nj * 16 * 64 + pl * GB_PLANE);
#pragma unroll
        for (int prod = 0; prod < 3; ++prod) {          // smallest terms first
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a16[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[mi][prod == 0 ? 1 : 0], B[nj][prod == 1 ? 1 : 0], a16[mi][nj], 0, 0, 0);
            if (prod == 0) sstore(cur ^ 1, ra, rb);
        }
        gload(t, to, kc, q, ra, rb);
        if (pf_q < q_last) TAP_WALK_NEXT();
    };
    // A wavefront whose 64 columns lie entirely beyond N (the second column pair of the last tile of a 320-channel layer: a
    // sixth of the launch's matrix work) only takes part in the staging: same loads, LDS writes and barriers, no LDS reads and
    // no MFMAs.  The choice is made once per wavefront, OUTSIDE the woven block (a per-sub-tile branch inside it was slower).
    const bool dead = bn0 + wn0 >= a.N;
    auto step_dead = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[BPC]) {
        const int t = __builtin_amdgcn_readfirstlane(pf_t), kc = __builtin_amdgcn_readfirstlane(pf_kc), q = __builtin_amdgcn_readfirstlane(pf_q);
        const int to = __builtin_amdgcn_readfirstlane(pf_to);
        sstore(cur ^ 1, ra, rb);
        gload(t, to, kc, q, ra, rb);
        if (pf_q < q_last) TAP_WALK_NEXT();
    };
    auto chunk_of = [&](int q, int &t, int &kc) {
        q = q < q_last ? q : q_last;
        kc = q / ntaps;
        t = q - kc * ntaps;
        return q;
    };
    if (q_begin < q_end) {
        int t, kc, q;
        q = chunk_of(q_begin, t, kc);
        gload(t, TAP_OFFSET(t), kc, q, raA, rbA);
        sstore(0, raA, rbA);
        q = chunk_of(q_begin + 1, t, kc);
        gload(t, TAP_OFFSET(t), kc, q, raA, rbA);
        q = chunk_of(q_begin + 2, t, kc);
        gload(t, TAP_OFFSET(t), kc, q, raB, rbB);
        pf_q = chunk_of(q_begin + 3, pf_t, pf_kc);
        pf_ts = pf_t % tw_S;
        pf_to = TAP_OFFSET(pf_t);
    }
    // scales of the epilogue, computed while the first chunks are in flight (see conv_f16x3_kernel; every split of a tile
    // computes the same values, the last arriver uses them)
    __shared__ float qred[16];
    const float fac = q_inv(a.xq) * q_inv(a.wq);                   // the operands were stored times 2^ex, 2^ew
    float oscale = 1.f;
    if (a.yp) {     // scale of the planes output from an upper bound of |output| (see conv_f16x3_kernel)
        const float xmax = q_amax(a.xq, qred), wmax = q_amax(a.wq, qred);
        float bm = 0.f;
        if (a.bias)
            for (int n = tid; n < a.N; n += GNT) bm = fmaxf(bm, fabsf(a.bias[n]));
        bm = block_max(bm, qred);
        const int oe = q_exp((float)(a.C * (PH ? a.btaps : ntaps)) * xmax * wmax + bm);
        oscale = q_pow2(oe);
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.yq[1] = q_pow2(-oe);
    }
    __syncthreads();
    if (!dead && MS == 16) {
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step16(0, raA, rbA);
            __syncthreads();
            step16(1, raB, rbB);
            __syncthreads();
        }
        if (q < q_end) {
            step16(0, raA, rbA);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = a16[(r >> 2) & 1][2 * j + (r >> 3)][r & 3];
    } else if (!dead) {
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step(0, raA, rbA);
            __syncthreads();
            step(1, raB, rbB);
            __syncthreads();
        }
        if (q < q_end) {
            step(0, raA, rbA);
            __syncthreads();
        }
    } else {
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step_dead(0, raA, rbA);
            __syncthreads();
            step_dead(1, raB, rbB);
            __syncthreads();
        }
        if (q < q_end) {
            step_dead(0, raA, rbA);
            __syncthreads();
        }
    }

    // ---- sums of this workgroup -> the epilogue tile T (directly, or through the split-K workspace) -----------------------
    float *T = reinterpret_cast<float *>(smem);                     // [GBM][GTP]
    const int Npad = gridDim.y * GBN;
    // every pass of the epilogue gives a thread the same pieces: column group ec4 (4 floats), tile rows er0 + ERS k
    constexpr int ERS = GNT / 32, ER = GBM / ERS;                   // 8 pieces per thread
    const int ec4 = tid & 31, er0 = tid >> 5;
    f32x4 ev[ER];
    // output pixel of tile pixel m: itself, or -- phases -- the fine pixel (2 qy + py, 2 qx + px) of the OHf x OWf output
    auto OROW = [&](int m) -> size_t {
        if constexpr (!PH) return (size_t)m;
        const int ohw = a.OH * a.OW, b = m / ohw, rem = m - b * ohw, qy = rem / a.OW, qx = rem - qy * a.OW;
        return ((size_t)b * a.OHf + 2 * qy + a.ph[phase].ooy) * a.OWf + 2 * qx + a.ph[phase].oox;
    };
    // the tile pixel exists (and -- phases -- its fine pixel lies inside an odd-sized fine grid OHf = 2 OH - 1)
    auto MOK = [&](int m) -> bool {
        if (m >= Mtot) return false;
        if constexpr (!PH) return true;
        const int ohw = a.OH * a.OW, rem = m % ohw, qy = rem / a.OW, qx = rem - qy * a.OW;
        return 2 * qy + a.ph[phase].ooy < a.OHf && 2 * qx + a.ph[phase].oox < a.OWf;
    };
    if (nsplit > 1) {
        float *wsp = wsb + (size_t)zsplit * Mtot * Npad;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm0 + ROWL(r), n = bn0 + COLL(j, r);
                if (m < Mtot) __hip_atomic_store(&wsp[(size_t)m * Npad + n], acc[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // arrival protocol of igemm.hip: partials moved with agent-scope accesses (no device-wide fence), stores acknowledged,
        // one ticket per workgroup; the last arriver owns the tile and re-zeroes the counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int *c = a.cnt + blockIdx.y * gridDim.x + blockIdx.x;
            tapi[0] = splitk_last_arriver(c, nsplit);
        }
        __syncthreads();
        if (!tapi[0]) return;
        constexpr int SC1 = 16;               // sc1: read at the device-coherent level, not this XCD's L2
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(wsb, 0, (int)((size_t)nsplit * Mtot * Npad * 4), 0x00020000);
        const int sstep = Mtot * Npad * 4;
        // the last arriver sums the slabs in split order, 4 pieces x 4 splits in flight (the read is latency-bound: ~1 us
        // per dependent round of cross-XCD sc1 loads; it used to keep 4 loads in flight per thread)
        int eoff[ER];
#pragma unroll
        for (int k = 0; k < ER; ++k) {
            const int m = bm0 + er0 + ERS * k;
            eoff[k] = m < Mtot ? (m * Npad + bn0 + ec4 * 4) * 4 : OOR;
        }
#pragma unroll
        for (int k = 0; k < ER; ++k) ev[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (4 pieces x 4 splits = 16 loads = 64 registers per round: the kernel keeps its two workgroups per CU -- 128 registers)
#pragma unroll
        for (int kb = 0; kb < ER; kb += 4)
            for (int sp = 0; sp < nsplit; sp += 4) {
                f32x4 tt[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        tt[k][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, sp + u < nsplit ? eoff[kb + k] : OOR,
                                                                                                      (sp + u < nsplit ? sp + u : 0) * sstep, SC1));
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u) ev[kb + k] += tt[k][u];          // beyond nsplit: zeros (out-of-range loads)
            }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[ROWL(r) * GTP + COLL(j, r)] = acc[j][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ER; ++k) ev[k] = *reinterpret_cast<const f32x4 *>(&T[(er0 + ERS * k) * GTP + ec4 * 4]);
    }
    float omax = 0.f;
    // ---- bias / activation, fp32 rows (16 bytes per thread), activated values into T for the planes pass.  A thread's four
    // columns are the same in every pass (GNT % 32 == 0): its bias values are loaded once, the z rows of the DACT epilogue all at
    // once (per-row dependent loads cost ~0.5 us each: 8 us of a 26 us launch)
    const int ecol = bn0 + ec4 * 4;
    const bool colok = ecol < a.N;                      // N % 4 == 0 (host check)
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && colok) {               // parameters may sit at any 4-byte offset of a flat buffer: scalar loads
#pragma unroll
        for (int c = 0; c < 4; ++c) ebias[c] = a.bias[ecol + c];
    }
    f32x4 ez[ER];
    if (a.epi == GEN_EPI_DACT) {
#pragma unroll
        for (int k = 0; k < ER; ++k) {
            const int m = bm0 + er0 + ERS * k;
            ez[k] = (colok && MOK(m)) ? *reinterpret_cast<const f32x4 *>(a.z + OROW(m) * a.ldz + ecol) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
#pragma unroll
    for (int k = 0; k < ER; ++k) {
        const int row = er0 + ERS * k, m = bm0 + row;
        f32x4 v = ev[k] * fac + ebias;
        if (a.epi == GEN_EPI_LRELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : v[c] * a.slope;
        } else if (a.epi == GEN_EPI_DACT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = ez[k][c] > 0.f ? v[c] : v[c] * a.slope;
        }
        if (colok && MOK(m)) {
            if (a.y) *reinterpret_cast<f32x4 *>(a.y + OROW(m) * a.ldy + ecol) = v;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        if (a.yp) *reinterpret_cast<f32x4 *>(&T[row * GTP + ec4 * 4]) = v;
    }
    if (a.yq) {
        omax = block_max(omax, qred);
        if (tid == 0) {
            a.yq[QREC_HDR + blockIdx.y * gridDim.x + blockIdx.x] = omax;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                q_header(a.yq, gridDim.x * gridDim.y);
                if (!a.yp) a.yq[1] = 1.f;
            }
        }
    }
    if (a.yp) {
        __syncthreads();
        const int oslab = a.N / KC, opix = oslab * SLAB;
        unsigned char *yp = static_cast<unsigned char *>(a.yp);
        for (int e = tid; e < GBM * (GBN / 8); e += GNT) {
            const int row = e / (GBN / 8), c8 = e - row * (GBN / 8);
            const int m = bm0 + row, n = bn0 + c8 * 8;
            if (!MOK(m) || n >= a.N) continue;              // N % 32 == 0 for planes (host check)
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&T[row * GTP + c8 * 8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&T[row * GTP + c8 * 8 + 4]);
            h16x8 h0, h1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hp_t x0, x1;
                q_split(v0[c], oscale, x0, x1);
                h0[c] = x0; h1[c] = x1;
                q_split(v1[c], oscale, x0, x1);
                h0[4 + c] = x0; h1[4 + c] = x1;
            }
            unsigned char *dst = yp + OROW(m) * opix + (n >> 5) * SLAB + ((n >> 3) & 3) * 16;
            *reinterpret_cast<h16x8 *>(dst) = h0;
            *reinterpret_cast<h16x8 *>(dst + 64) = h1;
        }
    }
}

// weights for conv_f16x3_gen_kernel: [N tile][chunk q = slab * R*S + tap][plane][128 rows][64 B].  flip: the input-gradient of
// a stride-1 convolution is a convolution of dy with w'[c][k][r][s] = w[k][c][R-1-r][S-1-s]: `w` is still the torch weight
// [K][C][R][S], the packed rows are its input channels c (N = C outputs) and the packed channels its output channels k.
// T <= RS: only the first T taps (row-major) are packed -- the live taps of a type-A / type-B masked convolution (layers.py:21-47)
__global__ __launch_bounds__(256) void pack_weight_gen_kernel(const float *w, unsigned char *wp, int N, int C, int RS, int T, int flip, long npieces,
                                                              float *wq)
{
    __shared__ float qred[16];
    const int we = q_exp(q_amax(wq, qred));
    const float wscale = q_pow2(we);
    if (blockIdx.x == 0 && threadIdx.x == 0) wq[1] = q_pow2(-we);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npieces) return;
    const int nchunks = (C / 32) * T;
    const int p = (int)(e & 3), nl = (int)((e >> 2) % GBN);
    const long qq = e / (4 * GBN);                       // ntile * nchunks + q
    const int ntile = (int)(qq / nchunks), q = (int)(qq - (long)ntile * nchunks);
    const int slab = q / T, tap = q - slab * T, n = ntile * GBN + nl;
    h16x8 h[NPL];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = slab * 32 + p * 8 + c;
        float v = 0.f;
        if (n < N) v = flip ? w[((size_t)ch * N + n) * RS + (RS - 1 - tap)] : w[((size_t)n * C + ch) * RS + tap];
        hp_t x0, x1;
        q_split(v, wscale, x0, x1);
        h[0][c] = x0; h[1][c] = x1;
    }
    unsigned char *dst = wp + qq * GB_BUF + nl * 64 + ((p ^ ((nl >> 2) & 3)) << 4);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<h16x8 *>(dst + pl * GB_PLANE) = h[pl];
}

// every layer's weight image with one launch: blockIdx.y = descriptor (the weights of a training model change every step).  The
// table travels by value in the kernel arguments: no staging buffer whose lifetime would have to outlast the queued launch.
constexpr int MAXPACK = 24;
struct PackTable {
    stem_f16x2_pack_desc d[MAXPACK];
};
constexpr int PKR = 8;                                  // output rows per unit
__global__ __launch_bounds__(256) void pack_weight_gen_multi_kernel(const PackTable tab)
{
    // One workgroup per unit = (8 consecutive output rows n, one 32-channel slab), grid-strided.  The 8 x 32 x RS source values
    // of a unit are read in long contiguous runs of the torch weight -- forward role: per row n the slab's 32 x RS values are
    // one run; flip role (rows = input channels of the forward layer): per contraction channel the 8 rows' RS values are one
    // run -- into LDS, then every thread emits whole 16-byte pieces (8 channels of one tap and row, three planes).  (The first
    // version handled one row per unit: 100-byte runs in flip mode and two barriers per 800 values.)
    __shared__ float tile[PKR * 32 * MAXTAP];         // [row][channel in slab][tap]
    const stem_f16x2_pack_desc &d = tab.d[blockIdx.y];
    const int nslab = d.C / 32, ntile = cdiv_dev(d.N, GBN);
    const float *w = static_cast<const float *>(d.w);
    unsigned char *wp = static_cast<unsigned char *>(d.wp);
    float *wq = reinterpret_cast<float *>(wp + (size_t)ntile * nslab * d.R * d.S * GB_BUF);    // the scale record sits behind the FULL image's size
    __shared__ float qred[16];
    float wmax;
    if (d.bmax) {       // maxima of the optimiser pass's chunks that cover this tensor (an upper bound: neighbours may share a chunk)
        float mm = 0.f;
        for (int i = threadIdx.x; i < 4 * d.nb; i += 256) mm = fmaxf(mm, d.bmax[4 * d.b0 + i]);      // four wavefront maxima per chunk
        wmax = block_max(mm, qred);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            q_header(wq, 1);
            wq[QREC_HDR] = wmax;
        }
    } else {
        wmax = q_amax(wq, qred);                                                       // amax_multi_kernel ran
    }
    const int we = q_exp(wmax);
    const float wscale = q_pow2(we);
    if (blockIdx.x == 0 && threadIdx.x == 0) wq[1] = q_pow2(-we);
    float *wz = (d.bmax && d.taps > 0 && d.taps < d.R * d.S) ? static_cast<float *>(const_cast<void *>(d.w)) : nullptr;      // masked taps zeroed here (no maximum pass did it)
    // flip == 2: the four phase images of a transposed face (tconv_slot): transposed indexing like the flip role, taps permuted into
    // phase order, one image per phase behind the other (sum of the phases' taps = RS: the record sits where it always does)
    __shared__ int pslot[MAXTAP], pT[4], pbase[4];
    if (d.flip == 2) {
        if (threadIdx.x < d.R * d.S) pslot[threadIdx.x] = tconv_slot(d.R, d.S, d.R / 2, threadIdx.x / d.S, threadIdx.x % d.S);
        if (threadIdx.x < 4) {
            int b = 0;
            for (int p = 0; p < (int)threadIdx.x; ++p) b += tconv_axis(d.R, d.R / 2, p >> 1).cnt * tconv_axis(d.S, d.S / 2, p & 1).cnt;
            pbase[threadIdx.x] = b;
            pT[threadIdx.x] = tconv_axis(d.R, d.R / 2, threadIdx.x >> 1).cnt * tconv_axis(d.S, d.S / 2, threadIdx.x & 1).cnt;
        }
        __syncthreads();
    }
    const int units = ntile * (GBN / PKR) * nslab;
    // the element loops divide by RS and 32 RS / PKR RS per element: with the window size a compile-time constant (1, 9, 25: every
    // layer of the model) those are multiplications -- the pass is bound by its index arithmetic, not by HBM, otherwise
    auto run = [&](auto rs_const) {
        constexpr int CRS = decltype(rs_const)::value;
        const int RS = CRS > 0 ? CRS : d.R * d.S;
        const int T = d.taps > 0 ? d.taps : RS, nchunks = nslab * T;
        for (int u = blockIdx.x; u < units; u += gridDim.x) {
            const int slab = u % nslab, n0 = (u / nslab) * PKR;   // n runs over the padded rows of all N tiles
            __syncthreads();
            if (d.flip) {
                // element (row r, channel c, tap): w[((slab * 32 + c) * N + n0 + r) * RS + (RS - 1 - tap)]: for fixed c, PKR * RS contiguous floats
                for (int e = threadIdx.x; e < 32 * PKR * RS; e += 256) {
                    const int c = e / (PKR * RS), rem = e - c * (PKR * RS), r = rem / RS, tp = rem - r * RS;
                    const int n = n0 + r;
                    tile[(r * 32 + c) * MAXTAP + (d.flip == 2 ? pslot[tp] : RS - 1 - tp)] = n < d.N ? w[((size_t)(slab * 32 + c) * d.N + n) * RS + tp] : 0.f;
                }
            } else {
                // element (row r, channel c, tap): w[((n0 + r) * C + slab * 32 + c) * RS + tap]: for fixed r, 32 * RS contiguous floats
                for (int e = threadIdx.x; e < PKR * 32 * RS; e += 256) {
                    const int r = e / (32 * RS), rem = e - r * (32 * RS), c = rem / RS, tp = rem - c * RS;
                    const int n = n0 + r;
                    const size_t wi = ((size_t)n * d.C + slab * 32 + c) * RS + tp;
                    float wv = n < d.N ? w[wi] : 0.f;
                    if (wz && tp >= T && n < d.N) {
                        wz[wi] = 0.f;
                        wv = 0.f;
                    }
                    tile[(r * 32 + c) * MAXTAP + tp] = wv;
                }
            }
            __syncthreads();
            for (int e = threadIdx.x; e < PKR * T * 4; e += 256) {
                const int p = e & 3, r = (e >> 2) % PKR, tap = (e >> 2) / PKR;
                const int n = n0 + r, nt = n / GBN, nl = n - nt * GBN;
                h16x8 h[NPL];
    #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    hp_t x0, x1;
                    q_split(tile[(r * 32 + p * 8 + c) * MAXTAP + tap], wscale, x0, x1);
                    h[0][c] = x0; h[1][c] = x1;
                }
                long qq = (long)nt * nchunks + slab * T + tap;
                if (d.flip == 2) {            // image of phase ph: [N tile][slab * T_ph + local tap], behind the earlier phases' images
                    const int ph = tap >= pbase[3] ? 3 : (tap >= pbase[2] ? 2 : (tap >= pbase[1] ? 1 : 0));
                    qq = (long)ntile * nslab * pbase[ph] + (long)nt * (nslab * pT[ph]) + slab * pT[ph] + (tap - pbase[ph]);
                }
                unsigned char *dst = wp + qq * GB_BUF + nl * 64 + ((p ^ ((nl >> 2) & 3)) << 4);
    #pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<h16x8 *>(dst + pl * GB_PLANE) = h[pl];
            }
        }
    };
    const int rs_rt = d.R * d.S;
    if (rs_rt == 25)
        run(std::integral_constant<int, 25>{});
    else if (rs_rt == 9)
        run(std::integral_constant<int, 9>{});
    else if (rs_rt == 1)
        run(std::integral_constant<int, 1>{});
    else
        run(std::integral_constant<int, 0>{});
}

// fp32 NHWC -> planes: one thread per (pixel, slab, 8-channel piece)
__global__ __launch_bounds__(256) void split_nhwc_kernel(const float *x, int ldx, unsigned char *xp, long npieces, int nslab, float *q,
                                                         const float *qsrc)
{
    // qsrc: the record whose slots hold max |x| -- q itself (amax_nhwc_kernel ran on it) or the record of the kernel that produced x
    __shared__ float qred[16];
    const float xmax = q_amax(qsrc, qred);
    const int xe = q_exp(xmax);
    const float xscale = q_pow2(xe);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        q[1] = q_pow2(-xe);
        if (qsrc != q) {
            q_header(q, 1);
            q[QREC_HDR] = xmax;
        }
    }
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npieces) return;
    const long pix = e / (nslab * 4);
    const int rem = (int)(e - pix * (nslab * 4)), sl = rem >> 2, p = rem & 3;
    const float *src = x + pix * ldx + sl * 32 + p * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
    h16x8 h0, h1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hp_t x0, x1;
        q_split(v0[c], xscale, x0, x1);
        h0[c] = x0; h1[c] = x1;
        q_split(v1[c], xscale, x0, x1);
        h0[4 + c] = x0; h1[4 + c] = x1;
    }
    unsigned char *dst = xp + pix * (long)(nslab * SLAB) + sl * SLAB + p * 16;
    *reinterpret_cast<h16x8 *>(dst) = h0;
    *reinterpret_cast<h16x8 *>(dst + 64) = h1;
}

// max |x| of an NHWC tensor (rows of C floats at pitch ldx) -> one slot per workgroup of the record q: the pass in front of a
// split whose input no kernel of this library has measured
__global__ __launch_bounds__(256) void amax_nhwc_kernel(const float *x, int ldx, long npix, int c4n, float *q)
{
    __shared__ float qred[16];
    float m = 0.f;
    const long n4 = npix * c4n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long pix = e / c4n;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * ldx + (e - pix * c4n) * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = block_max(m, qred);
    if (threadIdx.x == 0) {
        q[QREC_HDR + blockIdx.x] = m;
        if (blockIdx.x == 0) q_header(q, gridDim.x);
    }
}

// the same for flat fp32 arrays (weights), blockIdx.y = tensor
struct AmaxTable {
    float *w[24];
    float *q[24];
    long n[24];
    short rs[24], taps[24];        // taps > 0: a masked convolution -- taps >= taps[i] of every filter are ZEROED IN PLACE (what the
                                   // reference does at every forward: layers.py:44 `self.weight.data *= self.mask`) and not counted
};
__global__ __launch_bounds__(256) void amax_multi_kernel(const AmaxTable tab)
{
    __shared__ float qred[16];
    float *w = tab.w[blockIdx.y];
    const long n = tab.n[blockIdx.y];
    float m = 0.f;
    const long stride = (long)gridDim.x * 256;
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (tab.taps[blockIdx.y] > 0) {
        const int rs = tab.rs[blockIdx.y], live = tab.taps[blockIdx.y];
        for (; e < n; e += stride) {
            if ((int)(e % rs) >= live)
                w[e] = 0.f;
            else
                m = fmaxf(m, fabsf(w[e]));
        }
    }
    // eight independent loads in flight per thread (a flat parameter buffer guarantees 4-byte alignment only: scalar loads)
    for (; e + 7 * stride < n; e += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w[e + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
    }
    for (; e < n; e += stride) m = fmaxf(m, fabsf(w[e]));
    m = block_max(m, qred);
    if (threadIdx.x == 0) {
        float *q = tab.q[blockIdx.y];
        q[QREC_HDR + blockIdx.x] = m;
        if (blockIdx.x == 0) q_header(q, gridDim.x);
    }
}

// dy * (z > 0 ? 1 : slope) -> planes: the gradient that reaches a convolution whose output was activated (z = that output),
// split for the fp16 input-gradient / weight-gradient kernels in the pass that applies the leaky-ReLU derivative
__global__ __launch_bounds__(256) void split_dact_nhwc_kernel(const float *x, int ldx, const float *z, int ldz, float slope, unsigned char *xp,
                                                              long npieces, int nslab, float *q, const float *qsrc)
{
    __shared__ float qred[16];
    const float xmax = q_amax(qsrc, qred) * fmaxf(1.f, fabsf(slope));        // slots: max |dy|
    const int xe = q_exp(xmax);
    const float xscale = q_pow2(xe);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        q[1] = q_pow2(-xe);
        if (qsrc != q) {
            q_header(q, 1);
            q[QREC_HDR] = xmax;
        }
    }
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npieces) return;
    const long pix = e / (nslab * 4);
    const int rem = (int)(e - pix * (nslab * 4)), sl = rem >> 2, p = rem & 3;
    const float *src = x + pix * ldx + sl * 32 + p * 8, *zs = z + pix * ldz + sl * 32 + p * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
    const f32x4 z0 = *reinterpret_cast<const f32x4 *>(zs), z1 = *reinterpret_cast<const f32x4 *>(zs + 4);
    h16x8 h0, h1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hp_t x0, x1;
        q_split(z0[c] > 0.f ? v0[c] : v0[c] * slope, xscale, x0, x1);
        h0[c] = x0; h1[c] = x1;
        q_split(z1[c] > 0.f ? v1[c] : v1[c] * slope, xscale, x0, x1);
        h0[4 + c] = x0; h1[4 + c] = x1;
    }
    unsigned char *dst = xp + pix * (long)(nslab * SLAB) + sl * SLAB + p * 16;
    *reinterpret_cast<h16x8 *>(dst) = h0;
    *reinterpret_cast<h16x8 *>(dst + 64) = h1;
}

// planes -> fp32 NHWC (tests / debugging): the three planes add up to the fp32 value exactly
__global__ __launch_bounds__(256) void merge_planes_kernel(const unsigned char *xp, float *x, int ldx, long nelem, int C, const float *q)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nelem) return;
    const long pix = e / C;
    const int c = (int)(e - pix * C), sl = c >> 5, k = c & 31;
    const hp_t *src = reinterpret_cast<const hp_t *>(xp + pix * (long)((C / 32) * SLAB) + sl * SLAB);
    x[pix * ldx + c] = ((float)src[k] + (float)src[32 + k]) * q_inv(q);
}

// torch Conv2d weight [N][C][R][S] fp32 -> per chunk q = slab * R*S + tap the LDS image [plane][192 rows][64 B], piece p of
// row n stored at p ^ ((n >> 2) & 3); rows n >= N are zero
// flip: the operand of the INPUT GRADIENT of a stride-1 convolution whose torch weight is w[C][N][R][S] (rows n = the forward
// layer's input channels, contraction channels = its output channels, taps mirrored)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *w, unsigned char *wp, int N, int C, int RS, long npieces, int flip,
                                                          float *wq)
{
    __shared__ float qred[16];
    const int we = q_exp(q_amax(wq, qred));
    const float wscale = q_pow2(we);
    if (blockIdx.x == 0 && threadIdx.x == 0) wq[1] = q_pow2(-we);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npieces) return;
    const int p = (int)(e & 3), n = (int)((e >> 2) % BN);
    const long q = e / (4 * BN);
    const int slab = (int)(q / RS), tap = (int)(q - (long)slab * RS);
    h16x8 h[NPL];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = slab * 32 + p * 8 + c;
        const float v = n < N ? (flip ? w[((size_t)ch * N + n) * RS + (RS - 1 - tap)] : w[((size_t)n * C + ch) * RS + tap]) : 0.f;
        hp_t x0, x1;
        q_split(v, wscale, x0, x1);
        h[0][c] = x0; h[1][c] = x1;
    }
    unsigned char *dst = wp + q * B_BUF + n * 64 + ((p ^ ((n >> 2) & 3)) << 4);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<h16x8 *>(dst + pl * B_PLANE) = h[pl];
}

}   // namespace

namespace {
constexpr int WQ_SLOTS = 64;                                        // workgroups of the weights' max pass
constexpr size_t WQ_BYTES = (QREC_HDR + WQ_SLOTS) * sizeof(float);  // the record behind every packed weight image
size_t planes_payload(long npix, int C) { return (size_t)npix * (C / 32) * SLAB; }
// one slot per producing workgroup: the smallest output tile of any producer is 64 pixels x 128 channels
long planes_slots(long npix, int C) { return (long)cdivz((size_t)npix, 64) * cdiv(C, 128); }
size_t w_image_bytes(int C, int R, int S) { return (size_t)(C / 32) * R * S * B_BUF; }
size_t gen_image_bytes(int N, int C, int R, int S) { return (size_t)cdiv(N, GBN) * (C / 32) * R * S * GB_BUF; }

int amax_flat(const float *w, long n, float *q, hipStream_t st, int rs = 0, int taps = 0)
{
    AmaxTable t;
    memset(&t, 0, sizeof(t));
    t.w[0] = const_cast<float *>(w); t.q[0] = q; t.n[0] = n; t.rs[0] = (short)rs; t.taps[0] = (short)taps;
    hipLaunchKernelGGL(amax_multi_kernel, dim3(WQ_SLOTS, 1), dim3(256), 0, st, t);      // always all slots: images compare equal
    return 0;
}
}   // namespace

STEM_EXPORT size_t stem_f16x2_planes_qrec_offset(long npix, int C) { return C % 32 ? 0 : planes_payload(npix, C); }

STEM_EXPORT size_t stem_f16x2_planes_bytes(long npix, int C)
{
    return C % 32 ? 0 : planes_payload(npix, C) + (((QREC_HDR + planes_slots(npix, C)) * sizeof(float) + 15) & ~(size_t)15);
}

STEM_EXPORT size_t stem_f16x2_conv_weight_bytes(int C, int R, int S) { return C % 32 ? 0 : w_image_bytes(C, R, S) + WQ_BYTES; }

STEM_EXPORT int stem_amax_nhwc(const float *x, int ldx, long npix, int C, float *q, long max_slots, void *stream)
{
    STEM_CHECK_ARG(x && q && npix >= 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0 && max_slots >= 1,
                   "stem_amax_nhwc: rows of C %% 4 == 0 floats, 16-byte aligned (C=%d ldx=%d)", C, ldx);
    long wg = (long)cdivz((size_t)npix * (C / 4), 2048);
    if (wg < 1) wg = 1;
    if (wg > max_slots) wg = max_slots;
    if (wg > 1024) wg = 1024;
    hipLaunchKernelGGL(amax_nhwc_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, x, ldx, npix, C / 4, q);
    STEM_LAUNCH_CHECK("stem_amax_nhwc");
    return 0;
}

STEM_EXPORT int stem_f16x2_split_nhwc(const float *x, int ldx, void *xp, float *xq, const float *src_q, long npix, int C, void *stream)
{
    STEM_CHECK_ARG(x && xp && xq && npix >= 0 && C > 0 && C % 32 == 0 && ldx >= C && ldx % 4 == 0,
                   "stem_f16x2_split_nhwc: channels must be a multiple of 32 and rows 16-byte aligned (C=%d ldx=%d)", C, ldx);
    const long np = npix * (C / 32) * 4;
    if (np == 0) return 0;
    if (!src_q && stem_amax_nhwc(x, ldx, npix, C, xq, planes_slots(npix, C), stream)) return -2;
    hipLaunchKernelGGL(split_nhwc_kernel, dim3((unsigned)cdivz(np, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       static_cast<unsigned char *>(xp), np, C / 32, xq, src_q ? src_q : xq);
    STEM_LAUNCH_CHECK("stem_f16x2_split_nhwc");
    return 0;
}

STEM_EXPORT int stem_f16x2_split_dact_nhwc(const float *x, int ldx, const float *z, int ldz, float slope, void *xp, float *xq, const float *src_q,
                                            long npix, int C, void *stream)
{
    STEM_CHECK_ARG(x && z && xp && xq && npix >= 0 && C > 0 && C % 32 == 0 && ldx >= C && ldx % 4 == 0 && ldz >= C && ldz % 4 == 0,
                   "stem_f16x2_split_dact_nhwc: channels must be a multiple of 32 and rows 16-byte aligned (C=%d ldx=%d ldz=%d)", C, ldx, ldz);
    const long np = npix * (C / 32) * 4;
    if (np == 0) return 0;
    if (!src_q && stem_amax_nhwc(x, ldx, npix, C, xq, planes_slots(npix, C), stream)) return -2;
    hipLaunchKernelGGL(split_dact_nhwc_kernel, dim3((unsigned)cdivz(np, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, z, ldz, slope,
                       static_cast<unsigned char *>(xp), np, C / 32, xq, src_q ? src_q : xq);
    STEM_LAUNCH_CHECK("stem_f16x2_split_dact_nhwc");
    return 0;
}

STEM_EXPORT int stem_f16x2_merge_nhwc(const void *xp, const float *xq, float *x, int ldx, long npix, int C, void *stream)
{
    STEM_CHECK_ARG(x && xp && xq && npix >= 0 && C > 0 && C % 32 == 0 && ldx >= C, "stem_f16x2_merge_nhwc: bad arguments (C=%d ldx=%d)", C, ldx);
    const long ne = npix * C;
    if (ne == 0) return 0;
    hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)cdivz(ne, 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const unsigned char *>(xp), x, ldx, ne, C, xq);
    STEM_LAUNCH_CHECK("stem_f16x2_merge_nhwc");
    return 0;
}

static int pack_conv_weight(const float *w, void *wp, int N, int C, int R, int S, int flip, void *stream, const char *who)
{
    STEM_CHECK_ARG(w && wp && N >= 1 && N <= BN && C > 0 && C % 32 == 0 && R >= 1 && S >= 1 && R * S <= MAXTAP,
                   "%s: N <= %d, C %% 32 == 0, R*S <= %d (N=%d C=%d R=%d S=%d)", who, BN, MAXTAP, N, C, R, S);
    const long np = (long)(C / 32) * R * S * BN * 4;
    float *wq = reinterpret_cast<float *>(static_cast<unsigned char *>(wp) + w_image_bytes(C, R, S));
    amax_flat(w, (long)N * C * R * S, wq, (hipStream_t)stream);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdivz(np, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<unsigned char *>(wp), N, C, R * S, np, flip, wq);
    STEM_LAUNCH_CHECK(who);
    return 0;
}

STEM_EXPORT int stem_f16x2_pack_conv_weight(const float *w, void *wp, int N, int C, int R, int S, void *stream)
{
    return pack_conv_weight(w, wp, N, C, R, S, 0, stream, "stem_f16x2_pack_conv_weight");
}

STEM_EXPORT int stem_f16x2_pack_conv_weight_flip(const float *w, void *wp, int N, int C, int R, int S, void *stream)
{
    return pack_conv_weight(w, wp, N, C, R, S, 1, stream, "stem_f16x2_pack_conv_weight_flip");
}

static int conv2d_f16x3_launch(const void *xp, const float *xq, const void *wp, const float *bias, const float *beta, const void *gp,
                                float beta_min, int act, float slope, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N,
                                int R, int S, int stride, int pad, void *stream);

STEM_EXPORT int stem_conv2d_f16x3_fwd(const void *xp, const float *xq, const void *wp, const float *bias, const float *beta, const void *gp,
                                       float beta_min, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R, int S,
                                       int stride, int pad, void *stream)
{
    return conv2d_f16x3_launch(xp, xq, wp, bias, beta, gp, beta_min, 0, 0.f, y, ldy, yp, yq, B, H, W, C, N, R, S, stride, pad, stream);
}

STEM_EXPORT int stem_conv2d_f16x3_fwd_act(const void *xp, const float *xq, const void *wp, const float *bias, int act, float slope, float *y, int ldy,
                                           void *yp, float *yq, int B, int H, int W, int C, int N, int R, int S, int stride, int pad, void *stream)
{
    STEM_CHECK_ARG(act == 0 || act == 1, "stem_conv2d_f16x3_fwd_act: act is 0 (none) or 1 (leaky ReLU with `slope`)");
    STEM_CHECK_ARG(!act || fabsf(slope) <= 1.f, "stem_conv2d_f16x3_fwd_act: |slope| <= 1");
    return conv2d_f16x3_launch(xp, xq, wp, bias, nullptr, nullptr, 1e-6f, act, slope, y, ldy, yp, yq, B, H, W, C, N, R, S, stride, pad, stream);
}

static int conv2d_f16x3_launch(const void *xp, const float *xq, const void *wp, const float *bias, const float *beta, const void *gp,
                                float beta_min, int act, float slope, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N,
                                int R, int S, int stride, int pad, void *stream)
{
    STEM_CHECK_ARG(xp && xq && wp && (y || yp) && (yq || !yp), "stem_conv2d_f16x3_fwd: null pointer (planes come with their scale records)");
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 32 && C % 32 == 0 && N >= 1 && N <= BN && R >= 1 && S >= 1 && R * S <= MAXTAP &&
                   stride >= 1 && pad >= 0, "stem_conv2d_f16x3_fwd: C %% 32 == 0, N <= %d, R*S <= %d (C=%d N=%d R=%d S=%d)", BN, MAXTAP, C, N, R, S);
    STEM_CHECK_ARG(!yp || N % 32 == 0, "stem_conv2d_f16x3_fwd: planes output needs N %% 32 == 0 (N=%d)", N);
    STEM_CHECK_ARG(!y || ldy >= N, "stem_conv2d_f16x3_fwd: ldy < N");
    STEM_CHECK_ARG((beta == nullptr) == (gp == nullptr), "stem_conv2d_f16x3_fwd: beta and the packed gamma come together");
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    STEM_CHECK_ARG(OH >= 1 && OW >= 1, "stem_conv2d_f16x3_fwd: empty output");
    const size_t xb = planes_payload((long)B * H * W, C), wb = w_image_bytes(C, R, S);
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && wb < 0x7FFFFF00ull && (size_t)B * OH * OW < 0x7FFFFFFFull,
                   "stem_conv2d_f16x3_fwd: operand views must stay below 2 GiB (split the batch)");
    Fx3Args a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.wp = wp; a.bias = bias; a.beta = beta; a.gp = gp; a.y = y; a.yp = yp; a.ldy = ldy;
    if (gp) {       // gamma' = the packed image of an [N][ceil32(N)] 1x1 weight, its scale record behind it
        const int cpad = cdiv(N, 32) * 32;
        a.gbytes = (int)w_image_bytes(cpad, 1, 1);
        a.gq = reinterpret_cast<const float *>(static_cast<const unsigned char *>(gp) + a.gbytes);
    }
    a.xq = xq; a.yq = yq; a.wq = reinterpret_cast<const float *>(static_cast<const unsigned char *>(wp) + wb);
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.OH = OH; a.OW = OW; a.stride = stride; a.ntaps = R * S; a.S = S;
    a.xbytes = (int)xb; a.wbytes = (int)wb;
    a.fuse = gp ? 1 : 0;
    a.epi = act; a.slope = slope;
    a.beta_bound = (float)sqrt((double)beta_min + 1.4551915228366852e-11);
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            a.dy[r * S + s] = (signed char)(r - pad);
            a.dx[r * S + s] = (signed char)(s - pad);
        }
    static bool attr_done = false;      // > 64 KiB of dynamic LDS needs an explicit opt-in
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<128, 2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(128, 2));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<128, 3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(128, 3));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<128, 3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(128, 3));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<64, 2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(64, 2));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<64, 3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(64, 3));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_kernel<64, 3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total(64, 3));
        attr_done = true;
    }
    const int M = B * OH * OW;
    hipStream_t st = (hipStream_t)stream;
    // 128-pixel tiles (8 wavefronts) when there is one for at least half of the 256 CUs, 64-pixel tiles (4 wavefronts) below
    // that.  Alone, 64-pixel tiles are faster up to 255 tiles (g_a.4 at B=16, 128 tiles: 127 against 140 us); next to other
    // streams' work -- where such a launch runs in the training step -- one workgroup per CU on half the chip beats two small
    // ones per CU everywhere (bench step 15.46-15.54 -> 15.27-15.33 ms), so the threshold follows the loaded machine.
    const int tile = stem_tuning(STEM_TUNE_FX3_TILE);
    const bool small = tile ? tile == 64 : cdiv(M, 128) < 128;
    // main-loop form (see the kernel): stem_tuning_set("fx3_depth", 2 | 3); 0 = default
    // default: three stages with 128-pixel tiles (g_a.2 365 -> 357 us), two with 64-pixel tiles -- 96 KB of LDS would leave room
    // for ONE such workgroup per CU, and layers with 256 .. 511 tiles (g_a.4) run two per CU next to other streams' work
    const int dsel = stem_tuning(STEM_TUNE_FX3_DEPTH);
    const int depth = dsel ? dsel : (small ? 2 : 3);
    // MFMA shape (see the kernel): 16x16x32 where the launch fills the chip with 128-pixel tiles and runs the three-stage loop,
    // 32x32x16 otherwise; stem_tuning_set("fx3_mfma", 16 | 32) forces it where the loop form allows
    const int msel = stem_tuning(STEM_TUNE_FX3_MFMA);
    const bool m16 = depth == 3 && (msel ? msel == 16 : !small);
    if (small && m16)
        hipLaunchKernelGGL((conv_f16x3_kernel<64, 3, 16>), dim3(cdiv(M, 64)), dim3(256), lds_total(64, 3), st, a);
    else if (small && depth == 3)
        hipLaunchKernelGGL((conv_f16x3_kernel<64, 3, 32>), dim3(cdiv(M, 64)), dim3(256), lds_total(64, 3), st, a);
    else if (small)
        hipLaunchKernelGGL((conv_f16x3_kernel<64, 2, 32>), dim3(cdiv(M, 64)), dim3(256), lds_total(64, 2), st, a);
    else if (m16)
        hipLaunchKernelGGL((conv_f16x3_kernel<128, 3, 16>), dim3(cdiv(M, 128)), dim3(512), lds_total(128, 3), st, a);
    else if (depth == 3)
        hipLaunchKernelGGL((conv_f16x3_kernel<128, 3, 32>), dim3(cdiv(M, 128)), dim3(512), lds_total(128, 3), st, a);
    else
        hipLaunchKernelGGL((conv_f16x3_kernel<128, 2, 32>), dim3(cdiv(M, 128)), dim3(512), lds_total(128, 2), st, a);
    STEM_LAUNCH_CHECK("stem_conv2d_f16x3_fwd");
    return 0;
}

// ---- general variant: N tiles of 128, split-K, activation epilogues (training-time STEM layers) --------------------------------
STEM_EXPORT size_t stem_f16x2_conv_weight_gen_bytes(int N, int C, int R, int S)
{
    return C % 32 ? 0 : gen_image_bytes(N, C, R, S) + WQ_BYTES;
}

STEM_EXPORT int stem_f16x2_pack_conv_weight_gen(const float *w, void *wp, int N, int C, int R, int S, int flip, int taps, void *stream)
{
    STEM_CHECK_ARG(w && wp && N >= 1 && C > 0 && C % 32 == 0 && R >= 1 && S >= 1 && R * S <= MAXTAP,
                   "stem_f16x2_pack_conv_weight_gen: C %% 32 == 0, R*S <= %d (N=%d C=%d R=%d S=%d)", MAXTAP, N, C, R, S);
    STEM_CHECK_ARG(taps >= 0 && taps <= R * S && (taps == 0 || !flip), "stem_f16x2_pack_conv_weight_gen: 0 <= taps <= R*S, forward role only (taps=%d)", taps);
    const int T = taps > 0 ? taps : R * S;
    const long np = (long)cdiv(N, GBN) * (C / 32) * T * GBN * 4;
    float *wq = reinterpret_cast<float *>(static_cast<unsigned char *>(wp) + gen_image_bytes(N, C, R, S));
    amax_flat(w, (long)N * C * R * S, wq, (hipStream_t)stream, R * S, taps < R * S ? taps : 0);
    hipLaunchKernelGGL(pack_weight_gen_kernel, dim3((unsigned)cdivz(np, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<unsigned char *>(wp), N, C, R * S, T, flip, np, wq);
    STEM_LAUNCH_CHECK("stem_f16x2_pack_conv_weight_gen");
    return 0;
}

STEM_EXPORT int stem_f16x2_pack_conv_weights_multi(const stem_f16x2_pack_desc *descs_host, int n, void *stream)
{
    STEM_CHECK_ARG(descs_host && n >= 1 && n <= MAXPACK, "stem_f16x2_pack_conv_weights_multi: 1..%d descriptors per call, got %d", MAXPACK, n);
    size_t maxu = 0;
    for (int i = 0; i < n; ++i) {
        const stem_f16x2_pack_desc &d = descs_host[i];
        STEM_CHECK_ARG(d.w && d.wp && d.N >= 1 && d.C > 0 && d.C % 32 == 0 && d.R >= 1 && d.S >= 1 && d.R * d.S <= MAXTAP,
                       "stem_f16x2_pack_conv_weights_multi: descriptor %d: C %% 32 == 0, R*S <= %d (N=%d C=%d R=%d S=%d)", i, MAXTAP, d.N, d.C, d.R, d.S);
        const size_t nu = (size_t)cdiv(d.N, GBN) * (GBN / PKR) * (d.C / 32);
        if (nu > maxu) maxu = nu;
    }
    PackTable tab;
    memset(&tab, 0, sizeof(tab));
    memcpy(tab.d, descs_host, n * sizeof(stem_f16x2_pack_desc));
    AmaxTable mt;                           // max |w| of every tensor first: the scale its image is stored with
    memset(&mt, 0, sizeof(mt));
    for (int i = 0; i < n; ++i) {
        const stem_f16x2_pack_desc &d = descs_host[i];
        STEM_CHECK_ARG(d.taps >= 0 && d.taps <= d.R * d.S && (d.taps == 0 || !d.flip), "stem_f16x2_pack_conv_weights_multi: descriptor %d: taps", i);
        STEM_CHECK_ARG(d.flip >= 0 && d.flip <= 2 && (d.flip != 2 || (d.R == d.S && (d.R & 1))),
                       "stem_f16x2_pack_conv_weights_multi: descriptor %d: flip 0 | 1 | 2 (2 = phase images of a transposed face: odd square windows)", i);
        mt.w[i] = static_cast<float *>(const_cast<void *>(d.w));
        mt.q[i] = reinterpret_cast<float *>(static_cast<unsigned char *>(d.wp) + gen_image_bytes(d.N, d.C, d.R, d.S));
        mt.n[i] = (long)d.N * d.C * d.R * d.S;
        mt.rs[i] = (short)(d.R * d.S);
        mt.taps[i] = (short)(d.taps > 0 && d.taps < d.R * d.S ? d.taps : 0);
    }
    bool all_bmax = true;
    for (int i = 0; i < n; ++i) all_bmax = all_bmax && descs_host[i].bmax != nullptr && descs_host[i].nb >= 1;
    if (!all_bmax) {
        for (int i = 0; i < n; ++i) tab.d[i].bmax = nullptr;        // one source of the maximum per call
        hipLaunchKernelGGL(amax_multi_kernel, dim3(WQ_SLOTS, n), dim3(256), 0, (hipStream_t)stream, mt);
    }
    const unsigned gx = (unsigned)(maxu < 2048 ? maxu : 2048);
    hipLaunchKernelGGL(pack_weight_gen_multi_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, tab);
    STEM_LAUNCH_CHECK("stem_f16x2_pack_conv_weights_multi");
    return 0;
}

namespace {
// ---- both weight images of a layer from ONE read of its weights (round 5) ---------------------------------------------------------
// The training step re-packs every weight after every optimiser step (stem/trainSTEM.py:213: the weights changed), once per ROLE:
// the forward image and the input-gradient image of a layer are two transposes of the same numbers.  pack_weight_gen_multi_kernel
// makes one image per descriptor: 72 MB read + 72 MB written per role, 82 + 108 us per P-frame step, the second on the
// weight-gradient stream where it competes with the forward's kernels for HBM.  Here a workgroup takes a 32 x 32 x RS tile
// [a][b][tap] of the torch tensor [A][B][R][S] into LDS once and emits the pieces of BOTH images from it: rows = the tile's a
// index with 8 consecutive b per 16-byte piece (a Conv2d's forward image, a ConvTranspose2d's input-gradient image), or rows = b
// with 8 consecutive a (transposed indexing: flip / phases), taps in identity, mirrored or phase order.
constexpr int PAIR_T = 32;                              // tile edge (channels)
constexpr int PAIR_NT = 512;
constexpr int PAIR_PAD = 4;                             // floats added to the LDS pitch of a tile row (bank spread, see pair_pack_tile)
struct PairRole {
    void *wp;                 // null: the layer has no image of this role
    int rows_b;               // 0: image rows = the tensor's leading index a (contraction over b); 1: rows = b (contraction over a)
    int tapmode;              // 0 identity, 1 mirrored (input gradient of a stride-1 convolution), 2 sub-pixel phase order (tconv_slot)
    int taps;                 // identity only: > 0 = the image holds the first `taps` taps, the others are ZEROED in w (masked convolution)
};
struct PairDesc {
    float *w;
    int A, B, R, S;
    PairRole role[2];
    const float *bmax;        // per-chunk maxima of the optimiser pass (stem_adam_step_bmax) covering the tensor: chunks b0 .. b0 + nb - 1
    int b0, nb;
    int tile0;                // first workgroup of this tensor in the launch
};
constexpr int MAXPAIR = 20;
struct PairTable {
    PairDesc d[MAXPAIR];
    int n;
};

// TA = rows a of the tile: 32, or 16 for 5 x 5 windows (51 KB of LDS instead of 102: three workgroups per CU instead of one --
// the pass is HBM-bound and a lone workgroup per CU alternates between loading and storing)
__host__ __device__ inline int pair_ta(int RS) { return RS > 16 ? 16 : PAIR_T; }      // (8 for 5x5 and 16 for 3x3 measured 66.0 against 64.6 us)
template <int CRS, int TA>
__device__ inline void pair_pack_tile(const PairDesc &d, int tile, float *lds, int *ptap, float wscale)
{
    const int RS = CRS > 0 ? CRS : d.R * d.S;
    const int tb = cdiv_dev(d.B, PAIR_T);
    const int a0 = (tile / tb) * TA, b0 = (tile % tb) * PAIR_T;
    const int na = d.A - a0 < TA ? d.A - a0 : TA, nb = d.B - b0 < PAIR_T ? d.B - b0 : PAIR_T;
    const int row = PAIR_T * RS;                       // floats of one a-row of the tile (b-major, taps innermost): the tensor's own order
    // LDS pitch of an a-row: 32 * RS floats are a multiple of 32 banks for every window (25, 9, 1), so the emission's readers --
    // lanes = (piece p, row rr) -- all hit the banks of ONE row: 16-way conflicts.  Four more floats per row keep the float4
    // stores aligned and spread the 16 rows of a wavefront over eight bank groups
    const int lrow = row + PAIR_PAD;
    // masked convolution: the taps beyond the live prefix are zeroed IN PLACE (layers.py:44 `weight.data *= mask` at every forward)
    const int live = (d.role[0].wp && d.role[0].tapmode == 0 && d.role[0].taps > 0 && d.role[0].taps < RS) ? d.role[0].taps : RS;
    if (na == TA && nb == PAIR_T && live == RS && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 && ((size_t)d.B * RS) % 4 == 0) {
        // full tile: every a-row of the tile is 32 * RS consecutive floats starting on a 16-byte boundary -- float4 loads, four
        // independent ones in flight per thread
        constexpr int U = 4;
        const int row4 = row / 4, n4 = TA * row4;                      // 32 * RS is a multiple of 4
        for (int base = threadIdx.x; base < n4; base += U * PAIR_NT) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = base + u * PAIR_NT;
                const int a = i4 / row4, r4 = i4 - a * row4;
                v[u] = i4 < n4 ? *reinterpret_cast<const f32x4 *>(d.w + ((size_t)(a0 + a) * d.B + b0) * RS + 4 * r4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = base + u * PAIR_NT;
                if (i4 < n4) {
                    const int a = i4 / row4, r4 = i4 - a * row4;
                    *reinterpret_cast<f32x4 *>(lds + a * lrow + 4 * r4) = v[u];
                }
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < TA * row; idx += PAIR_NT) {
            const int a = idx / row, rem = idx - a * row;
            float v = 0.f;
            if (a < na && rem < nb * RS) {
                float *src = d.w + ((size_t)(a0 + a) * d.B + b0) * RS + rem;
                v = *src;
                if (live < RS && rem % RS >= live) {
                    *src = 0.f;
                    v = 0.f;
                }
            }
            lds[a * lrow + rem] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const PairRole &r = d.role[ri];
        if (!r.wp) continue;
        unsigned char *wp = static_cast<unsigned char *>(r.wp);
        const int N = r.rows_b ? d.B : d.A, C = r.rows_b ? d.A : d.B;           // image rows / contraction channels
        const int n0 = r.rows_b ? b0 : a0, nrows = r.rows_b ? nb : na, slab = (r.rows_b ? a0 : b0) / 32;
        const int nslab = C / 32, ntile = cdiv_dev(N, GBN);
        const int T = (r.tapmode == 0 && r.taps > 0) ? r.taps : RS;
        const int nchunks = nslab * T;
        int pT[4] = {0, 0, 0, 0}, pbase[4] = {0, 0, 0, 0};
        if (r.tapmode == 2) {
            int acc = 0;
            for (int p = 0; p < 4; ++p) {
                pbase[p] = acc;
                pT[p] = tconv_axis(d.R, d.R / 2, p >> 1).cnt * tconv_axis(d.S, d.S / 2, p & 1).cnt;
                acc += pT[p];
            }
        }
        // pieces of this tile in the image: rows = a: TA rows x 4 pieces (the tile's 32 b are one contraction slab); rows = b: 32 rows x
        // TA / 8 pieces (the tile's TA a are pieces pofs .. of their slab)
        const int ppr = r.rows_b ? TA / 8 : 4, prow = r.rows_b ? PAIR_T : TA;
        const int pofs = r.rows_b ? (a0 & 31) >> 3 : 0;
        for (int e = threadIdx.x; e < prow * ppr * T; e += PAIR_NT) {
            const int pp = e % ppr, rr = (e / ppr) % prow, tap = e / (ppr * prow);      // piece, row of the tile, image tap
            if (rr >= nrows) continue;
            const int p = pofs + pp;
            // source tap of image tap `tap`: identity, mirrored, or the tap that sits at phase-ordered position `tap`
            const int st = r.tapmode == 0 ? tap : (r.tapmode == 1 ? RS - 1 - tap : ptap[tap]);
            h16x8 h[NPL];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ch = pp * 8 + c;                                     // contraction channel inside the tile
                const float v = r.rows_b ? lds[ch * lrow + rr * RS + st] : lds[rr * lrow + ch * RS + st];
                hp_t x0, x1;
                q_split(v, wscale, x0, x1);
                h[0][c] = x0; h[1][c] = x1;
            }
            const int n = n0 + rr, nt = n / GBN, nl = n - nt * GBN;
            long qq = (long)nt * nchunks + slab * T + tap;
            if (r.tapmode == 2) {
                const int ph = tap >= pbase[3] ? 3 : (tap >= pbase[2] ? 2 : (tap >= pbase[1] ? 1 : 0));
                qq = (long)ntile * nslab * pbase[ph] + (long)nt * (nslab * pT[ph]) + slab * pT[ph] + (tap - pbase[ph]);
            }
            unsigned char *dst = wp + qq * GB_BUF + nl * 64 + ((p ^ ((nl >> 2) & 3)) << 4);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<h16x8 *>(dst + pl * GB_PLANE) = h[pl];
        }
    }
}

__global__ __launch_bounds__(PAIR_NT) void pack_weight_pair_multi_kernel(const PairTable tab)
{
    extern __shared__ __attribute__((aligned(16))) float pair_lds[];            // [32][32][RS] floats, then the phase tap table
    __shared__ float qred[16];
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.d[i + 1].tile0) ++i;
    const PairDesc &d = tab.d[i];
    const int tile = blockIdx.x - d.tile0, RS = d.R * d.S;
    int *ptap = reinterpret_cast<int *>(pair_lds + pair_ta(RS) * (PAIR_T * RS + PAIR_PAD));
    if ((int)threadIdx.x < RS) ptap[tconv_slot(d.R, d.S, d.R / 2, threadIdx.x / d.S, threadIdx.x % d.S)] = threadIdx.x;
    // one scale for both images: the maximum of the optimiser pass's chunks that cover the tensor (an upper bound: neighbours may
    // share a chunk), as in pack_weight_gen_multi_kernel
    float mm = 0.f;
    for (int k = threadIdx.x; k < 4 * d.nb; k += PAIR_NT) mm = fmaxf(mm, d.bmax[4 * d.b0 + k]);
    const float wmax = block_max(mm, qred);
    const int we = q_exp(wmax);
    if (tile == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            const PairRole &r = d.role[ri];
            if (!r.wp) continue;
            const int N = r.rows_b ? d.B : d.A, C = r.rows_b ? d.A : d.B;
            float *wq = reinterpret_cast<float *>(static_cast<unsigned char *>(r.wp) + (size_t)cdiv_dev(N, GBN) * (C / 32) * RS * GB_BUF);
            q_header(wq, 1);
            wq[1] = q_pow2(-we);
            wq[QREC_HDR] = wmax;
        }
    }
    __syncthreads();
    const float wscale = q_pow2(we);
    if (RS == 25)
        pair_pack_tile<25, 16>(d, tile, pair_lds, ptap, wscale);
    else if (RS == 9)
        pair_pack_tile<9, PAIR_T>(d, tile, pair_lds, ptap, wscale);
    else if (RS == 1)
        pair_pack_tile<1, PAIR_T>(d, tile, pair_lds, ptap, wscale);
    else if (RS > 16)
        pair_pack_tile<0, 16>(d, tile, pair_lds, ptap, wscale);
    else
        pair_pack_tile<0, PAIR_T>(d, tile, pair_lds, ptap, wscale);
}

}   // namespace

/* Both images of every layer from one read of its weights (csrc: pack_weight_pair_multi_kernel): descs[i] names the torch tensor
 * [A][B][R][S], up to two images (role 0 / 1: destination, rows = a or b, tap order, live-tap prefix) and the optimiser pass's
 * chunk maxima that cover it (mandatory here: the launch has no maximum pass of its own).  Images of a layer whose row count is
 * not a multiple of 128 must have been zero-filled once (the padding rows are never written).  What it replaces:
 * stem_f16x2_pack_conv_weights_multi called once per role after every optimiser step (stem/trainSTEM.py:213). */
STEM_EXPORT int stem_f16x2_pack_conv_weights_pair_multi(const stem_f16x2_pair_desc *descs, int n, void *stream)
{
    STEM_CHECK_ARG(descs && n >= 1 && n <= MAXPAIR, "stem_f16x2_pack_conv_weights_pair_multi: 1..%d descriptors per call, got %d", MAXPAIR, n);
    PairTable tab;
    memset(&tab, 0, sizeof(tab));
    int tiles = 0;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        const stem_f16x2_pair_desc &h = descs[i];
        STEM_CHECK_ARG(h.w && h.A >= 1 && h.B >= 1 && h.R >= 1 && h.S >= 1 && h.R * h.S <= MAXTAP && h.bmax && h.nb >= 1,
                       "stem_f16x2_pack_conv_weights_pair_multi: descriptor %d: tensor, window (<= %d taps) and chunk maxima are mandatory", i, MAXTAP);
        PairDesc &d = tab.d[i];
        d.w = static_cast<float *>(const_cast<void *>(h.w));
        d.A = h.A; d.B = h.B; d.R = h.R; d.S = h.S; d.bmax = h.bmax; d.b0 = h.b0; d.nb = h.nb;
        const void *wps[2] = {h.wp0, h.wp1};
        const int modes[2] = {h.mode0, h.mode1}, taps[2] = {h.taps0, h.taps1};
        for (int r = 0; r < 2; ++r) {
            d.role[r].wp = const_cast<void *>(wps[r]);
            if (!wps[r]) continue;
            d.role[r].rows_b = modes[r] & 1;
            d.role[r].tapmode = modes[r] >> 1;
            d.role[r].taps = taps[r];
            const int C = d.role[r].rows_b ? h.A : h.B;
            STEM_CHECK_ARG(d.role[r].tapmode >= 0 && d.role[r].tapmode <= 2 && C % 32 == 0 && taps[r] >= 0 && taps[r] <= h.R * h.S &&
                           (taps[r] == 0 || (d.role[r].tapmode == 0 && r == 0)) && (d.role[r].tapmode != 2 || (h.R == h.S && (h.R & 1))),
                           "stem_f16x2_pack_conv_weights_pair_multi: descriptor %d role %d: contraction channels %% 32, tap order 0..2, a live-tap "
                           "prefix only for role 0 in identity order, phase order for odd square windows", i, r);
        }
        d.tile0 = tiles;
        tiles += cdiv(h.A, pair_ta(h.R * h.S)) * cdiv(h.B, PAIR_T);
        const size_t need = (size_t)pair_ta(h.R * h.S) * (PAIR_T * h.R * h.S + PAIR_PAD) * sizeof(float) + 32 * sizeof(int);
        if (need > lds) lds = need;
    }
    tab.n = n;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute((const void *)pack_weight_pair_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(pack_weight_pair_multi_kernel, dim3(tiles), dim3(PAIR_NT), lds, (hipStream_t)stream, tab);
    STEM_LAUNCH_CHECK("stem_f16x2_pack_conv_weights_pair_multi");
    return 0;
}

namespace {
constexpr size_t kGenCntBytes = 64 * 1024;          // arrival counters in front of the split-K slabs: the layout of igemm.hip's
                                                    // workspace, so that one zero-headed buffer per stream serves both kernels

// Split factor.  Two workgroups fit a CU; the launch runs in R = ceil(workgroups / 256) "CU rounds", each as long as one
// workgroup's chunks (+ ~6 chunks of ramp-up per workgroup) -- except that a lone workgroup per CU (R = 1) leaves the matrix
// pipes ~40 % idle (one wavefront per SIMD).  Every split also pays its slab in the reduction.  Measured on TPM.0 / .2 / .4
// with splits 1..10 (tools/debug/f16x3_gen_check.py, STEM_FX3_SPLIT): the model ranks them as measured, optimum 4 / 4 / 4.
int gen_split(int tiles, int nchunks)
{
    const int forced = stem_tuning(STEM_TUNE_FX3_SPLIT);      // stem_tuning_set("fx3_split", n): tests / sweeps
    if (forced > 0) return forced < nchunks ? forced : nchunks;
    // 1x1 layers (EPM: 18-36 chunks): a split costs its slab round trip and a second epilogue pass, more than the shorter loop
    // returns (tools/debug/f16x3_split_sweep.py, round 3: unsplit 39 / 29 / 20 us against 43 / 33 / 26 us)
    if (nchunks <= 40 && tiles >= 128) return 1;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 16 && s * 8 <= nchunks; ++s) {
        const int cps = cdiv(nchunks, s), ns = cdiv(nchunks, cps), rounds = cdiv(tiles * ns, 256);
        const double cost = rounds * (cps + 6.0) * (rounds == 1 ? 1.67 : 1.0) + (ns > 1 ? 0.3 * ns : 0.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = ns;
        }
    }
    return best;
}

// Pixel tile: 128 when that still leaves enough tiles to spread over the chip with a moderate split and the loop is long enough
// to pay for the larger epilogue, else 64 (tools/debug/f16x3_split_sweep.py: TPM.0 / .2 / .4 52 / 78 / 100 -> 49 / 73 / 90 us,
// EPM.0 / .2 41 / 31 -> 38 / 28 us, EPM.4 with its 18 chunks 20 -> 23 us)
int gen_bm(int M, int ntn, int nchunks)
{
    const int forced = stem_tuning(STEM_TUNE_FX3_GEN_TILE);   // stem_tuning_set("fx3_gen_tile", 64 | 128): tests / sweeps
    if (forced) return forced;
    // (the nchunks > 20 clause of the isolated sweep is gone: inside the training step the short 1x1 launches -- EPM.2 / EPM.4 and
    // the EPM input gradients, 22-26 us on either tile alone -- run 0.07 ms per step faster on the larger workgroups, six
    // alternating pairs 15.21-15.26 against 15.16-15.20 ms: fewer, larger workgroups share the loaded chip better)
    (void)nchunks;
    return cdiv(M, 128) * ntn >= 64 ? 128 : 64;
}
}   // namespace

STEM_EXPORT size_t stem_conv2d_f16x3_gen_workspace_bytes(int B, int H, int W, int C, int N, int R, int S, int stride, int pad, int taps)
{
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    if (OH < 1 || OW < 1 || C % 32) return 0;
    const int M = B * OH * OW, nchunks = (C / 32) * (taps > 0 ? taps : R * S), tiles = cdiv(M, gen_bm(M, cdiv(N, GBN), nchunks)) * cdiv(N, GBN);
    int s = gen_split(tiles, nchunks);
    if (stem_fx3_img_eligible(B, H, W, N, R, S, stride, pad)) {       // the image-tile form plans its own split: room for either
        const int si = stem_fx3_img_split(stem_fx3_img_tiles(B, H, W, N), nchunks);
        if (si > s) s = si;
    }
    while (s > 1 && (size_t)s * M * cdiv(N, GBN) * GBN * sizeof(float) >= 0x7FFFFF00ull) --s;
    return s > 1 ? kGenCntBytes + (size_t)s * M * cdiv(N, GBN) * GBN * sizeof(float) : 0;
}

namespace {
// N_image / n0: the weight image holds N_image rows and this launch computes its rows [n0, n0 + N) -- n0 a multiple of the 128-row
// N tile, so the sub-image is a contiguous range of the image's N tiles; the scale record is the whole image's
int gen_fwd_impl(const void *wp_image, int N_image, int n0, const void *xp, const float *xq, int xpix, const float *bias, int epi, float slope,
                 const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R,
                 int S, int stride, int pad, int taps, void *ws, size_t ws_bytes, void *stream)
{
    STEM_CHECK_ARG(taps >= 0 && taps <= R * S, "stem_conv2d_f16x3_gen_fwd: 0 <= taps <= R*S (taps=%d)", taps);
    const int T = taps > 0 ? taps : R * S;
    STEM_CHECK_ARG(wp_image && n0 >= 0 && n0 % GBN == 0 && n0 + N <= N_image && (n0 + N == N_image || N % GBN == 0),
                   "stem_conv2d_f16x3_gen_fwd: rows [%d, %d) of an image of %d rows must be whole 128-row tiles", n0, n0 + N, N_image);
    const void *wp = static_cast<const unsigned char *>(wp_image) + (size_t)(n0 / GBN) * (C / 32) * T * GB_BUF;
    STEM_CHECK_ARG(xp && xq && wp && (y || yp) && (yq || !yp), "stem_conv2d_f16x3_gen_fwd: null pointer (planes come with their scale records)");
    STEM_CHECK_ARG(epi == GEN_EPI_BIAS || fabsf(slope) <= 1.f, "stem_conv2d_f16x3_gen_fwd: |slope| <= 1");
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 32 && C % 32 == 0 && N >= 4 && N % 4 == 0 && R >= 1 && S >= 1 && R * S <= MAXTAP &&
                   stride >= 1 && pad >= 0, "stem_conv2d_f16x3_gen_fwd: C %% 32 == 0, N %% 4 == 0, R*S <= %d (C=%d N=%d R=%d S=%d)", MAXTAP, C, N, R, S);
    STEM_CHECK_ARG(!yp || N % 32 == 0, "stem_conv2d_f16x3_gen_fwd: planes output needs N %% 32 == 0 (N=%d)", N);
    STEM_CHECK_ARG(!y || (ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0), "stem_conv2d_f16x3_gen_fwd: y rows must be 16-byte aligned, ldy >= N");
    STEM_CHECK_ARG(epi >= GEN_EPI_BIAS && epi <= GEN_EPI_DACT, "stem_conv2d_f16x3_gen_fwd: unknown epilogue %d", epi);
    STEM_CHECK_ARG(epi != GEN_EPI_DACT || (z && ldz >= N && ldz % 4 == 0 && ((uintptr_t)z & 15) == 0), "stem_conv2d_f16x3_gen_fwd: DACT needs z (16-byte aligned rows)");
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    STEM_CHECK_ARG(OH >= 1 && OW >= 1, "stem_conv2d_f16x3_gen_fwd: empty output");
    if (xpix == 0) xpix = (C / 32) * SLAB;
    STEM_CHECK_ARG(xpix >= (C / 32) * SLAB && xpix % SLAB == 0, "stem_conv2d_f16x3_gen_fwd: xpix must be a multiple of %d bytes covering C channels", SLAB);
    const size_t xb = (size_t)B * H * W * xpix, wb = gen_image_bytes(N, C, R, S);
    const int M = B * OH * OW, ntn = cdiv(N, GBN), nchunks = (C / 32) * T, bm = gen_bm(M, ntn, nchunks), tiles = cdiv(M, bm) * ntn;
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && wb < 0x7FFFFF00ull && (size_t)M * ntn * GBN * 4 < 0x7FFFFF00ull,
                   "stem_conv2d_f16x3_gen_fwd: operand views must stay below 2 GiB (split the batch)");
    Fx3Args a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.wp = wp; a.bias = bias; a.y = y; a.yp = yp; a.ldy = ldy; a.z = z; a.ldz = ldz; a.epi = epi; a.slope = slope;
    const float *wq_image = reinterpret_cast<const float *>(static_cast<const unsigned char *>(wp_image) + gen_image_bytes(N_image, C, R, S));
    a.xq = xq; a.yq = yq; a.wq = wq_image;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.OH = OH; a.OW = OW; a.stride = stride; a.ntaps = T; a.S = S;
    a.xbytes = (int)xb; a.wbytes = (int)wb; a.xpix = xpix;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            a.dy[r * S + s] = (signed char)(r - pad);
            a.dx[r * S + s] = (signed char)(s - pad);
        }
    // stride-1 "same" layers on images of 16x16 blocks: the image-tile form (conv_f16x3_img.hip: halo in LDS, weights by LDS-DMA)
    const bool img = N == N_image && stem_fx3_img_eligible(B, H, W, N, R, S, stride, pad);
    const int itiles = img ? stem_fx3_img_tiles(B, H, W, N) : 0;
    int split = img ? stem_fx3_img_split(itiles, nchunks) : gen_split(tiles, nchunks);
    while (split > 1 && (size_t)split * M * ntn * GBN * sizeof(float) >= 0x7FFFFF00ull) --split;      // the slabs are read through one buffer view
    const size_t need = kGenCntBytes + (size_t)split * M * ntn * GBN * sizeof(float);
    if (split > 1 && (!ws || ws_bytes < need || (size_t)tiles * sizeof(int) > kGenCntBytes)) split = 1;      // no workspace: unsplit, same result up to summation order
    if (img)
        return stem_fx3_img_launch(xp, xq, xpix, (int)xb, wp, wq_image, (int)wb,
                                   bias, epi, slope, z, ldz, y, ldy, yp, yq, B, H, W, C, N, R, T, split,
                                   split > 1 ? reinterpret_cast<float *>(static_cast<unsigned char *>(ws) + kGenCntBytes) : nullptr,
                                   split > 1 ? static_cast<int *>(ws) : nullptr, stream);
    a.nsplit = split;
    a.cps = cdiv(nchunks, split);
    a.nsplit = cdiv(nchunks, a.cps);
    if (a.nsplit > 1) {
        a.cnt = static_cast<int *>(ws);
        a.ws = reinterpret_cast<float *>(static_cast<unsigned char *>(ws) + kGenCntBytes);
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<64, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(64));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<128, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(128));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<64, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(64));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<128, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(128));
        attr_done = true;
    }
    const dim3 grid(cdiv(M, bm), ntn, a.nsplit);
    const int gmsel = stem_tuning(STEM_TUNE_FX3_GEN_MFMA);         // MFMA shape: stem_tuning_set("fx3_gen_mfma", 16 | 32); 0 = default
    const bool g16 = gmsel ? gmsel == 16 : GEN_DEFAULT_MFMA16;
    if (bm == 128 && g16)
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<128, 16>), grid, dim3(512), glds(128), (hipStream_t)stream, a);
    else if (bm == 128)
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<128, 32>), grid, dim3(512), glds(128), (hipStream_t)stream, a);
    else if (g16)
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<64, 16>), grid, dim3(256), glds(64), (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<64, 32>), grid, dim3(256), glds(64), (hipStream_t)stream, a);
    STEM_LAUNCH_CHECK("stem_conv2d_f16x3_gen_fwd");
    return 0;
}
}   // namespace

STEM_EXPORT int stem_conv2d_f16x3_gen_fwd(const void *xp, const float *xq, int xpix, const void *wp, const float *bias, int epi, float slope,
                                           const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R,
                                           int S, int stride, int pad, int taps, void *ws, size_t ws_bytes, void *stream)
{
    return gen_fwd_impl(wp, N, 0, xp, xq, xpix, bias, epi, slope, z, ldz, y, ldy, yp, yq, B, H, W, C, N, R, S, stride, pad, taps, ws, ws_bytes, stream);
}

/* the same for rows [n0, n0 + N) of a weight image of N_image rows (n0 and, unless the range ends the image, N multiples of 128):
 * a layer's outputs computed range by range, so that the consumer of one range need not wait for the others (the input gradient
 * of EPM.0 -- spatiotemporalpriors.py:832-838 in backward -- feeds three independent chains) */
STEM_EXPORT int stem_conv2d_f16x3_gen_fwd_rows(const void *xp, const float *xq, int xpix, const void *wp_image, int N_image, int n0,
                                                const float *bias, int epi, float slope, const float *z, int ldz, float *y, int ldy, void *yp,
                                                float *yq, int B, int H, int W, int C, int N, int R, int S, int stride, int pad, int taps, void *ws,
                                                size_t ws_bytes, void *stream)
{
    return gen_fwd_impl(wp_image, N_image, n0, xp, xq, xpix, bias, epi, slope, z, ldz, y, ldy, yp, yq, B, H, W, C, N, R, S, stride, pad, taps, ws,
                        ws_bytes, stream);
}

namespace {
// chunks per workgroup of a four-phase launch: every workgroup gets (about) the same number of chunks whatever its phase; the
// cost model of gen_split with the phases' workgroups added up
int tconv_cps(int tiles_per_phase, const int (&nchunks)[4])
{
    const int forced = stem_tuning(STEM_TUNE_FX3_SPLIT);
    int maxc = 0;
    for (int p = 0; p < 4; ++p) maxc = nchunks[p] > maxc ? nchunks[p] : maxc;
    if (forced > 0) return cdiv(maxc, forced < maxc ? forced : maxc);
    if (stem_tuning(STEM_TUNE_TCONV_CPS) > 0) return stem_tuning(STEM_TUNE_TCONV_CPS);       // stem_tuning_set("tconv_cps", n): sweeps
    // These launches are latency-bound (tools/debug/tconv_sweep.py, round 5: HD.0 24-29 us and HD.2 38-40 us for 12-24 chunks per
    // workgroup, 49-64 us at 4-8, 45-62 us at 36-72): the shortest loop that keeps the launch within one round of 512 workgroups
    // (two per CU) and the reduction within 16 slabs, but not below 12 chunks -- under that the slab round trip outweighs the loop
    for (int cps = 12; cps < maxc; ++cps) {
        int wgs = 0, maxns = 0;
        for (int p = 0; p < 4; ++p) {
            const int ns = cdiv(nchunks[p], cps);
            wgs += tiles_per_phase * ns;
            maxns = ns > maxns ? ns : maxns;
        }
        if (wgs <= 512 && maxns <= 16) return cps;
    }
    return maxc;
}
}   // namespace

STEM_EXPORT size_t stem_tconv2d_f16x3_workspace_bytes(int B, int H, int W, int C, int N, int R)
{
    if (C % 32 || R < 1 || R * R > MAXTAP || !(R & 1)) return 0;
    const int M = B * H * W, ntn = cdiv(N, GBN);
    // room for any plan: 16 splits of every phase
    return kGenCntBytes + (size_t)4 * 16 * M * ntn * GBN * sizeof(float);
}

/* The transposed face of a stride-2, R x R, padding R/2 layer whose fine grid is exactly twice the coarse one:
 *   forward of nn.ConvTranspose2d(C, N, R, stride=2, padding=R/2, output_padding=1)   (HD.0 / HD.2, spatiotemporalpriors.py:822-826)
 *   input gradient of nn.Conv2d(N, C, R, stride=2, padding=R/2) on an even-sized input   (HE.2 / HE.4, :814-818, torch autograd)
 * x: planes of the coarse tensor [B, H, W, C]; wp: the four phase images (stem_f16x2_pack_conv_weights_multi, flip = 2, rows =
 * the N outputs, contraction channels C); y / yp: fp32 rows / planes of the fine tensor [B, OHf, OWf, N], OHf = 2H (or 2H - 1: the
 * input gradient of a strided convolution whose input had an odd number of rows; same for OWf); epi / z as in
 * stem_conv2d_f16x3_gen_fwd (z: rows of the FINE tensor).  One launch: blockIdx.x = phase x coarse pixel tile. */
STEM_EXPORT int stem_tconv2d_f16x3_fwd(const void *xp, const float *xq, int xpix, const void *wp, const float *bias, int epi, float slope,
                                        const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R,
                                        int OHf, int OWf, void *ws, size_t ws_bytes, void *stream)
{
    STEM_CHECK_ARG(xp && xq && wp && (y || yp) && (yq || !yp), "stem_tconv2d_f16x3_fwd: null pointer (planes come with their scale records)");
    STEM_CHECK_ARG((OHf == 2 * H || OHf == 2 * H - 1) && (OWf == 2 * W || OWf == 2 * W - 1),
                   "stem_tconv2d_f16x3_fwd: the fine grid is 2H or 2H - 1 rows (the odd-sized input of a strided convolution), got %d x %d for %d x %d", OHf, OWf, H, W);
    STEM_CHECK_ARG(epi >= GEN_EPI_BIAS && epi <= GEN_EPI_DACT && (epi == GEN_EPI_BIAS || fabsf(slope) <= 1.f), "stem_tconv2d_f16x3_fwd: epilogue %d", epi);
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 32 && C % 32 == 0 && N >= 4 && N % 4 == 0 && R >= 3 && (R & 1) && R * R <= MAXTAP,
                   "stem_tconv2d_f16x3_fwd: C %% 32 == 0, N %% 4 == 0, odd 3 <= R, R*R <= %d (C=%d N=%d R=%d)", MAXTAP, C, N, R);
    STEM_CHECK_ARG(!yp || N % 32 == 0, "stem_tconv2d_f16x3_fwd: planes output needs N %% 32 == 0 (N=%d)", N);
    STEM_CHECK_ARG(!y || (ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0), "stem_tconv2d_f16x3_fwd: y rows must be 16-byte aligned, ldy >= N");
    STEM_CHECK_ARG(epi != GEN_EPI_DACT || (z && ldz >= N && ldz % 4 == 0 && ((uintptr_t)z & 15) == 0), "stem_tconv2d_f16x3_fwd: DACT needs z (16-byte aligned rows)");
    if (xpix == 0) xpix = (C / 32) * SLAB;
    STEM_CHECK_ARG(xpix >= (C / 32) * SLAB && xpix % SLAB == 0, "stem_tconv2d_f16x3_fwd: xpix must be a multiple of %d bytes covering C channels", SLAB);
    const int pad = R / 2, nslab = C / 32, M = B * H * W, ntn = cdiv(N, GBN);
    const size_t xb = (size_t)B * H * W * xpix, wb = gen_image_bytes(N, C, R, R);
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && wb < 0x7FFFFF00ull && (size_t)4 * M * ntn * GBN * 4 < 0x7FFFFF00ull,
                   "stem_tconv2d_f16x3_fwd: operand views must stay below 2 GiB (split the batch)");
    int nch[4], maxT = 0;
    Fx3Args a;
    memset(&a, 0, sizeof(a));
    int tapbase = 0;
    for (int p = 0; p < 4; ++p) {
        const TAxis ay = tconv_axis(R, pad, p >> 1), ax = tconv_axis(R, pad, p & 1);
        Fx3Phase &ph = a.ph[p];
        ph.ntaps = ay.cnt * ax.cnt;
        ph.S = ax.cnt;
        ph.dy0 = ay.d0;
        ph.dx0 = ax.d0;
        ph.ooy = p >> 1;
        ph.oox = p & 1;
        ph.wofs = (int)((size_t)ntn * nslab * tapbase * GB_BUF);
        tapbase += ph.ntaps;
        nch[p] = nslab * ph.ntaps;
        maxT = ph.ntaps > maxT ? ph.ntaps : maxT;
        STEM_CHECK_ARG(ph.ntaps >= 1, "stem_tconv2d_f16x3_fwd: empty phase (R=%d)", R);
    }
    // 64-pixel workgroups unless the coarse grid alone fills the chip (HD.2 at B = 16: 34 against 49 us)
    const int forced_bm = stem_tuning(STEM_TUNE_FX3_GEN_TILE);
    const int bm = forced_bm ? forced_bm : (4 * cdiv(M, 128) * ntn >= 256 ? 128 : 64), ptiles = cdiv(M, bm);
    int cps = tconv_cps(ptiles * ntn, nch);
    // the slabs of all phases sit behind each other in ws; without room for the plan the launch is unsplit (same result up to order)
    size_t wsfl = 0;
    int maxns = 1;
    for (int pass = 0; pass < 2; ++pass) {
        wsfl = 0;
        maxns = 1;
        for (int p = 0; p < 4; ++p) {
            Fx3Phase &ph = a.ph[p];
            ph.cps = cps < nch[p] ? cps : nch[p];
            ph.nsplit = cdiv(nch[p], ph.cps);
            ph.wsofs = (int)wsfl;
            if (ph.nsplit > 1) wsfl += (size_t)ph.nsplit * M * ntn * GBN;
            maxns = ph.nsplit > maxns ? ph.nsplit : maxns;
        }
        const bool fits = ws && kGenCntBytes + wsfl * sizeof(float) <= ws_bytes && (size_t)4 * ptiles * ntn * sizeof(int) <= kGenCntBytes &&
                          wsfl * sizeof(float) < 0x7FFFFF00ull;
        if (maxns == 1 || fits) break;
        cps = 1 << 30;                              // second pass: unsplit
    }
    a.xp = xp; a.wp = wp; a.bias = bias; a.y = y; a.yp = yp; a.ldy = ldy; a.z = z; a.ldz = ldz; a.epi = epi; a.slope = slope;
    a.xq = xq; a.yq = yq; a.wq = reinterpret_cast<const float *>(static_cast<const unsigned char *>(wp) + wb);
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.OH = H; a.OW = W; a.stride = 1;
    a.xbytes = (int)xb; a.wbytes = (int)wb; a.xpix = xpix;
    a.ptiles = ptiles; a.btaps = maxT; a.OHf = OHf; a.OWf = OWf;
    if (maxns > 1) {
        a.cnt = static_cast<int *>(ws);
        a.ws = reinterpret_cast<float *>(static_cast<unsigned char *>(ws) + kGenCntBytes);
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<64, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(64));
        (void)hipFuncSetAttribute((const void *)conv_f16x3_gen_kernel<128, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, glds(128));
        attr_done = true;
    }
    const dim3 grid(4 * ptiles, ntn, maxns);
    if (bm == 128)
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<128, 16, true>), grid, dim3(512), glds(128), (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_f16x3_gen_kernel<64, 16, true>), grid, dim3(256), glds(64), (hipStream_t)stream, a);
    STEM_LAUNCH_CHECK("stem_tconv2d_f16x3_fwd");
    return 0;
}
