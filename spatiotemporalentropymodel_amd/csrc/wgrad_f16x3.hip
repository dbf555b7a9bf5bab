// Weight gradient of a stride-1 convolution on the fp16 matrix cores with fp32-class products (see conv_f16x3.hip for the
// arithmetic: operands pre-split into two fp16 planes, three MFMAs per fp32 product, fp32 accumulate).
//
//   dW[k][c][r][s] = sum over output pixels m = (b, y, x) of dy[m][k] * x[b, y + r - pad, x + s - pad][c]
//
// Per tap this is a GEMM dW_t[K x C] = dY^T[K x M] . X_t[M x C] whose contraction runs over PIXELS, while both operands are
// stored pixel-major ([pixel][C/32][2][32] fp16).  The MFMA wants, per lane, 8 consecutive contraction elements of one row /
// column, i.e. 8 pixels of one channel: the chunk (16 pixels x 128 channels per operand and plane) is staged in LDS as it comes
// (256-byte pixel rows) and read back with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 4-pixel x 16-channel block per
// 16 lanes, delivered channel-major) -- two reads per operand fragment, no shuffles.  LDS image: the guide's 256-byte-row form
// off(row, chunk) = 256 row + 16 (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))), conflict-free for these reads.
//
// Workgroup: 128 (k) x 128 (c) outputs of ONE tap over a range of pixels, 4 wavefronts x (64 x 64), 32 KiB of LDS (two workgroups
// per CU); grid = (k tiles x c tiles, taps, pixel splits); partial sums go to slabs [split][tap][K][C] -- the layout of wgrad.hip, summed and transposed into the
// torch layout by stem_unpack_wgrads_multi.  The tap's shift is applied when the x rows are fetched (rows outside the image read
// as zeros through the range-checked buffer loads).
#include <stdlib.h>

#include "stem_common.h"

namespace {

typedef hp8 h16x8;
typedef short v4s __attribute__((ext_vector_type(4)));

constexpr int TK = 128, TC = 128, NT = 256, PX = 16;      // output tile, threads, pixels per chunk (= one MFMA k-step)
constexpr int PLANE = PX * 256;                            // 4096 B: 16 pixel rows x 128 channels of one plane
constexpr int NPL = 2, SLAB = NPL * 64;                    // planes per value; bytes per pixel and 32-channel slab
constexpr int OP_BUF = NPL * PLANE;                        // one operand, one buffer
constexpr int LDS_BYTES = 2 * 2 * OP_BUF;                  // 32768: two workgroups per CU (register-limited), 2 wavefronts per SIMD
constexpr int OOR = 0x7FFFFF00;
constexpr int MAXTAP = 25;

struct Wg3Args {
    const void *xp, *dyp;
    const float *xq, *dyq;         // scale records of the two planes tensors (stem_common.h)
    float *dwp;
    float *bias_part;              // optional [nsplit][K]: per-split column sums of dy (the bias gradient's first stage)
    int xpix, dypix;               // bytes per pixel of the planes buffers (channel views allowed)
    int B, H, W, C, K, OH, OW, T;
    int nsplit, cps;               // pixel chunks per split
    int xbytes, dybytes;
    int stride;                    // per-tap form only: output pixel (oy, ox) of tap t reads x at (oy * stride + dy[t], ox * stride + dx[t])
    signed char dy[MAXTAP], dx[MAXTAP];
};

// ---- filter-row form ---------------------------------------------------------------------------------------------------------
// The per-tap kernel below fetches 16 KiB of operands per 48 MFMAs and workgroup and re-reads dy once per tap.  For 'same'
// convolutions whose output rows are multiples of 16 pixels a pixel chunk is a run of 16 pixels of ONE image row, and the S taps
// of a filter row read the same x row shifted by one pixel each: this form's workgroup owns 128 (k) x 64 (c) outputs of ALL S taps
// of one filter row.  Per chunk it stages the dy rows once and the x row once with its halo (16 + S - 1 pixels, zeros outside
// the image; LDS-DMA into a four-stage ring, three chunks ahead), reads the dy fragments once and the x fragments of tap s from
// LDS rows s .. s + 15 -- S times the flop per staged byte and per dy read.
// 4 wavefronts x (64 k x 32 c) x S taps: 32 S accumulator registers per lane (160 for a 5-tap row), two workgroups per CU.
// LDS image of x: [row][plane][128 B] in 256-byte rows with the same XOR swizzle (the plane is bit 3 of the 16-byte chunk index:
// plane 1's address is plane 0's ^ 128); any four consecutive rows hit distinct bank groups, so the shifted reads stay
// conflict-free.
constexpr int RT_K = 128, RT_C = 64, RNT = 256;
constexpr int R_DY = NPL * PLANE;                          // 8192: dy [plane][16 px][256 B]
constexpr int R_XROWS = 32;                                // 16 + S - 1 <= 20 used; the copy engine fills whole 1 KiB blocks
constexpr int R_BUF = R_DY + R_XROWS * 256;                // 16384
constexpr int R_LDS = 4 * R_BUF;                           // four stages: chunk q is multiplied while q + 1 .. q + 3 are in flight
constexpr int R_MINCHUNKS = 64;                            // pixel chunks per workgroup the planner keeps (slab traffic).  Round 5: 64 for every
                                                           // window (32, 16 for 1x1 until round 4): at most four splits at the 4096-pixel
                                                           // training size -- fewer slabs to write and sum.  In the bench step 11.54 ms against
                                                           // 11.77 (48: 11.59, 86: 12.09, 128: 12.09, unsplit: 13.99; profiles/r05_ab_wg3_minch.log)

__device__ inline int lds_off(int row, int chunk) { return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ inline h16x8 tr_frag(const unsigned char *base, int a0, int a1)
{
    // two transposing reads: pixels 8h .. 8h+3 and 8h+4 .. 8h+7 of this lane's channel
    typedef __attribute__((address_space(3))) v4s *lp;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + a0));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + a1));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(h16x8, v);
}

__global__ __launch_bounds__(NT, 2) void wgrad_f16x3_kernel(const Wg3Args a)
{
    constexpr int PL = NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *As = smem;                        // [2][NPL][16 px][256 B]   dy
    unsigned char *Bs = smem + 2 * OP_BUF;           // [2][NPL][16 px][256 B]   x (shifted by the tap)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_c = (a.C + TC - 1) / TC;
    const int tk = blockIdx.x / tiles_c, tc = blockIdx.x - tk * tiles_c;
    const int k0 = tk * TK, c0 = tc * TC, tap = blockIdx.y, split = blockIdx.z;
    const int M = a.B * a.OH * a.OW, nchunks = (M + PX - 1) / PX;
    const int q_begin = split * a.cps, q_end = q_begin + a.cps < nchunks ? q_begin + a.cps : nchunks;
    const int tdy = a.dy[tap], tdx = a.dx[tap];

    // ---- staging: thread -> pixel row (tid / 16), 32-channel slab (tid / 4 % 4), 16-byte piece of the slab's 64-byte plane rows ----
    const int srow = tid >> 4, sslab = (tid >> 2) & 3, spc = tid & 3;
    const int st0 = lds_off(srow, sslab * 4 + spc);
    const bool ka_ok = k0 / 32 + sslab < a.K / 32, cb_ok = c0 / 32 + sslab < a.C / 32;
    const int a_col = (k0 / 32 + sslab) * SLAB + spc * 16, b_col = (c0 / 32 + sslab) * SLAB + spc * 16;
    const int ohw = a.OH * a.OW;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.xp), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.dyp), 0, a.dybytes, 0x00020000);

    f32x4 raA[PL], rbA[PL], raB[PL], rbB[PL];
    auto gload = [&](int q, f32x4 (&ra)[PL], f32x4 (&rb)[PL]) {
        const int m = q * PX + srow;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw, oy = rem / a.OW, ox = rem - oy * a.OW;
        const int iy = oy * a.stride + tdy, ix = ox * a.stride + tdx;
        const int offa = (ok && ka_ok) ? m * a.dypix + a_col : OOR;
        const int offb = (ok && cb_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? ((b * a.H + iy) * a.W + ix) * a.xpix + b_col : OOR;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            ra[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, offa + pl * 64, 0, 0));
            rb[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, offb + pl * 64, 0, 0));
        }
    };
    // bias gradient: the workgroups of tap 0 / channel tile 0 see every dy row of their k tile and pixel range exactly once on
    // its way to LDS; they add the three planes up again (exact) and keep per-thread column sums of their 8 channels
    const bool do_bias = a.bias_part != nullptr && tap == 0 && tc == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto sstore = [&](int buf, f32x4 (&ra)[PL], f32x4 (&rb)[PL], bool fresh) {       // fresh: not a clamped repeat of the last chunk
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            *reinterpret_cast<f32x4 *>(As + buf * OP_BUF + pl * PLANE + st0) = ra[pl];
            *reinterpret_cast<f32x4 *>(Bs + buf * OP_BUF + pl * PLANE + st0) = rb[pl];
        }
        if (do_bias && fresh) {
            const hp8 p0 = __builtin_bit_cast(hp8, ra[0]), p1 = __builtin_bit_cast(hp8, ra[1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                bsum[c] += (float)p0[c] + (float)p1[c];
            }
        }
    };

    // ---- operand fragments: lane (r = lane & 31, h = lane >> 5) needs pixels 8h .. 8h+7 of channel r of its 32-channel tile ------
    // 16-lane group g = lane / 16 covers channels 16 (g & 1) .. +15 and the pixel half h = g >> 1; inside the group lane 4q + p
    // supplies the address of pixel row q, channels 4p .. 4p+3 of the block (cdna guide, T10)
    const int wk0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
    const int g = lane >> 4, cg = g & 1, hh = g >> 1, qq = (lane >> 2) & 3, pp = lane & 3;
    int adr[2][2][2];          // [operand 0 = dy / 1 = x][32-channel tile][read 0 / 1]
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = 8 * hh + 4 * u + qq;
                const int chunk = ((op ? wc0 : wk0) + t * 32) / 8 + 2 * cg + (pp >> 1);
                adr[op][t][u] = lds_off(row, chunk) + 8 * (pp & 1);
            }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto step = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[PL], int qcur, int qn) {
        const unsigned char *Ab = As + cur * OP_BUF, *Bb = Bs + cur * OP_BUF;
        h16x8 af[2][PL], bf[2][PL];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
                af[t][pl] = tr_frag(Ab + pl * PLANE, adr[0][t][0], adr[0][t][1]);
                bf[t][pl] = tr_frag(Bb + pl * PLANE, adr[1][t][0], adr[1][t][1]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = STEM_MFMA16(af[i][1], bf[j][0], acc[i][j]);
                acc[i][j] = STEM_MFMA16(af[i][0], bf[j][1], acc[i][j]);
                acc[i][j] = STEM_MFMA16(af[i][0], bf[j][0], acc[i][j]);
            }
            if (i == 0)
                sstore(cur ^ 1, ra, rb, qcur + 1 < q_end);
            else
                gload(qn, ra, rb);
        }
    };
    // A wavefront whose 64 x 64 outputs lie entirely beyond K or C (half of the last tile of a 320-channel operand: a sixth of
    // the launch's matrix work) only takes part in the staging -- same loads, LDS writes, bias sums and barriers, no LDS reads,
    // no MFMAs.  Decided once per wavefront, outside the woven block.
    const bool dead = k0 + wk0 >= a.K || c0 + wc0 >= a.C;
    auto step_dead = [&](int cur, f32x4 (&ra)[PL], f32x4 (&rb)[PL], int qcur, int qn) {
        sstore(cur ^ 1, ra, rb, qcur + 1 < q_end);
        gload(qn, ra, rb);
    };
    const int q_last = q_end - 1;
    auto clampq = [&](int q) { return q < q_last ? q : q_last; };
    if (q_begin < q_end) {
        gload(q_begin, raA, rbA);
        sstore(0, raA, rbA, true);
        gload(clampq(q_begin + 1), raA, rbA);
        gload(clampq(q_begin + 2), raB, rbB);
    }
    __syncthreads();
    if (!dead) {
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step(0, raA, rbA, q, clampq(q + 3));
            __syncthreads();
            step(1, raB, rbB, q + 1, clampq(q + 4));
            __syncthreads();
        }
        if (q < q_end) {
            step(0, raA, rbA, q, clampq(q + 3));
            __syncthreads();
        }
    } else {
        int q = q_begin;
        for (; q + 1 < q_end; q += 2) {
            step_dead(0, raA, rbA, q, clampq(q + 3));
            __syncthreads();
            step_dead(1, raB, rbB, q + 1, clampq(q + 4));
            __syncthreads();
        }
        if (q < q_end) {
            step_dead(0, raA, rbA, q, clampq(q + 3));
            __syncthreads();
        }
    }

    if (do_bias) {       // 16 pixel rows x 128 channels of per-thread sums -> one row per split
        float *red = reinterpret_cast<float *>(smem);                 // the main loop ended with a barrier
#pragma unroll
        for (int c = 0; c < 8; ++c) red[srow * 128 + sslab * 32 + spc * 8 + c] = bsum[c];
        __syncthreads();
        if (tid < 128 && k0 + tid < a.K) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += red[r * 128 + tid];
            a.bias_part[(size_t)split * a.K + k0 + tid] = v * q_inv(a.dyq);
        }
    }

    // ---- partial sums -> slab [split][tap][K][C]; lane holds column c = .. + (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) --
    float *out = a.dwp + ((size_t)split * a.T + tap) * a.K * a.C;
    const float fac = q_inv(a.xq) * q_inv(a.dyq);                  // the planes hold x * 2^ex and dy * 2^ed
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc0 + j * 32 + lr;
            if (c >= a.C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < a.K) out[(size_t)k * a.C + c] = acc[i][j][r] * fac;
            }
        }
}

template <int S>
__global__ __launch_bounds__(RNT, 2) void wgrad_f16x3_row_kernel(const Wg3Args a)
{
    constexpr int XR = 16 + S - 1, PADX = S / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_c = (a.C + RT_C - 1) / RT_C;
    // Workgroups are dealt to the 8 XCDs round-robin by their linear id.  The launch is one-dimensional and XCD x takes the
    // x-th eighth of the list ordered (split, filter row, tile): the workgroups that stream the same pixel range -- the same dy and
    // x rows -- run on one XCD at about the same time and share them through its L2.
    const int ntile = ((a.K + RT_K - 1) / RT_K) * tiles_c, nper = gridDim.x >> 3;
    const int vid = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
    if (vid >= ntile * S * a.nsplit) return;
    const int split = vid / (ntile * S), vrem = vid - split * ntile * S, frow = vrem / ntile, tile = vrem - frow * ntile;
    const int tk = tile / tiles_c, tc = tile - tk * tiles_c;
    const int k0 = tk * RT_K, c0 = tc * RT_C;
    const int ohw = a.OH * a.OW, nchunks = (a.B * ohw) / PX;                 // OW % 16 == 0: a chunk never leaves its image row
    const int q_begin = split * a.cps, q_end = q_begin + a.cps < nchunks ? q_begin + a.cps : nchunks;
    const int tdy = frow - PADX;

    // ---- operand stream: LDS-DMA, four instructions per wavefront and chunk (1 KiB each, lane l -> byte 16 l of the destination).
    //      The LDS images are XOR-swizzled inside their 256-byte rows, so the lane at row r, position c' fetches piece c' ^ f(r) of
    //      that row.  dy: planes 0 / 1, rows 4 w .. 4 w + 3.  x: rows 4 j .. 4 j + 3 for j = w and w + 4; rows >= XR, pixels outside
    //      the image and slabs outside the channel range take the offset every buffer view rejects (they read as zeros).
    //      Issued as inline assembly: the compiler orders every LDS read after ALL pending LDS-DMA it knows of (vmcnt(0) in front
    //      of the first fragment read of a chunk), which would collapse the three-chunk look-ahead to one; the ring is kept
    //      consistent by the counted s_waitcnt + barrier at the top of a chunk instead. ----
    typedef int rsrc4 __attribute__((ext_vector_type(4)));
    const unsigned long long xa = reinterpret_cast<unsigned long long>(a.xp), da = reinterpret_cast<unsigned long long>(a.dyp);
    const rsrc4 rx = {(int)(unsigned)xa, (int)(unsigned)(xa >> 32), a.xbytes, 0x00020000};
    const rsrc4 rdy = {(int)(unsigned)da, (int)(unsigned)(da >> 32), a.dybytes, 0x00020000};
    auto dma16 = [](const rsrc4 r, unsigned lds_base, int voff) {        // lane l: 16 bytes at voff -> LDS lds_base + 16 l
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(r) : "memory");      // M0 is reserved (not allocatable, not declarable as a clobber); this kernel has no other user of it
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)smem;
    const int drow = 4 * wave + (lane >> 4);
    const int dch = (lane & 15) ^ (((drow & 3) << 2) | ((drow >> 2) & 3));
    const bool dok = k0 / 32 + (dch >> 2) < a.K / 32;
    const int dyv = drow * a.dypix + (k0 / 32 + (dch >> 2)) * SLAB + (dch & 3) * 16;
    int xv[2], xr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (wave + 4 * i) + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const int sl = (ch >> 2) & 1;
        xr[i] = (row < XR && c0 / 32 + sl < a.C / 32) ? row - PADX : -0x100000;      // the pixel's distance from the chunk's first (never valid)
        xv[i] = (row - PADX) * a.xpix + (c0 / 32 + sl) * SLAB + (ch >> 3) * 64 + (ch & 3) * 16;
    }
    // position of the next chunk to fetch (image, row, first column), walked chunk by chunk; past the last chunk it stays where
    // it is (the look-ahead then re-fetches the last chunk into a stage nobody reads)
    int nq = q_begin, nb, noy, nox;
    {
        const int m0 = q_begin * PX;
        nb = m0 / ohw;
        const int rem = m0 - nb * ohw;
        noy = rem / a.OW;
        nox = rem - noy * a.OW;
    }
    auto dma = [&](int stage) {
        const int iy = noy + tdy;
        const bool rowok = iy >= 0 && iy < a.H;
        const unsigned st = __builtin_amdgcn_readfirstlane(lds0 + stage * R_BUF + wave * 1024);
        const int va = dok ? dyv + nq * PX * a.dypix : OOR;
        dma16(rdy, st, va);
        dma16(rdy, st + PLANE, va + 64);
        const int sbase = ((nb * a.H + iy) * a.W + nox) * a.xpix;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ix = nox + xr[i];
            const int vx = (rowok && ix >= 0 && ix < a.W) ? sbase + xv[i] : OOR;
            dma16(rx, st + R_DY + i * 4096, vx);
        }
        const bool adv = nq + 1 < q_end, wrapx = nox + PX == a.OW, wrapy = wrapx && noy + 1 == a.OH;       // branch-free: the
        nq += adv ? 1 : 0;                                                                                   // multiply block stays one
        nox = adv ? (wrapx ? 0 : nox + PX) : nox;                                                            // basic block
        noy = (adv && wrapx) ? (wrapy ? 0 : noy + 1) : noy;
        nb += (adv && wrapy) ? 1 : 0;
    };

    // ---- fragments: as in the per-tap kernel; the x fragment of tap s starts s rows further down, plane 1 at address ^ 128 ----
    const int wk0 = (wave >> 1) * 64, wc0 = (wave & 1) * 32;
    const int g = lane >> 4, cg = g & 1, hh = g >> 1, qq = (lane >> 2) & 3, pp = lane & 3;
    int adrA[2][2], adrX[S][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = 8 * hh + 4 * u + qq;
#pragma unroll
        for (int t = 0; t < 2; ++t) adrA[t][u] = lds_off(row, (wk0 + t * 32) / 8 + 2 * cg + (pp >> 1)) + 8 * (pp & 1);
#pragma unroll
        for (int s = 0; s < S; ++s) adrX[s][u] = R_DY + lds_off(row + s, wc0 / 8 + 2 * cg + (pp >> 1)) + 8 * (pp & 1);
    }
    f32x16 acc[S][2];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][i][r] = 0.f;

    // bias gradient: the workgroups of filter row 0 / channel tile 0 see every dy row of their k tile and pixel range exactly once;
    // each thread re-reads one 16-byte piece per plane from the LDS image (every piece of the 16 x 128 tile is covered once)
    const bool do_bias = a.bias_part != nullptr && frow == 0 && tc == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 4, sslab = (tid >> 2) & 3, spc = tid & 3;
    const int st0 = lds_off(srow, sslab * 4 + spc);

    const bool dead = k0 + wk0 >= a.K || c0 + wc0 >= a.C;
    if (q_begin < q_end) {
        dma(0);
        dma(1);
        dma(2);
    }
    for (int q = q_begin; q < q_end; ++q) {
        const int stage = (q - q_begin) & 3;
        // chunk q has landed (this wavefront's part: all but the 8 instructions of chunks q + 1, q + 2), then everybody's; the
        // barrier also says that nobody still reads stage (q + 3) & 3, chunk q - 1's
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (dead) dma((stage + 3) & 3);
        const unsigned char *Ab = smem + stage * R_BUF;
        if (do_bias) {
            const hp8 p0 = *reinterpret_cast<const hp8 *>(Ab + st0), p1 = *reinterpret_cast<const hp8 *>(Ab + PLANE + st0);
#pragma unroll
            for (int c = 0; c < 8; ++c) bsum[c] += (float)p0[c] + (float)p1[c];
        }
        if (!dead) {
            h16x8 af[2][NPL], bf[2][NPL];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) af[t][pl] = tr_frag(Ab + pl * PLANE, adrA[t][0], adrA[t][1]);
            bf[0][0] = tr_frag(Ab, adrX[0][0], adrX[0][1]);
            bf[0][1] = tr_frag(Ab, adrX[0][0] ^ 128, adrX[0][1] ^ 128);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) {           // the next tap's fragments are on their way while this tap multiplies
                    bf[(s + 1) & 1][0] = tr_frag(Ab, adrX[s + 1][0], adrX[s + 1][1]);
                    bf[(s + 1) & 1][1] = tr_frag(Ab, adrX[s + 1][0] ^ 128, adrX[s + 1][1] ^ 128);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[s][i] = STEM_MFMA16(af[i][1], bf[s & 1][0], acc[s][i]);
                    acc[s][i] = STEM_MFMA16(af[i][0], bf[s & 1][1], acc[s][i]);
                    acc[s][i] = STEM_MFMA16(af[i][0], bf[s & 1][0], acc[s][i]);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                if (s == 0) dma((stage + 3) & 3);          // chunk q + 3, issued in the shadow of the first tap's products
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the re-fetches of the last chunk are still landing in LDS
    __syncthreads();

    if (do_bias) {
        float *red = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int c = 0; c < 8; ++c) red[srow * 128 + sslab * 32 + spc * 8 + c] = bsum[c];
        __syncthreads();
        if (tid < 128 && k0 + tid < a.K) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += red[r * 128 + tid];
            a.bias_part[(size_t)split * a.K + k0 + tid] = v * q_inv(a.dyq);
        }
    }
    if (dead) return;
    const float fac = q_inv(a.xq) * q_inv(a.dyq);
    const int lr = lane & 31, lh = lane >> 5;
    const int c = c0 + wc0 + lr;
    if (c >= a.C) return;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float *out = a.dwp + ((size_t)split * a.T + frow * S + s) * a.K * a.C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < a.K) out[(size_t)k * a.C + c] = acc[s][i][r] * fac;
            }
    }
}

// the filter-row form applies: 3 x 3 / 5 x 5 window with 'same' padding, output rows in whole 16-pixel chunks.  In the bench step
// (same box, ms per step): per-tap form everywhere 14.33-14.36, 5 x 5 layers here 14.08-14.13, 3 x 3 layers too 13.96-14.07
// (isolated the 3 x 3 layers are equal, 31.7 against 31.5 us: fewer, longer workgroups next to the step's other streams).
// 1 x 1 layers take the same kernel with S = 1 (no tap reuse, but the four-stage operand ring instead of a one-chunk look-ahead):
// EPM.2 / EPM.4 25.1 / 21.8 -> 20.9 / 16.0 us alone at equal splits (bit-identical sums), EPM.0 equal.
// stem_tuning_set("wg3_row", 1): per-tap form everywhere; 3: 5 x 5 only; 6: not the 1 x 1 layers.
bool row_form(int OH, int OW, int H, int W, int R, int S, int pad)
{
    const int sel = stem_tuning(STEM_TUNE_WG3_ROW);
    return sel != 1 && R == S && (R == 5 || (R == 3 && sel != 3) || (R == 1 && sel != 3 && sel != 6)) && pad == R / 2 && OH == H && OW == W && OW % PX == 0;
}

int plan_splits(int B, int OH, int OW, int C, int K, int T, bool rows = false, int R = 0)
{
    const int forced = stem_tuning(STEM_TUNE_WG3_SPLIT);      // stem_tuning_set("wg3_split", n): tests / sweeps
    if (rows) {       // workgroups = tiles x filter rows x splits: at most 512 (two per CU), at least R_MINCHUNKS pixel chunks each
        const int nchunks = B * OH * OW / PX, wgs = cdiv(K, RT_K) * cdiv(C, RT_C) * R;
        int s = forced > 0 ? forced : 512 / wgs;
        const int minch = stem_tuning(STEM_TUNE_WG3_MINCH) > 0 ? stem_tuning(STEM_TUNE_WG3_MINCH) : R_MINCHUNKS;      // stem_tuning_set("wg3_minch", n): sweeps
        if (forced <= 0 && s > nchunks / minch) s = nchunks / minch;
        if (s > nchunks) s = nchunks;
        if (s < 1) s = 1;
        return cdiv(nchunks, cdiv(nchunks, s));
    }
    const int nchunks = cdiv(B * OH * OW, PX), tiles = cdiv(K, TK) * cdiv(C, TC) * T;
    int s = forced > 0 ? forced : (512 + tiles / 2) / tiles;       // two workgroups per CU: aim at ~512 workgroups
    if (s < 1) s = 1;
    if (s > nchunks / 16) s = nchunks / 16 > 0 ? nchunks / 16 : 1;
    // one-tap layers have few output tiles: past 32 pixel chunks per workgroup the slabs' write + unpack traffic costs more
    // than the extra workgroups return (sweep: EPM.0 / EPM.2 63 / 42 us at 4-8 splits against 69 / 49 us at 9 / 16)
    if (forced <= 0 && T == 1 && s > nchunks / 32 && nchunks >= 64) s = nchunks / 32;
    const int cps = cdiv(nchunks, s);
    return cdiv(nchunks, cps);
}

}   // namespace

STEM_EXPORT int stem_wgrad_f16x3_splits(int B, int H, int W, int C, int K, int R, int S, int pad)
{
    const int OH = H + 2 * pad - R + 1, OW = W + 2 * pad - S + 1;
    if (OH < 1 || OW < 1) return 0;
    return plan_splits(B, OH, OW, C, K, R * S, row_form(OH, OW, H, W, R, S, pad), R);
}

STEM_EXPORT int stem_conv2d_wgrad_f16x3(const void *xp, const float *xq, int xpix, const void *dyp, const float *dyq, int dypix, float *dwp,
                                         float *bias_part, int B, int H, int W, int C, int K, int R, int S, int pad, int splits, void *stream)
{
    STEM_CHECK_ARG(xp && xq && dyp && dyq && dwp, "stem_conv2d_wgrad_f16x3: null pointer");
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 32 && C % 32 == 0 && K >= 32 && K % 32 == 0 && R >= 1 && S >= 1 && R * S <= MAXTAP && pad >= 0,
                   "stem_conv2d_wgrad_f16x3: C %% 32 == 0, K %% 32 == 0, R*S <= %d (C=%d K=%d R=%d S=%d)", MAXTAP, C, K, R, S);
    const int OH = H + 2 * pad - R + 1, OW = W + 2 * pad - S + 1;
    STEM_CHECK_ARG(OH >= 1 && OW >= 1, "stem_conv2d_wgrad_f16x3: empty output");
    if (xpix == 0) xpix = (C / 32) * SLAB;
    if (dypix == 0) dypix = (K / 32) * SLAB;
    STEM_CHECK_ARG(xpix >= (C / 32) * SLAB && xpix % SLAB == 0 && dypix >= (K / 32) * SLAB && dypix % SLAB == 0,
                   "stem_conv2d_wgrad_f16x3: pixel pitches must be multiples of %d bytes covering the channels", SLAB);
    const size_t xb = (size_t)B * H * W * xpix, db = (size_t)B * OH * OW * dypix;
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && db < 0x7FFFFF00ull, "stem_conv2d_wgrad_f16x3: operand views must stay below 2 GiB");
    const bool rows = row_form(OH, OW, H, W, R, S, pad);
    STEM_CHECK_ARG(splits == plan_splits(B, OH, OW, C, K, R * S, rows, R), "stem_conv2d_wgrad_f16x3: splits must come from stem_wgrad_f16x3_splits");
    Wg3Args a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.dyp = dyp; a.xq = xq; a.dyq = dyq; a.dwp = dwp; a.bias_part = bias_part; a.xpix = xpix; a.dypix = dypix;
    a.B = B; a.H = H; a.W = W; a.C = C; a.K = K; a.OH = OH; a.OW = OW; a.T = R * S;
    a.xbytes = (int)xb; a.dybytes = (int)db;
    a.stride = 1;
    const int nchunks = cdiv(B * OH * OW, PX);
    a.nsplit = splits;
    a.cps = cdiv(nchunks, splits);
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            a.dy[r * S + s] = (signed char)(r - pad);
            a.dx[r * S + s] = (signed char)(s - pad);
        }
    if (rows) {
        static bool attr_rows = false;
        if (!attr_rows) {
            (void)hipFuncSetAttribute((const void *)wgrad_f16x3_row_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
            (void)hipFuncSetAttribute((const void *)wgrad_f16x3_row_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
            (void)hipFuncSetAttribute((const void *)wgrad_f16x3_row_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
            attr_rows = true;
        }
        const dim3 grid(cdiv(cdiv(K, RT_K) * cdiv(C, RT_C) * R * splits, 8) * 8);
        if (R == 1)
            hipLaunchKernelGGL(wgrad_f16x3_row_kernel<1>, grid, dim3(RNT), R_LDS, (hipStream_t)stream, a);
        else if (R == 3)
            hipLaunchKernelGGL(wgrad_f16x3_row_kernel<3>, grid, dim3(RNT), R_LDS, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(wgrad_f16x3_row_kernel<5>, grid, dim3(RNT), R_LDS, (hipStream_t)stream, a);
        STEM_LAUNCH_CHECK("stem_conv2d_wgrad_f16x3");
        return 0;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)wgrad_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    const dim3 grid(cdiv(K, TK) * cdiv(C, TC), R * S, splits);
        hipLaunchKernelGGL(wgrad_f16x3_kernel, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    STEM_LAUNCH_CHECK("stem_conv2d_wgrad_f16x3");
    return 0;
}

/* Weight gradient of a STRIDED layer from planes operands (per-tap form; HE.2 / HE.4 and -- with the operands' roles swapped --
 * HD.0 / HD.2: spatiotemporalpriors.py:814-829, torch autograd):
 *   dW[k][c][r][s] = sum over coarse pixels (b, oy, ox) of g[b, oy, ox][k] * f[b, oy * stride + r - pad, ox * stride + s - pad][c]
 * g (`dyp`, K channels) lives on the coarse grid OH x OW = ((H + 2 pad - R) / stride + 1) x ..., f (`xp`, C channels) on the fine grid
 * H x W.  nn.Conv2d: f = the layer's input, g = the gradient of its output, slabs [t][K][C].  nn.ConvTranspose2d (weight [Cin][Cout]):
 * g = the layer's INPUT (K = Cin), f = the gradient of its OUTPUT (C = Cout), slabs [t][Cin][Cout] (STEM_UNPACK_DECONV). */
STEM_EXPORT int stem_wgrad_f16x3_strided_splits(int B, int H, int W, int C, int K, int R, int S, int stride, int pad)
{
    if (stride < 1) return 0;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    if (OH < 1 || OW < 1) return 0;
    return plan_splits(B, OH, OW, C, K, R * S);
}

STEM_EXPORT int stem_conv2d_wgrad_f16x3_strided(const void *xp, const float *xq, int xpix, const void *dyp, const float *dyq, int dypix, float *dwp,
                                                 float *bias_part, int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int splits,
                                                 void *stream)
{
    STEM_CHECK_ARG(xp && xq && dyp && dyq && dwp, "stem_conv2d_wgrad_f16x3_strided: null pointer");
    STEM_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && C >= 32 && C % 32 == 0 && K >= 32 && K % 32 == 0 && R >= 1 && S >= 1 && R * S <= MAXTAP && pad >= 0 &&
                   stride >= 1 && stride <= 4,
                   "stem_conv2d_wgrad_f16x3_strided: C %% 32 == 0, K %% 32 == 0, R*S <= %d, 1 <= stride <= 4 (C=%d K=%d R=%d S=%d stride=%d)", MAXTAP, C, K, R, S, stride);
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
    STEM_CHECK_ARG(OH >= 1 && OW >= 1, "stem_conv2d_wgrad_f16x3_strided: empty output");
    if (xpix == 0) xpix = (C / 32) * SLAB;
    if (dypix == 0) dypix = (K / 32) * SLAB;
    STEM_CHECK_ARG(xpix >= (C / 32) * SLAB && xpix % SLAB == 0 && dypix >= (K / 32) * SLAB && dypix % SLAB == 0,
                   "stem_conv2d_wgrad_f16x3_strided: pixel pitches must be multiples of %d bytes covering the channels", SLAB);
    const size_t xb = (size_t)B * H * W * xpix, db = (size_t)B * OH * OW * dypix;
    STEM_CHECK_ARG(xb < 0x7FFFFF00ull && db < 0x7FFFFF00ull, "stem_conv2d_wgrad_f16x3_strided: operand views must stay below 2 GiB");
    STEM_CHECK_ARG(splits == plan_splits(B, OH, OW, C, K, R * S), "stem_conv2d_wgrad_f16x3_strided: splits must come from stem_wgrad_f16x3_strided_splits");
    Wg3Args a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.dyp = dyp; a.xq = xq; a.dyq = dyq; a.dwp = dwp; a.bias_part = bias_part; a.xpix = xpix; a.dypix = dypix;
    a.B = B; a.H = H; a.W = W; a.C = C; a.K = K; a.OH = OH; a.OW = OW; a.T = R * S;
    a.xbytes = (int)xb; a.dybytes = (int)db;
    a.stride = stride;
    const int nchunks = cdiv(B * OH * OW, PX);
    a.nsplit = splits;
    a.cps = cdiv(nchunks, splits);
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
            a.dy[r * S + s] = (signed char)(r - pad);
            a.dx[r * S + s] = (signed char)(s - pad);
        }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)wgrad_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done = true;
    }
    const dim3 grid(cdiv(K, TK) * cdiv(C, TC), R * S, splits);
    hipLaunchKernelGGL(wgrad_f16x3_kernel, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    STEM_LAUNCH_CHECK("stem_conv2d_wgrad_f16x3_strided");
    return 0;
}
