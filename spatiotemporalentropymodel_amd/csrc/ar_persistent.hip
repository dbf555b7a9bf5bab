// Raster-order decoding of ONE image (compressai/models/spatiotemporalpriors.py:1015-1054) as a single persistent kernel.
//
// stem_ar_decode_image (ar.hip) pays, per latent position, four DEPENDENT dispatches (~6 us each on the queue), a stream
// synchronisation and the host coder: ~35 us x 8160 positions = 0.29 s per 1080p P frame, whatever the arithmetic costs.
// Here the whole loop is one launch (0.12-0.13 s per 1080p P frame, bit-identical):
//   * 32 workgroups of 512 threads stay resident -- one per CU of ONE XCD (every candidate reads its XCC id; the first arrival
//     picks the XCD, the first 32 workgroups of that XCD stay, all others exit), so that everything they exchange lives in one L2
//     -- and walk through the positions together; global wavefront g owns output rows g, g + 256, ... of all four products
//     (context window -> ctx, EPM.0, EPM.2, EPM.4) for the whole image and keeps their weights in registers / LDS;
//   * values travel between workgroups, and between device and host, as 8-byte words {value, position + 1} written by ONE store
//     each: the consumer spins on the words it needs until they carry its position's tag.  No barriers, no separate flags;
//   * only the part of a position's arithmetic that depends on the symbol just decoded is on the dependent path: the rest is
//     accumulated ahead, as lane partials, while the host decodes (see below);
//   * the host takes part through a pinned mailbox: workgroup 0 forwards the CDF indexes, the host thread (spinning inside
//     stem_ar_decode_image_persistent) runs the rANS symbol decoder and posts the symbols, workgroup 0 commits
//     y_hat = symbol + mean to the latent buffer and to a two-pixel ring of tagged words, which releases the other workers.
//     Every wait is bounded; a timeout raises an abort word that ends the kernel (the caller then decodes with the loop).
//
// Arithmetic: every output row is one wavefront's dot product in the order of gemv3_decode_kernel (segments in order, columns
// lane * 4 + 256 t ascending, xor-shuffle reduction, bias, leaky ReLU), compiled without FMA contraction like ar.hip -- the
// entropy parameters, hence symbols and bytes, are those of the per-position loop and of the encoder.
#include <chrono>
#include <mutex>
#include <vector>

#include "stem_common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWG_DEFAULT = 32;         // resident workers (512 threads each, one per CU of an XCD): 256 wavefronts, up to 3 output rows per wavefront
constexpr int GRID = 1024;              // candidates: 128 per XCD
constexpr int XMAX = 2304;              // longest input vector (the 12-tap context window at M = 192)
constexpr long SPIN_LIMIT = 4000000;    // ~1-2 s of polling

struct ArpArgs {
    const float *w_ctx, *b_ctx, *w0, *b0, *w1, *b1, *w2, *b2;
    int ld_ctx, ld0, n0, ld1, n1, ld2;
    float *buf;
    int H, W, M, pad;
    const float *tp, *hp;
    float *ctx, *h1, *h2, *gp;          // device scratch: [2M] [n0] [n1] [2M]
    const float *table;
    int T;
    float bound, slope;
    int *mail;                          // pinned: [16] host abort (-1); 8-byte words {value, position + 1}: [32 + slot * 2M ..] idx, [32 + 2 * 2M + slot * 2M ..] sym (slot = position parity)
    int nwg;                            // workers
    int want_xcc;                       // -1: the first candidate to arrive picks the XCD; 0..7: this one (several images at once: one XCD each)
    long long *words;                   // device: tagged 8-byte words {value, position + 1}: ctx [2M] | h1 [n0] | h2 [n1] | gp [2M] | idx [M] | pixel ring [2][M]
    float *dbg;                         // experiments build: [position][2M + n0 + n1 + 2M] copies of ctx | h1 | h2 | gp (null: off)
    int *dev;                           // device, one 128-byte line per word group: [0] worker tickets, [1] chosen XCC (-1), [96] abort;
                                        // [32..] the experiments build's timers
};

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }      // HW_REG_XCC_ID[3:0]

// plain-memory accesses that must not be served from this CU's L1 (the latent buffer other workgroups read next image row)
__device__ inline void st_agent(float *p, float v) { *reinterpret_cast<volatile float *>(p) = v; }

// ---- what a position costs, and what this kernel keeps off that path ---------------------------------------------------------------
// The first version (round 3) streamed all 9.7 MB of fp32 weights from the memory side for every position: 7-11 us per product,
// 0.49 s per 1080p frame.  Of the four products only a part depends on the symbol that has just been decoded:
//   ctx  = b_c + W_c . (row h-2: 5 pixels | row h-1: 5 pixels | row h: pixels w-2, w-1)      only the LAST segment is new
//   h1   = lrelu(b_0 + W_0 . (tp | hp | ctx))                                                  only the ctx segment is new
//   h2, gp                                                                                     entirely
// and a wavefront's dot product accumulates its segments IN ORDER (lane partial sums over columns lane * 4 + 256 t, then the
// xor-shuffle reduction): the lane partials after the leading segments are a well-defined intermediate state.  So
//   * while the host decodes the symbols of position p, every wavefront computes the lane partials of position p + 1 over the
//     segments that are already known (rows above: complete since the previous image row; tp, hp: inputs);
//   * once the symbols are in, it CONTINUES those sums with the new segment -- same additions in the same order, hence the same
//     floats as gemv3_decode_kernel (ar.hip) and as the encoder, bit for bit;
//   * each global wavefront g owns output rows g, g + 256, ... of every product for the whole image and keeps the weights it needs
//     on the dependent path in REGISTERS (148 VGPRs at M = 192: the last segment of its ctx rows, its EPM.0 / EPM.2 / EPM.4 rows)
//     and the weights of the rows-above part of its ctx rows in LDS (16 KB per wavefront): nothing but the 384..768-float
//     vectors handed from product to product moves on the dependent path.
// 32 workgroups of 512 threads (one per CU of one XCD, 256 wavefronts); hand-overs by tagged words (further down).
constexpr int NT = 512, NWAVES = 256;
constexpr int RC = 2, R0 = 3, R1 = 3, R2 = 2;           // rows per wavefront: ctx / gp (<= 512 rows), EPM.0 and EPM.2 (<= 768 rows)
constexpr int TCA = 8, TCL = 2, T0A = 4, T0C = 2, T1 = 3, T2 = 3;      // 256-column steps: ctx rows-above (2 x 4), ctx left, EPM.0 tp|hp, EPM.0 ctx, EPM.2, EPM.4
constexpr int WL_FLOATS = (NT / 64) * RC * TCA * 64 * 4;               // LDS image of the rows-above ctx weights: 32768 floats
constexpr int XA_FLOATS = 2048, XP_FLOATS = 1024, XV_FLOATS = 768;     // staging: rows-above window (10 M) and tp | hp for the look-ahead; the vector of the current product
constexpr int ARP_LDS = (WL_FLOATS + XA_FLOATS + XP_FLOATS + XV_FLOATS) * 4;

__device__ inline float dot4(const f32x4 xv, const f32x4 wv) { return xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3]; }
// (a tagged word's value goes through a scalar before it is reinterpreted: hipcc of ROCm 7.2 reads the wrong element when
// __builtin_bit_cast is applied directly to an element of an ext-vector -- tools/debug/probe/vec_even_elements.hip)

// global -> LDS.  COHERENT: data other workgroups write during the launch (the latent buffer, the vectors handed from product to
// product) -- `sc1` loads, which are served by the L2 whatever this CU's L1 holds.  Round 3's kernel read them with plain loads after
// `buffer_inv sc0` and was right only because the 150 KB of weights it streamed per position had evicted every line by then: with the
// weights resident the L1 keeps last position's lines and `buffer_inv sc0` does not drop them (tools/debug/arp_probe.py).
template <bool COHERENT>
__device__ inline void stage512(float *dst, const float *src, int n)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    for (int i = threadIdx.x * 4; i < n; i += NT * 4)
        *reinterpret_cast<f32x4 *>(dst + i) = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, COHERENT ? 16 : 0));
}

__device__ inline float wave_sum_xor(float acc)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

__global__ __launch_bounds__(NT, 1) void ar_decode_persistent_kernel(const ArpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds, *xa = lds + WL_FLOATS, *xp = xa + XA_FLOATS, *xv = xp + XP_FLOATS;
    __shared__ int role;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *abort_w = a.dev + 96;
    if (tid == 0) {
        // worker selection: the first workgroup to arrive fixes the XCD, the first NWG workgroups of that XCD are the workers
        const int me = xcc_id();
        int chosen = a.want_xcc;
        if (chosen < 0) {
            int expected = -1;
            __hip_atomic_compare_exchange_strong(a.dev + 1, &expected, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chosen = __hip_atomic_load(a.dev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int r = -1;
        if (chosen == me) {
            r = __hip_atomic_fetch_add(a.dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r >= a.nwg) r = -1;
        }
        role = r;
    }
    __syncthreads();
    const int wg = role;
    if (wg < 0) return;

    const int M = a.M, P = 2 * M, Wp = a.W + 2 * a.pad, N = a.H * a.W;
    const int g = wg * (NT / 64) + wave;                    // global wavefront: rows g, g + 256, ... of every product
    const int l4 = lane * 4;
    volatile int *host_abort = a.mail + 16;             // the host's only flag: -1 = give up

    // ---- 256-column steps of the products' segments (the same for every row; step t of a segment covers its columns
    // [256 t, 256 t + 256): `left` columns remain from there, a lane takes part while lane * 4 < left).  Index arithmetic only, so
    // that the weight arrays below are indexed by compile-time constants and stay in registers.
    const int tpP = a.tp ? P : 0;
    auto ca_woff = [&](int t) { return (t >> 2) * 5 * M + (t & 3) * 256; };        // ctx, rows above: 2 segments x <= 4 steps; x offset = weight offset
    auto ca_left = [&](int t) { return 5 * M - (t & 3) * 256; };
    auto cl_left = [&](int t) { return 2 * M - t * 256; };                           // ctx, this row: weight offset 10 M + 256 t, x offset 256 t
    auto ea_woff = [&](int t) { return (t >> 1) * tpP + (t & 1) * 256; };            // EPM.0 over tp | hp: 2 segments x <= 2 steps (tp may be absent)
    auto ea_left = [&](int t) { return (t >> 1) == 0 && !a.tp ? 0 : P - (t & 1) * 256; };
    auto ec_left = [&](int t) { return P - t * 256; };                               // EPM.0 over ctx: weight offset tpP + P + 256 t
    auto e1_left = [&](int t) { return a.n0 - t * 256; };
    auto e2_left = [&](int t) { return a.n1 - t * 256; };

    // ---- this wavefront's weights: registers for everything behind the new symbol, LDS for the rows-above part of ctx ---------------
    auto wload = [&](const float *Wm, int ldw, int n, int N_, int woff, int left) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N_ && l4 < left) v = *reinterpret_cast<const f32x4 *>(Wm + (size_t)n * ldw + woff + l4);
        return v;
    };
    f32x4 wcl[RC][TCL], w0c[R0][T0C], w1[R1][T1], w2[R2][T2];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
#pragma unroll
        for (int t = 0; t < TCL; ++t) wcl[r][t] = wload(a.w_ctx, a.ld_ctx, g + NWAVES * r, P, 10 * M + t * 256, cl_left(t));
#pragma unroll
        for (int t = 0; t < TCA; ++t)
            *reinterpret_cast<f32x4 *>(wl + (((wave * RC + r) * TCA + t) * 64 + lane) * 4) = wload(a.w_ctx, a.ld_ctx, g + NWAVES * r, P, ca_woff(t), ca_left(t));
    }
#pragma unroll
    for (int r = 0; r < R0; ++r) {
#pragma unroll
        for (int t = 0; t < T0C; ++t) w0c[r][t] = wload(a.w0, a.ld0, g + NWAVES * r, a.n0, tpP + P + t * 256, ec_left(t));
    }
#pragma unroll
    for (int r = 0; r < R1; ++r)
#pragma unroll
        for (int t = 0; t < T1; ++t) w1[r][t] = wload(a.w1, a.ld1, g + NWAVES * r, a.n1, t * 256, e1_left(t));
#pragma unroll
    for (int r = 0; r < R2; ++r)
#pragma unroll
        for (int t = 0; t < T2; ++t) w2[r][t] = wload(a.w2, a.ld2, g + NWAVES * r, P, t * 256, e2_left(t));
    // biases of this wavefront's rows (lane 0 finishes a row)
    float bc[RC], b0v[R0], b1v[R1], b2v[R2];
#pragma unroll
    for (int r = 0; r < RC; ++r) bc[r] = g + NWAVES * r < P ? a.b_ctx[g + NWAVES * r] : 0.f;
#pragma unroll
    for (int r = 0; r < R0; ++r) b0v[r] = g + NWAVES * r < a.n0 ? a.b0[g + NWAVES * r] : 0.f;
#pragma unroll
    for (int r = 0; r < R1; ++r) b1v[r] = g + NWAVES * r < a.n1 ? a.b1[g + NWAVES * r] : 0.f;
#pragma unroll
    for (int r = 0; r < R2; ++r) b2v[r] = g + NWAVES * r < P ? a.b2[g + NWAVES * r] : 0.f;

    const float tb = lane < a.T - 1 ? a.table[lane] : 0.f;          // this lane's entry of the scale table
    // lane partials of position q over the segments that do not depend on the symbols still to come: ctx over the two rows above,
    // EPM.0 over tp | hp.  Everything it reads has been final for at least a position (the rows above: for an image row).
    float cpart[RC], epart[R0];
    auto lookahead = [&](int q) {
        const int qh = q / a.W, qw = q - qh * a.W;
        const float *r0 = a.buf + ((size_t)qh * Wp + qw) * M;
        __syncthreads();                                     // the previous users of xa / xp are done
        stage512<true>(xa, r0, 5 * M);
        stage512<true>(xa + 5 * M, r0 + (size_t)Wp * M, 5 * M);
        if (a.tp) stage512<false>(xp, a.tp + (size_t)q * P, P);
        stage512<false>(xp + tpP, a.hp + (size_t)q * P, P);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < TCA; ++t)
                if (l4 < ca_left(t))
                    acc += dot4(*reinterpret_cast<const f32x4 *>(xa + ca_woff(t) + l4), *reinterpret_cast<const f32x4 *>(wl + (((wave * RC + r) * TCA + t) * 64 + lane) * 4));
            cpart[r] = acc;
            __builtin_amdgcn_sched_barrier(0);               // one row's operands at a time: the registers belong to the resident weights
        }
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            // these weights (2.4 MB for all rows at M = 192) are not on the dependent path: they stay in the XCD's L2 and are read
            // again for every position, which leaves the registers to the weights that are
            f32x4 w0a[T0A];
            const float *w0p = a.w0;
            asm volatile("" ::: "memory");           // the loads below must not be hoisted out of the position loop
#pragma unroll
            for (int t = 0; t < T0A; ++t) w0a[t] = wload(w0p, a.ld0, g + NWAVES * r, a.n0, ea_woff(t), ea_left(t));
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < T0A; ++t)
                if (l4 < ea_left(t)) acc += dot4(*reinterpret_cast<const f32x4 *>(xp + ea_woff(t) + l4), w0a[t]);
            epart[r] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();                                         // the LDS weight image is complete
    lookahead(0);

    // ---- tagged words: everything the workgroups hand to each other (and the host mailbox) is an 8-byte word {value, position + 1},
    // written by ONE 8-byte store.  A consumer loads the words it needs and spins until they carry its position's tag: the data is
    // its own flag -- one L2 round trip from producer to consumer, where a grid barrier (stores acknowledged, atomic arrival, polls,
    // then the loads) took four.  Overwriting is safe without a second buffer: nobody starts position p + 1 before the pixel of
    // position p is committed, which needs the host's symbols, which need ALL indexes of position p -- every product of p is done.
    long long *ctxw = a.words, *h1w = ctxw + P, *h2w = h1w + a.n0, *gpw = h2w + a.n1, *idxw = gpw + P, *pixw = idxw + M;      // pixw: [2][M] by position parity
    auto pack = [](float v, int tag) { return ((long long)tag << 32) | (unsigned)__builtin_bit_cast(int, v); };
    auto packi = [](int v, int tag) { return ((long long)tag << 32) | (unsigned)v; };
    auto put = [&](long long *dst, long long word) { __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    bool dead = false;                                        // a bounded wait ran out (or somebody else's did): leave
    auto give_up = [&](long &spins) {
        if ((++spins & 255) != 0) return false;
        if (spins > SPIN_LIMIT || __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_fetch_add(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    };
    // a tagged vector into LDS, the workgroup's 512 threads one word each (and again 512 further on): every thread spins on ITS
    // words until they carry `tag`.  (All 256 wavefronts polling whole vectors -- 1.5 MB per round through one L2 -- took 3.4-4 us
    // per hand-over; one copy per workgroup is 32 x 6 KB.)
    auto xwait = [&](const long long *src, int n, int tag) {
        __syncthreads();                                     // the previous readers of xv are done
        for (int c = tid; c < n; c += NT) {
            long spins = 0;
            long long v;
            for (;;) {
                v = __hip_atomic_load(src + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v >> 32) == tag) break;
                if (give_up(spins)) {
                    dead = true;
                    break;
                }
            }
            xv[c] = __builtin_bit_cast(float, (int)v);
        }
        dead = __syncthreads_or(dead);                       // a workgroup leaves as one
    };
    auto xget = [&](int t, int n) {                          // columns 256 t + 4 lane .. + 3 (zeros beyond the vector)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t * 256 + l4 < n) v = *reinterpret_cast<const f32x4 *>(xv + t * 256 + l4);
        return v;
    };
#ifdef STEM_EXPERIMENTS
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    const long long cyc0 = __builtin_readcyclecounter(), wall0 = tlast;
#define ARP_MARK(i)                                  \
    do {                                             \
        const long long tn_ = wall_clock64();        \
        tacc[i] += tn_ - tlast;                      \
        tlast = tn_;                                 \
    } while (0)
#else
#define ARP_MARK(i)
#endif
    // workgroup 0, one thread per channel: the pixel of position pp = symbol + mean, once the host has posted the symbol
    auto commit_pixel = [&](int pp) {
        const int ph = pp / a.W, pw = pp - ph * a.W;
        const long long *symw = reinterpret_cast<const long long *>(a.mail + 32 + 2 * P) + (size_t)(pp & 1) * M;
        float *pix = a.buf + ((size_t)(ph + a.pad) * Wp + (pw + a.pad)) * M;
        for (int c = tid; c < M; c += NT) {
            long spins = 0;
            long long sv, mv;
            for (;;) {                                        // the mean first: it has been there since EPM.4, the symbol is what takes time
                mv = __hip_atomic_load(gpw + M + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(mv >> 32) == pp + 1) break;
                if (give_up(spins)) {
                    dead = true;
                    return;
                }
            }
            spins = 0;
            for (;;) {
                sv = __hip_atomic_load(symw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((int)(sv >> 32) == pp + 1) break;
                if ((++spins & 63) == 0 && (spins > SPIN_LIMIT / 16 || __hip_atomic_load((int *)host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 0 ||
                                            __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_fetch_add(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const float y = (float)(int)sv + __builtin_bit_cast(float, (int)mv);
            st_agent(pix + c, y);
            put(pixw + (size_t)(pp & 1) * M + c, pack(y, pp + 1));
        }
    };
    for (int p = 0; p < N; ++p) {
        const int h = p / a.W, w = p - h * a.W, tag = p + 1;
        if (p > 0 && wg == 0) commit_pixel(p - 1);
        if (dead) return;
        // ---- ctx: the partials of the rows above, continued over pixels (h, w-2), (h, w-1) -- positions p - 2 and p - 1 of the pixel
        // ring (zeros at the left border).  Every wavefront waits for position p - 1's pixel here whether or not the window holds it:
        // that wait is what keeps the workgroups within one position of each other.
        {
            __syncthreads();
            for (int c = tid; c < 2 * M; c += NT) {
                float v = 0.f;
                if (p > 0) {
                    const int newer = c >= M ? 1 : 0, cc = c - newer * M, pp = p - 2 + newer;               // position of that pixel
                    if (newer || w >= 2) {                                                                    // p - 2 exists whenever w >= 2
                        long spins = 0;
                        long long wv;
                        for (;;) {
                            wv = __hip_atomic_load(pixw + (size_t)(pp & 1) * M + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((int)(wv >> 32) == pp + 1) break;
                            if (give_up(spins)) {
                                dead = true;
                                break;
                            }
                        }
                        if (newer ? w >= 1 : w >= 2) v = __builtin_bit_cast(float, (int)wv);
                    }
                }
                xv[c] = v;
            }
            if (__syncthreads_or(dead)) return;
            f32x4 x[TCL];
#pragma unroll
            for (int t = 0; t < TCL; ++t) x[t] = xget(t, 2 * M);
            ARP_MARK(0);
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const int n = g + NWAVES * r;
                float acc = cpart[r];
#pragma unroll
                for (int t = 0; t < TCL; ++t)
                    acc += dot4(x[t], wcl[r][t]);
                acc = wave_sum_xor(acc);
                if (lane == 0 && n < P) put(ctxw + n, pack(acc + bc[r], tag));
            }
        }
        ARP_MARK(1);
        // ---- h1 = lrelu(b_0 + W_0 . (tp | hp | ctx)): the partials over tp | hp, continued over ctx
        {
            xwait(ctxw, P, tag);
            if (dead) return;
            f32x4 x[T0C];
#pragma unroll
            for (int t = 0; t < T0C; ++t) x[t] = xget(t, P);
            ARP_MARK(2);
#ifdef STEM_EXPERIMENTS
            if (a.dbg && p == 0) {           // what this wavefront multiplies at position 0: its ctx columns and its look-ahead partials
                float *e = a.dbg + (size_t)N * (2 * P + a.n0 + a.n1) + (size_t)g * (256 + 64 * R0);
                for (int c = 0; c < 4; ++c) e[l4 + c] = x[0][c];
                for (int r = 0; r < R0; ++r) e[256 + 64 * r + lane] = epart[r];
            }
#endif
#pragma unroll
            for (int r = 0; r < R0; ++r) {
                const int n = g + NWAVES * r;
                float acc = epart[r];
#pragma unroll
                for (int t = 0; t < T0C; ++t)
                    acc += dot4(x[t], w0c[r][t]);
                acc = wave_sum_xor(acc);
                if (lane == 0 && n < a.n0) {
                    float v = acc + b0v[r];
                    v = v > 0.f ? v : v * a.slope;
                    put(h1w + n, pack(v, tag));
                }
            }
        }
        ARP_MARK(3);
        // ---- h2 = lrelu(b_1 + W_1 . h1)
        {
            xwait(h1w, a.n0, tag);
            if (dead) return;
            f32x4 x[T1];
#pragma unroll
            for (int t = 0; t < T1; ++t) x[t] = xget(t, a.n0);
            ARP_MARK(4);
#pragma unroll
            for (int r = 0; r < R1; ++r) {
                const int n = g + NWAVES * r;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < T1; ++t)
                    acc += dot4(x[t], w1[r][t]);
                acc = wave_sum_xor(acc);
                if (lane == 0 && n < a.n1) {
                    float v = acc + b1v[r];
                    v = v > 0.f ? v : v * a.slope;
                    put(h2w + n, pack(v, tag));
                }
            }
        }
        ARP_MARK(5);
        // ---- gp = b_2 + W_2 . h2 (scales | means) and the scales' CDF indexes
        {
            xwait(h2w, a.n1, tag);
            if (dead) return;
            f32x4 x[T2];
#pragma unroll
            for (int t = 0; t < T2; ++t) x[t] = xget(t, a.n1);
            ARP_MARK(4);
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                const int n = g + NWAVES * r;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < T2; ++t)
                    acc += dot4(x[t], w2[r][t]);
                acc = wave_sum_xor(acc);                 // every lane holds the sum
                const float v = acc + b2v[r];
                // index = #(table[:-1] < scale) (entropy_models.py:598-604): every lane compares the scale with ITS table entry
                const float sc = fmaxf(v, a.bound);
                int k = a.T - 1 - __builtin_popcountll(__ballot(lane < a.T - 1 && sc <= tb));
                for (int t = 64; t < a.T - 1; ++t) k -= (sc <= a.table[t]) ? 1 : 0;            // tables beyond 65 levels
                if (lane == 0 && n < P) {
                    put(gpw + n, pack(v, tag));
                    if (n < M) put(idxw + n, packi(k, tag));
                }
            }
        }
        ARP_MARK(6);
        // ---- workgroup 0 forwards the indexes to the host mailbox (written lane by lane from the products they were separate small
        // PCIe writes from 24 CUs: ~20 us per position)
        if (wg == 0) {
            long long *dst = reinterpret_cast<long long *>(a.mail + 32) + (size_t)(p & 1) * M;
            for (int c = tid; c < M; c += NT) {
                long spins = 0;
                long long v;
                for (;;) {
                    v = __hip_atomic_load(idxw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v >> 32) == tag) break;
                    if (give_up(spins)) return;
                }
                __hip_atomic_store(dst + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        ARP_MARK(7);
#ifdef STEM_EXPERIMENTS
        if (a.dbg && wg == 1) {
            // (after the look-ahead's first barrier below nobody of this workgroup is still multiplying; the other workgroups cannot
            // start the next position before the host has answered)
            float *d = a.dbg + (size_t)p * (2 * P + a.n0 + a.n1);
            for (int c = tid; c < 2 * P + a.n0 + a.n1; c += NT) {
                long spins = 0;
                long long v;
                for (;;) {
                    v = __hip_atomic_load(ctxw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v >> 32) == tag || give_up(spins)) break;
                }
                d[c] = __builtin_bit_cast(float, (int)v);
            }
        }
#endif
        // while the host decodes: the known part of the next position
        if (p + 1 < N) lookahead(p + 1);
    }
#ifdef STEM_EXPERIMENTS
    if (wg == 0 && tid == 0)
    {
        for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(a.dev + 32)[i] = tacc[i];
        reinterpret_cast<long long *>(a.dev + 32)[8] = __builtin_readcyclecounter() - cyc0;
        reinterpret_cast<long long *>(a.dev + 32)[9] = wall_clock64() - wall0;
    }
#endif
    if (wg == 0) commit_pixel(N - 1);                         // the last position's symbols
}

#ifdef STEM_EXPERIMENTS
float *g_arp_dbg = nullptr;
#endif
struct ArpState {
    int *pinned = nullptr, *dev = nullptr;
    long long *words = nullptr;
    size_t pinned_ints = 0, nwords = 0;
    int device = -1;               // the HIP device `dev` / `words` were allocated on
};
// start values of the device flags: word 1 = -1 ("no XCD claimed yet"), everything else 0.  Constant: eight host threads copy from it at once
const int kArpInit[192] = {0, -1};
std::once_flag g_arp_attr_once;
thread_local ArpState g_arp;
thread_local int g_arp_want_xcc = -1;

}   // namespace

// Calling thread: its next images run on XCD `xcc` (0..7; -1: whichever candidate arrives first picks).  For decoding several images
// at once from several host threads -- one XCD, one stream, one host thread each: the kernel needs an XCD to itself.
STEM_EXPORT int stem_ar_decode_image_persistent_prefer_xcc(int xcc)
{
    STEM_CHECK_ARG(xcc >= -1 && xcc < 8, "stem_ar_decode_image_persistent_prefer_xcc: XCD %d (0..7, or -1)", xcc);
    g_arp_want_xcc = xcc;
    return 0;
}

// 1 if the persistent kernel's fixed row -> wavefront map holds a model of these widths (M latent channels, n0 / n1 = EPM.0 / EPM.2 outputs)
STEM_EXPORT int stem_ar_decode_image_persistent_supported(int M, int n0, int n1)
{
    return M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && 5 * M <= 1024 && 10 * M <= XA_FLOATS && 4 * M <= XP_FLOATS && 2 * M <= NWAVES * RC &&
           n0 <= NWAVES * R0 && n1 <= NWAVES * R1 && n0 <= XV_FLOATS && n1 <= XV_FLOATS && 2 * M <= XV_FLOATS;
}

// C ABI: same contract as stem_ar_decode_image (ar.hip) minus the caller's mailboxes (kept here, pinned, per host thread).
STEM_EXPORT int stem_ar_decode_image_persistent(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                                                const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                                                float *buf, int H, int W, int M, int pad, const float *tp, const float *hp, float *ctx, float *h1,
                                                float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                                                stem_symbol_decoder_fn decode, void *dec, const int32_t *cdfs, int ncdf, int cdf_stride,
                                                const int32_t *sizes, const int32_t *offsets, void *stream)
{
    STEM_CHECK_ARG(w_ctx && b_ctx && w0 && b0 && w1 && b1 && w2 && b2 && buf && hp && ctx && h1 && h2 && gp && table && decode,
                   "stem_ar_decode_image_persistent: null pointer");
    STEM_CHECK_ARG(H > 0 && W > 0 && M > 0 && M % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && ld_ctx % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 &&
                   ld2 % 4 == 0 && T >= 1 && pad == 2, "stem_ar_decode_image_persistent: bad sizes");
    STEM_CHECK_ARG(stem_ar_decode_image_persistent_supported(M, n0, n1),
                   "stem_ar_decode_image_persistent: built for M <= 204 and EPM widths <= 768 (M=%d n0=%d n1=%d)", M, n0, n1);
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, N = H * W;
    const size_t need = 32 + 4 * (size_t)P;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) {
        stem_set_error("stem_ar_decode_image_persistent: no current device");
        return -2;
    }
    if (g_arp.device != device) {                  // this host thread last decoded on another GPU: its device buffers live there
        if (g_arp.dev) (void)hipFree(g_arp.dev);
        if (g_arp.words) (void)hipFree(g_arp.words);
        g_arp.dev = nullptr;
        g_arp.words = nullptr;
        g_arp.nwords = 0;
        g_arp.device = device;
    }
    if (g_arp.pinned_ints < need) {
        if (g_arp.pinned) (void)hipHostFree(g_arp.pinned);
        g_arp.pinned = nullptr;
        if (hipHostMalloc((void **)&g_arp.pinned, need * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            stem_set_error("stem_ar_decode_image_persistent: cannot allocate the pinned mailboxes");
            return -2;
        }
        g_arp.pinned_ints = need;
    }
    if (!g_arp.dev && hipMalloc((void **)&g_arp.dev, (192 + XMAX) * sizeof(int)) != hipSuccess) {
        stem_set_error("stem_ar_decode_image_persistent: cannot allocate the device flags");
        return -2;
    }
    const size_t nwords = 2 * (size_t)P + n0 + n1 + 3 * (size_t)M;
    if (g_arp.nwords < nwords) {
        if (g_arp.words) (void)hipFree(g_arp.words);
        g_arp.words = nullptr;
        g_arp.nwords = 0;
        if (hipMalloc((void **)&g_arp.words, nwords * sizeof(long long)) != hipSuccess) {
            stem_set_error("stem_ar_decode_image_persistent: cannot allocate the tagged vectors");
            return -2;
        }
        g_arp.nwords = nwords;
    }
    if (hipMemsetAsync(g_arp.words, 0, nwords * sizeof(long long), st) != hipSuccess) return -2;       // tag 0: nothing written yet
    int *pin = g_arp.pinned;
    memset(pin, 0, need * sizeof(int));            // sequence numbers of the previous image must not match this one's
    if (hipMemcpyAsync(g_arp.dev, kArpInit, sizeof(kArpInit), hipMemcpyHostToDevice, st) != hipSuccess) return -2;

    ArpArgs a;
    memset(&a, 0, sizeof(a));
    a.w_ctx = w_ctx; a.b_ctx = b_ctx; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.ld_ctx = ld_ctx; a.ld0 = ld0; a.n0 = n0; a.ld1 = ld1; a.n1 = n1; a.ld2 = ld2;
    a.buf = buf; a.H = H; a.W = W; a.M = M; a.pad = pad; a.tp = tp; a.hp = hp; a.ctx = ctx; a.h1 = h1; a.h2 = h2; a.gp = gp;
    a.table = table; a.T = T; a.bound = scale_bound; a.slope = slope; a.mail = pin; a.dev = g_arp.dev; a.words = g_arp.words;
#ifdef STEM_EXPERIMENTS
    a.dbg = g_arp_dbg;
#endif
    a.want_xcc = g_arp_want_xcc;
    a.nwg = NWG_DEFAULT;             // the row -> wavefront map is fixed: 32 workgroups x 8 wavefronts (the "arp_workers" selector of round 3 is ignored)
    std::call_once(g_arp_attr_once, [] {
        (void)hipFuncSetAttribute((const void *)ar_decode_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ARP_LDS);
    });
    hipLaunchKernelGGL(ar_decode_persistent_kernel, dim3(GRID), dim3(NT), ARP_LDS, st, a);
    if (hipGetLastError() != hipSuccess) {
        stem_set_error("stem_ar_decode_image_persistent: launch failed");
        return -2;
    }
    int rc_out = 0;
    std::vector<int32_t> idx_v((size_t)M), sym_v((size_t)M);
    // stem_tuning_set("arp_giveup_at", k): the host gives up at position k - 1 exactly as a wait that ran out does (abort word, error
    // return, a half-written latent buffer behind it) -- the test of the caller's fall-back to the per-position loop; 0 = never
    const int giveup_at = stem_tuning(STEM_TUNE_ARP_GIVEUP_AT);
    for (int p = 0; p < N; ++p) {
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        bool lost = giveup_at > 0 && p == giveup_at - 1;
        const long long *idxw = reinterpret_cast<const long long *>(pin + 32) + (size_t)(p & 1) * M;
        long long *symw = reinterpret_cast<long long *>(pin + 32 + 2 * P) + (size_t)(p & 1) * M;
        // all M words carry this position's sequence number (they arrive in any order).  The first position also waits for the launch
        // itself: behind whatever the stream still holds, and -- several images at once -- behind another image's kernel that shares
        // its hardware queue
        for (int c = M - 1; c >= 0 && !lost;) {
            const long long v = __atomic_load_n(idxw + c, __ATOMIC_ACQUIRE);
            if ((int)(v >> 32) == p + 1) {
                idx_v[(size_t)c] = (int32_t)v;
                --c;
                continue;
            }
            if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(p == 0 ? 120 : 5)) {
                lost = true;
                break;
            }
            __builtin_ia32_pause();
        }
        if (lost) {
            stem_set_error("stem_ar_decode_image_persistent: timed out waiting for the device at position %d", p);
            rc_out = -2;
            break;
        }
        if (int rc = decode(dec, idx_v.data(), (size_t)M, cdfs, ncdf, cdf_stride, sizes, offsets, sym_v.data())) {
            stem_set_error("stem_ar_decode_image_persistent: host symbol decoder failed (%d) at position %d", rc, p);
            rc_out = -3;
            break;
        }
        for (int c = 0; c < M; ++c)
            __atomic_store_n(symw + c, ((long long)(p + 1) << 32) | (unsigned)sym_v[(size_t)c], __ATOMIC_RELEASE);
    }
    if (rc_out) __atomic_store_n(pin + 16, -1, __ATOMIC_RELEASE);           // the kernel's polls stop
    if (hipStreamSynchronize(st) != hipSuccess) {
        stem_set_error("stem_ar_decode_image_persistent: device error: %s", hipGetErrorString(hipGetLastError()));
        return -2;
    }
    if (rc_out) return rc_out;
#ifdef STEM_EXPERIMENTS
    {
        long long t[10];
        if (hipMemcpy(t, g_arp.dev + 32, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[ar persistent] shader clock %.0f MHz (s_memtime cycles per 100 MHz tick x 100)\n", 100.0 * (double)t[8] / (double)t[9]);
            fprintf(stderr, "[ar persistent] 100 MHz ticks per position (workgroup 0): mail + look-ahead + host + commit %.1f, ctx %.1f + hand-over %.1f, h1 %.1f, h2 %.1f, "
                            "hand-overs of h1 and h2 %.1f, gp %.1f + mail %.1f\n", (double)t[0] / N, (double)t[1] / N, (double)t[2] / N, (double)t[3] / N,
                    (double)t[5] / N, (double)t[4] / N, (double)t[6] / N, (double)t[7] / N);
        }
    }
#endif
    int flags[2] = {0, 0};
    if (hipMemcpy(flags + 1, g_arp.dev + 96, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || flags[1]) {
        stem_set_error("stem_ar_decode_image_persistent: a device-side wait timed out");
        return -2;
    }
    return 0;
}

#ifdef STEM_EXPERIMENTS
// tools/debug/arp_probe.py: per-position copies of the four products' outputs
STEM_EXPORT void stem_exper_arp_debug(float *p) { g_arp_dbg = p; }
#endif
