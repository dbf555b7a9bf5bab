// Raster-order decoding of ONE image (compressai/models/spatiotemporalpriors.py:1015-1054) as a single persistent kernel.
//
// stem_ar_decode_image (ar.hip) pays, per latent position, four DEPENDENT dispatches (~6 us each on the queue), a stream
// synchronisation and the host coder: ~35 us x 8160 positions = 0.29 s per 1080p P frame, whatever the arithmetic costs.
// Here the whole loop is one launch:
//   * NW workgroups stay resident and walk through the positions together; the four matrix-vector products of a position
//     (context window -> ctx, EPM.0, EPM.2, EPM.4) are separated by grid barriers (one atomic counter, monotonic targets)
//     instead of kernel boundaries;
//   * the workers are taken from ONE XCD (every workgroup reads its XCC id; the first arrival picks the XCD, the first NW
//     workgroups of that XCD stay, all others exit): the barrier counter and the vectors handed from product to product then
//     live in one L2.  All cross-workgroup accesses are agent-scope (sc1) loads / stores / atomics, so the kernel is correct
//     wherever the workgroups land -- the XCD choice only decides how fast the barriers are;
//   * the host takes part through two pinned-memory mailboxes per position parity: the last product writes the CDF indexes and
//     a sequence flag, the host thread (spinning inside stem_ar_decode_image_persistent) runs the rANS symbol decoder and
//     posts the symbols with its own flag, workgroup 0 polls that flag, commits y_hat = symbol + mean to the latent buffer
//     and releases the other workers.  Every wait is bounded; a timeout raises an abort word that ends the kernel.
//
// Arithmetic: every output row is one wavefront's dot product in the order of gemv3_decode_kernel (segments in order, columns
// lane * 4 + 256 t ascending, xor-shuffle reduction, bias, leaky ReLU), compiled without FMA contraction like ar.hip -- the
// entropy parameters, hence symbols and bytes, are those of the per-position loop and of the encoder.
#include <chrono>

#include "stem_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NWG_DEFAULT = 64;         // resident workers (256 threads each): 256 wavefronts, up to 3 output rows per wavefront
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
    int *mail;                          // pinned: [0] flag_idx, [16] flag_sym, [32 + slot * 2M ..] idx, [32 + 2 * 2M + slot * 2M ..] sym
    int nwg;                            // workers
    int *dev;                           // device, one 128-byte line per word group: [0] worker tickets, [1] chosen XCC (-1) -- agent scope;
                                        // [64] barrier counter, [96] abort, [128] committed positions, [192..] CDF indexes -- L2-local
};

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }      // HW_REG_XCC_ID[3:0]

// The workers share ONE XCD by construction (same HW_REG_XCC_ID), i.e. one L2, and everything they exchange -- barrier counter,
// flags, the vectors handed from product to product -- is kept coherent AT that L2: stores are written through the CU's L1
// (it is a write-through cache) and waited for; the counter / flags are read with an atomic RMW (fetch_add 0), which executes in
// the L2; before data written by other CUs is loaded, the CU's L1 is invalidated (`buffer_inv sc0`).  Agent-scope (sc1)
// accesses would be correct wherever the workgroups run but go to the memory side: measured 2.7-3.3 us per barrier.
#define ARP_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
__device__ inline int poll_l2(int *p)
{
    // written in assembly: the compiler turns an idempotent fetch_add(p, 0) into a plain load, which the L1 may serve forever
    int v;
    const int zero = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    return v;
}
__device__ inline void l1_invalidate() { asm volatile("buffer_inv sc0" ::: "memory"); }
__device__ inline float ld_agent(const float *p) { return *reinterpret_cast<const volatile float *>(p); }       // after l1_invalidate()
__device__ inline void st_agent(float *p, float v) { *reinterpret_cast<volatile float *>(p) = v; }

// all workers arrive (their stores have been waited for), then every worker sees the counter reach `target`
__device__ inline bool grid_barrier(int *bar, int target, int *abort_w, int tid)
{
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, ARP_SCOPE);
        long spins = 0;
        int good = 1;
        while (poll_l2(bar) < target) {
            if ((++spins & 1023) == 0 && (spins > SPIN_LIMIT || poll_l2(abort_w))) {
                __hip_atomic_fetch_add(abort_w, 1, __ATOMIC_RELAXED, ARP_SCOPE);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    l1_invalidate();
    return ok != 0;
}

// rows [first, N) in steps of `stride`: y[n] = act(bias[n] + W[n] . x), x in LDS, one wavefront per row.  The weight loads of a
// row (up to 9 x 16 bytes per lane) are issued back to back -- the products are latency-bound, what counts is loads in flight --
// and then accumulated in the order of gemv3_decode_kernel: segments in order, columns lane * 4 + 256 t ascending.
constexpr int MAXT = 12;                // 256-column steps of the longest product: 4 + 4 + 2 for the 12-tap window at M = 192
__device__ inline void rows(const float *Wm, int ldw, const float *bias, const float *xs, const int *seg_len, const int *seg_woff, const int *seg_xoff,
                            float *y, int N, int first, int stride, bool lrelu, float slope, const ArpArgs &a, int *idx_out)
{
    const int lane = threadIdx.x & 63;
    // flatten the (segment, step) pairs of this product: the same list for every row
    int woffs[MAXT], xoffs[MAXT], nt = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        for (int k = 0; k < seg_len[q]; k += 256)
            if (nt < MAXT) {
                woffs[nt] = seg_woff[q] + k;
                xoffs[nt] = seg_xoff[q] + k;
                // the last step of a segment may be partial: lanes beyond its end do not contribute
                if (k + lane * 4 >= seg_len[q]) woffs[nt] = -1;
                ++nt;
            }
    for (int n = first; n < N; n += stride) {
        const float *wr = Wm + (size_t)n * ldw + lane * 4;
        f32x4 wv[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < nt) wv[t] = *reinterpret_cast<const f32x4 *>(wr + (woffs[t] >= 0 ? woffs[t] : 0));
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < nt && woffs[t] >= 0) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + xoffs[t] + lane * 4);
                acc += xv[0] * wv[t][0] + xv[1] * wv[t][1] + xv[2] * wv[t][2] + xv[3] * wv[t][3];
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) {
            float v = acc + (bias ? bias[n] : 0.f);
            if (lrelu) v = v > 0.f ? v : v * slope;
            st_agent(y + n, v);
            if (idx_out && n < a.M) {
                const float sc = fmaxf(v, a.bound);
                int k = a.T - 1;
                for (int t = 0; t < a.T - 1; ++t) k -= (sc <= a.table[t]) ? 1 : 0;
                *reinterpret_cast<volatile int *>(idx_out + n) = k;
            }
        }
    }
}

// a vector other workgroups have just written (the L1 was invalidated after the barrier; 16 bytes per lane, all of a thread's loads in
// flight together), or read-only data (plain loads), into LDS.  n is a multiple of 4 and at most 3 x 1024 floats.
__device__ inline void stage(float *dst, const float *src, int n, bool coherent)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    const int o = threadIdx.x * 16;
    f32x4 v[3];
    if (coherent) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + j * 4096, 0, 0));
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o + j * 4096, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (threadIdx.x * 4 + j * 1024 < n) *reinterpret_cast<f32x4 *>(dst + threadIdx.x * 4 + j * 1024) = v[j];
}

__global__ __launch_bounds__(256) void ar_decode_persistent_kernel(const ArpArgs a)
{
    __shared__ __attribute__((aligned(16))) float xs[XMAX];
    __shared__ int role;
    const int tid = threadIdx.x;
    int *bar = a.dev + 64, *abort_w = a.dev + 96, *commit = a.dev + 128;
    if (tid == 0) {
        // worker selection: the first workgroup to arrive fixes the XCD, the first NWG workgroups of that XCD are the workers
        const int me = xcc_id();
        int expected = -1;
        __hip_atomic_compare_exchange_strong(a.dev + 1, &expected, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int chosen = __hip_atomic_load(a.dev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const int wave = tid >> 6, first = wg * 4 + wave, stride = a.nwg * 4, NWG = a.nwg;
    volatile int *flag_sym = a.mail + 16;
    int nbar = 0;
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
    for (int p = 0; p < N; ++p) {
        const int h = p / a.W, w = p - h * a.W;
        // ---- previous position: y_hat = symbol + mean, once the host has posted the symbols (workgroup 0), then everybody goes on
        if (p > 0) {
            if (wg == 0) {
                __shared__ int got;
                if (tid == 0) {
                    long spins = 0;
                    int good = 1;
                    for (;;) {
                        // relaxed: an acquire at system scope would invalidate the L2 on every poll; the symbol loads below are
                        // system-scope loads issued after this one has returned
                        const int v = __hip_atomic_load((int *)flag_sym, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (v >= p) break;
                        if (v < 0 || ++spins > SPIN_LIMIT || ((spins & 255) == 0 && poll_l2(abort_w))) {
                            good = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!good) __hip_atomic_fetch_add(abort_w, 1, __ATOMIC_RELAXED, ARP_SCOPE);
                    got = good;
                }
                __syncthreads();
                if (got) {
                    const int pp = p - 1, ph = pp / a.W, pw = pp - ph * a.W;
                    const int *sym = a.mail + 32 + 2 * P + (pp & 1) * P;
                    float *pix = a.buf + ((size_t)(ph + a.pad) * Wp + (pw + a.pad)) * M;
                    for (int c = tid; c < M; c += 256) {
                        const int sv = __hip_atomic_load(sym + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        st_agent(pix + c, (float)sv + ld_agent(a.gp + M + c));
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(commit, 1, __ATOMIC_RELAXED, ARP_SCOPE);          // commit counts committed positions
            } else if (tid == 0) {
                long spins = 0;
                while (poll_l2(commit) < p) {
                    if ((++spins & 1023) == 0 && (spins > SPIN_LIMIT || poll_l2(abort_w))) break;
                }
            }
            __syncthreads();
            l1_invalidate();
            if (tid == 0) role = poll_l2(abort_w);
            __syncthreads();
            if (role) return;
        }
        ARP_MARK(0);
        // ---- ctx = b_c + W_c . window: rows h, h + 1 (5 pixels each) and h + 2 (2 pixels) of the padded buffer from column w
        {
            const float *r0 = a.buf + ((size_t)h * Wp + w) * M;
            stage(xs, r0, 5 * M, true);
            stage(xs + 5 * M, r0 + (size_t)Wp * M, 5 * M, true);
            stage(xs + 10 * M, r0 + 2 * (size_t)Wp * M, 2 * M, true);
            __syncthreads();
            const int len[3] = {5 * M, 5 * M, 2 * M}, woff[3] = {0, 5 * M, 10 * M}, xoff[3] = {0, 5 * M, 10 * M};
            rows(a.w_ctx, a.ld_ctx, a.b_ctx, xs, len, woff, xoff, a.ctx, P, first, stride, false, 0.f, a, nullptr);
        }
        ARP_MARK(1);
        if (!grid_barrier(bar, ++nbar * NWG, abort_w, tid)) return;
        ARP_MARK(2);
        // ---- h1 = lrelu(b_0 + W_0 . (tp | hp | ctx))
        {
            int len[3], woff[3], xoff[3];
            int o = 0;
            if (a.tp) {
                stage(xs, a.tp + (size_t)p * P, P, false);
                o = P;
            }
            stage(xs + o, a.hp + (size_t)p * P, P, false);
            stage(xs + o + P, a.ctx, P, true);
            __syncthreads();
            if (a.tp) {
                len[0] = P; len[1] = P; len[2] = P; woff[0] = 0; woff[1] = P; woff[2] = 2 * P; xoff[0] = 0; xoff[1] = P; xoff[2] = 2 * P;
            } else {
                len[0] = P; len[1] = P; len[2] = 0; woff[0] = 0; woff[1] = P; woff[2] = 0; xoff[0] = 0; xoff[1] = P; xoff[2] = 0;
            }
            rows(a.w0, a.ld0, a.b0, xs, len, woff, xoff, a.h1, a.n0, first, stride, true, a.slope, a, nullptr);
        }
        ARP_MARK(3);
        if (!grid_barrier(bar, ++nbar * NWG, abort_w, tid)) return;
        ARP_MARK(4);
        // ---- h2 = lrelu(b_1 + W_1 . h1)
        {
            stage(xs, a.h1, a.n0, true);
            __syncthreads();
            const int len[3] = {a.n0, 0, 0}, woff[3] = {0, 0, 0}, xoff[3] = {0, 0, 0};
            rows(a.w1, a.ld1, a.b1, xs, len, woff, xoff, a.h2, a.n1, first, stride, true, a.slope, a, nullptr);
        }
        ARP_MARK(5);
        if (!grid_barrier(bar, ++nbar * NWG, abort_w, tid)) return;
        ARP_MARK(4);
        // ---- gp = b_2 + W_2 . h2 (scales | means); the scales' CDF indexes go to the host mailbox of this position's parity
        {
            stage(xs, a.h2, a.n1, true);
            __syncthreads();
            const int len[3] = {a.n1, 0, 0}, woff[3] = {0, 0, 0}, xoff[3] = {0, 0, 0};
            rows(a.w2, a.ld2, a.b2, xs, len, woff, xoff, a.gp, P, first, stride, false, 0.f, a, a.dev + 192);
        }
        ARP_MARK(6);
        if (!grid_barrier(bar, ++nbar * NWG, abort_w, tid)) return;
        ARP_MARK(7);
        // indexes of position p are complete in device memory: workgroup 0 copies them to the host mailbox as ONE contiguous
        // store (M x 4 bytes; written lane by lane from the products they were 4-byte PCIe writes, ~20 us per position), waits
        // for it and raises the flag (relaxed: a system-scope release would write back the whole L2 first)
        if (wg == 0) {
            int *dst = a.mail + 32 + (p & 1) * P;
            for (int c = tid; c < M; c += 256)
                __hip_atomic_store(dst + c, *reinterpret_cast<const volatile int *>(a.dev + 192 + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.mail, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#ifdef STEM_EXPERIMENTS
    if (wg == 0 && tid == 0)
    {
        for (int i = 0; i < 8; ++i) reinterpret_cast<long long *>(a.dev + 32)[i] = tacc[i];
        reinterpret_cast<long long *>(a.dev + 32)[8] = __builtin_readcyclecounter() - cyc0;
        reinterpret_cast<long long *>(a.dev + 32)[9] = wall_clock64() - wall0;
    }
#endif
    // ---- last position's symbols
    if (wg == 0) {
        __shared__ int got2;
        if (tid == 0) {
            long spins = 0;
            int good = 1;
            for (;;) {
                const int v = __hip_atomic_load((int *)flag_sym, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v >= N) break;
                if (v < 0 || ++spins > SPIN_LIMIT) {
                    good = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!good) __hip_atomic_fetch_add(abort_w, 1, __ATOMIC_RELAXED, ARP_SCOPE);
            got2 = good;
        }
        __syncthreads();
        if (got2) {
            const int pp = N - 1, ph = pp / a.W, pw = pp - ph * a.W;
            const int *sym = a.mail + 32 + 2 * P + (pp & 1) * P;
            float *pix = a.buf + ((size_t)(ph + a.pad) * Wp + (pw + a.pad)) * M;
            for (int c = tid; c < M; c += 256) {
                const int sv = __hip_atomic_load(sym + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                st_agent(pix + c, (float)sv + ld_agent(a.gp + M + c));
            }
        }
    }
}

struct ArpState {
    int *pinned = nullptr, *dev = nullptr;
    size_t pinned_ints = 0;
};
thread_local ArpState g_arp;

}   // namespace

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
    STEM_CHECK_ARG(12 * M <= XMAX && n0 <= XMAX && n1 <= XMAX && 6 * M <= XMAX, "stem_ar_decode_image_persistent: vectors longer than %d floats (M=%d n0=%d n1=%d)",
                   XMAX, M, n0, n1);
    STEM_CHECK_ARG(2 * cdiv(5 * M, 256) + cdiv(2 * M, 256) <= MAXT && 3 * cdiv(2 * M, 256) <= MAXT && cdiv(n0, 256) <= MAXT && cdiv(n1, 256) <= MAXT,
                   "stem_ar_decode_image_persistent: a product needs more than %d 256-column steps (M=%d n0=%d n1=%d)", MAXT, M, n0, n1);
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * M, N = H * W;
    const size_t need = 32 + 4 * (size_t)P;
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
    int *pin = g_arp.pinned;
    pin[0] = 0;
    pin[16] = 0;
    static int init[192];
    memset(init, 0, sizeof(init));
    init[1] = -1;
    if (hipMemcpyAsync(g_arp.dev, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return -2;

    ArpArgs a;
    memset(&a, 0, sizeof(a));
    a.w_ctx = w_ctx; a.b_ctx = b_ctx; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.ld_ctx = ld_ctx; a.ld0 = ld0; a.n0 = n0; a.ld1 = ld1; a.n1 = n1; a.ld2 = ld2;
    a.buf = buf; a.H = H; a.W = W; a.M = M; a.pad = pad; a.tp = tp; a.hp = hp; a.ctx = ctx; a.h1 = h1; a.h2 = h2; a.gp = gp;
    a.table = table; a.T = T; a.bound = scale_bound; a.slope = slope; a.mail = pin; a.dev = g_arp.dev;
    a.nwg = stem_tuning(STEM_TUNE_ARP_WORKERS) > 0 ? stem_tuning(STEM_TUNE_ARP_WORKERS) : NWG_DEFAULT;
    if (a.nwg > GRID / 8) a.nwg = GRID / 8;
    hipLaunchKernelGGL(ar_decode_persistent_kernel, dim3(GRID), dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) {
        stem_set_error("stem_ar_decode_image_persistent: launch failed");
        return -2;
    }
    int rc_out = 0;
    for (int p = 0; p < N; ++p) {
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        bool lost = false;
        while (__atomic_load_n(pin, __ATOMIC_ACQUIRE) < p + 1) {
            if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
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
        const int32_t *idx = pin + 32 + (p & 1) * P;
        int32_t *sym = pin + 32 + 2 * P + (p & 1) * P;
        if (int rc = decode(dec, idx, (size_t)M, cdfs, ncdf, cdf_stride, sizes, offsets, sym)) {
            stem_set_error("stem_ar_decode_image_persistent: host symbol decoder failed (%d) at position %d", rc, p);
            rc_out = -3;
            break;
        }
        __atomic_store_n(pin + 16, p + 1, __ATOMIC_RELEASE);
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
            fprintf(stderr, "[ar persistent] 100 MHz ticks per position (workgroup 0): wait host + commit %.1f, ctx %.1f + barrier %.1f, h1 %.1f, h2 %.1f, "
                            "barriers after h1 / h2 %.1f, gp %.1f + barrier %.1f\n", (double)t[0] / N, (double)t[1] / N, (double)t[2] / N, (double)t[3] / N,
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
