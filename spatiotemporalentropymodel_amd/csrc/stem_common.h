// Shared helpers for libstem_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/stem_hip.h"

#define STEM_EXPORT extern "C" __attribute__((visibility("default")))

void stem_set_error(const char *fmt, ...);

#define STEM_CHECK_ARG(cond, ...)        \
    do {                                 \
        if (!(cond)) {                   \
            stem_set_error(__VA_ARGS__); \
            return -1;                   \
        }                                \
    } while (0)

// Launch errors are reported without synchronising (hipGetLastError only).
#define STEM_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            stem_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- switches ---------------------------------------------------------------------------------------------------------------
// Plan selectors (a tile shape, a split factor: every choice computes the same contraction, differing at most in summation
// order) are integers set through the exported stem_tuning_set() -- tests and sweep tools use them; nothing reads the
// environment per launch.
enum { STEM_TUNE_FX3_TILE = 0, STEM_TUNE_FX3_SPLIT, STEM_TUNE_WG3_SPLIT, STEM_TUNE_ARP_WORKERS, STEM_TUNE_FX3_DEPTH, STEM_TUNE_FX3_GEN_TILE, STEM_TUNE_FX3_MFMA, STEM_TUNE_FX3_GEN_MFMA, STEM_TUNE_FX3_GEN_IMG, STEM_TUNE_FX3_IMG_W, STEM_TUNE_WG3_ROW, STEM_TUNE_WG3_MINCH, STEM_TUNE_TCONV_CPS, STEM_TUNE_ARP_GIVEUP_AT, STEM_TUNE_UNPACK_MB, STEM_TUNE_COUNT };
int stem_tuning(int id);
// image-tile form of the general split-operand convolution (conv_f16x3_img.hip), dispatched from stem_conv2d_f16x3_gen_fwd
bool stem_fx3_img_eligible(int B, int H, int W, int N, int R, int S, int stride, int pad);
int stem_fx3_img_tiles(int B, int H, int W, int N);
int stem_fx3_img_split(int tiles, int nchunks);
int stem_fx3_img_launch(const void *xp, const float *xq, int xpix, int xbytes, const void *wp, const float *wq, int wbytes, const float *bias, int epi,
                        float slope, const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int KS,
                        int T, int split, float *ws, int *cnt, void *stream);
// Switches that CHANGE RESULTS (ablated kernel stages, instrumentation) exist only in a library built with
// -DSTEM_EXPERIMENTS (`make experiments` -> libstem_hip_exper.so, for tools/debug): in the shipped library the macro below is
// a constant null pointer.
#ifdef STEM_EXPERIMENTS
#include <stdlib.h>
#define STEM_EXPER_ENV(name) getenv(name)
#else
#define STEM_EXPER_ENV(name) (static_cast<const char *>(nullptr))
#endif

// parameters per workgroup of the optimiser pass that leaves per-chunk maxima for the fp16 weight packing (optim.hip, conv_f16x3.hip)
#define STEM_ADAM_CHUNK 4096

// ---- the 16-bit operands of the split-operand kernels (conv_f16x3 / wgrad_f16x3 / c4gdn_f16x3) -------------------------------
// Every fp32 value a travels as TWO fp16 numbers a0 = rn(a * 2^e), a1 = rn(a * 2^e - a0): |a * 2^e - a0 - a1| <= 2^-22 |a| 2^e
// (two 11-bit significands), and the three products a0.b0, a0.b1, a1.b0 are exact in the MFMA's fp32 accumulator input, so
// three v_mfma_f32_32x32x16_f16 carry an fp32 product to ~2^-21 (the dropped a1.b1 is <= 2^-22 |a.b|).  fp16 has 5 exponent
// bits, so every planes tensor / packed weight image carries a power-of-two scale 2^e chosen by its producer from an upper
// bound of its values (scale records below); fp16 subnormals are kept by the MFMA (tools/debug/probe/f16_denorm.hip), the
// absolute floor of a stored value is 2^-25 2^-e.
typedef _Float16 hp_t;
typedef hp_t hp8 __attribute__((ext_vector_type(8)));
#define STEM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// Scale record of a planes tensor / packed weight image (device memory, behind the payload; include/stem_hip.h: stem_qrec):
//   word 0: int nslots    word 1: float inv = 2^-e (the planes hold v * 2^e)    words 16 .. 16 + nslots: float max|v| per
//   producing workgroup.  Plain stores only (one slot per workgroup, the header by the workgroup that owns slot 0): nothing
//   to zero, no atomics, bit-reproducible; a consumer takes the maximum over the slots in its prologue / epilogue.
constexpr int QREC_HDR = 16;                       // floats in front of the slots
#ifdef __HIPCC__
__device__ inline int q_nslots(const float *q) { return reinterpret_cast<const int *>(q)[0]; }
__device__ inline float q_inv(const float *q) { return q[1]; }
// header of a record with n slots (one thread): the reserved words are zeroed so that records compare equal byte for byte
__device__ inline void q_header(float *q, int n)
{
    reinterpret_cast<int *>(q)[0] = n;
#pragma unroll
    for (int i = 2; i < QREC_HDR; ++i) q[i] = 0.f;
}
// e with bound * 2^e in [2^14, 2^15): twice the room fp16 (max 65504) needs, so that bounds rounded in fp32 stay safe
__device__ inline int q_exp(float bound)
{
    const unsigned b = __float_as_uint(bound) & 0x7FFFFFFFu;
    int eb = (int)(b >> 23) - 127;                 // floor(log2(bound)) of a normal number
    if (b == 0u || eb == 128) return 0;            // zero tensor / non-finite bound: nothing to protect
    if (eb < -126) eb = -126;
    const int e = 14 - eb;
    return e > 126 ? 126 : e;                      // eb <= 127, so e >= -113: q_pow2(+-e) stays a normal number
}
__device__ inline float q_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ inline float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// maximum over the workgroup (nthreads a multiple of 64, <= 1024); red: >= 16 floats of LDS; two barriers
__device__ inline float block_max(float v, float *red)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
    return m;
}
// max over the slots of a record (every thread gets it)
__device__ inline float q_amax(const float *q, float *red)
{
    int ns = q_nslots(q);
    ns = ns < 0 ? 0 : (ns > (1 << 20) ? (1 << 20) : ns);      // a record nobody wrote must not turn into an unbounded read
    float m = 0.f;
    for (int i = threadIdx.x; i < ns; i += blockDim.x) m = fmaxf(m, q[QREC_HDR + i]);
    return block_max(m, red);
}
// a producer of an fp32 tensor records its workgroup's max |value| (every thread of the workgroup calls this; 256-thread 1-D grids)
__device__ inline void record_block_max(float *q, float m)
{
    __shared__ float qred_rec[16];
    m = block_max(m, qred_rec);
    if (threadIdx.x == 0) {
        q[QREC_HDR + blockIdx.x] = m;
        if (blockIdx.x == 0) {
            q_header(q, gridDim.x);
            q[1] = 1.f;
        }
    }
}
// Split-K arrival: ONE thread of a workgroup calls this after the workgroup's partial tile went to the workspace (agent-scope
// 4-byte stores = sc1 write-through, every thread's s_waitcnt vmcnt(0), workgroup barrier); true for the workgroup that arrives
// last, which then owns the tile (after another barrier it reads the nsplit partials with sc1 loads, which are served at the
// device-coherent level, not from this XCD's L2) and has re-armed the counter for the next launch.
//
// Memory order of the ticket.  In the terms of the HIP memory model the protocol wants a RELEASE before and an ACQUIRE after the
// ticket; -DSTEM_SPLITK_ORDER=__ATOMIC_ACQ_REL builds exactly that (`buffer_wbl2 sc1` + atomic + `buffer_inv sc1` by the one
// thread).  Measured in the training step (round 5, profiles/r05_ab_splitk_order.log, alternating passes on one box):
// 14.36 / 14.44 ms per bench step against 13.83 / 13.85 ms with the relaxed ticket -- +4 %: the write-back / invalidate act on the
// XCD's whole L2, i.e. on the working sets of the kernels running next to this one.  The shipped form therefore stays RELAXED and
// rests on what the ISA guarantees for the accesses involved rather than on a fence: an sc1 store is acknowledged (vmcnt) only
// once it is visible at agent scope, so "all stores acknowledged -> barrier -> ticket" orders data before ticket; the reader's
// loads are issued after a barrier that follows the ticket's return value (control dependence through LDS) and carry sc1, so they
// cannot be served from a stale line of the local L2.  What guards it: tests/test_hip_fullsize.py::
// test_first_launch_on_fresh_workspaces_is_reproducible (the case a wider store form failed in round 4) and the bit-reproducibility
// tests of the training step.
//
// STEM_SPLITK_ACQUIRE_LAST (round 6): the reader's half of that pair where it is needed and nowhere else -- an agent-scope ACQUIRE
// fence (`buffer_inv sc1`) executed by the one thread of the ONE workgroup per tile whose ticket says "last", relaxed tickets for
// everybody else, stores as they are.  nsplit times fewer cache operations than the acquire-release ticket and no write-back at
// all; A/B in the step: profiles/r06_ab_splitk_acquire_last.log.
#ifndef STEM_SPLITK_ORDER
#define STEM_SPLITK_ORDER __ATOMIC_RELAXED
#endif
#ifndef STEM_SPLITK_ACQUIRE_LAST
#define STEM_SPLITK_ACQUIRE_LAST 1
#endif
__device__ inline bool splitk_last_arriver(int *counter, int nsplit)
{
    const int ticket = __hip_atomic_fetch_add(counter, 1, STEM_SPLITK_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == nsplit - 1;
    if (last) {
#if STEM_SPLITK_ACQUIRE_LAST
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return last;
}
// the two fp16 numbers whose sum is x * s (s a power of two)
__device__ inline void q_split(const float x, const float s, hp_t &h0, hp_t &h1)
{
    const float xs = x * s;
    h0 = (hp_t)xs;
    h1 = (hp_t)(xs - (float)h0);                   // the subtraction is exact
}
#endif
