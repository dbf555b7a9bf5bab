// Launch tape: the native step executor of the explicit training schedule.
//
// The P-frame optimisation step (trainer.FusedPFrameStep: stem/trainSTEM.py:203-218 as ~85 launches on four streams) is static
// per geometry: the same C-ABI calls with the same pointers every step, a handful of integers that advance by a fixed amount
// (Adam's step count, the Philox offsets of the training noise), the same stream hand-overs.  Python + ctypes need ~20 us per
// launch to walk that schedule (10.9 ms of host time per bench step against 14.3 ms of GPU time); the HIP runtime itself needs
// 3-4 us.  A tape records the calls once -- function address, integer-class and float arguments, per-argument step increments,
// event record / wait pairs between streams -- and stem_tape_replay() re-issues a range of it from C++: no interpreter, no
// argument conversion, no allocator.  Unlike a hipGraph the launches go to the very streams they were recorded on, so stream
// priorities and CU masks stay in force and the side streams keep overlapping (a graph replay put most side-stream kernels on one
// queue and lost the overlap).
//
// Calling convention: every recorded entry point returns int and takes only integer-class (pointers, int, long, size_t) and
// float arguments.  Under the x86-64 System V ABI the two classes are assigned to registers / stack slots independently and in
// order within their class, so calling f(a0, f0, a1) through the prototype (long long a0, long long a1, double f0) places every
// argument where the callee looks for it (int arguments read the low half of their 8-byte slot; a float argument reads the low
// 32 bits of its vector register, where the recorded pattern carries the float's bits).
#include <algorithm>
#include <array>
#include <type_traits>
#include <utility>
#include <vector>

#include "stem_common.h"

namespace {

constexpr int TAPE_MAXI = 40, TAPE_MAXF = 6;
enum { TAPE_CALL = 0, TAPE_WAIT = 1, TAPE_EVENT_RECORD = 2, TAPE_EVENT_WAIT = 3 };

struct TapeEntry {
    int kind;
    void *fn;
    int ni, nf;
    long long iv[TAPE_MAXI];
    long long idelta[TAPE_MAXI];       // added once per replay index: iv[i] + n * idelta[i]
    double fv[TAPE_MAXF];             // SSE-class arguments as 64-bit patterns: a double, or a float's bits in the low half
    hipStream_t s0, s1;                // WAIT: s0 waits for what s1 holds now; EVENT_RECORD / EVENT_WAIT: s0
    hipEvent_t ev;                     // WAIT: owned by the tape; EVENT_*: the caller's
    bool dynamic;
};

struct Tape {
    std::vector<TapeEntry> e;
    std::vector<hipEvent_t> owned;
};

template <size_t... I, size_t... Fx>
int call_seq(void *fn, const long long *iv, const double *fv, std::index_sequence<I...>, std::index_sequence<Fx...>)
{
    using Fn = int (*)(decltype((void)I, 0LL)..., decltype((void)Fx, 0.0)...);
    return reinterpret_cast<Fn>(fn)(iv[I]..., fv[Fx]...);
}
template <size_t NI>
int call_ni(void *fn, const long long *iv, int nf, const double *fv)
{
    switch (nf) {
    case 0: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<0>{});
    case 1: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<1>{});
    case 2: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<2>{});
    case 3: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<3>{});
    case 4: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<4>{});
    case 5: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<5>{});
    default: return call_seq(fn, iv, fv, std::make_index_sequence<NI>{}, std::make_index_sequence<6>{});
    }
}
using Caller = int (*)(void *, const long long *, int, const double *);
template <size_t... N>
constexpr auto make_callers(std::index_sequence<N...>) -> std::array<Caller, sizeof...(N)>
{
    return {{&call_ni<N>...}};
}
const auto kCallers = make_callers(std::make_index_sequence<TAPE_MAXI + 1>{});

// ---- the trampolines' contract, checked per entry point at COMPILE time -----------------------------------------------------------
// call_seq() calls f(a0, f0, a1, ...) through int(long long..., double...): that lands every argument where the callee reads it
// only if (1) each parameter is a scalar of the INTEGER class (integers, enums, pointers; <= 8 bytes) or of the SSE class (float,
// double) -- no struct / union / long double / vector by value, whose classification interleaves the two files --, (2) at most
// TAPE_MAXF <= 8 SSE arguments (all in xmm0-7: none on the stack, where they would interleave with the integer overflow), and
// (3) at most TAPE_MAXI integer ones (the trampoline table's size).  tape_entries.inc (tools/gen_tape_table.py) lists every
// int-returning declaration of include/stem_hip.h; a new entry point that breaks the contract fails the build here.
static_assert(TAPE_MAXF <= 8, "SSE arguments beyond xmm7 go to the stack and would interleave with the integer overflow area");
template <class T>
constexpr bool tape_int_class = (std::is_integral<T>::value || std::is_enum<T>::value || std::is_pointer<T>::value) && sizeof(T) <= 8;
template <class T>
constexpr bool tape_sse_class = std::is_same<T, float>::value || std::is_same<T, double>::value;
template <class... A>
constexpr bool tape_recordable(int (*)(A...))
{
    constexpr int ni = (0 + ... + (tape_int_class<A> ? 1 : 0)), nf = (0 + ... + (tape_sse_class<A> ? 1 : 0));
    return ni + nf == (int)sizeof...(A) && ni <= TAPE_MAXI && nf <= TAPE_MAXF;
}
#define STEM_TAPE_ENTRY(name) \
    static_assert(tape_recordable(&name), #name ": prototype outside the launch tape's calling contract (csrc/tape.hip)");
#include "tape_entries.inc"
#undef STEM_TAPE_ENTRY

#define STEM_TAPE_ENTRY(name) reinterpret_cast<const void *>(&name),
const void *const kRecordable[] = {
#include "tape_entries.inc"
};
#undef STEM_TAPE_ENTRY

}   // namespace

/* 1 if `fn` is the address of an int-returning entry point of this library whose prototype passed the compile-time check of the
 * trampolines' calling contract (tape.LaunchTape.add_call refuses anything else), 0 otherwise */
STEM_EXPORT int stem_tape_entry_recordable(void *fn)
{
    return std::find(std::begin(kRecordable), std::end(kRecordable), (const void *)fn) != std::end(kRecordable) ? 1 : 0;
}

STEM_EXPORT void *stem_tape_create(void) { return new Tape(); }

STEM_EXPORT void stem_tape_destroy(void *tape)
{
    Tape *t = static_cast<Tape *>(tape);
    if (!t) return;
    for (hipEvent_t ev : t->owned) (void)hipEventDestroy(ev);
    delete t;
}

STEM_EXPORT int stem_tape_length(void *tape) { return tape ? (int)static_cast<Tape *>(tape)->e.size() : -1; }

/* kinds[i]: 0 = integer class (ivals[i]); otherwise SSE class: fvals[i] holds the 64-bit pattern the register receives -- a double, or
 * a float's bits in its low half (the callee reads the low 32 bits).  ideltas may be null (no argument advances).  Returns the entry's index. */
STEM_EXPORT int stem_tape_add_call(void *tape, void *fn, int nargs, const unsigned char *kinds, const long long *ivals, const double *fvals,
                                   const long long *ideltas)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t && fn && nargs >= 0 && (nargs == 0 || (kinds && ivals && fvals)), "stem_tape_add_call: null argument");
    TapeEntry en;
    memset(&en, 0, sizeof(en));
    en.kind = TAPE_CALL;
    en.fn = fn;
    for (int i = 0; i < nargs; ++i) {
        if (kinds[i] != 0) {
            STEM_CHECK_ARG(en.nf < TAPE_MAXF, "stem_tape_add_call: more than %d float arguments", TAPE_MAXF);
            en.fv[en.nf++] = fvals[i];
        } else {
            STEM_CHECK_ARG(en.ni < TAPE_MAXI, "stem_tape_add_call: more than %d integer-class arguments", TAPE_MAXI);
            en.iv[en.ni] = ivals[i];
            en.idelta[en.ni] = ideltas ? ideltas[i] : 0;
            en.dynamic = en.dynamic || en.idelta[en.ni] != 0;
            ++en.ni;
        }
    }
    t->e.push_back(en);
    return (int)t->e.size() - 1;
}

/* `waiting` continues after everything `signalling` holds at this point of the replay (an event owned by the tape) */
STEM_EXPORT int stem_tape_add_wait(void *tape, void *waiting, void *signalling)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t, "stem_tape_add_wait: null tape");
    TapeEntry en;
    memset(&en, 0, sizeof(en));
    en.kind = TAPE_WAIT;
    en.s0 = (hipStream_t)waiting;
    en.s1 = (hipStream_t)signalling;
    if (hipEventCreateWithFlags(&en.ev, hipEventDisableTiming) != hipSuccess) {
        stem_set_error("stem_tape_add_wait: hipEventCreate failed");
        return -2;
    }
    t->owned.push_back(en.ev);
    t->e.push_back(en);
    return (int)t->e.size() - 1;
}

/* record / wait on an event the caller owns (it must outlive the tape's use): what the schedule hands to code outside the tape */
STEM_EXPORT int stem_tape_add_event(void *tape, void *event, void *stream, int wait)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t && event, "stem_tape_add_event: null argument");
    TapeEntry en;
    memset(&en, 0, sizeof(en));
    en.kind = wait ? TAPE_EVENT_WAIT : TAPE_EVENT_RECORD;
    en.s0 = (hipStream_t)stream;
    en.ev = (hipEvent_t)event;
    t->e.push_back(en);
    return (int)t->e.size() - 1;
}

/* Re-issue entries [lo, hi) for the n-th time after the recording (n = 1: the step after the recorded one): integer arguments
 * advance by n * delta.  Stops at the first call that fails: returns -(index + 1), the callee's message in stem_last_error(). */
STEM_EXPORT int stem_tape_replay(void *tape, int lo, int hi, long long n)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t && lo >= 0 && hi <= (int)t->e.size() && lo <= hi, "stem_tape_replay: bad range [%d, %d)", lo, hi);
    for (int i = lo; i < hi; ++i) {
        const TapeEntry &en = t->e[i];
        switch (en.kind) {
        case TAPE_CALL: {
            int rc;
            if (en.dynamic) {
                long long iv[TAPE_MAXI];
                for (int k = 0; k < en.ni; ++k) iv[k] = en.iv[k] + n * en.idelta[k];
                rc = kCallers[en.ni](en.fn, iv, en.nf, en.fv);
            } else {
                rc = kCallers[en.ni](en.fn, en.iv, en.nf, en.fv);
            }
            if (rc != 0) return -(i + 1);
            break;
        }
        case TAPE_WAIT:
            if (hipEventRecord(en.ev, en.s1) != hipSuccess || hipStreamWaitEvent(en.s0, en.ev, 0) != hipSuccess) {
                stem_set_error("stem_tape_replay: stream hand-over of entry %d failed", i);
                return -(i + 1);
            }
            break;
        case TAPE_EVENT_RECORD:
            if (hipEventRecord(en.ev, en.s0) != hipSuccess) {
                stem_set_error("stem_tape_replay: event record of entry %d failed", i);
                return -(i + 1);
            }
            break;
        default:
            if (hipStreamWaitEvent(en.s0, en.ev, 0) != hipSuccess) {
                stem_set_error("stem_tape_replay: event wait of entry %d failed", i);
                return -(i + 1);
            }
        }
    }
    return 0;
}

/* overwrite integer-class argument `arg` (position among the entry's integer-class arguments) of call entry `entry`: the
 * addresses that change from step to step (the latents a prefetcher hands over in a different buffer per frame) */
STEM_EXPORT int stem_tape_set_iarg(void *tape, int entry, int arg, long long value)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t && entry >= 0 && entry < (int)t->e.size() && t->e[entry].kind == TAPE_CALL && arg >= 0 && arg < t->e[entry].ni,
                   "stem_tape_set_iarg: no integer argument %d in entry %d", arg, entry);
    t->e[entry].iv[arg] = value;
    return 0;
}

/* overwrite SSE-class argument `arg` (position among the entry's float / double arguments) of call entry `entry` with the 64-bit
 * pattern `pattern` (as in stem_tape_add_call): the optimiser hyper-parameters a scheduler edits between steps
 * (stem/trainSTEM.py:123,290: ReduceLROnPlateau rewrites param_groups[0]["lr"]) */
STEM_EXPORT int stem_tape_set_farg(void *tape, int entry, int arg, double pattern)
{
    Tape *t = static_cast<Tape *>(tape);
    STEM_CHECK_ARG(t && entry >= 0 && entry < (int)t->e.size() && t->e[entry].kind == TAPE_CALL && arg >= 0 && arg < t->e[entry].nf,
                   "stem_tape_set_farg: no float argument %d in entry %d", arg, entry);
    t->e[entry].fv[arg] = pattern;
    return 0;
}

/* clear `nbytes` of device memory on a stream (optimizer.zero_grad(), stem/trainSTEM.py:203, as a library call: recordable,
 * unlike Tensor.zero_()) */
STEM_EXPORT int stem_zero_bytes(void *dst, size_t nbytes, void *stream)
{
    STEM_CHECK_ARG(dst || !nbytes, "stem_zero_bytes: null pointer");
    if (nbytes && hipMemsetAsync(dst, 0, nbytes, (hipStream_t)stream) != hipSuccess) {
        stem_set_error("stem_zero_bytes: hipMemsetAsync failed");
        return -2;
    }
    return 0;
}

/* ---- stream flags: order a stream behind work that the HOST has not issued yet --------------------------------------------------
 * A data-parallel rank's helper thread issues the gradient all-reduces (torch.distributed: stem_roi/train_stem_roi.py wraps the
 * models in DistributedDataParallel) only once their inputs are final, so that the communication queue never holds a pending wait
 * (DESIGN.md 8); the compute stream, whose optimiser launches the host enqueues long before, waits for "flag >= step" instead of for
 * an event that does not exist yet.  The flag is 8 bytes of signal memory (hipMallocSignalMemory: what hipStreamWaitValue32 takes). */
STEM_EXPORT int stem_stream_flag_create(void **flag)
{
    STEM_CHECK_ARG(flag, "stem_stream_flag_create: null pointer");
    int dev = 0, ok = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !ok) {
        stem_set_error("stem_stream_flag_create: this device / runtime has no stream wait-value operation");
        return -3;
    }
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess || !p) {
        stem_set_error("stem_stream_flag_create: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
        return -2;
    }
    *static_cast<volatile uint64_t *>(p) = 0;           /* signal memory is host-visible */
    *flag = p;
    return 0;
}

STEM_EXPORT int stem_stream_flag_destroy(void *flag)
{
    if (flag && hipFree(flag) != hipSuccess) {
        stem_set_error("stem_stream_flag_destroy: hipFree failed");
        return -2;
    }
    return 0;
}

/* stream proceeds once *flag >= value (the caller counts steps: monotonic) */
STEM_EXPORT int stem_stream_flag_wait_ge(void *flag, unsigned value, void *stream)
{
    STEM_CHECK_ARG(flag, "stem_stream_flag_wait_ge: null flag");
    if (hipStreamWaitValue32((hipStream_t)stream, flag, value, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
        stem_set_error("stem_stream_flag_wait_ge: hipStreamWaitValue32 failed");
        return -2;
    }
    return 0;
}

/* *flag <- value once the stream reaches this point; stream == (void*)-1: from the host, now (the error path of the helper thread:
 * nothing may be left waiting) */
STEM_EXPORT int stem_stream_flag_write(void *flag, unsigned value, void *stream)
{
    STEM_CHECK_ARG(flag, "stem_stream_flag_write: null flag");
    if (stream == (void *)-1) {
        __atomic_store_n(static_cast<uint32_t *>(flag), value, __ATOMIC_RELEASE);
        return 0;
    }
    if (hipStreamWriteValue32((hipStream_t)stream, flag, value, 0) != hipSuccess) {
        stem_set_error("stem_stream_flag_write: hipStreamWriteValue32 failed");
        return -2;
    }
    return 0;
}

/* device-to-device copy on a stream (the private copies a step hands out: recorded like any other launch) */
STEM_EXPORT int stem_copy_d2d(void *dst, const void *src, size_t nbytes, void *stream)
{
    STEM_CHECK_ARG(dst && src, "stem_copy_d2d: null pointer");
    if (nbytes && hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        stem_set_error("stem_copy_d2d: hipMemcpyAsync failed");
        return -2;
    }
    return 0;
}
