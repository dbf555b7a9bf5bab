// libstem_dp.so: see include/stem_dp.h.  Host code only (HIP runtime + RCCL); built by hipcc for the include paths.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/stem_dp.h"

#define STEM_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[512] = "";
int fail(int rc, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return rc;
}

constexpr int MAXEV = 8;
struct Task {
    int kind = 0;                  // 0 exchange, 1 fence, 2 stop
    hipEvent_t ev[MAXEV];
    int nev = 0;
    float *buf = nullptr;
    size_t count = 0;
    unsigned seq = 0;
};

struct Dp {
    std::atomic<ncclComm_t> comm{nullptr};
    hipStream_t cs = nullptr;      // the communicator's stream: collectives and flag writes, nothing that waits
    unsigned *flag = nullptr;      // signal memory: "fences completed"
    unsigned seq = 0;
    int device = 0;
    int world = 0;
    bool connected = false;
    long fault_at = -1;            // STEM_DP_FAULT=<n>: the n-th exchange (0-based) fails in the helper as a refused collective would
    long exchanges = 0;
    std::thread helper;
    std::thread aborter;           // runs ncclCommAbort after a failure (see fail_all)
    std::mutex mu;
    std::mutex comm_mu;            // a collective is enqueued, or the communicator aborted / destroyed, by one thread at a time
    std::condition_variable cv;
    std::deque<Task> q;
    std::vector<hipEvent_t> pool;
    std::atomic<int> status{0};
    std::atomic<int> claimed{0};   // the one thread that fills err[]
    char err[256] = "";

    hipEvent_t event()
    {
        {
            std::lock_guard<std::mutex> l(mu);
            if (!pool.empty()) {
                hipEvent_t e = pool.back();
                pool.pop_back();
                return e;
            }
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return e;
    }
    void recycle(const hipEvent_t *ev, int n)
    {
        std::lock_guard<std::mutex> l(mu);
        for (int i = 0; i < n; ++i)
            if (ev[i]) pool.push_back(ev[i]);
    }
    void push(const Task &t)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            q.push_back(t);
        }
        cv.notify_one();
    }
    // First failure wins: the message is complete BEFORE the status becomes visible (readers look at err once status != 0).
    // Abort-all: this rank's communicator is aborted, so that its own queued collectives end and every later submit / fence of this
    // rank returns the status -- the rank leaves with a non-zero exit code and its launcher stops the peers (bench.launch_ranks,
    // torch.distributed.run), which would otherwise sit inside a collective this rank never joins.
    void fail_all(int rc, const char *what)
    {
        int zero = 0;
        if (!claimed.compare_exchange_strong(zero, 1)) return;
        snprintf(err, sizeof(err), "%s", what);
        status.store(rc, std::memory_order_release);
        ncclComm_t c = nullptr;
        {
            std::lock_guard<std::mutex> l(comm_mu);
            c = comm.exchange(nullptr);
        }
        // ncclCommAbort synchronises with the DEVICE (observed: it does not return while a consumer stream sits in the
        // hipStreamWaitValue32 of a fence), and the flags of the fences queued so far are released by the helper thread right after
        // this function: the abort gets a thread of its own (joined at destruction), so that neither waits for the other
        if (c) {
            const int dev = device;
            aborter = std::thread([c, dev] {
                (void)hipSetDevice(dev);
                (void)ncclCommAbort(c);
            });
        }
    }
    void release_flag(unsigned seq_)
    {
        // nothing may be left waiting for a value nobody writes: after a failure the flag is still written -- on the communication
        // stream, which is ours and outlives the aborted communicator; should even that enqueue fail, by a copy from the host
        if (hipStreamWriteValue32(cs, flag, seq_, 0) == hipSuccess) return;
        (void)hipStreamSynchronize(cs);
        const unsigned v[2] = {seq_, 0};
        (void)hipMemcpy(flag, v, sizeof(v), hipMemcpyHostToDevice);
    }
    void run()
    {
        (void)hipSetDevice(device);
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return !q.empty(); });
                t = q.front();
                q.pop_front();
            }
            if (t.kind == 2) return;
            if (t.kind == 1) {
                // behind the collectives on the same stream
                if (status.load(std::memory_order_acquire) != 0) {
                    release_flag(t.seq);
                } else if (hipStreamWriteValue32(cs, flag, t.seq, 0) != hipSuccess) {
                    fail_all(-2, "hipStreamWriteValue32 failed");
                    (void)hipStreamSynchronize(cs);
                    const unsigned v[2] = {t.seq, 0};
                    (void)hipMemcpy(flag, v, sizeof(v), hipMemcpyHostToDevice);
                }
                continue;
            }
            // the slice is final once its producers' events have completed: polled on the host, so that the communicator's stream
            // gets the collective only when it can start
            for (int i = 0; i < t.nev; ++i) {
                hipError_t e;
                while ((e = hipEventQuery(t.ev[i])) == hipErrorNotReady) __builtin_ia32_pause();
                if (e != hipSuccess) fail_all(-2, "hipEventQuery failed");
            }
            recycle(t.ev, t.nev);
            const long index = exchanges++;
            if (status.load(std::memory_order_acquire) != 0) continue;
            if (index == fault_at) {
                fail_all(-3, "injected fault (STEM_DP_FAULT): collective refused");
                continue;
            }
            ncclResult_t r = ncclSuccess;
            {
                std::lock_guard<std::mutex> l(comm_mu);
                ncclComm_t c = comm.load();
                if (!c) continue;
                r = ncclAllReduce(t.buf, t.buf, t.count, ncclFloat, ncclSum, c, cs);
            }
            if (r != ncclSuccess) fail_all(-3, ncclGetErrorString(r));
        }
    }
    void teardown()
    {
        if (helper.joinable()) {
            Task t;
            t.kind = 2;
            push(t);
            helper.join();
        }
        (void)hipSetDevice(device);
        if (aborter.joinable()) aborter.join();
        if (cs) (void)hipStreamSynchronize(cs);
        {
            std::lock_guard<std::mutex> l(comm_mu);
            ncclComm_t c = comm.exchange(nullptr);
            if (c) (void)ncclCommDestroy(c);
        }
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        if (cs) (void)hipStreamDestroy(cs);
        if (flag) (void)hipFree(flag);
    }
};

}   // namespace

STEM_EXPORT const char *stem_dp_last_error(void) { return g_err; }

STEM_EXPORT int stem_dp_unique_id(unsigned char *id128)
{
    if (!id128) return fail(-1, "stem_dp_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == STEM_DP_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(-3, "stem_dp_unique_id: %s", ncclGetErrorString(r));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

STEM_EXPORT int stem_dp_prepare(void **handle, int device)
{
    if (!handle) return fail(-1, "stem_dp_prepare: null pointer");
    *handle = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail(-2, "stem_dp_prepare: hipSetDevice(%d) failed", device);
    int ok = 0;
    if (hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !ok)
        return fail(-4, "stem_dp_prepare: this device / runtime has no stream wait-value operation");
    Dp *d = new Dp;
    d->device = device;
    if (const char *f = getenv("STEM_DP_FAULT"))
        if (*f) d->fault_at = atol(f);
    void *f = nullptr;
    if (hipExtMallocWithFlags(&f, 8, hipMallocSignalMemory) != hipSuccess || !f) {
        delete d;
        return fail(-2, "stem_dp_prepare: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
    }
    d->flag = static_cast<unsigned *>(f);
    *reinterpret_cast<volatile unsigned long long *>(f) = 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&d->cs, hipStreamNonBlocking, hi) != hipSuccess) {
        d->cs = nullptr;
        d->teardown();
        delete d;
        return fail(-2, "stem_dp_prepare: cannot create the communication stream");
    }
    *handle = d;
    return 0;
}

STEM_EXPORT int stem_dp_connect(void *handle, const unsigned char *id128, int world, int rank)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d || !id128 || world < 1 || rank < 0 || rank >= world) return fail(-1, "stem_dp_connect: bad arguments (world %d, rank %d)", world, rank);
    if (d->connected) return fail(-1, "stem_dp_connect: this handle is connected already");
    if (hipSetDevice(d->device) != hipSuccess) return fail(-2, "stem_dp_connect: hipSetDevice(%d) failed", d->device);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return fail(-3, "stem_dp_connect: ncclCommInitRank: %s", ncclGetErrorString(r));
    int n = 0;
    if (ncclCommCount(c, &n) != ncclSuccess || n != world) {
        (void)ncclCommAbort(c);
        return fail(-3, "stem_dp_connect: the communicator reports %d ranks, %d expected", n, world);
    }
    d->comm.store(c);
    d->world = world;
    d->connected = true;
    d->helper = std::thread([d] { d->run(); });
    return 0;
}

STEM_EXPORT int stem_dp_create(void **handle, const unsigned char *id128, int world, int rank, int device)
{
    if (!handle || !id128 || world < 1 || rank < 0 || rank >= world) return fail(-1, "stem_dp_create: bad arguments (world %d, rank %d)", world, rank);
    void *h = nullptr;
    int rc = stem_dp_prepare(&h, device);
    if (rc != 0) return rc;
    rc = stem_dp_connect(h, id128, world, rank);
    if (rc != 0) {
        (void)stem_dp_destroy(h);
        return rc;
    }
    *handle = h;
    return 0;
}

STEM_EXPORT int stem_dp_nranks(void *handle)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_nranks: null handle");
    std::lock_guard<std::mutex> l(d->comm_mu);
    ncclComm_t c = d->comm.load();
    if (!c) return fail(d->status.load() ? d->status.load() : -1, "stem_dp_nranks: no communicator (%s)", d->connected ? d->err : "not connected");
    int n = 0;
    const ncclResult_t r = ncclCommCount(c, &n);
    if (r != ncclSuccess) return fail(-3, "stem_dp_nranks: %s", ncclGetErrorString(r));
    return n;
}

STEM_EXPORT int stem_dp_submit(void *handle, void *const *streams, int n, float *buf, size_t count)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d || !buf || n < 0 || n > MAXEV || (n && !streams)) return fail(-1, "stem_dp_submit: bad arguments (%d streams)", n);
    if (!d->connected) return fail(-1, "stem_dp_submit: not connected");
    if (int s = d->status.load(std::memory_order_acquire)) return fail(s, "stem_dp_submit: this rank's exchange failed earlier: %s", d->err);
    if (!count) return 0;
    Task t;
    t.buf = buf;
    t.count = count;
    t.nev = n;
    for (int i = 0; i < n; ++i) {
        t.ev[i] = d->event();
        if (!t.ev[i] || hipEventRecord(t.ev[i], (hipStream_t)streams[i]) != hipSuccess) {
            // the collective cannot be issued on this rank: its peers would wait for it -- abort-all, and the events go back
            d->recycle(t.ev, i + 1);
            d->fail_all(-2, "stem_dp_submit: hipEventRecord failed");
            return fail(-2, "stem_dp_submit: hipEventRecord failed");
        }
    }
    d->push(t);
    return 0;
}

STEM_EXPORT int stem_dp_fence(void *handle, void *stream)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_fence: null handle");
    if (!d->connected) return fail(-1, "stem_dp_fence: not connected");
    if (int s = d->status.load(std::memory_order_acquire)) return fail(s, "stem_dp_fence: this rank's exchange failed earlier: %s", d->err);
    Task t;
    t.kind = 1;
    // 32-bit sequence compared with >=: it would wrap after 2^32 fences (one per optimiser step: ~1e8 septuplets of 6 steps at
    // 10 ms each = 8 years of training); a run that long re-creates its reducer
    t.seq = ++d->seq;
    d->push(t);
    if (hipStreamWaitValue32((hipStream_t)stream, d->flag, t.seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
        d->fail_all(-2, "stem_dp_fence: hipStreamWaitValue32 failed");
        return fail(-2, "stem_dp_fence: hipStreamWaitValue32 failed");
    }
    return 0;
}

STEM_EXPORT int stem_dp_status(void *handle)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_status: null handle");
    const int s = d->status.load(std::memory_order_acquire);
    if (s) (void)fail(s, "%s", d->err);
    return s;
}

STEM_EXPORT int stem_dp_abort(void *handle, int code, const char *why)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_abort: null handle");
    d->fail_all(code < 0 ? code : -5, why && *why ? why : "aborted by the host");
    return 0;
}

STEM_EXPORT int stem_dp_destroy(void *handle)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return 0;
    d->teardown();
    delete d;
    return 0;
}
