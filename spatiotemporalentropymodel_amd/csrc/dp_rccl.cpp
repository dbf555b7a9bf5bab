// libstem_dp.so: see include/stem_dp.h.  Host code only (HIP runtime + RCCL); built by hipcc for the include paths.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
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
    ncclComm_t comm = nullptr;
    hipStream_t cs = nullptr;      // the communicator's stream: collectives and flag writes, nothing that waits
    unsigned *flag = nullptr;      // signal memory: "fences completed"
    unsigned seq = 0;
    int device = 0;
    std::thread helper;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> q;
    std::vector<hipEvent_t> pool;
    std::atomic<int> status{0};
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
    void push(const Task &t)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            q.push_back(t);
        }
        cv.notify_one();
    }
    void note(int rc, const char *what)
    {
        int zero = 0;
        if (status.compare_exchange_strong(zero, rc)) snprintf(err, sizeof(err), "%s", what);
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
                // behind the collectives on the same stream; should the enqueue fail the flag is written from here: nothing may be
                // left waiting for a value nobody writes
                if (hipStreamWriteValue32(cs, flag, t.seq, 0) != hipSuccess) {
                    note(-2, "hipStreamWriteValue32 failed");
                    (void)hipStreamSynchronize(cs);
                    __atomic_store_n(flag, t.seq, __ATOMIC_RELEASE);
                }
                continue;
            }
            // the slice is final once its producers' events have completed: polled on the host, so that the communicator's stream
            // gets the collective only when it can start
            for (int i = 0; i < t.nev; ++i) {
                hipError_t e;
                while ((e = hipEventQuery(t.ev[i])) == hipErrorNotReady) __builtin_ia32_pause();
                if (e != hipSuccess) note(-2, "hipEventQuery failed");
            }
            {
                std::lock_guard<std::mutex> l(mu);
                for (int i = 0; i < t.nev; ++i) pool.push_back(t.ev[i]);
            }
            if (status.load() == 0) {
                const ncclResult_t r = ncclAllReduce(t.buf, t.buf, t.count, ncclFloat, ncclSum, comm, cs);
                if (r != ncclSuccess) note(-3, ncclGetErrorString(r));
            }
        }
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

STEM_EXPORT int stem_dp_create(void **handle, const unsigned char *id128, int world, int rank, int device)
{
    if (!handle || !id128 || world < 1 || rank < 0 || rank >= world) return fail(-1, "stem_dp_create: bad arguments (world %d, rank %d)", world, rank);
    if (hipSetDevice(device) != hipSuccess) return fail(-2, "stem_dp_create: hipSetDevice(%d) failed", device);
    int ok = 0;
    if (hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !ok)
        return fail(-4, "stem_dp_create: this device / runtime has no stream wait-value operation");
    Dp *d = new Dp;
    d->device = device;
    void *f = nullptr;
    if (hipExtMallocWithFlags(&f, 8, hipMallocSignalMemory) != hipSuccess || !f) {
        delete d;
        return fail(-2, "stem_dp_create: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
    }
    d->flag = static_cast<unsigned *>(f);
    *reinterpret_cast<volatile unsigned long long *>(f) = 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&d->cs, hipStreamNonBlocking, hi) != hipSuccess) {
        (void)hipFree(f);
        delete d;
        return fail(-2, "stem_dp_create: cannot create the communication stream");
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&d->comm, world, id, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(d->cs);
        (void)hipFree(f);
        delete d;
        return fail(-3, "stem_dp_create: ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    d->helper = std::thread([d] { d->run(); });
    *handle = d;
    return 0;
}

STEM_EXPORT int stem_dp_submit(void *handle, void *const *streams, int n, float *buf, size_t count)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d || !buf || n < 0 || n > MAXEV || (n && !streams)) return fail(-1, "stem_dp_submit: bad arguments (%d streams)", n);
    if (int s = d->status.load()) return fail(s, "stem_dp_submit: the helper thread failed earlier: %s", d->err);
    if (!count) return 0;
    Task t;
    t.buf = buf;
    t.count = count;
    t.nev = n;
    for (int i = 0; i < n; ++i) {
        t.ev[i] = d->event();
        if (!t.ev[i] || hipEventRecord(t.ev[i], (hipStream_t)streams[i]) != hipSuccess) return fail(-2, "stem_dp_submit: hipEventRecord failed");
    }
    d->push(t);
    return 0;
}

STEM_EXPORT int stem_dp_fence(void *handle, void *stream)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_fence: null handle");
    if (int s = d->status.load()) return fail(s, "stem_dp_fence: the helper thread failed earlier: %s", d->err);
    Task t;
    t.kind = 1;
    t.seq = ++d->seq;
    d->push(t);
    if (hipStreamWaitValue32((hipStream_t)stream, d->flag, t.seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess)
        return fail(-2, "stem_dp_fence: hipStreamWaitValue32 failed");
    return 0;
}

STEM_EXPORT int stem_dp_status(void *handle)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return fail(-1, "stem_dp_status: null handle");
    const int s = d->status.load();
    if (s) (void)fail(s, "%s", d->err);
    return s;
}

STEM_EXPORT int stem_dp_destroy(void *handle)
{
    Dp *d = static_cast<Dp *>(handle);
    if (!d) return 0;
    Task t;
    t.kind = 2;
    d->push(t);
    if (d->helper.joinable()) d->helper.join();
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->cs);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    for (hipEvent_t e : d->pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(d->cs);
    (void)hipFree(d->flag);
    delete d;
    return 0;
}
