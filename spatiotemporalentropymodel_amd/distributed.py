"""Data parallelism for STEM training: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all.  Samples of a batch are independent through
every op of the path and EMLoss normalises by the LOCAL pixel count (utils.py:19-24), so the global-batch
gradient is the mean of the per-rank gradients: one sum all-reduce of the flat gradient buffer per
optimiser step, followed by the fused clip+Adam with grad_scale = 1/world (clipping therefore acts on
the averaged gradient, as stem/trainSTEM.py:213-214 does on a single device).  The auxiliary loss depends
on parameters only, so `.quantiles` gradients are identical on every rank and need no exchange.

The flat buffer is split into a few contiguous buckets (xGMI is point-to-point, ~153 GB/s per link: a
72 MB ring all-reduce costs ~0.8 ms, small against a >5 ms step; few large messages beat many small
ones).  Buckets are reduced on a side stream so that the exchange of one bucket overlaps whatever the
compute stream still has queued.
"""
from __future__ import annotations

import ctypes
import os
import sys

import torch
import torch.distributed as dist

from . import config as _config


def init_from_env(backend=None, single=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torch.distributed.run contract).

    Backend: STEM_DIST_BACKEND, else RCCL ("nccl") when every rank has a GPU of its own, else gloo -- RCCL refuses two
    ranks on one device, so LOCAL_WORLD_SIZE > device count (the one-GPU test box, the CPU tests) exchanges through the host
    (`all_reduce_sum_`).  `single` (or STEM_DIST_SINGLE=1) creates the process group at world size 1 as well, so that a
    one-GPU box executes the very RCCL calls, stream ordering and reducer bookkeeping of a multi-rank run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()                   # counting devices does not initialise the GPU
    # ranks PER NODE against devices per node: a multi-node job (WORLD_SIZE=16 over two 8-GPU nodes) shares nothing and stays on
    # RCCL; torch.distributed.run sets LOCAL_WORLD_SIZE, a bare launch (bench.launch_ranks, the tests) is one node = WORLD_SIZE
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared = ndev > 0 and local_world > ndev
    if shared:
        local = local % ndev                           # more ranks than GPUs (single-GPU test box): share devices
    cfg = _config.runtime()
    if single is None:
        single = cfg.dist_single
    if (world > 1 or single) and not dist.is_initialized():
        if backend is None:
            backend = cfg.dist_backend or ("nccl" if ndev > 0 and not shared else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            # RCCL's kernels run on a stream torch makes for the process group: at HIP's high priority, like the schedule's own
            # side streams (functional.make_stream) -- at normal priority an all-reduce queued behind the long kernels of the
            # latent-prefetch stream reached the CUs up to ~0.6 ms late, and the optimiser step waits for it (measured with a
            # one-rank group: 25.5 ms per bench step against 22.4 ms without a group)
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:
                pass
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


_HOST_STAGE = {}


def _side_stream(device):
    from . import functional as F
    return F.make_stream(device, "side")


def all_reduce_sum_(t: torch.Tensor):
    """In-place sum all-reduce of `t` on the current stream.  RCCL ("nccl") takes device tensors directly and is
    stream-ordered.  With the gloo backend (CPU tests; the 2-ranks-on-one-GPU tests, where RCCL refuses to put two ranks
    on one device) a device tensor is staged through a pinned host buffer: copy out on the current stream, wait for
    it, exchange on the host, copy back on the same stream -- so callers see the same ordering contract as with RCCL
    (the result is ordered after prior work and before later work of the current stream), just without the overlap."""
    if not t.is_cuda or dist.get_backend() != "gloo":
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return
    key = (t.device, t.dtype, t.numel() > (1 << 22))
    host = _HOST_STAGE.get(key)
    if host is None or host.numel() < t.numel():
        host = _HOST_STAGE[key] = torch.empty(max(t.numel(), 1 << 16), dtype=t.dtype).pin_memory()
    h = host[:t.numel()].view(t.shape)
    st = torch.cuda.current_stream(t.device)
    h.copy_(t, non_blocking=True)
    st.synchronize()
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    t.copy_(h, non_blocking=True)
    st.synchronize()                       # the pinned buffer is reused by the next call


def broadcast_parameters(module: torch.nn.Module, src=0):
    """Replicas must start identical (weights stay replicated afterwards: same gradient, same update)."""
    if not dist.is_initialized():
        return
    stage = dist.get_backend() == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if not t.numel():
            continue
        if stage and t.is_cuda:                      # gloo: through the host (see all_reduce_sum_)
            h = t.data.cpu()
            dist.broadcast(h, src)
            t.data.copy_(h)
        else:
            dist.broadcast(t.data, src)


class FlatGradReducer:
    """Sum all-reduce of a flat gradient tensor in `n_buckets` contiguous pieces."""

    def __init__(self, flat_grad: torch.Tensor, n_buckets: int = 4, min_bucket_elems: int = 1 << 20):
        self.grad = flat_grad
        n = flat_grad.numel()
        n_buckets = max(1, min(n_buckets, n // max(1, min_bucket_elems) or 1))
        step = (n + n_buckets - 1) // n_buckets
        step = (step + 3) // 4 * 4
        self.ranges = [(s, min(n, s + step)) for s in range(0, n, step)]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()            # a world-size-1 group still runs its collectives (init_from_env(single=True))
        self._stream = _side_stream(flat_grad.device) if flat_grad.is_cuda else None

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def all_reduce(self):
        """Call after backward and before the optimiser step.  Returns once the reduced gradient is
        ordered before subsequent work on the current stream."""
        if not self.active:
            return
        if self._stream is None:                       # CPU / gloo
            for s, e in self.ranges:
                dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM)
            return
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)                  # gradients are complete on the compute stream
        with torch.cuda.stream(self._stream):
            for s, e in self.ranges:
                all_reduce_sum_(self.grad[s:e])
        cur.wait_stream(self._stream)


class _CollectiveIssuer:
    """Issues RCCL all-reduces from a helper thread, each one only AFTER the events its slice depends on have completed.

    Why.  torch's process group makes ITS stream wait for an event recorded on the caller's stream.  The host runs several P-frame
    steps ahead of the GPU, so that wait sits unsatisfied in the process group's hardware queue for milliseconds -- and on this chip a
    PENDING wait in a fifth queue is what a data-parallel rank pays for (round 5, profiles/r05_ab_fifth_queue.log: a fifth stream
    that records events, or waits for events that are already complete, costs nothing; the same stream with a pending wait costs
    +2.4 ms per bench step, with or without RCCL behind it).  Here the helper thread synchronises with the producer's event on the
    HOST and then calls all_reduce from an otherwise idle stream: the event the process group records is complete when it is
    recorded, its stream never holds a pending wait, and the collective starts at once.  All ranks issue the same collectives in the
    same order (the order of submit()), as before.

    The other direction -- the compute stream has to wait for collectives that do not exist yet when the host enqueues the optimiser
    step -- goes through a stream flag (stem_stream_flag_*: hipStreamWaitValue32 on signal memory): fence() enqueues "wait until
    flag >= n" on the caller's stream and tells the helper, which orders a flag write behind the step's last collective.  Without
    the wait-value operation (`use_flag=False`, or the runtime lacks it) fence() blocks the host until the helper has issued
    everything and waits for the last Work handle instead: correct, but the host loses its run-ahead."""

    def __init__(self, device, use_flag=True):
        import queue
        import threading
        self.device = device
        self.q = queue.SimpleQueue()
        self.free = queue.SimpleQueue()                        # events the helper is done with
        self.cv = threading.Condition()
        self.submitted = self.issued = 0
        self.works, self.err = [], None
        self.idle = torch.cuda.Stream(device=device)          # never runs kernels: what is recorded on it completes at once
        self.flag, self.seq = None, 0
        if use_flag:
            from . import _lib
            f = ctypes.c_void_p()
            with torch.cuda.device(device):
                if _lib.hip().stem_stream_flag_create(ctypes.byref(f)) == 0:
                    self.flag = f
        self.thread = threading.Thread(target=self._run, name="stem-collectives", daemon=True)
        self.thread.start()

    def event(self):
        try:
            return self.free.get_nowait()
        except Exception:
            return torch.cuda.Event()

    def submit(self, events, tensor):
        if self.err is not None:
            self._raise()
        self.submitted += 1
        self.q.put((events, tensor))

    def _raise(self):
        with self.cv:
            err, self.err = self.err, None
        raise RuntimeError("data-parallel helper thread: a collective failed") from err

    def _run(self):
        from . import _lib
        torch.cuda.set_device(self.device)
        last = None
        while True:
            item = self.q.get()
            if item is None:
                return
            if item[0] == "fence":                             # everything submitted before it has been issued (one thread, in order)
                rc = -1
                try:
                    if last is not None:
                        with torch.cuda.stream(self.idle):
                            last.wait()                        # idle stream behind the process group's stream: for the collective's own duration
                    rc = _lib.hip().stem_stream_flag_write(self.flag, item[1], self.idle.cuda_stream)
                except BaseException as e:
                    with self.cv:
                        self.err = self.err or e
                if rc != 0:                                    # nothing may be left waiting for a flag nobody writes
                    _lib.hip().stem_stream_flag_write(self.flag, item[1], ctypes.c_void_p(-1))
                last = None
                continue
            events, g = item
            w, err = None, None
            try:
                for ev in events:
                    ev.synchronize()                           # host-side: the slice is final
                with torch.cuda.stream(self.idle):
                    w = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
            except BaseException as e:                         # handed to the thread that submits / fences
                err = e
            for ev in events:
                self.free.put(ev)
            last = w if w is not None else last
            with self.cv:
                if w is not None and self.flag is None:
                    self.works.append(w)
                self.err = self.err or err
                self.issued += 1
                self.cv.notify_all()

    def fence(self):
        """order the caller's current stream behind every collective submitted so far"""
        from . import functional as F_
        from . import _lib
        if self.err is not None:
            self._raise()
        if self.flag is not None:
            self.seq += 1
            self.q.put(("fence", self.seq))
            F_._chk(_lib.hip().stem_stream_flag_wait_ge(self.flag, self.seq, F_._stream()))
            return
        with self.cv:
            while self.issued < self.submitted:
                self.cv.wait()
            works, self.works = self.works, []
        if self.err is not None:
            self._raise()
        if works:
            works[-1].wait()                                   # one stream, in order: the last one's completion covers them all

    def close(self):
        self.q.put(None)


class NativeRouteUnavailable(RuntimeError):
    """AGREED by every rank (one MIN all-reduce): some rank cannot prepare the native RCCL issue path, none of them uses it"""


def _agree_all(ok: bool, device) -> bool:
    """True only if `ok` on every rank: one MIN all-reduce on the torch process group (host tensor under gloo)"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class _NativeIssuer:
    """The same job as _CollectiveIssuer without Python on the path: libstem_dp.so (include/stem_dp.h, csrc/dp_rccl.cpp) owns an RCCL
    communicator of its own, a C++ helper thread that polls the producers' events and then enqueues ncclAllReduce, and the stream
    flag the compute stream waits for.  The communicator id travels through torch.distributed once, at construction (collective:
    every rank builds its reducers in the same order).

    Construction never leaves a peer behind.  Step 1 is LOCAL on every rank (library load, stem_dp_prepare: device, wait-value
    capability, flag, stream; rank 0 also draws the id); then the ranks agree (MIN) -- if any of them failed, ALL raise
    NativeRouteUnavailable and the caller may choose another route, collectively.  Step 2 is the collective stem_dp_connect
    (ncclCommInitRank), followed by a second agreement: past this point there is no fallback, a failure raises RuntimeError on
    every rank.  `lib` injects a stand-in for libstem_dp.so (CPU tests of this protocol)."""

    def __init__(self, device, lib=None):
        from . import _lib
        dev = torch.device(device)
        self.h, self.lib, why = ctypes.c_void_p(), None, ""
        ident = (ctypes.c_ubyte * 128)()
        try:
            self.lib = lib if lib is not None else _lib.dp()
            rc = self.lib.stem_dp_prepare(ctypes.byref(self.h), dev.index or 0)
            if rc == 0 and dist.get_rank() == 0:
                rc = self.lib.stem_dp_unique_id(ident)
            if rc != 0:
                why = self._last_error()
        except Exception as e:                               # library not loadable on this rank
            rc, why = -1, str(e)
        if not _agree_all(rc == 0, dev):
            self.close()
            raise NativeRouteUnavailable("the native RCCL issue path cannot be prepared on every rank"
                                         + (f" (this rank: {why})" if rc != 0 else " (this rank is fine)"))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        rc = self.lib.stem_dp_connect(self.h, ident, dist.get_world_size(), dist.get_rank())
        why = self._last_error() if rc != 0 else ""
        n = self.lib.stem_dp_nranks(self.h) if rc == 0 else -1
        if not _agree_all(rc == 0 and n == dist.get_world_size(), dev):
            self.close()
            raise RuntimeError("libstem_dp: the RCCL communicator could not be built on every rank"
                               + (f" (this rank: {why or f'{n} ranks reported, {dist.get_world_size()} expected'})" if rc != 0 or n != dist.get_world_size() else ""))
        self.nranks = int(n)
        import atexit
        atexit.register(self.close)                          # the helper thread and the communicator go before the runtime does

    def _last_error(self):
        try:
            return (self.lib.stem_dp_last_error() or b"").decode()
        except Exception:
            return ""

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libstem_dp: " + self._last_error())

    def submit_streams(self, streams, tensor):
        arr = (ctypes.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
        self._chk(self.lib.stem_dp_submit(self.h, arr, len(streams), tensor.data_ptr(), tensor.numel()))

    def fence(self):
        from . import functional as F_
        self._chk(self.lib.stem_dp_fence(self.h, F_._stream()))

    def check(self):
        """raises if this rank's exchange has failed (the helper thread works behind the host: call after the step's fence, and
        after a synchronise before trusting the replicas)"""
        self._chk(self.lib.stem_dp_status(self.h))

    def abort(self, why="aborted by the host"):
        if self.h:
            self.lib.stem_dp_abort(self.h, -5, str(why).encode()[:200])

    def close(self):
        if self.h and self.lib is not None:
            self.lib.stem_dp_destroy(self.h)
        self.h = ctypes.c_void_p()


class OverlappedGradReducer:
    """Sum all-reduce of a FlatParameters gradient buffer, one contiguous slice per module group, started from
    StemEngine.grad_ready_hook while the rest of backward is still running (RCCL on a side stream).

        red = OverlappedGradReducer(opt.flat); red.attach(stem.engine())
        loss.backward()          # slices are exchanged as they become final
        red.finish()             # compute stream waits for the last slice; then opt.step(red.grad_scale)
    """

    def __init__(self, flat, min_bytes=None):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()
        self._off = {id(p): (o, p.numel()) for p, o in zip(flat.params, flat.offsets)}
        # RCCL: the collective is issued asynchronously FROM the stream that made the slice final (the process group's own stream waits
        # for that stream; nobody waits for the collective until finish()) -- no reducer stream of its own: the step's streams
        # already fill the four hardware queues of their priority, a fifth shares one and its event waits block that queue's other
        # stream (one-rank RCCL group, bench step: 15.02 -> 14.86 ms).  gloo (host staging, synchronous): a side
        # stream as before, so that the staging does not block the weight-gradient stream.
        self._direct = self.active and flat.grad.is_cuda and dist.get_backend() == "nccl"
        self._stream = _side_stream(flat.grad.device) if (flat.grad.is_cuda and not self._direct) else None
        self._works = []           # asynchronous collectives of this step (direct mode)
        self._done = []            # [lo, hi) slices reported since the last finish()
        self._pending = []         # reported, final, not yet exchanged: merged contiguous runs [lo, hi, {stream: event}] -- the events
                                   # after which the run's slices are final on the streams that reported them (direct mode)
        self._events = {}          # stream handle -> its event, re-used from step to step
        # RCCL: collectives are issued by a helper thread once their producers' events have completed (_CollectiveIssuer): the
        # process group's queue never holds a pending wait.  STEM_DP_THREADED=0: issued here, from the reporting stream (round 4)
        # STEM_DP_THREADED: 3 (default) native helper thread with its own RCCL communicator (libstem_dp.so) + stream flag; 2 Python
        # helper thread issuing through torch.distributed + stream flag; 1 that thread, finish() blocking the host; 0 off
        mode = int(_config.runtime().dp_threaded)
        self._issuer = None
        if self._direct and mode >= 3 and flat.grad.dtype == torch.float32:
            try:
                self._issuer = _NativeIssuer(flat.grad.device)
            except NativeRouteUnavailable as e:              # agreed by ALL ranks (no librccl / no wait-value operation somewhere):
                import warnings                              # every rank takes the Python helper; any later failure raises everywhere
                warnings.warn(f"native RCCL issue path unavailable ({e}); issuing through torch.distributed")
        if self._issuer is None and self._direct and mode > 0:
            self._issuer = _CollectiveIssuer(flat.grad.device, use_flag=mode >= 2)
        self.calls = 0             # slices reported (schedule bookkeeping)
        self.collectives = 0       # all-reduce calls issued
        #: a run of final gradients is exchanged once it holds this many bytes (and whatever is left at finish()).  What is
        #: exchanged at finish() is exposed: nothing of backward is left to hide it.  With RCCL's stream at high priority a call no
        #: longer costs stream time of its own (one-rank RCCL group, 24 / 12 / 8 / 4 MB: 16.7-16.9 ms per bench step each; it was
        #: +4 ms for 6 calls per P-frame step before), so the threshold only keeps tiny groups (the bottleneck's 60 KB) from
        #: travelling alone: 8 MB leaves the last group (the hyper encoder, ~10 MB) for finish() instead of a 39 MB run.
        #: STEM_DP_MIN_BYTES=0 exchanges every group as soon as it is final
        self.min_bytes = _config.runtime().dp_min_bytes if min_bytes is None else int(min_bytes)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def attach(self, engine):
        engine.grad_ready_hook = self.reduce_params
        return self

    def _exchange(self, lo, hi, deps=None):
        self.collectives += 1
        if not self.active:
            return
        g = self.flat.grad[lo:hi]
        if self._direct and isinstance(self._issuer, _NativeIssuer):
            self._issuer.submit_streams(list(deps or {torch.cuda.current_stream(): None}), g)
            return
        if self._direct and self._issuer is not None:
            # one fresh event per stream that reported a slice of this run, recorded NOW (its stream's tail covers the reports)
            evs = []
            for st in (deps or {torch.cuda.current_stream(): None}):
                ev = self._issuer.event()          # not the per-stream events of reduce_params: the host is steps ahead of the helper
                ev.record(st)
                evs.append(ev)
            self._issuer.submit(evs, g)
            return
        if self._direct:
            cur = torch.cuda.current_stream()
            for st, ev in (deps or {}).items():    # slices of THIS run that became final on another stream (nothing else is waited for)
                if st != cur:
                    cur.wait_event(ev)
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
            return
        if self._stream is None:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return
        # every slice of the run was ordered in front of the reducer's stream when it was REPORTED (reduce_params); the wait below
        # only covers a caller that exchanges from a stream nobody reported on
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):
            all_reduce_sum_(g)

    def reduce_params(self, params):
        """Report the gradients of `params` as final.  They join the pending runs (module groups are contiguous in the flat
        buffer because its order is sorted by parameter name; neighbouring groups merge); a run is exchanged as soon as it
        holds `min_bytes`."""
        spans = sorted(self._off[id(p)] for p in params if id(p) in self._off)
        if not spans:
            return
        dep = {}
        if self._direct:
            # one event per report: "these slices are final on this stream from here on".  A run that is exchanged from the very
            # stream that reported all of its slices needs none of them (stream order); the events are recorded anyway because a
            # pending run may leave later, merged with slices of another stream
            # (one event per stream, recorded again at every report: a wait refers to the latest record before it, which on the
            # same stream covers the earlier ones)
            cur = torch.cuda.current_stream()
            ev = self._events.get(cur.cuda_stream)
            if ev is None:
                ev = self._events[cur.cuda_stream] = torch.cuda.Event()
            ev.record(cur)
            dep = {cur: ev}
        if self.active and self._stream is not None:
            # a reported slice may wait in `_pending` and leave merged with slices reported later from OTHER streams (a second
            # weight-gradient lane, a hook caller with its own streams): order it in front of the reducer's stream now, on the
            # stream that made it final, not on whichever stream happens to be current when the run is exchanged
            self._stream.wait_stream(torch.cuda.current_stream())
        # merge neighbours (tensors are padded to 4 elements) into maximal runs; never bridge over foreign tensors,
        # whose gradients may not be final yet
        runs = []
        for o, n in spans:
            e = (o + n + 3) // 4 * 4
            if runs and o <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], e)
            else:
                runs.append([o, e])
        for lo, hi in runs:
            hi = min(hi, self.flat.grad.numel())
            self._done.append((lo, hi))
            self.calls += 1
            self._pending.append([lo, hi, dict(dep)])
        self._pending.sort(key=lambda r: (r[0], r[1]))
        merged = []
        for lo, hi, d in self._pending:
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
                merged[-1][2].update(d)             # the LATEST event of a stream covers its earlier reports
            else:
                merged.append([lo, hi, dict(d)])
        keep = []
        for lo, hi, d in merged:
            if (hi - lo) * self.flat.grad.element_size() >= self.min_bytes:
                self._exchange(lo, hi, d)
            else:
                keep.append([lo, hi, d])
        self._pending = keep

    def finish(self):
        """Order the exchanged gradient before whatever the compute stream does next.  Every element of the flat buffer
        must have been exchanged exactly once since the previous finish(): a slice reported twice would be summed twice
        (scaled by `world` relative to the rest), a missing one would stay rank-local and the replicas would drift apart
        -- both are programming errors of the schedule that feeds reduce_params(), so they raise instead of being patched
        up with a second collective."""
        done, self._done = sorted(self._done), []
        pending, self._pending = self._pending, []
        pos, n = 0, self.flat.grad.numel()
        for lo, hi in done:
            if lo < pos:
                raise RuntimeError(f"OverlappedGradReducer: gradient slice [{lo}, {hi}) overlaps one already exchanged "
                                   f"(up to {pos}) in this step: it would be summed twice")
            if lo > pos:
                raise RuntimeError(f"OverlappedGradReducer: gradient elements [{pos}, {lo}) were never reported by the "
                                   f"backward schedule ({self._names(pos, lo)}): they would stay rank-local")
            pos = hi
        if pos < n:
            raise RuntimeError(f"OverlappedGradReducer: gradient elements [{pos}, {n}) were never reported by the "
                               f"backward schedule ({self._names(pos, n)}): they would stay rank-local")
        for lo, hi, d in pending:                  # what never reached min_bytes on its own
            self._exchange(lo, hi, d)
        if self._direct and self._issuer is not None:
            self._issuer.fence()
            if isinstance(self._issuer, _NativeIssuer):
                self._issuer.check()               # a failure the helper thread met since the last step: abort-all, this rank raises
        elif self._direct:
            if self._works:                        # the process group runs its collectives on ONE stream, in issue order: the last
                self._works[-1].wait()             # one's completion covers them all -- one cross-queue wait instead of one per call
            self._works = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def check(self):
        """raises if the native exchange of this rank has failed (call after a device synchronise, before trusting the replicas)"""
        if isinstance(self._issuer, _NativeIssuer):
            self._issuer.check()

    @property
    def rccl_nranks(self):
        """ranks as the communicator that carries the gradients reports them: ncclCommCount of libstem_dp's communicator on the native
        route, the torch process group's size otherwise; None without a process group"""
        if isinstance(self._issuer, _NativeIssuer):
            return self._issuer.nranks
        return dist.get_world_size() if self.active else None

    @property
    def route(self):
        if not self.active:
            return "none"
        if isinstance(self._issuer, _NativeIssuer):
            return "libstem_dp (native helper thread, own RCCL communicator)"
        if self._issuer is not None:
            return f"torch.distributed {dist.get_backend()} from a Python helper thread"
        return f"torch.distributed {dist.get_backend()}"

    def _names(self, lo, hi):
        hit = [n for n, p, o in zip(self.flat.names, self.flat.params, self.flat.offsets) if o < hi and o + p.numel() > lo]
        return ", ".join(hit[:4]) + (", ..." if len(hit) > 4 else "")


class GopGradAccumulator:
    """Data-parallel form of the variable-rate loop's gradient handling (stem_roi/train_stem_roi.py:515-566): gradients
    of the frames of a GOP accumulate and the running sum is clipped after every frame.  Clipping is not linear, so the
    ranks cannot exchange only at the end: after each frame's backward the *frame* gradient (this rank's samples) is
    sum-all-reduced and added, scaled by 1/world, to the running global gradient, which is what then gets clipped --
    the sequence of tensors every rank sees equals the single-device full-batch run (the criterion normalises by the local
    batch, equal on every rank).  One all-reduce per model per frame (≈0.2 GB each, a few ms per GOP over xGMI against a
    ≈1 s iteration).  `.quantiles` gradients depend on parameters only and are accumulated without exchange.

        acc = GopGradAccumulator([opt_i.flat, opt_p.flat], [aux_i.flat, aux_p.flat]); acc.begin()
        loss.backward(retain_graph=True); acc.end_frame()          # per frame, before the clip
        clip over acc.running(...); aux.backward(); acc.end_aux()
        acc.finish()                                               # running sums -> .grad buffers, then optimiser steps
    """

    def __init__(self, exchanged, local=()):
        self.exchanged, self.local = list(exchanged), list(local)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()
        self.sums = {id(f): torch.zeros_like(f.grad) for f in self.exchanged + self.local}

    def running(self, flat):
        return self.sums[id(flat)]

    def begin(self):
        for f in self.exchanged + self.local:
            self.sums[id(f)].zero_()
            f.zero_grad()

    def _fold(self, flat, scale):
        acc = self.sums[id(flat)]
        if acc.is_cuda:
            from . import functional as F
            F.axpy_(acc, flat.grad, scale)
        else:
            acc.add_(flat.grad, alpha=scale)
        flat.zero_grad()

    def end_frame(self):
        for f in self.exchanged:
            if self.active:
                all_reduce_sum_(f.grad)
            self._fold(f, 1.0 / self.world)

    def end_aux(self):
        for f in self.local:
            self._fold(f, 1.0)

    def any_rank(self, flag: bool) -> bool:
        """Logical OR of a host-side flag over all ranks (one tiny MAX all-reduce): collective control-flow decisions
        such as the "skip this GOP" break must be taken by every rank or by none."""
        if not self.active:
            return bool(flag)
        dev = self.exchanged[0].grad.device if self.exchanged and dist.get_backend() != "gloo" else torch.device("cpu")
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def finish(self):
        for f in self.exchanged + self.local:
            f.grad.copy_(self.sums[id(f)])


def replica_checksum(t: torch.Tensor) -> int:
    """a 64-bit position-weighted checksum of the BITS of an fp32 tensor (wrapping int64 arithmetic): equal on two replicas only if
    they are bit-identical (up to the 2^-64 collision odds of a checksum)"""
    bits = t.detach().reshape(-1).view(torch.int32).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64) % 65521 + 1
    return int((bits * w).sum().item())


def replicas_identical(t: torch.Tensor):
    """-> (identical, checksum of this rank): MIN and MAX all-reduce of replica_checksum(t) must agree.  Weights of a
    data-parallel job stay replicated only if every rank applied the same reduced gradient: this is the check of that."""
    c = replica_checksum(t)
    if not dist.is_initialized():
        return True, c
    dev = "cpu" if dist.get_backend() == "gloo" else t.device
    lo = torch.tensor([c], dtype=torch.int64, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(int(lo.item()) == int(hi.item())), c


def shard_seed(base_seed: int, rank: int) -> int:
    """Per-rank data / noise seed (seed 1234 + rank)."""
    return base_seed + rank


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


# ---- host-side placement of a rank: the cores next to its GPU ----------------------------------------------------------------------
def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_cpu_lists(sysfs_root="/sys", with_ids=False):
    """[cpus next to physical GPU 0, GPU 1, ...]: the GPU nodes of the KFD topology in node order (= HIP's device order when no
    *_VISIBLE_DEVICES variable re-maps it: visible_device_map), each with the `local_cpulist` of its PCI device (the cores of the
    GPU's NUMA node).  Reads sysfs only -- no HIP call, so a rank can use it before it initialises its GPU.  [] when the topology
    cannot be read.  with_ids: -> (lists, [unique_id of each GPU as a hex string or None])."""
    nodes_dir = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    out, uids = [], []
    try:
        ids = sorted(int(n) for n in os.listdir(nodes_dir) if n.isdigit())
    except OSError:
        return ([], []) if with_ids else []
    for nid in ids:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(nodes_dir, str(nid), "properties")) if " " in line.strip())
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:           # a CPU node of the topology
            continue
        minor = props.get("drm_render_minor", "").strip()
        try:
            cpus = _parse_cpulist(open(os.path.join(sysfs_root, "class", "drm", f"renderD{minor}", "device", "local_cpulist")).read())
        except (OSError, ValueError):
            cpus = []
        out.append(cpus)
        try:
            uids.append(format(int(props.get("unique_id", "0").strip()), "x") or None)
        except ValueError:
            uids.append(None)
    return (out, uids) if with_ids else out


def visible_device_map(n_physical, environ=None, unique_ids=None):
    """HIP device index -> physical GPU index (position among the KFD topology's GPU nodes) under the two filters the runtime
    applies, in its order: ROCR_VISIBLE_DEVICES (indices, or "GPU-<unique id>") selects and re-orders what the ROCr layer
    exposes, HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias) then indexes into THAT list.  As in the runtime, a list stops
    at its first invalid entry.  No variable set: the identity."""
    environ = os.environ if environ is None else environ
    phys = list(range(n_physical))

    def apply(spec, cur, allow_uuid):
        out = []
        for tok in spec.split(","):
            tok = tok.strip()
            idx = None
            if allow_uuid and tok.upper().startswith("GPU-") and unique_ids:
                want = tok[4:].lower().lstrip("0")
                idx = next((i for i, p in enumerate(cur) if unique_ids[p] and unique_ids[p].lstrip("0") == want), None)
            else:
                try:
                    idx = int(tok)
                except ValueError:
                    idx = None
            if idx is None or not 0 <= idx < len(cur) or cur[idx] in out:
                break
            out.append(cur[idx])
        return out

    rocr = environ.get("ROCR_VISIBLE_DEVICES")
    if rocr is not None:
        phys = apply(rocr, phys, True)
    hip = environ.get("HIP_VISIBLE_DEVICES", environ.get("CUDA_VISIBLE_DEVICES"))
    if hip is not None:
        phys = apply(hip, phys, False)
    return phys


def pin_rank_to_gpu_cores(local_rank, local_world=1, sysfs_root="/sys", apply=True, environ=None):
    """Restrict this process to the host cores of its GPU's NUMA node, BEFORE it makes any GPU call (the HIP runtime's helper
    threads inherit the mask): with 8 ranks on one host the enqueueing thread of a rank (10+ ms of Python + ctypes per bench
    step) otherwise migrates across sockets, away from its GPU's PCIe root.  Ranks whose GPUs share a node split that node's
    cores evenly (SMT siblings stay together when the list enumerates them pairwise).  `local_rank` is a HIP device index: a
    launcher's HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES re-mapping is followed to the physical GPU
    (visible_device_map).  Returns the core list, or None when the topology is unknown / STEM_PIN_RANKS=0 -- the run then keeps
    the inherited mask."""
    if not _config.runtime().pin_ranks:
        return None
    phys_lists, uids = gpu_cpu_lists(sysfs_root, with_ids=True)
    vis = visible_device_map(len(phys_lists), environ, uids)
    lists = [phys_lists[p] for p in vis]                      # indexed by HIP device, as the ranks are
    if not lists or local_rank >= len(lists) or not lists[local_rank]:
        return None
    mine = lists[local_rank]
    sharers = [r for r in range(min(local_world, len(lists))) if lists[r] == mine]
    if local_rank in sharers and len(sharers) > 1 and len(mine) >= len(sharers):
        per = len(mine) // len(sharers)
        i = sharers.index(local_rank)
        mine = mine[i * per:(i + 1) * per]
    try:
        allowed = os.sched_getaffinity(0)
        cpus = sorted(c for c in mine if c in allowed) or None
        if cpus and apply:
            os.sched_setaffinity(0, cpus)
        return cpus
    except (AttributeError, OSError):
        return None
