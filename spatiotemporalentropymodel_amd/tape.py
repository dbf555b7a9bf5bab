"""Launch tapes: the training schedule recorded once and replayed by the native executor (csrc/tape.hip).

`trainer.FusedPFrameStep.step` walks ~85 C-ABI launches per P-frame step through Python + ctypes (~20 us each: 10.9 ms of host
time per bench step against ~14 ms of GPU time).  The schedule is static per geometry, so `TapedPFrameStep` records it --

  * every int-returning call into libstem_hip.so (function address, arguments by class, the arrays it passes kept alive),
  * the stream hand-overs (`functional.stream_wait`) and the event records / waits of the schedule (`functional.event_record`,
    `functional.event_wait`),
  * Python callables that must run in between (`functional.tape_py`: the data-parallel exchange hook) --

from two consecutive ordinary steps: the first with every tensor it allocates kept alive (no address is reused inside the step:
a replay has no allocator to arbitrate cross-stream reuse), the second to find the integer arguments that advance from step to
step (Adam's step count, Philox offsets) and by how much.  From then on a step is: copy the two input latents into the recorded
input buffers, `stem_tape_replay` per segment, advance the Python-side counters.  Results are bit-identical to the untaped
schedule (tests/test_hip_trainer.py); stream priorities and CU masks stay in force because the launches go to the recorded
streams (a hipGraph replay lost them).

What a tape assumes, and how each assumption is enforced:

  * the same shapes / strides / reducer / grad_scale / current stream / clipping on-off every step: they form the replay key,
    anything else starts a fresh recording;
  * optimiser hyper-parameters may CHANGE between steps (stem/trainSTEM.py:123,290: ReduceLROnPlateau edits
    param_groups[0]["lr"]): the optimiser binds its launch's float arguments to a provider (`functional.tape_bind_floats`),
    every replay reads the provider and patches the recorded slots that moved (`stem_tape_set_farg`); any OTHER float argument
    that differs between the two recordings is an error;
  * every device operation of the step is a call into libstem_hip.so: a recording runs under `_ForeignOpGuard`, a
    TorchDispatchMode that notes every torch operator touching device memory other than allocations and views (Tensor.zero_(),
    copy_(), .contiguous() of a strided input, .item() ...) -- a replay would silently drop them.  Such a recording is refused
    (TapeRefused) and the step stays on the ordinary schedule;
  * no host-side data dependence inside the step (the fused schedule has none; .item() is caught by the guard);
  * every recorded function is an entry point whose prototype passed the trampolines' compile-time contract check
    (`stem_tape_entry_recordable`: csrc/tape.hip, tape_entries.inc).
"""
from __future__ import annotations

import ctypes as C
import struct
import warnings

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib
from . import functional as F


class TapeRefused(RuntimeError):
    """the schedule cannot be replayed faithfully (it is not static, or it contains device work a tape does not hold)"""


def _pattern(ty, value):
    """the 64-bit pattern an SSE register receives for an argument of ctypes type `ty`"""
    if ty is C.c_float:             # the callee reads the low 32 bits
        return struct.unpack("<d", struct.pack("<fI", float(value), 0))[0]
    return float(value)


class LaunchTape:
    """One recorded schedule.  Use through `recording()`; `finalize(other)` compares with a second recording and builds the
    native tape; `replay(n)` re-issues it."""

    def __init__(self):
        self.entries = []           # ("call", name, addr, kinds, ivals, fvals, keep) | ("wait", dst, src) | ("evrec", ev, st) | ("evwait", st, ev) | ("py", fn, stream)
        self.keep = []              # tensors / ctypes arrays the recorded addresses point into
        self.in_py = 0              # > 0 while a tape_py callable runs (its torch ops are replayed with it)
        self.foreign = []           # torch operators on device memory seen by the recording's guard (a replay would drop them)
        self.problems = []          # calls that could not be recorded
        self.float_bindings = {}    # entry index -> provider() of that call's float arguments (functional.tape_bind_floats)
        self._bound_now = {}        # entry index -> the patterns the native tape holds now
        self._native = None
        self._segments = None

    # ---- recording ---------------------------------------------------------------------------------------------------------------
    def add_call(self, name, fn, args):
        sig = _lib._HIP_SIG[name]
        addr = C.cast(fn, C.c_void_p).value
        # the call itself has already run (the step goes on whatever happens here): what cannot be recorded is noted and makes
        # finalize() refuse the tape
        if not _lib.hip().stem_tape_entry_recordable(addr):
            self.problems.append(f"{name} is not in the library's table of recordable entry points (csrc/tape_entries.inc)")
            return
        if len(args) != len(sig):
            self.problems.append(f"{name}: {len(args)} arguments for a prototype of {len(sig)}")
            return
        kinds, ivals, fvals, keep = [], [], [], []
        isptr = [ty is C.c_void_p for ty in sig]            # by declared type: addresses never count as step counters
        for a, ty in zip(args, sig):
            if ty is C.c_float or ty is C.c_double:
                kinds.append(1 if ty is C.c_float else 2); ivals.append(0); fvals.append(_pattern(ty, a))
                continue
            kinds.append(0); fvals.append(0.0)
            if a is None:
                ivals.append(0)
            elif isinstance(a, int):
                ivals.append(a)
            elif isinstance(a, (C.Array, C.Structure)):
                ivals.append(C.addressof(a)); keep.append(a)
            elif isinstance(a, C.c_void_p):
                ivals.append(a.value or 0)
            else:
                self.problems.append(f"{name}: argument of type {type(a).__name__} is not recordable (pass addresses as integers)")
                return
        self.entries.append(("call", name, addr, kinds, ivals, fvals, keep, isptr))

    def bind_floats(self, provider):
        """the call recorded last reads its float arguments from provider() at every replay"""
        i = len(self.entries) - 1
        if i < 0 or self.entries[i][0] != "call":
            raise RuntimeError("tape_bind_floats: no recorded call to bind to")
        nf = sum(1 for k in self.entries[i][3] if k != 0)
        if len(provider()) != nf:
            raise RuntimeError(f"tape_bind_floats: {self.entries[i][1]} has {nf} float arguments, the provider returns {len(provider())}")
        self.float_bindings[i] = provider

    def _float_patterns(self, i):
        x = self.entries[i]
        sig = _lib._HIP_SIG[x[1]]
        ftys = [ty for ty in sig if ty is C.c_float or ty is C.c_double]
        return [_pattern(ty, v) for ty, v in zip(ftys, self.float_bindings[i]())]

    def refresh_floats(self):
        """bound float arguments whose provider moved since the last replay -> patched into the native tape; returns how many"""
        lib, n = _lib.hip(), 0
        for i in self.float_bindings:
            now = self._float_patterns(i)
            old = self._bound_now[i]
            if now != old and [struct.pack("<d", v) for v in now] != [struct.pack("<d", v) for v in old]:
                for pos, (a, b) in enumerate(zip(now, old)):
                    if struct.pack("<d", a) != struct.pack("<d", b):
                        _lib.check(lib.stem_tape_set_farg(self._native, self.native_index[i], pos, a))
                        n += 1
                self._bound_now[i] = now
        return n

    # ---- native tape -------------------------------------------------------------------------------------------------------------
    def finalize(self, second: "LaunchTape"):
        """`second`: the next step recorded the ordinary way.  Same entry sequence required; integer arguments below 2^32 that
        differ are the per-step counters (their difference = the increment per replay); anything else that differs is an error."""
        a, b = self.entries, second.entries
        if len(a) != len(b) or any(x[0] != y[0] or (x[0] == "call" and x[1] != y[1]) for x, y in zip(a, b)):
            raise TapeRefused("LaunchTape: two consecutive steps issued different launch sequences -- the schedule is not static")
        if self.problems or second.problems:
            raise TapeRefused("LaunchTape: " + "; ".join(sorted(set(self.problems + second.problems))))
        if self.foreign or second.foreign:
            ops = sorted(set(self.foreign + second.foreign))
            raise TapeRefused("LaunchTape: the step runs torch operators on device memory that a replay would drop: " + ", ".join(ops))
        if set(self.float_bindings) != set(second.float_bindings):
            raise TapeRefused("LaunchTape: the two recordings bound different launches to float providers")
        lib = _lib.hip()
        nat = lib.stem_tape_create()
        segs, lo = [], 0
        ndyn = 0
        self.native_index = []      # per entry of self.entries: its index in the native tape (calls only; -1 otherwise)
        for ei, (x, y) in enumerate(zip(a, b)):
            if x[0] != "call":
                self.native_index.append(-1)
            if x[0] == "py":
                hi = lib.stem_tape_length(nat)
                if hi > lo:
                    segs.append((lo, hi))
                segs.append(x)
                lo = hi
                continue
            if x[0] == "wait":
                _lib.check(min(0, lib.stem_tape_add_wait(nat, x[1], x[2])))
            elif x[0] == "evrec":
                _lib.check(min(0, lib.stem_tape_add_event(nat, x[1], x[2], 0)))
            elif x[0] == "evwait":
                _lib.check(min(0, lib.stem_tape_add_event(nat, x[2], x[1], 1)))
            else:
                _, name, addr, kinds, iv, fv, _keep, isptr = x
                self.native_index.append(lib.stem_tape_length(nat))
                deltas = [0] * len(iv)
                bound = ei in self.float_bindings
                if bound:                       # the provider is the source of truth from here on (it may have moved since step A)
                    cur = self._float_patterns(ei)
                    self._bound_now[ei] = cur
                    it = iter(cur)
                    fv = [next(it) if k != 0 else 0.0 for k in kinds]
                for i, (k, va, vb, fa, fb) in enumerate(zip(kinds, iv, y[4], fv, y[5])):
                    if k != 0:
                        if not bound and struct.pack('<d', fa) != struct.pack('<d', fb):
                            raise TapeRefused(f"LaunchTape: float argument {i} of {name} changed between two steps ({fa} -> {fb}) and no "
                                              "provider is bound to it (functional.tape_bind_floats)")
                    elif va != vb:
                        if isptr[i]:
                            continue            # addresses of the second (ordinary) step's tensors: the tape keeps the first step's
                        # what advances from step to step is a counter (Adam's step count, a Philox offset): it grows.  A leading
                        # dimension or a shape that differs between two recordings of the same key is a bug the tape must not paper over
                        if vb < va:
                            raise TapeRefused(f"LaunchTape: integer argument {i} of {name} went from {va} to {vb} between two steps: not a step counter")
                        deltas[i] = vb - va
                        ndyn += 1
                n = len(iv)
                _lib.check(min(0, lib.stem_tape_add_call(nat, addr, n, (C.c_ubyte * n)(*kinds), (C.c_longlong * n)(*iv), (C.c_double * n)(*fv),
                                                         (C.c_longlong * n)(*deltas))))
        hi = lib.stem_tape_length(nat)
        if hi > lo:
            segs.append((lo, hi))
        self._native, self._segments, self.dynamic_args = nat, segs, ndyn
        return self

    def pointer_slots(self, address):
        """[(native entry, integer-argument position)] of every recorded pointer argument equal to `address`"""
        out = []
        for x, ni in zip(self.entries, self.native_index):
            if x[0] != "call":
                continue
            pos = 0
            for k, v, isp in zip(x[3], x[4], x[7]):
                if k != 0:
                    continue
                if isp and v == address:
                    out.append((ni, pos))
                pos += 1
        return out

    def set_pointer(self, slots, address):
        lib = _lib.hip()
        for ni, pos in slots:
            _lib.check(lib.stem_tape_set_iarg(self._native, ni, pos, address))

    def replay(self, n):
        lib = _lib.hip()
        for seg in self._segments:
            if len(seg) == 2:
                rc = lib.stem_tape_replay(self._native, seg[0], seg[1], n)
                if rc != 0:
                    raise RuntimeError(f"LaunchTape: entry {-rc - 1} ({self.entries[-rc - 1][1]}) failed: {(lib.stem_last_error() or b'').decode()}")
            else:
                _, fn, stream = seg
                with F.on_stream(stream):
                    fn()

    def __len__(self):
        return len(self.entries)

    def __del__(self):
        try:
            if self._native:
                _lib.hip().stem_tape_destroy(self._native)
        except Exception:
            pass


class _RecordingLibrary:
    """stands in for the CDLL while a tape records: int-returning entry points are called AND logged"""

    def __init__(self, lib, tape):
        self._lib, self._tape, self._cache = lib, tape, {}

    def __getattr__(self, name):
        w = self._cache.get(name)
        if w is not None:
            return w
        fn = getattr(self._lib, name)
        if (name not in _lib._HIP_SIG or name in _lib._RESTYPE or name.startswith("stem_tape_") or name.startswith("stem_tuning")
                or name.startswith("stem_stream_flag_")):
            # size queries, error strings, the tape API itself: not part of a schedule.  Stream flags carry a step counter and are
            # written by the data-parallel helper THREAD (distributed._CollectiveIssuer): they belong to the reducer's Python
            # entries, which a replay calls again
            return fn
        tape = self._tape

        def recorded(*args, _fn=fn, _name=name):
            rc = _fn(*args)
            if rc == 0:                         # a launch (status 0); anything else is a query's answer (stem_rate_partials) or an error the caller raises
                tape.add_call(_name, _fn, args)
            return rc
        self._cache[name] = recorded
        return recorded


class _ForeignOpGuard(TorchDispatchMode):
    """Notes every torch operator that reads or writes DEVICE memory while a tape records, other than allocations, views and
    allocator bookkeeping: such work is not a library call, so a replay would skip it (a cleared buffer that is never cleared
    again, a .contiguous() copy whose temporary the tape reads after it was freed, an .item() the host waits for).  Operators
    issued inside a functional.tape_py callable are that callable's own: it runs again at every replay."""

    #: operator names (aten::<name>) that move no data
    BENIGN = frozenset("""empty empty_like empty_strided new_empty new_empty_strided view _unsafe_view reshape _reshape_alias
        as_strided slice select narrow permute transpose t unsqueeze squeeze expand expand_as detach alias unbind split
        split_with_sizes chunk view_as flatten unflatten movedim diagonal record_stream is_pinned size stride numel dim
        is_contiguous storage_offset sym_size sym_stride sym_numel sym_storage_offset is_same_size _has_compatible_shallow_copy_type
        set_ resize_ lift_fresh _to_copy_meta is_nonzero_meta""".split())

    def __init__(self, tape, is_device=None):
        super().__init__()
        self.tape = tape
        self.is_device = is_device or (lambda t: t.is_cuda)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.tape.in_py == 0:
            name = getattr(getattr(func, "overloadpacket", func), "__name__", str(func))
            if name not in self.BENIGN:
                import torch.utils._pytree as pytree
                leaves = pytree.tree_leaves((args, kwargs or {}, out))
                if any(isinstance(t, torch.Tensor) and self.is_device(t) for t in leaves):
                    self.tape.foreign.append(f"aten::{name}")
        return out


class recording:
    """`with recording(tape, keep_allocations=True): step()` -- logs the step into `tape`.  keep_allocations: every tensor torch
    allocates inside the block stays alive with the tape (unique addresses, valid for as long as the tape is).  The block runs
    under `_ForeignOpGuard`: torch operators on device memory end up in `tape.foreign`."""

    _ALLOC = ("empty", "zeros", "empty_like", "zeros_like", "empty_strided")

    def __init__(self, tape, keep_allocations):
        self.tape, self.keep_alloc = tape, keep_allocations

    def __enter__(self):
        if F._TAPE is not None:
            raise RuntimeError("a launch tape is already recording")
        self._real = _lib.hip()
        _lib._hip = _RecordingLibrary(self._real, self.tape)
        F._TAPE = self.tape
        self._guard = _ForeignOpGuard(self.tape)
        self._guard.__enter__()
        self._saved = {}
        if self.keep_alloc:
            keep = self.tape.keep
            for n in self._ALLOC:
                real = getattr(torch, n)
                self._saved[n] = real

                def held(*a, _real=real, **kw):
                    t = _real(*a, **kw)
                    keep.append(t)
                    return t
                setattr(torch, n, held)
            self._clone = torch.Tensor.clone

            def clone(t, *a, _real=self._clone, **kw):
                r = _real(t, *a, **kw)
                keep.append(r)
                return r
            torch.Tensor.clone = clone
        return self.tape

    def __exit__(self, *exc):
        self._guard.__exit__(*exc)
        _lib._hip = self._real
        F._TAPE = None
        for n, real in self._saved.items():
            setattr(torch, n, real)
        if self.keep_alloc:
            torch.Tensor.clone = self._clone
        return False


class TapedPFrameStep:
    """`trainer.FusedPFrameStep` behind a launch tape: same arguments, same results (bit for bit), ~3 ms instead of ~11 ms of host
    time per bench step.  Steps 1-2 run the ordinary schedule (weights packed, workspaces at their final sizes), step 3 is
    recorded with its allocations kept, step 4 recorded again to find the per-step counters; from step 5 on the tape replays.

    The tensors a replayed step returns are the recorded step's (static buffers, overwritten by the next step; `y_hat` alternates
    between two buffers so that the next step can read it as its `y_cond` in place): consume them as the training loop does, or
    copy them before the next call.

    Optimiser hyper-parameters (lr, betas, eps, max_norm of both optimisers) are read again at every replay: a scheduler that
    edits `param_groups[0]["lr"]` between steps (stem/trainSTEM.py:123,290) is honoured.  A schedule the tape cannot hold
    (TapeRefused: device-resident optimiser state, torch operators on device memory inside the step, a launch sequence that
    differs between two steps) keeps running on the ordinary `FusedPFrameStep.step`, with one warning; `taped` tells which."""

    WARMUP = 2

    def __init__(self, fused):
        self.fused = fused
        self.refused = None             # the reason a recording was given up for the current key (the step then runs untaped)
        self._reset(None)

    def _reset(self, key):
        self.key, self.calls, self.tape, self.first = key, 0, None, None
        self.replays = 0
        self.refused = None

    taped = property(lambda self: self.tape is not None)

    def _counters(self):
        f = self.fused
        objs = [(f.opt, "t"), (f.aux_opt, "t"), (f.stem.entropy_bottleneck, "_noise_offset"), (f.stem.gaussian_conditional, "_noise_offset")]
        return [(o, a) for o, a in objs if hasattr(o, a)]

    def _refuse(self, why):
        self.refused, self.tape, self.first = str(why), None, None
        warnings.warn(f"TapedPFrameStep: running the ordinary schedule instead of a launch tape -- {why}", RuntimeWarning, stacklevel=3)

    def _key(self, y_cur, y_cond, num_pixels, grad_scale, reducer):
        f = self.fused
        clip = tuple(bool(o.max_norm is not None and o.max_norm > 0) for o in (f.opt, f.aux_opt))
        return (tuple(y_cur.shape), tuple(y_cond.shape), tuple(y_cur.stride()), tuple(y_cond.stride()), y_cur.dtype, y_cond.dtype, y_cur.device,
                int(num_pixels), float(grad_scale), id(reducer), F.cur_stream(y_cur.device).cuda_stream, clip, bool(f.clear_grad_in_adam),
                bool(f.adam_block_max), bool(f.overwrite_grads))

    def step(self, y_cur, y_cond, num_pixels, grad_scale=1.0, reducer=None):
        key = self._key(y_cur, y_cond, num_pixels, grad_scale, reducer)
        if key != self.key:
            self._reset(key)
        f = self.fused
        if self.refused is not None:
            return f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
        if self.calls == 0:
            # device-resident optimiser state is mirrored from the host by torch fill_ calls a tape does not hold
            if f.opt._dev is not None or f.aux_opt._dev is not None:
                self._refuse("the optimiser keeps its step count / learning rate in device memory (enable_device_state)")
                return f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
            for name, t in (("y_cur", y_cur), ("y_cond", y_cond)):
                if F.nhwc_ld(t) is None and not t.is_contiguous():
                    self._refuse(f"{name} is neither NHWC nor contiguous: the schedule would convert it with a torch copy")
                    return f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
        self.calls += 1
        if self.calls <= self.WARMUP:
            return f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
        if self.calls == self.WARMUP + 1:                       # recording A: inputs in static buffers, allocations kept
            # the recorded input buffers carry the CALLER's strides (empty_like of a non-dense view -- a channel slice of an NHWC
            # tensor -- would be dense: the recording would bake ld = C into the launches' integer arguments and set_pointer would
            # then aim them at a tensor with another ld)
            self.in_cur = torch.empty_strided(y_cur.size(), y_cur.stride(), dtype=y_cur.dtype, device=y_cur.device)
            self.in_cond = torch.empty_strided(y_cond.size(), y_cond.stride(), dtype=y_cond.dtype, device=y_cond.device)
            if self.in_cur.stride() != y_cur.stride() or self.in_cond.stride() != y_cond.stride():
                self._refuse("the input latents' strides cannot be reproduced in a recorded buffer")
                return f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
            self.in_cur.copy_(y_cur)
            self.in_cond.copy_(y_cond)
            self.first = LaunchTape()
            before = [getattr(o, a) for o, a in self._counters()]
            try:
                with recording(self.first, keep_allocations=True):
                    self.result = f.step(self.in_cur, self.in_cond, num_pixels, grad_scale, reducer)
            except BaseException:
                self._reset(None)                                # the step itself failed: nothing of this recording survives
                raise
            self.deltas = [getattr(o, a) - b for (o, a), b in zip(self._counters(), before)]
            return self.result
        if self.calls == self.WARMUP + 2:                       # recording B: the ordinary step, logged for the comparison
            second = LaunchTape()
            try:
                with recording(second, keep_allocations=False):
                    res = f.step(y_cur, y_cond, num_pixels, grad_scale, reducer)
            except BaseException:
                self._reset(None)
                raise
            # the step above has run and its result stands whatever happens to the tape
            try:
                self.tape = self.first.finalize(second)
            except TapeRefused as e:
                self._refuse(e)
                return res
            self.replays = 1                                     # this step was "replay index 1" in the counters' progression
            # the frame's latents arrive in a different buffer every step (the prefetcher's): where the schedule reads the recorded
            # input buffer, the tape is pointed at the caller's tensor instead of copying it (same shape / strides: the key)
            self.cur_slots = self.tape.pointer_slots(self.in_cur.data_ptr())
            # the conditioning latents are the previous step's y_hat (stem/trainSTEM.py:179): instead of copying them into the
            # recorded input buffer, the tape reads them where they are and writes this step's y_hat into the OTHER of two
            # output buffers (the schedule reads y_cond after it has written y_hat, so the two must not alias)
            self.cond_slots = self.tape.pointer_slots(self.in_cond.data_ptr())
            y_hat = self.result[0]["y_hat"]
            self.yhat_slots = self.tape.pointer_slots(y_hat.data_ptr())
            self.yhat_bufs = [y_hat, torch.empty_like(y_hat)] if (self.cond_slots and self.yhat_slots) else None
            return res
        if self.cur_slots:
            self.tape.set_pointer(self.cur_slots, y_cur.data_ptr())
        else:
            self.in_cur.copy_(y_cur)                             # on the key's stream, which the replay's first launches are ordered after
        result = self.result
        if self.yhat_bufs is not None:
            out = self.yhat_bufs[1] if y_cond.data_ptr() == self.yhat_bufs[0].data_ptr() else self.yhat_bufs[0]
            self.tape.set_pointer(self.cond_slots, y_cond.data_ptr())
            self.tape.set_pointer(self.yhat_slots, out.data_ptr())
            if out is not self.yhat_bufs[0]:
                result = (dict(self.result[0], y_hat=out),) + tuple(self.result[1:])
        else:
            self.in_cond.copy_(y_cond)
        self.tape.refresh_floats()                               # lr / betas / eps / max_norm as the optimisers hold them NOW
        self.replays += 1
        self.tape.replay(self.replays)
        for (o, a), d in zip(self._counters(), self.deltas):
            setattr(o, a, getattr(o, a) + d)
        f.after_replay()
        return result

    def finish(self):
        self.fused.finish()
