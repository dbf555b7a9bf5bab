"""The P-frame optimisation step of stem/trainSTEM.py:203-218 as ONE explicit launch schedule (no torch autograd, no
torch elementwise kernels):

    zero main gradients -> STEM forward with the fused glue (engine.forward(rate_coef=...): EMLoss value and
    d loss / d likelihood come out of the forward) -> engine.backward -> [data-parallel exchange] -> global norm ->
    clip + Adam -> auxiliary loss + quantile gradient -> aux Adam

~65 launches instead of ~160 through nn.Module / autograd / torch.optim (selfcheck.p_frame_step, the generic route every
parity test against the reference's goldens goes through; tests/test_hip_trainer.py ties this route to it: same Philox
noise -> same likelihoods, gradients and updated parameters).  The arithmetic is the same kernels; what disappears is
the per-dispatch latency of ~50 few-microsecond glue kernels per step and most of the host time.

Only `EMLoss` (rate-only, the trainSTEM criterion) has this closed form; other criteria use the generic route.
"""
from __future__ import annotations

import contextlib
import math
import os

import torch

from . import config as _config
from . import functional as F
from .layers import join_wgrad_stream


#: the stream set-up bench.py measures; the environment variables of the same names override them ("" disables).
#: Round 5: 160 of the 256 CUs for the latent prefetch (192 until round 4): the P-frame step got shorter and its three streams now
#: keep the chip busier, 11.75 against 11.99 ms per bench step (128: 12.15, 144: 12.07, 176: 11.83, 224: 12.10, no mask: 12.08;
#: profiles/r05_ab_cumask*.log)
SCHEDULE_DEFAULTS = {"STEM_STREAM_PRIO": "latents=0,side=-1,compute=-1", "STEM_STREAM_CUMASK": "latents=block:160"}


def tuned_schedule(device):
    """-> a context manager that makes a high-priority compute stream current.  Call it BEFORE anything creates a stream (the
    first FusedPFrameStep / LatentPrefetcher / engine call), enter it around the training loop:

        sched = tuned_schedule(dev)
        fused, pf = FusedPFrameStep(stem, opt, aux_opt), LatentPrefetcher(imodel)
        with sched:
            for frames in loader: ...

    Installs the stream priorities and the CU mask of SCHEDULE_DEFAULTS unless the environment already sets them: the P-frame
    step's own streams at HIP's high priority, the latent-prefetch stream at normal priority and confined to 192 of the 256 CUs,
    so that the step's short kernels never queue behind running workgroups of the long analysis-transform kernels.  Scheduling
    only: no result depends on it."""
    for k, v in SCHEDULE_DEFAULTS.items():
        os.environ.setdefault(k, v)                    # config.runtime() parses them (fields stream_prio / stream_cumask)
    F.make_stream(device, "side")                      # parses STEM_STREAM_PRIO once
    if "compute" in (F._STREAM_PRIO or {}):
        return torch.cuda.stream(F.make_stream(device, "compute"))
    return contextlib.nullcontext()


class LatentPrefetcher:
    """Latents of the frozen I-frame model (getY, compressai priors.py:686-694; frozen in stem/trainSTEM.py:128) computed on their
    own stream, ahead of the P-frame steps that consume them.  The reference calls getY(frame t) inside the loop over the
    frames (stem/trainSTEM.py:171-179); the transform does not depend on the STEM weights, so frame t + `ahead`'s latents can be
    in flight while P-step t runs: their long matrix kernels fill the stretches where a P-frame step cannot use the chip (norm
    reduction -> clip -> Adam -> weight packing, and the small 4x4 / 8x8 layers).

        pf = LatentPrefetcher(imodel)
        pf.start(frames)                      # enqueues frames 0 .. ahead
        for t in range(1, len(frames)):
            y_cur, _ = pf.get(t)              # compute stream waits for frame t's event; frame t + ahead is enqueued
            ...

    Across sequences (a training loop whose loader holds the next batch): `pf.start(frames, next_frames=following)` also computes
    the following sequence's first two latents during this sequence's last two P-frame steps; `pf.start(following, ...)` adopts them.
    """

    def __init__(self, imodel, ahead=1):
        self.imodel, self.ahead = imodel, int(ahead)
        self._stream = None
        self._frames = self._ys = self._events = None
        self._next = 0
        self._nxt = None            # {"frames", "ys", "events", "n"}: the FOLLOWING sequence, whose first latents are computed early

    def _enqueue_into(self, frames, ys, events, t):
        dev = frames[t].device
        if self._stream is None or self._stream.device != dev:
            self._stream = F.make_stream(dev, "latents")
        with F.on_stream(self._stream), torch.no_grad():
            ys[t] = self.imodel.getY(frames[t])
            events[t] = torch.cuda.Event()
            events[t].record(self._stream)

    def _enqueue(self, t):
        self._enqueue_into(self._frames, self._ys, self._events, t)

    def start(self, frames, frames_ready=False, next_frames=None):
        """frames_ready=True: the frame tensors are not being written by work queued on the compute stream (a data loader's
        finished batch), so the first transforms need not wait for what that stream still has queued -- they then overlap the
        tail of the previous optimisation step.

        next_frames: the sequence that will be started NEXT (the loader's following batch, already resident).  A sequence cannot
        begin its first P-frame step before the latents of its frames 0 and 1 exist -- two analysis transforms (~1.5 ms at the
        bench size) during which the step's own streams idle.  With next_frames those two are computed during the LAST two
        P-frame steps of this sequence, like every other frame's, and start(next_frames) then finds them in flight.  The transform
        is frozen (stem/trainSTEM.py:128) and the calls keep their order (this sequence's frames, then the next one's), so every
        result -- noise draws included -- is what the unpipelined loop computes."""
        frames = list(frames) if not isinstance(frames, list) else frames
        n = len(frames)
        adopted = 0
        if self._nxt is not None and (self._nxt["frames"] is frames or
                                      (len(self._nxt["frames"]) == n and all(a is b for a, b in zip(self._nxt["frames"], frames)))):
            self._ys, self._events, adopted = self._nxt["ys"], self._nxt["events"], self._nxt["n"]
        else:
            self._ys, self._events = [None] * n, [None] * n
        self._frames = frames
        self._nxt = None
        if next_frames is not None:
            nf = list(next_frames) if not isinstance(next_frames, list) else next_frames
            self._nxt = {"frames": nf, "ys": [None] * len(nf), "events": [None] * len(nf), "n": 0}
        dev = self._frames[0].device
        if self._stream is None or self._stream.device != dev:
            self._stream = F.make_stream(dev, "latents")
        if not frames_ready:
            F.stream_wait(self._stream, F.cur_stream(dev))      # the frames were produced on the compute stream
        self._next = adopted
        while self._next < n and self._next <= self.ahead:
            self._enqueue(self._next)
            self._next += 1
        return self

    def get(self, t):
        """(y, y + noise) of frame t, ordered before whatever the current stream does next"""
        n = len(self._frames)
        while self._next < n and self._next <= t + self.ahead:
            self._enqueue(self._next)
            self._next += 1
        if self._nxt is not None and self._next >= n:
            # positions beyond this sequence: the next one's opening frames, one more than `ahead` (a sequence opens with TWO latents)
            want = min(t + self.ahead + 2 - n, self.ahead + 1, len(self._nxt["frames"]))
            while self._nxt["n"] < want:
                self._enqueue_into(self._nxt["frames"], self._nxt["ys"], self._nxt["events"], self._nxt["n"])
                self._nxt["n"] += 1
        cur = F.cur_stream(self._frames[t].device)
        cur.wait_event(self._events[t])
        y, yq = self._ys[t]
        y.record_stream(cur)
        yq.record_stream(cur)
        return y, yq


class LazyScalar:
    """A scalar that the step left in device memory on the auxiliary stream: a private copy (later steps do not overwrite it)
    plus the event after which it is valid.  float() / tensor() wait for that event -- no host synchronisation unless
    somebody looks at the value."""

    def __init__(self, value, event, scale=1.0, sqrt=False):
        self._v, self._e, self._scale, self._sqrt = value, event, scale, sqrt

    def tensor(self):
        cur = F.cur_stream(self._v.device)
        cur.wait_event(self._e)
        self._v.record_stream(cur)          # allocated on the auxiliary stream, read here: the allocator must not recycle it under this reader
        v = self._v.reshape(())
        return (v.sqrt() if self._sqrt else v) * self._scale if (self._sqrt or self._scale != 1.0) else v

    def __float__(self):
        return float(self.tensor())


class FusedPFrameStep:
    def __init__(self, stem, optimizer, aux_optimizer):
        self.stem, self.opt, self.aux_opt = stem, optimizer, aux_optimizer
        self.eng = stem.engine()
        eb = stem.entropy_bottleneck
        if eb.quantiles.grad is None or getattr(eb.quantiles, "_flat_grad_view", None) is None:
            raise ValueError("FusedPFrameStep needs the flat-buffer optimisers of optim.configure_optimizers(fused=True)")
        self._aux_loss = torch.zeros(1, dtype=torch.float32, device=eb.quantiles.device)
        self._grad_clean = False          # True once our own Adam pass has cleared the flat gradient buffer
        #: clear the gradient buffer inside the Adam pass (saves the 72 MB memset of the next step); set False to leave the
        #: gradients in place after step() for inspection -- they are then zeroed at the start of the next step instead
        self.clear_grad_in_adam = True
        #: this schedule produces every gradient exactly once per step, so its backward OVERWRITES the flat gradient buffer instead
        #: of adding into a cleared one (engine.accumulate_grads = False for the backward): no clearing inside the Adam pass (72 MB
        #: written) and no read of the old values by the slab sums (72 MB read) -- and the gradients stay in place after step() for
        #: inspection.  False restores the clear-then-accumulate form (`clear_grad_in_adam` then decides where the clearing happens)
        self.overwrite_grads = _config.runtime().trainer_overwrite_grads
        #: the optimiser pass leaves per-chunk maxima of the updated parameters for the fp16 weight packing (no maximum launches)
        self.adam_block_max = _config.runtime().adam_block_max
        # the auxiliary work (one workgroup of latency-bound launches after the optimiser pass) shares the weight-gradient stream, which
        # is idle then: one high-priority stream less (compute, hyper branch, weight gradients + this, and the process group's own
        # = the four hardware queues of that priority; with a fifth the one-rank RCCL run took 14.86 instead of 14.69 ms per step)
        self._aux_stream = self.eng.side_stream(eb.quantiles.device) or F.make_stream(eb.quantiles.device, "side")
        self._aux_pending = False
        self._done_event = torch.cuda.Event()            # re-recorded every step: the LazyScalars of the LATEST step wait on it

    def step(self, y_cur, y_cond, num_pixels, grad_scale=1.0, reducer=None):
        """y_cur / y_cond: the frame's and the conditioning latents [B,C,h,w]; num_pixels = N*H*W of the FRAMES (EMLoss
        normalisation).  Returns (out, criterion_out, aux_loss, grad_norm) like selfcheck.p_frame_step; the loss entries
        are 0-dim fp64 device tensors; aux_loss and grad_norm are LazyScalars (private copies, valid whenever they are read).
        Call finish() before reading `entropy_bottleneck.quantiles` / the aux optimiser outside of step()."""
        stem, opt, aux_opt, eng = self.stem, self.opt, self.aux_opt, self.eng
        eb = stem.entropy_bottleneck
        overwrite = bool(self.overwrite_grads)                      # (the data-parallel reducers sum in place: unaffected)
        if overwrite or (self._grad_clean and opt._dev is None):    # nothing to clear / cleared by the previous step's Adam pass
            for p in opt.flat.params:
                p.grad = p._flat_grad_view
        else:
            opt.flat.zero_grad()                                    # one memset; the aux gradient is overwritten below
        coef = -1.0 / (math.log(2.0) * num_pixels)
        y_hat, lik_y, lik_z, k = eng.forward(y_cur, y_cond, True, rate_coef=(coef, -1.0 / num_pixels))
        eng.accumulate_grads = not overwrite
        try:
            eng.backward(k, k["dlik_y"], k["dlik_z"])               # an attached OverlappedGradReducer exchanges slices in here
        finally:
            eng.accumulate_grads = True
        if reducer is not None:
            F.tape_py(reducer.finish if hasattr(reducer, "finish") else reducer.all_reduce)
        join_wgrad_stream()
        main = F.cur_stream(y_hat.device)
        if self._aux_pending:                    # the previous step's auxiliary work reads the parameters Adam is about to change
            F.stream_wait(main, self._aux_stream)
            self._aux_pending = False
        F.sumsq(opt.flat.grad, opt._sumsq, overwrite=True)
        clean = opt._dev is None and self.clear_grad_in_adam and not overwrite
        use_bmax = opt._dev is None and eng.use_fx3 and self.adam_block_max
        opt.step(grad_scale=grad_scale, norm_is_current=True, zero_grad=clean, block_max=use_bmax)
        self._grad_clean = clean
        # auxiliary loss on the UPDATED parameters (stem/trainSTEM.py:216-218); its gradient goes straight into the aux
        # optimiser's flat buffer (the only aux parameter is `entropy_bottleneck.quantiles`).  One workgroup of latency-bound
        # work on its own stream; the training forward does not read the quantiles, so nothing waits for it until the NEXT
        # optimiser step (above) or until the caller looks at the returned values (LazyScalar).
        F.stream_wait(self._aux_stream, main)
        with F.on_stream(self._aux_stream):
            # private copies (the persistent buffers are overwritten by the next step), made by a library call: recordable
            gn_copy = F.copy_d2d(torch.empty_like(opt._sumsq[:1]), opt._sumsq[:1])
            pack = F.eb_pack(eb._tensors14())
            F.eb_aux_loss_grad(eb.quantiles.detach(), pack, eb.target, eb.quantiles._flat_grad_view, loss_out=self._aux_loss)
            eb.quantiles.grad = eb.quantiles._flat_grad_view
            aux_opt.step()
            aux_copy = F.copy_d2d(torch.empty_like(self._aux_loss), self._aux_loss)
            done = self._done_event
            F.event_record(done, self._aux_stream)
        self._aux_pending = True
        # the next forward's weight packing, issued now; the fp16 images take their scales from the maxima the optimiser pass left
        eng.ensure_packed(block_max=(opt.block_maxima, opt.flat.data) if use_bmax else None)
        loss3 = k["loss3"]
        out = {"y_hat": y_hat, "likelihoods": {"y": lik_y, "z": lik_z}}
        oc = {"y_bpp_loss": loss3[0], "z_bpp_loss": loss3[1], "loss": loss3[2]}
        return out, oc, LazyScalar(aux_copy, done), LazyScalar(gn_copy, done, scale=grad_scale, sqrt=True)

    def after_replay(self):
        """host-side state a replayed step (tape.TapedPFrameStep) leaves as the ordinary one does: the weights changed (the packed
        copies' keys), the auxiliary stream holds work, the gradient buffer was cleared by the optimiser pass"""
        from .layers import bump_weight_epoch
        bump_weight_epoch(self.opt.flat.params)
        self._aux_pending = True

    def finish(self):
        """order everything the step left on its auxiliary stream before the current stream (end of training, checkpointing,
        evaluation: anything that reads `entropy_bottleneck.quantiles` or the aux optimiser's state)"""
        if self._aux_pending:
            F.stream_wait(F.cur_stream(self._aux_loss.device), self._aux_stream)
            self._aux_pending = False


def subsample_septuplet(images, rand):
    """The temporal subsampling of stem/trainSTEM.py:175-182: with probability 1/4 each, frames 1,3,5,7 / 1,4,7 / 1,7 / all seven
    of the septuplet (`rand` = the loop's random.random() draw)."""
    if rand <= 0.25:
        return images[0:7:2]
    if rand <= 0.50:
        return images[0:7:3]
    if rand <= 0.75:
        return images[0:7:6]
    return images


class SeptupletTrainer:
    """The loop body of stem/trainSTEM.py:174-226 for ONE loader item (a septuplet: list of 7 frame batches [B,3,H,W]) -- the unit
    BASELINE.json's metric counts and `bench.py` times:

        frames = subsample_septuplet(images, random.random())          # :175-182
        _, y_condition = IFrameCompressor.getY(frames[0])              # :199 (frozen I-frame model, :128)
        for every later frame:  y_cur = getY(frame) -> stem(y_cur, y_condition.detach()) -> y_condition = y_hat ->
                                criterion -> backward -> clip -> optimizer.step -> aux_loss.backward -> aux_optimizer.step   (:203-218)

    route "taped" (default): the explicit schedule FusedPFrameStep replayed from its launch tape; "fused": the explicit schedule
    issued launch by launch; "generic": nn.Module / autograd / optimiser calls (selfcheck.p_frame_step), which needs `criterion`.
    prefetch: getY of frame t + `ahead` on its own stream while step t runs (LatentPrefetcher); otherwise all latents first.
    reducer / grad_scale: data parallel (distributed.OverlappedGradReducer attached to stem.engine(), 1 / world).
    `rng`: a random.Random for the subsampling draw (seed it per rank and per epoch as the training script seeds `random`)."""

    def __init__(self, imodel, stem, optimizer, aux_optimizer, *, route="taped", prefetch=True, ahead=1, reducer=None, grad_scale=1.0,
                 criterion=None, rng=None):
        import random
        self.imodel, self.stem, self.opt, self.aux_opt = imodel, stem, optimizer, aux_optimizer
        self.route, self.reducer, self.grad_scale, self.criterion = route, reducer, float(grad_scale), criterion
        self.rng = rng or random.Random(0)
        self.step_fn = None
        if route in ("taped", "fused"):
            self.step_fn = FusedPFrameStep(stem, optimizer, aux_optimizer)
            if route == "taped":
                from .tape import TapedPFrameStep
                self.step_fn = TapedPFrameStep(self.step_fn)
        elif route != "generic":
            raise ValueError(f"SeptupletTrainer: unknown route {route!r}")
        elif criterion is None:
            raise ValueError("SeptupletTrainer: the generic route needs the criterion module")
        self.prefetch = LatentPrefetcher(imodel, ahead=ahead) if prefetch else None
        self.on_step = None             # called after every P-frame step with (t, out, criterion_out, aux, grad_norm)

    def train_septuplet(self, images, rand=None, next_images=None, frames_ready=True):
        """One loader item.  rand: the subsampling draw (None: self.rng.random(); 1.0 keeps all seven frames).  next_images: the
        FOLLOWING item's (already subsampled) frames, whose first two latents are then computed during this item's last steps.
        Returns [(criterion_out, aux, grad_norm)] of the P-frame steps (device-resident values; no host synchronisation)."""
        frames = list(subsample_septuplet(list(images), self.rng.random() if rand is None else rand))
        num_pixels = frames[0].shape[0] * frames[0].shape[-2] * frames[0].shape[-1]
        if self.prefetch is not None:
            self.prefetch.start(frames, frames_ready=frames_ready, next_frames=next_images)
            ys = None
            y_cond = self.prefetch.get(0)[1]
        else:
            with torch.no_grad():
                ys = [self.imodel.getY(f) for f in frames]
            y_cond = ys[0][1]
        log = []
        for t in range(1, len(frames)):
            y_cur = self.prefetch.get(t)[0] if self.prefetch is not None else ys[t][0]
            if self.step_fn is not None:
                out, oc, aux, gn = self.step_fn.step(y_cur, y_cond, num_pixels, grad_scale=self.grad_scale, reducer=self.reducer)
            else:
                from .selfcheck import p_frame_step
                out, oc, aux, gn = p_frame_step(self.imodel, self.stem, self.criterion, self.opt, self.aux_opt, frames[t], y_cond,
                                                grad_scale=self.grad_scale, reducer=self.reducer, y_cur=y_cur)
            y_cond = out["y_hat"]
            log.append((oc, aux, gn))
            if self.on_step is not None:
                self.on_step(t, out, oc, aux, gn)
        return log

    def finish(self):
        if self.step_fn is not None:
            self.step_fn.finish()
