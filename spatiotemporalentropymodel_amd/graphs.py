"""hipGraph capture of the fixed P-frame optimisation step (stem/trainSTEM.py:203-218).

One P-frame step of the big configuration is ~160 kernel launches, most of them far shorter than the ~10-20 us the
Python + ctypes + autograd path needs to issue one: measured on MI355X (tools/timeline.py) the GPU sat idle 10 % of a
bench step waiting for the host in the glue between the convolutions, and the host needed 11-22 ms per 32 ms step.  The
schedule is static for a given latent geometry, so it is captured once (torch.cuda.CUDAGraph = hipGraph on ROCm: every
kernel this package launches through the C ABI goes to torch's current stream and is recorded like any other) and
replayed with one host call per step.

What makes the step capturable (kernel ARGUMENTS are frozen at capture, device MEMORY is not):
  * inputs / outputs live in static buffers (`y_cur`, `y_cond` in; `y_hat`, the loss scalars, the gradient norm out);
  * the optimiser's step count and learning rate live in device memory (optim.FusedClipAdam.enable_device_state:
    stem_adam_step_dev), so bias correction advances and an LR scheduler keeps working across replays;
  * the quantisation noise is drawn from the Philox stream at (host offset frozen at capture) + epoch * 2^40 with a
    device-resident epoch that the graph itself increments (stem_uniform_noise_epoch / stem_counter_add): fresh noise on
    every replay;
  * the packed weight copies are rebuilt inside the graph (the weights change every step); the host-side caches that
    track "which weights are packed" are invalidated after every replay so that eager calls stay correct;
  * the weight-gradient side stream forks from and re-joins the capturing stream (event edges in the graph).
Not captured: the frozen I-frame transform `getY` (a handful of long kernels; bench.py brackets its dominant kernel
with HIP events, which cannot be timed inside a graph), data-parallel runs (the RCCL exchanges inside backward stay
eager; see distributed.OverlappedGradReducer) and parity runs with host-injected noise (`noise_source`).
"""
from __future__ import annotations

import torch

from . import functional as F
from .layers import bump_weight_epoch


class GraphedPFrameStep:
    """`step(y_cur, y_cond) -> (out, criterion_out, aux_loss, grad_norm)` with the semantics of selfcheck.p_frame_step minus
    getY: zero_grad -> stem forward -> criterion -> backward -> grad norm -> clip + Adam -> aux loss / aux Adam, replayed
    from one hipGraph.  The returned tensors are the graph's static output buffers: valid until the next step()."""

    def __init__(self, stem, criterion, optimizer, aux_optimizer, target_hw, warmup=2):
        self.stem, self.criterion, self.opt, self.aux_opt = stem, criterion, optimizer, aux_optimizer
        self.target_hw = tuple(target_hw)            # (H, W) of the frames: the criterion only uses it to count pixels
        self.warmup = int(warmup)
        self.graph = None
        self._key = None

    # -------------------------------------------------------------------------------------------
    def _body(self):
        F.counter_add_(self._epoch, 1)
        self.opt.zero_grad()
        self.aux_opt.zero_grad()
        out = self.stem(self._y_cur, self._y_cond)
        oc = self.criterion(out, self._target)
        oc["loss"].backward()
        gn = self.opt.grad_norm()
        self.opt.step(norm_is_current=True)
        aux = self.stem.aux_loss()
        aux.backward()
        self.aux_opt.step()
        return out, oc, aux, gn

    def _capture(self, y_cur, y_cond):
        dev = y_cur.device
        stem = self.stem
        for m in (stem.entropy_bottleneck, stem.gaussian_conditional):
            if m.noise_source is not None:
                raise RuntimeError("GraphedPFrameStep: host-injected noise (noise_source) cannot be captured; run p_frame_step eagerly")
        self.opt.enable_device_state()
        self.aux_opt.enable_device_state()
        self._epoch = torch.zeros(1, dtype=torch.int64, device=dev)
        stem.entropy_bottleneck.noise_epoch = stem.gaussian_conditional.noise_epoch = self._epoch
        B = y_cur.shape[0]
        self._y_cur = torch.empty_like(y_cur, memory_format=torch.channels_last).copy_(y_cur)
        self._y_cond = torch.empty_like(y_cond, memory_format=torch.channels_last).copy_(y_cond)
        self._target = torch.empty((B, 3) + self.target_hw, device=dev)          # shape carrier only (EMLoss: N*H*W)
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        # Warm-up on the capture stream: populates every per-geometry / per-stream cache (split-K workspaces, wgrad gather
        # tables, LDS-size attributes) so that the capture itself records nothing but the steady-state launches.  The
        # warm-up steps run on the caller's data but must not count as training: parameters and optimiser state are
        # saved before and restored after, so the first replay is THE first step.
        saved = [(o, o.flat.data.clone(), o.m.clone(), o.v.clone(), o.t) for o in (self.opt, self.aux_opt)]
        with torch.cuda.stream(stream):
            for _ in range(self.warmup):
                self._body()
            for o, data, m, v, t in saved:
                o.flat.data.copy_(data)
                o.m.copy_(m)
                o.v.copy_(v)
                o.t = t
                o.sync_device_state()
            self._epoch.zero_()
            bump_weight_epoch(self.opt.flat.params)       # weights restored + the capture must CONTAIN the packing launch
            bump_weight_epoch(self.aux_opt.flat.params)
            # host-side Philox offsets as frozen into the graph (tests replay the same stream eagerly)
            self._capture_offsets = {id(m): m._noise_offset for m in (stem.entropy_bottleneck, stem.gaussian_conditional)}
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=stream):
                self._static = self._body()
            for o, _d, _m, _v, t in saved:                  # capturing executed nothing: undo the host-side increments
                o.t = t
        torch.cuda.current_stream(dev).wait_stream(stream)
        self._key = (tuple(y_cur.shape), tuple(y_cond.shape), dev)

    # -------------------------------------------------------------------------------------------
    def step(self, y_cur, y_cond):
        key = (tuple(y_cur.shape), tuple(y_cond.shape), y_cur.device)
        if self.graph is None:
            self._capture(y_cur, y_cond)
        if key != self._key:
            raise RuntimeError(f"GraphedPFrameStep was captured for {self._key[:2]}, got {key[:2]}: one instance per geometry")
        self._y_cur.copy_(y_cur)
        self._y_cond.copy_(y_cond)
        for o in (self.opt, self.aux_opt):               # a scheduler / checkpoint may have changed lr or the step count
            d = o._dev
            lr = float(o.param_groups[0]["lr"])
            if d["lr_host"] != lr:
                d["lr"].fill_(lr)
                d["lr_host"] = lr
        self.graph.replay()
        self.opt.t += 1
        self.aux_opt.t += 1
        bump_weight_epoch(self.opt.flat.params)           # weights moved on the device: host-side pack caches are stale
        bump_weight_epoch(self.aux_opt.flat.params)
        out, oc, aux, gn = self._static
        return ({"y_hat": out["y_hat"].detach(), "likelihoods": {k: v.detach() for k, v in out["likelihoods"].items()}},
                {k: v.detach() for k, v in oc.items()}, aux.detach(), gn)
