"""Flat parameter / gradient buffers and the fused clip + Adam step.

The reference's per-step optimiser work (stem/trainSTEM.py:213-218, utils.py:104-135) is
`clip_grad_norm_` over 41 tensors followed by two torch Adam steps: ~250 small kernels.  Here every
STEM parameter is a view into ONE flat fp32 buffer (and its gradient a view into a second one), so the
global norm is one reduction, clip + Adam one elementwise pass, and the data-parallel gradient
exchange one RCCL all-reduce over the same buffer (distributed.py).
"""
from __future__ import annotations

import torch

from . import functional as F
from .layers import bump_weight_epoch, join_wgrad_stream


#: module groups of the STEM models in the order the explicit backward makes their gradients FINAL (engine.StemEngine.backward:
#: entropy-parameter network, context model, temporal prior, hyper decoder + bottleneck, hyper encoder)
BACKWARD_ORDER = ("EPM.", "context_prediction.", "TPM.", "HD.", "entropy_bottleneck.", "HE.")


def backward_layout_key(name):
    """sort key that lays parameters out group by group in backward-completion order (names inside a group stay sorted): groups
    that become final one after the other are then NEIGHBOURS in the flat gradient buffer, so the data-parallel reducer
    (distributed.OverlappedGradReducer) merges small ones into its runs instead of leaving them for a collective of their own at
    the end of backward (the bottleneck's 60 KB used to travel alone, after everything else)"""
    for i, pre in enumerate(BACKWARD_ORDER):
        if name.startswith(pre):
            return (i, name)
    return (len(BACKWARD_ORDER), name)


class FlatParameters:
    """Re-homes `params` into one contiguous buffer; `.grad`s become views of a second buffer.  `names` / `params` / `offsets` keep
    the order of `named_params` (the optimiser's parameter order: utils.py:104-135 sorts by name, and state dicts index by it);
    `layout_key(name)` only decides where each tensor sits INSIDE the buffers."""

    def __init__(self, named_params, layout_key=None):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        dev = self.params[0].device
        order = sorted(range(len(self.params)), key=(lambda i: layout_key(self.names[i])) if layout_key else None)
        self.offsets, off = [0] * len(self.params), 0
        for i in order:
            self.offsets[i] = off
            off += (self.params[i].numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned for the float4 kernels
        self.numel = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p._flat_grad_view = self.grad[o:o + n].view(p.shape)
            p.grad = p._flat_grad_view
        bump_weight_epoch(self.params)

    def zero_grad(self):
        join_wgrad_stream()
        if self.grad.is_cuda:
            F.zero_bytes(self.grad)          # a library call: a launch tape records it (Tensor.zero_() would be dropped on replay)
        else:
            self.grad.zero_()
        for p in self.params:
            p.grad = p._flat_grad_view


class FusedClipAdam(torch.optim.Optimizer):
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps) over a FlatParameters in two kernel launches.
    `max_norm=None` disables clipping (the reference does not clip the aux optimiser's `.quantiles`).

    It IS a torch.optim.Optimizer: `ReduceLROnPlateau(optimizer, "min")` (stem/trainSTEM.py:123) attaches to it and its
    `param_groups[0]["lr"]` is what the step uses; `state_dict()` / `load_state_dict()` speak torch.optim.Adam's layout
    (per-parameter "step" / "exp_avg" / "exp_avg_sq" + param_groups), so the `"optimizer"` / `"aux_optimizer"` entries of
    a reference checkpoint (stem/trainSTEM.py:286-297) load here and ours load into torch.optim.Adam.  The moments live
    in two flat buffers; the per-parameter state tensors are views into them."""

    def __init__(self, flat: FlatParameters, lr, max_norm=None, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=0, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(flat.params, defaults)
        self.flat, self.max_norm = flat, max_norm
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.t = 0
        self._sumsq = F.sumsq_accumulator(flat.data.device)          # [0] = sum of squares, rest = reduction scratch
        self._dev = None             # device-resident (step, lr, scratch) once enable_device_state() was called

    # the scalar hyper-parameters live in param_groups[0] (schedulers and checkpoints edit them there)
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def grad_norm(self):
        """Global L2 norm of the current flat gradient as a 0-dim device tensor (no host sync)."""
        join_wgrad_stream()
        self._sumsq[:1].zero_()
        F.sumsq(self.flat.grad, self._sumsq)
        return self._sumsq[0].sqrt().reshape(())

    def enable_device_state(self):
        """Keep the step count and the learning rate in device memory (stem_adam_step_dev), which is what makes `step()`
        capturable in a hipGraph: kernel arguments are frozen at capture, device memory is not.  `self.t` and
        `param_groups[0]["lr"]` stay the source of truth on the host: they are mirrored to the device by step() /
        sync_device_state() (a replayed graph advances the device counter itself; graphs.GraphedPFrameStep advances
        `self.t` alongside)."""
        if self._dev is None:
            dev = self.flat.data.device
            self._dev = {"step": torch.zeros(1, dtype=torch.int64, device=dev), "lr": torch.zeros(1, dtype=torch.float32, device=dev),
                         "scal": torch.zeros(2, dtype=torch.float32, device=dev), "lr_host": None}
        self.sync_device_state()
        return self

    def sync_device_state(self):
        """host -> device: step count always, learning rate when a scheduler / checkpoint changed it"""
        d = self._dev
        if d is None:
            return
        d["step"].fill_(self.t)
        lr = float(self.param_groups[0]["lr"])
        if d["lr_host"] != lr:
            d["lr"].fill_(lr)
            d["lr_host"] = lr

    def step(self, closure=None, *, grad_scale: float = 1.0, norm_is_current: bool = False, zero_grad: bool = False,
             block_max: bool = False):
        """grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).  norm_is_current: the caller
        has just called grad_norm() on these very gradients (as the training loop does to report the norm), so the
        reduction is not repeated.  zero_grad: clear the flat gradient buffer in the same pass (explicit schedules that
        accumulate into it next step; not available with device-resident state).  block_max: the pass also leaves the
        maximum |parameter| of every chunk of F.adam_chunk() parameters in `self.block_maxima` (host-resident state only):
        valid until the parameters change again."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        join_wgrad_stream()
        self.t += 1
        g = self.param_groups[0]
        use_clip = self.max_norm is not None and self.max_norm > 0
        if use_clip and not norm_is_current:
            self._sumsq[:1].zero_()
            F.sumsq(self.flat.grad, self._sumsq)
        if self._dev is not None:
            d = self._dev
            if not torch.cuda.is_current_stream_capturing():
                if d["lr_host"] != float(g["lr"]):
                    d["lr"].fill_(float(g["lr"]))
                    d["lr_host"] = float(g["lr"])
            F.adam_step_dev(self.flat.data, self.flat.grad, self.m, self.v, self._sumsq if use_clip else None,
                            float(self.max_norm) if use_clip else 0.0, float(grad_scale), d["lr"], g["betas"][0], g["betas"][1],
                            float(g["eps"]), d["step"], d["scal"])
        elif block_max:
            if getattr(self, "block_maxima", None) is None:
                ch = F.adam_chunk()
                self.block_maxima = torch.empty(4 * ((self.flat.data.numel() + ch - 1) // ch), device=self.flat.data.device, dtype=torch.float32)
            F.adam_step_bmax(self.flat.data, self.flat.grad, self.m, self.v, self._sumsq if use_clip else None,
                             float(self.max_norm) if use_clip else 0.0, float(grad_scale), float(g["lr"]), g["betas"][0], g["betas"][1],
                             float(g["eps"]), self.t, self.block_maxima, zero_grad=zero_grad)
        else:
            F.adam_step(self.flat.data, self.flat.grad, self.m, self.v, self._sumsq if use_clip else None,
                        float(self.max_norm) if use_clip else 0.0, float(grad_scale), float(g["lr"]), g["betas"][0], g["betas"][1],
                        float(g["eps"]), self.t, zero_grad=zero_grad)
        if self._dev is None:
            # a launch tape replays the call above: it must read these values again on every replay, not keep the recorded ones
            # (whether there is clipping at all decides a pointer argument: that is part of the tape's key, tape.TapedPFrameStep)
            gs = float(grad_scale)
            F.tape_bind_floats(lambda: (float(self.max_norm) if use_clip else 0.0, gs, float(g["lr"]), float(g["betas"][0]),
                                        float(g["betas"][1]), float(g["eps"])))
        bump_weight_epoch(self.flat.params)
        return loss

    # ---- torch.optim.Adam-compatible (de)serialisation ----------------------------------------------------------
    def _views(self, buf):
        return [buf[o:o + p.numel()].view(p.shape) for p, o in zip(self.flat.params, self.flat.offsets)]

    def _sync_state(self):
        """Expose the flat moments as Adam's per-parameter state (views, no copies); empty before the first step, as
        torch.optim.Adam's is."""
        self.state.clear()
        if self.t == 0:
            return
        for p, m, v in zip(self.flat.params, self._views(self.m), self._views(self.v)):
            self.state[p] = {"step": torch.tensor(float(self.t)), "exp_avg": m, "exp_avg_sq": v}

    def state_dict(self):
        self._sync_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError(f"FusedClipAdam: expected one parameter group of {len(self.flat.params)} tensors "
                             f"(utils.py:104-135 order: sorted parameter names), got {[len(g['params']) for g in groups]}")
        super().load_state_dict(state_dict)          # validates, casts to the parameters' device, fills self.state
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for p, m, v in zip(self.flat.params, self._views(self.m), self._views(self.v)):
            st = self.state.get(p)
            if not st:
                continue
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))        # int in torch 1.7 checkpoints, 0-dim tensor since 1.12
        if len(steps) > 1:
            raise ValueError(f"FusedClipAdam: parameters carry different step counts {sorted(steps)}; the fused step keeps one")
        self.t = steps.pop() if steps else 0
        self._sync_state()
        self.sync_device_state()


def clip_grad_norm_(optimizers, max_norm, tensors=None):
    """torch.nn.utils.clip_grad_norm_ over the union of the optimisers' flat gradient buffers, without a host sync:
    one reduction per buffer into a shared fp64 accumulator, then one scaling pass per buffer.  Returns the norm
    before clipping as a 0-dim device tensor.  (The variable-rate training loop clips after every frame while the
    gradients of a GOP keep accumulating: stem_roi/train_stem_roi.py:536,563.)  `tensors` overrides which flat
    buffers are clipped (the data-parallel loop clips the running global sums, distributed.GopGradAccumulator)."""
    join_wgrad_stream()
    optimizers = [o for o in optimizers if o is not None]
    bufs = list(tensors) if tensors is not None else [o.flat.grad for o in optimizers]
    acc = optimizers[0]._sumsq
    acc[:1].zero_()
    for g in bufs:
        F.sumsq(g, acc)
    for g in bufs:
        F.clip_scale(g, acc, max_norm)
    return acc[0].sqrt().reshape(())


def configure_optimizers(net, args, fused=True, max_norm=1.0):
    """utils.py:104-135: main Adam over everything but `.quantiles`, aux Adam over `.quantiles`.
    fused=True returns FusedClipAdam objects (clip folded into the main step); fused=False returns the
    reference's plain torch.optim.Adam pair."""
    named = list(net.named_parameters())
    main = sorted([(n, p) for n, p in named if not n.endswith(".quantiles") and p.requires_grad], key=lambda t: t[0])
    aux = sorted([(n, p) for n, p in named if n.endswith(".quantiles") and p.requires_grad], key=lambda t: t[0])
    assert len(main) + len(aux) == len([1 for _, p in named if p.requires_grad])
    if not fused:
        return (torch.optim.Adam((p for _, p in main), lr=args.learning_rate),
                torch.optim.Adam((p for _, p in aux), lr=args.aux_learning_rate))
    return (FusedClipAdam(FlatParameters(main, layout_key=backward_layout_key), args.learning_rate, max_norm=max_norm),
            FusedClipAdam(FlatParameters(aux), args.aux_learning_rate, max_norm=None))
