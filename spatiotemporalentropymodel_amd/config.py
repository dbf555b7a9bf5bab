"""One runtime configuration object for the hot path (the reference keeps its settings in one argparse
namespace, stem/trainSTEM.py:13-97).

Every switch of the package lives in `StemRuntimeConfig`: which arithmetic route a layer family takes (each pair of routes meets
the same parity gates), how the training step is scheduled over streams, the data-parallel exchange, the decoder's loop form.
The environment is an OVERRIDE, parsed in ONE place (`runtime()`; the variables of `_ENV`; parsed again only when one of them
has changed).  Code asks `runtime().field`; tests and tools change fields through `override(field=value)` (a context manager)
or by setting the variable -- no module reads `os.environ` for these on its own any more.

    from spatiotemporalentropymodel_amd import config
    cfg = config.runtime()                       # defaults + environment overrides
    with config.override(engine_f16x3=False):    # e.g. the fp32-MFMA route for the training layers
        ...                                      # (build the model's engine INSIDE the block: routes are chosen per engine)

`engine.StemEngine`'s switches (`use_fx3`, `overlap_wgrad`, ...) are descriptors that read these fields at look-up time, so an
override block or a changed variable reaches them; assigning `StemEngine.<attr> = value` pins a switch regardless of the
configuration.  Booleans parse "", "0", "false", "no", "off" as False (until round 4 an EMPTY `STEM_ENGINE_*=` counted as enabled).
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
from dataclasses import dataclass


@dataclass
class StemRuntimeConfig:
    # ---- routes: fp16 split-operand kernels (three fp16 MFMAs per fp32 product) or the fp32-MFMA kernels of igemm.hip / wgrad.hip
    analysis_f16x3: bool = True          #: g_a.2 .. g_a.6 of the frozen analysis transform (csrc/conv_f16x3.hip)
    first_layer_f16x3: bool = True       #: g_a.0 + GDN (csrc/c4gdn_f16x3.hip)
    engine_f16x3: bool = True            #: stride-1 STEM layers, forward and input gradient (TPM, HE.0, HD.4, EPM, context model)
    engine_strided_f16x3: bool = True    #: strided faces of the hyper path's stride-2 layers
    engine_transposed_f16x3: bool = True #: ... their transposed faces (four sub-pixel phases, one launch) and weight gradients
    engine_ctx_f16x3: bool = True        #: the masked context convolution over its live taps
    engine_wgrad_f16x3: bool = True      #: weight gradients of the stride-1 layers (csrc/wgrad_f16x3.hip)
    engine_records: bool = True          #: producers of fp32 tensors record their maxima (no maximum pass in front of a split)
    layers_f16x3: bool = True            #: stride-1 convolutions of the layer-wise (variable-rate) models
    layers_f16x3_maxpix: int = 1 << 30   #: ... up to this many pixels per batch
    layers_wide_minpix: int = 32768      #: from this many pixels on (<= 192 outputs) the 192-column kernel
    trainer_overwrite_grads: bool = True #: the explicit P-frame step's backward overwrites the gradient buffer (no clearing pass, no accumulate read)
    adam_block_max: bool = True          #: the optimiser pass leaves per-chunk parameter maxima for the fp16 weight packing
    # ---- schedule of the training step (none of these changes a result)
    engine_overlap: bool = True          #: weight gradients on a side stream
    engine_split_pack: bool = True       #: input-gradient weight images packed on the side stream
    engine_pack_pair: bool = True        #: both images of a layer from one read of its weights (after an optimiser pass with maxima)
    engine_pack_first: bool = False      #: of the forward-role images only the opening layers' on the compute stream (measured: +0.07 ms)
    engine_branch: bool = True           #: hyper path on its own stream
    engine_tpm_first: bool = True        #: temporal-prior chain enqueued ahead of the hyper branch (forward)
    engine_tpm_first_bwd: bool = True    #: ... in backward (the hyper chain waits for the EPM input gradient through an event)
    engine_tpm_wgrad_inline: bool = True #: the TPM chain's weight gradients on the compute stream (behind its input gradients)
    engine_epm_dgrad_by_prior: bool = False #: EPM.0's input gradient range by range, the hyper chain's range first (measured: +0.19 ms)
    engine_share_in_planes: bool = True  #: he_in's planes double as the TPM chain's input (channel view)
    engine_ctx_on_side: bool = False     #: the context model's forward on the weight-gradient stream (experiment)
    engine_ctx_split_on_side: bool = True #: the planes of t_hat (the context model's input) are made on the weight-gradient stream, idle during the forward, instead of between TPM.4 and the context model
    engine_fuse_gc_backward: bool = True #: GaussianConditional backward inside the fused forward glue kernel
    engine_bias_multi: bool = True       #: one launch for a module group's bias-gradient second stages
    stream_prio: str = ""                #: "latents=0,side=-1,compute=-1" (trainer.tuned_schedule installs it)
    stream_cumask: str = ""              #: "latents=block:160"
    # ---- data parallel
    dp_threaded: int = 3                 #: RCCL collectives issued by a helper thread once their producers' events completed (no pending wait in the communication queue): 3 native thread + own communicator (libstem_dp.so), 2 Python thread through torch.distributed, both with a stream flag for the way back; 1 host-blocking finish(); 0 off
    dp_min_bytes: int = 8 << 20          #: a run of final gradients is exchanged once it holds this many bytes
    dist_backend: str = ""               #: "" = RCCL when every rank has its own GPU, else gloo
    dist_single: bool = False            #: a process group at world size 1 (one-GPU boxes execute the RCCL calls)
    pin_ranks: bool = True               #: ranks pin themselves to their GPU's NUMA cores
    # ---- decoder loop (all forms are bit-identical)
    ar_persistent: bool = True
    ar_pipeline: bool = False
    ar_stepwise: bool = False
    ar_force_batch: bool = False
    ar_concurrent: bool = True           #: several images: one persistent decoder per image, up to 8 at once (one XCD, one stream, one host thread each); off: the lockstep batch loop
    ar_no_batch: bool = False


#: field -> environment variable (the historical spellings)
_ENV = {
    "analysis_f16x3": "STEM_F16X3", "first_layer_f16x3": "STEM_C4GDN_F16X3", "engine_f16x3": "STEM_ENGINE_F16X3",
    "engine_strided_f16x3": "STEM_ENGINE_STRIDED_F16X3", "engine_transposed_f16x3": "STEM_ENGINE_TRANSPOSED_F16X3", "engine_ctx_f16x3": "STEM_ENGINE_CTX_F16X3",
    "engine_wgrad_f16x3": "STEM_ENGINE_WGRAD_F16X3", "engine_records": "STEM_ENGINE_RECORDS", "layers_f16x3": "STEM_LAYERS_F16X3",
    "layers_f16x3_maxpix": "STEM_LAYERS_F16X3_MAXPIX", "layers_wide_minpix": "STEM_LAYERS_WIDE_MINPIX",
    "adam_block_max": "STEM_ADAM_BLOCK_MAX", "trainer_overwrite_grads": "STEM_TRAINER_OVERWRITE_GRADS", "engine_overlap": "STEM_ENGINE_OVERLAP",
    "engine_split_pack": "STEM_ENGINE_SPLIT_PACK", "engine_pack_first": "STEM_ENGINE_PACK_FIRST", "engine_pack_pair": "STEM_ENGINE_PACK_PAIR", "engine_branch": "STEM_ENGINE_BRANCH", "engine_tpm_first": "STEM_ENGINE_TPM_FIRST", "engine_tpm_first_bwd": "STEM_ENGINE_TPM_FIRST_BWD", "engine_tpm_wgrad_inline": "STEM_ENGINE_TPM_WGRAD_INLINE",
    "engine_bias_multi": "STEM_ENGINE_BIAS_MULTI", "engine_fuse_gc_backward": "STEM_ENGINE_FUSE_GC_BACKWARD", "engine_ctx_on_side": "STEM_ENGINE_CTX_ON_SIDE", "engine_ctx_split_on_side": "STEM_ENGINE_CTX_SPLIT_ON_SIDE", "engine_share_in_planes": "STEM_ENGINE_SHARE_IN_PLANES", "engine_epm_dgrad_by_prior": "STEM_ENGINE_EPM_DGRAD_BY_PRIOR",
    "stream_prio": "STEM_STREAM_PRIO", "stream_cumask": "STEM_STREAM_CUMASK", "dp_min_bytes": "STEM_DP_MIN_BYTES", "dp_threaded": "STEM_DP_THREADED",
    "dist_backend": "STEM_DIST_BACKEND", "dist_single": "STEM_DIST_SINGLE", "pin_ranks": "STEM_PIN_RANKS",
    "ar_persistent": "STEM_AR_PERSISTENT", "ar_pipeline": "STEM_AR_PIPELINE", "ar_stepwise": "STEM_AR_STEPWISE",
    "ar_force_batch": "STEM_AR_FORCE_BATCH", "ar_concurrent": "STEM_AR_CONCURRENT", "ar_no_batch": "STEM_AR_NO_BATCH",
}


def _parse(kind, text):
    if kind is bool:
        return text.strip() not in ("", "0", "false", "False", "no", "off")
    if kind is int:
        return int(text)
    return text


def from_env(environ=None) -> StemRuntimeConfig:
    """defaults with every `STEM_*` variable of `_ENV` that is set applied on top"""
    environ = os.environ if environ is None else environ
    cfg = StemRuntimeConfig()
    for f in dataclasses.fields(cfg):
        name = _ENV[f.name]
        if name in environ:
            setattr(cfg, f.name, _parse(type(f.default), environ[name]))
    return cfg


_RUNTIME = None
_SNAPSHOT = None
_OVERRIDES = {}


def _snapshot():
    env = os.environ
    return tuple(env.get(v) for v in _ENV.values())


def runtime() -> StemRuntimeConfig:
    """the process's configuration: defaults, the environment's `STEM_*` overrides, then the fields an `override()` block holds.
    The environment is parsed here and nowhere else; it is parsed again only when one of the variables of `_ENV` has changed
    since the last call (tests that set one before exercising a code path)."""
    global _RUNTIME, _SNAPSHOT
    snap = _snapshot()
    if _RUNTIME is None or snap != _SNAPSHOT:
        _RUNTIME, _SNAPSHOT = from_env(), snap
        for k, v in _OVERRIDES.items():
            setattr(_RUNTIME, k, v)
    return _RUNTIME


def reload() -> StemRuntimeConfig:
    """parse the environment again now"""
    global _RUNTIME
    _RUNTIME = None
    return runtime()


@contextlib.contextmanager
def override(**fields):
    """`with override(engine_f16x3=False): ...` -- change fields of the runtime configuration for a block"""
    cfg = runtime()
    unknown = [k for k in fields if not hasattr(cfg, k)]
    if unknown:
        raise AttributeError(f"StemRuntimeConfig has no field(s) {unknown}")
    old = {k: (_OVERRIDES[k] if k in _OVERRIDES else None, k in _OVERRIDES, getattr(cfg, k)) for k in fields}
    try:
        for k, v in fields.items():
            _OVERRIDES[k] = v
            setattr(runtime(), k, v)
        yield runtime()
    finally:
        for k, (ov, had, val) in old.items():
            if had:
                _OVERRIDES[k] = ov
            else:
                _OVERRIDES.pop(k, None)
            setattr(runtime(), k, ov if had else val)
