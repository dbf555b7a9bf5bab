"""ctypes loader for the two C-ABI libraries (include/stem_hip.h, include/stem_rans.h).

The HIP library is the product compute path: there is NO CPU or PyTorch fallback.  If
libstem_hip.so is missing every device op raises (loudly), it never silently degrades.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# STEM_HIP_LIBRARY names another build of the same ABI (tools/debug use `make EXPERIMENTS=1` -> libstem_hip_exper.so)
HIP_SO = os.environ.get("STEM_HIP_LIBRARY") or os.path.join(_PKG, "libstem_hip.so")
# STEM_RANS_LIBRARY names another build of the host codec (`make sanitize` -> libstem_rans_asan.so, the sanitizer test)
RANS_SO = os.environ.get("STEM_RANS_LIBRARY") or os.path.join(_PKG, "libstem_rans.so")

DP_SO = os.environ.get("STEM_DP_LIBRARY") or os.path.join(_PKG, "libstem_dp.so")
_hip = None
_rans = None
_dp = None

vp, ci, cf, sz, u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPE)
_HIP_SIG = {
    "stem_pack_weight": [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "stem_unpack_wgrad": [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "stem_pack_weights_multi": [vp, ci, vp],
    "stem_unpack_wgrads_multi": [vp, ci, vp],
    "stem_bias_grad_final_multi": [vp, ci, vp],
    "stem_conv2d_fwd": [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp, sz, vp],
    "stem_conv2d_fwd_c4": [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_conv2d_dgrad": [vp, ci, vp, vp, ci, vp, ci, cf, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp],
    "stem_conv2d_wgrad": [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_wgrad_splits": [ci, ci, ci, ci, ci, ci, ci],
    "stem_wgrad_workspace_elems": [ci, ci, ci, ci, ci, ci],
    "stem_conv_workspace_bytes": [ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci],
    "stem_deconv2d_fwd": [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp, sz, vp],
    "stem_deconv2d_dgrad": [vp, ci, vp, vp, ci, vp, ci, cf, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp],
    "stem_deconv2d_wgrad": [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_wgrad_bias_parts": [vp, ci, vp, ci, C.c_long, ci, ci, ci, ci, ci],
    "stem_gdn_fwd": [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp],
    "stem_conv2d_gdn_fwd": [vp, ci, vp, vp, vp, vp, vp, ci] + [ci] * 10 + [cf, vp],
    "stem_conv2d_fwd_c4_gdn": [vp, vp, vp, vp, vp, vp, ci] + [ci] * 9 + [cf, vp],
    "stem_deconv2d_gdn_fwd": [vp, ci, vp, vp, vp, vp, vp, ci] + [ci] * 11 + [cf, vp],
    "stem_gdn_bwd": [vp, ci, vp, ci, vp, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, cf, vp, sz, vp],
    "stem_gdn_bwd_workspace_bytes": [ci, ci, ci, ci],
    "stem_lrelu_bwd": [vp, vp, vp, sz, cf, vp],
    "stem_lrelu_fwd": [vp, vp, sz, cf, vp],
    "stem_sft_fwd": [vp, vp, vp, vp, sz, cf, vp],
    "stem_sft_bwd": [vp, vp, vp, vp, vp, vp, vp, sz, cf, vp],
    "stem_avgpool_fwd": [vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_avgpool_bwd": [vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_crop_u8_to_f32": [vp, vp, vp, ci, ci, ci, ci, ci, vp],
    "stem_qmap_params_per_sample": [],
    "stem_qmap_render": [vp, vp, ci, ci, cf, vp],
    "stem_weighted_sqerr_sum": [vp, vp, vp, ci, ci, sz, vp, vp],
    "stem_weighted_sqerr_bwd": [vp, vp, vp, vp, ci, ci, sz, vp, cf, vp],
    "stem_nchw_to_nhwc": [vp, vp, ci, ci, ci, ci, ci, vp],
    "stem_nhwc_to_nchw": [vp, ci, vp, ci, ci, ci, ci, ci, vp],
    "stem_nchw3_to_nhwc4": [vp, vp, ci, ci, ci, vp, vp],
    "stem_nhwc4_qrec_floats": [ci, ci, ci],
    "stem_copy_channels": [vp, ci, vp, ci, sz, ci, vp],
    "stem_eb_pack": [vp, vp, ci, vp],
    "stem_eb_unpack_grads": [vp, vp, ci, ci, vp],
    "stem_eb_forward": [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, vp],
    "stem_eb_backward": [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp],
    "stem_eb_backward_rec": [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp],
    "stem_eb_aux_loss": [vp, vp, vp, vp, vp, ci, vp],
    "stem_gc_forward": [vp, vp, vp, vp, ci, vp, vp, sz, ci, ci, cf, cf, vp],
    "stem_gc_backward": [vp, vp, vp, ci, vp, vp, vp, ci, vp, sz, ci, cf, cf, vp, vp],
    "stem_log2_sum": [vp, sz, vp, vp],
    "stem_dlog": [vp, vp, sz, cf, vp],
    "stem_sub": [vp, vp, vp, sz, vp],
    "stem_add": [vp, vp, vp, sz, vp],
    "stem_round": [vp, vp, sz, vp],
    "stem_uniform_noise": [vp, sz, u64, u64, vp],
    "stem_uniform_noise_epoch": [vp, sz, u64, u64, vp, u64, vp],
    "stem_counter_add": [vp, C.c_longlong, vp],
    "stem_prior_prologue": [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, u64, u64, vp, u64, sz, ci, ci, ci, vp, vp, vp],
    "stem_rate_partials": [sz],
    "stem_eb_forward_train": [vp, ci, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, sz, ci, cf, cf, vp],
    "stem_eb_forward_train_rec": [vp, ci, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, sz, ci, cf, cf, vp, vp],
    "stem_gc_forward_train": [vp, vp, vp, ci, vp, u64, u64, vp, u64, vp, vp, vp, vp, sz, ci, cf, cf, cf, vp],
    "stem_gc_forward_backward_train": [vp, vp, vp, ci, vp, u64, u64, vp, u64, vp, vp, vp, vp, sz, ci, cf, cf, cf, vp, vp, ci, vp, vp],
    "stem_em_loss_finalize": [vp, ci, vp, ci, C.c_double, vp, vp],
    "stem_eb_aux_loss_grad": [vp, vp, vp, vp, vp, ci, ci, vp],
    "stem_build_indexes": [vp, ci, vp, ci, vp, sz, ci, cf, vp],
    "stem_gemv3": [vp, ci, vp, vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, ci, ci, cf, vp],
    "stem_pack_ctx_gemv": [vp, vp, ci, ci, vp],
    "stem_ar_finish_encode": [vp, vp, ci, cf, vp, vp, vp, ci, vp],
    "stem_ar_index": [vp, vp, ci, cf, vp, ci, vp],
    "stem_ar_finish_decode": [vp, vp, vp, ci, vp],
    "stem_gemv3_wave": [vp, ci, vp, vp, vp, ci, ci, ci, cf, ci, ci, ci, vp],
    "stem_ar_finish_encode_wave": [vp, vp, ci, cf, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "stem_gemv3_decode": [vp, ci, vp, vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, ci, ci, cf, vp, vp, vp, ci, ci, vp, ci, cf, vp, vp],
    "stem_ar_decode_image": [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf,
                             vp, vp, vp, vp, vp, ci, ci, vp, vp, vp],
    "stem_f16x2_planes_bytes": [C.c_long, ci],
    "stem_f16x2_conv_weight_bytes": [ci, ci, ci],
    "stem_f16x2_planes_qrec_offset": [C.c_long, ci],
    "stem_amax_nhwc": [vp, ci, C.c_long, ci, vp, C.c_long, vp],
    "stem_f16x2_split_nhwc": [vp, ci, vp, vp, vp, C.c_long, ci, vp],
    "stem_f16x2_merge_nhwc": [vp, vp, vp, ci, C.c_long, ci, vp],
    "stem_f16x2_split_dact_nhwc": [vp, ci, vp, ci, cf, vp, vp, vp, C.c_long, ci, vp],
    "stem_f16x2_pack_conv_weight": [vp, vp, ci, ci, ci, ci, vp],
    "stem_f16x2_conv_weight_gen_bytes": [ci, ci, ci, ci],
    "stem_f16x2_pack_conv_weight_gen": [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "stem_conv2d_f16x3_gen_workspace_bytes": [ci, ci, ci, ci, ci, ci, ci, ci, ci, ci],
    "stem_conv2d_f16x3_gen_fwd_rows": [vp, vp, ci, vp, ci, ci, vp, ci, cf, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp],
    "stem_tconv2d_f16x3_workspace_bytes": [ci, ci, ci, ci, ci, ci],
    "stem_tconv2d_f16x3_fwd": [vp, vp, ci, vp, vp, ci, cf, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp],
    "stem_wgrad_f16x3_strided_splits": [ci, ci, ci, ci, ci, ci, ci, ci, ci],
    "stem_conv2d_wgrad_f16x3_strided": [vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_f16x2_pack_conv_weights_multi": [vp, ci, vp],
    "stem_f16x2_pack_conv_weights_pair_multi": [vp, ci, vp],
    "stem_wgrad_f16x3_splits": [ci, ci, ci, ci, ci, ci, ci, ci],
    "stem_conv2d_wgrad_f16x3": [vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_bias_grad_final": [vp, ci, ci, vp, ci, vp],
    "stem_bias_grad_scratch_elems": [C.c_long, ci],
    "stem_bias_grad": [vp, ci, C.c_long, ci, vp, vp, ci, vp],
    "stem_conv2d_f16x3_gen_fwd": [vp, vp, ci, vp, vp, ci, cf, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp],
    "stem_c4gdn_supported": [ci, ci, ci],
    "stem_c4gdn_stream_bytes": [ci, ci, ci],
    "stem_c4gdn_pack": [vp, vp, vp, ci, ci, ci, vp],
    "stem_conv2d_c4_gdn_f16x3": [vp, vp, vp, vp, vp, cf, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_conv2d_f16x3_fwd_act": [vp, vp, vp, vp, ci, cf, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_f16x2_pack_conv_weight_flip": [vp, vp, ci, ci, ci, ci, vp],
    "stem_conv2d_f16x3_fwd": [vp, vp, vp, vp, vp, vp, cf, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "stem_ar_decode_image_persistent_supported": [ci, ci, ci],
    "stem_ar_decode_image_persistent_prefer_xcc": [ci],
    "stem_ar_decode_image_persistent": [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf,
                                        vp, vp, vp, ci, ci, vp, vp, vp],
    "stem_ar_decode_batch_pipelined": [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf,
                                       vp, vp, vp, ci, ci, vp, vp, vp],
    "stem_ar_decode_batch": [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf,
                             vp, vp, vp, vp, vp, ci, ci, vp, vp, vp],
    "stem_ar_encode_image": [vp, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf,
                             vp, vp, vp],
    "stem_sumsq": [vp, sz, vp, vp],
    "stem_sumsq_set": [vp, sz, vp, vp],
    "stem_clip_scale": [vp, sz, vp, cf, vp],
    "stem_axpy": [vp, vp, cf, sz, vp],
    "stem_adam_step": [vp, vp, vp, vp, sz, vp, cf, cf, cf, cf, cf, cf, ci, vp],
    "stem_adam_step_zero": [vp, vp, vp, vp, sz, vp, cf, cf, cf, cf, cf, cf, ci, vp],
    "stem_adam_step_dev": [vp, vp, vp, vp, sz, vp, cf, cf, vp, cf, cf, cf, vp, vp, vp],
    "stem_packed_weight_elems": [ci, ci, ci, ci, ci],
    "stem_abi_version": [],
    "stem_adam_chunk": [],
    "stem_adam_step_bmax": [vp, vp, vp, vp, sz, vp, cf, cf, cf, cf, cf, cf, ci, ci, vp, vp],
    "stem_built_with_experiments": [],
    "stem_tape_create": [],
    "stem_tape_destroy": [vp],
    "stem_tape_length": [vp],
    "stem_tape_add_call": [vp, vp, ci, vp, vp, vp, vp],
    "stem_tape_add_wait": [vp, vp, vp],
    "stem_tape_add_event": [vp, vp, vp, ci],
    "stem_tape_replay": [vp, ci, ci, C.c_longlong],
    "stem_tape_set_iarg": [vp, ci, ci, C.c_longlong],
    "stem_tape_set_farg": [vp, ci, ci, C.c_double],
    "stem_tape_entry_recordable": [vp],
    "stem_zero_bytes": [vp, sz, vp],
    "stem_copy_d2d": [vp, vp, sz, vp],
    "stem_stream_flag_create": [vp],
    "stem_stream_flag_destroy": [vp],
    "stem_stream_flag_wait_ge": [vp, C.c_uint, vp],
    "stem_stream_flag_write": [vp, C.c_uint, vp],
    "stem_tuning_set": [C.c_char_p, ci],
    "stem_tuning_get": [C.c_char_p],
    "stem_last_error": [],
}
_RESTYPE = {"stem_c4gdn_stream_bytes": sz, "stem_adam_chunk": sz, "stem_f16x2_planes_qrec_offset": sz, "stem_nhwc4_qrec_floats": sz, "stem_bias_grad_scratch_elems": sz, "stem_f16x2_conv_weight_gen_bytes": sz, "stem_conv2d_f16x3_gen_workspace_bytes": sz, "stem_tconv2d_f16x3_workspace_bytes": sz, "stem_f16x2_planes_bytes": sz, "stem_f16x2_conv_weight_bytes": sz, "stem_packed_weight_elems": sz, "stem_gdn_bwd_workspace_bytes": sz, "stem_wgrad_workspace_elems": sz, "stem_conv_workspace_bytes": sz, "stem_last_error": C.c_char_p,
             "stem_tape_create": vp, "stem_tape_destroy": None}

_RANS_SIG = {
    "stem_rans_encode": [vp, vp, sz, vp, ci, ci, vp, vp, vp, sz],
    "stem_rans_decode": [vp, sz, vp, sz, vp, ci, ci, vp, vp, vp],
    "stem_rans_encoder_create": [],
    "stem_rans_encoder_destroy": [vp],
    "stem_rans_encoder_push": [vp, vp, vp, sz, vp, ci, ci, vp, vp],
    "stem_rans_encoder_flush": [vp, vp, sz],
    "stem_rans_encoder_pending_bytes": [vp],
    "stem_rans_decoder_create": [],
    "stem_rans_decoder_destroy": [vp],
    "stem_rans_decoder_set_stream": [vp, vp, sz],
    "stem_rans_decoder_decode": [vp, vp, sz, vp, ci, ci, vp, vp, vp],
    "stem_pmf_to_quantized_cdf": [vp, ci, ci, vp],
    "stem_rans_last_error": [],
}
_RANS_RESTYPE = {"stem_rans_encode": C.c_long, "stem_rans_encoder_flush": C.c_long, "stem_rans_encoder_create": vp,
                 "stem_rans_decoder_create": vp, "stem_rans_last_error": C.c_char_p,
                 "stem_rans_encoder_pending_bytes": sz, "stem_rans_encoder_destroy": None, "stem_rans_decoder_destroy": None}


class PackDesc(C.Structure):
    _fields_ = [("w", vp), ("wp", vp), ("K", ci), ("C", ci), ("R", ci), ("S", ci), ("role", ci), ("masked", ci)]


class F16PackDesc(C.Structure):
    _fields_ = [("w", vp), ("wp", vp), ("N", ci), ("C", ci), ("R", ci), ("S", ci), ("flip", ci), ("taps", ci), ("bmax", vp), ("b0", ci), ("nb", ci), ("rsv0", ci), ("rsv1", ci)]


class F16PairDesc(C.Structure):
    _fields_ = [("w", vp), ("A", ci), ("B", ci), ("R", ci), ("S", ci), ("wp0", vp), ("mode0", ci), ("taps0", ci), ("wp1", vp), ("mode1", ci),
                ("taps1", ci), ("bmax", vp), ("b0", ci), ("nb", ci)]


class UnpackDesc(C.Structure):
    _fields_ = [("dwp", vp), ("dw", vp), ("K", ci), ("C", ci), ("R", ci), ("S", ci), ("splits", ci), ("flags", ci)]


class BiasFinalDesc(C.Structure):
    _fields_ = [("part", vp), ("db", vp), ("K", ci), ("parts", ci), ("accumulate", ci), ("reserved", ci)]


class WaveSeg(C.Structure):
    _fields_ = [("x", vp), ("len", ci), ("woff", ci), ("sh", C.c_long), ("sw", C.c_long), ("sp", C.c_long)]


class StemLibraryError(RuntimeError):
    pass


def _bind(lib, sigs, restypes):
    for name, args in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = restypes.get(name, C.c_int)
    return lib


def hip():
    """libstem_hip.so, or raise: the product path has no fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise StemLibraryError(
                f"{HIP_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the STEM kernels.")
        _hip = _bind(C.CDLL(HIP_SO), _HIP_SIG, _RESTYPE)
    return _hip


def rans():
    global _rans
    if _rans is None:
        if not os.path.exists(RANS_SO):
            raise StemLibraryError(f"{RANS_SO} is missing: run `make -C {os.path.join(_PKG, 'csrc')}`")
        _rans = _bind(C.CDLL(RANS_SO), _RANS_SIG, _RANS_RESTYPE)
    return _rans


def check(rc: int):
    if rc != 0:
        raise RuntimeError((hip().stem_last_error() or b"").decode() or f"libstem_hip error {rc}")


def declared_hip_symbols():
    return sorted(_HIP_SIG)


_DP_SIG = {
    "stem_dp_unique_id": [C.c_void_p],
    "stem_dp_create": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int],
    "stem_dp_prepare": [C.c_void_p, C.c_int],
    "stem_dp_connect": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "stem_dp_nranks": [C.c_void_p],
    "stem_dp_abort": [C.c_void_p, C.c_int, C.c_char_p],
    "stem_dp_submit": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t],
    "stem_dp_fence": [C.c_void_p, C.c_void_p],
    "stem_dp_status": [C.c_void_p],
    "stem_dp_destroy": [C.c_void_p],
    "stem_dp_last_error": [],
}


def dp():
    """libstem_dp.so (include/stem_dp.h): the native RCCL issue path of a data-parallel rank; links librccl"""
    global _dp
    if _dp is None:
        if not os.path.exists(DP_SO):
            raise StemLibraryError(f"{DP_SO} is missing: run `make -C {os.path.join(_PKG, 'csrc')}`")
        _dp = _bind(C.CDLL(DP_SO), _DP_SIG, {"stem_dp_last_error": C.c_char_p})
    return _dp


def declared_dp_symbols():
    return sorted(_DP_SIG)


def declared_rans_symbols():
    return sorted(_RANS_SIG)
