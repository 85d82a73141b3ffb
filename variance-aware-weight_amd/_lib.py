"""ctypes binding of libvaw_hip.so (include/vaw_hip.h).  No fallback: if the library is missing,
or a call is made without a GPU tensor, this raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAW_HIP_LIB") or os.path.join(_HERE, "libvaw_hip.so")   # (override: measurement builds only)

F32, BF16, FP8, BF8 = 0, 1, 2, 3
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Epilogue(C.Structure):
    _fields_ = [("bias", _p), ("act", _i), ("aux_in", _p), ("aux_out", _p), ("gate", _p), ("gate_ld", _l),
                ("resid", _p), ("rowadd", _p), ("rows_per_batch", _i), ("alpha", _f), ("beta", _f), ("out_f32", _i),
                ("colsum_out", _p), ("colsum_beta", _f), ("resid_is_act", _i), ("rowsum_a_out", _p), ("rowsum_a_beta", _f),
                ("colsum_partial_out", _p), ("colsum_rows_out", C.POINTER(C.c_int64))]


class AttnDesc(C.Structure):
    _fields_ = [("B", _i), ("H", _i), ("T", _i), ("hd", _i), ("q_sb", _l), ("q_sh", _l), ("q_st", _l), ("q_sd", _l),
                ("o_sb", _l), ("o_sh", _l), ("o_st", _l), ("o_sd", _l), ("scale", _f)]


# name -> argtypes (every function returns int status unless noted)
_PROTOS = {
    "vaw_qsample_fwd": [_p, _p, _p, _p, _p, _i, _p, _i, _l, _p],
    "vaw_mix_rows": [_p, _p, _p, _p, _p, _i, _l, _p],
    "vaw_wmse_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _l, _p],
    "vaw_wmse_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _p],
    "vaw_gemm": [_i, _i, _i, _l, _l, _l, _p, _l, _p, _l, _p, _l, C.POINTER(Epilogue), _p, _l, _p],
    "vaw_gemm_uses_bf16_mfma": [_i, _l, _l, _l, _p, _l, _p, _l],
    "vaw_colsum": [_i, _p, _l, _l, _l, _p, _f, _p, _l, _p],
    "vaw_ln_modulate_fwd": [_i, _p, _p, _p, _l, _p, _p, _p, _i, _i, _i, _f, _p],
    "vaw_ln_modulate_bwd": [_i, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l, _i, _i, _i, _p, _l, _p],
    "vaw_gate_bwd": [_i, _p, _p, _p, _l, _p, _p, _l, _p, _i, _i, _i, _p, _l, _p],
    "vaw_reduce_rows": [_p, _l, _l, _p, _f, _p],
    "vaw_reduce_rows_batched": [_i, _p, _f, _p, _i, _p],
    "vaw_ln_modulate_bwd_gate": [_i, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _p, _l, _p],
    "vaw_ln_modulate_bwd_gate_fp8": [_p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _l, _p],
    "vaw_patchify": [_i, _p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_patchify_bwd": [_p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_unpatchify": [_i, _p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_unpatchify_bwd": [_i, _p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_timestep_embedding": [_i, _p, _p, _i, _i, _f, _p],
    "vaw_silu_fwd": [_i, _p, _p, _l, _p],
    "vaw_silu_bwd": [_p, _p, _p, _l, _p],
    "vaw_add_embedding": [_p, _p, _p, _p, _i, _i, _i, _p],
    "vaw_embedding_bwd": [_p, _p, _p, _i, _i, _i, _f, _p],
    "vaw_attn_fwd": [_i, C.POINTER(AttnDesc), _p, _p, _p, _p, _p, _p],
    "vaw_attn_bwd": [_i, C.POINTER(AttnDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "vaw_attn_bwd_colsum": [_i, C.POINTER(AttnDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(C.c_int64), _p],
    "vaw_comm_unique_id": [_p],
    "vaw_comm_init": [_p, _i, _i],
    "vaw_comm_world": [],
    "vaw_comm_destroy": [],
    "vaw_allreduce_bucket_start": [_p, _l, _i, _p],
    "vaw_reduce_scatter_bucket_start": [_p, _l, _i, _p],
    "vaw_allgather_bucket_start": [_p, _l, _i, _p],
    "vaw_allreduce_bucket_wait": [_p],
    "vaw_groupnorm_fwd": [_i, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p],
    "vaw_groupnorm_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _f, _p, _p, _l, _i, _i, _i, _i, _p, _p],
    "vaw_im2col3x3": [_i, _p, _p, _i, _i, _i, _i, _p],
    "vaw_col2im3x3": [_i, _p, _p, _i, _i, _i, _i, _p],
    "vaw_conv3x3": [_i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(Epilogue), _p, _l, _p],
    "vaw_vb_fwd": [_p, _p, _p, _p, _p, _i, _i, _f, _p, _i, _l, _p],
    "vaw_vb_bwd": [_p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _i, _l, _p],
    "vaw_sample_step": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _i, _l, _p],
    "vaw_conv3x3_narrow": [_i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_conv3x3_wgrad_small": [_i, _p, _p, _p, _f, _i, _i, _i, _i, _i, _p, _l, _p],
    "vaw_resample2": [_i, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vaw_concat_channels": [_i, _p, _p, _p, _l, _i, _i, _i, _p],
    "vaw_add_inplace": [_i, _p, _p, _l, _p],
    "vaw_mul": [_i, _p, _p, _p, _l, _p],
    "vaw_subsample2": [_i, _p, _p, _i, _i, _i, _i, _i, _p],
    "vaw_rowvec_add": [_i, _p, _p, _l, _i, _i, _i, _p],
    "vaw_rowvec_sum": [_i, _p, _p, _l, _i, _i, _i, _f, _p],
    "vaw_nchw_to_nhwc": [_i, _p, _p, _i, _i, _i, _p],
    "vaw_nhwc_to_nchw": [_i, _p, _p, _i, _i, _i, _p],
    "vaw_sumsq": [_p, _l, _p, _i, _p, _p],
    "vaw_adamw_ema_step_dev": [_p, _p, _p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _f, _p, _f, _i, _p],
    "vaw_adamw_ema_step": [_p, _p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _f, _f, _p, _f, _i, _p],
    "vaw_ema_update": [_p, _p, _l, _f, _p],
    "vaw_cast_bf16": [_p, _p, _l, _p],
    "vaw_uncast_bf16": [_p, _p, _l, _f, _p],
    "vaw_wgrad_grouped": [_i, _i, _p, _l, _f, _p, _i, _p, _l, _p],
    "vaw_fp8_quantize": [_i, _i, _p, _l, _l, _l, _p, _l, _p, _l, _p, _p, _l, _p],
    "vaw_fp8_quantize_delayed": [_i, _i, _p, _l, _l, _l, _p, _l, _p, _l, _p, _p],
    "vaw_fp8_scale_update": [_p, _l, _p],
    "vaw_fp8_quantize_delayed_batched": [_i, _p, _p, _i, _p],
    "vaw_gemm_fp8": [_i, _l, _l, _l, _p, _l, _p, _p, _l, _p, _p, _l, C.POINTER(Epilogue), _p, _i, _p, _l, _p],
    "vaw_fp8_transpose": [_p, _l, _l, _l, _p, _l, _p],
    "vaw_ln_modulate_fwd_fp8": [_p, _p, _p, _l, _p, _p, _i, _p, _p, _i, _i, _i, _f, _p],
    "vaw_gate_bwd_fp8": [_p, _p, _p, _l, _p, _p, _i, _p, _l, _p, _i, _i, _i, _p, _l, _p],
}

_lib = None


class VawError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL.  Raises if the HIP library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VawError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH)
        for name, args in _PROTOS.items():
            if not hasattr(L, name) and os.environ.get("VAW_HIP_LIB"):
                continue          # an older measurement build behind VAW_HIP_LIB (A/B runs): entry points added since are simply absent
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _i
        L.vaw_version.restype = _i
        L.vaw_last_error_string.restype = C.c_char_p
        L.vaw_colsum_workspace_floats.argtypes = [_l, _l]
        L.vaw_colsum_workspace_floats.restype = _l
        L.vaw_row_bwd_workspace_floats.argtypes = [_i, _i, _i]
        L.vaw_row_bwd_workspace_floats.restype = _l
        L.vaw_conv3x3_wgrad_small_workspace_floats.argtypes = [_i, _i, _i, _i, _i]
        L.vaw_conv3x3_wgrad_small_workspace_floats.restype = _l
        L.vaw_groupnorm_workspace_floats.argtypes = [_i, _i, _i]
        L.vaw_groupnorm_workspace_floats.restype = _l
        L.vaw_wgrad_grouped_desc_bytes.argtypes = [_i]
        L.vaw_wgrad_grouped_desc_bytes.restype = _l
        L.vaw_reduce_rows_batched_desc_bytes.argtypes = [_i]
        L.vaw_reduce_rows_batched_desc_bytes.restype = _l
        L.vaw_fp8_quantize_batched_desc_bytes.argtypes = [_i]
        L.vaw_fp8_quantize_batched_desc_bytes.restype = _l
        L.vaw_fp8_quantize_workspace_floats.argtypes = []
        L.vaw_fp8_quantize_workspace_floats.restype = _l
        L.vaw_sumsq_workspace_floats.argtypes = []
        L.vaw_sumsq_workspace_floats.restype = _l
        L.vaw_debug_force_rowwise_attention.argtypes = [_i]
        L.vaw_debug_force_rowwise_attention.restype = None
        L.vaw_debug_force_generic_gemm.argtypes = [_i]
        L.vaw_debug_force_generic_gemm.restype = None
        L.vaw_debug_gemm_tile.argtypes = [_i]
        L.vaw_debug_gemm_tile.restype = None
        for dbg in ("vaw_debug_gn_coop", "vaw_debug_gn_flat"):      # absent from older measurement builds loaded through VAW_HIP_LIB
            if hasattr(L, dbg):
                getattr(L, dbg).argtypes = [_i]
                getattr(L, dbg).restype = None
        L.vaw_p8_set_reserved_cus.argtypes = [_i]
        L.vaw_p8_set_reserved_cus.restype = None
        L.vaw_debug_cu_hog.argtypes = [_i, _i, _p]
        L.vaw_debug_cu_hog.restype = _i
        _lib = L
    return _lib


def exported_symbols():
    return sorted(list(_PROTOS) + ["vaw_version", "vaw_last_error_string", "vaw_colsum_workspace_floats",
                                   "vaw_sumsq_workspace_floats", "vaw_groupnorm_workspace_floats", "vaw_wgrad_grouped_desc_bytes",
                                   "vaw_conv3x3_wgrad_small_workspace_floats", "vaw_row_bwd_workspace_floats",
                                   "vaw_fp8_quantize_workspace_floats", "vaw_p8_set_reserved_cus", "vaw_reduce_rows_batched_desc_bytes", "vaw_fp8_quantize_batched_desc_bytes"])


def check(rc, what):
    if rc != 0:
        raise VawError(f"{what} failed ({rc}): {lib().vaw_last_error_string().decode()}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VawError("vaw_amd runs on the GPU only: got a CPU tensor (no CPU fallback exists by design)")
