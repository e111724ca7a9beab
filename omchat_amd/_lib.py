"""ctypes binding of libomchat_hip.so (include/omchat_hip.h).  The product path has NO CPU fallback: if the library
is missing or cannot be loaded, everything that needs it raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMCHAT_LIB") or os.path.join(HERE, "lib", "libomchat_hip.so")      # OMCHAT_LIB: A/B another build of the same ABI
_lib = None

F16, BF16, F32 = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_LS_RESID, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3, 4
EPI_LS_RESID_STATS, EPI_NONE_STATS = 7, 8      # round 6: the same epilogues + per-row sum-of-squares slots (kernels.h)
PAD_ROW = -(2 ** 31)
PROF_DECODE_GATEUP, PROF_PREFILL_GATEUP, PROF_VIT_FC1 = 0, 1, 2


class OmchatConfig(C.Structure):
    _fields_ = [
        ("v_hidden", C.c_int), ("v_heads", C.c_int), ("v_qk_channels", C.c_int), ("v_mlp", C.c_int), ("v_layers", C.c_int),
        ("v_patch", C.c_int), ("v_image", C.c_int), ("v_eps", C.c_float),
        ("t_hidden", C.c_int), ("t_layers", C.c_int), ("t_heads", C.c_int), ("t_kv_heads", C.c_int), ("t_mlp", C.c_int),
        ("t_vocab", C.c_int), ("t_vocab_total", C.c_int), ("t_eps", C.c_float), ("rope_theta", C.c_float),
        ("max_seq", C.c_int), ("max_batch", C.c_int), ("max_tiles", C.c_int), ("max_prefill_rows", C.c_int), ("dtype", C.c_int),
        ("v_head_dim", C.c_int), ("v_norm_type", C.c_int), ("v_no_qk_norm", C.c_int),
    ]


class OmchatError(RuntimeError):
    pass


_vp, _i, _f, _i64, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64, C.c_size_t
_SIGS = {
    "omchat_last_error": (C.c_char_p, []),
    "omchat_version": (C.c_char_p, []),
    "omchat_ctx_create": (_i, [C.POINTER(OmchatConfig), _i, _i, _vp, C.POINTER(_vp)]),
    "omchat_ctx_destroy": (None, [_vp]),
    "omchat_load_tensor": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i, _i]),
    "omchat_fill_synthetic": (_i, [_vp, _u64]),
    "omchat_weights_missing": (_i, [_vp]),
    "omchat_device_bytes": (_sz, [_vp]),
    "omchat_vit_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "omchat_projector_forward": (_i, [_vp, _vp, _i, _vp, _vp]),
    "omchat_encode_images": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "omchat_splice_plan": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, C.POINTER(_i), _i]),
    "omchat_splice_gather": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "omchat_prefill": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "omchat_prefill_left": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "omchat_decode_step": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "omchat_lm_head": (_i, [_vp, _vp, _i, _vp, _vp]),
    "omchat_greedy": (_i, [_vp, _vp, _i, _vp, _vp]),
    "omchat_kv_lengths": (_i, [_vp, _vp, _i]),
    "omchat_kv_rewind": (_i, [_vp, _i, _i, _vp]),
    "omchat_decode_step_masked": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "omchat_masked_decode_begin": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "omchat_decode_step_masked_next": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "omchat_fused_status": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_uint)]),
    "omchat_has_experiments": (_i, []),
    "omchat_set_allreduce_hook": (_i, [_vp, _vp, _vp]),
    "omchat_allreduce_noop": (_i, [_vp, _vp, C.c_size_t, _i, _vp]),
    "omchat_op_gemv_norm": (_i, [_i, _vp, _vp, _i, _vp, _i, _i, _vp, _f, _vp, _i, _i, _vp]),
    "omchat_prof_enable": (_i, [_vp, _i]),
    "omchat_prof_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long), _i]),
    "omchat_mha_fwd": (_i, [_vp, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "omchat_mha_fwd_varlen": (_i, [_vp, _i, _i, _i, _i, _vp, _f, _i, _vp, _i, _vp]),
    "omchat_op_gemm": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "omchat_op_gemm_fused": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _vp, _vp]),
    "omchat_op_stats_finish": (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "omchat_op_row_sumsq": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "omchat_op_fold_cols": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp]),
    "omchat_op_vit_knorm_slots": (_i, [_i, _vp, _i, _vp, _i, _i, _i, _f, _vp, _i, _i, _vp, _vp]),
    "omchat_op_mha_qnorm": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _f, _f, _vp, _vp]),
    "omchat_ctx_sp_stats": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "omchat_op_set_tuning": (_i, [_i, _i]),
    "omchat_gemm_tune_load": (_i, [C.c_char_p]),
    "omchat_gemm_tune_dump": (_i, [C.c_char_p]),
    "omchat_gemm_tune_runs": (C.c_long, []),
    "omchat_op_gemm_sk_ws": (_sz, []),
    "omchat_op_gemm_sk": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _i, _vp]),
    "omchat_op_gemv": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "omchat_op_gemv_packed": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "omchat_op_pack_x": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "omchat_op_gemm_fp8": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "omchat_op_quant_rows_fp8": (_i, [_i, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]),
    "omchat_enable_fp8_decode": (_i, [_vp, _i]),
    "omchat_enable_fp8_kv": (_i, [_vp, _i]),
    "omchat_enable_fp8_prefill": (_i, [_vp, _i]),
    "omchat_enable_decode_graph": (_i, [_vp, _i]),
    "omchat_decode_graph_stats": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "omchat_op_quant_fp8": (_i, [_i, _vp, _i, _i, _vp, _vp, _vp]),
    "omchat_op_gemv_fp8": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "omchat_op_rmsnorm": (_i, [_i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "omchat_op_resid_rmsnorm": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "omchat_op_vit_qknorm": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "omchat_op_attn_prefill": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _f, _vp]),
    "omchat_op_attn_prefill_d": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _f, _vp]),
    "omchat_op_layernorm": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "omchat_op_attn_decode_ws": (_sz, [_i, _i, _i]),
    "omchat_op_attn_decode": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _f, _vp, _sz, _vp]),
    "omchat_op_attn_decode_kv8": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _f, _vp, _sz, _vp]),
    "omchat_op_attn_decode_kv8_append": (_i, [_i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _sz, _vp]),
    "omchat_op_rope_kv": (_i, [_i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "omchat_op_rope_kv_q8": (_i, [_i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "omchat_op_argmax": (_i, [_vp, _i, _i, _vp, _vp]),
    "omchat_op_fill_uniform": (_i, [_i, _vp, _i64, _u64, _f, _f, _vp]),
    "omchat_preproc_plan": (_i, [_i, _i, _vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "omchat_preproc_anyres": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "omchat_preproc_dynamic": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "omchat_resample_coeffs": (_i, [_i, _i, C.POINTER(_i), _vp, _vp, _i]),
    "omchat_normalize_lut": (_i, [_vp, _vp, _vp]),
    "omchat_comm_unique_id": (_i, [C.c_char_p]),
    "omchat_comm_init": (_i, [C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "omchat_comm_allreduce": (_i, [_vp, _vp, _sz, _i, _vp]),
    "omchat_comm_destroy": (None, [_vp]),
    "omchat_comm_count": (_i, [_vp, C.POINTER(_i)]),
    "omchat_peer_create": (_i, [_i, _i, _sz, C.POINTER(_vp), C.c_char_p]),
    "omchat_peer_connect": (_i, [_vp, C.c_char_p]),
    "omchat_peer_base": (_vp, [_vp]),
    "omchat_peer_connect_local": (_i, [_vp, C.POINTER(_vp)]),
    "omchat_peer_set_mode": (_i, [_vp, _i, _sz, _i]),
    "omchat_peer_capacity": (_sz, [_vp]),
    "omchat_peer_allreduce": (_i, [_vp, _vp, _sz, _i, _vp]),
    "omchat_peer_reduce_scatter": (_i, [_vp, _vp, _sz, _i, _vp]),
    "omchat_peer_all_gather": (_i, [_vp, _vp, _sz, _i, _vp]),
    "omchat_peer_resid_rmsnorm": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "omchat_peer_error": (_i, [_vp, C.POINTER(_i)]),
    "omchat_peer_destroy": (None, [_vp]),
    "omchat_ctx_set_peer": (_i, [_vp, _vp, _sz, _i]),
    "omchat_ctx_allreduce": (_i, [_vp, _vp, _sz, _i, _vp]),
    "omchat_ctx_comm_stats": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
}
EXPORTS = sorted(_SIGS)


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


def lib():
    """Load (once) and return the shared library; raises if it is absent -- build it with `python -m omchat_amd.build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OmchatError(f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
                              "Run `python -m omchat_amd.build` (hipcc --offload-arch=gfx950).")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        # measured GEMM tile choices for the OmChat-13B shapes (TP 1 / 2 / 4 / 8) on MI355X: no first-use tuning on known shapes
        tune = os.environ.get("OMCHAT_GEMM_TUNE_FILE") or os.path.join(HERE, "gemm_tune_gfx950.txt")
        if os.path.exists(tune):
            l.omchat_gemm_tune_load(tune.encode())
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().omchat_last_error().decode()
        if rc == 1:
            raise ValueError(msg)      # argument / shape errors: the reference raises ValueError at these seams
        if rc == 4:
            raise IndexError(msg)      # token id outside the embedding table (torch's embed_tokens raises IndexError)
        raise OmchatError(msg)


def dtype_code(torch_dtype):
    import torch
    return {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}[torch_dtype]


def torch_dtype(code):
    import torch
    return {F16: torch.float16, BF16: torch.bfloat16, F32: torch.float32}[code]


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
