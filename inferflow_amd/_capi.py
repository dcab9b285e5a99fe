"""ctypes binding of include/inferflow_amd.h.  Fails loudly when the HIP
library is missing -- there is no CPU fallback on the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IFA_LIB: another build of the SAME library (tools/sweep_variants.py: one .so per tuning setting); tuning only
_LIB_PATH = os.environ.get("IFA_LIB") or os.path.join(_HERE, "lib", "libinferflow_amd.so")
_lib = None


class IfaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("inferflow_amd error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return _LIB_PATH


def lib():
    """Load libinferflow_amd.so (built by inferflow_amd/build.py / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                "inferflow_amd: %s is missing -- run `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % _LIB_PATH)
        L = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
        _declare(L)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise IfaError(rc, lib().ifa_last_error().decode(errors="replace"))
    return rc


_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float

# name -> (restype, argtypes); mirrors include/inferflow_amd.h one to one
SIGNATURES = {
    "ifa_version": (C.c_char_p, []),
    "ifa_last_error": (C.c_char_p, []),
    "ifa_device_count": (_i, []),
    "ifa_set_device": (_i, [_i]),
    "ifa_malloc": (_i, [C.POINTER(_vp), _sz]),
    "ifa_free": (_i, [_vp]),
    "ifa_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "ifa_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "ifa_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "ifa_memset": (_i, [_vp, _i, _sz, _vp]),
    "ifa_stream_create": (_i, [C.POINTER(_vp)]),
    "ifa_stream_destroy": (_i, [_vp]),
    "ifa_stream_sync": (_i, [_vp]),
    "ifa_block_capacity": (_i, [_i]),
    "ifa_block_bytes": (_i, [_i]),
    "ifa_row_bytes": (_sz, [_i, _sz]),
    "ifa_dtype_from_name": (_i, [C.c_char_p]),
    "ifa_quantize": (_i, [_i, _vp, _sz, _sz, _vp, _vp]),
    "ifa_quantize_f32": (_i, [_i, _vp, _sz, _sz, _vp, _vp]),
    "ifa_dequantize": (_i, [_i, _vp, _sz, _sz, _vp, _vp]),
    "ifa_quantize_act_q8": (_i, [_vp, _sz, _sz, _vp, _vp]),
    "ifa_gemv": (_i, [_i, _vp, _sz, _sz, _i, _vp, _vp, _vp, _vp]),
    "ifa_gemm": (_i, [_i, _vp, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
    "ifa_attention_two_pass_min": (_i, [_i]),
    "ifa_attention_two_pass_min_keys": (_i, [_i]),
    "ifa_gemm_big_tiles": (_i, [_i]),
    "ifa_wait_grid_decision": (_i, [_i, _i, C.c_longlong]),
    "ifa_visible_cus_from_mask": (_i, [C.c_char_p, _i, _i]),
    "ifa_inlaunch_waits_enabled": (_i, []),
    "ifa_gemm_release_stream": (_i, [_vp]),
    "ifa_tiled_row_bytes": (_sz, [_i, _sz]),
    "ifa_repack_weights": (_i, [_i, _vp, _sz, _sz, _vp, _vp]),
    "ifa_gemv_tiled": (_i, [_i, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "ifa_layernorm": (_i, [_i, _vp, _sz, _sz, _vp, _vp, _f, _f, _vp, _vp]),
    "ifa_exact_rmsnorm": (_i, [_vp, _sz, _sz, _vp, _vp, _f, _f, _vp, _vp]),
    "ifa_exact_gemv": (_i, [_i, _vp, _sz, _sz, _i, _vp, _vp, _vp, _vp]),
    "ifa_exact_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "ifa_exact_activation_mul": (_i, [_i, _vp, _vp, _sz, _vp, _vp]),
    "ifa_rope": (_i, [_vp, _i, _i, _i, _i, _f, _i, _f, _vp]),
    "ifa_alibi": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "ifa_softmax": (_i, [_vp, _i, _i, _i, _i, _f, _vp]),
    "ifa_activation": (_i, [_i, _i, _vp, _sz, _sz, _vp, _vp]),
    "ifa_mul": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "ifa_add": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "ifa_scale": (_i, [_vp, _f, _sz, _vp, _vp]),
    "ifa_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp]),
    "ifa_moe_route_topk": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp, _vp]),
    "ifa_kv_store": (_i, [_i, _vp, _sz, _sz, _vp, _sz, _vp]),
    "ifa_argmax": (_i, [_vp, _sz, _vp, _vp]),
    "ifa_argmax_masked": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "ifa_model_create": (_i, [_vp, C.POINTER(_vp)]),
    "ifa_model_destroy": (_i, [_vp]),
    "ifa_model_set_tensor": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _sz]),
    "ifa_model_set_tensor_f16": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _sz]),
    "ifa_model_finalize": (_i, [_vp]),
    "ifa_model_reset": (_i, [_vp]),
    "ifa_model_set_option": (_i, [_vp, C.c_char_p, _i]),
    "ifa_model_set_excluded_tokens": (_i, [_vp, _vp, _i]),
    "ifa_model_perf_stat": (_i, [_vp, _vp, _vp, _i, _vp, _i]),
    "ifa_model_fused_supported": (_i, [_vp, C.c_char_p, _sz]),
    "ifa_model_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ifa_model_decode": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ifa_model_decode_prepare": (_i, [_vp, _i, _i]),
    "ifa_add_by_row_index": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "ifa_model_kv_slots": (_i, [_vp, _i]),
    "ifa_model_select_kv": (_i, [_vp, _i]),
    "ifa_model_decode_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ifa_model_get_buffer": (_i, [_vp, C.c_char_p, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "ifa_model_stream": (_vp, [_vp]),
    "ifa_model_time_kernel": (_i, [_vp, _i, _i, _vp]),
    "ifa_model_set_stream": (_i, [_vp, _vp]),
    "ifa_model_tp_begin": (_i, [_vp, _i, _i]),
    "ifa_model_tp_begin_hidden": (_i, [_vp, _vp, _i]),
    "ifa_model_tp_hidden": (_i, [_vp, _vp]),
    "ifa_model_tp_attn": (_i, [_vp, _i, _vp]),
    "ifa_model_tp_post_attn": (_i, [_vp, _i, _vp]),
    "ifa_model_tp_ffn": (_i, [_vp, _i, _vp]),
    "ifa_model_tp_post_ffn": (_i, [_vp, _i, _vp]),
    "ifa_model_tp_logits": (_i, [_vp, _vp]),
    "ifa_model_tp_set_token": (_i, [_vp, _vp]),
    "ifa_comm_unique_id": (_i, [_vp]),
    "ifa_comm_init_rank": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "ifa_comm_init_all": (_i, [_vp, _i, C.POINTER(_vp)]),
    "ifa_comm_destroy": (_i, [_vp]),
    "ifa_comm_capturable": (_i, [_vp]),
    "ifa_comm_serial": (C.c_ulonglong, [_vp]),
    "ifa_comm_abort": (_i, [_vp]),
    "ifa_comm_oneshot": (_i, [_vp]),
    "ifa_comm_oneshot_export": (_i, [_vp, _vp]),
    "ifa_comm_oneshot_import": (_i, [_vp, _vp]),
    "ifa_comm_set_oneshot": (_i, [_vp, _i]),
    "ifa_comm_status": (_i, [_vp]),
    "ifa_comm_rank": (_i, [_vp]),
    "ifa_comm_size": (_i, [_vp]),
    "ifa_comm_group_start": (_i, []),
    "ifa_comm_group_end": (_i, []),
    "ifa_allreduce_sum_f16": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "ifa_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "ifa_broadcast": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ifa_send": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ifa_recv": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ifa_model_tp_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ifa_model_tp_prefill": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ifa_model_tp_decode_batch": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ifa_model_get_tensor": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz)]),
    "ifa_model_get_expert_tensor": (_i, [_vp, _i, _i, _i, _vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz)]),
}


# include/inferflow_engine.h (the C++ InferenceEngine facade)
_ip = C.POINTER(C.c_int)
ENGINE_SIGNATURES = {
    "ifa_engine_create": (_vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "ifa_engine_destroy": (None, [_vp]),
    "ifa_engine_last_error": (C.c_char_p, []),
    "ifa_engine_add_query": (_i, [_vp, _ip, _i]),
    "ifa_engine_add_query_ex": (_i, [_vp, _ip, _i, _i, _i, _f]),
    "ifa_engine_strategy_id": (_i, [_vp, C.c_char_p]),
    "ifa_sampling_choose": (_i, [_vp, _i, _i, _i, _f, _i, _f, C.c_longlong, _i, _ip, C.POINTER(_f), _ip, C.POINTER(_f), _i]),
    "ifa_sampling_choose_ex": (_i, [_vp, _i, _i, C.POINTER(_f), _f, C.c_longlong, _i, _ip, C.POINTER(_f), _ip, C.POINTER(_f), _i, C.POINTER(_f), _ip, _i]),
    "ifa_sampling_random_doubles": (_i, [C.c_longlong, _i, C.POINTER(C.c_double)]),
    "ifa_engine_query_count": (_i, [_vp]),
    "ifa_engine_remove_query": (_i, [_vp, _i]),
    "ifa_engine_infer": (_i, [_vp, _ip, _ip, _i]),
    "ifa_engine_commit": (_i, [_vp, _ip, _ip, _ip, _i]),
    "ifa_engine_last_logits": (_i, [_vp, _i, _vp, _sz, _ip, _ip]),
    "ifa_engine_perf_stat": (_i, [_vp, _vp, _vp, _i]),
    "ifa_engine_generate": (_i, [_vp, _i, _i, _ip, C.POINTER(_f)]),
    "ifa_engine_perplexity": (_i, [_vp, _ip, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "ifa_perplexity_token_nll": (C.c_double, [_vp, _i, _i]),
    "ifa_engine_model_info": (_i, [_vp, C.c_char_p]),
    "ifa_engine_worker": (_vp, [_vp, _i]),
    "ifa_engine_worker_plan": (_i, [_vp, _i, _ip]),
    "ifa_service_parse_request": (_i, [C.c_char_p, _i, C.c_char_p, _sz]),
    "ifa_service_format_response": (_i, [_ip, _i, _i, _i, _i, _i, C.c_char_p, _sz]),
    "ifa_service_selftest_loop": (_i, [_i, _i, _i, _ip, _i, _i, _i, _i, _i, C.c_char_p, _sz]),
    "ifa_partition_slice": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _sz, _sz, C.POINTER(_sz)]),
    "ifa_partition_split_layers": (_i, [_i, _i, _ip, _i]),
}


class ModelConfig(C.Structure):
    """ifa_model_config (include/inferflow_amd.h)"""
    _fields_ = [(n, C.c_int) for n in (
        "dim", "layers", "heads", "kv_heads", "head_dim", "ffn", "vocab", "max_ctx",
        "norm_kind", "act_kind", "is_glu", "rope_order", "use_alibi", "parallel_attn",
        "share_input")] + [(n, C.c_float) for n in (
            "rope_theta", "partial_rotary", "kq_scale", "eps")] + [(n, C.c_int) for n in (
                "kv_dtype", "full_quant_gemv", "experts", "moe_top_k", "moe_norm_topk",
                "tp_rank", "tp_size", "device")] + [(n, C.c_float) for n in (
                    "attn_norm_base", "ffn_norm_base", "out_norm_base", "attn_out_scale", "ffn_out_scale", "out_scale", "embd_scale")]


def _declare(L):
    missing = []
    for name, (res, args) in list(SIGNATURES.items()) + list(ENGINE_SIGNATURES.items()):
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and os.environ.get("IFA_LIB"):
        return      # a tuning build (tools/sweep_variants.py) may predate the newest entry points; bench.py does not call them
    if missing:
        raise ImportError("libinferflow_amd.so lacks symbols declared in include/*.h: %s" % missing)
