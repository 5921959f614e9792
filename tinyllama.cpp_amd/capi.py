"""ctypes binding of the C-ABI in include/tlhip.h (csrc/libtlhip.so).

There is NO fallback: if the library is missing or a call fails, an exception is raised (TlError carries
tl_last_error()). Nothing here touches the CPU oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libtlhip.so")

F32, Q8_0, Q4_K, Q6_K, Q8_K = 0, 8, 12, 14, 15
(T_ATTN_Q, T_ATTN_K, T_ATTN_V, T_ATTN_OUTPUT, T_FFN_GATE, T_FFN_UP, T_FFN_DOWN, T_ATTN_NORM, T_FFN_NORM,
 T_TOKEN_EMBD, T_OUTPUT_NORM, T_OUTPUT) = range(12)


class TlError(RuntimeError):
    pass


class ModelConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_attention_heads", C.c_int),
                ("num_key_value_heads", C.c_int), ("num_hidden_layers", C.c_int), ("vocab_size", C.c_int),
                ("max_position_embeddings", C.c_int), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
                ("tp_rank", C.c_int), ("tp_size", C.c_int)]


_vp, _fp, _ip, _sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t  # device pointers travel as integers

# name -> argtypes; every symbol include/tlhip.h declares (tests/test_capi_symbols.py checks the two agree)
SIGNATURES = {
    "tl_init": [C.c_int],
    "tl_device_count": [C.POINTER(C.c_int)],
    "tl_malloc": [C.POINTER(C.c_void_p), _sz],
    "tl_shutdown": [],
    "tl_free": [_vp],
    "tl_memset": [_vp, C.c_int, _sz, _vp],
    "tl_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "tl_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "tl_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "tl_stream_create": [C.POINTER(C.c_void_p)],
    "tl_stream_destroy": [_vp],
    "tl_stream_sync": [_vp],
    "tl_device_sync": [],
    "tl_qweight_upload": [C.c_int, _vp, C.c_int, C.c_int, C.POINTER(C.c_void_p), _vp],
    "tl_qweight_free": [_vp],
    "tl_qweight_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz)],
    "tl_qweight_expand": [_vp, C.POINTER(_sz), _vp],
    "tl_dequantize_rows": [_vp, C.c_int, C.c_int, _fp, _vp],
    "tl_matvec_q": [_vp, _fp, _fp, _vp],
    "tl_matmul_q": [_vp, _fp, _fp, C.c_int, _vp],
    "tl_rmsnorm_vector": [_fp, _fp, _fp, C.c_int, C.c_float, _vp],
    "tl_rmsnorm_batch": [_fp, _fp, _fp, C.c_int, C.c_int, C.c_float, _vp],
    "tl_rope_table_upload": [_fp, C.c_int, C.c_int, C.c_float, _vp],
    "tl_rope": [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _vp],
    "tl_rope_batch": [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "tl_update_kv_cache": [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "tl_update_kv_cache_batch": [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "tl_attention": [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _vp],
    "tl_attention_noalloc": [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _vp],
    "tl_attention_batch_prefill": [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, _vp],
    "tl_add_residual": [_fp, _fp, _fp, _sz, _vp],
    "tl_swiglu": [_fp, _fp, _fp, _sz, C.c_int, _vp],
    "tl_lookup_embedding": [_vp, C.c_int, _fp, _vp],
    "tl_argmax": [_fp, C.c_int, _ip, _vp],
    "tl_model_create": [C.POINTER(ModelConfig), C.POINTER(C.c_void_p)],
    "tl_model_free": [_vp],
    "tl_model_set_tensor": [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int],
    "tl_model_finalize": [_vp],
    "tl_model_reset_kv": [_vp],
    "tl_model_decode": [_vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)],
    "tl_model_generate_greedy": [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp],
    "tl_model_prefill": [_vp, _vp, C.c_int, C.c_int, _vp, _vp],
    "tl_quantize_q8_k": [_vp, _sz, _vp, _vp],
    "tl_model_set_prefill": [_vp, C.c_int, C.c_int],
    "tl_model_generate_sampled": [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp, C.c_int, C.c_float, C.c_int, C.c_float,
                                  C.c_uint64, _vp],
    "tl_sample_top_k_top_p": [_vp, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp],
    "tl_gemm_set_terms": [C.c_int],
    "tl_model_device_ptr": [_vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "tl_model_bytes_per_token": [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "tl_model_prefill_planes_bytes": [_vp, C.POINTER(_sz)],
    "tl_tp_unique_id": [_vp],
    "tl_tp_local_group_prefill": [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp],
    "tl_model_graph_captured": [_vp, _vp],
    "tl_model_batch_init": [_vp, C.c_int],
    "tl_model_batch_prefill": [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp],
    "tl_model_batch_decode": [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp],
    "tl_model_batch_prefill_packed": [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
    "tl_model_batch_generate_greedy": [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp],
    "tl_model_tp_init": [_vp, _vp],
    "tl_model_tp_peer_export": [_vp, _vp],
    "tl_model_tp_peer_import": [_vp, _vp, _vp],
    "tl_model_tp_peer_enable": [_vp, C.c_int],
    "tl_model_tp_peer_enabled": [_vp, _vp],
    "tl_model_tp_peer_fused": [_vp, _vp],
    "tl_model_graph_nodes": [_vp, _vp],
    "tl_tp_local_group_decode": [_vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)],
}

# include/tlhip_debug.h: measurement / diagnostic hooks (not part of the boundary)
DEBUG_SIGNATURES = {
    "tl_bench_matvec": [_vp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.POINTER(C.c_float), _vp],
    "tl_bench_matvec_prefetch": [_vp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.POINTER(C.c_float)],
    "tl_debug_gemv_stamps": [_vp, _fp, _fp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp],
    "tl_model_profile_decode": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "tl_model_bench_kernel": [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)],
    "tl_chain_set_stamps": [_vp, _sz],
    "tl_debug_gemm_px_launches": [C.POINTER(C.c_longlong)],
}

# libtlhip_tools.so (csrc/tl_stream.hip): experiments kept out of the product library; declared in include/tlhip_debug.h
TOOLS_LIB_PATH = os.path.join(HERE, "csrc", "libtlhip_tools.so")
TOOLS_SIGNATURES = {
    "tl_stream_matvec": [_vp, C.c_int, _fp, _fp, C.c_int, C.POINTER(C.c_float), _vp],
}

_lib = None
_tools_lib = None


def lib():
    """The loaded library with typed entry points; raises if csrc/libtlhip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TlError(f"{LIB_PATH} is missing: build it with `make -C tinyllama.cpp_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.tl_last_error.restype = C.c_char_p
        L.tl_last_error.argtypes = []
        for name, argtypes in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = argtypes
        _lib = L
    return _lib


def tools_lib():
    """libtlhip_tools.so with typed entry points (it links against libtlhip.so, which is loaded first)."""
    global _tools_lib
    if _tools_lib is None:
        lib()
        if not os.path.exists(TOOLS_LIB_PATH):
            raise TlError(f"{TOOLS_LIB_PATH} is missing: build it with `make -C tinyllama.cpp_amd/csrc`")
        L = C.CDLL(TOOLS_LIB_PATH)
        for name, argtypes in TOOLS_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = argtypes
        _tools_lib = L
    return _tools_lib


def check(rc, what=""):
    if rc != 0:
        raise TlError(f"{what} failed (rc={rc}): {lib().tl_last_error().decode(errors='replace')}")
