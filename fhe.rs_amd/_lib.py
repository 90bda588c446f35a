"""ctypes loader for the C-ABI shared library (include/fhe_hip.h).

The product path loads exactly one thing: the HIP build `libfhe_hip.so` that lives next to
this file (built in-tree by `__graft_entry__.build()`).  If it is missing this module raises
-- there is no CPU fallback and nothing here ever touches `oracle/`.

`_load_for_tests(path)` exists for the CPU test-suite only: it lets tests/ point the same
bindings at `tests/emu/_build/libfhe_emu.so` (the kernel sources compiled for the host with a
fiber emulator) to validate kernel logic in a container without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfhe_hip.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
szp = C.POINTER(C.c_size_t)
vp = C.c_void_p
sz = C.c_size_t
u64 = C.c_uint64
i32 = C.c_int

# fhe_ntt_tables_fn: int (*)(void *user, uint64_t modulus, size_t degree, 4 x uint64_t *[degree], 2 x uint64_t *)
NTT_TABLES_FN = C.CFUNCTYPE(C.c_int, vp, u64, sz, u64p, u64p, u64p, u64p, u64p, u64p)

# name -> (restype, argtypes); mirrors include/fhe_hip.h one to one
SIGNATURES = {
    "fhe_last_error": (C.c_char_p, []),
    "fhe_version": (C.c_char_p, []),
    "fhe_device_count": (i32, []),
    "fhe_buf_alloc": (i32, [i32, sz, C.POINTER(vp)]),
    "fhe_buf_free": (i32, [vp]),
    "fhe_buf_alloc_async": (i32, [i32, sz, vp, C.POINTER(vp)]),
    "fhe_buf_free_async": (i32, [vp, vp]),
    "fhe_buf_upload": (i32, [vp, vp, sz, vp]),
    "fhe_buf_upload_async": (i32, [vp, vp, sz, vp]),
    "fhe_buf_download": (i32, [vp, vp, sz, vp]),
    "fhe_buf_download_async": (i32, [vp, vp, sz, vp]),
    "fhe_buf_copy_async": (i32, [vp, vp, sz, vp]),
    "fhe_buf_zero_async": (i32, [vp, sz, vp]),
    "fhe_host_alloc": (i32, [sz, C.POINTER(vp)]),
    "fhe_host_free": (i32, [vp]),
    "fhe_stream_create": (i32, [i32, C.POINTER(vp)]),
    "fhe_stream_sync": (i32, [vp]),
    "fhe_stream_destroy": (i32, [vp]),
    "fhe_device_sync": (i32, [i32]),
    "fhe_device_mem_info": (i32, [i32, szp, szp]),
    "fhe_poly_switch_down_to": (i32, [vp, vp, u64p, u64p, sz]),
    "fhe_poly_switch_down_to_dev": (i32, [vp, vp, vp, vp, sz, vp]),
    "fhe_bfv_switch_to_level": (i32, [vp, sz, sz, u64p, u64p, sz]),
    "fhe_bfv_switch_to_level_dev": (i32, [vp, sz, sz, vp, vp, sz, vp]),
    "fhe_ubench_int": (i32, [i32, i32, C.c_double, C.POINTER(C.c_double)]),
    "fhe_ubench_scaler": (i32, [vp, C.c_double, C.POINTER(C.c_double)]),
    "fhe_ubench_copy": (i32, [i32, sz, C.c_double, C.POINTER(C.c_double)]),
    "fhe_ctx_create": (i32, [i32, sz, sz, u64p, u64p, u64p, u64p, u64p, u64p, u64p, C.POINTER(vp)]),
    "fhe_ctx_destroy": (None, [vp]),
    "fhe_ctx_at_level": (i32, [vp, sz, C.POINTER(vp)]),
    "fhe_ctx_niterations_to": (i32, [vp, vp, szp]),
    "fhe_ctx_degree": (sz, [vp]),
    "fhe_ctx_nmoduli": (sz, [vp]),
    "fhe_ctx_device": (i32, [vp]),
    "fhe_ctx_moduli": (i32, [vp, u64p]),
    "fhe_ctx_get_table": (i32, [vp, i32, u64p]),
    "fhe_ntt_forward": (i32, [vp, u64p, sz]),
    "fhe_ntt_backward": (i32, [vp, u64p, sz]),
    "fhe_ntt_forward_dev": (i32, [vp, vp, sz, vp]),
    "fhe_ntt_backward_dev": (i32, [vp, vp, sz, vp]),
    "fhe_poly_add": (i32, [vp, u64p, u64p, sz]),
    "fhe_poly_sub": (i32, [vp, u64p, u64p, sz]),
    "fhe_poly_mul": (i32, [vp, u64p, u64p, sz]),
    "fhe_poly_neg": (i32, [vp, u64p, sz]),
    "fhe_poly_mul_shoup": (i32, [vp, u64p, u64p, u64p, sz]),
    "fhe_poly_shoup": (i32, [vp, u64p, u64p, sz]),
    "fhe_poly_add_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_poly_sub_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_poly_mul_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_poly_neg_dev": (i32, [vp, vp, sz, vp]),
    "fhe_poly_mul_shoup_dev": (i32, [vp, vp, vp, vp, sz, vp]),
    "fhe_poly_substitute": (i32, [vp, sz, u64p, u64p, sz, i32]),
    "fhe_poly_substitute_dev": (i32, [vp, sz, vp, vp, sz, i32, vp]),
    "fhe_poly_serialized_size": (sz, [vp]),
    "fhe_poly_serialize": (i32, [vp, u64p, u8p, sz, i32]),
    "fhe_poly_serialize_dev": (i32, [vp, vp, vp, sz, i32, vp]),
    "fhe_poly_deserialize": (i32, [vp, u8p, u64p, sz, i32]),
    "fhe_poly_deserialize_dev": (i32, [vp, vp, vp, sz, i32, vp]),
    "fhe_poly_switch_down": (i32, [vp, u64p, u64p, sz]),
    "fhe_poly_switch_down_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_scaler_create": (i32, [vp, vp, u64p, sz, u64p, sz, C.POINTER(vp)]),
    "fhe_scaler_create_from_constants": (i32, [vp, vp, sz, i32, u64p, u64p, u64p, u64p, u64, u64, i32,
                                               u64p, u64p, u8p, u64p, u64p, sz, C.POINTER(vp)]),
    "fhe_switcher_create": (i32, [vp, vp, C.POINTER(vp)]),
    "fhe_scaler_destroy": (None, [vp]),
    "fhe_scaler_number_common_moduli": (sz, [vp]),
    "fhe_scaler_get_constants": (i32, [vp, i32, u64p]),
    "fhe_poly_scale": (i32, [vp, u64p, u64p, sz, i32]),
    "fhe_poly_scale_dev": (i32, [vp, vp, vp, sz, i32, vp]),
    "fhe_ksk_create": (i32, [vp, vp, sz, u64p, u64p, u64p, u64p, sz, C.POINTER(vp)]),
    "fhe_ksk_create_dev": (i32, [vp, vp, sz, vp, vp, sz, vp, C.POINTER(vp)]),
    "fhe_ksk_destroy": (None, [vp]),
    "fhe_ksk_set_mode": (i32, [vp, C.c_int, sz]),
    "fhe_ksk_get_mode": (i32, [vp, C.POINTER(C.c_int), szp]),
    "fhe_key_switch": (i32, [vp, u64p, u64p, u64p, sz]),
    "fhe_key_switch_dev": (i32, [vp, vp, vp, vp, sz, vp]),
    "fhe_bfv_relinearize": (i32, [vp, u64p, u64p, sz]),
    "fhe_bfv_relinearize_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_bfv_galois": (i32, [vp, sz, u64p, u64p, sz]),
    "fhe_bfv_galois_dev": (i32, [vp, sz, vp, vp, sz, vp]),
    "fhe_bfv_switch_down": (i32, [vp, sz, u64p, u64p, sz]),
    "fhe_bfv_switch_down_dev": (i32, [vp, sz, vp, vp, sz, vp]),
    "fhe_bfv_dot_product_scalar": (i32, [vp, sz, sz, u64p, i32, u64p, i32, u64p, sz]),
    "fhe_bfv_dot_product_scalar_dev": (i32, [vp, sz, sz, vp, i32, vp, i32, vp, sz, vp]),
    "fhe_bfv_mul_plain": (i32, [vp, sz, u64p, u64p, i32, u64p, sz]),
    "fhe_bfv_mul_plain_dev": (i32, [vp, sz, vp, vp, i32, vp, sz, vp]),
    "fhe_poly_from_seed": (i32, [vp, u8p, u64p, sz]),
    "fhe_poly_from_seed_dev": (i32, [vp, vp, vp, sz, vp]),
    "fhe_bfv_decrypt": (i32, [vp, u64, u64p, u64p, sz, u64p, sz]),
    "fhe_bfv_decrypt_dev": (i32, [vp, u64, vp, vp, sz, vp, sz, vp]),
    "fhe_bfv_rgsw_mul": (i32, [vp, vp, u64p, u64p, sz]),
    "fhe_bfv_rgsw_mul_dev": (i32, [vp, vp, vp, vp, sz, vp]),
    "fhe_bfv_inner_sum": (i32, [C.POINTER(vp), szp, sz, u64p, u64p, sz]),
    "fhe_bfv_inner_sum_dev": (i32, [C.POINTER(vp), szp, sz, vp, vp, sz, vp]),
    "fhe_bfv_expand": (i32, [C.POINTER(vp), sz, u64p, u64p, sz, sz]),
    "fhe_bfv_expand_dev": (i32, [C.POINTER(vp), sz, vp, vp, sz, sz, vp]),
    "fhe_mul_create": (i32, [vp, vp, vp, vp, i32, C.POINTER(vp)]),
    "fhe_mul_destroy": (None, [vp]),
    "fhe_mul_out_shape": (i32, [vp, szp, szp]),
    "fhe_mul_basis": (i32, [vp, szp, u64p]),
    "fhe_mul_set_chunk": (i32, [vp, sz]),
    "fhe_mul_set_streams": (i32, [vp, sz]),
    "fhe_mul_get_options": (i32, [vp, szp, szp]),
    "fhe_bfv_mul": (i32, [vp, u64p, u64p, u64p, sz]),
    "fhe_bfv_mul_dev": (i32, [vp, vp, vp, vp, sz, vp]),
    "fhe_bfv_tensor": (i32, [vp, sz, sz, u64p, u64p, u64p, sz]),
    "fhe_bfv_tensor_dev": (i32, [vp, sz, sz, vp, vp, vp, sz, vp]),
    "fhe_params_create": (i32, [i32, sz, sz, u64p, u64, C.POINTER(vp)]),
    "fhe_params_create_with_tables": (i32, [i32, sz, sz, u64p, u64, NTT_TABLES_FN, vp, C.POINTER(vp)]),
    "fhe_params_destroy": (None, [vp]),
    "fhe_params_max_level": (sz, [vp]),
    "fhe_params_ctx": (i32, [vp, sz, C.POINTER(vp)]),
    "fhe_params_mul_ctx": (i32, [vp, sz, C.POINTER(vp)]),
    "fhe_params_extender": (i32, [vp, sz, C.POINTER(vp)]),
    "fhe_params_down_scaler": (i32, [vp, sz, C.POINTER(vp)]),
    "fhe_mul_create_default": (i32, [vp, sz, vp, i32, C.POINTER(vp)]),
    "fhe_generate_prime": (u64, [sz, u64, u64]),
    "fhe_supports_opt": (i32, [u64]),
    "fhe_is_prime": (i32, [u64]),
    "fhe_generate_moduli": (i32, [szp, sz, sz, u64p]),
    "fhe_synth_uniform_dev": (i32, [vp, u64, u64, u64, sz, vp, sz, vp]),
    "fhe_workspace_trim": (sz, []),
    "fhe_workspace_set_limit": (i32, [sz, sz]),
    "fhe_workspace_get_limit": (i32, [szp, szp]),
    "fhe_workspace_stats": (i32, [szp, szp, szp, szp, szp]),
    "fhe_workspace_pool_stats": (i32, [i32, szp, szp, szp, szp]),
    "fhe_prof_enable": (None, [i32]),
    "fhe_prof_reset": (None, []),
    "fhe_prof_count": (sz, []),
    "fhe_prof_get": (i32, [sz, C.c_char_p, sz, u64p, C.POINTER(C.c_double)]),
    "fhe_prof_get_symbol": (i32, [sz, C.c_char_p, sz]),
    "fhe_engine_set_f64": (None, [i32]),
    "fhe_engine_get_f64": (i32, []),
}

_lib = None
_loaded_path = None


def _bind(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The loaded HIP library; raises if it has not been built."""
    global _lib, _loaded_path
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        _lib = _bind(C.CDLL(LIB_PATH))
        _loaded_path = LIB_PATH
    return _lib


def loaded_path():
    return _loaded_path


def _load_for_tests(path):
    """TEST HOOK: bind to an explicitly given library (the host-emulation build)."""
    global _lib, _loaded_path
    _lib = _bind(C.CDLL(path))
    _loaded_path = path
    return _lib


class FheError(RuntimeError):
    """Non-zero fhe_status; `.code` is the status (see include/fhe_hip.h), the message is
    fhe_last_error()."""

    def __init__(self, code, message):
        super().__init__(f"fhe_status {code}: {message}")
        self.code = code


def check(status):
    if status != 0:
        raise FheError(status, lib().fhe_last_error().decode(errors="replace"))
