"""ctypes binding of libvcengine.so (include/vcengine.h).

This is the binding a maintainer of the reference would add (INTEGRATION.md shows it verbatim).
There is deliberately no fallback: if the shared library is missing or fails to load, importing the
compute path raises, so a run can never silently proceed on a non-HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VC_ENGINE_LIB", os.path.join(_HERE, "libvcengine.so"))   # override: A/B of two builds

VC_OK = 0
VC_E_INVALID, VC_E_HIP, VC_E_STATE, VC_E_NOMEM, VC_E_UNSUPPORTED = -1, -2, -3, -4, -5
VC_FWD_RUN_MAIN_BLOCKS, VC_FWD_STORE_RESIDUAL, VC_FWD_USE_RESIDUAL, VC_FWD_SHARED_CFG_INPUT = 1, 2, 4, 8
VC_FWD_RESIDUAL_UNCOND = 16
VC_MAX_GEOADA_LAYERS = 64
VC_ABI_VERSION = 3
VC_RCCL_UNIQUE_ID_BYTES = 128
VC_SP_FORCE_EXCHANGE = 1


class vc_config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("ffn_dim", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32),
                ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("geoada_in_dim", C.c_int32),
                ("text_dim", C.c_int32), ("text_len", C.c_int32), ("freq_dim", C.c_int32),
                ("eps", C.c_float), ("num_geoada_layers", C.c_int32),
                ("geoada_layers", C.c_int32 * VC_MAX_GEOADA_LAYERS)]


class vc_vae_config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("z_dim", C.c_int32), ("dim_mult", C.c_int32 * 4), ("num_res_blocks", C.c_int32),
                ("temporal_downsample", C.c_int32 * 3)]


class vc_t5_config(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("dim", C.c_int32), ("dim_attn", C.c_int32), ("dim_ffn", C.c_int32),
                ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("num_buckets", C.c_int32),
                ("max_distance", C.c_int32), ("eps", C.c_float)]


ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALL_TO_ALL_SUB_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p)

# every symbol include/vcengine.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "vc_abi_version": (_I, []),
    "vc_last_error": (C.c_char_p, [_P]),
    "vc_create": (_I, [C.POINTER(vc_config), C.POINTER(_P)]),
    "vc_destroy": (None, [_P]),
    "vc_load_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L)]),
    "vc_missing_weights": (_I, [_P]),
    "vc_set_rope_table": (_I, [_P, C.POINTER(C.c_double), _I, _I]),
    "vc_sp_init": (_I, [_P, _I, _I, ALL_TO_ALL_FN, ALL_GATHER_FN, _P]),
    "vc_sp_set_ring": (_I, [_P, _I, ALL_TO_ALL_SUB_FN, SENDRECV_FN]),
    "vc_sp_ring_degree": (_I, [_P]),
    "vc_rccl_available": (_I, []),
    "vc_rccl_unique_id": (_I, [_P, _I]),
    "vc_sp_init_rccl": (_I, [_P, _I, _I, _P, _I, C.c_uint32]),
    "vc_sp_comm_ranks": (_I, [_P]),
    "vc_sp_all_to_all": (_I, [_P, _I, _P, _P, _L, _P]),
    "vc_sp_all_to_all_n": (_I, [_P, _I, _P, _P, _L, _I, _P]),
    "vc_sp_all_gather": (_I, [_P, _P, _P, _L, _P]),
    "vc_sp_all_to_all_sub": (_I, [_P, _I, _P, _P, _L, _I, _I, _I, _P]),
    "vc_sp_sendrecv": (_I, [_P, _I, _P, _I, _P, _I, _L, _P]),
    "vc_sp_init_sim": (_I, [_P, _I, _I, C.c_double]),
    "vc_prepare_video": (_I, [_P, _P, C.POINTER(_P), C.POINTER(C.c_int32), _I, _I, _I, _I, _I, _P]),
    "vc_forward": (_I, [_P, _P, _P, _P, _F, C.c_uint32, _P]),
    "vc_time_embedding": (_I, [_P, _P, _I, _P, _P]),
    "vc_workspace_bytes": (_L, [_P]),
    "vc_reset_residuals": (_I, [_P]),
    "vc_graph_replays": (_L, [_P]),
    "vc_profile_enable": (_I, [_P, _I]),
    "vc_profile_read": (_I, [_P, _I, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double)]),
    "vc_op_gemm_bf16": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P, _L, _F, _I, _P]),
    "vc_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L),
                             C.POINTER(_L), _I, _F, _P]),
    "vc_op_attention_variant": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L),
                                     C.POINTER(_L), _I, _F, _I, _P]),
    "vc_op_attention_fp8_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "vc_op_attention_fp8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), _I, _F, _I, _I,
                                 _P, _L, _P]),
    "vc_op_attention_lse": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), _I, _F, _P]),
    "vc_op_attention_merge": (_I, [C.POINTER(_P), C.POINTER(_P), _I, _P, _I, _I, _I, C.POINTER(_L), _P]),
    "vc_op_attention_segmented": (_I, [_P, _P, _P, _P, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L),
                                       C.POINTER(_L), _I, _I, _F, _P]),
    "vc_op_attention_padmerge": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L),
                                      C.POINTER(_L), C.POINTER(C.c_int32), _F, _P]),
    "vc_render_last_error": (C.c_char_p, []),
    "vc_op_render_composite": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "vc_op_render_depth_gray": (_I, [_P, _P, _L, _I, _F, _F, _P]),
    "vc_op_render_gauss_density": (_I, [_P, _I, _P, _I, _I, _P]),
    "vc_op_render_gauss_frame": (_I, [_P, _I, _P, _F, _F, _P, _P, _I, _I, _P]),
    "vc_op_render_blend": (_I, [_P, _P, _P, _P, _L, _I, _P]),
    "vc_set_fp8_linear": (_I, [_P, _I]),
    "vc_fp8_linear": (_I, [_P]),
    "vc_set_fp8_attention": (_I, [_P, _I, _I]),
    "vc_fp8_attention": (_I, [_P]),
    "vc_op_quantize_rows_fp8": (_I, [_P, _L, _P, _L, _P, _I, _I, _P]),
    "vc_op_gemm_fp8": (_I, [_P, _L, _P, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P, _L, _I, _I, _P]),
    "vc_h264_pcm_last_error": (C.c_char_p, []),
    "vc_op_h264_pcm_bytes": (_L, [_I, _I, _I]),
    "vc_op_h264_pcm_pack": (_I, [_P, _P, _I, _I, _I, _P]),
    "vc_op_h264_pcm_unpack": (_I, [_P, _P, _I, _I, _I, _P]),
    "vc_fit_last_error": (C.c_char_p, []),
    "vc_op_fit_erode_mask": (_I, [_P, _P, _I, _I, _I, _P]),
    "vc_op_fit_points_scratch_bytes": (_L, [_I, _I]),
    "vc_op_fit_points": (_I, [_P, _P, C.POINTER(_F), C.POINTER(_F), _I, _I, _P, _P, _P, _P]),
    "vc_op_fit_moments_scratch_bytes": (_L, []),
    "vc_op_fit_moments": (_I, [_P, _L, _P, _P, _P]),
    "vc_op_fit_project": (_I, [C.POINTER(_F), _P, _P, _P, _I, _I, _P]),
    "vc_op_fit_blend": (_I, [_P, _P, _P, _F, C.POINTER(_F), _P, _P, _L, _P]),
    "vc_op_fit_picture_u8": (_I, [_P, _P, _L, _P]),
    "vc_op_render_points_scratch_bytes": (_L, [_L, _I, _I, _I]),
    "vc_op_render_points": (_I, [_P, _P, _L, C.POINTER(_F), C.POINTER(_F), _I, _I, _F, _I, _F, _P, _P, _P, _P, _P]),
    "vc_op_render_mesh_scratch_bytes": (_L, [_I, _I, _I]),
    "vc_op_render_mesh": (_I, [_P, _P, _I, _P, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _I, _I, _I, _P, _P, _P, _P,
                               _P]),
    "vc_op_layernorm": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _P, _L, _P]),
    "vc_op_rmsnorm_rope": (_I, [_P, _L, _I, _I, _P, _F, _P, C.POINTER(C.c_int32), _P]),
    "vc_op_qkv_front": (_I, [_P, _I, _I, _P, _P, _F, _P, C.POINTER(C.c_int32), _P, _I, _P]),
    "vc_op_geoada_context": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vc_op_unipc_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, C.POINTER(_F), _I, _P]),
    "vc_vae_create": (_I, [C.POINTER(vc_vae_config), C.POINTER(_P)]),
    "vc_vae_load_weight": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_L)]),
    "vc_vae_missing_weights": (_I, [_P]),
    "vc_vae_encode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vc_vae_decode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "vc_vae_last_error": (C.c_char_p, [_P]),
    "vc_vae_workspace_bytes": (_L, [_P]),
    "vc_vae_set_time_chunk": (_I, [_P, _I]),
    "vc_vae_last_time_chunk": (_I, [_P]),
    "vc_vae_release_workspace": (_I, [_P]),
    "vc_vae_destroy": (None, [_P]),
    "vc_t5_create": (_I, [C.POINTER(vc_t5_config), C.POINTER(_P)]),
    "vc_t5_load_weight": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_L)]),
    "vc_t5_encode": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "vc_t5_relative_bucket": (_I, [_I, _I, _I]),
    "vc_t5_last_error": (C.c_char_p, [_P]),
    "vc_t5_workspace_bytes": (_L, [_P]),
    "vc_t5_destroy": (None, [_P]),
}

_lib = None


def load():
    """Load libvcengine.so (once).  Raises RuntimeError if it is not built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C versecrafter_amd/csrc`).  versecrafter_amd has no non-HIP fallback.")
    # torch must own the HIP runtime first so that both share one libamdhip64 (same soname) instance
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.vc_abi_version() != VC_ABI_VERSION:
        raise RuntimeError("libvcengine ABI version mismatch")
    _lib = lib
    return lib


class VcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvcengine error {code}: {msg}")
        self.code = code


def check(code, handle=None):
    if code == VC_OK:
        return
    msg = load().vc_last_error(handle)
    msg = msg.decode() if msg else ""
    if code == VC_E_INVALID:
        raise ValueError(f"libvcengine: {msg}")
    raise VcError(code, msg)


def i64x3(a, b, c):
    return (C.c_int64 * 3)(a, b, c)
