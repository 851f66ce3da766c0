"""ctypes binding of libucdir_hip.so (include/ucdir_hip.h).

There is no CPU or PyTorch fallback: if the shared library is missing or a call fails the
caller gets an exception.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UCDIR_LIB") or os.path.join(_HERE, "libucdir_hip.so")   # UCDIR_LIB: A/B-test another build
MAX_MULTS = 8
ABI_VERSION = 5


class UcdirConfig(Structure):
    _fields_ = [
        ("in_channel", c_int32), ("out_channel", c_int32), ("inner_channel", c_int32),
        ("n_mults", c_int32), ("channel_mults", c_int32 * MAX_MULTS),
        ("n_attn_res", c_int32), ("attn_res", c_int32 * MAX_MULTS),
        ("res_blocks", c_int32), ("image_size", c_int32), ("device", c_int32), ("attn_fp16", c_int32),
    ]


class UcdirError(RuntimeError):
    pass


_SIGS = {
    "ucdir_abi_version": (c_int32, []),
    "ucdir_last_error": (c_char_p, []),
    "ucdir_create": (c_int32, [POINTER(UcdirConfig), POINTER(c_void_p)]),
    "ucdir_destroy": (None, [c_void_p]),
    "ucdir_load_weight": (c_int32, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int32]),
    "ucdir_finalize_weights": (c_int32, [c_void_p]),
    "ucdir_num_weights": (c_int32, [c_void_p]),
    "ucdir_weight_name": (c_char_p, [c_void_p, c_int32]),
    "ucdir_prepare_guide": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ucdir_unet_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ucdir_set_graph": (c_int32, [c_void_p, c_int32]),
    "ucdir_sampler_step_rng": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                         ctypes.c_uint64, ctypes.c_uint32, c_void_p]),
    "ucdir_fill_normal": (c_int32, [c_void_p, c_int64, ctypes.c_uint64, ctypes.c_uint32, c_void_p]),
    "ucdir_sampler_step_rng_batched": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float,
                                                 c_void_p, ctypes.c_uint32, c_void_p]),
    "ucdir_fill_normal_batched": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, ctypes.c_uint32, c_void_p]),
    "ucdir_gather_windows": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "ucdir_sampler_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                     c_float, c_void_p]),
    "ucdir_debug_read": (c_int32, [c_void_p, c_char_p, c_char_p, c_void_p, c_int64, c_void_p]),
    "ucdir_workspace_bytes": (c_int64, [c_void_p]),
    "ucdir_debug_flag": (c_int32, [c_char_p, c_int32]),
    "ucdir_debug_launch_plan": (c_int32, [c_char_p, c_int32, c_int32, c_int32, c_double]),
    "ucdir_profile_enable": (c_int32, [c_int32]),
    "ucdir_profile_read": (c_int32, [c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_double), POINTER(c_double),
                                     POINTER(c_double), POINTER(c_int32), c_void_p]),
    "ucdir_forward_flops": (c_double, [c_void_p]),
    "ucdir_matrix_rate": (c_int32, [c_int32, c_int32, POINTER(c_double), c_void_p]),
    "ucdir_predictor_create": (c_int32, [c_int32, POINTER(c_void_p)]),
    "ucdir_predictor_destroy": (None, [c_void_p]),
    "ucdir_predictor_load_weight": (c_int32, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int32]),
    "ucdir_predictor_finalize": (c_int32, [c_void_p]),
    "ucdir_predictor_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ucdir_op_conv": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "ucdir_op_conv_res": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ucdir_op_akgm": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ucdir_op_attention": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
}
EXPORTED = tuple(_SIGS)

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UcdirError(
            f"{LIB_PATH} is missing: build it with `python -m ucdir_amd.build` (hipcc, gfx950). "
            "ucdir_amd has no CPU / PyTorch fallback for the denoiser.")
    # The library shares device memory and streams with PyTorch, so both must sit on ONE HIP runtime: import torch first - its
    # bundled libamdhip64 is then the one this library's dependency resolves to.  (Loaded the other way round the process holds
    # two runtimes and the second one to initialise sees no device: `python __graft_entry__.py smoke` failed with "no HIP
    # device available" behind build(), which loads the library before anything imported torch.)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.ucdir_abi_version() != ABI_VERSION:
        raise UcdirError("libucdir_hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ucdir_last_error()
        raise UcdirError(msg.decode() if msg else f"libucdir_hip call failed (rc={rc})")
