"""ctypes binding of libcfhip.so (the C-ABI declared in include/cfhip.h).

The library is built in-tree by `__graft_entry__.build()` / `build_lib.sh` and must sit next to
this file.  There is NO fallback: if it is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import c_uint64, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# CFHIP_LIB selects another build of the same C-ABI (the benchmark tools use it for the -DCFHIP_ABLATE library)
LIB_PATH = os.environ.get("CFHIP_LIB") or os.path.join(_HERE, "libcfhip.so")

_lib: Optional[ctypes.CDLL] = None


class GemmProblem(ctypes.Structure):
    """`cfhip_gemm_problem` of include/cfhip.h (one weight-gradient GEMM of a grouped launch)."""

    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias_grad", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64),
        ("accumulate", c_int), ("bias_grad_accumulate", c_int),
    ]

# name -> (restype, argtypes); mirrors include/cfhip.h line by line
_P = c_void_p
SIGNATURES = {
    "cfhip_version": (c_int, []),
    "cfhip_last_error": (c_char_p, []),
    "cfhip_set_option": (c_int, [c_char_p, c_int]),
    "cfhip_gemm_bf16": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int,
         c_int, c_int, c_int, _P, c_size_t, _P, c_int, _P],
    ),
    "cfhip_gemm_bf16_grouped_tn": (c_int, [_P, c_int, _P]),
    "cfhip_colsum_workspace": (c_size_t, [c_int, c_int]),
    "cfhip_colsum_bf16": (c_int, [_P, _P, c_int, c_int, c_int64, c_int, _P, c_size_t, _P]),
    "cfhip_layernorm_fwd": (
        c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int64, c_int64, c_float, _P]
    ),
    "cfhip_layernorm_bwd_workspace": (c_size_t, [c_int, c_int]),
    "cfhip_layernorm_bwd": (
        c_int,
        [_P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int64, c_int64, c_int64, c_int, _P,
         c_size_t, _P],
    ),
    "cfhip_layernorm_bwd_partials": (
        c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int64, c_int64, c_int64, _P, c_size_t, _P, _P]
    ),
    "cfhip_layernorm_bwd_reduce": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P]),
    "cfhip_layernorm_bwd2": (
        c_int,
        [_P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int64, c_int64, c_int64, c_int, _P, c_size_t, _P, _P],
    ),
    "cfhip_layernorm4d_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cfhip_layernorm4d_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "cfhip_layernorm4d_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "cfhip_split_f32_bf16x2": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_join_bf16x2_f32": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_attn_fwd": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, _P],
    ),
    "cfhip_attn_bwd": (
        c_int,
        [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_int, _P],
    ),
    "cfhip_attn_fwd_dh": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, _P],
    ),
    "cfhip_conv2d_grouped_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 11 + [_P]),
    "cfhip_conv2d_grouped_bwd_input": (c_int, [_P, _P, _P] + [c_int] * 11 + [_P]),
    "cfhip_conv2d_grouped_bwd_weight": (c_int, [_P, _P, _P, c_int, _P, c_int] + [c_int] * 11 + [_P]),
    "cfhip_attn_probs": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
         c_float, c_int, _P],
    ),
    "cfhip_attn_probs_bwd": (
        c_int,
        [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
         c_int64, c_float, c_int, _P],
    ),
    "cfhip_attn_bwd_dh": (
        c_int,
        [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_int, _P],
    ),
    "cfhip_attn_fwd_dropout": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_float, c_uint64, c_uint64, _P],
    ),
    "cfhip_attn_bwd_dropout": (
        c_int,
        [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
         c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_int, c_float, c_uint64,
         c_uint64, _P],
    ),
    "cfhip_attn_dropout_mask": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_uint64, _P]),
    "cfhip_im2row": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cfhip_assemble_tokens_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cfhip_assemble_tokens_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cfhip_cast_f32_to_bf16": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_cast_bf16_to_f32": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_gelu_fwd": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_gelu_bwd": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_add_bf16": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_transpose_bf16": (c_int, [_P, _P, c_int, c_int, c_int64, c_int64, _P]),
    "cfhip_adam_step": (
        c_int,
        [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_int,
         c_float, _P],
    ),
    "cfhip_adam_step_dev": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, c_int, _P]),
    "cfhip_sumsq_f32": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_softmax_xent": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cfhip_softmax_focal": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, c_float, _P]),
    "cfhip_conv_im2row": (c_int, [_P, c_int, _P] + [c_int] * 10 + [_P]),
    "cfhip_conv_row2im": (c_int, [_P, _P] + [c_int] * 10 + [_P]),
    "cfhip_transpose_batched": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "cfhip_batchnorm_fwd": (
        c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P]
    ),
    "cfhip_batchnorm_bwd": (
        c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]
    ),
    "cfhip_leaky_relu_fwd": (c_int, [_P, _P, c_int64, c_float, _P]),
    "cfhip_leaky_relu_bwd": (c_int, [_P, _P, _P, c_int64, c_float, _P]),
    "cfhip_avgpool_fwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "cfhip_avgpool_bwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "cfhip_groupnorm_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "cfhip_groupnorm_bwd": (c_int, [_P, _P, c_int] + [_P] * 9 + [c_int] * 5 + [_P]),
    "cfhip_groupnorm_affine_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int,
                                           c_int, _P]),
    "cfhip_groupnorm_affine_bwd": (c_int, [_P, _P, c_int] + [_P] * 9 + [c_int] * 6 + [_P]),
    "cfhip_groupnorm_split_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int,
                                          c_int, c_int, _P, _P]),
    "cfhip_groupnorm_split_bwd": (c_int, [_P, _P, c_int] + [_P] * 9 + [c_int] * 7 + [_P, _P]),
    "cfhip_diffusion_loss": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int, _P]),
    "cfhip_silu_f32_fwd": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_silu_f32_bwd": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_time_proj_fwd": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P]),
    "cfhip_time_proj_bwd": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P, _P]),
    "cfhip_upsample2_fwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "cfhip_upsample2_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "cfhip_groupnorm_nhwc_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cfhip_groupnorm_nhwc_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, _P, _P]),
    "cfhip_groupnorm_nhwc_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "cfhip_upsample2_nhwc_fwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P]),
    "cfhip_upsample2_nhwc_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P]),
    "cfhip_reflect_pad2d_fwd": (c_int, [_P, c_int, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cfhip_reflect_pad2d_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cfhip_avgpool2_fwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "cfhip_avgpool2_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "cfhip_timestep_embedding": (c_int, [_P, _P, c_int, c_int, c_float, _P]),
    "cfhip_ema_update": (c_int, [_P, _P, c_int64, c_float, c_float, _P]),
    "cfhip_spin": (c_int, [c_int, _P]),
    "cfhip_sgemm_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int64, c_int64, c_int, c_int, _P, c_float, _P]),
    "cfhip_dot_f32": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_conv3x3_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cfhip_conv3x3_nhwc_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "cfhip_conv3x3_wgrad_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cfhip_conv3x3_wgrad_nhwc_bf16": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                              c_size_t, _P]),
    "cfhip_conv3x3_pack_filters": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "cfhip_conv3x3_pack_filters_grouped": (c_int, [_P, c_int, _P]),
    "cfhip_colreduce_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "cfhip_colreduce2_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cfhip_q_sample": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, _P]),
    "cfhip_mse_loss": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_float, _P]),
    "cfhip_copy_strided_bf16": (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int64, _P]),
    "cfhip_copy_strided2_bf16": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int64, _P]),
    "cfhip_geglu_fwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "cfhip_geglu_bwd": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cfhip_quick_gelu_fwd": (c_int, [_P, _P, c_int64, _P]),
    "cfhip_quick_gelu_bwd": (c_int, [_P, _P, _P, c_int64, _P]),
    "cfhip_embedding_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int64, _P]),
    "cfhip_embedding_bwd": (c_int, [_P, c_int, _P, _P, c_int64, c_int, c_int64, c_int64, _P]),
    "cfhip_l2norm_fwd": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cfhip_l2norm_bwd": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "cfhip_dropout": (c_int, [_P, _P, c_int, c_int64, c_float, c_uint64, c_uint64, _P, _P, _P]),
    "cfhip_drop_path_mask": (c_int, [_P, c_int64, c_float, c_uint64, c_uint64, _P]),
    "cfhip_drop_path": (c_int, [_P, _P, c_int, _P, c_float, c_int64, c_int64, _P]),
    "cfhip_ml_encode_fwd": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int, _P, _P, _P]),
    "cfhip_ml_encode_indices": (c_int, [_P, c_int64, c_int64, _P, _P, c_int, _P, _P]),
    "cfhip_ml_encode_bwd": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, c_int, _P, _P, _P]),
    "cfhip_gemm_kernel_name": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t]),
    "cfhip_comm_unique_id": (c_int, [_P]),
    "cfhip_comm_init": (c_int, [c_int, c_int, _P, _P]),
    "cfhip_comm_destroy": (c_int, [_P]),
    "cfhip_comm_count": (c_int, [_P, _P, _P]),
    "cfhip_comm_allreduce": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "cfhip_comm_allgather": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "cfhip_comm_reduce_scatter": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "cfhip_comm_broadcast": (c_int, [_P, _P, c_size_t, c_int, c_int, _P]),
}


def lib_exists() -> bool:
    return os.path.isfile(LIB_PATH)


# Launch recording (fused.StackPlan, round 4): while RECORDER is a list, every C-ABI call made through `load().<name>(...)`
# is executed AND appended as (function, argument tuple) — raw pointers, sizes, stream handles: exactly what a replay needs.
RECORDER: Optional[list] = None


class _Recording:
    __slots__ = ("lib",)

    def __init__(self, lib: ctypes.CDLL) -> None:
        self.lib = lib

    def __getattr__(self, name: str):
        fn = getattr(self.lib, name)
        rec = RECORDER
        if name.endswith("_workspace") or name in ("cfhip_version", "cfhip_last_error", "cfhip_set_option", "cfhip_gemm_kernel_name"):
            return fn  # host-side queries: nothing to replay

        def call(*args):
            rc = fn(*args)
            if rec is not None:
                rec.append((0, fn, args))
            return rc

        return call


# Vectorcall wrappers (csrc/gen_fastcall.py -> _cfhip_fast.<abi>.so, built next to libcfhip.so): the same entry points of the
# same loaded library, entered through METH_FASTCALL functions that read the integer arguments straight out of the call's
# argument array instead of through ctypes' per-argument converters — ~0.3 us instead of 3-6 us per launch on the one thread
# that issues every launch of a step.  Optional acceleration of the HOST path only (CFHIP_FASTCALL=0 or a missing module:
# plain ctypes); entry points whose callers pass ctypes objects stay on ctypes either way.
FAST_PATH = os.path.join(_HERE, "_cfhip_fast" + (__import__("sysconfig").get_config_var("EXT_SUFFIX") or ".so"))
fast_bound = 0  # entry points served by the vectorcall module in this process


class _FastLib:
    """attribute access = the vectorcall wrapper when there is one, else the ctypes function of the same CDLL"""

    def __init__(self, cdll: ctypes.CDLL, fast) -> None:
        global fast_bound
        self.__dict__["_cdll"] = cdll
        n = 0
        for name in fast.names():
            if name in SIGNATURES and fast.bind(name, ctypes.cast(getattr(cdll, name), c_void_p).value):
                self.__dict__[name] = getattr(fast, name)
                n += 1
        fast_bound = n

    def __getattr__(self, name: str):
        return getattr(self.__dict__["_cdll"], name)


def _load_fast():
    if os.environ.get("CFHIP_FASTCALL", "1") == "0" or not os.path.isfile(FAST_PATH):
        return None
    import importlib.util

    spec = importlib.util.spec_from_file_location("_cfhip_fast", FAST_PATH)
    if spec is None or spec.loader is None:
        return None
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load() -> ctypes.CDLL:
    """Load libcfhip.so once; raises if it has not been built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib if RECORDER is None else _Recording(_lib)  # type: ignore[return-value]
    if not lib_exists():
        raise RuntimeError(
            f"libcfhip.so not found at {LIB_PATH}: run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or ./build_lib.sh) first — the HIP path has no fallback"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    fast = _load_fast()
    _lib = lib if fast is None else _FastLib(lib, fast)  # type: ignore[assignment]
    return _lib  # type: ignore[return-value]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().cfhip_last_error()
        raise RuntimeError(f"cfhip {what} failed (code {rc}): {msg.decode() if msg else ''}")
