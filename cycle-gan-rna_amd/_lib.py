"""Binding of libcgrna.so (the C ABI declared in include/cgrna.h).

load() returns the thin PyTorch-ROCm C++ extension built over the C ABI (lib/_cgrna_torch*.so, generated from the
signature table below by build.py): one function per entry point, pointers passed as integers
(tensor.data_ptr()), and a stream argument that is either a hipStream_t handle or -(device index + 1) for "the
current HIP stream of that device", looked up in C++.  load_ctypes() binds the same library with ctypes (plain C ABI,
used by tests that call an entry point directly and by the export check).

torch must be imported before the library is loaded so that both share one HIP
runtime (the library's libamdhip64.so.7 dependency then resolves to the copy
torch already mapped).  There is no fallback: a missing library is an error.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcgrna.so")

P = C.c_void_p
I = C.c_int
F = C.c_float
Z = C.c_size_t
L = C.c_longlong


class GlossDesc(C.Structure):
    _fields_ = [("n_terms", C.c_int * 6), ("slot", (C.c_int * 4) * 6), ("weight", (C.c_double * 4) * 6),
                ("lambda_", C.c_double * 6), ("ema_alpha", C.c_double), ("weights", C.c_void_p)]


SIGNATURES = {
    "cg_version": ([], I),
    "cg_gemm": ([P, I, P, I, P, I, I, I, I, I, I, I, P, I, P, I, P, I, F, I, P, Z, P], I),
    "cg_gemm_ex": ([P, I, P, I, P, I, I, I, I, I, I, I, P, I, P, I, P, I, F, I, P, P, Z, P], I),
    "cg_gemm_slabs": ([P, I, P, I, P, I, I, I, I, I, I, I, P, P, Z, P, P], I),
    "cg_bn_fwd_slabs_ok": ([I, I], I),
    "cg_bn_fwd_slabs": ([P, I, P, I, P, I, P, I, I, P, P, F, P, P, P, P, F, F, P, I, P, I, P, I, P], I),
    "cg_interp_hidden": ([P, I, P, P, I, I, P, I, I, I, P], I),
    "cg_bn_workspace_bytes": ([I, I], Z),
    "cg_bn_fwd": ([P, I, P, I, I, I, P, P, F, P, P, P, P, F, F, P, I, P, I, P, I, P, Z, P], I),
    "cg_bn_fwd_static": ([P, I, P, I, I, I, P, P, F, P, P, P, P, F, F, P, I, P, I, P, I, P], I),
    "cg_bn_fwd_seg": ([P, I, P, I, I, I, C.c_uint, I, P, P, F, P, P, P, P, F, F, P, I, P, I, P, I, P, Z, P], I),
    "cg_bn_bwd_seg": ([P, I, P, I, P, I, P, I, P, I, P, P, P, P, P, I, P, P, F, I, I, I, P, I, P, Z, P], I),
    "cg_tok_blocks": ([I], I),
    "cg_tok_lin_fwd": ([P, P, P, P, P, I, P], I),
    "cg_tok_bn_lin_fwd": ([P, P, P, P, P, P, P, F, P, P, P, P, F, P, P, I, P], I),
    "cg_tok_bn_res_fwd": ([P, P, P, P, P, F, P, P, P, P, F, P, I, P], I),
    "cg_tok_bn_bwd_stats": ([P, P, P, P, P, I, P], I),
    "cg_tok_bwd_mid": ([P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P], I),
    "cg_tok_bwd_in": ([P, P, P, P, P, P, P, P, P, P, P, P, P, I, P], I),
    "cg_tok_wgrad_reduce": ([P, P, P, P, I, P], I),
    "cg_bn_running_update": ([P, P, P, P, I, I, F, F, P], I),
    "cg_bn_eval": ([P, I, P, I, I, I, P, P, P, P, F, F, P, I, P], I),
    "cg_bn_bwd": ([P, I, P, I, P, I, P, I, P, I, P, P, P, P, P, I, P, P, F, I, I, P, I, P, Z, P], I),
    "cg_bn_bwd_slabs_ok": ([I, I], I),
    "cg_bn_bwd_slabs": ([P, I, P, I, P, I, P, I, P, I, P, P, P, P, P, I, F, I, I, P, I, P], I),
    "cg_bn_bwd_bwd": ([P, I, P, I, P, I, P, I, P, I, P, I, P, P, P, P, P, I, I, P, P, I, P, Z, P], I),
    "cg_axpby": ([P, I, F, P, I, F, P, I, I, I, P], I),
    "cg_mul": ([P, I, P, I, P, I, I, I, P], I),
    "cg_fill_coef": ([P, I, I, I, P, I, F, P], I),
    "cg_to_bf16": ([P, I, P, I, I, I, P], I),
    "cg_embed_gather": ([P, I, P, I, P, I, I, I, P, P], I),
    "cg_copy_bf16": ([P, P, I, P, I, I, I, P], I),
    "cg_label_colsum": ([P, P, I, P, I, I, I, I, F, I, P, I, P], I),
    "cg_embed_scatter": ([P, I, P, I, P, I, I, I, I, P], I),
    "cg_reparam_fwd": ([P, I, P, I, P, I, I, P], I),
    "cg_reparam_bwd": ([P, I, P, I, P, I, P, P, I, F, I, I, P], I),
    "cg_libsize_fwd": ([P, I, P, I, P, F, I, I, P], I),
    "cg_libsize_bwd": ([P, I, P, I, P, I, P, F, I, I, P], I),
    "cg_gp_interp": ([P, I, P, I, P, I, P, I, I, P], I),
    "cg_critic_inputs": ([P, I, P, I, P, I, P, I, I, P, P, P, I, I, P, P], I),
    "cg_gp_combine": ([P, I, P, I, P, P, I, P, I, I, P, I, I, I, P, P], I),
    "cg_gp_penalty": ([P, I, P, P, I, F, I, I, P, P], I),
    "cg_gp_penalty_parts": ([P, I, P, P, P, I, F, I, I, P, P], I),
    "cg_reduce_sum": ([P, P, P, I, I, I, P], I),
    "cg_reduce_l1": ([P, P, P, I, P, I, I, I, P], I),
    "cg_reduce_poisson": ([P, P, P, I, P, I, I, I, P], I),
    "cg_reduce_centroid_sq": ([P, P, P, I, P, I, P, I, I, I, I, P], I),
    "cg_reduce_kl": ([P, P, P, I, I, I, P], I),
    "cg_colsum": ([P, P, I, I, I, F, I, P], I),
    "cg_copy4": ([P, P, L, P, P, L, P, P, L, P, P, L, P], I),
    "cg_colsum3": ([P, P, I, I, I, F, I, P, P, I, I, I, F, I, P, P, I, I, I, F, I, P], I),
    "cg_reduce_sum_seg": ([P, P, P, I, I, I, I, P], I),
    "cg_reduce_sqdiff": ([P, P, P, I, P, I, I, I, P], I),
    "cg_colvar_mean": ([P, P, P, I, I, I, P], I),
    "cg_l1_bwd": ([P, I, P, I, P, I, P, I, F, I, I, I, P], I),
    "cg_poisson_bwd": ([P, I, P, I, P, I, P, I, F, I, I, I, P], I),
    "cg_centroid_bwd": ([P, I, P, I, P, I, P, I, P, I, F, I, I, I, I, P], I),
    "cg_fm_bwd": ([P, I, P, P, P, I, F, I, I, I, P], I),
    "cg_gloss_finalize": ([P, P, P, P, P, P, P], I),
    "cg_sum_parts": ([P, P, P, P, P, I, P], I),
    "cg_adam_amsgrad": ([P, P, P, P, P, Z, P, F, F, F, P, P], I),
    "cg_attn_ln_fwd": ([P, P, P, P, I, I, I, P], I),
    "cg_attn_block_blocks": ([I], I),
    "cg_attn_block_bwd": ([P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, I, P], I),
    "cg_attn_param_reduce_blocks": ([P, L, P, L, I, I, I, P], I),
    "cg_tok_wgrad_reduce_blocks": ([P, P, L, P, P, L, I, I, P], I),
    "cg_tok_bwd_mid_from": ([P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, P], I),
    "cg_attn_block_fwd": ([P, P, P, P, P, I, P, P, F, P, P, P, P, F, P, P, I, I, I, P, P, P, P, P], I),
    "cg_tok_bn_lin_fwd_from": ([P, P, P, P, P, I, P, P, F, P, P, P, P, F, P, P, I, P], I),
    "cg_attn_bwd_blocks": ([I], I),
    "cg_attn_bwd_ws": ([I, I, I], C.c_longlong),
    "cg_attn_ln_bwd": ([P, P, P, P, P, P, P, I, I, I, I, P], I),
    "cg_gemm_timing_begin": ([I], I),
    "cg_gemm_timing_end": ([P, I], I),
}

_lib = None
_ext = None


def _missing(path):
    return RuntimeError(f"{path} is missing: build it with `python cycle-gan-rna_amd/build.py` "
                        "(this package has no CPU or PyTorch fallback for its kernels)")


def load_ctypes():
    """The library bound with ctypes, argtypes set; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CGRNA_LIB", LIB_PATH)     # tuning aid: a differently built copy of the same library
    if not os.path.exists(path):
        raise _missing(path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def load():
    """The C++ extension over the C ABI (see the module docstring); raises if it or the library is absent."""
    global _ext
    if _ext is not None:
        return _ext
    if os.environ.get("CGRNA_LIB"):      # a differently built copy of the library: the ctypes binding can take it
        return load_ctypes()
    if not os.path.exists(LIB_PATH):
        raise _missing(LIB_PATH)
    import glob
    import importlib.util
    found = glob.glob(os.path.join(_HERE, "lib", "_cgrna_torch*.so"))
    if not found:
        raise _missing(os.path.join(_HERE, "lib", "_cgrna_torch*.so"))
    spec = importlib.util.spec_from_file_location("_cgrna_torch", found[0])
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    for name in SIGNATURES:
        getattr(ext, name)               # AttributeError if the extension is older than the table
    _ext = ext
    return ext


class CgError(RuntimeError):
    pass


def check(status: int, what: str):
    if status != 0:
        kind = {-1: "bad argument", -2: "unsupported", -3: "workspace too small"}.get(status, f"hipError {status}")
        raise CgError(f"{what}: {kind}")
