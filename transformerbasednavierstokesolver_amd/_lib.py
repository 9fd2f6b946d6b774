"""ctypes binding of libpa2d.so (include/pa2d.h).  The library is built in-tree by
`__graft_entry__.build()` / `make -C transformerbasednavierstokesolver_amd/csrc`.

There is deliberately NO fallback: if the shared object is missing or a symbol cannot be bound the
import of the compute path raises, and every non-zero return code becomes a RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PA2D_LIB") or os.path.join(_HERE, "csrc", "libpa2d.so")   # PA2D_LIB: A/B builds

_f = C.c_void_p      # const float* / float* (device pointers travel as integers)
_i = C.c_int
_ll = C.c_longlong
_sz = C.c_size_t
_st = C.c_void_p     # hipStream_t

# name -> (restype, argtypes); mirrors include/pa2d.h one to one
SIGNATURES = {
    "pa2d_version": (C.c_char_p, []),
    "pa2d_default_engine": (_i, []),
    "pa2d_reload_env": (None, []),
    "pa2d_layernorm_fwd": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, C.c_float, _st]),
    "pa2d_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "pa2d_layernorm_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _st]),
    "pa2d_gemm_fwd_workspace": (_sz, [_i, _i, _i]),
    "pa2d_gemm_weight_image": (_i, [_f, _ll, _i, _f, _sz, _i, _i, _i, _st]),
    "pa2d_gemm_bias_act_fwd": (_i, [_f, _ll, _f, _ll, _f, _f, _ll, _f, _ll, _f, _ll, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "pa2d_gemm_bwd_data_workspace": (_sz, [_i, _i, _i]),
    "pa2d_gemm_bwd_data": (_i, [_f, _ll, _f, _ll, _f, _ll, _i, _f, _ll, _f, _f, _sz, _i, _i, _i, _i, _st]),
    "pa2d_gemm_bwd_weight_workspace": (_sz, [_i, _i, _i, _i]),
    "pa2d_gemm_bwd_weight": (_i, [_f, _ll, _f, _ll, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "pa2d_conv3x3x2_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "pa2d_conv3x3x2_fwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "pa2d_conv3x3x2_pack_bytes": (_sz, [_i]),
    "pa2d_conv3x3x2_pack": (_i, [_f, _f, _f, _sz, _i, _i, _i, _i, _i, _i, _st]),
    "pa2d_conv3x3x2_fwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_conv3x3x2_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _i, _st, _st,
                                _st]),
    "pa2d_slice_nchunk": (_i, [_i, _i, _i]),
    "pa2d_slice_scatter": (_i, [_f, _ll, _f, _ll, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_token_attn_lds_bytes": (_sz, [_i, _i, _i]),
    "pa2d_token_attn_fwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _st]),
    "pa2d_token_attn_bwd_workspace": (_sz, [_i, _i]),
    "pa2d_token_attn_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "pa2d_deslice_fwd": (_i, [_f, _ll, _f, _f, _f, _f, _f, _ll, _i, _i, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_slice_bwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "pa2d_slice_bwd_points": (_i, [_f, _ll, _f, _ll, _f, _ll, _f, _f, _f, _f, _f, _f, _f, _ll, _f, _ll, _f, _f, _f,
                                   _f, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_head_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _st]),
    "pa2d_head_bwd_workspace": (_sz, [_i, _i, _i]),
    "pa2d_head_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _st]),
    "pa2d_act_bwd": (_i, [_f, _f, _f, _ll, _i, _st]),
    "pa2d_sumsq_workspace": (_sz, [_ll]),
    "pa2d_sumsq": (_i, [_f, _ll, _f, _f, _sz, _st]),
    "pa2d_adamw_step": (_i, [_f, _f, _f, _f, _ll, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _f,
                             C.c_float, _st]),
    "pa2d_rel_l2_fwd": (_i, [_f, _f, _f, _f, _f, _i, _ll, _st]),
    "pa2d_rel_l2_bwd": (_i, [_f, _f, _f, _f, _f, _f, _i, _ll, _st]),
}
# bf16-storage variants: identical argument lists (activation pointers simply hold bf16); the GEMM / conv ones have no
# `engine` argument (they ARE the bf16 engine)
for _name in ("pa2d_layernorm_fwd", "pa2d_layernorm_bwd", "pa2d_head_fwd", "pa2d_head_bwd"):
    SIGNATURES[_name + "_bf16"] = SIGNATURES[_name]
# the slice stages: the fp32-storage entry points take `engine` (exact-fp32 MFMA or bf16 splits), the bf16-storage ones do not
SIGNATURES["pa2d_slice_scatter_bf16"] = (_i, [_f, _ll, _f, _ll, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _st, _st, _st])
SIGNATURES["pa2d_deslice_fwd_bf16"] = (_i, [_f, _ll, _f, _f, _f, _f, _f, _ll, _i, _i, _i, _i, _i, _i, _st, _st, _st])
SIGNATURES["pa2d_slice_bwd_points_bf16"] = (_i, [_f, _ll, _f, _ll, _f, _ll, _f, _f, _f, _f, _f, _f, _f, _ll, _f, _ll, _f, _f,
                                                 _f, _f, _sz, _i, _i, _i, _i, _i, _i, _i, _st, _st, _st])
SIGNATURES.update({
    "pa2d_planes_bytes": (_sz, [_ll, _i, _i]),
    "pa2d_conv3x3x2_planes_mask": (_i, [_i, _i, _i, _i, _i]),
    "pa2d_layernorm_fwd_planes": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, C.c_float, _i, _st]),
    "pa2d_conv3x3x2_fwd_planes": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_conv3x3x2_workspace_planes": (_sz, [_i, _i, _i, _i, _i]),
    "pa2d_conv3x3x2_bwd_planes": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_slice_bwd_points_planes": (_i, [_f, _ll, _f, _ll, _f, _ll, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f,
                                          _sz, _i, _i, _i, _i, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_gemm_bias_act_fwd_bf16": (_i, [_f, _ll, _f, _ll, _f, _f, _ll, _f, _ll, _f, _ll, _i, _i, _i, _i, _st]),
    "pa2d_gemm_bwd_data_bf16": (_i, [_f, _ll, _f, _ll, _f, _ll, _i, _f, _ll, _f, _i, _i, _i, _st]),
    "pa2d_gemm_bwd_weight_workspace_bf16": (_sz, [_i, _i, _i]),
    "pa2d_gemm_bwd_weight_bf16": (_i, [_f, _ll, _f, _ll, _f, _f, _f, _sz, _i, _i, _i, _i, _st]),
    "pa2d_conv3x3x2_pack_bf16": (_i, [_f, _f, _f, _sz, _i, _i, _st]),
    "pa2d_conv3x3x2_workspace_bf16": (_sz, [_i, _i, _i, _i]),
    "pa2d_conv3x3x2_fwd_workspace_bf16": (_sz, [_i, _i, _i, _i]),
    "pa2d_conv3x3x2_fwd_bf16": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _st, _st, _st]),
    "pa2d_conv3x3x2_bwd_bf16": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st, _st, _st]),
})

ERRORS = {1001: "PA2D_ERR_ARG (alignment/shape contract)", 1002: "PA2D_ERR_UNSUPPORTED (size outside kernel grid)",
          1003: "PA2D_ERR_WORKSPACE (workspace too small)"}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def _try_build():
    """The shared object is a build artefact (git-ignored).  If it is absent but the toolchain is present
    (same ROCm image), compile it once in-tree — this builds the native path, it is not a fallback."""
    import shutil
    import subprocess
    csrc = os.path.dirname(LIB_PATH)
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if hipcc is None or shutil.which("make") is None or os.environ.get("PA2D_NO_AUTOBUILD"):
        return
    try:
        subprocess.run(["make", "-C", csrc, "-j4", f"HIPCC={hipcc}"], check=True, stdout=subprocess.DEVNULL)
    except Exception:      # the caller raises NativeLibraryMissing with the manual instructions
        pass


def load():
    """Load libpa2d.so once and bind every symbol of include/pa2d.h; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        _try_build()
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C transformerbasednavierstokesolver_amd/csrc`). There is no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libpa2d: {what} failed with code {rc} ({ERRORS.get(rc, 'hipError_t')})")
