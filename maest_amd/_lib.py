"""ctypes binding of libmaest_hip.so (C ABI: include/maest_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a tensor that is not
on a HIP device reaches a kernel wrapper, an exception is raised.  (``_testing_override`` exists
so that tests/emu can run the same kernel sources under the host SIMT emulator; nothing in the
package calls it.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MAEST_HIP_LIB: another build of the same library -- e.g. the fallback form of maest_amd/build.py --leave-out, for running the suite against it)
LIB_PATH = os.environ.get("MAEST_HIP_LIB") or os.path.join(_HERE, "libmaest_hip.so")

F32 = 0
BF16 = 1
F16 = 3     # IEEE half, input of maest_patch_im2col only
BF16_QS = 4 # bf16 qkv tensor with q columns pre-multiplied by scale * log2(e) (maest_attn_* dtype only)
SPLIT3_A, SPLIT3_B, F32X3_A3 = 5, 6, 7   # the split-bf16 product as one bf16 GEMM of 3 K (include/maest_hip.h)
F32X3 = 2   # fp32 tensors, split-bf16 matrix products (maest_gemm_nt in_dtype / maest_attn_fwd dtype only)
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_MUL, EPI_ATOMIC = 0, 1, 2, 3, 4

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> argtypes, in the order of include/maest_hip.h
SIGNATURES = {
    "maest_gemm_nt": [_P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P, _I, _P, _P, _L, _I, _P],
    "maest_gemm_nt_rowdot": [_P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P, _P, _L, _P, _I, _P],
    "maest_gemm_tn": [_P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _P, _I, _P],
    "maest_gemm_tn_workspace_bytes": [_I, _I, _I, _I, _I, _P],
    "maest_gemm_tn_ws": [_P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _P, _I, _P, _L, _P],
    "maest_transpose": [_P, _L, _P, _L, _I, _I, _I, _P],
    "maest_cast_weights": [_P, _P, _P, _I, _I, _I, _P],
    "maest_cast_weights_multi": [_I, _P, _P, _P, _P, _P, _P, _F, _I, _P],
    "maest_layernorm_fwd": [_P, _L, _P, _P, _P, _L, _I, _P, _P, _I, _I, _F, _P],
    "maest_add_layernorm_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P],
    "maest_layernorm_bwd": [_P, _L, _I, _P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P],
    "maest_attn_fwd": [_P, _P, _P, _I, _I, _I, _F, _P],
    "maest_attn_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "maest_attn_fwd_rows": [_P, _P, _P, _I, _I, _I, _F, _I, _P],
    "maest_attn_bwd_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "maest_layernorm_bwd_headres": [_P, _L, _I, _P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "maest_gather_head_rows": [_P, _I, _I, _I, _I, _P, _P],
    "maest_scatter_head_rows": [_P, _I, _I, _I, _I, _I, _P, _P],
    "maest_patch_im2col": [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "maest_patch_im2col_strided": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "maest_token_assemble": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _P, _P],
    "maest_token_assemble_bwd": [_P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "maest_head_pool_fwd": [_P, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P],
    "maest_head_pool_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "maest_embed_pool": [_P, _I, _I, _P, _P],
    "maest_bce_logits": [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P],
    "maest_sigmoid_mean": [_P, _I, _I, _P, _P],
    "maest_colsum": [_P, _L, _I, _I, _I, _P, _P],
    "maest_spec_mask": [_P, _I, _I, _I, _P, _I, _P, _I, _P],
    "maest_swa_update_multi": [_I, _P, _P, _P, _F, _P],
    "maest_affine_f32": [_P, _L, _F, _F, _P],
    "maest_augment_mel": [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _P],
    "maest_melfile_assemble": [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
    "maest_logmel": [_P, _I, _I, _P, _P, _P, _P, _P, _I, _F, _F, _F, _P, _P],
    "maest_scale_f32": [_P, _L, _F, _P],
    "maest_scale_dev_f32": [_P, _L, _P, _P],
    "maest_cast_rows": [_P, _L, _P, _L, _I, _I, _I, _P],
    "maest_set_option": [_I, _I, _I],
    "maest_get_option": [_I, _P],
    "maest_set_option_thread": [_I, _I, _I],
    "maest_kernel_forms": [_P],
}
FORM_GEMM_NT_OW, FORM_GEMM_TN_OW, FORM_ATTN_FWD_PW = 1, 2, 4

ABI_VERSION = 8
OPTIONS = {"gemm_min_m": 0, "gemm_variant": 1, "gemm_epilogue": 2, "attn_bwd": 3, "ln_bwd_blocks": 4, "gemm_tail": 5, "attn_fwd": 6, "attn_fwd_waves": 7,
           "tn_reduce": 8, "gemm_wgs": 9, "gemm_panel": 10}

_lib = None
_lib_f16 = None
_host_emulation = False  # set only by tests/emu
# The library's second build: the same sources with IEEE half as the 16-bit operand type (csrc/common.h: MAEST_16BIT_F16).  A thread selects it
# for the calls it makes inside `with flavour("f16"):` (maest.py: precision="fp16" evaluation forwards); tensors keep the bf16 dtype TAG -- a
# 16-bit container whose bits the kernels of the selected build interpret.
LIB_PATH_F16 = os.environ.get("MAEST_HIP_LIB_F16") or os.path.join(_HERE, "libmaest_hip_f16.so")
import threading as _threading
_tls = _threading.local()


class flavour:
    """``with _lib.flavour("f16"): ...`` -- C-ABI calls of this thread go to libmaest_hip_f16.so inside the block."""

    def __init__(self, name):
        assert name in ("bf16", "f16")
        self.name = name

    def __enter__(self):
        self.prev = getattr(_tls, "flavour", "bf16")
        _tls.flavour = self.name
        return self

    def __exit__(self, *a):
        _tls.flavour = self.prev


def current_flavour():
    return getattr(_tls, "flavour", "bf16")


class MaestHipError(RuntimeError):
    pass


def _bind(lib):
    lib.maest_version.restype = c_int
    lib.maest_version.argtypes = []
    lib.maest_last_error.restype = c_char_p
    lib.maest_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = c_int
        fn.argtypes = argtypes
    return lib


def load():
    """Load (once) and return the bound library -- the build the calling thread's flavour selects; raise loudly when it is absent."""
    global _lib, _lib_f16
    if getattr(_tls, "flavour", "bf16") == "f16" and not _host_emulation:
        if _lib_f16 is None:
            if not os.path.exists(LIB_PATH_F16):
                raise MaestHipError(f"{LIB_PATH_F16} not found: precision=\"fp16\" needs the half-precision build of the kernels "
                                    "(`python -c 'import __graft_entry__ as g; g.build()'` builds both).")
            _lib_f16 = _bind(ctypes.CDLL(LIB_PATH_F16))
            if _lib_f16.maest_version() != ABI_VERSION:
                raise MaestHipError("libmaest_hip_f16.so ABI version mismatch")
        return _lib_f16
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MaestHipError(
                f"{LIB_PATH} not found: the MI355X kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "maest_amd has no CPU fallback.")
        _lib = _bind(ctypes.CDLL(LIB_PATH))
        if _lib.maest_version() != ABI_VERSION:
            raise MaestHipError("libmaest_hip.so ABI version mismatch")
    return _lib


def _testing_override(path):
    """tests/emu only: bind a host-emulation build of the same sources."""
    global _lib, _host_emulation
    _lib = _bind(ctypes.CDLL(path))
    _host_emulation = True
    return _lib


def _testing_restore():
    global _lib, _host_emulation
    _lib = None
    _host_emulation = False


def host_emulation():
    return _host_emulation


def kernel_forms():
    """Bit mask of the owned-register kernels present in this build (include/maest_hip.h: MAEST_FORM_*)."""
    m = c_int(0)
    call("maest_kernel_forms", ctypes.byref(m))
    return m.value


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MaestHipError(f"{name} failed (status {rc}): {lib.maest_last_error().decode()}")
