"""ctypes binding of ``libpfa_hip.so`` (C ABI declared in ``include/pfa_hip.h``).

PyTorch is plumbing here: it owns device memory and streams; the library gets raw device
pointers, element strides and the current ``hipStream_t``.  There is NO CPU or eager
fallback in this module: if the native library is missing or refuses the arguments the
call raises.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

PFA_ABI_VERSION = 7
PFA_DTYPE_BF16, PFA_DTYPE_FP16, PFA_DTYPE_FP32 = 0, 1, 2
PFA_FLAG_SPLIT_P = 0x1
PFA_FLAG_NO_XCD_MAP = 0x2

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libpfa_hip.so")

EXPORTS = (
    "pfa_abi_version", "pfa_status_string", "pfa_device_supported", "pfa_last_hip_error",
    "pfa_fa3_workspace_bytes", "pfa_fa3_check", "pfa_fa3_fwd", "pfa_fa3_describe", "pfa_fa3_weights",
    "pfa_fa3_bwd", "pfa_fa3_bwd_workspace_bytes", "pfa_fa3_bwd_mask_workspace_bytes", "pfa_fa3_prepare", "pfa_probe_mfma",
)


class PfaFa3Args(C.Structure):
    """Mirror of ``struct pfa_fa3_args`` (include/pfa_hip.h)."""
    _fields_ = [
        ("size", C.c_uint32), ("flags", C.c_uint32),
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("lse", C.c_void_p), ("seqlens_k", C.c_void_p), ("key_mask", C.c_void_p),
        ("q_stride_b", C.c_int64), ("q_stride_h", C.c_int64), ("q_stride_s", C.c_int64),
        ("k_stride_b", C.c_int64), ("k_stride_h", C.c_int64), ("k_stride_s", C.c_int64),
        ("v_stride_b", C.c_int64), ("v_stride_h", C.c_int64), ("v_stride_s", C.c_int64),
        ("o_stride_b", C.c_int64), ("o_stride_h", C.c_int64), ("o_stride_s", C.c_int64),
        ("key_mask_stride_b", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32), ("Sk", C.c_int32), ("D", C.c_int32),
        ("dtype_in", C.c_int32), ("dtype_out", C.c_int32), ("causal", C.c_int32),
        ("softmax_scale", C.c_float), ("device_id", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("mask", C.c_void_p),
        ("mask_stride_b", C.c_int64), ("mask_stride_h", C.c_int64), ("mask_stride_q", C.c_int64),
        ("mask_stride_k", C.c_int64),
        ("kv_group", C.c_int32), ("reserve_cus", C.c_int32),
        ("drop_mask", C.c_void_p), ("drop_scale", C.c_float), ("reserved1", C.c_int32),
    ]


class PfaFa3BwdArgs(C.Structure):
    """Mirror of ``struct pfa_fa3_bwd_args`` (include/pfa_hip.h)."""
    _fields_ = (
        [("size", C.c_uint32), ("flags", C.c_uint32)]
        + [(n, C.c_void_p) for n in ("q", "k", "v", "o", "dout", "lse", "dq", "dk", "dv", "delta", "seqlens_k")]
        + [(f"{t}_stride_{a}", C.c_int64) for t in ("q", "k", "v", "o", "do", "dq", "dk", "dv") for a in "bhs"]
        + [(n, C.c_int32) for n in ("B", "H", "Sq", "Sk", "D", "dtype", "dtype_grad", "causal")]
        + [("softmax_scale", C.c_float), ("device_id", C.c_int32)]
        + [("mask", C.c_void_p)] + [(f"mask_stride_{a}", C.c_int64) for a in "bhqk"]
        + [("drop_mask", C.c_void_p), ("drop_scale", C.c_float), ("kv_group", C.c_int32)]
        + [("mask_workspace", C.c_void_p), ("mask_workspace_bytes", C.c_size_t)]
    )


class PfaError(RuntimeError):
    """A non-zero ``pfa_status`` from the native library."""

    def __init__(self, status: int, text: str):
        super().__init__(f"libpfa_hip: {text} (status {status})")
        self.status = status


_lib = None
_lock = threading.Lock()


def load(path: Optional[str] = None):
    """Load (once) and type the library.  Raises ``OSError`` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        p = path or os.environ.get("PFA_HIP_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise OSError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C photonic_flash_attention_amd/csrc` (there is no CPU fallback)")
        lib = C.CDLL(p)
        lib.pfa_abi_version.restype = C.c_int
        lib.pfa_status_string.restype = C.c_char_p
        lib.pfa_status_string.argtypes = [C.c_int]
        lib.pfa_device_supported.restype = C.c_int
        lib.pfa_device_supported.argtypes = [C.c_int]
        lib.pfa_last_hip_error.restype = C.c_int
        lib.pfa_fa3_workspace_bytes.restype = C.c_size_t
        lib.pfa_fa3_workspace_bytes.argtypes = [C.POINTER(PfaFa3Args)]
        lib.pfa_fa3_check.restype = C.c_int
        lib.pfa_fa3_check.argtypes = [C.POINTER(PfaFa3Args)]
        lib.pfa_fa3_fwd.restype = C.c_int
        lib.pfa_fa3_fwd.argtypes = [C.POINTER(PfaFa3Args), C.c_void_p]
        lib.pfa_fa3_weights.restype = C.c_int
        lib.pfa_fa3_weights.argtypes = [C.POINTER(PfaFa3Args), C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_void_p]
        lib.pfa_fa3_bwd.restype = C.c_int
        lib.pfa_fa3_bwd.argtypes = [C.POINTER(PfaFa3BwdArgs), C.c_void_p]
        lib.pfa_fa3_bwd_workspace_bytes.restype = C.c_size_t
        lib.pfa_fa3_bwd_workspace_bytes.argtypes = [C.POINTER(PfaFa3BwdArgs)]
        lib.pfa_fa3_bwd_mask_workspace_bytes.restype = C.c_size_t
        lib.pfa_fa3_bwd_mask_workspace_bytes.argtypes = [C.POINTER(PfaFa3BwdArgs)]
        lib.pfa_fa3_describe.restype = C.c_int
        lib.pfa_fa3_describe.argtypes = [C.POINTER(PfaFa3Args), C.c_char_p, C.c_size_t]
        lib.pfa_fa3_prepare.restype = C.c_int
        lib.pfa_fa3_prepare.argtypes = [C.c_int]
        lib.pfa_probe_mfma.restype = C.c_int
        lib.pfa_probe_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        v = lib.pfa_abi_version()
        if v != PFA_ABI_VERSION:
            raise OSError(f"{p}: ABI version {v}, binding expects {PFA_ABI_VERSION}")
        _lib = lib
    return _lib


def status_string(status: int) -> str:
    return load().pfa_status_string(int(status)).decode()


def check_status(status: int) -> None:
    if status != 0:
        extra = ""
        if status == -9:
            extra = f" [hipError {load().pfa_last_hip_error()}]"
        raise PfaError(status, status_string(status) + extra)


def make_args(**kw) -> PfaFa3Args:
    a = PfaFa3Args()
    a.size = C.sizeof(PfaFa3Args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def describe(args: PfaFa3Args):
    """-> (kernel variant name, number of workgroups) the library would launch."""
    buf = C.create_string_buffer(128)
    n = load().pfa_fa3_describe(C.byref(args), buf, 128)
    if n < 0:
        check_status(n)
    return buf.value.decode(), n
