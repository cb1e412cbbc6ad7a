"""ORACLE (test infrastructure): ctypes loader for the plain-C restatement ``oracle/fa3_oracle.c``."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_fa3.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise OSError(f"{_PATH} missing: run `make -C oracle`")
        lib = C.CDLL(_PATH)
        lib.oracle_fa3_fwd_f32.restype = C.c_int
        lib.oracle_fa3_fwd_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p]
        lib.oracle_fa3_tile.restype = C.c_int
        lib.oracle_fa3_tile.argtypes = [C.c_int, C.c_int]
        _lib = lib
    return _lib


def attention_bshd(q, k, v, causal=False, seqlens_k=None, scaling=None) -> torch.Tensor:
    """q [B,Sq,H,D], k/v [B,Sk,H,D] (any float dtype) -> fp32 [B,Sq,H,D] via the C restatement."""
    lib = load()
    qf, kf, vf = (t.float().contiguous() for t in (q, k, v))
    B, Sq, H, D = qf.shape
    Sk = kf.shape[1]
    out = torch.empty_like(qf)
    sl = None
    if seqlens_k is not None:
        sl = np.ascontiguousarray(np.asarray(seqlens_k, dtype=np.int32))
    lib.oracle_fa3_fwd_f32(qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), out.data_ptr(), B, H, Sq, Sk, D,
                           float(D ** -0.5 if scaling is None else scaling), int(bool(causal)),
                           None if sl is None else sl.ctypes.data)
    return out
