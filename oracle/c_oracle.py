"""ORACLE -- test infrastructure.  ctypes access to oracle/_build/liboracle.so (oracle/c/rba_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def rba_reduce(mask, prob, want_sem=True):
    lib = load()
    mask, prob = _f(mask), _f(prob)
    Q, HW = mask.shape[0], int(np.prod(mask.shape[1:]))
    K = prob.shape[1]
    rba = np.empty(HW, np.float32)
    sem = np.empty((K, HW), np.float32) if want_sem else None
    arg = np.empty(HW, np.int32)
    vp = ctypes.c_void_p
    rc = lib.oracle_rba_reduce_f32(vp(mask.ctypes.data), vp(prob.ctypes.data), vp(rba.ctypes.data),
                                   vp(sem.ctypes.data) if want_sem else None, vp(arg.ctypes.data),
                                   ctypes.c_int(Q), ctypes.c_int(K), ctypes.c_int64(HW))
    assert rc == 0
    sp = mask.shape[1:]
    return (sem.reshape((K,) + sp) if want_sem else None), rba.reshape(sp), arg.reshape(sp)


def ms_deform_attn(value, shapes, lsi, loc, w):
    lib = load()
    value, loc, w = _f(value), _f(loc), _f(w)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((N, Lq, M * D), np.float32)
    vp = ctypes.c_void_p
    rc = lib.oracle_ms_deform_attn_f32(vp(value.ctypes.data), vp(shapes.ctypes.data), vp(lsi.ctypes.data),
                                       vp(loc.ctypes.data), vp(w.ctypes.data), vp(out.ctypes.data),
                                       N, S, M, D, L, Lq, P)
    assert rc == 0
    return out
