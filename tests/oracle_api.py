"""ctypes helpers around oracle/liboracle.so (the CPU restatement).  Test infrastructure only."""
import ctypes

import numpy as np


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _real(prec):
    return (np.float32, np.complex64, ctypes.c_float, "_f32") if prec == "f32" else (np.float64, np.complex128, ctypes.c_double, "_f64")


def ct_c2c(lib, x, inverse, reorder, prec="f32"):
    rt, ctp, cty, suf = _real(prec)
    x = np.ascontiguousarray(x, dtype=ctp)
    out = np.empty_like(x)
    n = x.shape[-1]
    getattr(lib, "oracle_ct_c2c" + suf)(_ptr(x, cty), _ptr(out, cty), n, x.size // n, int(inverse), int(reorder))
    return out


def st_c2c(lib, x, inverse=True, prec="f32"):
    rt, ctp, cty, suf = _real(prec)
    x = np.ascontiguousarray(x, dtype=ctp)
    out = np.empty_like(x)
    n = x.shape[-1]
    getattr(lib, "oracle_st_c2c" + suf)(_ptr(x, cty), _ptr(out, cty), n, x.size // n, int(inverse))
    return out


def r2c(lib, x, prec="f32"):
    rt, ctp, cty, suf = _real(prec)
    x = np.ascontiguousarray(x, dtype=rt)
    n = x.shape[-1]
    out = np.empty(x.shape[:-1] + (n // 2,), dtype=ctp)
    getattr(lib, "oracle_r2c_c2r" + suf)(_ptr(x, cty), _ptr(out, cty), n, x.size // n, 0)
    return out


def c2r(lib, xp, prec="f32"):
    rt, ctp, cty, suf = _real(prec)
    xp = np.ascontiguousarray(xp, dtype=ctp)
    half = xp.shape[-1]
    out = np.empty(xp.shape[:-1] + (2 * half,), dtype=rt)
    getattr(lib, "oracle_r2c_c2r" + suf)(_ptr(xp, cty), _ptr(out, cty), 2 * half, xp.size // half, 1)
    return out
