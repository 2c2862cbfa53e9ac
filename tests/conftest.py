import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_product():
    """The product library must exist before anything imports smfft_amd (it has no CPU fallback and
    refuses to import without libsmfft_amd.so).  hipcc cross-compiles for gfx950 without a GPU."""
    lib = os.path.join(ROOT, "smfft_amd", "libsmfft_amd.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "smfft_amd", "csrc"), "-j", "8"])
    return lib


@pytest.fixture(scope="session")
def oracle_lib():
    """The C restatement (oracle/liboracle.so), built on demand.  Checker only."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "smfft_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    lib = ctypes.CDLL(path)
    fp, dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    for suf, p in (("_f32", fp), ("_f64", dp)):
        getattr(lib, "oracle_ct_c2c" + suf).argtypes = [p, p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]
        getattr(lib, "oracle_st_c2c" + suf).argtypes = [p, p, ctypes.c_int, ctypes.c_long, ctypes.c_int]
        getattr(lib, "oracle_r2c_c2r" + suf).argtypes = [p, p, ctypes.c_int, ctypes.c_long, ctypes.c_int]
    lib.oracle_get_error.restype = ctypes.c_float
    lib.oracle_get_error.argtypes = [ctypes.c_float, ctypes.c_float]
    lib.oracle_compare_data.restype = ctypes.c_long
    lib.oracle_compare_data.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_double, dp, dp]
    lib.oracle_compare_r2c.restype = ctypes.c_long
    lib.oracle_compare_r2c.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_double]
    lib.oracle_compare_c2r.restype = ctypes.c_long
    lib.oracle_compare_c2r.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_double]
    return lib


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "smfft_golden.npz"))
