"""BASELINE.json config 1: N=256 C2C forward, 1024 FFTs, CPU only (plumbing, no GPU): the C
restatement (both arithmetic builds) against the NumPy fp64 reference, the FFTW-API baseline shim
(when an FFTW3 provider resolves) against the same reference, and the reference harness's own
comparison metric (max_error = 1e-4) evaluated between the two fp32 results, as upstream does
between cuFFT and smFFT."""
import ctypes
import os

import numpy as np

from oracle import np_reference as ref
from tests import oracle_api as oa

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_config1_plumbing(oracle_lib):
    n, nffts = 256, 1024
    rng = np.random.default_rng(20200720)
    x = (rng.random((nffts, n), dtype=np.float32) + 1j * rng.random((nffts, n), dtype=np.float32)).astype(np.complex64)
    want = ref.ct_c2c(x, False, True)
    got64 = oa.ct_c2c(oracle_lib, x, 0, 1, "f64")
    l2, mx = ref.fft_errors(got64, want)
    assert l2 < 1e-13 and mx < 1e-13
    got32 = oa.ct_c2c(oracle_lib, x, 0, 1, "f32")
    l2, mx = ref.fft_errors(got32, want)
    assert l2 < 2e-6

    path = os.path.join(ROOT, "oracle", "fftw_baseline.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "fftw_baseline.so"])
    fb = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    fb.fftw_baseline_init.argtypes = [ctypes.c_int]
    fb.fftw_baseline_c2c.restype = ctypes.c_double
    fb.fftw_baseline_c2c.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    fb.fftw_baseline_c2c_sliced.restype = ctypes.c_double
    fb.fftw_baseline_c2c_sliced.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if fb.fftw_baseline_init(1):
        out = np.empty_like(x)
        t = fb.fftw_baseline_c2c(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, nffts, 0, 2)
        assert t > 0
        ref.assert_close_fp32(out, want, "FFTW-API baseline, config 1")
        out2 = np.empty_like(x)
        t = fb.fftw_baseline_c2c_sliced(x.ctypes.data_as(fp), out2.ctypes.data_as(fp), n, nffts, 0, 2, 3)
        assert t > 0
        ref.assert_close_fp32(out2, want, "FFTW-API baseline (sliced), config 1")
        # the harness metric between two fp32 results (CT/FFT.c:52-77): PASS = 0 elements above 1e-4
        f = lambda v: v.view(np.float32).ctypes.data_as(fp)  # noqa: E731
        assert oracle_lib.oracle_compare_data(f(out), f(got32), n, nffts, 1e-4, None, None) == 0
