"""The lane-accurate emulation of the reference's CUDA thread blocks (oracle/lane_emulation.py: every thread,
register, shared-memory cell and warp shuffle of CT:54-551, ST:97-258, RC:106-365, in complex128) must compute
what the oracle says the reference computes -- for every length, direction and reorder setting.  This is what
ties the oracle (and through it every parity test) to the reference's actual choreography rather than to a
reading of it.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

from oracle import lane_emulation as emu
from oracle import np_reference as ref
from tests import oracle_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]
TOL = 1e-12   # fp64 against fp64: relative L2 per FFT


@pytest.fixture(scope="module")
def olib():
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


def _cplx(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def _rel(a, b):
    return float(np.max(np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("reorder", [True, False])
def test_ct_block_emulation_equals_oracle(olib, n, inverse, reorder):
    rng = np.random.default_rng(n * 4 + 2 * inverse + reorder)
    x = _cplx(rng, (8 if n <= 64 else 2, n))
    got = emu.ct_external(x, inverse, reorder)
    assert _rel(got, ref.ct_c2c(x, inverse, reorder)) < TOL                       # S1 / S2 as stated with numpy.fft
    assert _rel(got, oracle_api.ct_c2c(olib, x, inverse, reorder, "f64")) < TOL   # the C restatement


def test_ct_4096_inverse_noreorder_class_is_forward_upstream():
    """FFT_4096_inverse_noreorder::fft_direction = 0 upstream (SM_FFT_parameters.cuh:388): with the class's real
    member the reference computes the FORWARD no-reorder transform; this library implements the inverse its
    name promises (documented deviation)."""
    rng = np.random.default_rng(4096)
    x = _cplx(rng, (1, 4096))
    upstream = emu.ct_external(x, True, False, direction_override=0)
    assert _rel(upstream, ref.ct_c2c(x, False, False)) < TOL
    assert _rel(emu.ct_external(x, True, False), ref.ct_c2c(x, True, False)) < TOL


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096])
def test_stockham_block_emulation_equals_oracle(olib, n):
    rng = np.random.default_rng(n)
    x = _cplx(rng, (2, n))
    got = emu.st_external(x)                       # the ST program: + sign only (ST:76)
    assert _rel(got, ref.st_c2c(x, True)) < TOL
    assert _rel(got, oracle_api.st_c2c(olib, x, True, "f64")) < TOL
    fwd = emu.st_external(x, inverse=False)        # RC's direction-templated C2C (RC:158-163)
    assert _rel(fwd, ref.st_c2c(x, False)) < TOL


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_r2c_c2r_block_emulation_equals_oracle(olib, n):
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal((2, n))
    packed = emu.rc_external(x, inverse=False)
    assert _rel(packed, ref.r2c_packed(x)) < TOL                                   # S5, element 0 = (DC, Nyquist)
    assert _rel(packed, oracle_api.r2c(olib, x, "f64")) < TOL
    back = emu.rc_external(packed, inverse=True)
    assert _rel(back, ref.c2r_packed(packed)) < TOL                                # S6
    assert _rel(back, oracle_api.c2r(olib, packed, "f64")) < TOL
    assert _rel(back, (n // 2) * x) < 1e-11                                        # C2R(R2C(x)) = (N/2) x  (RC:613)
