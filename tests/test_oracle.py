"""Pins the CPU oracle (oracle/smfft_oracle.c) against the committed fp64 NumPy fixtures, analytic
known-answer tests and the S1..S6 identities of SURVEY.md 8(a)/(c).  No GPU needed."""
import numpy as np
import pytest

from oracle import np_reference as ref
from tests import oracle_api as oa

C2C_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]
R2C_SIZES = [512, 1024, 2048, 4096]


@pytest.mark.parametrize("n", C2C_SIZES)
@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("reo", [0, 1])
def test_ct_matches_golden(oracle_lib, golden, n, inv, reo):
    x = golden[f"c2c_in_u01_{n}"]
    want = golden[f"ct_out_u01_{n}_inv{inv}_reo{reo}"]
    got64 = oa.ct_c2c(oracle_lib, x, inv, reo, "f64")
    l2, mx = ref.fft_errors(got64, want)
    assert l2 < 1e-13 and mx < 1e-13, (l2, mx)
    got32 = oa.ct_c2c(oracle_lib, x, inv, reo, "f32")
    # the fp32 restatement uses the reference's radix-2 ladder with fp32 sincosf twiddles: it is
    # less accurate than the HIP engine, so it gets its own (looser) documented bound
    l2, mx = ref.fft_errors(got32, want)
    assert l2 < 2e-6 and mx < 4e-6, (l2, mx)


@pytest.mark.parametrize("n", C2C_SIZES)
def test_ct_zero_mean_golden(oracle_lib, golden, n):
    x = golden[f"c2c_in_u11_{n}"]
    want = golden[f"ct_out_u11_{n}_inv0_reo1"]
    l2, mx = ref.fft_errors(oa.ct_c2c(oracle_lib, x, 0, 1, "f64"), want)
    assert l2 < 1e-13 and mx < 1e-13


@pytest.mark.parametrize("n", [32, 64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("inv", [0, 1])
def test_stockham_matches_golden(oracle_lib, golden, n, inv):
    x = golden[f"c2c_in_u01_{n}"]
    want = golden[f"ct_out_u01_{n}_inv{inv}_reo1"]  # S3/S4 == S1 with the same sign
    l2, mx = ref.fft_errors(oa.st_c2c(oracle_lib, x, inv, "f64"), want)
    assert l2 < 1e-13 and mx < 1e-13
    l2, mx = ref.fft_errors(oa.st_c2c(oracle_lib, x, inv, "f32"), want)
    assert l2 < 2e-6 and mx < 4e-6


@pytest.mark.parametrize("n", R2C_SIZES)
def test_r2c_c2r_match_golden(oracle_lib, golden, n):
    l2, mx = ref.fft_errors(oa.r2c(oracle_lib, golden[f"r2c_in_{n}"], "f64"), golden[f"r2c_out_{n}"])
    assert l2 < 1e-13 and mx < 1e-13
    l2, mx = ref.fft_errors(oa.c2r(oracle_lib, golden[f"c2r_in_{n}"], "f64"), golden[f"c2r_out_{n}"])
    assert l2 < 1e-13 and mx < 1e-13
    l2, mx = ref.fft_errors(oa.r2c(oracle_lib, golden[f"r2c_in_{n}"], "f32"), golden[f"r2c_out_{n}"])
    assert l2 < 2e-6 and mx < 4e-6


# ---------------------------------------------------------------- known-answer tests (8(c) item 3)
@pytest.mark.parametrize("n", [32, 256, 1024, 4096])
def test_kat_impulse_constant_tone(oracle_lib, n):
    k = np.arange(n)
    n0 = 5
    x = np.zeros((1, n), np.complex128)
    x[0, n0] = 1
    np.testing.assert_allclose(oa.ct_c2c(oracle_lib, x, 0, 1, "f64")[0], np.exp(-2j * np.pi * k * n0 / n), atol=1e-12)
    np.testing.assert_allclose(oa.ct_c2c(oracle_lib, x, 1, 1, "f64")[0], np.exp(+2j * np.pi * k * n0 / n), atol=1e-12)
    x = np.ones((1, n), np.complex128)
    want = np.zeros(n)
    want[0] = n
    np.testing.assert_allclose(oa.ct_c2c(oracle_lib, x, 0, 1, "f64")[0], want, atol=1e-10)
    k0 = 7
    x = np.exp(2j * np.pi * k0 * k / n)[None]
    want = np.zeros(n)
    want[k0] = n
    np.testing.assert_allclose(oa.ct_c2c(oracle_lib, x, 0, 1, "f64")[0], want, atol=1e-9)


def test_kat_two_tone_generate_signal(oracle_lib):
    # Generate_signal (SMFFT_CooleyTukey_C2C/FFT.c:14-21): sin at f1 = 1/8 and f2 = 2/8 of the sample rate
    n = 1024
    f = np.arange(n)
    sig = 1.0 * np.sin(2 * np.pi * f / 8) + 0.5 * np.sin(2 * np.pi * 2 * f / 8 + 3 * np.pi / 4)
    spec = oa.ct_c2c(oracle_lib, sig.astype(np.complex128)[None], 0, 1, "f64")[0]
    mag = np.abs(spec)
    peaks = set(np.argsort(mag)[-4:].tolist())
    assert peaks == {n // 8, n - n // 8, n // 4, n - n // 4}
    assert abs(mag[n // 8] - n / 2) < 1e-8 and abs(mag[n // 4] - n / 4) < 1e-8


@pytest.mark.parametrize("n", [64, 512, 2048])
def test_identities(oracle_lib, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))
    y = rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))
    F = lambda v, inv=0, reo=1: oa.ct_c2c(oracle_lib, v, inv, reo, "f64")  # noqa: E731
    # linearity
    np.testing.assert_allclose(F(2 * x - 3j * y), 2 * F(x) - 3j * F(y), atol=1e-9)
    # Parseval
    np.testing.assert_allclose((np.abs(F(x)) ** 2).sum(-1), n * (np.abs(x) ** 2).sum(-1), rtol=1e-12)
    # inv(fwd(x)) = N x ; FFT(FFT(x))[m] = N x[(-m) mod N]
    np.testing.assert_allclose(F(F(x), 1), n * x, atol=1e-9)
    np.testing.assert_allclose(F(F(x)), n * np.roll(x[:, ::-1], 1, axis=-1), atol=1e-9)
    # S2: no-reorder = FFT of the bit-reversed input
    br = ref.bitrev_indices(n)
    np.testing.assert_allclose(F(x, 0, 0), F(x[:, br]), atol=1e-9)
    np.testing.assert_allclose(F(x, 1, 0), F(x[:, br], 1), atol=1e-9)


@pytest.mark.parametrize("n", R2C_SIZES)
def test_r2c_packing_and_roundtrip(oracle_lib, n):
    rng = np.random.default_rng(n)
    x = rng.random((2, n))
    xp = oa.r2c(oracle_lib, x, "f64")
    full = np.fft.rfft(x, axis=-1)
    np.testing.assert_allclose(xp[:, 0].real, full[:, 0].real, atol=1e-10)
    np.testing.assert_allclose(xp[:, 0].imag, full[:, n // 2].real, atol=1e-10)
    np.testing.assert_allclose(xp[:, 1:], full[:, 1 : n // 2], atol=1e-10)
    # C2R(R2C(x)) = (N/2) x   (SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:613)
    np.testing.assert_allclose(oa.c2r(oracle_lib, xp, "f64"), (n / 2) * x, atol=1e-9)


# ------------------------------------------------------------- harness comparison metric restated
def test_get_error_semantics(oracle_lib):
    g = oracle_lib.oracle_get_error
    assert g(1.0, 1.00005) == pytest.approx(5e-5, rel=1e-2)
    assert g(-1.0, 1.0) == 0.0                      # signs are dropped (CT/FFT.c:26-27)
    assert g(523.0, 524.0) == pytest.approx(0.01)   # smaller > 10: scaled by 10^floor(log10)
    assert g(5.0, 1000.0) == pytest.approx(995.0)   # smaller <= 10: absolute


def test_compare_data_counts(oracle_lib):
    import ctypes
    a = np.zeros((2, 32), np.complex64)
    b = a.copy()
    b[1, 3] = 1e-3
    f = lambda v: v.view(np.float32).ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
    assert oracle_lib.oracle_compare_data(f(a), f(b), 32, 2, 1e-4, None, None) == 1
    assert oracle_lib.oracle_compare_data(f(a), f(a), 32, 2, 1e-4, None, None) == 0
