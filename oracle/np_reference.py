"""fp64 NumPy statement of the six transform semantics S1..S6.

TEST INFRASTRUCTURE ONLY (see oracle/smfft_oracle.c): used by tests/ and by
tests/golden/make_golden.py to pin the C restatement and the HIP path.  Everything is evaluated
in complex128 with numpy.fft, the double-precision reference north_star names.

Reference citations (relative to the reference checkout):
  S1/S2  SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu:334-532  (do_SMFFT_CT_DIT, fft_reorder 1 / 0)
  S3     SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu:70-78,97-240 (sign +, un-normalised)
  S4     SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:106-266
  S5/S6  SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:269-344
"""
import numpy as np


def bitrev_indices(n: int) -> np.ndarray:
    bits = n.bit_length() - 1
    idx = np.arange(n)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(bits):
        rev |= ((idx >> b) & 1) << (bits - 1 - b)
    return rev


def ct_c2c(x: np.ndarray, inverse: bool, reorder: bool) -> np.ndarray:
    """x: (nFFTs, N) complex.  S1 (reorder) / S2 (no reorder).  Un-normalised both directions."""
    x = np.asarray(x, dtype=np.complex128)
    n = x.shape[-1]
    if not reorder:
        x = x[..., bitrev_indices(n)]
    return np.fft.ifft(x, axis=-1) * n if inverse else np.fft.fft(x, axis=-1)


def st_c2c(x: np.ndarray, inverse: bool = True) -> np.ndarray:
    """S3 (inverse=True, the ST program) / S4 (either direction).  Natural order, un-normalised."""
    return ct_c2c(x, inverse, True)


def r2c_packed(x: np.ndarray) -> np.ndarray:
    """S5: x (nFFTs, N) real -> (nFFTs, N/2) complex, element 0 = (X[0].re, X[N/2].re)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    full = np.fft.rfft(x, axis=-1)
    out = full[..., : n // 2].copy()
    out[..., 0] = full[..., 0].real + 1j * full[..., n // 2].real
    return out


def c2r_packed(xp: np.ndarray) -> np.ndarray:
    """S6: packed (nFFTs, N/2) complex -> (nFFTs, N) real = (N/2) * irfft-normalised signal."""
    xp = np.asarray(xp, dtype=np.complex128)
    half = xp.shape[-1]
    n = 2 * half
    full = np.zeros(xp.shape[:-1] + (half + 1,), dtype=np.complex128)
    full[..., :half] = xp
    full[..., 0] = xp[..., 0].real
    full[..., half] = xp[..., 0].imag
    return np.fft.irfft(full, n=n, axis=-1) * (n / 2)


# ---- the stated fp32 tolerance -------------------------
REL_L2_TOL = 5e-7      # per FFT: ||y - ref||_2 / ||ref||_2
MAX_ABS_TOL = 1e-6     # per FFT: max|y - ref| / max|ref|


def fft_errors(got: np.ndarray, ref: np.ndarray):
    """Worst per-FFT relative-L2 and max-abs/max-magnitude errors of got vs the fp64 reference."""
    got = np.asarray(got).astype(ref.dtype if np.iscomplexobj(ref) else np.float64)
    d = got - ref
    ax = -1
    l2 = np.sqrt((np.abs(d) ** 2).sum(axis=ax)) / np.maximum(np.sqrt((np.abs(ref) ** 2).sum(axis=ax)), 1e-300)
    mx = np.abs(d).max(axis=ax) / np.maximum(np.abs(ref).max(axis=ax), 1e-300)
    return float(l2.max()), float(mx.max())


def assert_close_fp32(got, ref, what=""):
    l2, mx = fft_errors(got, ref)
    assert l2 <= REL_L2_TOL and mx <= MAX_ABS_TOL, f"{what}: relL2={l2:.3e} (tol {REL_L2_TOL}), maxabs={mx:.3e} (tol {MAX_ABS_TOL})"
    return l2, mx
