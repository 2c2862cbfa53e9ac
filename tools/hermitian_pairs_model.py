"""NumPy model of the pair-wise Hermitian split / merge of the in-LDS R2C / C2R kernels (include/smfft/smfft_planar.hpp,
PlanarEngine::hermitian_apply_pairs): who holds what, who fetches what, which rows the second results travel through -- replayed
thread by thread and register by register and compared with numpy.fft.rfft / irfft in the reference's packed layout (RC:269-344:
element 0 = (DC, Nyquist)).  CPU only; tests/test_planar_layout_model.py runs check().

Layout: the complex transform of length L lives on T = L / 16 threads; the thread with role k holds r[q] = x[k + T q], q = 0..15.
Element i pairs with L - i = (T - k) + T (15 - q): register 15 - q of the thread with role T - k (role 0: its OWN register 16 - q).
  1  every thread exposes its registers 8..15 in rows 8..15 of the image (lane linear: row j, the thread's dword)
  2  pair q = 0..7 of a thread: A = r[q], B = row 15 - q at the PARTNER's dword (role 0: its own registers);
         out = S/2 + V D,  other = conj(S - out),  S = A + conj(B), D = A - conj(B), V = (-+i/2) W_2L^i;   r[q] = out
  3  row j <- other[15 - j] (role 0: other[16 - j], row 8 <- conj(r[8])), lane linear again
  4  r[j] <- row j at the partner's dword, j = 8..15."""
import numpy as np


def split_merge_pairs(r, L, inverse):
    """r[k][q]: registers of the thread with role k -> the same after the split (R2C, inverse = False) or merge (C2R)"""
    T = L // 16
    sign = 1.0 if inverse else -1.0
    rows = {j: [r[k][j] for k in range(T)] for j in range(8, 16)}                  # step 1
    other = [[0j] * 8 for _ in range(T)]
    out = [list(rk) for rk in r]
    for k in range(T):
        partner = (T - k) % T
        for q in range(8):
            A = r[k][q]
            if k == 0:
                B = r[0][8] if q == 0 else r[0][16 - q]
            else:
                B = rows[15 - q][partner]
            S, D = A + np.conj(B), A - np.conj(B)
            i = k + T * q
            V = (sign * 0.5j) * np.exp(sign * 2j * np.pi * i / (2 * L))
            o = 0.5 * S + V * D
            other[k][q] = np.conj(S - o)
            if k == 0 and q == 0:
                o = 0.5 * complex(A.real + A.imag, A.real - A.imag) if inverse else complex(A.real + A.imag, A.real - A.imag)
                other[k][q] = np.conj(r[0][8])
            out[k][q] = o
    new_rows = {}
    for j in range(8, 16):                                                          # step 3
        new_rows[j] = [(other[k][0] if j == 8 else other[k][16 - j]) if k == 0 else other[k][15 - j] for k in range(T)]
    for k in range(T):                                                              # step 4
        for j in range(8, 16):
            out[k][j] = new_rows[j][(T - k) % T]
    return out


def to_registers(x, L):
    T = L // 16
    return [[x[k + T * q] for q in range(16)] for k in range(T)]


def from_registers(r, L):
    T = L // 16
    x = np.zeros(L, complex)
    for k in range(T):
        for q in range(16):
            x[k + T * q] = r[k][q]
    return x


def check():
    """max relative error of split(FFT_L(z)) against rfft in the packed layout and of the merge against its inverse, L = 256 ... 2048"""
    rng = np.random.default_rng(1)
    worst = 0.0
    for L in (256, 512, 1024, 2048):
        xr = rng.standard_normal(2 * L)
        z = xr[0::2] + 1j * xr[1::2]                         # the real input read as L float2 (RC:406)
        X = np.fft.rfft(xr)
        want = X[:L].copy()
        want[0] = complex(X[0].real, X[L].real)
        got = from_registers(split_merge_pairs(to_registers(np.fft.fft(z), L), L, False), L)
        worst = max(worst, np.abs(got - want).max() / np.abs(want).max())
        # C2R: the merge of the packed spectrum followed by the inverse complex transform gives (N/2) * x as L float2 (S6)
        merged = from_registers(split_merge_pairs(to_registers(want, L), L, True), L)
        back = np.fft.ifft(merged) * L
        worst = max(worst, np.abs(back - L * z).max() / np.abs(L * z).max())
    return worst


if __name__ == "__main__":
    print("worst relative error", check())
