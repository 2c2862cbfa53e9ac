"""Generates tests/golden/smfft_golden.npz: seeded fp32 inputs and their fp64 NumPy outputs for
every transform semantic S1..S6 and every supported size.  Run from the repo root:

    python tests/golden/make_golden.py

The reference (CUDA + cuFFT) cannot run here and holds no vectors of its own, so these fixtures are
generated from numpy.fft in complex128 -- the double-precision reference north_star names.
Inputs follow the reference harness distribution U[0,1) (SMFFT_CooleyTukey_C2C/FFT.c:139-143) plus
a zero-mean U[-1,1) set; PCG64 with a fixed seed instead of srand(time(NULL)).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import np_reference as ref  # noqa: E402

SEED = 20200720
NFFT = 9  # FFTs per size: eight plus one, so that every kernel shape (4096-element tiles, 1024-element waves) ends in a ragged tail
C2C_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]
R2C_SIZES = [512, 1024, 2048, 4096]


def main():
    rng = np.random.Generator(np.random.PCG64(SEED))
    out = {}
    for n in C2C_SIZES:
        for dist in ("u01", "u11"):
            re = rng.random((NFFT, n), dtype=np.float32)
            im = rng.random((NFFT, n), dtype=np.float32)
            if dist == "u11":
                re, im = 2 * re - 1, 2 * im - 1
            x = (re + 1j * im).astype(np.complex64)
            out[f"c2c_in_{dist}_{n}"] = x
            for inv in (0, 1):
                for reo in (0, 1):
                    if dist == "u11" and (inv, reo) != (0, 1):
                        continue
                    out[f"ct_out_{dist}_{n}_inv{inv}_reo{reo}"] = ref.ct_c2c(x, bool(inv), bool(reo))
    for n in R2C_SIZES:
        x = rng.random((NFFT, n), dtype=np.float32)
        out[f"r2c_in_{n}"] = x
        out[f"r2c_out_{n}"] = ref.r2c_packed(x)
        xp = (rng.random((NFFT, n // 2), dtype=np.float32) + 1j * rng.random((NFFT, n // 2), dtype=np.float32)).astype(np.complex64)
        out[f"c2r_in_{n}"] = xp
        out[f"c2r_out_{n}"] = ref.c2r_packed(xp)
    path = os.path.join(os.path.dirname(__file__), "smfft_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
