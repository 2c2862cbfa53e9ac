"""One-off stress: the seeded random geometry sweep of tests/test_gpu_parity.py with many more cases and seeds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
from oracle import np_reference as ref
C2C = [32, 64, 128, 256, 512, 1024, 2048, 4096]; RC = [512, 1024, 2048, 4096]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(seed0)
worst = 0.0
for case in range(ncases):
    fam = ("ct", "ct", "st", "rc")[int(rng.integers(0, 4))]
    cap = int(rng.choice([0, 1, 2, 3, 5, 13, 64, 12288]))
    sm.lib.smfft_set_grid_cap(cap)
    if fam == "rc":
        n = int(rng.choice(RC)); nffts = int(rng.integers(1, 10 * (8192 // n) + 3))
        x = rng.random((nffts, n), dtype=np.float32) - 0.5
        got, want = sm.r2c(x), ref.r2c_packed(x)
        e1 = ref.fft_errors(got, want)
        xp = (rng.random((nffts, n // 2), dtype=np.float32) + 1j * rng.random((nffts, n // 2), dtype=np.float32)).astype(np.complex64)
        e2 = ref.fft_errors(sm.c2r(xp), ref.c2r_packed(xp))
        errs = (e1, e2)
    else:
        n = int(rng.choice(C2C)); nffts = int(rng.integers(1, 10 * (4096 // n) + 3))
        x = ((rng.random((nffts, n), dtype=np.float32) - 0.5) + 1j * (rng.random((nffts, n), dtype=np.float32) - 0.5)).astype(np.complex64)
        inv, reo = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if fam == "st":
            errs = (ref.fft_errors(sm.stockham_c2c(x, inverse=inv), ref.ct_c2c(x, inv, True)),)
        else:
            errs = (ref.fft_errors(sm.c2c(x, inv, reo), ref.ct_c2c(x, inv, reo)),)
    for rel, mx in errs:
        worst = max(worst, float(np.max(rel)))
        assert np.max(rel) <= ref.REL_L2_TOL and np.max(mx) <= ref.MAX_ABS_TOL, (case, fam, n, nffts, cap, float(np.max(rel)), float(np.max(mx)))
print(f"{ncases} random cases from seed {seed0}: all within tolerance; worst per-FFT relL2 {worst:.2e}")
