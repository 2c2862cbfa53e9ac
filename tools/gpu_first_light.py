"""First-light check on the GPU box: every transform, every size, error vs fp64 NumPy, printed as
a table (more informative than pytest -x while bringing a kernel up).  Exit code = failures."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402
from oracle import np_reference as ref  # noqa: E402

fails = 0
rng = np.random.default_rng(0)
for n in [32, 64, 128, 256, 512, 1024, 2048, 4096]:
    nffts = 3 * (4096 // n) + 1
    x = (rng.random((nffts, n), dtype=np.float32) + 1j * rng.random((nffts, n), dtype=np.float32)).astype(np.complex64)
    for inv in (0, 1):
        for reo in (1, 0):
            got = sm.c2c(x, bool(inv), bool(reo))
            l2, mx = ref.fft_errors(got, ref.ct_c2c(x, bool(inv), bool(reo)))
            ok = l2 <= ref.REL_L2_TOL and mx <= ref.MAX_ABS_TOL
            fails += not ok
            print(f"CT  N={n:5d} inv={inv} reorder={reo} relL2={l2:.2e} maxabs={mx:.2e} {'ok' if ok else 'FAIL'}")
for n in [512, 1024, 2048, 4096]:
    nffts = 2 * (4096 // (n // 2)) + 1
    x = rng.random((nffts, n), dtype=np.float32)
    l2, mx = ref.fft_errors(sm.r2c(x), ref.r2c_packed(x))
    ok = l2 <= ref.REL_L2_TOL and mx <= ref.MAX_ABS_TOL
    fails += not ok
    print(f"R2C N={n:5d} relL2={l2:.2e} maxabs={mx:.2e} {'ok' if ok else 'FAIL'}")
    xp = (rng.random((nffts, n // 2), dtype=np.float32) + 1j * rng.random((nffts, n // 2), dtype=np.float32)).astype(np.complex64)
    l2, mx = ref.fft_errors(sm.c2r(xp), ref.c2r_packed(xp))
    ok = l2 <= ref.REL_L2_TOL and mx <= ref.MAX_ABS_TOL
    fails += not ok
    print(f"C2R N={n:5d} relL2={l2:.2e} maxabs={mx:.2e} {'ok' if ok else 'FAIL'}")
print("failures:", fails)
sys.exit(min(fails, 100))
