"""In-LDS R2C / C2R path (`multiple`: 100 applications per slot in LDS) at the README batch (2 GiB of reals), several builds of
the library in one process, next to the C2C transform of the same complex length L = N/2 on the same number of slots (the
natural-order inverse Stockham program's `multiple` kernel): what the Hermitian split / merge costs on top of it.
    python tools/ab_rc_multiple.py [name=lib.so ...]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

libs = []
for a in sys.argv[1:]:
    name, path = a.split("=", 1)
    libs.append((name, ctypes.CDLL(os.path.abspath(path))))
if not libs:
    libs = [("product", sm.lib)]
vp, i, dp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
for _, l in libs:
    l.smfft_launch.argtypes = [i, i, vp, vp, i, i, i, i, vp]
TOTAL = 1 << 29
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
sm.lib.smfft_memset(A.ptr, 0, TOTAL * 8)


def med_launch(l, family, size, n, inv):
    ts = []
    for _ in range(11):
        sm.lib.smfft_synchronize()
        t0 = time.perf_counter()
        l.smfft_launch(family, 1, A.ptr, B.ptr, size, n, inv, 1, None)
        sm.lib.smfft_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts[3:])[3]


for rn in (512, 1024, 2048, 4096):
    n = TOTAL * 2 // rn
    cells = []
    for name, l in libs:
        c2c = med_launch(l, 1, rn // 2, n, 1)
        r2c = med_launch(l, 2, rn, n, 0)
        c2r = med_launch(l, 2, rn, n, 1)
        cells.append(f"{name}: C2C(L) {c2c:.4f} R2C {r2c:.4f} (+{(r2c / c2c - 1) * 100:.0f} %) C2R {c2r:.4f} (+{(c2r / c2c - 1) * 100:.0f} %)")
    print(f"real N={rn} ({n // 100} slots x 100, wall ms incl. launch): " + " | ".join(cells), flush=True)
