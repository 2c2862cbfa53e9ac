"""External C2C N=1024 at 1, 2, 4 GiB per direction and R2C/C2R real N=2048 at 1, 2 GiB on one smfft_malloc_pair pair:
how much of the R2C/C2R gap to config 2 is the shorter launch (fixed ramp + tail) rather than the kernel."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

TOTAL = 1 << 29
nbytes = TOTAL * 8
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
print("pair:", sm.last_pair_info())
a, b = pa.value, pb.value
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(a, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(a + filled, a, step)
    filled += step


def med(call, rounds=11):
    ts = []
    for r in range(rounds + 2):
        t = ctypes.c_double(0)
        assert call(ctypes.byref(t)) == 0
        if r >= 2:
            ts.append(t.value)
    return sorted(ts)[len(ts) // 2]


for rep in range(2):
    for gib, off in ((4, 0), (2, 0), (2, 2), (1, 0), (1, 1), (1, 2), (1, 3)):
        n = gib * (1 << 27) // 1024
        o = off * (1 << 30)
        ms = med(lambda t: sm.lib.smfft_ct_external_benchmark(a + o, b + o, 1024, n, 0, 1, t))
        print(f"C2C N=1024 {gib} GiB at +{off} GiB: {ms:.4f} ms  {2 * gib * (1 << 30) / ms / 1e6:.0f} GB/s  frac {2 * gib * (1 << 30) / ms / 1e6 / 8000:.3f}")
    for gib, off in ((2, 0), (2, 2), (1, 0), (1, 2)):
        n = gib * (1 << 28) // 2048
        o = off * (1 << 30)
        for inv in (0, 1):
            ms = med(lambda t: sm.lib.smfft_rc_external_benchmark(a + o, b + o, 2048, n, inv, t))
            print(f"{'C2R' if inv else 'R2C'} real N=2048 {gib} GiB at +{off} GiB: {ms:.4f} ms  {2 * gib * (1 << 30) / ms / 1e6:.0f} GB/s  frac {2 * gib * (1 << 30) / ms / 1e6 / 8000:.3f}")
