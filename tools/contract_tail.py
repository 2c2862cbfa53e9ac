"""What upstream's launch shape costs the in-LDS contract path at the README batch: SMFFT_DIT_multiple<P> in the reference's shape
(one block of fft_length / 4 threads per transform, CT:669-683) is a grid of equal, short-lived blocks; the README batches are 2.56 rounds of
the blocks a device holds at every length >= 256.  Timed here: the README grid, and grids of exactly 2 and 3 rounds (per-transform time)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

ex = ctypes.CDLL(os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
ex.smfft_example_reference_shape_ct_multiple.argtypes = [vp, vp, ci, ci, ci, vp]
TOTAL = 1 << 29
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(A.ptr, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < TOTAL * 8:
    step = min(filled, TOTAL * 8 - filled)
    sm.lib.smfft_memcpy_d2d(A.ptr + filled, A.ptr, step)
    filled += step


def once(fn, reps=5):
    for _ in range(3):
        fn()
    sm.lib.smfft_synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sm.lib.smfft_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


CUS = 256
for n in (256, 512, 1024, 2048, 4096):
    slots = CUS * 2048 // (n // 4)          # blocks of N/4 threads a device of 256 CUs x 2048 threads holds
    readme = (TOTAL // n) // 100
    for reo in (1, 0):
        row = []
        for blocks in (readme, 2 * slots, 3 * slots):
            ms = once(lambda: ex.smfft_example_reference_shape_ct_multiple(A.ptr, B.ptr, n, blocks, reo, None))
            row.append(f"{blocks} blocks ({blocks / slots:.2f} rounds) {ms:.4f} ms = {ms * 1e6 / (blocks * 100):.3f} ns per transform")
        print(f"N={n} reorder={reo}: " + " | ".join(row), flush=True)
