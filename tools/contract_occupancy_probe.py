import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
ex = ctypes.CDLL(os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
ex.smfft_example_reference_shape_ct_multiple.argtypes = [vp, vp, ci, ci, ci, vp]
TOTAL = 1 << 27
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(A.ptr, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < TOTAL * 8:
    step = min(filled, TOTAL * 8 - filled); sm.lib.smfft_memcpy_d2d(A.ptr + filled, A.ptr, step); filled += step
def once(fn, reps=7):
    for _ in range(10): fn()
    sm.lib.smfft_synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); sm.lib.smfft_synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
for n in (256, 1024, 4096):
    slots = 256 * 2048 // (n // 4)
    for reo in (1, 0):
        row = []
        for frac in (8, 4, 2, 1):
            blocks = slots // frac
            if blocks < 256: continue
            ms = once(lambda: ex.smfft_example_reference_shape_ct_multiple(A.ptr, B.ptr, n, blocks, reo, None))
            row.append(f"{blocks} blocks (1/{frac} of the slots) {ms:.4f} ms")
        print(f"N={n} reorder={reo}: " + " | ".join(row), flush=True)
