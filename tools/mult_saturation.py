"""FFT_multiple_benchmark at the README batch (4 GiB-equivalent nFFTs -> nFFTs/100 slots) and at
10x that batch (the kernel only touches the first nFFTs/100 FFTs, so the same 4 GiB buffers hold
it): shows how much of the in-LDS rate the README batch loses to launch quantization on 256 CUs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
TOTAL = 1 << 29
a, b = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, TOTAL * 8, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, chunk.nbytes)
for n in (32, 256, 1024, 4096):
    for reo in (True, False):
        row = []
        for mult in (1, 4, 10):
            nffts = TOTAL // n * mult
            for _ in range(2):
                sm.FFT_multiple_benchmark(a.ptr, b.ptr, n, nffts, False, reo)
            ts = sorted(sm.FFT_multiple_benchmark(a.ptr, b.ptr, n, nffts, False, reo)[1] for _ in range(7))
            done = (nffts // 400 * 400) if n == 32 else (nffts // 100 * 100)
            row.append(f"x{mult}: {ts[3]:.3f} ms {done / ts[3] * 1e3:.3e} FFT/s")
        print(f"N={n:5d} reorder={int(reo)}  " + " | ".join(row), flush=True)
