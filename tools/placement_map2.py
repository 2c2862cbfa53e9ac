"""Finer placement map: output offset in 1 GiB steps, 15 launches each (median), input at 0."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
nb = n * nffts * 8
G = 1 << 30
arena = sm.DeviceBuffer(80 * G)
base = arena.ptr
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, nb, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(base + off, chunk.ctypes.data, chunk.nbytes)
def t(i, o):
    sm.FFT_external_benchmark(i, o, n, nffts)
    return sorted(sm.FFT_external_benchmark(i, o, n, nffts)[1] for _ in range(15))[7]
for lo, hi in ((28, 48), (48, 68)):
    print(" ".join(f"{g}:{t(base, base + g * G):.3f}" for g in range(lo, hi)))
# sub-GiB: offsets 40 GiB + k*64 MiB
print(" ".join(f"40G+{k*64}M:{t(base, base + 40 * G + k * (64 << 20)):.3f}" for k in range(0, 16, 2)))
