"""Isolated (one event-timed, synchronised launch at a time: FFT_external_benchmark, what the reference
times) vs sustained (K launches back to back between one pair of events) duration of the
N=1024 external kernel, same buffers, same process."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
a, b = sm.DeviceBuffer(n * nffts * 8), sm.DeviceBuffer(n * nffts * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, n * nffts * 8, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, chunk.nbytes)
for _ in range(5):
    sm.FFT_external_benchmark(a.ptr, b.ptr, n, nffts)
for rep in range(3):
    iso = sorted(sm.FFT_external_benchmark(a.ptr, b.ptr, n, nffts)[1] for _ in range(20))
    sm.lib.smfft_synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sm.launch("ct", "external", a.ptr, b.ptr, n, nffts)
    sm.lib.smfft_synchronize()
    sus = (time.perf_counter() - t0) / 50 * 1e3
    print(f"isolated median {iso[10]:.4f} ms ({2*n*nffts*8/iso[10]/1e6:.0f} GB/s)   sustained {sus:.4f} ms ({2*n*nffts*8/sus/1e6:.0f} GB/s)")
