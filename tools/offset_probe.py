"""Does the relative placement of the input and output buffers matter for the external kernel?
(HBM channel / bank conflicts between the read and the write stream.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

n, nffts = 1024, 524288
nbytes = n * nffts * 8
slack = 64 << 20
a, b = sm.DeviceBuffer(nbytes + slack), sm.DeviceBuffer(nbytes + slack)
rng = np.random.default_rng(0)
chunk = rng.random(1 << 22, dtype=np.float32)
for off in range(0, nbytes, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, min(chunk.nbytes, nbytes - off))
print(f"in  {a.ptr:#x}  out {b.ptr:#x}  delta {(b.ptr - a.ptr) % (1 << 32):#x}")
for off in [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (4 << 20) + 65536, 16 << 20, (32 << 20) + 8192]:
    ts = []
    for _ in range(3):
        sm.FFT_external_benchmark(a.ptr, b.ptr + off, n, nffts)
    for _ in range(15):
        rc, ms = sm.FFT_external_benchmark(a.ptr, b.ptr + off, n, nffts)
        ts.append(ms)
    ts.sort()
    print(f"out offset {off:>10d}: median {ts[7]:.4f} ms ({2 * nbytes / ts[7] / 1e6:.1f} GB/s) min {ts[0]:.4f}")
