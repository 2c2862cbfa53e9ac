"""Copy-ceiling probe: the stream-copy calibration kernel over grid caps (and tile mappings via
SMFFT_COPY_CONTIGUOUS=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n = 1 << 29
a, b = sm.DeviceBuffer(n * 8), sm.DeviceBuffer(n * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, n * 8, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, chunk.nbytes)
for cap in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1024,2048,4096,8192,12288,16384,24576,32768,65536,0").split(",")]:
    sm.lib.smfft_set_grid_cap(cap)
    for _ in range(3):
        sm.lib.smfft_copy_launch(a.ptr, b.ptr, n, None)
    sm.lib.smfft_synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            sm.lib.smfft_copy_launch(a.ptr, b.ptr, n, None)
        sm.lib.smfft_synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"cap {cap:6d}: {best*1e3:.4f} ms  {2*n*8/best/1e9:.1f} GB/s", flush=True)
