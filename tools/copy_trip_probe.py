"""Does an LDS round trip between the loads and the stores change the stream-copy rate?  (SMFFT_COPY_TRIPS
experiment knob of the calibration kernel; same placement-probed buffers for every variant.)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n = 1 << 29
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(n * 8, ctypes.byref(pa), ctypes.byref(pb)) == 0
sm.lib.smfft_memset(pa.value, 1, n * 8)
res = {}
for rnd in range(7):
    for trips in (0, 1, 2):
        os.environ["SMFFT_COPY_TRIPS"] = str(trips)
        sm.lib.smfft_copy_launch(pa.value, pb.value, n, None)
        sm.lib.smfft_synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            sm.lib.smfft_copy_launch(pa.value, pb.value, n, None)
        sm.lib.smfft_synchronize()
        res.setdefault(trips, []).append((time.perf_counter() - t0) / 10)
for trips, v in res.items():
    v.sort()
    print(f"copy variant {trips:2d}: median {v[len(v)//2]*1e3:.4f} ms  {2*n*8/v[len(v)//2]/1e9:.0f} GB/s   min {v[0]*1e3:.4f}")
