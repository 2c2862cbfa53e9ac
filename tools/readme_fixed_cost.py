"""README batch of the in-LDS path (5242 chains of 1024 elements) on the product schedule with 25 ... 400 applications per chain:
the intercept of time against applications is what a `multiple` launch pays whatever the applications.
    python tools/readme_fixed_cost.py [N ...]"""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smfft_amd as sm
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)
def med(fn, reps=15):
    for _ in range(3): fn(None)
    sp = ctypes.c_double(0)
    while sp.value < 30: fn(ctypes.byref(sp))
    ts = []
    for _ in range(reps):
        t = ctypes.c_double(0); fn(ctypes.byref(t)); ts.append(t.value)
    return sorted(ts)[len(ts)//2]
sizes = [int(x) for x in sys.argv[1:]] or [32, 64, 128, 256, 512, 1024, 2048, 4096]
for n in sizes:
    for reo in (1, 0):
        nffts = total // n
        ks, ms = [], []
        for k in (25, 50, 100, 200, 400):
            sm.lib.smfft_set_nreuses(k)
            ks.append(k); ms.append(med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, reo, t)))
        slope, icpt = np.polyfit(ks[1:], ms[1:], 1)
        print(f"N={n} reorder={reo} README batch, us by applications:", " ".join(f"{k}:{m*1e3:.1f}" for k, m in zip(ks, ms)),
              f"| per application {slope*1e3:.3f} us, intercept {icpt*1e3:.1f} us = {100*icpt/ms[2]:.1f} % of the 100-application launch", flush=True)
sm.lib.smfft_set_nreuses(0)
