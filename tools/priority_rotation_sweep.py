"""In-LDS reorder kernels, one round of co-resident chains / the README batch / a saturating batch, one chain per workgroup
(balance 0) against the balanced schedule, for the rotation period of the wave priorities given by SMFFT_PRIO_ROTATE (log2 of
shader clocks, 0 = off):   for k in 0 9 11 13 15 17 19; do SMFFT_PRIO_ROTATE=$k python tools/priority_rotation_sweep.py; done"""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)
def med(fn, reps=7):
    for _ in range(3): fn(None)
    sp = ctypes.c_double(0)
    while sp.value < 40: fn(ctypes.byref(sp))
    ts = []
    for _ in range(reps):
        t = ctypes.c_double(0); fn(ctypes.byref(t)); ts.append(t.value)
    return sorted(ts)[len(ts)//2]
print("SMFFT_PRIO_ROTATE =", os.environ.get("SMFFT_PRIO_ROTATE"))
for n, slots in ((4096, 1024), (1024, 4096), (2048, 2048), (256, 4096)):
    tile = max(1, 1024 // n)
    for label, ntiles in (("1 x slots", slots), ("README", (total // n) // 100 // tile), ("8 x slots + 37%", int(8.37 * slots))):
        nffts = ntiles * tile * 100
        row = []
        for bal in (0, 1):
            sm.lib.smfft_set_multiple_balance(bal)
            row.append(med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, 1, t)))
        print(f"N={n} {label:18s} chains {ntiles:6d}: unbalanced {row[0]:.4f} ms | balanced {row[1]:.4f}", flush=True)
