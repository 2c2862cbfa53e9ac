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
for n, slots in ((4096, 1024), (1024, 4864), (2048, 2048)):
    tile = max(1, 1024 // n)
    for label, ntiles in (("8 x slots (aligned)", 8 * slots), ("8 x slots + 37%", int(8.37 * slots)), ("README", (total // n) // 100 // tile), ("1 x slots", slots), ("2 x slots", 2 * slots), ("1.5 x slots", slots * 3 // 2)):
        nffts = ntiles * tile * 100
        row = []
        for bal in (0, 1, slots // 2, slots * 3 // 4):
            sm.lib.smfft_set_multiple_balance(bal)
            row.append(med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, 1, t)))
        print(f"N={n} {label:22s} chains {ntiles:6d}: unbalanced {row[0]:.4f} ms | balanced {row[1]:.4f} | G=slots/2 {row[2]:.4f} | G=3/4 slots {row[3]:.4f}   per chain-round (unbalanced) {row[0] / (ntiles / slots):.4f}", flush=True)
