"""One round of co-resident chains (reorder kernels, old schedule) with 1 ... 100 applications per chain: the intercept is what a
`multiple` launch pays whatever the applications (dispatch, the tile loads and stores of all workgroups at once, the event bracket)."""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)
def med(fn, reps=9):
    for _ in range(3): fn(None)
    sp = ctypes.c_double(0)
    while sp.value < 20: fn(ctypes.byref(sp))
    ts = []
    for _ in range(reps):
        t = ctypes.c_double(0); fn(ctypes.byref(t)); ts.append(t.value)
    return sorted(ts)[len(ts)//2]
sm.lib.smfft_set_multiple_balance(0)
for n, slots in ((4096, 1024), (1024, 4096), (256, 4096)):
    tile = max(1, 1024 // n)
    nffts = slots * tile * 100
    row = []
    for k in (1, 2, 4, 8, 16, 32, 100):
        sm.lib.smfft_set_nreuses(k)
        row.append((k, med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, 1, t))))
    print(f"N={n} one round of {slots} chains, ms by applications:", " ".join(f"{k}:{ms*1e3:.1f}us" for k, ms in row), flush=True)
sm.lib.smfft_set_nreuses(0)
