"""Where the balanced persistent grid stops paying: in-LDS reorder kernels with R rounds' worth of chains (R = chains / co-resident
workgroups), the balanced schedule with rotating priorities (forced: balance = the slot count) against one chain per workgroup in
the arbiter's own order.  The launcher switches at R = 4 (smfft_inst.hip, launch_compact)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smfft_amd as sm  # noqa: E402

total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)


def med(fn, reps=7):
    for _ in range(3):
        fn(None)
    sp = ctypes.c_double(0)
    while sp.value < 40:
        fn(ctypes.byref(sp))
    ts = []
    for _ in range(reps):
        t = ctypes.c_double(0)
        fn(ctypes.byref(t))
        ts.append(t.value)
    return sorted(ts)[len(ts) // 2]


sizes = [int(v) for v in sys.argv[1:]] or [32, 64, 256, 1024, 4096]
for n in sizes:
  for reo in (1, 0):
    slots = ctypes.c_int(0)
    sm.lib.smfft_measure_multiple_residency(0, n, 0, reo, 1, ctypes.byref(slots))
    slots = slots.value
    tile = max(1, 1024 // n)
    for rounds in (1.28, 2.3, 4.3, 6.3, 8.37):
        ntiles = int(rounds * slots)
        nffts = ntiles * tile * 100
        sm.lib.smfft_set_multiple_balance(slots)
        sm.lib.smfft_set_multiple_rotation(15)
        bal = med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, reo, t))
        sm.lib.smfft_set_multiple_balance(0)
        sm.lib.smfft_set_multiple_rotation(0)
        old = med(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, reo, t))
        print(f"N={n} reorder={reo} {slots} slots, {rounds:5.2f} rounds ({ntiles} chains): balanced + rotation {bal:.4f} ms | one chain per workgroup, oldest first {old:.4f} ms | ratio {old / bal:.3f}", flush=True)
sm.lib.smfft_set_multiple_balance(-1)
sm.lib.smfft_set_multiple_rotation(-1)
