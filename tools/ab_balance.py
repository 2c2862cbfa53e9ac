#!/usr/bin/env python3
"""In-LDS `multiple` path at the README batch (2^29 / N FFTs): one chain per workgroup (balance 0, rounds 1-3) against the
balanced schedule (balance 1: a persistent grid of the co-resident workgroups shares the applications evenly), and the kernel
without cross-application fusion (path 2); a saturating batch (8 x the slots) for reference.  One process, one box.
    python tools/ab_balance.py [N ...] > profiles/r04_ab_balance.txt"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smfft_amd as sm  # noqa: E402

PEAK = 157.3e12


def median_ms(fn, reps=9, settle_ms=40.0):
    for _ in range(3):
        fn(None)
    spent = ctypes.c_double(0.0)
    while spent.value < settle_ms:
        fn(ctypes.byref(spent))
    ts = []
    for _ in range(reps):
        t = ctypes.c_double(0.0)
        fn(ctypes.byref(t))
        ts.append(t.value)
    return sorted(ts)[len(ts) // 2]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 512, 1024, 2048, 4096]
    total = 1 << 29
    a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
    sm.lib.smfft_memset(a.ptr, 0, total * 8)
    sm.FFT_init()
    print(f"{'N':>5} {'order':>9} {'unbalanced ms':>13} {'frac':>6} {'balanced ms':>11} {'frac':>6} {'gain':>6} {'saturating frac':>15} {'unfused bal. ms':>15} {'frac':>6}")
    for n in sizes:
        bn = total // n
        done = (bn // 400 * 400) if n == 32 else (bn // 200 * 200) if n == 64 else (bn // 100 * 100)
        flops = done * 5 * n * math.log2(n)
        for reo in (1, 0):
            row = {}
            for bal in (0, 1):
                sm.lib.smfft_set_multiple_balance(bal)
                row[bal] = median_ms(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, bn, 0, reo, t))
            sm.lib.smfft_set_multiple_balance(1)
            sat = median_ms(lambda t: sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, 8 * bn, 0, reo, t), reps=5)
            unf = None
            if reo and n >= 64:
                unf = median_ms(lambda t: sm.lib.smfft_ct_multiple_unfused_benchmark(a.ptr, b.ptr, n, bn, 0, t))
            f = lambda ms: flops / (ms * 1e-3) / PEAK
            print(f"{n:5d} {'reorder' if reo else 'noreorder':>9} {row[0]:13.4f} {f(row[0]):6.3f} {row[1]:11.4f} {f(row[1]):6.3f} {row[0] / row[1]:6.3f} "
                  f"{8 * flops / (sat * 1e-3) / PEAK:15.3f} " + (f"{unf:15.4f} {f(unf):6.3f}" if unf else f"{'-':>15} {'-':>6}"), flush=True)
    sm.lib.smfft_set_multiple_balance(-1)


if __name__ == "__main__":
    main()
