"""In-LDS path, N = 1024: time of one launch as a function of the number of single-wave workgroups (k per SIMD, 1024 SIMDs),
each transforming its slot 100 times.  While the waves all fit on the chip together the time grows with the per-SIMD load
only; the first step up shows how many waves per SIMD really are resident, and the slope what a SIMD does per FFT.
    python tools/occupancy_steps.py [N]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
TOTAL = 1 << 27
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
sm.lib.smfft_memset(A.ptr, 0, TOTAL * 8)
ffts_per_wave = max(1, 1024 // n)
for reo in (1, 0):
    print(f"N={n} reorder={reo}: waves per SIMD -> ms per launch, ns per FFT per SIMD")
    for k16 in (4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160):
        waves = 64 * k16                      # k16 / 16 waves per SIMD
        slots = waves * ffts_per_wave
        ts = []
        for _ in range(7):
            sm.lib.smfft_synchronize()
            t0 = time.perf_counter()
            sm.lib.smfft_launch(0, 1, A.ptr, B.ptr, n, slots * 100, 0, reo, None)
            sm.lib.smfft_synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ts)[2]
        print(f"  {k16 / 16:5.2f} waves/SIMD ({waves:6d} waves): {ms:.4f} ms   {ms * 1e6 / (100 * ffts_per_wave * k16 / 16):.1f} ns per FFT per SIMD", flush=True)
