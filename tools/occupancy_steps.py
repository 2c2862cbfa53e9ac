"""In-LDS path: time of one launch as a function of the number of compact workgroups resident per SIMD, each
transforming its slots 100 times.  While the workgroups all fit on the chip together the time grows with the per-SIMD
load only; the first step up shows how many waves per SIMD really are resident, and the slope what a SIMD does per FFT.
Several builds of the library can be compared in one process (same box, same clocks):
    python tools/occupancy_steps.py [N] [name=lib.so ...] [--reorder 0,1] [--steps 4,8,16,...]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 1024
reorders, steps, libs = (1, 0), (4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160), []
while args:
    a = args.pop(0)
    if a == "--reorder":
        reorders = tuple(int(v) for v in args.pop(0).split(","))
    elif a == "--steps":
        steps = tuple(int(v) for v in args.pop(0).split(","))
    else:
        name, path = a.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.smfft_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        libs.append((name, lib))
if not libs:
    libs = [("product", sm.lib)]
TOTAL = 1 << 27
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
sm.lib.smfft_memset(A.ptr, 0, TOTAL * 8)
ffts_per_wave = max(1, 1024 // n)
waves_per_group = max(1, n // 1024)          # N = 2048 / 4096: one FFT per workgroup of 2 / 4 waves
for reo in reorders:
    print(f"N={n} reorder={reo}: waves per SIMD -> ms per launch, ns per FFT per SIMD  [" + " | ".join(name for name, _ in libs) + "]")
    for k16 in steps:
        waves = 64 * k16                      # k16 / 16 waves per SIMD
        slots = waves * ffts_per_wave // waves_per_group
        cells = []
        for name, lib in libs:
            ts = []
            for _ in range(7):
                sm.lib.smfft_synchronize()
                t0 = time.perf_counter()
                lib.smfft_launch(0, 1, A.ptr, B.ptr, n, slots * 100, 0, reo, None)
                sm.lib.smfft_synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(ts)[2]
            cells.append(f"{ms:.4f} ms {ms * 1e6 / (100 * ffts_per_wave / waves_per_group * k16 / 16):7.1f} ns")
        print(f"  {k16 / 16:5.2f} waves/SIMD ({waves:6d} waves): " + " | ".join(cells), flush=True)
