"""Host-side cost of one launch through the C ABI at a batch so small that nothing else matters (nFFTs = 4): the time per
smfft_launch call (asynchronous, 2000 calls, one synchronise at the end) and per FFT_external_benchmark call (event-timed,
synchronous), with 0 and with 64 built outputs alive (the per-launch pacing lookup walks... nothing: it reads a sorted
snapshot).  Several builds can be compared:  python tools/launch_latency.py [name=lib.so ...]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

libs = []
for a in sys.argv[1:]:
    name, path = a.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i, dp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
    lib.smfft_launch.argtypes = [i, i, vp, vp, i, i, i, i, vp]
    lib.smfft_ct_external_benchmark.argtypes = [vp, vp, i, i, i, i, dp]
    lib.smfft_malloc_written.argtypes = [ctypes.c_ulonglong, ctypes.POINTER(vp)]
    lib.smfft_free_written.argtypes = [vp]
    libs.append((name, lib))
if not libs:
    libs = [("product", sm.lib)]
A, B = sm.DeviceBuffer(1 << 20), sm.DeviceBuffer(1 << 20)
for name, lib in libs:
    for alive in (0, 64):
        outs = []
        for _ in range(alive):
            w = ctypes.c_void_p()
            assert lib.smfft_malloc_written(256 << 20, ctypes.byref(w)) == 0
            outs.append(w)
        for _ in range(200):
            lib.smfft_launch(0, 0, A.ptr, B.ptr, 1024, 4, 0, 1, None)
        sm.lib.smfft_synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            lib.smfft_launch(0, 0, A.ptr, B.ptr, 1024, 4, 0, 1, None)
        sm.lib.smfft_synchronize()
        t_async = (time.perf_counter() - t0) / 2000 * 1e6
        t = ctypes.c_double(0)
        t0 = time.perf_counter()
        for _ in range(500):
            lib.smfft_ct_external_benchmark(A.ptr, B.ptr, 1024, 4, 0, 1, ctypes.byref(t))
        t_sync = (time.perf_counter() - t0) / 500 * 1e6
        print(f"{name}: {alive:2d} built outputs alive: smfft_launch {t_async:.2f} us per call; FFT_external_benchmark(nFFTs = 4) {t_sync:.1f} us per call (kernel {t.value / 500 * 1e3:.1f} us)", flush=True)
        for w in outs:
            lib.smfft_free_written(w)
