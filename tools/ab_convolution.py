"""A/B of builds of the reference-shaped convolution example (examples/reference_shape_kernel.hip) in one process:
every build_ab/libconv_<variant>.so (or name=path) is loaded, its contract-form and register-form kernels run on the config-2 pair,
timed round robin (so that drift of the box hits all alike) and compared with the first variant's output."""
import ctypes
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import smfft_amd as sm  # noqa: E402

N, NS = 1024, 524288
vp, ci = ctypes.c_void_p, ctypes.c_int
names = sys.argv[1:] or sorted(os.path.basename(p)[len("libconv_"):-3] for p in glob.glob(os.path.join(ROOT, "build_ab", "libconv_*.so")))
paths = {a.split("=")[0]: (os.path.abspath(a.split("=")[1]) if "=" in a else os.path.join(ROOT, "build_ab", f"libconv_{a}.so")) for a in names}
names = list(paths)
libs = {n: ctypes.CDLL(p) for n, p in paths.items()}
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(NS * N * 8, ctypes.byref(pa), ctypes.byref(pb)) == 0
chunk = (np.random.default_rng(0).random(1 << 22, dtype=np.float32) - 0.5)
sm.lib.smfft_memcpy_h2d(pa.value, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < NS * N * 8:
    step = min(filled, NS * N * 8 - filled)
    sm.lib.smfft_memcpy_d2d(pa.value + filled, pa.value, step)
    filled += step
h = np.zeros(N, np.complex128)
h[:5] = [0.4, 0.3, 0.2, 0.1, -0.05j]
H = sm.DeviceBuffer.from_host(np.fft.fft(h).astype(np.complex64))
SYMS = ("smfft_example_reference_shape_convolve_1024", "smfft_example_reference_shape_convolve_1024_registers")
head = np.empty(1 << 20, np.float32)
ref = {}
for sym in SYMS:
    for n in names:
        fn = getattr(libs[n], sym)
        fn.argtypes = [vp, vp, vp, ci, vp]
        assert fn(pa.value, H.ptr, pb.value, NS, None) == 0
        sm.lib.smfft_synchronize()
        sm.lib.smfft_memcpy_d2h(head.ctypes.data, pb.value + (NS * N * 8 - head.nbytes), head.nbytes)
        if sym not in ref:
            ref[sym] = head.copy()
        else:
            d = float(np.max(np.abs(head - ref[sym])))
            print(f"{sym[len('smfft_example_reference_shape_'):]} {n}: max |difference to {names[0]}| = {d:.3g}", flush=True)
for sym in SYMS:
    ts = {n: [] for n in names}
    for rep in range(13):
        for n in names:
            fn = getattr(libs[n], sym)
            if rep == 0:
                for _ in range(2):
                    fn(pa.value, H.ptr, pb.value, NS, None)
                sm.lib.smfft_synchronize()
            t0 = time.perf_counter()
            fn(pa.value, H.ptr, pb.value, NS, None)
            sm.lib.smfft_synchronize()
            ts[n].append((time.perf_counter() - t0) * 1e3)
    for n in names:
        v = sorted(ts[n])
        print(f"{sym[len('smfft_example_reference_shape_'):]:28s} {n:12s} median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}", flush=True)
