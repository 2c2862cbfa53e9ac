"""Times the two example convolution kernels (LDS-contract device function vs register-level
engine interface) on 524288 series of 1024 points (4 GiB in, 4 GiB out)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
ex = ctypes.CDLL(os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"))
n, ns = 1024, 524288
a, b = sm.DeviceBuffer(n * ns * 8), sm.DeviceBuffer(n * ns * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, n * ns * 8, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, chunk.nbytes)
H = sm.DeviceBuffer.from_host(np.fft.fft(np.r_[0.5, 0.3, 0.2, np.zeros(n - 3)]).astype(np.complex64))
for sym in ("smfft_example_convolve_1024", "smfft_example_convolve_1024_registers"):
    fn = getattr(ex, sym)
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    for _ in range(3):
        fn(a.ptr, H.ptr, b.ptr, ns, None)
    sm.lib.smfft_synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn(a.ptr, H.ptr, b.ptr, ns, None)
    sm.lib.smfft_synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{sym:40s} {ms:.3f} ms  {2 * n * ns * 8 / ms / 1e6:.0f} GB/s  {ns / ms * 1e3:.3e} convolutions/s")
