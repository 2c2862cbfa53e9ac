"""The library use case end to end (reference README.md:10-16): batched circular convolution y = IFFT(FFT(x) .* H) / N of
524288 series of 1024 points (4 GiB in, 4 GiB out) inside ONE user kernel, three ways -- on the reference's contract
(blockDim = N/4, examples/reference_shape_kernel.hip), on the engine's tiled contract and on its register-level interface
(examples/fft_convolution.hip) -- next to one external FFT launch and the same-shape copy on the same pair of buffers."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

N, NS = 1024, 524288
ex = ctypes.CDLL(os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
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


def timed(fn, reps=11, warm=25):   # 25 launches = 35...60 ms: the clocks have settled (profiles/r03_warm_ramp.txt)
    import time
    for _ in range(warm):
        fn()
    sm.lib.smfft_synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sm.lib.smfft_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


gb = 2 * NS * N * 8 / 1e9
for name, sym in (("reference contract (blockDim = N/4)", "smfft_example_reference_shape_convolve_1024"),
                  ("reference thread shape, register form", "smfft_example_reference_shape_convolve_1024_registers"),
                  ("tiled device functions", "smfft_example_convolve_1024"),
                  ("register-level engine", "smfft_example_convolve_1024_registers")):
    fn = getattr(ex, sym)
    fn.argtypes = [vp, vp, vp, ci, vp]
    ms = timed(lambda: fn(pa.value, H.ptr, pb.value, NS, None))
    print(f"convolution, {name}: {ms:.3f} ms = {gb / ms:.2f} TB/s of input + output", flush=True)
t = ctypes.c_double(0)
ts = []
for k in range(9):
    t.value = 0
    sm.lib.smfft_ct_external_benchmark(pa.value, pb.value, N, NS, 0, 1, ctypes.byref(t))
    ts.append(t.value)
print(f"one external FFT launch on the same pair: {sorted(ts)[4]:.3f} ms; same-shape copy: {timed(lambda: sm.lib.smfft_copy_launch(pa.value, pb.value, NS * N, None)):.3f} ms")
