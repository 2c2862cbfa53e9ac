"""The device functions in the REFERENCE'S OWN contract (blockDim.x = fft_length / 4, data contiguous in s[0 .. fft_length),
every thread of the block calls): the two-argument kernels of include/smfft/smfft_device_functions.hpp, launched in the
reference's shape, timed next to the library's tiled / compact kernels on the same buffers.
    python tools/reference_contract.py [--sizes 1024] [--plain]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024")
ap.add_argument("--plain", action="store_true")
ap.add_argument("--examples", default=os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"), help="another build of examples/ to time")
ap.add_argument("--in-lds-only", action="store_true")
args = ap.parse_args()
ex = ctypes.CDLL(os.path.abspath(args.examples))
vp, i = ctypes.c_void_p, ctypes.c_int
ex.smfft_example_reference_shape_ct.argtypes = [vp, vp, i, i, i, i, i, vp]
ex.smfft_example_reference_shape_st.argtypes = [vp, vp, i, i, vp]
ex.smfft_example_reference_shape_rc.argtypes = [vp, vp, i, i, i, vp]
ex.smfft_example_reference_shape_multiple_one.argtypes = [vp, vp, i, i, vp]
ex.smfft_example_reference_shape_ct_multiple.argtypes = [vp, vp, i, i, i, vp]
has_wave64 = hasattr(ex, "smfft_example_reference_shape_ct_multiple_wave64")
if has_wave64:
    ex.smfft_example_reference_shape_ct_multiple_wave64.argtypes = [vp, vp, i, i, i, vp]

TOTAL = 1 << 29
nbytes = TOTAL * 8
if args.plain:
    A, B = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
    a, b = A.ptr, B.ptr
else:
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
    a, b = pa.value, pb.value
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(a, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(a + filled, a, step)
    filled += step


def timed(fn, reps=10, warm=3):
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


def lib_ms(call):
    ts = []
    for k in range(9):
        t = ctypes.c_double(0)
        call(ctypes.byref(t))
        if k >= 2:
            ts.append(t.value)
    return sorted(ts)[len(ts) // 2]


for n in [int(v) for v in args.sizes.split(",")]:
    nffts = TOTAL // n
    gb = 2 * nbytes / 1e9
    for reo in (1, 0):
        # in-LDS path: SMFFT_DIT_multiple<P> in the reference's shape next to the compact kernel, README batch
        per_block = max(n, 128) // n if n <= 128 else 1
        if n <= 128:
            per_block = 128 // n
        blocks = (nffts // 100) // per_block
        ref = timed(lambda: ex.smfft_example_reference_shape_ct_multiple(a, b, n, blocks, reo, None), reps=15, warm=25)
        lib = lib_ms(lambda t: sm.lib.smfft_ct_multiple_benchmark(a, b, n, nffts, 0, reo, t))
        done = blocks * per_block * 100
        print(f"in-LDS N={n} reorder={reo}: reference contract {ref:.3f} ms {done / ref * 1e3:.3e} FFT/s | compact {lib:.3f} ms | ratio {lib / ref:.2f}", flush=True)
        if n <= 128 and has_wave64:       # the wave64-full classes: 64-thread blocks of 256 elements
            per64 = 256 // n
            blocks64 = (nffts // 100) // per64
            ref64 = timed(lambda: ex.smfft_example_reference_shape_ct_multiple_wave64(a, b, n, blocks64, reo, None), reps=15, warm=25)
            done64 = blocks64 * per64 * 100
            print(f"in-LDS N={n} reorder={reo} _wave64: reference contract {ref64:.3f} ms {done64 / ref64 * 1e3:.3e} FFT/s | ratio to compact {lib * done64 / done / ref64:.2f}", flush=True)
    if args.in_lds_only:
        continue
    for reo in (1, 0):
        ref = timed(lambda: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, reo, 1, None))
        usr = timed(lambda: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, reo, 0, None))
        lib = lib_ms(lambda t: sm.lib.smfft_ct_external_benchmark(a, b, n, nffts, 0, reo, t))
        print(f"external N={n} reorder={reo}: two-argument kernel {ref:.3f} ms {gb / ref:.2f} TB/s | user kernel (fill / call / drain) {usr:.3f} ms | tiled {lib:.3f} ms {gb / lib:.2f} TB/s | ratios {lib / ref:.2f} {lib / usr:.2f}", flush=True)
        if n <= 128 and has_wave64:
            ref64 = timed(lambda: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, reo, 3, None))
            usr64 = timed(lambda: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, reo, 2, None))
            print(f"external N={n} reorder={reo} _wave64: two-argument kernel {ref64:.3f} ms | user kernel {usr64:.3f} ms | ratios {lib / ref64:.2f} {lib / usr64:.2f}", flush=True)
    if n >= 256:
        ref = timed(lambda: ex.smfft_example_reference_shape_st(a, b, n, nffts, None))
        lib = lib_ms(lambda t: sm.lib.smfft_st_external_benchmark(a, b, n, nffts, t))
        print(f"Stockham external N={n}: reference contract {ref:.3f} ms {gb / ref:.2f} TB/s | tiled {lib:.3f} ms {gb / lib:.2f} TB/s | ratio {lib / ref:.2f}", flush=True)
    if n >= 512:
        # R2C / C2R program, real length n: FFT_GPU_R2C_C2R_external<FFT_{n/2}, D><<<nFFTs, n/8>>> against the tiled kernels (2 GiB in + 2 GiB out)
        rnffts = TOTAL // n
        rgb = 2 * rnffts * n * 4 / 1e9
        for inv in (0, 1):
            ref = timed(lambda: ex.smfft_example_reference_shape_rc(a, b, n, rnffts, inv, None))
            lib = lib_ms(lambda t: sm.lib.smfft_rc_external_benchmark(a, b, n, rnffts, inv, t))
            print(f"{'C2R' if inv else 'R2C'} external real N={n}: two-argument kernel {ref:.3f} ms {rgb / ref:.2f} TB/s | tiled {lib:.3f} ms {rgb / lib:.2f} TB/s | ratio {lib / ref:.2f}", flush=True)
    if n == 1024:
        slots = nffts // 100
        for which, name, call in ((0, "CT multiple reorder", lambda t: sm.lib.smfft_ct_multiple_benchmark(a, b, n, nffts, 0, 1, t)),
                                  (1, "CT multiple no-reorder", lambda t: sm.lib.smfft_ct_multiple_benchmark(a, b, n, nffts, 0, 0, t)),
                                  (2, "Stockham multiple", lambda t: sm.lib.smfft_st_multiple_benchmark(a, b, n, nffts, t))):
            ref = timed(lambda: ex.smfft_example_reference_shape_multiple_one(a, b, slots, which, None))
            lib = lib_ms(call)
            print(f"{name} N={n} ({slots} slots x 100): reference contract {ref:.3f} ms {slots * 100 / ref * 1e3:.3e} FFT/s | compact {lib:.3f} ms {slots * 100 / lib * 1e3:.3e} FFT/s | ratio {lib / ref:.2f}", flush=True)
