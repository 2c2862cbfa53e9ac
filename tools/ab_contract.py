"""Several builds of examples/ (the reference-contract kernels) in ONE process on the SAME buffers, rounds interleaved: the in-LDS
loop SMFFT_DIT_multiple<P> in the reference's launch shape at the README batch, per length and ordering; optionally the fused
convolution kernel (N = 1024).
    python tools/ab_contract.py name=lib.so [name=lib.so ...] [--sizes 256,1024] [--rounds 7] [--conv]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

args = sys.argv[1:]
sizes, rounds, conv = [256, 512, 1024, 2048, 4096], 7, False
libs = []
while args:
    a = args.pop(0)
    if a == "--sizes":
        sizes = [int(v) for v in args.pop(0).split(",")]
    elif a == "--rounds":
        rounds = int(args.pop(0))
    elif a == "--conv":
        conv = True
    else:
        name, path = a.split("=", 1)
        libs.append((name, ctypes.CDLL(os.path.abspath(path))))
vp, ci = ctypes.c_void_p, ctypes.c_int
for _, ex in libs:
    ex.smfft_example_reference_shape_ct_multiple.argtypes = [vp, vp, ci, ci, ci, vp]

TOTAL = 1 << 29
nbytes = TOTAL * 8
A, B = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(A.ptr, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(A.ptr + filled, A.ptr, step)
    filled += step


def once(fn, reps=5):
    fn()
    sm.lib.smfft_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sm.lib.smfft_synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


for n in sizes:
    nffts = TOTAL // n
    per_block = 128 // n if n <= 128 else 1
    blocks = (nffts // 100) // per_block
    for reo in (1, 0):
        best = {name: [] for name, _ in libs}
        for _ in range(3):      # warm
            for name, ex in libs:
                ex.smfft_example_reference_shape_ct_multiple(A.ptr, B.ptr, n, blocks, reo, None)
        for r in range(rounds):
            for name, ex in libs:
                best[name].append(once(lambda: ex.smfft_example_reference_shape_ct_multiple(A.ptr, B.ptr, n, blocks, reo, None)))
        line = f"in-LDS contract N={n} reorder={reo}:"
        for name, _ in libs:
            ms = sorted(best[name])[len(best[name]) // 2]
            line += f" | {name} {ms:.4f} ms {blocks * per_block * 100 / ms * 1e3:.3e} FFT/s"
        print(line, flush=True)
if conv:
    NS = 524288
    h = np.zeros(1024, np.complex128)
    h[:5] = [0.4, 0.3, 0.2, 0.1, -0.05j]
    H = sm.DeviceBuffer.from_host(np.fft.fft(h).astype(np.complex64))
    for sym in ("smfft_example_reference_shape_convolve_1024", "smfft_example_reference_shape_convolve_1024_registers"):
        line = f"{sym}:"
        res = {name: [] for name, _ in libs}
        for r in range(rounds):
            for name, ex in libs:
                fn = getattr(ex, sym)
                fn.argtypes = [vp, vp, vp, ci, vp]
                res[name].append(once(lambda: fn(A.ptr, H.ptr, B.ptr, NS, None), reps=3))
        for name, _ in libs:
            line += f" | {name} {sorted(res[name])[len(res[name]) // 2]:.4f} ms"
        print(line, flush=True)
