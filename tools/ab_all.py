"""A/B/C... of several library builds over all lengths: external f1 (GB/s) and in-LDS f1/f0 (FFT/s), same
placement-probed buffers, interleaved rounds.  usage: python tools/ab_all.py libA.so libB.so [...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
paths = sys.argv[1:]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
sig = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for l in libs:
    l.smfft_ct_external_benchmark.argtypes = sig
    l.smfft_ct_multiple_benchmark.argtypes = sig
TOTAL = 1 << 29
nbytes = TOTAL * 8
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(pa.value, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(pa.value + filled, pa.value, step)
    filled += step
print("libs:", [os.path.basename(p) for p in paths])


def med(fn, rounds=9):
    res = [[] for _ in libs]
    for _ in range(rounds):
        for k, l in enumerate(libs):
            v = ctypes.c_double(0)
            for _ in range(2):
                fn(l, v)
            res[k].append(v.value / 2)
    return [sorted(r[1:])[len(r[1:]) // 2] for r in res]


for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    nffts = TOTAL // n
    ext = med(lambda l, v: l.smfft_ct_external_benchmark(pa.value, pb.value, n, nffts, 0, 1, ctypes.byref(v)))
    m1 = med(lambda l, v: l.smfft_ct_multiple_benchmark(pa.value, pb.value, n, nffts * 4, 0, 1, ctypes.byref(v)), 5)
    m0 = med(lambda l, v: l.smfft_ct_multiple_benchmark(pa.value, pb.value, n, nffts * 4, 0, 0, ctypes.byref(v)), 5)
    print(f"N={n:5d} ext f1 GB/s: " + " ".join(f"{2 * nbytes / t / 1e6:6.0f}" for t in ext)
          + " | mult x4 f1 ms: " + " ".join(f"{t:7.4f}" for t in m1) + " | f0 ms: " + " ".join(f"{t:7.4f}" for t in m0), flush=True)
