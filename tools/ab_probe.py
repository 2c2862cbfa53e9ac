"""A/B of two library builds in ONE process on the SAME buffers (placement-probed), interleaved rounds.
usage: python tools/ab_probe.py libA.so libB.so [N] [variant f1|f0|i1]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
var = sys.argv[4] if len(sys.argv) > 4 else "f1"
rc = var in ("r2c", "c2r")       # real transforms: n is the REAL length, half the bytes per FFT
inv, reo = (int(var == "c2r"), 1) if rc else (int(var[0] == "i"), int(var[1] == "1"))
nffts = (1 << 29) // n
nbytes = (1 << 29) * 8 if var not in ("r2c", "c2r") else (1 << 29) * 4
for l in libs:
    l.smfft_ct_external_benchmark.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    l.smfft_rc_external_benchmark.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
def t(l, i, o, k=1):
    v = ctypes.c_double(0)
    for _ in range(k):
        if rc:
            l.smfft_rc_external_benchmark(i, o, n, nffts, inv, ctypes.byref(v))
        else:
            l.smfft_ct_external_benchmark(i, o, n, nffts, inv, reo, ctypes.byref(v))
    return v.value / k
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0   # placement-probed pair
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(pa.value, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(pa.value + filled, pa.value, step)
    filled += step


class _B:
    def __init__(self, p):
        self.ptr = p


a, b = _B(pa.value), _B(pb.value)
res = [[], []]
for rnd in range(15):
    for k, l in enumerate(libs):
        res[k].append(t(l, a.ptr, b.ptr, 3))
for k in range(2):
    r = sorted(res[k])
    print(f"{os.path.basename(sys.argv[1 + k]):28s} N={n} {var}: median {r[7]:.4f} ms ({2*nbytes/r[7]/1e6:.0f} GB/s)  min {r[0]:.4f}")
