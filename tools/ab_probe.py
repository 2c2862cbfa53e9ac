"""A/B of two library builds in ONE process on the SAME buffers (placement-probed), interleaved rounds.
usage: python tools/ab_probe.py libA.so libB.so [N] [variant f1|f0|i1]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
var = sys.argv[4] if len(sys.argv) > 4 else "f1"
inv, reo = int(var[0] == "i"), int(var[1] == "1")
nffts = (1 << 29) // n
nbytes = (1 << 29) * 8
for l in libs:
    l.smfft_ct_external_benchmark.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
def t(l, i, o, k=1):
    v = ctypes.c_double(0)
    for _ in range(k):
        l.smfft_ct_external_benchmark(i, o, n, nffts, inv, reo, ctypes.byref(v))
    return v.value / k
cands = [sm.DeviceBuffer(nbytes) for _ in range(5)]
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, nbytes, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(cands[0].ptr + off, chunk.ctypes.data, chunk.nbytes)
for c in cands[1:]:
    sm.lib.smfft_memcpy_d2d(c.ptr, cands[0].ptr, nbytes)
best = min(((t(libs[0], a.ptr, b.ptr, 5), ia, ib) for ia, a in enumerate(cands) for ib, b in enumerate(cands) if ia != ib))
a, b = cands[best[1]], cands[best[2]]
res = [[], []]
for rnd in range(15):
    for k, l in enumerate(libs):
        res[k].append(t(l, a.ptr, b.ptr, 3))
for k in range(2):
    r = sorted(res[k])
    print(f"{os.path.basename(sys.argv[1 + k]):28s} N={n} {var}: median {r[7]:.4f} ms ({2*nbytes/r[7]/1e6:.0f} GB/s)  min {r[0]:.4f}")
