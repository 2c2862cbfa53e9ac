"""A/B of two library builds on the in-LDS `multiple` path, same buffers, interleaved rounds.
usage: python tools/ab_mult_probe.py libA.so libB.so [sizes=1024] [batch multipliers=1,10]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
sizes = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1024").split(",")]
mults = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,10").split(",")]
TOTAL = 1 << 29
for l in libs:
    l.smfft_ct_multiple_benchmark.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
a, b = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, TOTAL * 8, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, chunk.nbytes)
for n in sizes:
    for reo in (1, 0):
        for mult in mults:
            nffts = TOTAL // n * mult
            done = (nffts // 400 * 400) if n == 32 else (nffts // 200 * 200) if n == 64 else (nffts // 100 * 100)
            res = [[], []]
            for rnd in range(9):
                for k, l in enumerate(libs):
                    v = ctypes.c_double(0)
                    l.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, nffts, 0, reo, ctypes.byref(v))
                    res[k].append(v.value)
            out = []
            for k in range(2):
                r = sorted(res[k][2:])
                out.append(f"{os.path.basename(sys.argv[1 + k])}: {r[len(r)//2]:.4f} ms {done / r[len(r)//2] * 1e3:.3e} FFT/s")
            print(f"N={n} reorder={reo} x{mult}: " + " | ".join(out), flush=True)
