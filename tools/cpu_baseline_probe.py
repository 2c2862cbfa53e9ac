import ctypes, os, sys, time
import numpy as np
ROOT = os.getcwd()
n, m = 1024, 131072
rng = np.random.default_rng(1)
x = (rng.random((m, n), dtype=np.float32) + 1j * rng.random((m, n), dtype=np.float32)).astype(np.complex64)
out = np.empty_like(x)
fp = ctypes.POINTER(ctypes.c_float)
fb = ctypes.CDLL(os.path.join(ROOT, "oracle", "fftw_baseline.so"))
fb.fftw_baseline_init.argtypes = [ctypes.c_int]
fb.fftw_baseline_c2c_sliced.restype = ctypes.c_double
fb.fftw_baseline_c2c_sliced.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
print("init", fb.fftw_baseline_init(1), "with_torch" if "torch" in sys.modules else "no_torch")
for rep in range(2):
    for thr in (1, 8, 32, 64, 128, 256):
        t0 = time.time()
        t = fb.fftw_baseline_c2c_sliced(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, m, 0, 5, thr)
        print(f"threads {thr:3d}: best {t*1e3:8.2f} ms  {m/t:.3e} FFT/s  (wall {time.time()-t0:.1f}s)", flush=True)
