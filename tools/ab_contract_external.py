import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
libs=[(a.split("=")[0], ctypes.CDLL(os.path.abspath(a.split("=")[1]))) for a in sys.argv[1:]]
vp, ci = ctypes.c_void_p, ctypes.c_int
for _, ex in libs:
    ex.smfft_example_reference_shape_ct.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    ex.smfft_example_reference_shape_st.argtypes = [vp, vp, ci, ci, vp]
    ex.smfft_example_reference_shape_rc.argtypes = [vp, vp, ci, ci, ci, vp]
TOTAL = 1 << 29
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(TOTAL * 8, ctypes.byref(pa), ctypes.byref(pb)) == 0
a, b = pa.value, pb.value
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(a, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < TOTAL * 8:
    step = min(filled, TOTAL * 8 - filled); sm.lib.smfft_memcpy_d2d(a + filled, a, step); filled += step
def once(fn, reps=5):
    fn(); sm.lib.smfft_synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    sm.lib.smfft_synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
for n in (1024, 2048, 4096):
    nffts = TOTAL // n
    for label, call in ((f"CT two-arg external N={n} reorder=1", lambda ex: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, 1, 1, None)),
                        (f"CT two-arg external N={n} reorder=0", lambda ex: ex.smfft_example_reference_shape_ct(a, b, n, nffts, 0, 0, 1, None)),
                        (f"Stockham two-arg external N={n}", lambda ex: ex.smfft_example_reference_shape_st(a, b, n, nffts, None)),
                        (f"C2R two-arg external real N={n}", lambda ex: ex.smfft_example_reference_shape_rc(a, b, n, TOTAL // n, 1, None)),
                        (f"R2C two-arg external real N={n}", lambda ex: ex.smfft_example_reference_shape_rc(a, b, n, TOTAL // n, 0, None))):
        res = {name: [] for name, _ in libs}
        for r in range(7):
            for name, ex in libs:
                res[name].append(once(lambda: call(ex)))
        print(label + ": " + " | ".join(f"{name} {sorted(v)[3]:.4f} ms" for name, v in res.items()), flush=True)
    t = ctypes.c_double(0); ts = []
    for k in range(7):
        t.value = 0; sm.lib.smfft_ct_external_benchmark(a, b, n, nffts, 0, 1, ctypes.byref(t)); ts.append(t.value)
    print(f"  tiled N={n}: {sorted(ts)[3]:.4f} ms", flush=True)
