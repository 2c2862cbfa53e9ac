"""Reorder vs no-reorder external kernels, interleaved rounds on placement-probed buffers."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
nbytes = (1 << 29) * 8
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(pa.value, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(pa.value + filled, pa.value, step)
    filled += step
print("offset GiB", (pb.value - pa.value) / 2**30)
for n in (32, 128, 256, 512, 1024, 2048, 4096):
    nffts = (1 << 29) // n
    res = {"f1": [], "f0": [], "i1": []}
    for rnd in range(9):
        for var in res:
            inv, reo = var[0] == "i", var[1] == "1"
            t = 0
            for _ in range(3):
                t += sm.FFT_external_benchmark(pa.value, pb.value, n, nffts, inv, reo)[1]
            res[var].append(t / 3)
    print(f"N={n:5d} " + "  ".join(f"{v}: {sorted(r)[4]:.4f} ms ({2*nbytes/sorted(r)[4]/1e6:.0f})" for v, r in res.items()), flush=True)
