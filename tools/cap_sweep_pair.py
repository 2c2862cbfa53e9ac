"""Grid-cap sweep of the external kernels on a placement-searched pair (smfft_malloc_pair)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024").split(",")]
caps = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8192,12288,16384,20480,24576,32768,49152").split(",")]
nbytes = (1 << 29) * 8
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
sm.lib.smfft_memset(pa.value, 0, nbytes)
for n in sizes:
    nffts = (1 << 29) // n
    res = {c: [] for c in caps}
    for rnd in range(9):
        for c in caps:
            sm.lib.smfft_set_grid_cap(c)
            t = 0
            for _ in range(3):
                t += sm.FFT_external_benchmark(pa.value, pb.value, n, nffts, False, True)[1]
            res[c].append(t / 3)
    print(f"N={n}: " + "  ".join(f"{c}: {sorted(r[1:])[4]:.4f} ms ({2*nbytes/sorted(r[1:])[4]/1e6:.0f})" for c, r in res.items()), flush=True)
