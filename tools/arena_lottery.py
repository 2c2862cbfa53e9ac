"""Is the 'super-fast' placement (config 2 at 1.34 instead of 1.42-1.46 ms) a property of the arena an allocation
happens to get?  Allocates several 100 GiB arenas in one process (two alive at a time, so consecutive ones cannot be
the same physical memory), scans the (input, output) pairs smfft_malloc_pair would scan in each with the N=1024 FFT
kernel, and prints best / quartiles per arena."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
nb = n * nffts * 8
G = 1 << 30
span = 96
arenas = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    a = sm.DeviceBuffer((span + 4) * G)
    arenas.append(a)
    if len(arenas) > 2:
        arenas.pop(0).free()
    res = []
    for i in (0, 32, 64):
        for o in range(0, span + 1, 8):
            if abs(i - o) < 4:
                continue
            sm.FFT_external_benchmark(a.ptr + i * G, a.ptr + o * G, n, nffts)
            ms = min(sm.FFT_external_benchmark(a.ptr + i * G, a.ptr + o * G, n, nffts)[1] for _ in range(3))
            res.append((ms, i, o))
    res.sort()
    t = [r[0] for r in res]
    print(f"arena {k} at {a.ptr:#x}: best {t[0]:.4f} ms (in {res[0][1]}, out {res[0][2]} GiB)  5 best {[round(x, 3) for x in t[:5]]}  median {t[len(t)//2]:.4f}  worst {t[-1]:.4f}  pairs<1.38: {sum(x < 1.38 for x in t)}/{len(t)}", flush=True)
