"""Full (input offset, output offset) map of the N=1024 external kernel's time inside one arena, 4 GiB lattice.
usage: python tools/placement_map2.py [arena_GiB=192] [step_GiB=4]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
nb = n * nffts * 8
G = 1 << 30
total = int(sys.argv[1]) if len(sys.argv) > 1 else 192
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
arena = sm.DeviceBuffer(total * G)
sm.lib.smfft_memset(arena.ptr, 0, total * G)     # content does not matter for the timing; avoid NaN paths anyway
offs = list(range(0, total - 3, step))
print(f"arena {arena.ptr:#x} {total} GiB; ms of one launch (min of 3) for input offset (rows) x output offset (columns), GiB")
print("      " + " ".join(f"{o:5d}" for o in offs))
best = []
for i in offs:
    row = []
    for o in offs:
        if abs(i - o) < 4:
            row.append("    -")
            continue
        sm.FFT_external_benchmark(arena.ptr + i * G, arena.ptr + o * G, n, nffts)
        ms = min(sm.FFT_external_benchmark(arena.ptr + i * G, arena.ptr + o * G, n, nffts)[1] for _ in range(3))
        row.append(f"{ms:5.3f}")
        best.append((ms, i, o))
    print(f"{i:5d} " + " ".join(row), flush=True)
best.sort()
print("fastest:", [(round(m, 4), i, o) for m, i, o in best[:12]])
print("slowest:", [(round(m, 4), i, o) for m, i, o in best[-5:]])
import collections
h = collections.Counter(round(m, 2) for m, _, _ in best)
print("histogram (ms: pairs):", sorted(h.items()))
