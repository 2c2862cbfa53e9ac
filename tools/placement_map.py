"""Maps the streaming speed of the N=1024 external kernel as a function of WHERE the 4 GiB input
and output windows sit inside one big hipMalloc'ed arena (physical placement effects)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
nb = n * nffts * 8
G = 1 << 30
total_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 96
arena = sm.DeviceBuffer(total_gib * G)
base = arena.ptr
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
for off in range(0, nb, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(base + off, chunk.ctypes.data, chunk.nbytes)
print(f"arena {arena.ptr:#x} {total_gib} GiB")
def t(i, o):
    sm.FFT_external_benchmark(i, o, n, nffts)
    return sorted(sm.FFT_external_benchmark(i, o, n, nffts)[1] for _ in range(5))[2]
print("input at 0 GiB, output at offset (GiB):")
row = []
for g in range(4, total_gib - 3, 4):
    row.append(f"{g}:{t(base, base + g * G):.3f}")
print(" ".join(row))
mid = (total_gib // 2) // 4 * 4
sm.lib.smfft_memcpy_d2d(base + mid * G, base, nb)
print(f"input at {mid} GiB, output at offset (GiB):")
row = []
for g in range(0, total_gib - 3, 4):
    if abs(g - mid) >= 4:
        row.append(f"{g}:{t(base + mid * G, base + g * G):.3f}")
print(" ".join(row))
