"""Map of the N=1024 external kernel's time over (input chunk, output chunk) for separately allocated 16 GiB chunks that
together cover most of the device memory (which chunk pairs give the 1.34 ms 'super-fast' placement?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
G = 1 << 30
nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chunks = []
for k in range(nchunks):
    try:
        chunks.append(sm.DeviceBuffer(16 * G))
    except MemoryError:
        break
print(len(chunks), "chunks of 16 GiB; virtual addresses (GiB):", [round(c.ptr / G, 1) for c in chunks])
print("rows: input chunk, columns: output chunk (4 GiB windows at the start of each chunk; same chunk: input at 0, output at 8 GiB)")
best = []
for i, ci in enumerate(chunks):
    row = []
    for o, co in enumerate(chunks):
        ip, op = ci.ptr, (co.ptr if o != i else co.ptr + 8 * G)
        sm.FFT_external_benchmark(ip, op, n, nffts)
        ms = min(sm.FFT_external_benchmark(ip, op, n, nffts)[1] for _ in range(3))
        row.append(f"{ms:5.3f}")
        best.append((ms, i, o))
    print(f"{i:3d} " + " ".join(row), flush=True)
best.sort()
print("fastest:", [(round(m, 4), i, o) for m, i, o in best[:10]])
