"""End-to-end rate of smfft_host_transform on a host-resident config-2 batch (N=1024, 4 GiB in + 4 GiB out),
pinned and pageable, over lane counts and slab sizes; next to the L3 wrapper (one pageable copy each way).
usage: python tools/host_stream_probe.py [log2_elements=29]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
total = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 29)
n = 1024
nffts = total // n
gib = total * 8 / 2**30
rng = np.random.default_rng(0)
xp = sm.pinned_empty((nffts, n), np.complex64)
chunk = (rng.random((4096, n), dtype=np.float32) + 1j * rng.random((4096, n), dtype=np.float32)).astype(np.complex64)
for i in range(0, nffts, 4096):
    xp[i:i + 4096] = chunk[: min(4096, nffts - i)]
yp = sm.pinned_empty((nffts, n), np.complex64)
x = np.array(xp)                 # pageable copies
y = np.empty_like(x)
print(f"batch: {nffts} FFTs of {n} = {gib:.1f} GiB in + {gib:.1f} GiB out, host resident")
for rep in range(3):
    _, ms = sm.host_transform(xp, out=yp)
print(f"  pinned, zero copy (the kernel reads and writes the host buffers): {ms:8.1f} ms  = {2 * gib * 2**30 / ms / 1e6:6.1f} GB/s (in+out)  {nffts / ms * 1e3:.3e} FFT/s", flush=True)
os.environ["SMFFT_HOST_ZERO_COPY"] = "0"          # the slab pipeline (DMA copies to and from device slabs) from here on
for name, a, b in (("pinned", xp, yp), ("pageable", x, y)):
    for lanes, slab_mib in ((0, 32), (8, 32), (2, 32)):
        best = 1e9
        for rep in range(3):
            _, ms = sm.host_transform(a, out=b, slab_ffts=slab_mib * 2**20 // (n * 8), lanes=lanes)
            best = min(best, ms)
        print(f"  {name:9s} lanes={lanes:2d} slab={slab_mib:3d} MiB: {best:8.1f} ms  = {2 * gib * 2**30 / best / 1e6:6.1f} GB/s (in+out)  {nffts / best * 1e3:.3e} FFT/s", flush=True)
ref = sm.c2c(chunk[:64])
assert np.array_equal(yp[:64], ref) and np.array_equal(y[:64], ref)
import ctypes
dev = sm.DeviceBuffer(x.nbytes)
for rep in range(2):
    t0 = time.perf_counter(); sm.lib.smfft_memcpy_h2d(dev.ptr, x.ctypes.data, x.nbytes); dt = time.perf_counter() - t0
print(f"  plain hipMemcpy, resident pageable memory, H2D: {dt * 1e3:.1f} ms = {x.nbytes / dt / 1e9:.1f} GB/s")
for rep in range(2):
    t0 = time.perf_counter(); sm.lib.smfft_memcpy_d2h(y.ctypes.data, dev.ptr, x.nbytes); dt = time.perf_counter() - t0
print(f"  plain hipMemcpy, resident pageable memory, D2H: {dt * 1e3:.1f} ms = {x.nbytes / dt / 1e9:.1f} GB/s")
dev.free()
t0 = time.perf_counter()
s1, s2 = ctypes.c_double(0), ctypes.c_double(0)
sm.lib.smfft_gpu_ct(x.ctypes.data, y.ctypes.data, n, nffts, 0, 1, 1, ctypes.byref(s1), ctypes.byref(s2))
dt = time.perf_counter() - t0
print(f"  L3 wrapper smfft_gpu_ct (alloc, pageable H2D, 1 external + 1 multiple launch, D2H): {dt * 1e3:.1f} ms = {2 * gib * 2**30 / dt / 1e9:.1f} GB/s")
