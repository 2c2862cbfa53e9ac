"""Is a hipMallocAsync (stream-ordered pool) buffer a fast WRITE target without any search?  N=1024 external kernel:
input from hipMalloc, output from hipMallocAsync, against the pair smfft_malloc_pair finds in the same process."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
hip = ctypes.CDLL("libamdhip64.so.7")
n, nffts = 1024, 524288
nbytes = n * nffts * 8
def med(i, o, reps=9):
    sm.FFT_external_benchmark(i, o, n, nffts)
    return sorted(sm.FFT_external_benchmark(i, o, n, nffts)[1] for _ in range(reps))[reps // 2]
t0 = time.perf_counter()
a = sm.DeviceBuffer(nbytes)
pool_out = ctypes.c_void_p()
assert hip.hipMallocAsync(ctypes.byref(pool_out), ctypes.c_size_t(nbytes), None) == 0
sm.lib.smfft_synchronize()
dt = time.perf_counter() - t0
sm.lib.smfft_memset(a.ptr, 0, nbytes)
print(f"hipMalloc input + hipMallocAsync output ({dt*1e3:.0f} ms to allocate): {med(a.ptr, pool_out.value):.4f} ms")
pool_in = ctypes.c_void_p()
assert hip.hipMallocAsync(ctypes.byref(pool_in), ctypes.c_size_t(nbytes), None) == 0
sm.lib.smfft_synchronize()
sm.lib.smfft_memset(pool_in.value, 0, nbytes)
print(f"both from the pool: {med(pool_in.value, pool_out.value):.4f} ms; pool input + hipMalloc output: {med(pool_out.value, a.ptr):.4f} ms")
b = sm.DeviceBuffer(nbytes)
print(f"two hipMalloc buffers: {med(a.ptr, b.ptr):.4f} ms")
for gib in (8, 16):
    big_in, big_out = sm.DeviceBuffer(gib << 30), ctypes.c_void_p()
    assert hip.hipMallocAsync(ctypes.byref(big_out), ctypes.c_size_t(gib << 30), None) == 0
    sm.lib.smfft_synchronize()
    k = (gib << 30) // (n * 8)
    sm.FFT_external_benchmark(big_in.ptr, big_out.value, n, k)
    ms = sorted(sm.FFT_external_benchmark(big_in.ptr, big_out.value, n, k)[1] for _ in range(5))[2]
    print(f"{gib} GiB buffers (hipMalloc in, pool out): {ms:.4f} ms = {2 * (gib << 30) / ms / 1e6:.0f} GB/s")
    hip.hipFreeAsync(big_out, None); sm.lib.smfft_synchronize(); big_in.free()
t0 = time.perf_counter()
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb))
print(f"smfft_malloc_pair search ({time.perf_counter() - t0:.1f} s): {med(pa.value, pb.value):.4f} ms")
