"""What do hipMalloc / hipFree cost for the chunk search of smfft_malloc_pair?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
G = 1 << 30
def T(f):
    t0 = time.perf_counter(); r = f(); return r, time.perf_counter() - t0
for size, count in ((4, 60), (16, 15), (1, 60)):
    ptrs, dt = T(lambda: [sm.lib.smfft_malloc(size * G) for _ in range(count)])
    _, df = T(lambda: [sm.lib.smfft_free(p) for p in ptrs])
    print(f"{count} x hipMalloc({size} GiB): {dt:.2f} s ({dt/count*1e3:.0f} ms each); hipFree: {df:.2f} s ({df/count*1e3:.0f} ms each)")
p, dt = T(lambda: sm.lib.smfft_malloc(240 * G))
_, df = T(lambda: sm.lib.smfft_free(p))
print(f"1 x hipMalloc(240 GiB): {dt:.2f} s; hipFree {df:.2f} s")
