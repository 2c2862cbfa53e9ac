"""smfft_malloc_pair / smfft_free_pair in a loop, in a process that does NOT import torch (so the library runs on the system's
HIP runtime, as a C program would): the memory held while a pair exists and what is still missing after it is freed -- the
check that found the allocator's leak on ROCm 7.2 (physical memory of released handles comes back only when their virtual
range is freed).   python tools/allocator_soak.py [GiB per buffer = 4] [cycles = 60]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm
nb = (int(sys.argv[1]) if len(sys.argv) > 1 else 4) << 30
def free_mib():
    f = ctypes.c_ulonglong(); t = ctypes.c_ulonglong()
    sm.lib.smfft_mem_info(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20
f0 = free_mib()
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = sm.lib.smfft_malloc_pair(nb, ctypes.byref(a), ctypes.byref(b))
    dt = time.perf_counter() - t0
    info = sm.last_pair_info()
    held = f0 - free_mib()
    sm.lib.smfft_free_pair(a.value)
    print(f"cycle {i}: rc {rc} {dt * 1e3:.0f} ms chunks {info['candidates']} scanned {info['candidate_bytes'] >> 30} GiB; held while allocated {held:.0f} MiB; right after free {f0 - free_mib():.0f} MiB", flush=True)
for s in (0.5, 1, 2):
    time.sleep(s)
    print(f"  after {s} s more: still missing {f0 - free_mib():.0f} MiB", flush=True)
