"""In-LDS R2C / C2R path (`multiple`: 100 applications per slot in LDS), README batch, two library builds in one process:
    tools/build_variant.sh rcold -DSMFFT_RC_MULTIPLE_FUSED=0 ; python tools/ab_rc_multiple.py
old = split / merge as a separate LDS-resident pass per application; new = fused into the load of the following transform."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm
old = ctypes.CDLL(os.path.abspath("smfft_amd/libsmfft_amd_rcold.so"))
vp, i, dp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
for l in (old, sm.lib):
    l.smfft_rc_multiple_benchmark.argtypes = [vp, vp, i, i, dp]
    l.smfft_launch.argtypes = [i, i, vp, vp, i, i, i, i, vp]
TOTAL = 1 << 29
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
sm.lib.smfft_memset(A.ptr, 0, TOTAL * 8)
import time
def med(call):
    ts = []
    for _ in range(9):
        t = ctypes.c_double(0); call(ctypes.byref(t)); ts.append(t.value)
    return sorted(ts[2:])[3]
def med_launch(l, rn, n, inv):
    ts = []
    for _ in range(9):
        sm.lib.smfft_synchronize(); t0 = time.perf_counter()
        l.smfft_launch(2, 1, A.ptr, B.ptr, rn, n, inv, 1, None); sm.lib.smfft_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts[2:])[3]
for rep in range(2):
    for rn in (512, 1024, 2048, 4096):
        n = TOTAL * 2 // rn
        print(f"real N={rn}: R2C multiple old {med(lambda t: old.smfft_rc_multiple_benchmark(A.ptr, B.ptr, rn, n, t)):.4f} new {med(lambda t: sm.lib.smfft_rc_multiple_benchmark(A.ptr, B.ptr, rn, n, t)):.4f} ms | C2R multiple (wall) old {med_launch(old, rn, n, 1):.4f} new {med_launch(sm.lib, rn, n, 1):.4f} ms", flush=True)
