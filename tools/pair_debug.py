import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["SMFFT_PAIR_DEBUG"] = "1"
import smfft_amd as sm
for nbytes in (4 << 30, 1 << 30):
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    t0 = time.time()
    assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(a), ctypes.byref(b)) == 0
    print(time.time() - t0, sm.last_pair_info(), flush=True)
    sm.lib.smfft_free_pair(a.value)
