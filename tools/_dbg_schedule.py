import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
for n in (32, 64, 256, 1024, 2048, 4096):
    for reo in (1, 0):
        t = ctypes.c_double(0)
        sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, total // n, 0, reo, ctypes.byref(t))
ctypes.CDLL(None).fflush(None)
