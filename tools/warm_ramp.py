"""How many launches of the in-LDS kernel it takes after memory-bound work until its time settles (the device's clocks follow
the load: right after 4 GiB external launches the first in-LDS launches run ~10 % slower).  Prints the times of 600
consecutive FFT_multiple_benchmark calls at N=1024 after 30 external launches, for both orderings."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

N, NFFTS = 1024, 524288
A, B = sm.DeviceBuffer(NFFTS * N * 8), sm.DeviceBuffer(NFFTS * N * 8)
t = ctypes.c_double(0.0)
for reo in (0, 1):
    for _ in range(30):
        sm.lib.smfft_ct_external_benchmark(A.ptr, B.ptr, N, NFFTS, 0, 1, ctypes.byref(t))
    ts = []
    for _ in range(600):
        t.value = 0.0                      # the entry accumulates into *time like the reference's
        sm.lib.smfft_ct_multiple_benchmark(A.ptr, B.ptr, N, NFFTS, 0, reo, ctypes.byref(t))
        ts.append(t.value)
    print("reorder" if reo else "noreorder", "launches 0-9:", " ".join("%.3f" % x for x in ts[:10]))
    print("   every 25th:", " ".join("%.3f" % x for x in ts[::25]))
