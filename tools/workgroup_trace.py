"""One `multiple` launch with per-workgroup traces: SMFFT_SCHEDULE_TRACE=<tmp file> python tools/workgroup_trace.py N chains balance out.txt [nreuses] [reorder]
(each line: block, start and end of s_memtime on its CU, HW_ID, XCC_ID -- counters of different CUs are not aligned --, then start, end
and "first tile in LDS" on the device-wide 100 MHz clock)"""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
n, ntiles, bal, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
nreuses = int(sys.argv[5]) if len(sys.argv) > 5 else 100
reorder = int(sys.argv[6]) if len(sys.argv) > 6 else 1
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)
tile = max(1, 1024 // n)
sm.lib.smfft_set_multiple_balance(bal)
sm.lib.smfft_set_nreuses(nreuses)
ts = []
for _ in range(60):
    t = ctypes.c_double(0)
    sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, ntiles * tile * 100, 0, reorder, ctypes.byref(t))
    ts.append(t.value)
print(n, ntiles, bal, nreuses, "last launch ms", t.value, "median of the last 20 (traced: one hipMalloc + copy each)", sorted(ts[-20:])[10])
os.rename(os.environ["SMFFT_SCHEDULE_TRACE"], out)
