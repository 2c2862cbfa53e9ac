import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    for inv, reo, path in ((0, 1, 1), (0, 0, 1), (0, 1, 2)):
        if path == 2 and n < 64: continue
        a = ctypes.c_int(0)
        got = sm.lib.smfft_measure_multiple_residency(0, n, inv, reo, path, ctypes.byref(a))
        print(f"N={n} reorder={reo} path={path}: counted {got} assumed {a.value}", flush=True)
