"""In-LDS path at the README batch, every length, by the rotation period 2^k of the wave priorities (smfft_set_multiple_rotation), interleaved
in one process on the same buffers:   python tools/rotation_sweep.py [k ...]"""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import smfft_amd as sm
total = 1 << 29
a, b = sm.DeviceBuffer(total * 8), sm.DeviceBuffer(total * 8)
sm.lib.smfft_memset(a.ptr, 0, total * 8)
PEAK = 157.3e12
ks = [int(x) for x in sys.argv[1:]] or [0, 11, 12, 13, 14, 15, 16, 17]
def once(n, reo):
    t = ctypes.c_double(0)
    sm.lib.smfft_ct_multiple_benchmark(a.ptr, b.ptr, n, total // n, 0, reo, ctypes.byref(t))
    return t.value
for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    for reo in (1, 0):
        sp = 0.0
        while sp < 40: sp += once(n, reo)
        ts = {k: [] for k in ks}
        for _ in range(9):
            for k in ks:
                sm.lib.smfft_set_multiple_rotation(k)
                ts[k].append(once(n, reo))
        flops = 5 * n * (n.bit_length() - 1)
        ffts = 100 * (total // n // 100)
        print(f"N={n:4d} reorder={reo}: " + " | ".join(f"2^{k}: {sorted(ts[k])[4]*1e3:6.1f} us {ffts / (sorted(ts[k])[4] * 1e-3) * flops / PEAK:.3f}" for k in ks), flush=True)
sm.lib.smfft_set_multiple_rotation(-1)
