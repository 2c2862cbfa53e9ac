"""Looks for a box whose first quarter of memory is ONE memory class (smfft_malloc_pair finds nothing to mix or interleave) and,
when it has one, shows what a patient scan finds there: python tools/uniform_box_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402


def c2c_ms(a, b):
    t = ctypes.c_double(0)
    ts = []
    for _ in range(8):
        t.value = 0
        sm.lib.smfft_ct_external_benchmark(a, b, 1024, 524288, 0, 1, ctypes.byref(t))
        ts.append(t.value)
    return sorted(ts)[3]


pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(1 << 32, ctypes.byref(pa), ctypes.byref(pb)) == 0
info = sm.last_pair_info()
helped = info["first_ordinary_copy_ms"] > 0 and info["copy_ms"] < 0.97 * info["first_ordinary_copy_ms"]
print("default budget:", {k: info[k] for k in ("candidates", "good_enough", "classification", "mixed_bytes", "interleaved_bytes", "copy_ms", "first_ordinary_copy_ms", "search_ms")},
      "C2C %.4f ms" % c2c_ms(pa.value, pb.value), flush=True)
if info["good_enough"] or helped:
    print("an ordinary box")
    sys.exit(0)
sm.lib.smfft_free_pair(pa.value)
t0 = time.perf_counter()
os.environ["SMFFT_PAIR_DEBUG"] = "1"
assert sm.lib.smfft_malloc_pair_budget(1 << 32, ctypes.byref(pa), ctypes.byref(pb), 0.9, 20000.0) == 0
info = sm.last_pair_info()
print("UNIFORM BOX; patient budget (%.1f s):" % (time.perf_counter() - t0),
      {k: info[k] for k in ("candidates", "good_enough", "classification", "mixed_bytes", "interleaved_bytes", "copy_ms", "first_ordinary_copy_ms", "search_ms")},
      "C2C %.4f ms" % c2c_ms(pa.value, pb.value), flush=True)
