"""Runs BASELINE.json's configs 2, 3 and 4 on the GPU box and writes one JSON document
(profiles/rNN_configs.json).  Timing: FFT_*_benchmark-equivalent calls (one event-timed launch
each), median and min of `rounds` launches after 3 warm-ups, data resident, random U[0,1) input.

    python tools/run_configs.py --out gpurun_out/r1/configs.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="configs.json")
ap.add_argument("--rounds", type=int, default=21)
args = ap.parse_args()

TOTAL = 1 << 29                      # float2 elements = 4 GiB (README's "input data size is 4GB")
nbytes = TOTAL * 8
rng = np.random.default_rng(0)
chunk = rng.random(1 << 22, dtype=np.float32)
# buffer placement (same as bench.py and the L3 wrappers: where the two 4 GiB buffers land physically changes the
# streaming rate by 6-8 %): the library's placement-probed pair allocator
import ctypes  # noqa: E402


class _Raw:
    def __init__(self, ptr):
        self.ptr = ptr


pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
a, b = _Raw(pa.value), _Raw(pb.value)
sm.lib.smfft_memcpy_h2d(a.ptr, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(a.ptr + filled, a.ptr, step)
    filled += step
sm.FFT_init()


def timed(fn):
    for _ in range(3):
        fn()
    ts = sorted(fn()[1] for _ in range(args.rounds))
    return ts[len(ts) // 2], ts[0]


doc = {"grid_cap": sm.lib.smfft_get_grid_cap(), "rounds": args.rounds, "unit_time": "ms",
       "buffers": "smfft_malloc_pair"}

# ---- config 2: N=1024 C2C forward + inverse with reorder, 524288 FFTs, external path
c2 = {}
for name, inv in (("forward", False), ("inverse", True)):
    med, mn = timed(lambda: sm.FFT_external_benchmark(a.ptr, b.ptr, 1024, TOTAL // 1024, inv, True))
    c2[name] = {"median_ms": med, "min_ms": mn, "GB/s": 2 * nbytes / med / 1e6, "FFT/s": TOTAL // 1024 / med * 1e3,
                "frac_of_8TBps": 2 * nbytes / med / 1e6 / 8000.0}
doc["config2_N1024_c2c_reorder_external"] = c2

# ---- config 3: N = 32..4096, no-reorder (and reorder), forward, multiple path, 4 GiB-sized nFFTs
c3 = {}
for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    nffts = TOTAL // n
    row = {}
    for name, reo in (("noreorder", False), ("reorder", True)):
        med, mn = timed(lambda: sm.FFT_multiple_benchmark(a.ptr, b.ptr, n, nffts, False, reo))
        done = (nffts // 400 * 400) if n == 32 else (nffts // 200 * 200) if n == 64 else (nffts // 100 * 100)
        row[name] = {"median_ms": med, "FFT/s": done / med * 1e3, "GFLOP/s_5NlogN": done * 5 * n * np.log2(n) / med / 1e6}
    med, mn = timed(lambda: sm.FFT_external_benchmark(a.ptr, b.ptr, n, nffts, False, False))
    row["external_noreorder"] = {"median_ms": med, "GB/s": 2 * nbytes / med / 1e6}
    med, mn = timed(lambda: sm.FFT_external_benchmark(a.ptr, b.ptr, n, nffts, False, True))
    row["external_reorder"] = {"median_ms": med, "GB/s": 2 * nbytes / med / 1e6}
    c3[str(n)] = row
doc["config3_sweep_multiple_and_external"] = c3

# ---- Stockham C2C program (inverse sign) and config 4: R2C + C2R, real N=2048, 262144 FFTs
st = {}
for n in (256, 512, 1024, 2048, 4096):
    med, _ = timed(lambda: sm.FFT_external_benchmark(a.ptr, b.ptr, n, TOTAL // n, family="st"))
    med2, _ = timed(lambda: sm.FFT_multiple_benchmark(a.ptr, b.ptr, n, TOTAL // n, family="st"))
    st[str(n)] = {"external_ms": med, "external_GB/s": 2 * nbytes / med / 1e6, "multiple_ms": med2, "multiple_FFT/s": (TOTAL // n // 100 * 100) / med2 * 1e3}
doc["stockham_c2c"] = st
c4 = {}
for n in (512, 1024, 2048, 4096):
    nffts = 262144 * 2048 // n       # 2 GiB of reals in, 2 GiB packed out
    rbytes = n * nffts * 4
    med_f, _ = timed(lambda: sm.FFT_external_benchmark(a.ptr, b.ptr, n, nffts, inverse=False, family="rc"))
    med_i, _ = timed(lambda: sm.FFT_external_benchmark(b.ptr, a.ptr, n, nffts, inverse=True, family="rc"))
    med_m, _ = timed(lambda: sm.FFT_multiple_benchmark(a.ptr, b.ptr, n, nffts, family="rc"))
    c4[str(n)] = {"nFFTs": nffts, "r2c_ms": med_f, "r2c_GB/s": 2 * rbytes / med_f / 1e6, "c2r_ms": med_i, "c2r_GB/s": 2 * rbytes / med_i / 1e6,
                  "r2c_multiple_ms": med_m, "r2c_multiple_FFT/s": (nffts // 100 * 100) / med_m * 1e3}
doc["config4_r2c_c2r_external"] = c4

os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
json.dump(doc, open(args.out, "w"), indent=1)
print(json.dumps(doc["config2_N1024_c2c_reorder_external"]))
for n, row in c3.items():
    print(n, {k: round(v.get("FFT/s", v.get("GB/s")), 1) for k, v in row.items()})
print({k: (round(v["r2c_GB/s"]), round(v["c2r_GB/s"])) for k, v in c4.items()})
