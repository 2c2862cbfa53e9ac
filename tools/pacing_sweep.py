"""Sweep of the external kernels' rate limiter (smfft_set_pacing(K): K serialised LDS loads between a wave's loads and stores):
every length, C2C forward and R2C / C2R, on two plain allocations (default) or on a smfft_malloc_pair pair (--pair).
    python tools/pacing_sweep.py [--pair] [--ks 0,4,8,16] [--sizes 32,...,4096]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true")
ap.add_argument("--ks", default="0,2,4,6,8,12,16,24,32")
ap.add_argument("--sizes", default="32,64,128,256,512,1024,2048,4096")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--consecutive", action="store_true", help="all launches of one K in a row instead of interleaved rounds")
args = ap.parse_args()
ks = [int(v) for v in args.ks.split(",")]
TOTAL = 1 << 29
nbytes = TOTAL * 8
if args.pair:
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
    a, b = pa.value, pb.value
    print("pair:", sm.last_pair_info())
else:
    A, B = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
    a, b = A.ptr, B.ptr
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(a, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(a + filled, a, step)
    filled += step


def sweep(label, call, gbytes):
    res = {k: [] for k in ks}
    order = [(rnd, k) for k in ks for rnd in range(args.rounds + 1)] if args.consecutive else [(rnd, k) for rnd in range(args.rounds + 1) for k in ks]
    for rnd, k in order:
        sm.lib.smfft_set_pacing(k)
        t = ctypes.c_double(0)
        assert call(ctypes.byref(t)) == 0
        if rnd:
            res[k].append(t.value)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    best = min(med, key=med.get)
    print(f"{label}: " + " | ".join(f"K={k} {med[k]:.4f}" for k in ks) + f"  -> best K={best} ({(med[0] / med[best] - 1) * 100:+.1f} % vs K=0, {gbytes / med[best] / 8e6:.3f} of peak)", flush=True)


for n in [int(v) for v in args.sizes.split(",")]:
    sweep(f"C2C N={n}", lambda t: sm.lib.smfft_ct_external_benchmark(a, b, n, TOTAL // n, 0, 1, t), 2 * nbytes)
    rn = 2 * n
    if 512 <= rn <= 4096:
        nffts = TOTAL // rn
        for inv in (0, 1):
            sweep(f"{'C2R' if inv else 'R2C'} real N={rn}", lambda t: sm.lib.smfft_rc_external_benchmark(a, b, rn, nffts, inv, t), nbytes)
