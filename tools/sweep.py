"""Timing sweep on the GPU box (no torch): external/multiple paths over sizes, variants and grid caps.
usage: python tools/sweep.py [--sizes 1024,...] [--caps 0,768,1024] [--paths external,multiple] [--total-log2 29]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024")
ap.add_argument("--caps", default="0")
ap.add_argument("--paths", default="external")
ap.add_argument("--variants", default="f1")   # f1 = forward reorder, f0 = forward noreorder, i1, i0
ap.add_argument("--total-log2", type=int, default=29)
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--family", default="ct")
args = ap.parse_args()

total = 1 << args.total_log2
nbytes = total * 8
rng = np.random.default_rng(0)
chunk = rng.random(1 << 22, dtype=np.float32)
a, b = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
for off in range(0, nbytes, chunk.nbytes):
    sm.lib.smfft_memcpy_h2d(a.ptr + off, chunk.ctypes.data, min(chunk.nbytes, nbytes - off))
sm.FFT_init()
for n in [int(s) for s in args.sizes.split(",")]:
    nffts = total // n if args.family != "rc" else total * 2 // n
    for path in args.paths.split(","):
        for var in args.variants.split(","):
            inv, reo = var[0] == "i", var[1] == "1"
            for cap in [int(c) for c in args.caps.split(",")]:
                sm.lib.smfft_set_grid_cap(cap)
                f = sm.FFT_external_benchmark if path == "external" else sm.FFT_multiple_benchmark
                for _ in range(3):
                    f(a.ptr, b.ptr, n, nffts, inv, reo, args.family)
                ts = []
                for _ in range(args.rounds):
                    rc, ms = f(a.ptr, b.ptr, n, nffts, inv, reo, args.family)
                    ts.append(ms)
                ts.sort()
                med, mn = ts[len(ts) // 2], ts[0]
                if path == "external":
                    gb = 2 * nbytes / 1e9
                    print(f"{args.family} N={n:5d} {path:8s} {var} cap={cap:6d}: median {med:.4f} ms ({gb/med*1e3:7.1f} GB/s)  min {mn:.4f} ms ({gb/mn*1e3:7.1f} GB/s)", flush=True)
                else:
                    cnt = (nffts // 100) * 100
                    print(f"{args.family} N={n:5d} {path:8s} {var} cap={cap:6d}: median {med:.4f} ms ({cnt/med*1e3:.3e} FFT/s)  min {mn:.4f} ms", flush=True)
