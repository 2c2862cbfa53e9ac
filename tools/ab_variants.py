"""A/B/C... of several builds of libsmfft_amd.so in ONE process on the SAME buffers, rounds interleaved.

    python tools/ab_variants.py base=smfft_amd/libsmfft_amd.so v1=smfft_amd/libsmfft_amd_v1.so \\
           [--sizes 1024,4096] [--paths multiple,external,rc] [--mult 1,10] [--plain]

Build a variant with
    make -C smfft_amd/csrc LIB=../libsmfft_amd_v1.so OBJDIR=build_v1 EXTRA_HIPFLAGS="-DSMFFT_NT=0" ../libsmfft_amd_v1.so      (or any other build: another commit's library, tools/build_variant.sh)
multiple: FFT_multiple_benchmark (README batch 2^29/N FFTs, x the batch multipliers), reorder and no-reorder.
external: FFT_external_benchmark forward, reorder and no-reorder, 4 GiB in + 4 GiB out.
rc:       R2C and C2R external at real N = 2 * size (2 GiB each way).
Buffers: smfft_malloc_pair of the first library unless --plain (two plain allocations)."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+", help="name=path pairs")
ap.add_argument("--sizes", default="1024")
ap.add_argument("--paths", default="multiple")
ap.add_argument("--mult", default="1")
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--plain", action="store_true")
args = ap.parse_args()

names, libs = [], []
for spec in args.libs:
    name, path = spec.split("=", 1)
    path, _, opts = path.partition(":")          # name=path[:balance=K]
    lib = ctypes.CDLL(os.path.abspath(path))
    for opt in filter(None, opts.split(":")):
        key, val = opt.split("=")
        if key == "balance":
            lib.smfft_set_multiple_balance(int(val))
    vp, i, dp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
    lib.smfft_ct_multiple_benchmark.argtypes = [vp, vp, i, i, i, i, dp]
    lib.smfft_ct_external_benchmark.argtypes = [vp, vp, i, i, i, i, dp]
    lib.smfft_rc_external_benchmark.argtypes = [vp, vp, i, i, i, dp]
    names.append(name)
    libs.append(lib)
sizes = [int(v) for v in args.sizes.split(",")]
mults = [int(v) for v in args.mult.split(",")]
TOTAL = 1 << 29
nbytes = TOTAL * 8
if args.plain:
    A, B = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
    a, b = A.ptr, B.ptr
else:
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) == 0
    a, b = pa.value, pb.value
chunk = np.random.default_rng(0).random(1 << 22, dtype=np.float32)
sm.lib.smfft_memcpy_h2d(a, chunk.ctypes.data, chunk.nbytes)
filled = chunk.nbytes
while filled < nbytes:
    step = min(filled, nbytes - filled)
    sm.lib.smfft_memcpy_d2d(a + filled, a, step)
    filled += step


def compare(label, call, work, unit):
    res = [[] for _ in libs]
    for rnd in range(args.rounds + 2):
        for k, lib in enumerate(libs):
            t = ctypes.c_double(0)
            call(lib, ctypes.byref(t))
            if rnd >= 2:
                res[k].append(t.value)
    cells = []
    for k in range(len(libs)):
        r = sorted(res[k])
        med = r[len(r) // 2]
        cells.append(f"{names[k]} {med:.4f} ms {work / med * 1e3:.4g} {unit}")
    print(f"{label}: " + " | ".join(cells), flush=True)


for n in sizes:
    for path in args.paths.split(","):
        if path == "multiple":
            for reo in (1, 0):
                for m in mults:
                    nffts = min(TOTAL // n * m, 2**31 - 1)
                    done = (nffts // 400 * 400) if n == 32 else (nffts // 200 * 200) if n == 64 else (nffts // 100 * 100)
                    if nffts * n > TOTAL and m > 1:
                        pass   # the multiple path touches only the first nFFTs/100 slots: larger "batches" stay inside the buffers
                    compare(f"multiple N={n} reorder={reo} x{m}", lambda lib, t: lib.smfft_ct_multiple_benchmark(a, b, n, nffts, 0, reo, t), done, "FFT/s")
        elif path == "external":
            for reo in (1, 0):
                compare(f"external N={n} reorder={reo}", lambda lib, t: lib.smfft_ct_external_benchmark(a, b, n, TOTAL // n, 0, reo, t), 2 * nbytes / 1e9, "GB/s")
        elif path == "rc":
            rn = 2 * n
            if rn < 512 or rn > 4096:
                continue
            nffts = (TOTAL // 2) * 2 // rn   # 2 GiB of reals
            rbytes = rn * nffts * 4
            compare(f"R2C real N={rn}", lambda lib, t: lib.smfft_rc_external_benchmark(a, b, rn, nffts, 0, t), 2 * rbytes / 1e9, "GB/s")
            # C2R reads the packed spectra from the READ buffer's second half and writes into the WRITE buffer's second half
            # (same placement relation as every other transform here); the R2C input in the first half of `a` stays intact
            sm.lib.smfft_memcpy_d2d(a + nbytes // 2, b, rbytes)
            compare(f"C2R real N={rn}", lambda lib, t: lib.smfft_rc_external_benchmark(a + nbytes // 2, b + nbytes // 2, rn, nffts, 1, t), 2 * rbytes / 1e9, "GB/s")
