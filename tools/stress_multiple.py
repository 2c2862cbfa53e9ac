"""Stress of the in-LDS (`multiple`) path's schedules: seeded random (program, N, batch, NREUSES, number of co-resident
workgroups the balanced grid is forced to) -- the balanced schedule (chains cut and parked between workgroups, priorities
rotating) must give the BITS of one chain per workgroup, whatever the geometry.  One-off tool; the fixed cases of
tests/test_gpu_parity.py (test_*_balanced_schedule_is_bit_identical) are the regression form.
    python tools/stress_multiple.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm  # noqa: E402

C2C = [32, 64, 128, 256, 512, 1024, 2048, 4096]
RC = [512, 1024, 2048, 4096]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(seed0)
MAXB = 48 << 20
din, dout = sm.DeviceBuffer(MAXB), sm.DeviceBuffer(MAXB)
# scaled by 2^-100 so that N^(NREUSES/2) stays finite for most of the cases drawn (a result full of NaNs compares equal whatever happened)
host = ((rng.random(MAXB // 4, dtype=np.float32) - 0.5) * np.ldexp(np.float32(1.0), -100)).astype(np.float32)
sm.lib.smfft_memcpy_h2d(din.ptr, host.ctypes.data, MAXB)
tally = {}
finite_words = words = 0
for case in range(ncases):
    prog = ("ct_reorder", "ct_noreorder", "ct_unfused", "st", "r2c", "c2r")[int(rng.integers(0, 6))]
    if prog in ("r2c", "c2r"):
        n = int(rng.choice(RC))
        tile_ffts, bytes_per = max(1, 1024 // (n // 2)), n * 4
    else:
        n = int(rng.choice(C2C[1:] if prog == "ct_unfused" else C2C[3:] if prog == "st" else C2C))
        tile_ffts, bytes_per = max(1, 1024 // n), n * 8
    tiles = int(rng.integers(1, 40))
    slots = max(1, tiles * tile_ffts - int(rng.integers(0, tile_ffts)))
    unit = 400 if (n == 32 and prog.startswith("ct")) else 200 if (n == 64 and prog.startswith("ct")) else 100
    nffts = min(slots * 100 + int(rng.integers(0, 100)), MAXB // bytes_per)
    nffts = max(nffts, unit)
    reuses = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 100]))
    forced = [int(g) for g in rng.choice([1, 2, 3, 5, 7, 16, 64, 1000], size=2, replace=False)]
    fam, path, inv, reo = {"ct_reorder": (0, 1, 0, 1), "ct_noreorder": (0, 1, 0, 0), "ct_unfused": (0, 2, 0, 1), "st": (1, 1, 1, 1),
                           "r2c": (2, 1, 0, 1), "c2r": (2, 1, 1, 1)}[prog]
    if prog.startswith("ct") and rng.integers(0, 2):
        inv = 1
    sm.lib.smfft_set_nreuses(reuses)
    nbytes = nffts * bytes_per

    def run():
        sm.lib.smfft_memset(dout.ptr, 0xFF, nbytes)
        rc = sm.lib.smfft_launch(fam, path, din.ptr, dout.ptr, n, nffts, inv, reo, None)
        assert rc == 0 and sm.lib.smfft_synchronize() == 0, (case, prog, n, nffts, reuses, rc)
        return dout.to_host(np.uint32, (nbytes // 4,))
    sm.lib.smfft_set_multiple_balance(0)
    want = run()
    for g in forced + [-1]:
        sm.lib.smfft_set_multiple_balance(g)
        for rot in (1, 0):
            sm.lib.smfft_set_multiple_rotation(rot)
            got = run()
            # (NaN against NaN counts as equal whatever the sign bit: the lane engines of N = 32 / 64 keep some lanes negated between
            #  applications, and which of two NaNs an addition returns depends on the operand order of that copy of the loop body)
            bad = np.flatnonzero((got != want) & ~(np.isnan(got.view(np.float32)) & np.isnan(want.view(np.float32))))
            assert bad.size == 0, (case, prog, n, nffts, reuses, g, rot, bad.size, int(bad[0]))
    written = want != 0xFFFFFFFF                      # (the launch writes the first nFFTs / 100 slots; the rest keeps the fill)
    finite_words += int(np.count_nonzero(np.isfinite(want.view(np.float32)[written])))
    words += int(np.count_nonzero(written))
    sm.lib.smfft_set_multiple_rotation(-1)
    tally[prog] = tally.get(prog, 0) + 1
sm.lib.smfft_set_nreuses(0)
sm.lib.smfft_set_multiple_balance(-1)
print(f"{ncases} random in-LDS launches from seed {seed0}, each on 3 balanced grids x rotation on / off: bit-identical to one chain per workgroup ({finite_words / words:.0%} of the words the launches wrote are finite); {tally}")
