"""The MI355X counterpart of the reference's one published table (README.md:79-91 of KAdamek/SMFFT: `FFT.exe` outputs at the 4 GB
batches, 8 lengths x Cooley-Tukey no reorder / reorder, Stockham, vendor FFT; "multiple [external]" in ms), produced by the
HARNESS PROGRAMS themselves -- the three FFT.exe of harness/, 20 kernel executions each, as a user of the reference would run them:
    python tools/readme_table.py [out.md] [--runs 20] [--sizes 32,64,...]
Also records each run's own verdict line(s) (PASSED / FAILED and, for FAILED, the attribution lines of harness_common.h)."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V100 = {   # README.md:84-91 of the reference (V100 32GB, CUDA 10): multiple [external] ms
    32: ("2.04 [10.45]", "2.43 [10.45]", "NA", "NA [10.52]"), 64: ("2.54 [10.45]", "3.93 [10.45]", "NA", "NA [10.45]"),
    128: ("3.45 [10.47]", "4.89 [10.47]", "NA", "NA [10.47]"), 256: ("3.95 [10.46]", "5.63 [10.46]", "6.70 [10.46]", "NA [10.55]"),
    512: ("4.43 [10.40]", "6.07 [10.40]", "6.77 [10.39]", "NA [10.52]"), 1024: ("5.01 [10.41]", "6.16 [10.41]", "6.90 [10.41]", "NA [10.50]"),
    2048: ("5.77 [10.50]", "7.72 [10.50]", "7.63 [10.53]", "NA [10.49]"), 4096: ("6.80 [10.75]", "9.47 [10.75]", "8.95 [11.52]", "NA [10.65]")}
LABEL = {32: "16M", 64: "8M", 128: "4M", 256: "2M", 512: "1M", 1024: "524k", 2048: "262k", 4096: "131k"}

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default="")
ap.add_argument("--runs", type=int, default=20)
ap.add_argument("--sizes", default="32,64,128,256,512,1024,2048,4096")
ap.add_argument("--warmup-ms", type=float, default=40.0, help="SMFFT_WRAPPER_WARMUP_MS of the runs (0 = upstream's behaviour, the library's default since round 6)")
args = ap.parse_args()
sizes = [int(v) for v in args.sizes.split(",")]
# The wrappers time their nRuns launches directly behind the upload, as upstream does -- on a device whose clocks have fallen back, so that 20
# in-LDS launches are mostly ramp.  THIS table wants the settled figures: it asks the wrappers for 40 ms of untimed launches of the same kernel
# first (opt-in since round 6) and says so in its header; --warmup-ms 0 gives upstream's cold figures.
env = dict(os.environ, SMFFT_SEED="20200720", SMFFT_WRAPPER_WARMUP_MS=str(args.warmup_ms))


def run(prog, *a):
    p = subprocess.run([os.path.join(ROOT, "harness", prog)] + [str(v) for v in a], capture_output=True, text=True, env=env, cwd=ROOT)
    text = re.sub(r"\x1b\[[0-9;]*m", "", p.stdout)
    sm = re.search(r"SH FFT normal = ([0-9.]+) ms; SM FFT multiple times = ([0-9.]+) ms", text)
    vendor = re.search(r"cuFFT time = ([0-9.]+) ms", text)
    verdicts = [l.strip() for l in text.splitlines() if "FFT test:" in l or "Worst element" in l or "Distance from the fp64" in l or "no verification" in l]
    assert p.returncode == 0 and sm, (prog, a, p.returncode, text[-2000:], p.stderr[-2000:])
    return float(sm.group(1)), float(sm.group(2)), (float(vendor.group(1)) if vendor else None), verdicts


rows, notes = [], []
for n in sizes:
    nffts = (1 << 29) // n
    ext0, mul0, _, v0 = run("FFT_CooleyTukey_C2C.exe", n, nffts, args.runs, 0, 0)
    ext1, mul1, ven, v1 = run("FFT_CooleyTukey_C2C.exe", n, nffts, args.runs, 0, 1)
    if n >= 128:
        exts, muls, _, vs = run("FFT_Stockham_C2C.exe", n, nffts, args.runs)
        st = f"{muls:.3f} [{exts:.3f}]"
    else:
        st, vs = "NA", []
    rows.append((n, f"{mul0:.3f} [{ext0:.3f}]", f"{mul1:.3f} [{ext1:.3f}]", st, f"NA [{ven:.3f}]"))
    for name, v in (("Cooley-Tukey", v0), ("Cooley-Tukey reorder", v1), ("Stockham", vs)):
        for line in v:
            notes.append(f"| {n} | {name} | {line} |")
    print(rows[-1], flush=True)

lines = ["# `FFT.exe` at the reference's README batches on MI355X (harness programs, %d kernel executions each)" % args.runs, "",
         "Layout of `README.md:82-91` of KAdamek/SMFFT: time in milliseconds, first the `FFT_multiple_benchmark` time (the first nFFTs/100 FFTs transformed 100 times in LDS),",
         "in square brackets the `FFT_external_benchmark` time (device-memory bound: 4 GiB in + 4 GiB out).  Input 4 GiB; the number of FFTs in square brackets.",
         "The vendor column is hipFFT on the same buffers (one execution, as upstream's `GPU_cuFFT`).  V100 columns: the reference's published table.",
         ("`SMFFT_WRAPPER_WARMUP_MS=%g`: the wrappers ran that many milliseconds of untimed launches of the same kernel before the timed ones (settled clocks)." % args.warmup_ms) if args.warmup_ms > 0
         else "`SMFFT_WRAPPER_WARMUP_MS=0`: timed directly behind the upload, as upstream (cold clocks: the in-LDS figures include the ramp).", "",
         "FFT size | Cooley-Tukey | Cooley-Tukey reorder | Stockham | hipFFT | V100: Cooley-Tukey | V100: reorder | V100: Stockham | V100: cuFFT",
         "-------- | ------------ | -------------------- | -------- | ------ | ------------------ | ------------- | -------------- | -----------"]
for n, a, b, c, d in rows:
    v = V100[n]
    lines.append(f"{n} [{LABEL[n]}] | {a} | {b} | {c} | {d} | {v[0]} | {v[1]} | {v[2]} | {v[3]}")
lines += ["", "What each run printed about its own check (upstream's metric: two fp32 results compared under the absolute bound `max_error = 1e-4`, `FFT.c:12,23-49`;",
          "when it reports errors the harness adds which side is how far from an fp64 DFT of the same input, `harness_common.h` `harness_attribute`):", "",
          "| N | program | line |", "|---|---|---|"] + notes
text = "\n".join(lines) + "\n"
if args.out:
    open(args.out, "w").write(text)
print(text)
