"""The per-length tables of DESIGN.md sections 5.2 / 5.4 from profiles/r06_bench_detail.json (the round's final evidence call):
    python tools/design_tables_r06.py            prints the rows; --splice rewrites the two tables of DESIGN.md in place"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
c = d["configs"]
c3, by = c["config3_multiple"], c["reference_contract"]["by_length"]
r5 = {"256": "0.47 / 0.51", "512": "0.54 / 0.57", "1024": "0.52 / 0.56", "2048": "0.46 / 0.50", "4096": "0.36 / 0.47"}
rows52, rows54 = [], []
for k in ("32", "64", "128", "256", "512", "1024", "2048", "4096"):
    n, r, b = int(k), c3[k], by[k]
    frac = lambda v: v * 5 * n * math.log2(n) / 1e12 / 157.3      # noqa: E731
    fused = f"{r['reorder']['frac_fp32_peak']:.3f} / {r['noreorder']['frac_fp32_peak']:.3f}"
    if k == "32":
        fused += f" ({r['reorder']['FFT/s'] / 1e11:.2f}·10¹¹ FFT/s)"
    if k == "1024":
        fused += f" ({r['reorder']['FFT/s'] / 1e9:.2f} / {r['noreorder']['FFT/s'] / 1e9:.2f}·10⁹ FFT/s; V100 reference 1.05 / 0.85·10⁸)"
    percall = f"{r['reorder']['unfused']['frac_fp32_peak']:.3f} / " + (f"{r['noreorder']['unfused']['frac_fp32_peak']:.3f}" if n <= 64 else "= fused")
    con = f"{frac(b['reorder']['in_lds_FFT/s']):.3f} / {frac(b['noreorder']['in_lds_FFT/s']):.3f}"
    ratio = f"{b['reorder']['in_lds_ratio_to_compact']:.2f} / {b['noreorder']['in_lds_ratio_to_compact']:.2f}"
    if "wave64" in b["reorder"]:
        w = b["reorder"]["wave64"]["in_lds_ratio_to_compact"], b["noreorder"]["wave64"]["in_lds_ratio_to_compact"]
        con += f" — `_wave64` classes {w[0] * r['reorder']['frac_fp32_peak']:.2f} / {w[1] * r['noreorder']['frac_fp32_peak']:.2f}"
        ratio += f" — {w[0]:.2f} / {w[1]:.2f}"
    else:
        con, ratio = "**" + con + "**", "**" + ratio + f"** ({r5[k]})"
    if k == "1024":
        con += f" ({b['reorder']['in_lds_FFT/s'] / 1e9:.2f} / {b['noreorder']['in_lds_FFT/s'] / 1e9:.2f}·10⁹)"
    sat = f"{r['reorder']['saturating_batch']['frac_fp32_peak']:.3f} / {r['noreorder']['saturating_batch']['frac_fp32_peak']:.3f}"
    rows52.append(f"| {k} | {fused} | {percall} | {con} | {ratio} | {sat} |")
    ext = f"{b['reorder']['external_ratio_to_tiled']:.2f} / {b['noreorder']['external_ratio_to_tiled']:.2f}"
    usr = f"{b['reorder']['user_kernel_external_ratio_to_tiled']:.2f} / {b['noreorder']['user_kernel_external_ratio_to_tiled']:.2f}"
    name = k
    if "wave64" in b["reorder"]:
        ext += f" — {b['reorder']['wave64']['external_ratio_to_tiled']:.2f}"
        if k == "32":
            name = "32 upstream shape (32-thread blocks) — `_wave64`"
    v = b["reorder"]["in_lds_FFT/s"]
    e = int(math.floor(math.log10(v)))
    rows54.append(f"| {name} | {ext} | {usr} | {ratio} | {v / 10 ** e:.2f}·10{ {8: '⁸', 9: '⁹', 10: '¹⁰'}[e] } |")
print("\n".join(rows52))
print()
print("\n".join(rows54))
if "--splice" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    i, j = s.index("| 32 | 0."), s.index('("= fused": the planar no-reorder kernels')
    s = s[:i] + "\n".join(rows52) + "\n\n" + s[j:]
    i, j = s.index("| 32 upstream shape (32-thread blocks) — `_wave64` |"), s.index("In one process against the round-5 header on the same buffers")
    s = s[:i] + "\n".join(rows54) + "\n\n" + s[j:]
    open(p, "w").write(s)
