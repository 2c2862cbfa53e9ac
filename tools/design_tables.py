"""Prints the config-3 table of DESIGN.md section 5.2 from profiles/<tag>_bench.json (round-2 fractions in brackets):
python tools/design_tables.py [r03]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
c = d["configs"]["config3_multiple"]
sup = str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹")
r2 = {"32": ("0.39", "0.41"), "64": ("0.34", "0.36"), "128": ("0.36", "0.36"), "256": ("0.40", "0.40"), "512": ("0.39", "0.39"),
      "1024": ("0.37", "0.38"), "2048": ("0.36", "0.36"), "4096": ("0.39", "0.37")}


def sci(x):
    e = len(str(int(x))) - 1
    return ("%.2f" % (x / 10 ** e)) + "·10" + str(e).translate(sup)


def bold(s, cond):
    return "**%s**" % s if cond else s


for n in ("32", "64", "128", "256", "512", "1024", "2048", "4096"):
    a, b = c[n]["noreorder"], c[n]["reorder"]
    print("| %s | %s | %s (%s) | %.3f | %s | %.3f (%s) | %.3f |" % (
        n, bold(sci(a["FFT/s"]), n == "1024"), bold("%.3f" % a["frac_fp32_peak"], n in ("64", "2048", "4096")), r2[n][0], a["saturating_batch"]["frac_fp32_peak"],
        bold(sci(b["FFT/s"]), n == "1024"), b["frac_fp32_peak"], r2[n][1], b["saturating_batch"]["frac_fp32_peak"]))
