"""Prints the tables of DESIGN.md section 5.2 - 5.4 from profiles/<tag>_bench_detail.json (the full record behind a bench line):
python tools/design_tables.py [r04] [--patch]      (--patch: rewrite those tables in DESIGN.md in place)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r04"
d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_detail.json")))
c = d["configs"]["config3_multiple"]
by_len = d["configs"]["reference_contract"]["by_length"]
sup = str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹")
r3 = {"32": ("0.398", "0.413"), "64": ("0.428", "0.493"), "128": ("0.481", "0.524"), "256": ("0.507", "0.542"), "512": ("0.437", "0.447"),
      "1024": ("0.465", "0.481"), "2048": ("0.471", "0.476"), "4096": ("0.423", "0.441")}       # round 3's README-batch fractions (no reorder, reorder)
SIZES = ("32", "64", "128", "256", "512", "1024", "2048", "4096")


def sci(x):
    e = len(str(int(x))) - 1
    return ("%.2f" % (x / 10 ** e)).rstrip("0").rstrip(".") + "·10" + str(e).translate(sup)


print("what one call costs (FFT/s): contract (wave64 class) | unfused | fused")
for n in SIZES:
    b = c[n]["reorder"]
    con = by_len[n]["reorder"]
    w = con.get("wave64")
    print("| %s | %s%s | %s | %s |" % (n, sci(con["in_lds_FFT/s"]), " (%s)" % sci(w["in_lds_FFT/s"]) if w else "",
                                     sci(b["unfused"]["FFT/s"]) if "unfused" in b else "= fused", sci(b["FFT/s"])))
print("\nconfig 3: | N | reorder FFT/s | frac (round 3) | oldest-first | saturating | no reorder FFT/s | frac (round 3) | saturating |")
for n in SIZES:
    a, b = c[n]["noreorder"], c[n]["reorder"]
    print("| %s | %s | %.3f (%s) | %.3f | %.3f | %s | %.3f (%s) | %.3f |" % (
        n, sci(b["FFT/s"]), b["frac_fp32_peak"], r3[n][1], b["one_chain_per_workgroup_oldest_first"]["frac_fp32_peak"], b["saturating_batch"]["frac_fp32_peak"],
        sci(a["FFT/s"]), a["frac_fp32_peak"], r3[n][0], a["saturating_batch"]["frac_fp32_peak"]))
print("\ncontract: | N | external two-argument / user kernel (reorder) | in-LDS ratio (reorder / no reorder) | in-LDS FFT/s |")
for n in SIZES:
    r, nr = by_len[n]["reorder"], by_len[n]["noreorder"]
    print("| %s | %.2f / %.2f | %.2f / %.2f | %s |" % (n, r["external_ratio_to_tiled"], r["user_kernel_external_ratio_to_tiled"], r["in_lds_ratio_to_compact"], nr["in_lds_ratio_to_compact"], sci(r["in_lds_FFT/s"])))
    if "wave64" in r:
        w, wn = r["wave64"], nr["wave64"]
        print("| %s _wave64 | %.2f / %.2f | %.2f / %.2f | %s |" % (n, w["external_ratio_to_tiled"], w["user_kernel_external_ratio_to_tiled"], w["in_lds_ratio_to_compact"], wn["in_lds_ratio_to_compact"], sci(w["in_lds_FFT/s"])))
c4 = d["configs"]["config4_r2c_c2r_external"]
if "in_lds" in c4.get("512", {}):
    print("\nR2C / C2R in-LDS (4 GiB of reals, ms per 100 applications): | real N | R2C | C2R | C2C of the same complex length | R2C / C2R over it |")
    for n in ("512", "1024", "2048", "4096"):
        r = c4[n]["in_lds"]
        print("| %s | %.3f | %.3f | %.3f | +%.0f / +%.0f %% |" % (n, r["r2c_ms"], r["c2r_ms"], r["c2c_same_complex_length_ms"], 100 * r["r2c_over_c2c"], 100 * r["c2r_over_c2c"]))


def patch_design():
    """the three tables of DESIGN.md 5.2 / 5.4 and the in-LDS table of 5.3, rewritten from the record (headers and prose stay)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.abspath(__file__), tag], capture_output=True, text=True).stdout
    sec = [[l for l in blk.split("\n")[1:] if l.startswith("|")] for blk in out.split("\n\n")]
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()

    def replace_rows(text, header_start, end_marker, rows):
        a = text.index(header_start)
        hdr_end = text.index("\n", text.index("|---|", a)) + 1
        b = text.index(end_marker, hdr_end)
        return text[:hdr_end] + "\n".join(rows) + "\n\n" + text[b:]
    rows = []
    for l in sec[0]:
        c = [x.strip() for x in l.strip("|").split("|")]
        if c[0] == "32":
            c[2] = "= fused (the N = 32 kernel forwards its registers in either build)"
        if c[0] == "1024":
            c = [c[0]] + ["**" + x + "**" for x in c[1:]]
        rows.append("| " + " | ".join(c) + " |")
    s = replace_rows(s, "| N | the reference's contract", "(V100 reference, README.md:84-91", rows)
    s = replace_rows(s, "| N | reorder FFT/s | frac (round 3) | oldest-first | saturating |", "(Other runs of the round, other boxes:", sec[1])
    rows = []
    for l in sec[2]:
        c = [x.strip() for x in l.strip("|").split("|")]
        if "_wave64" in c[0]:
            n = int(c[0].split()[0])
            c[0] = "%d `_wave64` (64-thread blocks, %d transforms each)" % (n, 256 // n)
        elif int(c[0]) <= 128:
            n = int(c[0])
            c[0] = "%d upstream shape (32-thread blocks, %d transform%s each)" % (n, 128 // n, "s" if 128 // n > 1 else "")
        rows.append("| " + " | ".join(c) + " |")
    s = replace_rows(s, "| N | external, two-argument kernel / user's fill-call-drain kernel (reorder), ratio to tiled |", "The `_wave64` classes (`include/smfft/SM_FFT_parameters.hpp`) change ONE number", rows)
    if len(sec) > 3 and "| real N | R2C ms | C2R ms |" in s:
        s = replace_rows(s, "| real N | R2C ms | C2R ms |", "(`configs.config4_r2c_c2r_external.<N>.in_lds`", sec[3])
    open(path, "w").write(s)
    print("DESIGN.md tables rewritten from profiles/%s_bench_detail.json" % tag)


if "--patch" in sys.argv:
    patch_design()
