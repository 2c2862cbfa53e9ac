#!/usr/bin/env python3
"""ISA view of one smfft_inst.hip build (CPU only: hipcc cross-compiles): per kernel, the instruction mix of the
hottest loop (the innermost loop with the most instructions) and, with --dump, its text.
    python tools/isa_loop.py 1024 [--kernel multiple] [--dump] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    n = args.pop(0) if args and args[0].isdigit() else "1024"
    pick, dump = "multiple", False
    extra = []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            pick = args.pop(0)
        elif a == "--dump":
            dump = True
        else:
            extra.append(a)
    src = os.path.join(ROOT, "smfft_amd", "csrc", "smfft_inst.hip")
    out = f"/tmp/smfft_isa_{n}.s"
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from inst_flags import part_flags
    # the object the picked kernel ships in: part 2 (in-LDS kernels, with the Makefile's MULT_FLAGS_<N>) unless an external kernel is asked for
    part = 1 if "external" in pick else 2
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", f"-DSMFFT_N={n}", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", src, "-o", out] + part_flags(n, part) + extra
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-4000:])
        raise SystemExit(p.returncode)
    text = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pick not in name:
            continue
        lines = body.split("\n")
        # basic blocks: label lines ".LBBx_y:"; a loop = a block range closed by a backward branch to its label
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loops.append((labels[mm.group(1)], i))
        if not loops:
            continue
        # innermost = contains no other loop; take the one with most instructions
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        a, b = max(inner, key=lambda lp: lp[1] - lp[0])
        mix = {}
        for l in lines[a:b + 1]:
            t = l.strip()
            if not t or t.startswith((";", ".")):
                continue
            op = t.split()[0]
            key = op
            if op.startswith("v_") and not op.startswith(("v_permlane", "v_mov_b32_dpp", "v_pk_")):
                key = "v_dpp" if "dpp" in t else "valu"
            if op == "s_waitcnt":
                key = "s_waitcnt " + " ".join(t.split()[1:])
            mix[key] = mix.get(key, 0) + 1
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        print(f"== {short}: hot loop {b - a + 1} lines")
        print("   " + ", ".join(f"{k}: {v}" for k, v in sorted(mix.items(), key=lambda kv: -kv[1])))
        if dump:
            print("\n".join(lines[a:b + 1]))


if __name__ == "__main__":
    main()
