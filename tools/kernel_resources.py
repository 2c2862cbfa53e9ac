#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, SGPRs, waves/SIMD, LDS bytes, scratch) of one smfft_inst.hip build:
   python tools/kernel_resources.py 1024 [extra hipcc flags...]      (CPU only: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = sys.argv[1] if len(sys.argv) > 1 else "1024"
    extra = sys.argv[2:]
    src = os.path.join(ROOT, "smfft_amd", "csrc", "smfft_inst.hip")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from inst_flags import part_flags
    report = ""
    for part in (1, 2):      # the product's two objects per length, each with its flags (smfft_amd/csrc/Makefile)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", f"-DSMFFT_N={n}", "-I" + os.path.join(ROOT, "include"),
               "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + part_flags(n, part) + extra
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-4000:])
            raise SystemExit(p.returncode)
        report += p.stderr
    rows, cur = [], None
    for line in report.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            cur = {"name": name}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print(f"{'kernel':70s} vgpr sgpr occ   lds scratch")
    for r in rows:
        print(f"{r['name'][:70]:70s} {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('occ', -1):3d} {r.get('lds', -1):5d} {r.get('scratch', -1):4d}")


if __name__ == "__main__":
    main()
