"""Summarise a rocprofv3 --pmc csv directory: per kernel, mean counter values (and per-FFT values
when the kernel name carries the transform length and the batch is the 4 GiB sweep batch)."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
total = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 29)
agg = collections.defaultdict(list)
# (gpurun merges every call's files into gpurun_out/: only the LATEST run's csv of a directory is read)
import os
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(f)):
        # the reference-shaped kernels of include/smfft/smfft_device_functions.hpp carry the library's kernels' NAMES with two arguments:
        # kept apart by their signature
        full = r["Kernel_Name"]
        name = full.split("(")[0]
        if re.search(r"\(HIP_vector_type<float, 2u>( const)?\*, HIP_vector_type<float, 2u>\*\)$", full):
            name += " [reference shape <<<nFFTs, N/4>>>]"
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({k[0] for k in agg})
for name in names:
    m = re.search(r"FFT_(\d+)", name)
    n = int(m.group(1)) if m else None
    mult = "multiple" in name
    row = []
    for (k, c), v in sorted(agg.items()):
        if k != name:
            continue
        mean = sum(v) / len(v)
        if n:
            nfft = total // n
            if mult:
                nfft = (nfft // 100) * 100
            row.append(f"{c}={mean:.4g} ({mean / nfft:.4g}/FFT)")
        else:
            row.append(f"{c}={mean:.4g}")
    print(name.replace("void ", ""), "|", "  ".join(row))
