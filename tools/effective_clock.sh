#!/bin/bash
# Effective shader clock of a kernel = GRBM_GUI_ACTIVE (cycles the GPU is busy, counted at the shader clock) / the
# kernel's duration, per dispatch, from one rocprofv3 counter pass: the FMA-chain calibration kernel (tools/microbench/
# clock_ratio), the in-LDS kernels and the external kernel.  Usage (on the GPU box): tools/effective_clock.sh > out.txt
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/effclk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/chains -- $R/tools/microbench/clock_ratio > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/fft -- python3 $R/tools/sweep.py --sizes 1024 --paths multiple,external --variants f1,f0 --rounds 5 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("chains", "fft"):
    trace = {}
    for f in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size", 0) or 0))
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            name, dur, grid = trace.get(r["Dispatch_Id"], (r.get("Kernel_Name", "?"), 0, 0))
            if dur > 20000:
                acc[(name.split("(")[0][:70], grid)].append(float(r["Counter_Value"]) / dur)
    for (name, grid), v in sorted(acc.items()):
        v.sort()
        print(f"{sub}: {name} grid {grid}: GRBM_GUI_ACTIVE / ns = {v[len(v) // 2]:.3f} (median of {len(v)} dispatches; x 1 GHz if the counter is one instance)")
PY
