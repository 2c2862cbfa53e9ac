"""Summarises gpurun_out/<tag>/ (written by tools/profile_round.sh) into the tracked profiles/ files.
usage: python tools/collect_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the latest run's file."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]


def counters(sub):
    agg = collections.defaultdict(list)
    for f in newest(os.path.join(src, sub, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


# 1. kernel stats of the bench command
for f in newest(os.path.join(src, "stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, f"profiles/{tag}_bench_kernel_stats.csv")
for name in ("bench.json", "bench_detail.json", "bench_under_rocprof.json", "configs.json", "mult_saturation.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), f"profiles/{tag}_{name}")

# 2. HBM traffic
try:
    import subprocess
    HEAD = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    HEAD = None
out = {"commit": HEAD, "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports 1/2 of a coalesced streaming read (MI355X_MICROARCH.md, HBM)",
       "calibration": {"note": "tools/microbench/membw 268435456 under the same two passes: tile8 = the engine's access shape (8 B/lane, 512 B per wave instruction) moves 2097152 KiB each way"},
       "kernels": {}}
for sub, key in (("pmc_calib_fetch", "FETCH_SIZE"), ("pmc_calib_write", "WRITE_SIZE")):
    for (k, c), v in counters(sub).items():
        if k in ("tile8", "tile16", "copy16_nt") and c == key:
            out["calibration"].setdefault(k, {})[key + "_KiB_mean"] = sum(v) / len(v)
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for (k, c), v in counters(sub).items():
        if ("SMFFT" in k or "FFT_GPU" in k) and c == key:
            out["kernels"].setdefault(k, {})[key + "_KiB_mean"] = sum(v) / len(v)
            out["kernels"][k][key + "_launches"] = len(v)
k = "void SMFFT_DIT_external<FFT_1024_forward>"
if k in out["kernels"] and len(out["kernels"][k]) >= 4:
    e = out["kernels"][k]
    e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024
    e["algorithmic_bytes_per_launch"] = 2 * 1024 * 524288 * 8
    e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
    for rk, e2 in out["kernels"].items():       # config 4: real N = 2048, 262144 FFTs: 2 GiB in, 2 GiB out per launch
        if "FFT_GPU_R2C_C2R_external<FFT_1024" in rk and len(e2) >= 4:
            e2["hbm_bytes_per_launch"] = (2 * e2["FETCH_SIZE_KiB_mean"] + e2["WRITE_SIZE_KiB_mean"]) * 1024
            e2["algorithmic_bytes_per_launch"] = 2 * 2048 * 262144 * 4
            e2["traffic_over_algorithmic"] = e2["hbm_bytes_per_launch"] / e2["algorithmic_bytes_per_launch"]
    json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print("traffic/algorithmic =", e["traffic_over_algorithmic"])

# 3. LDS and wave-state counters of the sweep
rows = {}
total = 1 << 29
both = dict(counters("pmc_lds"))
both.update(counters("pmc_wait"))
for (k, c), v in both.items():
    m = re.search(r"(SMFFT_DIT_\w+)<FFT_(\d+)_forward(_noreorder)?>", k)
    if not m:
        continue
    path, n, nr = m.group(1), int(m.group(2)), bool(m.group(3))
    nfft = total // n
    if "multiple" in path:
        nfft = (nfft // 400 * 400) if n == 32 else (nfft // 200 * 200) if n == 64 else (nfft // 100 * 100)
    rows.setdefault(f"{path} N={n} {'noreorder' if nr else 'reorder'}", {})[c] = sum(v) / len(v) / nfft
for name, r in rows.items():
    if "SQ_LDS_IDX_ACTIVE" in r and r["SQ_LDS_IDX_ACTIVE"]:
        r["bank_conflict_ratio"] = r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"]
for name, r in rows.items():
    # SQ_ACTIVE_INST_VALU counts quad-cycles per wave; a SIMD serves two waves' VALU instructions in those four cycles
    if "SQ_ACTIVE_INST_VALU" in r and r.get("SQ_WAVE_CYCLES"):
        r["valu_active_fraction_of_wave_time"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES (pass 1), SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES (pass 2) -- python3 tools/sweep.py --sizes 32..4096 --paths multiple[,external] --variants f0,f1",
           "unit": "counter value per FFT (README batch: 2^29/N FFTs; multiple path: floor(nFFTs/100)*100 FFT executions)",
           "per_fft": dict(sorted(rows.items()))}, open(f"profiles/{tag}_pmc_lds.json", "w"), indent=1)
print("wrote profiles/ for", tag, ":", sorted(os.listdir("profiles")))
