#!/usr/bin/env python3
"""Summarises a placement-study run (tools/placement_round.sh -> gpurun_out/<tag>/) into the tracked files
profiles/<round>_placement_map.txt (the chunk map, verbatim) and profiles/<round>_placement_pmc.json (class sizes, the
class-pair timing matrix, and the PMC counters of the tagged dispatches, per kernel).

    python tools/placement_report.py gpurun_out/r02_placement r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

import numpy as np

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(os.path.join(src, "map.txt")).read().splitlines()
shutil.copy(os.path.join(src, "map.txt"), os.path.join(root, "profiles", f"{rnd}_placement_map.txt"))
i = [k for k, l in enumerate(lines) if l.startswith("matrix:")][0]
M = []
for l in lines[i + 1:]:
    f = l.split()
    if len(f) < 10:
        break
    M.append([int(x) for x in f[1:]])
M = np.array(M)
n = M.shape[0]
rw = {}
for l in lines:
    f = l.split()
    if len(f) == 6 and f[0].isdigit():
        rw[int(f[0])] = (float(f[1]), float(f[2]))
wr_med = float(np.median([rw[j][1] for j in range(n)]))
mixed = [j for j in range(n) if rw[j][1] < 0.93 * wr_med]
ordinary = [j for j in range(n) if j not in mixed]
cut = 153   # ms x 100: same-class pairs sit at 1.55-1.60, pairs of different classes at 1.48-1.52
cls, cid = {}, 0
for j in ordinary:
    if j in cls:
        continue
    cls[j] = cid
    for k in ordinary:
        if k not in cls and (M[j, k] >= cut or M[k, j] >= cut):
            cls[k] = cid
    cid += 1
groups = collections.defaultdict(list)
for j, c in cls.items():
    groups[c].append(j)
doc = {"source": src, "chunk_GiB": 4, "chunks": n, "window": "4 GiB read + 4 GiB written per copy, 16 x 8 B/lane tile shape of the external kernels",
       "classes": {f"class{c}": {"chunks": g, "GiB": 4 * len(g)} for c, g in groups.items()},
       "mixed": {"chunks": mixed, "GiB": 4 * len(mixed), "note": "pure write 15-20 %% faster, pure read 5-8 %% slower than an ordinary chunk; chunk %d is the hipMallocAsync allocation made first" % (n - 1)},
       "pure_read_ms": {"ordinary": float(np.mean([rw[j][0] for j in ordinary])), "mixed": float(np.mean([rw[j][0] for j in mixed])) if mixed else None},
       "pure_write_ms": {"ordinary": float(np.mean([rw[j][1] for j in ordinary])), "mixed": float(np.mean([rw[j][1] for j in mixed])) if mixed else None}}
pairs = {}
for a in groups:
    for b in groups:
        v = [M[x, y] for x in groups[a] for y in groups[b] if x != y]
        pairs[f"class{a}->class{b}"] = {"mean_ms": float(np.mean(v)) / 100, "min_ms": float(np.min(v)) / 100, "max_ms": float(np.max(v)) / 100}
    if mixed:
        v = [M[x, y] for x in groups[a] for y in mixed]
        w = [M[y, x] for x in groups[a] for y in mixed]
        pairs[f"class{a}->mixed"] = {"mean_ms": float(np.mean(v)) / 100, "min_ms": float(np.min(v)) / 100, "max_ms": float(np.max(v)) / 100}
        pairs[f"mixed->class{a}"] = {"mean_ms": float(np.mean(w)) / 100}
doc["copy_ms_by_class_pair"] = pairs

pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for row in csv.DictReader(open(files[0])):
        m = re.search(r"study_(\w+)<(\d+)", row["Kernel_Name"])
        if not m or m.group(2) == "0":
            continue
        name = f"{m.group(1)}<{m.group(2)}>"
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[name][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    tags = [l for l in open(d + ".txt").read().splitlines() if l.startswith("tags:")]
    pmc[os.path.basename(d)] = {"tags": tags[0] if tags else None,
                                "kernels": {k: dict({"dispatches": len(dur[k]), "mean_ms": sum(dur[k].values()) / len(dur[k])},
                                                    **{c: sum(v) / len(v) for c, v in agg[k].items()}) for k in sorted(agg)}}
doc["pmc"] = pmc
doc["legend"] = {"copy<1>": "chunk 0 -> S (slowest output = chunk 0's class)", "copy<2>": "0 -> X (median output = another class)", "copy<3>": "0 -> F (fastest output = a mixed chunk)",
                 "copy<4>": "F -> 0", "copy<5>": "S -> 0", "copy<6>": "X -> 0", "read<1,2,3>": "pure read of 0, X, F", "write<1,2,3>": "pure write of S, X, F",
                 "copy_paced<1,2,3>": "the copies 1-3 with 16 serialised flat LDS loads between a wave's loads and its stores (vmem_throttle)",
                 "units": "counter values are sums over all TCC / TCP instances per dispatch; S, X, F are re-chosen in every pass (fresh allocation)"}
json.dump(doc, open(os.path.join(root, "profiles", f"{rnd}_placement_pmc.json"), "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("classes", "mixed", "pure_read_ms", "pure_write_ms")}, indent=1)[:1500])
