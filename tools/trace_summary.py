"""Summary of a SMFFT_SCHEDULE_TRACE file (tools/workgroup_trace.py): per CU, when its workgroups start and end relative to the
CU's first start (cycle counters of different CUs are not aligned):  python tools/trace_summary.py trace.txt"""
import sys
from collections import defaultdict

import numpy as np

rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
print(open(sys.argv[1]).readline().strip())
cus = defaultdict(list)
wall = []
for row in rows:
    blk, start, end, hw, xcc = row[:5]
    if len(row) >= 8:
        wall.append((int(row[5]), int(row[6]), int(row[7])))
    hw = int(hw, 16)
    key = (int(xcc, 16) & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    cus[key].append((int(start), int(end), int(blk)))
ramp, spread, dur, steps = [], [], [], []
for key, wgs in cus.items():
    wgs.sort()
    t0 = wgs[0][0]
    starts = np.array([w[0] - t0 for w in wgs])
    ends = np.array([w[1] - t0 for w in wgs])
    ramp.append(starts[-1])
    steps.extend(np.diff(starts))
    spread.append(ends.max() - ends.min())
    dur.append(ends.max())
q = lambda a: " / ".join(f"{np.percentile(a, p):.0f}" for p in (5, 50, 95))
print(f"{len(cus)} CUs, {len(rows)} workgroups, {len(rows) / len(cus):.1f} per CU")
print(f"cycles (5 / 50 / 95 %% over CUs): last start after the CU's first {q(ramp)}; step between consecutive starts on a CU {q(steps)}")
print(f"  last end after the CU's first start {q(dur)}; last end - first end on a CU {q(spread)}")
one = sorted(cus.items())[0]
print("  one CU", one[0], ":", " ".join(f"[{b}: {s - one[1][0][0]}..{e - one[1][0][0]}]" for s, e, b in one[1]))
if wall:
    w = np.array(wall, dtype=np.int64)
    t0 = w[:, 0].min()
    us = lambda a: " / ".join(f"{np.percentile(a, p) / 100:.1f}" for p in (5, 50, 95, 100))
    print(f"  device-wide clock, us (5 / 50 / 95 / 100 %% over workgroups): start {us(w[:, 0] - t0)}; first tile in LDS after the start {us(w[:, 2] - w[:, 0])}; "
          f"end {us(w[:, 1] - t0)}; workgroup duration {us(w[:, 1] - w[:, 0])}")
