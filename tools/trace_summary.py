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
        wall.append([int(v) for v in row[5:]])
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
if wall and len(wall[0]) >= 14:
    w = np.array(wall, dtype=np.int64)
    t0 = w[:, 0].min()
    us = lambda a: " / ".join(f"{np.percentile(a, p) / 100:.1f}" for p in (5, 50, 95, 100))
    start, end = w[:, 0], w[:, 1]
    pieces = w[:, 2:14].reshape(len(w), 4, 3)
    used = pieces[:, :, 0] > 0
    first_tile = pieces[:, 0, 0] - start
    compute = np.where(used, pieces[:, :, 1] - pieces[:, :, 0], 0).sum(axis=1)
    store = np.where(used, pieces[:, :, 2] - pieces[:, :, 1], 0).sum(axis=1)
    # between pieces: from "tile stored" of piece k to "tile in LDS" of piece k + 1 (hand-over wait + tile load)
    gaps = np.zeros(len(w), dtype=np.int64)
    for k in range(3):
        both = used[:, k] & used[:, k + 1]
        gaps += np.where(both, pieces[:, k + 1, 0] - pieces[:, k, 2], 0)
    total = end - start
    print(f"  device-wide clock, us (5 / 50 / 95 / 100 %% over workgroups): start {us(start - t0)}; end {us(end - t0)}; workgroup duration {us(total)}")
    print(f"  inside a workgroup, us: first tile in LDS after its start {us(first_tile)}; applications {us(compute)}; tile stores (+ parking) {us(store)}; "
          f"between pieces (wait for a parked chain + tile load) {us(gaps)}; everything but applications {us(total - compute)}")
    print(f"  launch: first start to last end {(end.max() - t0) / 100:.1f} us; mean of the workgroups' application time {compute.mean() / 100:.1f} us "
          f"= {100 * compute.mean() / (end.max() - t0):.1f} % of it; the slowest workgroup's {compute.max() / 100:.1f} us")
