#!/usr/bin/env python3
"""Design check of the planar ("structure of arrays") in-LDS engine (include/smfft/smfft_planar.hpp), CPU only.

The in-LDS kernels keep every LDS image as two planes of dwords (re, im).  Every store is a lane-linear
ds_write_addtid_b32 (register j of thread tid -> row j, dword tid: 2 LDS cycles per dword against 6 per float2 of
ds_write_b64), so all freedom is in WHO READS WHAT: each exchange is arranged so that a reader's sixteen values are
runs of contiguous dwords of few rows (ds_read_b128 / ds_read_b64: full LDS rate) and the row bases are padded so
that those reads are bank-conflict free under the gfx950 rules (MI355X_MICROARCH.md, LDS table).

This script (1) emulates the whole choreography thread by thread for every N and both orderings and checks the
result against numpy.fft, and (2) counts bank conflicts of every read instruction.  It also prints the row-base
tables the header uses.
    python tools/soa_model.py [N ...]"""
import sys

import numpy as np

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
B64_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def conflict_cycles(addrs, width):
    """addrs: dword address per lane (64 lanes, None = inactive) of one read instruction of `width` dwords per lane.
    Returns extra LDS cycles (0 = conflict free)."""
    groups = B128_GROUPS if width == 4 else B64_GROUPS
    nbanks = 64 if width >= 2 else 32
    extra = 0
    for g in groups:
        per_bank = {}
        for lane in g:
            a = addrs[lane]
            if a is None:
                continue
            for k in range(width):
                per_bank.setdefault((a + k) % nbanks, set()).add(a + k)
        worst = max((len(v) for v in per_bank.values()), default=1)
        extra += worst - 1
    return extra


def rev(v, bits):
    r = 0
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def ilog2(n):
    return n.bit_length() - 1


class Geometry:
    def __init__(self, N):
        self.N = N
        self.T = N // 16
        self.R1 = N // 16 if N <= 256 else 16
        self.RM = 1 if N <= 256 else N // 256
        self.BM = 16 // self.RM
        self.B1 = 16 // self.R1
        self.T1 = N // self.R1
        self.TW = max(64, self.T)                 # threads per compact workgroup
        self.F = self.TW // self.T                # FFTs per workgroup
        self.tb = ilog2(self.T)


# ---- row bases (dwords inside a plane).  A row holds TW dwords (one per thread). ------------------------------------
def row_bases(G, kind):
    """kind: 'image' (16 rows, natural registers), 'x1' (exchange after pass 1), 'x2' (exchange before the last pass)."""
    TW = G.TW
    if G.RM > 1:
        if kind == "x1":
            return [TW * j for j in range(16)]
        if kind == "x2":
            return [TW * j + 4 * (j // G.RM) for j in range(16)] if G.RM >= 4 else [TW * j + 4 * ((j // 2) % 4) + 16 * (j // 8) for j in range(16)]
        if kind == "image":
            return IMAGE_BASES[G.N]
    else:
        if kind == "x2":
            return LAST_BASES[G.N]
        if kind == "image":
            return IMAGE_BASES[G.N]
    raise ValueError(kind)


IMAGE_BASES = {}
LAST_BASES = {}


def pass1_role(G, v):
    """thread position v inside its FFT -> pass-1 role t1 (three-pass sizes: v = RM * t2 + r2 -> t1 = t2 + 16 * r2)"""
    if G.RM > 1:
        return (v // G.RM) + 16 * (v % G.RM)
    return v


def search_bases(G, kind, addr_fn, widths, tries):
    """find row bases (multiples of 4 dwords, rows of TW dwords, not overlapping) that make every read conflict free"""
    best = None
    for name, bases in tries:
        total = 0
        for instr in addr_fn(bases):
            total += conflict_cycles(instr[0], instr[1])
        if best is None or total < best[0]:
            best = (total, name, bases)
        if total == 0:
            break
    return best


def bases_from_residues(TW, q):
    """row j gets quad residue q[j] (its base / 4 mod 16): rows are placed in the order of their residues, so the shifts
    never make rows overlap and the plane is at most 16 * TW + 60 dwords"""
    order = sorted(range(16), key=lambda j: (q[j], j))
    bases = [0] * 16
    for rank, j in enumerate(order):
        bases[j] = TW * rank + 4 * q[j]
    return bases


def candidate_bases(TW):
    """GF(2)-linear maps of the row index to the quad residue, simplest first"""
    out = [("q = 0", [TW * j for j in range(16)])]
    seen = set()
    import itertools
    # each residue bit is the XOR of a subset of the row-index bits: 16 choices per bit, tried in order of total weight
    masks = sorted(range(16), key=lambda m: (bin(m).count("1"), m))
    combos = sorted(itertools.product(masks, repeat=4), key=lambda ms: (sum(bin(m).count("1") for m in ms), ms))
    for ms in combos:
        q = tuple(sum(((bin(j & ms[b]).count("1") & 1) << b) for b in range(4)) for j in range(16))
        if q in seen:
            continue
        seen.add(q)
        out.append((f"q bits = parity(j & {ms})", bases_from_residues(TW, q)))
    return out


# ---- the reads of each exchange, as (addresses per lane of the wave, width) -------------------------------------------
def image_bitrev_reads(G, bases, wave=0):
    """thread with pass-1 role t1 reads the sixteen contiguous elements p = 16 * rho + i, rho = rev_T(t1), of its FFT:
    element p lies in row p // T at dword fft * T + p % T."""
    T, TW = G.T, G.TW
    run = min(16, T)                     # contiguous dwords per row
    width = 4 if run >= 4 else 2
    instrs = []
    for start in range(0, 16, width):
        addrs = []
        for lane in range(64):
            tid = 64 * wave + lane
            fft, v = tid // T, tid % T
            rho = rev(pass1_role(G, v), G.tb)
            p = 16 * rho + start
            addrs.append(bases[p // T] + fft * T + p % T)
        instrs.append((addrs, width))
    return instrs


def x1_reads(G, bases, wave=0):
    """middle thread (t2, a) = position v' = 16 a + t2 reads rows q1 = a * BM + c, dwords fft * T + RM * t2 + [0, RM)"""
    T, RM, BM = G.T, G.RM, G.BM
    width = min(RM, 4)
    instrs = []
    for c in range(BM):
        for start in range(0, RM, width):
            addrs = []
            for lane in range(64):
                tid = 64 * wave + lane
                fft, v = tid // T, tid % T
                t2, a = v % 16, v // 16
                addrs.append(bases[a * BM + c] + fft * T + RM * t2 + start)
            instrs.append((addrs, width))
    return instrs


def x2_reads(G, bases, klow_of, wave=0):
    """last-pass thread at position v'' with output index klow = klow_of(v'') reads t2 = 0..15:
    row j = c * RM + q2, dwords fft * T + 16 * a + [0, 16), with c = klow % BM, a = (klow % 16) // BM, q2 = klow // 16"""
    T, RM, BM = G.T, G.RM, G.BM
    instrs = []
    for start in range(0, 16, 4):
        addrs = []
        for lane in range(64):
            tid = 64 * wave + lane
            fft, v = tid // T, tid % T
            klow = klow_of(v)
            c, a, q2 = klow % BM, (klow % 16) // BM, klow // 16
            addrs.append(bases[c * RM + q2] + fft * T + 16 * a + start)
        instrs.append((addrs, 4))
    return instrs


def last_reads_two_pass(G, bases, wave=0):
    """two-pass sizes: last-pass thread q1 (= position v'') reads n1 = 0..15: n1 = t1 + T * b in row b * R1 + q1, dword fft * T + t1"""
    T, R1, B1 = G.T, G.R1, G.B1
    instrs = []
    for b in range(B1):
        for start in range(0, T, 4):
            addrs = []
            for lane in range(64):
                tid = 64 * wave + lane
                fft, v = tid // T, tid % T
                addrs.append(bases[b * R1 + v] + fft * T + start)
            instrs.append((addrs, 4))
    return instrs


def pick_bases(N):
    G = Geometry(N)
    waves = G.TW // 64
    cands = candidate_bases(G.TW)

    def over_waves(fn):
        return lambda bases: [i for w in range(waves) for i in fn(G, bases, w)]

    res = {}
    best = search_bases(G, "image", over_waves(image_bitrev_reads), None, cands)
    IMAGE_BASES[N] = best[2]
    res["image"] = best
    if G.RM > 1:
        res["x1"] = search_bases(G, "x1", over_waves(x1_reads), None, cands)
        for name, klow_of in (("x2 (klow = position)", lambda v: v), ("x2 (klow = pass-1 role)", lambda v: pass1_role(G, v))):
            res[name] = search_bases(G, "x2", lambda bases: [i for w in range(waves) for i in x2_reads(G, bases, klow_of, w)], None, cands)
    else:
        best = search_bases(G, "x2", over_waves(last_reads_two_pass), None, cands)
        LAST_BASES[N] = best[2]
        res["last"] = best
    return G, res


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]
    for N in sizes:
        G, res = pick_bases(N)
        print(f"N={N}: T={G.T} TW={G.TW} RM={G.RM}")
        for k, (extra, name, bases) in res.items():
            span = max(bases) + G.TW
            print(f"  {k:28s}: extra LDS cycles {extra:3d}  bases {name}  plane span {span} dwords")


if __name__ == "__main__":
    main()
