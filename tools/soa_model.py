#!/usr/bin/env python3
"""Design check of the planar ("structure of arrays") in-LDS engine (include/smfft/smfft_planar.hpp), CPU only.

The in-LDS kernels keep every LDS image as two planes of dwords (re, im).  Every store is a lane-linear
ds_write_addtid_b32 (register j of thread tid -> row j, dword tid: 2 LDS cycles per dword against 6 per float2 of
ds_write_b64), so all freedom is in WHO READS WHAT: each exchange is arranged so that a reader's sixteen values are
runs of contiguous dwords of few rows (ds_read_b128 / ds_read_b64: full LDS rate) and the row bases are padded so
that those reads are bank-conflict free under the gfx950 rules (MI355X_MICROARCH.md, LDS table).

This script states the read pattern of every exchange (who reads which dwords of which rows), counts the bank
conflicts of every read instruction and searches, per length and exchange, the shift of each row (its base / 4 mod 16:
the "residue") that makes all of them conflict free.  It prints the residue tables pasted into the header.  (That the
index maps compute the right transform is checked on the GPU by the parity tests of the `multiple` path.)
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
    groups = B128_GROUPS if width == 4 else B64_GROUPS     # ds_read_b64 / b32: two groups of 32 lanes
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


def pass1_role(G, v):
    """thread position v inside its FFT -> pass-1 role t1 (three-pass sizes: v = RM * t2 + r2 -> t1 = t2 + 16 * r2)"""
    if G.RM > 1:
        return (v // G.RM) + 16 * (v % G.RM)
    return v


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
    width = min(T, 4)
    instrs = []
    for b in range(B1):
        for start in range(0, T, width):
            addrs = []
            for lane in range(64):
                tid = 64 * wave + lane
                fft, v = tid // T, tid % T
                addrs.append(bases[b * R1 + v] + fft * T + start)
            instrs.append((addrs, width))
    return instrs


def bases_from_residues(TW, q):
    """row j gets quad residue q[j] (its base / 4 mod 16): rows are placed in the order of their residues, so the shifts
    never make rows overlap and the plane is at most 16 * TW + 60 dwords"""
    order = sorted(range(16), key=lambda j: (q[j], j))
    bases = [0] * 16
    for rank, j in enumerate(order):
        bases[j] = TW * rank + 4 * q[j]
    return bases


def total_conflicts(TW, q, instr_fn):
    bases = bases_from_residues(TW, q)
    return sum(conflict_cycles(a, w) for a, w in instr_fn(bases))


def search_residues(TW, instr_fn, seed=0):
    """coordinate descent over the sixteen residues, from a few structured starts and random restarts"""
    rng = np.random.default_rng(seed)
    starts = [[0] * 16, [j % 16 for j in range(16)], [(j >> 2) for j in range(16)], [(j >> 1) & 3 for j in range(16)],
              [(j & 3) + 8 * (j >> 3) for j in range(16)], [(j & 3) + 4 * ((j >> 2) & 1) for j in range(16)]]
    best = None
    for attempt in range(40):
        q = list(starts[attempt]) if attempt < len(starts) else [int(x) for x in rng.integers(0, 16, 16)]
        cur = total_conflicts(TW, q, instr_fn)
        improved = True
        while improved and cur > 0:
            improved = False
            for j in range(16):
                for val in range(16):
                    if val == q[j]:
                        continue
                    old = q[j]
                    q[j] = val
                    t = total_conflicts(TW, q, instr_fn)
                    if t < cur:
                        cur, improved = t, True
                    else:
                        q[j] = old
        if best is None or cur < best[0] or (cur == best[0] and max(q) < max(best[1])):
            best = (cur, list(q))
        if cur == 0 and attempt >= len(starts) - 1:
            break
        if cur == 0 and max(q) <= 3:
            break
    return best


def exchanges(N):
    """name -> function(bases) -> list of (addresses, width) over every wave of the workgroup"""
    G = Geometry(N)
    waves = G.TW // 64

    def over_waves(fn, *extra):
        return lambda bases: [i for w in range(waves) for i in fn(G, bases, *extra, wave=w)]

    ex = {"image (bit-reversed rows)": over_waves(image_bitrev_reads)}
    if G.RM > 1:
        ex["x1"] = over_waves(x1_reads)
        ex["x2, klow = position (no reorder)"] = over_waves(x2_reads, lambda v: v)
        ex["x2, klow = pass-1 role (reorder)"] = over_waves(x2_reads, lambda v: pass1_role(G, v))
    else:
        ex["last (two-pass)"] = over_waves(last_reads_two_pass)
    return G, ex


def bit(j, b):
    return (j >> b) & 1


# The residues include/smfft/smfft_planar.hpp uses (row_residue there, in closed form), per length and exchange, with the
# pass-1 role map of that length: what tests/test_planar_layout_model.py checks to be conflict free in this model.
# N = 512 exchanges 1 in registers (roles = positions); N = 2048 no reorder takes r2 bit-reversed; N = 4096 reads exchange 1
# in a per-lane rotated order, computes klow = pass-1 role in both orderings and reads its bit-reversed image as dwords.
HEADER_RESIDUES = {
    64: {"image": [4 * (j >> 2) for j in range(16)], "last": [4 * (j & 3) for j in range(16)]},
    128: {"image": [bit(j, 2) + 8 * bit(j, 3) for j in range(16)], "last": [bit(j, 0) + 8 * bit(j, 1) for j in range(16)]},
    256: {"image": [j for j in range(16)], "last": [(j & 3) + 8 * bit(j, 3) for j in range(16)]},
    512: {"image": [bit(j, 0) + 2 * bit(j, 2) + 8 * bit(j, 3) for j in range(16)], "x2": [bit(j, 0) + 2 * bit(j, 1) + 8 * bit(j, 2) for j in range(16)]},
    1024: {"image": [j >> 2 for j in range(16)], "x1": [0] * 16, "x2": [j >> 2 for j in range(16)], "x2_reorder": [(j & 3) + 8 * bit(j, 3) for j in range(16)]},
    2048: {"image": [j >> 2 for j in range(16)], "x1": [bit(j, 1) for j in range(16)], "x2": [bit(j, 0) + 2 * bit(j, 3) for j in range(16)],
           "x2_reorder": [(j & 3) + 8 * bit(j, 3) for j in range(16)]},
    4096: {"image": [4 * bit(j, 3) for j in range(16)], "x1": [2 * bit(j, 0) for j in range(16)], "x2_reorder": [(j & 3) + 8 * bit(j, 3) for j in range(16)]},
}


def header_conflicts(N):
    """extra LDS cycles of every read set of length N with the header's residues and role maps: {name: cycles}"""
    G = Geometry(N)
    waves = G.TW // 64
    res = HEADER_RESIDUES[N]
    out = {}

    def role_plain(v):
        return (v // G.RM) + 16 * (v % G.RM) if G.RM > 1 else v

    def role_noreorder(v):
        if N == 512:
            return v
        if N == 2048:
            return (v // G.RM) + 16 * rev(v % G.RM, ilog2(G.RM))
        return role_plain(v)

    def total(q, instrs):
        bases = bases_from_residues(G.TW, q)
        return sum(conflict_cycles(a, w) for a, w in instrs(bases))

    global pass1_role
    saved = pass1_role
    try:
        pass1_role = lambda G_, v: role_noreorder(v)       # noqa: E731  (the no-reorder kernels' map)
        if N == 4096:
            def image(bases):      # sixteen ds_read_b32 per plane: element 16 * (rho % 16) + i at dword 16 * i + rho % 16 of row rho / 16
                ins = []
                for w in range(waves):
                    for i in range(16):
                        addrs = []
                        for lane in range(64):
                            rho = rev(role_plain(64 * w + lane), 8)
                            addrs.append(bases[rho // 16] + 16 * i + rho % 16)
                        ins.append((addrs, 1))
                return ins
            out["image (dword reads)"] = total(res["image"], image)

            def x1(bases):         # quad (k + rot) % 4 of the run at step k, rot = t2 >> 3
                ins = []
                for w in range(waves):
                    for k in range(4):
                        addrs = []
                        for lane in range(64):
                            v = 64 * w + lane
                            t2, a = v % 16, v // 16
                            addrs.append(bases[a] + 16 * t2 + 4 * ((k + (t2 >> 3)) & 3))
                        ins.append((addrs, 4))
                return ins
            out["x1 (rotated)"] = total(res["x1"], x1)
        else:
            out["image"] = total(res["image"], lambda b: [i for w in range(waves) for i in image_bitrev_reads(G, b, wave=w)])
            if "x1" in res:
                out["x1"] = total(res["x1"], lambda b: [i for w in range(waves) for i in x1_reads(G, b, wave=w)])
        if "last" in res:
            out["last"] = total(res["last"], lambda b: [i for w in range(waves) for i in last_reads_two_pass(G, b, wave=w)])
        if "x2" in res:
            out["x2 (klow = position)"] = total(res["x2"], lambda b: [i for w in range(waves) for i in x2_reads(G, b, lambda v: v, wave=w)])
        if "x2_reorder" in res:
            out["x2 (klow = role)"] = total(res["x2_reorder"], lambda b: [i for w in range(waves) for i in x2_reads(G, b, role_plain, wave=w)])
    finally:
        pass1_role = saved
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--header":
        for N in sorted(HEADER_RESIDUES):
            print(N, header_conflicts(N))
        return
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]
    for N in sizes:
        G, ex = exchanges(N)
        print(f"N={N}: T={G.T} TW={G.TW} RM={G.RM}")
        for name, fn in ex.items():
            extra, q = search_residues(G.TW, fn)
            span = max(bases_from_residues(G.TW, q)) + G.TW
            print(f"  {name:36s}: extra LDS cycles {extra:3d}  residues {{{', '.join(str(x) for x in q)}}}  plane span {span} dwords", flush=True)


if __name__ == "__main__":
    main()
