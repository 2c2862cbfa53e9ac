"""Round 6 form of the reference-contract engine (include/smfft/smfft_device_functions.hpp, quarter_fft) for N >= 256: the radix-2^2
decimation-in-time ladder on four elements per thread, cut into PHASES in each of which a wave owns eight position bits -- six in its
lanes, two in the four slots of a thread -- and runs every pass over those bits on lanes and registers (one-bit lane <-> slot transposes
between passes); phases are joined by one trip through the LDS image and a workgroup barrier where the owners change.

  natural order   phase 0: thread t loads x[t + m N/4] and runs pass 0 (bits 0, 1; carries the bit reversal: results = positions 4 rev(t) + i)
                  image 1 (swizzled), read back with slots = position bits (2, 3)  -- the pass can start at once: no transpose in front of it
                  phase 1: passes (2,3) (4,5) (6,7) on the wave's aligned block of 256 positions
                  N = 256: results are final (slots = bits 6, 7; lane = position bits 0..5): natural store, conflict free
                  N = 512 / 1024: image 2 (swizzled); the last pass (radix 2 on bit 8 / radix 4 on bits 8, 9) in the thread, natural store
                                  (the form of -DSMFFT_QUARTER_PAIRS=0 and of the register-out variants; the header's default: transform_pairs below)
                  N = 2048 / 4096: image 2 (natural layout); LAST PHASE below
  no reorder      phase 1: thread t loads x[4 t + i], passes (0,1) ... (6,7) on the wave's aligned block; the exchange between the passes
                  (2,3) and (4,5) goes through the wave's own block of the image (no barrier) instead of through sixteen DPP-fed selects
                  N = 512 / 1024: as before; N = 2048 / 4096: image (natural layout); LAST PHASE
  LAST PHASE (N >= 2048): the wave owns position bits 8 ... n-1 and the low 16 - n bits; slots = bits (8, 9); pass (8,9), then
                  N = 4096: slots <-> lane bits 4, 5 (v_permlane16/32_swap), pass (10,11); N = 2048: slot bit 0 <-> lane bit 5, radix 2 on bit 10;
                  natural store.  Both cross-wave passes of these lengths between ONE pair of barriers (round 5: two trips through LDS).

This file replays threads, lanes, slots, transposes and images in NumPy against numpy.fft (check()), and counts the LDS cycles of
every access with the gfx950 lane-group rules (lds_report()).  CPU only; tests/test_quarter_swizzle_model.py runs both."""
import numpy as np


def rev(v, bits):
    r = 0
    for b in range(bits):
        r |= ((v >> b) & 1) << (bits - 1 - b)
    return r


def product_swizzle(i):
    return i ^ ((i >> 8) & 31) ^ ((i >> 4) & 30) ^ ((i >> 2) & 24)


# N = 256 natural order: image 1 is a GF(2)-linear bijection of the eight position bits (rows = address bits, given as the XOR of
# position bits) found by search_256(): scattered stores and slot-(2,3) reads both conflict free
M256 = None      # filled in below


def apply_rows(rows, p):
    a = 0
    for bit, mask in enumerate(rows):
        a |= (bin(p & mask).count("1") & 1) << bit
    return a


class Wave:
    """64 lanes x 4 slots of complex values with the position each one holds (tracking only)"""

    def __init__(self):
        self.e = np.zeros((64, 4), complex)
        self.pos = np.zeros((64, 4), int)

    def swap(self, slot_bit, lane_bit):
        ne, npos = self.e.copy(), self.pos.copy()
        for lane in range(64):
            partner = lane ^ (1 << lane_bit)
            for i in range(4):
                if (i >> slot_bit) & 1:
                    continue
                hi = i | (1 << slot_bit)
                if (lane >> lane_bit) & 1 == 0:
                    ne[lane, hi], npos[lane, hi] = self.e[partner, i], self.pos[partner, i]
                else:
                    ne[lane, i], npos[lane, i] = self.e[partner, hi], self.pos[partner, hi]
        self.e, self.pos = ne, npos


def quad(e, pos, P, sign):
    """fused radix-2^2 butterfly on slots (0,1,2,3) = positions k, k + P, k + 2P, k + 3P (in that slot order)"""
    k = int(pos[0]) & (P - 1)
    assert [int(q) for q in pos] == [int(pos[0]) + m * P for m in range(4)], (pos, P)
    w2 = np.exp(sign * 2j * np.pi * k / (4 * P))
    w1 = w2 * w2
    x0, x1, x2, x3 = e
    t1, t3 = x1 * w1, x3 * w1
    y0, y1, y2, y3 = x0 + t1, x0 - t1, x2 + t3, x2 - t3
    u2, v3 = y2 * w2, y3 * w2
    u3 = v3 * (1j * sign)
    return [y0 + u2, y1 + u3, y0 - u2, y1 - u3]


def transform(N, DIR, REORDER, x, log=None):
    """the whole device function; log: list that receives (kind, [float2 index per lane]) of every wave-level LDS access"""
    n = N.bit_length() - 1
    Q, TB = N // 4, n - 2
    sign = 1 if DIR else -1
    waves = N // 256
    img = np.zeros(N, complex)

    def note(kind, addr):
        if log is not None:
            log.append((kind, list(addr)))

    sw1 = (lambda p: apply_rows(M256, p)) if N == 256 else product_swizzle      # image 1 / the round-4 images
    sw2 = product_swizzle if n <= 10 else (lambda p: p)                          # image in front of the last pass(es)

    ladders = [Wave() for _ in range(waves)]
    if REORDER:
        # phase 0: pass 0 on x[t + m Q], scattered into image 1
        stores = [[] for _ in range(4)]
        for m in range(4):
            for w in range(waves):
                note("r", [64 * w + lane + m * Q for lane in range(64)])
        for t in range(Q):
            e = [0] * 4
            for m in range(4):
                e[((m & 1) << 1) | (m >> 1)] = x[t + m * Q]
            s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
            jd1 = d1 * (1j * sign)
            a = 4 * rev(t, TB)
            for i, v in enumerate([s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]):
                img[sw1(a + i)] = v
                stores[i].append(sw1(a + i))
        for i in range(4):
            for w in range(waves):
                note("w", stores[i][64 * w:64 * w + 64])
        # phase 1: slots = position bits (2, 3); lane bits 0, 1 = position bits 0, 1; lane bits 2 ... 5 = position bits 4 ... 7
        for w, L in enumerate(ladders):
            for j in range(4):
                addr = []
                for lane in range(64):
                    p = (lane & 3) + 4 * j + 16 * (lane >> 2) + 256 * w
                    L.e[lane, j], L.pos[lane, j] = img[sw1(p)], p
                    addr.append(sw1(p))
                note("r", addr)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 4, sign)
            for (b0, b1), P in (((2, 3), 16), ((4, 5), 64)):
                L.swap(0, b0)
                L.swap(1, b1)
                for lane in range(64):
                    L.e[lane] = quad(L.e[lane], L.pos[lane], P, sign)
    else:
        for w, L in enumerate(ladders):
            for lane in range(64):
                t = 64 * w + lane
                for i in range(4):
                    L.e[lane, i], L.pos[lane, i] = x[4 * t + i], 4 * t + i
                e = L.e[lane]
                s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
                jd1 = d1 * (1j * sign)
                L.e[lane] = [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]
            L.swap(0, 0)
            L.swap(1, 1)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 4, sign)
            # the exchange between the passes 1 and 2 through the wave's own block of the image (QuarterLanes::run_exchanged):
            # stored with slots = bits (2, 3), re-read with slots = bits (4, 5), lane bits 0 ... 3 = bits 0 ... 3, lane bits 4, 5 = bits 6, 7
            for j in range(4):
                addr = []
                for lane in range(64):
                    p = (lane & 3) + 4 * j + 16 * (lane >> 2) + 256 * w
                    assert int(L.pos[lane, j]) == p
                    img[256 * w + exchange_image(p & 255)] = L.e[lane, j]
                    addr.append(256 * w + exchange_image(p & 255))
                note("w", addr)
            for j in range(4):
                addr = []
                for lane in range(64):
                    p = (lane & 15) + 16 * j + 64 * (lane >> 4) + 256 * w
                    L.e[lane, j], L.pos[lane, j] = img[256 * w + exchange_image(p & 255)], p
                    addr.append(256 * w + exchange_image(p & 255))
                note("r", addr)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 16, sign)
            L.swap(0, 4)
            L.swap(1, 5)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 64, sign)
    # every thread of every wave now holds positions 256 w + lane + 64 i
    for w, L in enumerate(ladders):
        for lane in range(64):
            assert [int(q) for q in L.pos[lane]] == [256 * w + lane + 64 * i for i in range(4)]
    out = np.zeros(N, complex)
    if n == 8:
        for i in range(4):
            note("w", [lane + 64 * i for lane in range(64)])
        for lane in range(64):
            for i in range(4):
                out[lane + 64 * i] = ladders[0].e[lane, i]
        return out
    for i in range(4):
        for w, L in enumerate(ladders):
            addr = [sw2(256 * w + lane + 64 * i) for lane in range(64)]
            note("w", addr)
            for lane in range(64):
                img[addr[lane]] = L.e[lane, i]
    if n <= 10:
        # the last pass in the thread: radix 4 on (8, 9) / radix 2 on bit 8, elements t + m Q
        for m in range(4):
            for w in range(waves):
                note("r", [sw2(64 * w + lane + m * Q) for lane in range(64)])
        for t in range(Q):
            if n == 10:
                res = quad([img[sw2(t + m * 256)] for m in range(4)], [t + m * 256 for m in range(4)], 256, sign)
                for m in range(4):
                    out[t + m * 256] = res[m]
            else:
                wv = np.exp(sign * 2j * np.pi * t / N)
                x0, x1, x2, x3 = img[sw2(t)], img[sw2(t + N // 2)], img[sw2(t + Q)], img[sw2(t + 3 * Q)]
                t1, t3 = x1 * wv, x3 * wv * (1j * sign)
                out[t], out[t + N // 2], out[t + Q], out[t + 3 * Q] = x0 + t1, x0 - t1, x2 + t3, x2 - t3
        for m in range(4):
            for w in range(waves):
                note("w", [64 * w + lane + m * Q for lane in range(64)])
        return out
    # LAST PHASE (N = 2048 / 4096): lanes 0 ... low-1 = position bits 0 ... low-1 (low = 16 - n), the other lane bits = position bits 10 (, 11);
    # wave = the position bits between; slots = bits (8, 9)
    low = 16 - n
    for w in range(waves):
        L = Wave()
        for j in range(4):
            addr = []
            for lane in range(64):
                p = (lane & ((1 << low) - 1)) | (w << low) | (j << 8) | ((lane >> low) << 10)
                L.e[lane, j], L.pos[lane, j] = img[sw2(p)], p
                addr.append(sw2(p))
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 256, sign)
        if n == 12:
            L.swap(0, 4)
            L.swap(1, 5)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 1024, sign)
        else:
            L.swap(0, 5)            # slots = (bit 10, bit 9)
            for lane in range(64):
                p0 = int(L.pos[lane, 0])
                assert [int(q) for q in L.pos[lane]] == [p0, p0 + 1024, p0 + 512, p0 + 1536]
                wv = np.exp(sign * 2j * np.pi * p0 / N)
                x0, x1, x2, x3 = L.e[lane]
                t1, t3 = x1 * wv, x3 * wv * (1j * sign)
                L.e[lane] = [x0 + t1, x0 - t1, x2 + t3, x2 - t3]
        for i in range(4):
            note("w", [int(L.pos[lane, i]) for lane in range(64)])
            for lane in range(64):
                out[int(L.pos[lane, i])] = L.e[lane, i]
    return out


def transform_pairs(N, DIR, x, log=None):
    """Natural order in PAIRS of passes (round 6, last day; the header's form for N = 512 / 1024, SMFFT_QUARTER_PAIRS; the N = 256 form below
    measured 2 % slower than the three-pass phase on the GPU and is not in the header): a phase is two passes with ONE exchange between them, and
    that exchange is always the one of the lane bits 4, 5 (v_permlane16/32_swap) -- never the sixteen DPP-fed selects of a lane bit 0 ... 3.
      N = 256          [pass (0,1) on x[t + 64 m] | slots <-> lane bits 5, 4 | pass (2,3)]  image256  [pass (4,5) | <-> 4, 5 | pass (6,7)]
      N = 512 / 1024   pass (0,1) in the thread, scattered (as before); [pass (2,3) | <-> 4, 5 | pass (4,5)] on the wave's aligned block;
                       swizzled image; N = 1024: [pass (6,7) | <-> 4, 5 | pass (8,9)], wave = position bits 4, 5 -- a barrier in front of the natural store;
                       N = 512: [pass (6,7) | slot bit 0 <-> lane bit 5 | radix 2 on bit 8], wave = position bit 5: whole groups of 32, no barrier."""
    n = N.bit_length() - 1
    assert n in (8, 9, 10, 11, 12)
    Q, TB = N // 4, n - 2
    sign = 1 if DIR else -1
    waves = N // 256
    img = np.zeros(N, complex)
    if n >= 11:
        return transform_pairs_head(N, DIR, x, log)

    def note(kind, addr):
        if log is not None:
            log.append((kind, list(addr)))

    def pass0(e):
        s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
        jd1 = d1 * (1j * sign)
        return [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]

    out = np.zeros(N, complex)
    for m in range(4):
        for w in range(waves):
            note("r", [64 * w + lane + m * Q for lane in range(64)])
    if n == 8:
        L = Wave()
        for lane in range(64):
            e = [0] * 4
            for m in range(4):
                e[((m & 1) << 1) | (m >> 1)] = x[lane + m * Q]
            L.e[lane] = pass0(e)
            L.pos[lane] = [4 * rev(lane, 6) + i for i in range(4)]
        L.swap(0, 5)       # lane bit 5 holds position bit 2, lane bit 4 position bit 3
        L.swap(1, 4)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 4, sign)
        for j in range(4):
            addr = [image256(int(L.pos[lane, j])) for lane in range(64)]
            note("w", addr)
            for lane in range(64):
                img[addr[lane]] = L.e[lane, j]
        for j in range(4):
            addr = []
            for lane in range(64):
                p = (lane & 15) + 16 * j + 64 * (lane >> 4)
                L.e[lane, j], L.pos[lane, j] = img[image256(p)], p
                addr.append(image256(p))
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 16, sign)
        L.swap(0, 4)
        L.swap(1, 5)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 64, sign)
        for i in range(4):
            assert [int(L.pos[lane, i]) for lane in range(64)] == [lane + 64 * i for lane in range(64)]
            note("w", [lane + 64 * i for lane in range(64)])
            for lane in range(64):
                out[lane + 64 * i] = L.e[lane, i]
        return out
    sw = product_swizzle
    stores = [[] for _ in range(4)]
    for t in range(Q):
        e = [0] * 4
        for m in range(4):
            e[((m & 1) << 1) | (m >> 1)] = x[t + m * Q]
        a = 4 * rev(t, TB)
        for i, v in enumerate(pass0(e)):
            img[sw(a + i)] = v
            stores[i].append(sw(a + i))
    for i in range(4):
        for w in range(waves):
            note("w", stores[i][64 * w:64 * w + 64])
    # phase B: slots = position bits (2, 3); lane bits 0, 1 = bits 0, 1; lane bits 2, 3 = bits 6, 7; lane bits 4, 5 = bits 4, 5; wave = bits 8 (, 9)
    ladders = [Wave() for _ in range(waves)]
    for w, L in enumerate(ladders):
        for j in range(4):
            addr = []
            for lane in range(64):
                p = (lane & 3) | (j << 2) | ((lane >> 4) << 4) | (((lane >> 2) & 3) << 6) | (w << 8)
                L.e[lane, j], L.pos[lane, j] = img[sw(p)], p
                addr.append(sw(p))
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 4, sign)
        L.swap(0, 4)
        L.swap(1, 5)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 16, sign)
    for j in range(4):
        for w, L in enumerate(ladders):
            addr = [sw(int(L.pos[lane, j])) for lane in range(64)]
            note("w", addr)
            for lane in range(64):
                img[addr[lane]] = L.e[lane, j]
    # phase C
    for w in range(waves):
        L = Wave()
        for j in range(4):
            addr = []
            for lane in range(64):
                if n == 10:
                    p = (lane & 15) | (w << 4) | (j << 6) | ((lane >> 4) << 8)
                else:
                    p = (lane & 31) | (w << 5) | (j << 6) | ((lane >> 5) << 8)
                L.e[lane, j], L.pos[lane, j] = img[sw(p)], p
                addr.append(sw(p))
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 64, sign)
        if n == 10:
            L.swap(0, 4)
            L.swap(1, 5)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 256, sign)
        else:
            L.swap(0, 5)            # slots = (bit 8, bit 7)
            for lane in range(64):
                p0 = int(L.pos[lane, 0])
                assert [int(q) for q in L.pos[lane]] == [p0, p0 + 256, p0 + 128, p0 + 384]
                wv = np.exp(sign * 2j * np.pi * p0 / N)
                x0, x1, x2, x3 = L.e[lane]
                t1, t3 = x1 * wv, x3 * wv * (1j * sign)
                L.e[lane] = [x0 + t1, x0 - t1, x2 + t3, x2 - t3]
        for i in range(4):
            note("w", [int(L.pos[lane, i]) for lane in range(64)])
            for lane in range(64):
                out[int(L.pos[lane, i])] = L.e[lane, i]
        if n == 9:
            # in place: the words a wave stores to (natural layout) are the words it read (the swizzle permutes aligned groups of 32; the wave owns whole groups)
            read_words = {sw((lane & 31) | (w << 5) | (j << 6) | ((lane >> 5) << 8)) for lane in range(64) for j in range(4)}
            assert read_words == {int(L.pos[lane, i]) for lane in range(64) for i in range(4)}
    return out


def transform_pairs_head(N, DIR, x, log=None):
    """N = 2048 (the header's form; N = 4096 replayed as well -- measured no faster there, not in the header): [pass (0,1) | slots <-> lane bits 5, 4 |
    pass (2,3)] in front of the scattered store -- the thread loads x[(lane & 15) + 16 wave + 16 W (lane >> 4) + m N/4], W waves -- then
    [pass (4,5) | <-> 4, 5 | pass (6,7)] on the wave's aligned block, then the LAST PHASE of transform()."""
    n = N.bit_length() - 1
    Q = N // 4
    sign = 1 if DIR else -1
    waves, wb = N // 256, n - 8
    img = np.zeros(N, complex)
    sw = product_swizzle

    def note(kind, addr):
        if log is not None:
            log.append((kind, list(addr)))

    heads = []
    for w in range(waves):
        L = Wave()
        for m in range(4):
            note("r", [(lane & 15) + 16 * w + ((lane >> 4) << (4 + wb)) + m * Q for lane in range(64)])
        for lane in range(64):
            t_in = (lane & 15) + 16 * w + ((lane >> 4) << (4 + wb))          # the input index without the slots' bits
            e = [0] * 4
            for m in range(4):
                e[((m & 1) << 1) | (m >> 1)] = x[t_in + m * Q]
            s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
            jd1 = d1 * (1j * sign)
            L.e[lane] = [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]
            L.pos[lane] = [4 * rev(t_in, n - 2) + i for i in range(4)]
        L.swap(0, 5)
        L.swap(1, 4)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 4, sign)
            pw = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (rev(w, wb) << 4) | (rev(lane & 15, 4) << (4 + wb))
            assert [int(q) for q in L.pos[lane]] == [pw + 4 * j for j in range(4)]
        heads.append(L)
    for j in range(4):
        for L in heads:
            addr = [sw(int(L.pos[lane, j])) for lane in range(64)]
            note("w", addr)
            for lane in range(64):
                img[addr[lane]] = L.e[lane, j]
    blocks = []
    for w in range(waves):
        L = Wave()
        for j in range(4):
            addr = []
            for lane in range(64):
                p = (lane & 15) | (j << 4) | ((lane >> 4) << 6) | (w << 8)
                L.e[lane, j], L.pos[lane, j] = img[sw(p)], p
                addr.append(sw(p))
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 16, sign)
        L.swap(0, 4)
        L.swap(1, 5)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 64, sign)
            assert [int(q) for q in L.pos[lane]] == [256 * w + lane + 64 * i for i in range(4)]
        blocks.append(L)
    for i in range(4):                     # natural layout
        for w, L in enumerate(blocks):
            note("w", [256 * w + lane + 64 * i for lane in range(64)])
            for lane in range(64):
                img[256 * w + lane + 64 * i] = L.e[lane, i]
    out = np.zeros(N, complex)
    low = 16 - n
    for w in range(waves):
        L = Wave()
        for j in range(4):
            addr = []
            for lane in range(64):
                p = (lane & ((1 << low) - 1)) | (w << low) | (j << 8) | ((lane >> low) << 10)
                L.e[lane, j], L.pos[lane, j] = img[p], p
                addr.append(p)
            note("r", addr)
        for lane in range(64):
            L.e[lane] = quad(L.e[lane], L.pos[lane], 256, sign)
        if n == 12:
            L.swap(0, 4)
            L.swap(1, 5)
            for lane in range(64):
                L.e[lane] = quad(L.e[lane], L.pos[lane], 1024, sign)
        else:
            L.swap(0, 5)
            for lane in range(64):
                p0 = int(L.pos[lane, 0])
                wv = np.exp(sign * 2j * np.pi * p0 / N)
                x0, x1, x2, x3 = L.e[lane]
                t1, t3 = x1 * wv, x3 * wv * (1j * sign)
                L.e[lane] = [x0 + t1, x0 - t1, x2 + t3, x2 - t3]
        for i in range(4):
            note("w", [int(L.pos[lane, i]) for lane in range(64)])
            for lane in range(64):
                out[int(L.pos[lane, i])] = L.e[lane, i]
    return out


def lds_report_pairs(N):
    log = []
    transform_pairs(N, 0, np.zeros(N, complex), log)
    return sum(access_cycles(k, a) for k, a in log), sum(2 if k == "r" else 6 for k, a in log), len(log)


def access_cycles(kind, addr):
    """LDS-array cycles of one wave-level ds_read_b64 / ds_write_b64 (MI355X_MICROARCH.md, LDS): reads in two groups of 32 lanes on 32
    float2 banks, writes in four groups of 16 contiguous lanes on 16 float2 banks and never under 6 cycles (address / data transfer)"""
    group, banks, floor = (32, 32, 0) if kind == "r" else (16, 16, 6)
    cycles = 0
    for g0 in range(0, 64, group):
        load = {}
        for a in set(addr[g0:g0 + group]):
            load[a % banks] = load.get(a % banks, 0) + 1
        cycles += max(load.values())
    return max(floor, cycles)


def lds_report(N, REORDER):
    """(cycles, conflict-free cycles, accesses) of one transform of the block"""
    log = []
    transform(N, 0, REORDER, np.zeros(N, complex), log)
    total = sum(access_cycles(k, a) for k, a in log)
    ideal = sum(2 if k == "r" else 6 for k, a in log)
    return total, ideal, len(log)


def search_256():
    """image 1 of N = 256 natural order: address bit b = XOR of the position bits in rows[b].  Wanted: the scattered stores of pass 0
    (16 contiguous lanes = position bits 7, 6, 5, 4; slots = bits 0, 1) distinct mod 16, the slot-(2,3) reads (32 lanes = position
    bits 0, 1, 4, 5, 6) distinct mod 32.  Candidates: address bits 0 ... 3 each take ONE of the position bits 4 ... 7 in, bit 4 one of
    5 ... 7; bits 5 ... 7 stay.  Returns every conflict-free choice."""
    import itertools
    found = []
    for perm in itertools.permutations(range(4, 8)):
        for c4 in (5, 6, 7):
            r = [(1 << b) | (1 << perm[b]) for b in range(4)] + [(1 << 4) | (1 << c4), 1 << 5, 1 << 6, 1 << 7]
            if len({apply_rows(r, p) for p in range(256)}) != 256:
                continue
            ok = all(len({apply_rows(r, 4 * rev(t, 6) + i) % 16 for t in range(16 * g, 16 * g + 16)}) == 16 for i in range(4) for g in range(4))
            ok = ok and all(len({apply_rows(r, (l & 3) + 4 * j + 16 * (l >> 2)) % 32 for l in range(32 * h, 32 * h + 32)}) == 32 for j in range(4) for h in range(2))
            if ok:
                found.append(r)
    return found


# one of search_256()'s solutions: address bits 0 ... 3 take position bits 7, 6, 5, 4 in, bit 4 takes bit 6 in:
#   a = p ^ ((p >> 7) & 1) ^ ((p >> 5) & 2) ^ ((p >> 3) & 4) ^ ((p >> 1) & 8) ^ ((p >> 2) & 16)
M256 = [0x01 | 0x80, 0x02 | 0x40, 0x04 | 0x20, 0x08 | 0x10, 0x10 | 0x40, 0x20, 0x40, 0x80]


def exchange_image(p):
    """the wave's block in the exchange between the passes 1 and 2 of the no-reorder ladder (QuarterLanes::exchange_image)"""
    return p ^ ((p >> 2) & 28)


def image256(p):
    return p ^ ((p >> 7) & 1) ^ ((p >> 5) & 2) ^ ((p >> 3) & 4) ^ ((p >> 1) & 8) ^ ((p >> 2) & 16)


# ---- N = 32, 64, 128, natural order (round 6): the same road as N = 256 -- pass 0, ONE trip through the block's image, read back with
# slots = position bits (2, 3) -- for the transforms of which a block holds several (upstream: 32 threads = 128 / N transforms; the _wave64
# classes: 64 threads = 256 / N).  Images: address bit b of the block-level index q = f N + p takes the parity of q & SMALL_MASKS[N][b] in
# (found by search_small(): scattered stores and slot-(2,3) reads conflict free for 32- and 64-thread blocks).
SMALL_MASKS = {128: [16, 64, 96, 32, 0], 64: [16, 4, 72, 32, 0], 32: [144, 4, 32, 64, 0]}


def small_image(N, q):
    a = q
    for b, m in enumerate(SMALL_MASKS[N]):
        if bin(q & m).count("1") & 1:
            a ^= 1 << b
    return a


def small_readback_position(N, u, j):
    """position (within its transform) that the thread with index u in its transform reads into slot j"""
    if N == 128:
        return (u & 3) | (j << 2) | (((u >> 2) & 3) << 4) | ((u >> 4) << 6)
    return (u & 3) | (j << 2) | ((u >> 2) << 4)


def transform_small(N, DIR, xs, log=None):
    """xs: (transforms of the block, N) natural-order inputs; returns their DFTs.  One wave: lane = f Q + u"""
    F = xs.shape[0]
    Q, TB = N // 4, (N // 4).bit_length() - 1
    lanes = F * Q
    assert lanes in (32, 64)
    sign = 1 if DIR else -1
    img = np.zeros(F * N, complex)

    def note(kind, addr):
        if log is not None:
            log.append((kind, list(addr) + [addr[0]] * (64 - len(addr))))      # (inactive lanes of a 32-thread block take no part)
    stores = [[] for _ in range(4)]
    for m in range(4):
        note("r", [(l // Q) * N + (l % Q) + m * Q for l in range(lanes)])
    for l in range(lanes):
        f, u = l // Q, l % Q
        e = [0] * 4
        for m in range(4):
            e[((m & 1) << 1) | (m >> 1)] = xs[f, u + m * Q]
        s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
        jd1 = d1 * (1j * sign)
        a = f * N + 4 * rev(u, TB)
        for i, v in enumerate([s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]):
            img[small_image(N, a + i)] = v
            stores[i].append(small_image(N, a + i))
    for i in range(4):
        note("w", stores[i])
    E = np.zeros((lanes, 4), complex)
    P_ = np.zeros((lanes, 4), int)
    for j in range(4):
        addr = []
        for l in range(lanes):
            f, u = l // Q, l % Q
            p = small_readback_position(N, u, j)
            E[l, j], P_[l, j] = img[small_image(N, f * N + p)], p
            addr.append(small_image(N, f * N + p))
        note("r", addr)

    def swap(slot_bit, lane_bit):
        nonlocal E, P_
        ne, npos = E.copy(), P_.copy()
        for l in range(lanes):
            partner = l ^ (1 << lane_bit)
            for i in range(4):
                if (i >> slot_bit) & 1:
                    continue
                hi = i | (1 << slot_bit)
                if (l >> lane_bit) & 1 == 0:
                    ne[l, hi], npos[l, hi] = E[partner, i], P_[partner, i]
                else:
                    ne[l, i], npos[l, i] = E[partner, hi], P_[partner, hi]
        E, P_ = ne, npos
    for l in range(lanes):
        E[l] = quad(E[l], P_[l], 4, sign)
    if N >= 64:
        swap(0, 2)
        swap(1, 3)
        for l in range(lanes):
            E[l] = quad(E[l], P_[l], 16, sign)
    out = np.zeros((F, N), complex)
    if N == 64:
        for i in range(4):
            note("w", [(l // Q) * N + int(P_[l, i]) for l in range(lanes)])
        for l in range(lanes):
            for i in range(4):
                assert int(P_[l, i]) == (l % Q) + 16 * i
                out[l // Q, int(P_[l, i])] = E[l, i]
        return out
    swap(0, 4 if N == 128 else 2)
    half = N // 2
    for l in range(lanes):
        u = l % Q
        assert [int(q) for q in P_[l]] == [u, u + half, u + Q, u + half + Q], (N, l, P_[l])
        wv = np.exp(sign * 2j * np.pi * u / N)
        x0, x1, x2, x3 = E[l]
        t1, t3 = x1 * wv, x3 * wv * (1j * sign)
        out[l // Q, u], out[l // Q, u + half], out[l // Q, u + Q], out[l // Q, u + half + Q] = x0 + t1, x0 - t1, x2 + t3, x2 - t3
    for off in (0, half, Q, half + Q):
        note("w", [(l // Q) * N + (l % Q) + off for l in range(lanes)])
    return out


def transform_small_noreorder(N, DIR, xs, log=None):
    """N = 64, 128 without reorder (quarter_small_noreorder): pass 0 on x[4 u + i], the exchange with lane bits 0, 1, pass 1, the MIDDLE exchange
    through the block's image (exchange_image of the block-level index), pass 2, and N = 128's radix-2 pass behind a swap with lane bit 4"""
    F = xs.shape[0]
    Q = N // 4
    lanes = F * Q
    sign = 1 if DIR else -1
    img = np.zeros(F * N, complex)
    E = np.zeros((lanes, 4), complex)
    P_ = np.zeros((lanes, 4), int)

    def note(kind, addr):
        if log is not None:
            log.append((kind, list(addr) + [addr[0]] * (64 - len(addr))))

    def swap(slot_bit, lane_bit):
        nonlocal E, P_
        ne, npos = E.copy(), P_.copy()
        for l in range(lanes):
            partner = l ^ (1 << lane_bit)
            for i in range(4):
                if (i >> slot_bit) & 1:
                    continue
                hi = i | (1 << slot_bit)
                if (l >> lane_bit) & 1 == 0:
                    ne[l, hi], npos[l, hi] = E[partner, i], P_[partner, i]
                else:
                    ne[l, i], npos[l, i] = E[partner, hi], P_[partner, hi]
        E, P_ = ne, npos
    for l in range(lanes):
        f, u = l // Q, l % Q
        e = [xs[f, 4 * u + i] for i in range(4)]
        s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
        jd1 = d1 * (1j * sign)
        E[l] = [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]
        P_[l] = [4 * u + i for i in range(4)]
    swap(0, 0)
    swap(1, 1)
    for l in range(lanes):
        E[l] = quad(E[l], P_[l], 4, sign)
    for j in range(4):
        addr = []
        for l in range(lanes):
            f, u = l // Q, l % Q
            p = (u & 3) + 4 * j + 16 * (u >> 2)
            assert int(P_[l, j]) == p
            img[exchange_image(f * N + p)] = E[l, j]
            addr.append(exchange_image(f * N + p))
        note("w", addr)
    for j in range(4):
        addr = []
        for l in range(lanes):
            f, u = l // Q, l % Q
            p = (u & 15) + 16 * j + 64 * (u >> 4)
            E[l, j], P_[l, j] = img[exchange_image(f * N + p)], p
            addr.append(exchange_image(f * N + p))
        note("r", addr)
    for l in range(lanes):
        E[l] = quad(E[l], P_[l], 16, sign)
    out = np.zeros((F, N), complex)
    if N == 64:
        for l in range(lanes):
            for i in range(4):
                assert int(P_[l, i]) == (l % Q) + 16 * i
                out[l // Q, int(P_[l, i])] = E[l, i]
        return out
    swap(0, 4)
    for l in range(lanes):
        u = l % Q
        assert [int(q) for q in P_[l]] == [u, u + 64, u + 32, u + 96]
        wv = np.exp(sign * 2j * np.pi * u / N)
        x0, x1, x2, x3 = E[l]
        t1, t3 = x1 * wv, x3 * wv * (1j * sign)
        out[l // Q, u], out[l // Q, u + 64], out[l // Q, u + 32], out[l // Q, u + 96] = x0 + t1, x0 - t1, x2 + t3, x2 - t3
    return out


def search_small(N, trials=200000, seed=1):
    """random search over images whose address bits 0 ... 4 take the parity of at most two HIGHER bits of the block-level index in: the first
    found with scattered stores (groups of 16 contiguous lanes, mod 16) and slot-(2,3) reads (groups of 32, mod 32) conflict free for
    blocks of 32 and of 64 threads"""
    import random
    rnd = random.Random(seed)
    Q, TB = N // 4, (N // 4).bit_length() - 1
    cand = [[m for m in range(256) if m & ((1 << (b + 1)) - 1) == 0 and bin(m).count("1") <= 2] for b in range(5)]

    def image(q, masks):
        a = q
        for b, m in enumerate(masks):
            if bin(q & m).count("1") & 1:
                a ^= 1 << b
        return a

    def free(masks, lanes):
        for i in range(4):
            row = [image((l // Q) * N + 4 * rev(l % Q, TB) + i, masks) for l in range(lanes)]
            if any(len({a % 16 for a in row[g:g + 16]}) != 16 for g in range(0, lanes, 16)):
                return False
        for j in range(4):
            row = [image((l // Q) * N + small_readback_position(N, l % Q, j), masks) for l in range(lanes)]
            if any(len({a % 32 for a in row[g:g + 32]}) != 32 for g in range(0, lanes, 32)):
                return False
        return True
    for _ in range(trials):
        masks = [rnd.choice(cand[b]) for b in range(5)]
        if free(masks, 32) and free(masks, 64):
            return masks
    return None


def check(verbose=False):
    worst = 0.0
    rng = np.random.default_rng(1)
    n_rev = lambda N: np.array([rev(i, N.bit_length() - 1) for i in range(N)])
    for N in (256, 512, 1024, 2048, 4096):
        for DIR in (0, 1):
            for REO in (1, 0):
                x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
                got = transform(N, DIR, REO, x)
                xin = x if REO else x[n_rev(N)]
                want = (np.fft.ifft(xin) * N) if DIR else np.fft.fft(xin)
                err = np.abs(got - want).max() / np.abs(want).max()
                worst = max(worst, err)
                if verbose:
                    print(f"N={N} dir={DIR} reorder={REO}: max error {err:.2e}")
    for N in (32, 64, 128):
        for lanes in (32, 64):
            F = lanes // (N // 4)
            for DIR in (0, 1):
                xs = rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N))
                got = transform_small(N, DIR, xs)
                want = (np.fft.ifft(xs, axis=-1) * N) if DIR else np.fft.fft(xs, axis=-1)
                err = np.abs(got - want).max() / np.abs(want).max()
                worst = max(worst, err)
                if verbose:
                    print(f"N={N} natural order, {lanes}-thread block ({F} transforms) dir={DIR}: max error {err:.2e}")
                if N >= 64:
                    log = []
                    got = transform_small_noreorder(N, DIR, xs, log)
                    xr = xs[:, n_rev(N)]
                    want = (np.fft.ifft(xr, axis=-1) * N) if DIR else np.fft.fft(xr, axis=-1)
                    err = np.abs(got - want).max() / np.abs(want).max()
                    worst = max(worst, err)
                    for kind, addr in log:      # the middle exchange through the block's image: conflict free in both block shapes
                        group, banks = (32, 32) if kind == "r" else (16, 16)
                        for g in range(0, lanes, group):
                            assert len({a % banks for a in addr[g:g + group]}) == group, (N, lanes, kind)
                    if verbose:
                        print(f"N={N} no reorder, {lanes}-thread block dir={DIR}: max error {err:.2e}; exchange conflict free")
    return worst


def lds_report_small(N, lanes):
    log = []
    F = lanes // (N // 4)
    transform_small(N, 0, np.zeros((F, N), complex), log)
    groups = 1 if lanes == 32 else 2
    total = ideal = 0
    for kind, addr in log:
        a = addr[:lanes] + [addr[0]] * (64 - lanes)
        if kind == "r":
            c = sum(max(collections_count([x % 32 for x in set(a[g:g + 32])])) for g in range(0, lanes, 32))
            total, ideal = total + c, ideal + groups
        else:
            c = sum(max(collections_count([x % 16 for x in set(a[g:g + 16])])) for g in range(0, lanes, 16))
            total, ideal = total + max(6, c), ideal + 6
    return total, ideal


def collections_count(values):
    load = {}
    for v in values:
        load[v] = load.get(v, 0) + 1
    return load.values()


if __name__ == "__main__":
    import sys
    if "--search" in sys.argv:
        sols = search_256()
        print(len(sols), "conflict-free images of N = 256; the header's is one of them:", M256 in sols)
    assert all(apply_rows(M256, p) == image256(p) for p in range(256))
    print("worst relative error", check(verbose=True))
    for N in (32, 64, 128):
        for lanes in (32, 64):
            print(f"N={N} natural order, {lanes}-thread block: LDS cycles / conflict free {lds_report_small(N, lanes)}")
    for N in (256, 512, 1024, 2048, 4096):
        for REO in (1, 0):
            total, ideal, count = lds_report(N, REO)
            print(f"N={N} reorder={REO}: {count} wave-level LDS accesses per transform, {total} LDS cycles ({ideal} conflict free)")
    for N in (256, 512, 1024, 2048, 4096):
        total, ideal, count = lds_report_pairs(N)
        print(f"N={N} natural order in pairs of passes: {count} wave-level LDS accesses per transform, {total} LDS cycles ({ideal} conflict free)")
