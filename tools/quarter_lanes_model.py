"""NumPy model of the N <= 256 form of the reference-contract engine (include/smfft/smfft_device_functions.hpp, QuarterLanes): the
radix-2^2 decimation-in-time ladder on four elements per thread in which the exchange between two passes is a pair of one-bit
transposes between a slot bit of the thread's four registers and a lane bit -- no LDS.  Replays lanes, slots and swaps and
compares with numpy.fft for N = 32 ... 256, both directions, natural order and no reorder; prints where the results end up
(slot i of thread t holds element base + (N/4) * sigma(i), base = rev(t) or t).  N = 512 ... 4096 (model_large): the same lane
passes inside every aligned block of 256 elements -- all four of them without reorder, passes 1 ... 3 after the scattered first
pass in natural order -- and the passes that cross waves on the image.  CPU only; tests/test_quarter_swizzle_model.py runs check()."""
import numpy as np

def rev(v, bits):
    r = 0
    for b in range(bits):
        r |= ((v >> b) & 1) << (bits - 1 - b)
    return r

def model(N, DIR, REORDER, x, skip_pass0=False):
    """returns (result array in natural order assembled from the tracked positions, jmap[t][i]); skip_pass0: x holds the results of
    pass 0 already (the natural-order variants of N >= 512 enter the lane passes after their scattered first pass)"""
    n = N.bit_length() - 1
    Q = N // 4
    TB = n - 2
    kP = n // 2
    odd = n & 1
    sign = 1 if DIR else -1
    W = lambda M, k: np.exp(sign * 2j * np.pi * k / M)
    e = np.zeros((Q, 4), complex)      # e[t][slot]
    jidx = np.zeros((Q, 4), int)       # which j each register holds (tracking only)
    for t in range(Q):
        if REORDER:
            r = rev(t, TB)
            for m in range(4):
                i = ((m & 1) << 1) | (m >> 1)
                e[t, i] = x[t + m * Q]
                jidx[t, i] = 4 * r + i
        else:
            for i in range(4):
                e[t, i] = x[4 * t + i]
                jidx[t, i] = 4 * t + i
    def lane_bit_of_jbit(b):     # which t bit holds j bit (2 + b) initially
        return (TB - 1 - b) if REORDER else b
    def swap(slot_bit, lane_bit):
        nonlocal e, jidx
        ne, nj = e.copy(), jidx.copy()
        for t in range(Q):
            p = t ^ (1 << lane_bit)
            for i in range(4):
                if (i >> slot_bit) & 1:
                    continue
                hi = i | (1 << slot_bit)
                if (t >> lane_bit) & 1 == 0:
                    # keeps A (slot i), gets partner's A into slot hi
                    ne[t, hi] = e[p, i]; nj[t, hi] = jidx[p, i]
                else:
                    # keeps B (slot hi), gets partner's B into slot i
                    ne[t, i] = e[p, hi]; nj[t, i] = jidx[p, hi]
        e, jidx = ne, nj
    # pass 0
    for t in range(Q if not skip_pass0 else 0):
        e0, e1, e2, e3 = e[t]
        s0, d0, s1, d1 = e0 + e1, e0 - e1, e2 + e3, e2 - e3
        jd1 = d1 * (1j * sign)
        e[t] = [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]
    P = 4
    for p in range(1, kP):
        swap(0, lane_bit_of_jbit(2 * p - 2))
        swap(1, lane_bit_of_jbit(2 * p - 1))
        for t in range(Q):
            k = (rev(t, TB) if REORDER else t) & (P - 1)
            w2 = W(4 * P, k); w1 = w2 * w2
            x0, x1, x2, x3 = e[t]
            t1, t3 = x1 * w1, x3 * w1
            y0, y1, y2, y3 = x0 + t1, x0 - t1, x2 + t3, x2 - t3
            u2, v3 = y2 * w2, y3 * w2
            u3 = v3 * (1j * sign)
            e[t] = [y0 + u2, y1 + u3, y0 - u2, y1 - u3]
        P *= 4
    if odd:
        swap(0, lane_bit_of_jbit(n - 3))
        for t in range(Q):
            k = (rev(t, TB) if REORDER else t) & (N // 4 - 1)
            w = W(N, k)
            x0, x1, x2, x3 = e[t]
            t1 = x1 * w
            t3 = x3 * w * (1j * sign)
            e[t] = [x0 + t1, x0 - t1, x2 + t3, x2 - t3]
    # after the last pass the j of a register: the butterfly outputs replace ... positions: slot i <-> j bits; tracking of jidx
    # follows inputs; outputs of an in-place butterfly keep the slot's j
    out = np.zeros(N, complex)
    for t in range(Q):
        for i in range(4):
            out[jidx[t, i]] = e[t, i]
    return out, jidx

def model_large(N, DIR, REORDER, x):
    """N >= 512 (quarter_fft, kLanesHead / kLanesMiddle): the wave-local passes of every aligned block of 256 elements on lanes
    (the 256-point ladder above: all four passes without reorder, passes 1 ... 3 after the scattered pass 0 in natural order), the
    passes that cross waves (P >= 256, the radix-2 pass) on the image y, indexed naturally (the swizzle is a renaming)."""
    n = N.bit_length() - 1
    Q, TB = N // 4, n - 2
    sign = 1 if DIR else -1
    W = lambda M, k: np.exp(sign * 2j * np.pi * k / M)
    y = np.zeros(N, complex)
    if REORDER:
        for t in range(Q):                       # pass 0: loads x[t + m N/4], stores 4 rev(t) + i
            e = [0] * 4
            for m in range(4):
                e[((m & 1) << 1) | (m >> 1)] = x[t + m * Q]
            s0, d0, s1, d1 = e[0] + e[1], e[0] - e[1], e[2] + e[3], e[2] - e[3]
            jd1 = d1 * (1j * sign)
            a = 4 * rev(t, TB)
            y[a:a + 4] = [s0 + s1, d0 + jd1, s0 - s1, d0 - jd1]
    for w in range(N // 256):                    # the wave's block: thread (w, lane) enters with elements 256 w + 4 lane + i
        blk = (y if REORDER else x)[256 * w:256 * w + 256]
        out, jmap = model(256, DIR, 0, blk, skip_pass0=bool(REORDER))
        for lane in range(64):
            assert list(jmap[lane]) == [lane + 64 * i for i in range(4)]      # ... and leaves with 256 w + lane + 64 i
        y[256 * w:256 * w + 256] = out
    P = 256
    for p in range(4, n // 2):                   # passes through LDS: k = t mod P, base = (t - k) * 4 + k
        for t in range(Q):
            k = t & (P - 1)
            base = ((t - k) << 2) + k
            w2 = W(4 * P, k); w1 = w2 * w2
            x0, x1, x2, x3 = y[base], y[base + P], y[base + 2 * P], y[base + 3 * P]
            t1, t3 = x1 * w1, x3 * w1
            y0, y1, y2, y3 = x0 + t1, x0 - t1, x2 + t3, x2 - t3
            u2, v3 = y2 * w2, y3 * w2
            u3 = v3 * (1j * sign)
            y[base], y[base + P], y[base + 2 * P], y[base + 3 * P] = y0 + u2, y1 + u3, y0 - u2, y1 - u3
        P *= 4
    if n & 1:
        for t in range(Q):
            w = W(N, t)
            x0, x1, x2, x3 = y[t], y[t + N // 2], y[t + Q], y[t + 3 * Q]
            t1, t3 = x1 * w, x3 * w * (1j * sign)
            y[t], y[t + N // 2], y[t + Q], y[t + 3 * Q] = x0 + t1, x0 - t1, x2 + t3, x2 - t3
    return y


def bitrev_perm(N):
    n = N.bit_length() - 1
    return np.array([rev(i, n) for i in range(N)])

def check(verbose=False):
    """max relative error over every case, and the assertion that slot i of thread t ends with element base + (N/4) * sigma(i)"""
    worst = 0.0
    rng = np.random.default_rng(0)
    for N in (32, 64, 128, 256):
        n, Q = N.bit_length() - 1, N // 4
        for DIR in (0, 1):
            for REO in (1, 0):
                x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
                got, jmap = model(N, DIR, REO, x)
                xin = x if REO else x[bitrev_perm(N)]
                want = (np.fft.ifft(xin) * N) if DIR else np.fft.fft(xin)
                err = np.abs(got - want).max() / np.abs(want).max()
                worst = max(worst, err)
                sigma = [0, 2, 1, 3] if n & 1 else [0, 1, 2, 3]
                for t in range(Q):
                    base = rev(t, n - 2) if REO else t
                    assert list(jmap[t]) == [base + Q * sigma[i] for i in range(4)], (N, REO, t)
                if verbose:
                    print(f"N={N} dir={DIR} reorder={REO}: max error {err:.2e}; thread 1 ends with elements {list(map(int, jmap[1]))}")
    for N in (512, 1024, 2048, 4096):
        for DIR in (0, 1):
            for REO in (1, 0):
                x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
                got = model_large(N, DIR, REO, x)
                xin = x if REO else x[bitrev_perm(N)]
                want = (np.fft.ifft(xin) * N) if DIR else np.fft.fft(xin)
                err = np.abs(got - want).max() / np.abs(want).max()
                worst = max(worst, err)
                if verbose:
                    print(f"N={N} dir={DIR} reorder={REO}: lane passes inside every block of 256, max error {err:.2e}")
    return worst


if __name__ == "__main__":
    print("worst relative error", check(verbose=True))
