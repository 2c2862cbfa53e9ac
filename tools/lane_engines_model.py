"""NumPy replay of the lane engines of the in-LDS path (include/smfft/smfft_engine.hpp: PairEngine32, QuadEngine64) -- lanes, registers,
layouts, sign vectors, the renaming of the no-reorder variants, the pieces of a cut chain -- against numpy.fft (CPU only; a test
runs it).  It checks the BOOK-KEEPING the kernels rely on, not their arithmetic:
  * N = 32, natural order: applications alternate dit (layout A -> B) and dif (B -> A); lane 1 is negated after a dit, plain after a dif;
  * N = 32, no reorder: dit every time on the renamed registers, the sign vector negated on odd applications;
  * N = 64, no reorder: lane j holds the stored block rev2(j) before and after every application; s1 = (+,+,-,-), s2 = (+,-,+,-) on
    even applications, negated on odd ones; lanes 1 and 2 negated after an even application; lane 3 turns by -+i between the stages;
  * a chain cut anywhere (load / store with the sign-bit flip of the odd pieces) gives what the uncut chain gives.
    python tools/lane_engines_model.py"""
import numpy as np


def rev(v, bits):
    return int(format(v, f"0{bits}b")[::-1], 2) if bits else 0


def dft16(x, inverse):
    return np.fft.ifft(x) * 16 if inverse else np.fft.fft(x)


def w(n, m, inverse):
    return np.exp((2j if inverse else -2j) * np.pi * m / n)


class Pair32:
    """two lanes u = 0, 1 with sixteen registers each"""
    def __init__(self, inverse, reorder):
        self.inv, self.reorder = inverse, reorder
        self.tw = [np.array([w(32, u * q, inverse) for q in range(16)]) for u in (0, 1)]
        self.s_plain = np.array([1.0, -1.0])

    def cross(self, r, s):                       # own <- own + s * partner's own
        return [r[0] + s[0] * r[1], r[1] + s[1] * r[0]]

    def dit(self, r, s):
        y = [dft16(r[u], self.inv) * self.tw[u] for u in (0, 1)]
        return self.cross(y, s)

    def dif(self, r, s):
        r = self.cross(r, s)
        return [dft16(r[u] * self.tw[u], self.inv) for u in (0, 1)]

    def apply(self, r, odd):
        s = -self.s_plain if odd else self.s_plain
        if self.reorder:
            return self.dif(r, s) if odd else self.dit(r, s)
        x = [np.array([r[u][rev(c, 4)] for c in range(16)]) for u in (0, 1)]
        return self.dit(x, s)

    def load(self, image, f0):                   # image: the 32 stored elements
        odd = f0 & 1
        if self.reorder and not odd:
            return [image[u::2].copy() for u in (0, 1)]                       # layout A
        r = [image[16 * u:16 * u + 16].copy() for u in (0, 1)]              # layout B
        if odd:
            r[1] = -r[1]
        return r

    def store(self, r, f1):
        odd = f1 & 1
        image = np.zeros(32, complex)
        if self.reorder and not odd:
            for u in (0, 1):
                image[u::2] = r[u]
        else:
            image[0:16] = r[0]
            image[16:32] = -r[1] if odd else r[1]
        return image


class Quad64:
    """four lanes j = 0 ... 3 with sixteen registers each; no reorder"""
    def __init__(self, inverse):
        self.inv = inverse
        self.tw = [np.array([w(64, j * q, inverse) for q in range(16)]) for j in range(4)]
        self.s1 = np.array([1.0, 1.0, -1.0, -1.0])
        self.s2 = np.array([1.0, -1.0, 1.0, -1.0])
        self.turn = 1j if inverse else -1j

    def apply(self, r, odd):
        x = [np.array([r[j][rev(c, 4)] for c in range(16)]) for j in range(4)]
        y = [dft16(x[j], self.inv) * self.tw[j] for j in range(4)]
        s1 = -self.s1 if odd else self.s1
        y = [y[j] + s1[j] * y[j ^ 2] for j in range(4)]
        y[3] = self.turn * y[3]
        s2 = -self.s2 if odd else self.s2
        return [y[j] + s2[j] * y[j ^ 1] for j in range(4)]

    def load(self, image, f0):
        r = [image[16 * rev(j, 2):16 * rev(j, 2) + 16].copy() for j in range(4)]
        if f0 & 1:
            r[1], r[2] = -r[1], -r[2]
        return r

    def store(self, r, f1):
        image = np.zeros(64, complex)
        for j in range(4):
            flip = (f1 & 1) and j in (1, 2)
            image[16 * rev(j, 2):16 * rev(j, 2) + 16] = -r[j] if flip else r[j]
        return image


def reference(x, n, inverse, reorder, count):
    bits = n.bit_length() - 1
    perm = np.array([rev(i, bits) for i in range(n)])
    for _ in range(count):
        v = x if reorder else x[perm]
        x = np.fft.ifft(v) * n if inverse else np.fft.fft(v)
    return x


def run_chain(engine, x, cuts, count):
    """the chain's applications 0 ... count - 1 in pieces [0, c1), [c1, c2), ...: every piece loads the image, applies, stores it"""
    image = x.copy()
    bounds = [0] + list(cuts) + [count]
    for f0, f1 in zip(bounds[:-1], bounds[1:]):
        r = engine.load(image, f0)
        for f in range(f0, f1):
            r = engine.apply(r, f & 1)
        image = engine.store(r, f1)
    return image


def check(verbose=False):
    rng = np.random.default_rng(11)
    worst = 0.0
    for name, n, make in (("pair32 natural order", 32, lambda inv: Pair32(inv, True)), ("pair32 no reorder", 32, lambda inv: Pair32(inv, False)), ("quad64 no reorder", 64, lambda inv: Quad64(inv))):
        reorder = "natural" in name
        for inverse in (False, True):
            x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            for count in (1, 2, 3, 4, 7):
                want = reference(x, n, inverse, reorder, count)
                for cuts in [()] + [(c,) for c in range(1, count)] + ([(1, 2), (2, 5), (3, 4, 6)] if count == 7 else []):
                    got = run_chain(make(inverse), x, cuts, count)
                    err = np.abs(got - want).max() / np.abs(want).max()
                    worst = max(worst, err)
                    assert err < 1e-12, (name, inverse, count, cuts, err)
        if verbose:
            print(f"{name}: every count in (1, 2, 3, 4, 7), every single cut and three multiple cuts, both directions: equal to numpy.fft")
    return worst


if __name__ == "__main__":
    print(f"worst relative deviation {check(verbose=True):.1e}")
