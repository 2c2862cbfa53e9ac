"""Lane-accurate fp64 emulation of the reference's CUDA thread blocks (32-lane warps).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing in the product imports this.

Why it exists: oracle/smfft_oracle.c and oracle/np_reference.py state WHAT the reference computes
(semantics S1..S6 of SURVEY.md 8(a)).  That statement was derived by reading the reference; this
module closes the loop by replaying the reference's own thread choreography -- every thread of a
block, every register, every shared-memory cell, every warp shuffle -- in complex128 and checking
that the result is what the oracle says (tests/test_lane_emulation.py).  It follows, step by step:

  CT  SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu
        :54-124   reorder_{4,8,16,32}_register     -> _lane_bitrev
        :126-329  reorder_{32..4096}<P>            -> CtBlock.reorder
        :334-532  do_SMFFT_CT_DIT<P>               -> CtBlock.fft
        :534-551  SMFFT_DIT_external<P>            -> ct_external
      SMFFT_CooleyTukey_C2C/SM_FFT_parameters.cuh:1-390 (fft_length, fft_sm_required per class)
  ST  SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu:97-240, 243-258   -> stockham_block / st_external
  RC  SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:106-266       -> stockham_block
        :269-344  do_FFT_Stockham_R2C_C2R<P,D>     -> r2c_c2r_block
        :349-365  FFT_GPU_R2C_C2R_external<P,D>    -> rc_external

All threads of a block are evaluated together as NumPy vectors indexed by threadIdx.x; a statement
of the reference that reads shared memory / shuffles and a later statement that writes are
separated exactly where the reference has __syncthreads / __syncwarp, so the vectorised order of
evaluation is one of the interleavings the reference allows.  Twiddles are exact (fp64) instead of
the reference's fast-math sincosf: the emulation pins the ALGORITHM, the fp32 error budget is the
parity tests' business.
"""
import numpy as np

WARP = 32  # the reference's FFT_Params::warp (SM_FFT_parameters.cuh:5)


def _brev(v, bits):
    """bit reversal of the low `bits` bits of every entry of v (the reference's __brev(x) >> (32 - bits))"""
    v = np.asarray(v)
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (bits - 1 - b)
    return out


def _w(n, m, inverse):
    """Get_W_value / Get_W_value_inverse (CT:18-28, RC:83-94): e^{-+ 2 pi i m / n}"""
    ang = 2.0 * np.pi * np.asarray(m, dtype=np.float64) / float(n)
    return np.cos(ang) + (1j if inverse else -1j) * np.sin(ang)


# ------------------------------------------------------------------------------------------------
# Cooley-Tukey program
# ------------------------------------------------------------------------------------------------
def ct_params(n):
    """(fft_exp, fft_length, fft_sm_required) of the reference's classes, SM_FFT_parameters.cuh:
    N = 32: 128 elements per block (4 FFTs), 128 float2 of shared memory (:8-18); N = 64, 128: 128 per
    block, 132 (:56-66, :104-114); larger N: one FFT per block, (N / 32) * 33 (README.md:18)."""
    exp = int(np.log2(n))
    length = max(n, 128)
    sm = 128 if n == 32 else (length // 32) * 33
    return exp, length, sm


class CtBlock:
    """One thread block of SMFFT_DIT_external<P>: fft_length / 4 threads, 4 registers A..D per thread."""

    def __init__(self, n, inverse, reorder):
        self.n = n
        self.exp, self.length, sm = ct_params(n)
        self.inverse = bool(inverse)
        self.do_reorder = bool(reorder)
        self.nthreads = self.length // 4
        self.tid = np.arange(self.nthreads)
        self.lid = self.tid & (WARP - 1)
        self.wid = self.tid // WARP
        self.s = np.zeros(sm, dtype=np.complex128)
        self.reg = [np.zeros(self.nthreads, dtype=np.complex128) for _ in range(4)]

    # -- warp shuffles -------------------------------------------------------------------------
    def _shfl(self, v, target_lane):
        """__shfl_sync(full mask, v, lane): every thread reads v of lane `target_lane` of ITS warp"""
        return v[self.wid * WARP + target_lane]

    def _shfl_xor(self, v, mask):
        return v[self.tid ^ mask]

    def _lane_bitrev(self, bits):
        """reorder_{4,8,16,32}_register (CT:54-124): all four registers move to the lane whose low
        `bits` lane bits are reversed (lanes in groups of 2^bits)"""
        group = 1 << bits
        target = _brev(self.lid & (group - 1), bits) + group * (self.lid >> bits)
        self.reg = [self._shfl(r, target) for r in self.reg]

    # -- bit reversal of the block's data (CT:126-329) -----------------------------------------
    def _transpose_33(self, store_pos, read_pos, read_offsets):
        """the padded (stride 33) shared-memory transposition every reorder_N is built around:
        A, B, C, D -> s[store_pos + {0, 33, 66, 99}], barrier, A..D <- s[read_pos + read_offsets]"""
        for k, off in enumerate((0, 33, 66, 99)):
            self.s[store_pos + off] = self.reg[k]
        self.reg = [self.s[read_pos + off].copy() for off in read_offsets]

    def _reorder_64(self):      # CT:133-156
        lid, wid = self.lid, self.wid
        self._lane_bitrev(5)
        self._transpose_33((lid >> 4) + 2 * (lid & 15) + wid * 132, (lid & 1) * 32 + lid + wid * 132, (0, 1, 66, 67))

    def _reorder_128(self):     # CT:159-185
        lid, wid = self.lid, self.wid
        self._lane_bitrev(5)
        self._transpose_33((lid >> 3) + 4 * (lid & 7) + wid * 132, (lid & 3) * 32 + lid + wid * 132, (0, 1, 2, 3))
        self._lane_bitrev(2)

    def reorder(self):
        lid, wid, e = self.lid, self.wid, self.exp
        if e == 5:              # CT:126-129
            self._lane_bitrev(5)
        elif e == 6:
            self._reorder_64()
        elif e == 7:
            self._reorder_128()
        elif e in (8, 9, 10):   # CT:188-268: same shape, the split between lane bits and rows moves with N
            k = 10 - e          # 2, 1, 0
            self._lane_bitrev(5)
            store = (lid >> k) + (32 >> k) * (lid & ((1 << k) - 1)) + wid * 132
            read = (lid & ((32 >> k) - 1)) * 32 + lid + wid * 4
            self._transpose_33(store, read, (0, 1, 2, 3))
            self._lane_bitrev(5 - k)
        else:                   # CT:270-329: coarse transposition over the warps, then reorder_64 / reorder_128
            self._lane_bitrev(5)
            store = lid + wid * 132
            if e == 11:
                self._transpose_33(store, lid * 33 + wid * 2, (0, 1056, 1, 1057))
                self._reorder_64()
            else:
                self._transpose_33(store, lid * 33 + wid, (0, 1056, 2112, 3168))
                self._reorder_128()

    # -- the transform (CT:334-532) --------------------------------------------------------------
    def fft(self):
        lid, wid, tid, s = self.lid, self.wid, self.tid, self.s
        base = lid + (wid << 2) * WARP
        self.reg = [s[base + k * WARP].copy() for k in range(4)]                    # CT:345-350
        if self.do_reorder:
            self.reorder()                                                          # CT:352-361
        # stage 1 (CT:367-378): v = parity * v + partner, parity = +1 on even lanes
        parity = 1 - 2 * (lid & 1)
        self.reg = [parity * r + self._shfl_xor(r, 1) for r in self.reg]
        # stages 2..5 in registers (CT:381-411)
        pot, potp1 = 2, 4
        for q in range(1, 5):
            m = lid & (potp1 - 1)
            hi = m >> q
            w = _w(potp1, hi * m, self.inverse)
            t = [w * r for r in self.reg]
            self.reg = [x + (2 * hi - 1) * self._shfl_xor(x, pot) for x in t]
            pot, potp1 = pot << 1, potp1 << 1
        for k in range(4):                                                           # CT:413-417
            s[base + k * WARP] = self.reg[k]
        # shared-memory stages: two butterflies of two adjacent sub-FFTs per thread (CT:419-490)
        first = 5
        if self.exp == 6:                                                            # CT:419-454
            self._lds_stage(5, pot, potp1)
            pot, potp1 = pot << 1, potp1 << 1
            first = 6   # the generic loop below is empty for fft_exp = 6
        for q in range(first, self.exp - 1):                                         # CT:456-490
            self._lds_stage(q, pot, potp1)
            pot, potp1 = pot << 1, potp1 << 1
        if self.exp > 6:                                                             # CT:493-531
            m = tid
            w = _w(potp1, m, self.inverse)
            ia, ib, ic, idd = m, m + pot, m + (pot >> 1), m + 3 * (pot >> 1)
            a, b, c, d = s[ia].copy(), s[ib].copy(), s[ic].copy(), s[idd].copy()
            # second butterfly: twiddle index m + N/4, i.e. W * (-i) forward, W * (+i) inverse (CT:514-525)
            w2 = w * (1j if self.inverse else -1j)
            s[ia], s[ib] = a + w * b, a - w * b
            s[ic], s[idd] = c + w2 * d, c - w2 * d

    def _lds_stage(self, q, pot, potp1):
        s, tid = self.s, self.tid
        m = tid & (pot - 1)
        j = tid >> q
        w = _w(potp1, m, self.inverse)
        ia = j * (potp1 << 1) + m
        ib, ic, idd = ia + pot, ia + potp1, ia + 3 * pot
        a, b, c, d = s[ia].copy(), s[ib].copy(), s[ic].copy(), s[idd].copy()
        s[ia], s[ib] = a + w * b, a - w * b
        s[ic], s[idd] = c + w * d, c - w * d


def ct_external(x, inverse, reorder, direction_override=None):
    """SMFFT_DIT_external<P> over a batch (CT:534-551 + the launch shape of CT:586-595): x is
    (nFFTs, N); for N = 32 / 64 a block carries 4 / 2 FFTs (nFFTs must be a multiple).
    direction_override: the fft_direction member the class really has, for the one class whose
    value differs from its name (FFT_4096_inverse_noreorder::fft_direction = 0, SM_FFT_parameters.cuh:388)."""
    x = np.asarray(x, dtype=np.complex128)
    nffts, n = x.shape
    _, length, _ = ct_params(n)
    per_block = length // n
    assert nffts % per_block == 0
    flat = x.reshape(-1)
    out = np.empty_like(flat)
    direction = inverse if direction_override is None else direction_override
    for b in range(nffts // per_block):
        blk = CtBlock(n, direction, reorder)
        q = length // 4
        for k in range(4):       # CT:538-541
            blk.s[blk.tid + k * q] = flat[blk.tid + b * length + k * q]
        blk.fft()
        for k in range(4):       # CT:547-550
            out[blk.tid + b * length + k * q] = blk.s[blk.tid + k * q]
    return out.reshape(nffts, n)


# ------------------------------------------------------------------------------------------------
# Stockham programs
# ------------------------------------------------------------------------------------------------
def stockham_block(s, n, inverse):
    """do_FFT_Stockham_mk6<P> (ST:97-240; sign fixed to +, ST:76) and do_FFT_Stockham_C2C<P,D>
    (RC:106-266): N/4 threads, two radix-2 butterflies per thread per stage, autosort, in place on s[0..N)."""
    exp = int(np.log2(n))
    tid = np.arange(n // 4)
    half, quarter = n // 2, n // 4
    pot = 1
    for r in range(1, exp + 1):
        potm1, pot = pot, pot << 1
        if r < exp:
            j, k = tid >> (r - 1), tid & (potm1 - 1)
            wa = wb = _w(pot, k, inverse)          # r = 1: k = 0, W = 1 (the reference skips the multiply)
            oa = j * pot + k
            ob = oa + half
            oa2, ob2 = oa + potm1, ob + potm1
        else:                                      # last stage: two twiddles, written in place (RC:224-258)
            wa, wb = _w(n, tid, inverse), _w(n, tid + quarter, inverse)
            oa, oa2, ob, ob2 = tid, tid + half, tid + quarter, tid + 3 * quarter
        a, a2 = s[tid].copy(), s[tid + half].copy()
        b, b2 = s[tid + quarter].copy(), s[tid + 3 * quarter].copy()
        s[oa], s[oa2] = a + wa * a2, a - wa * a2
        s[ob], s[ob2] = b + wb * b2, b - wb * b2


def st_external(x, inverse=True):
    """FFT_GPU_external<P> of the Stockham C2C program (ST:243-258): one FFT per block."""
    x = np.asarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    for f in range(x.shape[0]):
        s = x[f].copy()
        stockham_block(s, x.shape[1], inverse)
        out[f] = s
    return out


def r2c_c2r_block(s, length, inverse):
    """do_FFT_Stockham_R2C_C2R<P,D> (RC:269-344) on s[0..L], L = complex length = real length / 2,
    L/4 threads; thread t handles the pairs (t + 1, L - t - 1) and (t + 1 + L/4, L - t - 1 - L/4)."""
    tid = np.arange(length // 4)
    ohx, ohy = (-0.5, 0.5) if inverse else (0.5, -0.5)
    if not inverse:
        stockham_block(s, length, False)                                   # RC:275
    else:
        z = s[0]
        s[0] = 0.5 * (z.real + z.imag) + 0.5j * (z.real - z.imag)        # RC:280-286
    for off in (0, length // 4):                                          # RC:289-309, 312-328
        ia, ib = tid + 1 + off, length - tid - 1 - off
        a, b = s[ia].copy(), s[ib].copy()
        h1 = 0.5 * (a.real + b.real) + 0.5j * (a.imag - b.imag)
        h2 = ohx * (a.imag + b.imag) + 1j * ohy * (a.real - b.real)
        wh = _w(2 * length, ia, inverse) * h2
        f1 = h1 + wh
        f2 = (h1.real - wh.real) + 1j * (-h1.imag + wh.imag)
        s[ia] = f1
        s[ib] = f2      # ia == ib == L/2 for the last thread of the second half: F2 is written last (RC:308)
    if not inverse:
        z = s[0]
        s[0] = (z.real + z.imag) + 1j * (z.real - z.imag)                 # RC:332-339
    else:
        stockham_block(s, length, True)                                    # RC:342


def rc_external(x, inverse):
    """FFT_GPU_R2C_C2R_external<P,D> (RC:349-365).  inverse = False: x is (nFFTs, N) real, read as N/2 float2
    (RC:406), result (nFFTs, N/2) complex packed; inverse = True: the reverse."""
    if not inverse:
        x = np.asarray(x, dtype=np.float64)
        nffts, n = x.shape
        length = n // 2
        out = np.empty((nffts, length), dtype=np.complex128)
        for f in range(nffts):
            s = np.zeros(length + 1, dtype=np.complex128)
            s[:length] = x[f, 0::2] + 1j * x[f, 1::2]
            r2c_c2r_block(s, length, False)
            out[f] = s[:length]
        return out
    x = np.asarray(x, dtype=np.complex128)
    nffts, length = x.shape
    out = np.empty((nffts, 2 * length), dtype=np.float64)
    for f in range(nffts):
        s = np.zeros(length + 1, dtype=np.complex128)
        s[:length] = x[f]
        r2c_c2r_block(s, length, True)
        out[f, 0::2] = s[:length].real
        out[f, 1::2] = s[:length].imag
    return out
