// smfft_device_functions.hpp -- the device-function surface under the reference's names.
//
// Two forms of every function:
//
// (1) THE REFERENCE'S CONTRACT (global namespace; what a kernel written for KAdamek/SMFFT calls):
//       do_SMFFT_CT_DIT<P>(s)              CT/FFT-GPU-32bit.cu:334-532, README.md:10-18
//       do_FFT_Stockham_mk6<P>(s)          ST/FFT-GPU-32bit-Stockham.cu:97-240
//       do_FFT_Stockham_C2C<P,D>(s)        RC/FFT-GPU-32bit-Stockham.cu:106-266
//       do_FFT_Stockham_R2C_C2R<P,D>(s)    RC/FFT-GPU-32bit-Stockham.cu:269-344
//     and the kernels in the reference's launch shape
//       SMFFT_DIT_external<P>(in, out), SMFFT_DIT_multiple<P>(in, out)        CT:534-572, <<<nFFTs*N/fft_length, fft_length/4>>>
//       FFT_GPU_external<P>(in, out), FFT_GPU_multiple<P>(in, out)            ST:243-278, <<<nFFTs, N/4, N*8>>>
//       FFT_GPU_R2C_C2R_external<P,D>(in, out), FFT_GPU_R2C_C2R_multiple<P,D> RC:349-384, <<<nFFTs, L/4>>>
//     Same contract as upstream: blockDim.x = fft_length / 4 (CT: 32 for N <= 128; Stockham: N / 4), the data in
//     s[0 .. fft_length) contiguous and in natural order before and after, in place, EVERY thread of the block calls,
//     the caller barriers before the call (CT also after it).  LDS the caller provides: CT P::fft_sm_required
//     (= 17 * fft_length / 16 here); Stockham exactly N float2, R2C/C2R L + 1 (as upstream: ST:319, RC:351).
//     How it runs on a 64-lane wave: EVERY thread of the block works on four elements, as upstream -- the reference's own
//     radix-2 decimation-in-time ladder with two stages fused per pass (quarter_fft_inplace below): in place, one
//     synchronisation per pass (a compiler fence while the N/4 threads share a wave), the bit reversal of the
//     natural-order variants folded into the first pass, and -- between the first reads and the last stores -- a swizzled
//     image of the same LDS words that takes the bank conflicts out of the strided passes (quarter_swizzle below).
//     Measured against this library's tiled / compact kernels on the same buffers (tools/reference_contract.py,
//     profiles/r03_quarter_swizzle.txt, r03_contract_registers.txt): a user's fill / call / drain kernel 0.65-0.94 of their
//     rate (round 2, one working wave per block: 0.14-0.55), the two-argument kernels below -- which for N >= 256 hand the
//     block's registers to do_SMFFT_CT_DIT_registers -- 0.72-0.97; in-LDS path 0.45-0.52 for N = 256 .. 2048 (0.24-0.39 for the half-wave blocks of N <= 128 and the
//     16-wave blocks of N = 4096) -- five LDS round trips of the whole FFT per transform (4 reads + 4 writes per thread
//     and pass) against two-and-a-half of the 16-elements-per-thread engine.  Kernels that want the
//     engine's speed use form (2) or the Engine directly (examples/fft_convolution.hip).
//
// (2) THE ENGINE'S TILED CONTRACT (namespace smfft::tiled; what this library's own kernels are built on):
//     256-thread workgroups own 4096 float2 = P::fft_per_block FFTs; `s` is an LDS array of P::tile_sm_required
//     (4352) float2; FFT j of the workgroup occupies s[j * P::fft_region + n], n in [0, N), natural order before
//     and after; all 256 threads call; callers barrier between filling s and the call and between the call and
//     reading s.
#pragma once
#include "smfft_engine.hpp"
#include "SM_FFT_stockham_parameters.hpp"

#ifndef NREUSES
#define NREUSES 100
#endif

namespace smfft {

// ------------------------------------------------------------------------------------------------
// R2C / C2R (real length 2L through a complex FFT of length L).  RC:269-344.
// Hermitian split (forward, after the C2C) / merge (inverse, before the C2C) on the natural
// layout in LDS; thread u of an FFT handles the 8 index pairs i = 1 + u + T*j, (i, L - i).
// ------------------------------------------------------------------------------------------------
template <int L, int DIR>
__device__ __forceinline__ void hermitian_pass(float2* sf, int u) {
    constexpr int T = L / 16;
    constexpr float ohx = DIR ? -0.5f : 0.5f;   // upstream's (ohx, ohy) = (1/2, -1/2) forward, (-1/2, 1/2) inverse (RC:289-328)
    if (DIR) {
        if (u == 0) {
            float2 z = sf[0];
            sf[0] = make_float2(0.5f * (z.x + z.y), 0.5f * (z.x - z.y));
        }
    }
    // H1 = S/2 with S = (A.x + B.x, A.y - B.y); H2 = (ohx * D.x, ohy * D.y) with D = (A.y + B.y, A.x - B.x) and ohy = -ohx,
    // so W * H2 = ((ohx W).x D.x + (ohx W).y D.y, (ohx W).y D.x - (ohx W).x D.y): with ohx folded into the twiddle (loop
    // invariant across the applications of the in-LDS kernels) a pair costs 12 instructions instead of 16
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 1 + u + T * j;
        const float2 A = sf[i], B = sf[L - i];
        const float2 W = twiddle<DIR>(i * (4096 / (2 * L)));
        const float2 Wh = make_float2(ohx * W.x, ohx * W.y);
        const float2 S = make_float2(A.x + B.x, A.y - B.y);
        const float2 D = make_float2(A.y + B.y, A.x - B.x);
        const float2 WH = make_float2(fmaf(Wh.x, D.x, Wh.y * D.y), fmaf(Wh.y, D.x, -Wh.x * D.y));
        sf[i] = make_float2(fmaf(0.5f, S.x, WH.x), fmaf(0.5f, S.y, WH.y));
        sf[L - i] = make_float2(fmaf(0.5f, S.x, -WH.x), fmaf(-0.5f, S.y, WH.y));   // for i == L/2 this value stays (RC:308)
    }
    if (!DIR) {
        if (u == 0) {   // sf[0] is not touched by the pair loop (i >= 1, L - i >= L/2)
            float2 z = sf[0];
            sf[0] = make_float2(z.x + z.y, z.x - z.y);
        }
    }
}

// In place on LDS, natural layout (device-function form; RC:269-344).
template <int L, int DIR>
__device__ __forceinline__ void r2c_c2r_lds_inplace(float2* s, const Engine<L, DIR, 1>& eng, int stride = Geometry<L>::SF) {
    using G = Geometry<L>;
    float2* sf = s + eng.fft * stride;
    if (DIR == 0) {
        fft_lds_inplace(s, eng, stride);
        fft_sync<G::kMultiWave>();
        hermitian_pass<L, 0>(sf, eng.u);
    } else {
        hermitian_pass<L, 1>(sf, eng.u);
        fft_sync<G::kMultiWave>();
        fft_lds_inplace(s, eng, stride);
    }
}

// ------------------------------------------------------------------------------------------------
// The reference's contract on ALL of its threads: blockDim.x = fft_length / 4, four elements per thread.
// (Round 2 ran the 16-elements-per-thread engine on the block's first wave and let the other waves wait at the
// caller's barrier: 0.14-0.55 of the tiled kernels' rate, profiles/r03_reference_contract_before.txt.)
//
// Plan: the reference's own radix-2 decimation-in-time ladder (CT:334-532), two stages fused per pass so that a
// thread's four elements {a, a+P, a+2P, a+3P} make one radix-2^2 butterfly IN PLACE -- no exchange inside a pass, one
// synchronisation between passes (nothing but a compiler fence while the FFT's N/4 threads share a wave, N <= 256):
//   pass p (P = 4^p):  k = t mod P, a = (t / P) * 4P + k
//        (e0, e1) <- (e0 + w1 e1, e0 - w1 e1), (e2, e3) likewise      w1 = W_2P^k  = (W_4P^k)^2
//        (e0, e2) <- (e0 + w2 e2, e0 - w2 e2)                          w2 = W_4P^k
//        (e1, e3) <- (e1 + w3 e3, e1 - w3 e3)                          w3 = W_4P^(k+P) = -+i w2
//   log2 N odd: a last radix-2 pass on (t, t + N/2) and (t + N/4, t + 3N/4) with W_N^t and -+i W_N^t (CT:493-531).
// Applied to natural-order data that ladder computes DFT(x o bitrev) -- the no-reorder transform S2 -- as it stands.
// Natural order (S1, Stockham S3/S4): the bit reversal is folded into pass 0, whose twiddles are trivial: thread t loads
// x[t + m N/4] (m = 0..3), which are the inputs bitrev(4j + i), i = rev2(m), of butterfly j = rev(t), and stores its
// four results at 4j + i: one scattered store instead of the reference's three-barrier reorder (CT:126-329).
// One twiddle load per thread and pass, from per-length rows (consecutive threads -> consecutive values).
// ------------------------------------------------------------------------------------------------
template <int N>
struct QuarterTwiddleRows {
    // row of pass p >= 1 (P = 4^p): W_4P^k, k < P; then, for odd log2 N, the row of the last radix-2 pass: W_N^t, t < N/4
    static constexpr int kBits = ilog2c(N);
    static constexpr int kPasses = kBits / 2;                 // fused radix-2^2 passes (pass 0 has no twiddles)
    static constexpr bool kOdd = (kBits & 1) != 0;
    static constexpr int row_start(int p) {                   // p = 1 .. kPasses (kPasses = the radix-2 row)
        int o = 0;
        for (int q = 1, P = 4; q < p; ++q, P *= 4) o += P;
        return o;
    }
    static constexpr int kCount = row_start(kPasses) + (kOdd ? N / 4 : 0);
    TwiddleValue w[kCount > 0 ? kCount : 1];
    constexpr QuarterTwiddleRows() : w{} {
        int P = 4;
        for (int p = 1; p < kPasses; ++p, P *= 4)
            for (int k = 0; k < P; ++k) w[row_start(p) + k] = twiddle_values[(k * (4096 / (4 * P))) & 4095];
        if (kOdd)
            for (int t = 0; t < N / 4; ++t) w[row_start(kPasses) + t] = twiddle_values[(t * (4096 / N)) & 4095];
    }
};
template <int N>
static __device__ const QuarterTwiddleRows<N> quarter_twiddle_rows = QuarterTwiddleRows<N>();

// Every twiddle of a thread's transform, fetched FIRST -- in front of the first synchronisation -- and kept: the index of a pass's
// twiddle depends on the thread only, so in a caller's loop around the device function (the `multiple` kernels: the reference's
// own benchmark shape, CT:553-572) the loads and the two products below are loop invariant, and at the top of the function the
// compiler may hoist them (behind a barrier or an asm statement it may not: a load is not speculated).  A fused radix-2^2 pass
// then costs three complex products and eight complex sums, b1 = w1 x1, b2 = w2 x2, b3 = w3 x3:
//     (x0 + b1) +- (b2 + b3),   (x0 - b1) +- (-+i)(b2 - b3)
// -- 28 instructions instead of the 36 of two radix-2 stages with w1 = w2^2 recomputed (N = 1024: 208 -> 176 vector instructions
// per thread and application, and no global load left in the loop).  A single call pays the same 36 as the two-stage form.
// The kept values cost registers -- six per pass -- and the blocks of N = 2048 / 4096 are 8 / 16 waves that must fit a CU several
// times over: with everything kept those kernels need 66 / 73 registers (measured: N = 4096 one block per CU instead of two,
// -20 %), so their last passes (quarter_late_passes) fetch at the pass as before ("late" passes: the load stays behind the pass's
// synchronisation, nothing of it is kept).  N <= 128 does the same in every pass: in upstream's 32-thread blocks the kept form
// measured 2-3 % slower, in the 64-thread blocks of the _wave64 classes within +-2 % (profiles/r05_contract_twiddles.txt).
struct QuadTwiddle { float2 w1, w2, w3; };      // W_2P^k, W_4P^k, W_4P^3k
// (measured, in-LDS loop of 100 calls at the README batch against fetching in every pass: N = 256 ... 1024 +7 ... +13 %, N = 2048 +9 %
//  natural order / +4 % no reorder, N = 4096 no reorder +3 %; N = 4096 natural order -2 % with one pass kept: all late there)
constexpr int quarter_late_passes(int n, bool reorder) {
    const int passes = ilog2c(n) / 2;
    return n >= 4096 ? (reorder ? passes : 4) : n >= 2048 ? 1 : n <= 128 ? passes : 0;
}
// KEEP_ALL (round 6): no late pass whatever the length.  The phased form of N = 2048 / 4096 (quarter_fft, kLastPhase) needs 56 / 64
// registers with every twiddle kept -- still eight waves per SIMD -- and a late pass's global load sits right behind the one barrier
// of its last phase, on the critical path of sixteen waves: kept, N = 4096 runs 13 ... 15 % faster in the in-LDS loop, N = 2048 1 ... 4 %
// (profiles/r06_contract_phases.txt).  The round-5 form (OUT_REGS at these lengths) keeps its late passes.
// HIGH_FROM: the first pass whose butterfly index comes from kb_high (4: the last phase of N >= 2048; 3: the natural-order transforms of
// N = 512 / 1024 in PAIRS of passes -- quarter_fft -- whose threads are numbered anew at the second image).
template <int N, int DIR, int REORDER, bool KEEP_ALL = false, int HIGH_FROM = 4>
struct QuarterTwiddles {
    using R = QuarterTwiddleRows<N>;
    static constexpr int kLatePasses = KEEP_ALL ? 0 : quarter_late_passes(N, REORDER != 0);
    static constexpr bool kRowSelects = N < 1024;      // how the wave-local ladder of THIS length swaps lane bits 2 / 3 (slots_swap; re-measured on the phased engine in round 6: selects at N >= 1024 within +-0.3 %, N = 4096 natural order 5 % slower)
    static constexpr bool late(int p) { return p >= R::kPasses - kLatePasses; }
    QuadTwiddle q[R::kPasses > 1 ? R::kPasses : 1];    // q[p], p = 1 .. kPasses - 1 (P = 4^p)
    float2 wr;                                         // the radix-2 pass of an odd log2 N
    int kbase, kbase_high, kbase_first;                // butterfly index of the passes with P <= 64 / of the passes above (the last phase of N >= 2048 numbers its threads differently) / of pass 1 where that differs
    __device__ static __forceinline__ QuadTwiddle fetch(int p, int k) {
        const TwiddleValue tv = quarter_twiddle_rows<N>.w[R::row_start(p) + k];
        QuadTwiddle t;
        t.w2 = make_float2(tv.x, DIR ? -tv.y : tv.y);
        t.w1 = make_float2(t.w2.x * t.w2.x - t.w2.y * t.w2.y, 2.f * t.w2.x * t.w2.y);
        t.w3 = late(p) ? t.w2 : cmul(t.w1, t.w2);      // (a late pass takes the two-stage butterfly: no third product)
        return t;
    }
    // kb: the thread's butterfly index (k = kb mod P in every pass); r: its index in the radix-2 row; kb_high: the index for the passes with
    // P >= 256 where it differs (quarter_fft's last phase)
    __device__ __forceinline__ void load(int kb, int r) { load(kb, r, kb); }
    __device__ __forceinline__ void load(int kb, int r, int kb_high) { load(kb, r, kb_high, kb); }
    __device__ __forceinline__ void load(int kb, int r, int kb_high, int kb_first) {
        kbase = kb;
        kbase_high = kb_high;
        kbase_first = kb_first;
#pragma unroll
        for (int p = 1; p < R::kPasses; ++p)
            if (!late(p)) q[p] = fetch(p, (p == 1 ? kb_first : p >= HIGH_FROM ? kb_high : kb) & ((1 << (2 * p)) - 1));
        if constexpr (R::kOdd) {
            const TwiddleValue tv = quarter_twiddle_rows<N>.w[R::row_start(R::kPasses) + r];
            wr = make_float2(tv.x, DIR ? -tv.y : tv.y);
        }
    }
    // the twiddles of pass `pass`, at the pass (pass: a constant where this is called -- a template argument or the counter of an unrolled loop)
    __device__ __forceinline__ QuadTwiddle of(int pass) const {
        return late(pass) ? fetch(pass, (pass == 1 ? kbase_first : pass >= HIGH_FROM ? kbase_high : kbase) & ((1 << (2 * pass)) - 1)) : q[pass];
    }
};
// the fused radix-2^2 butterfly on (x0, x1, x2, x3) = elements k, k + P, k + 2P, k + 3P; results in place
// (two_stage: the form for a twiddle fetched at the pass -- w2 (x2 +- w1 x3): four products, none of them behind w3 = w1 w2, whose
//  place in the dependency chain load -> w1 -> w3 -> b3 cost the 16-wave blocks of N = 4096 2 %)
template <int DIR>
__device__ __forceinline__ void quad_butterfly(float2& x0, float2& x1, float2& x2, float2& x3, const QuadTwiddle& w, bool two_stage) {
    const float2 b1 = cmul(x1, w.w1);
    const float2 y0 = cadd(x0, b1), y1 = csub(x0, b1);
    float2 y2, v3;
    if (two_stage) {
        const float2 t3 = cmul(x3, w.w1);
        y2 = cmul(cadd(x2, t3), w.w2), v3 = cmul(csub(x2, t3), w.w2);
    } else {
        const float2 b2 = cmul(x2, w.w2), b3 = cmul(x3, w.w3);
        y2 = cadd(b2, b3), v3 = csub(b2, b3);
    }
    const float2 u3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);   // (-+i) v3
    x0 = cadd(y0, y2), x1 = cadd(y1, u3), x2 = csub(y0, y2), x3 = csub(y1, u3);
}

// Inside the function the data lives in a SWIZZLED image of the same LDS words: element i (index within the block's region:
// the FFT's offset included, N <= 128 keeps 128 / N transforms per 32 threads) sits at
//     i ^ ((i >> 8) & 31) ^ ((i >> 4) & 30) ^ ((i >> 2) & 24)
// -- the five bank-selecting bits of a float2 index get the higher index bits mixed in (x = i >> 5: (x >> 3) ^ (x << 1) ^
// (x << 3), GF(2)-linear, so a pass's four addresses are one swizzled base XOR three constants).  In the natural layout a
// pass's stride-4P accesses hit 8 (P = 4) or 16 (P = 16) of the 32 float2 banks and the natural-order variants' scattered
// store of pass 0 two of them; with the swizzle every read of every pass is conflict free and the stores at most 2-way
// (tools/quarter_swizzle.py: LDS cycles per N = 1024 transform, both orderings together, 2912 -> 1344 of 1248 conflict free;
// measured SQ_LDS_BANK_CONFLICT 0.44-0.57 -> 0.10-0.25 of the LDS cycles, in-LDS rate x 1.5-2.0).  The
// price: the first pass must have read everything before anything is stored, and the last pass reads everything before it
// stores in natural order -- one more synchronisation at either end (the natural-order first pass had one already).
// One float2 from LDS as a ds_read_b64 of its own: hipcc merges neighbouring constant-stride loads into ds_read2(st64)_b64, which the LDS serves
// at HALF the rate of single ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 cycles for two float2 against 2 + 2) -- `volatile` 64-bit loads are
// neither merged nor reordered, and (unlike inline assembly) counted by the compiler's s_waitcnt.
// (used for the four natural-order loads of quarter_fft's pass 0, N = 256 ... 2048: in-LDS loop +1.7 ... 2.7 % at N = 256 / 512 / 2048, nothing at 1024;
//  N = 4096 natural order measured 6 % slower that way, the last phase's reads 1 %: they keep the merged form; profiles/r06_contract_phases.txt (5))
__device__ __forceinline__ float2 lds_load_single(const float2* p) {
    typedef __attribute__((address_space(3))) const float2 lds_float2;
    typedef __attribute__((address_space(3))) const volatile unsigned long long lds_u64;
    const unsigned long long w = *(lds_u64*)(lds_float2*)p;
    return make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
}
__host__ __device__ constexpr int quarter_swizzle(int i) { return i ^ ((i >> 8) & 31) ^ ((i >> 4) & 30) ^ ((i >> 2) & 24); }
// The image of the ONE-wave natural-order transform (N = 256, round 6: quarter_fft's phases): pass 0's results go through LDS once --
// scattered stores of positions 4 rev(t) + i, read back with slots = position bits (2, 3) -- and for eight position bits no XOR of shifts
// into the five low bits serves both (sixteen contiguous lanes of a store differ in position bits 7 ... 4, which must reach the four bank
// bits of a ds_write_b64; the 32 lanes of a read differ in bits 0, 1, 4, 5, 6).  Address bits 0 ... 3 take position bits 7, 6, 5, 4 in,
// bit 4 takes bit 6 in: GF(2)-linear, a bijection of 0 ... 255, the identity on 0 ... 15 (so a thread's four addresses are again one
// base XOR constants); tools/quarter_phases_model.py searched the family and counts both accesses conflict free.
__host__ __device__ constexpr int quarter_image256(int p) { return p ^ ((p >> 7) & 1) ^ ((p >> 5) & 2) ^ ((p >> 3) & 4) ^ ((p >> 1) & 8) ^ ((p >> 2) & 16); }

// ------------------------------------------------------------------------------------------------
// N <= 256 (round 4): the same radix-2^2 ladder with NO LDS between its passes.  The N/4 threads of a transform are 8 ... 64
// lanes of one wave, so the exchange between two passes -- the four elements a thread holds swap their two slot bits with the
// two lane bits that hold the next two index bits -- is a pair of one-bit lane <-> register transposes: v_permlane32_swap /
// v_permlane16_swap for lane bits 5 / 4 (one instruction per dword pair), two row-DPP moves per dword pair for bits 3 / 2, two
// DPP-fed selects per dword pair for bits 1 / 0 (the primitives of smfft_engine.hpp).  N = 256: 40 such instructions per
// thread replace three LDS round trips of the whole transform (12 ds_write_b64 + 12 ds_read_b64 per thread); the contract's
// first read and last write of s[] remain.  Index bookkeeping (checked by a NumPy model of the lanes and slots against numpy.fft,
// tools/quarter_lanes_model.py, and by the GPU parity tests of every contract kernel):
//   in:   natural order: thread t loads x[t + m N/4] into slot rev2(m) -> it holds y[4 rev(t) + slot] of the bit-reversed input y,
//         i.e. index bits 0, 1 in the slots and index bit 2 + b in lane bit T_BITS - 1 - b; no reorder: x[4 t + slot], bit 2 + b in lane bit b
//   pass p >= 1 (P = 4^p): slot bits 0 / 1 (index bits 2p - 2, 2p - 1) <-> the lane bits that hold index bits 2p, 2p + 1 (lane_bit_of
//         below), then the butterfly of quarter_fft with k = base mod P, base = rev(t) (natural order) or t (no reorder)
//   odd log2 N: slot bit 0 <-> the last lane bit, two radix-2 butterflies with W_N^base and -+i W_N^base
//   out:  slot i holds result element base + (N/4) * sigma(i), sigma = identity (even log2 N) or (0, 2, 1, 3) (odd)
// Measured (profiles/r04_contract_lanes.txt; in-LDS loop of 100 calls, README batch, against the LDS form): with the quad
// transposes as two DPP-fed selects per dword pair (smfft_engine.hpp, swap_bit_quad) N = 32 +83...+99 %, N = 64 +21...+26 %, N = 128
// +30...+32 %, N = 256 +2...+9 % (with hipcc's own v_mov_b32_dpp + v_cndmask pairs: +40...60 / -7...+4 / +9...13 / -1...-3.5 %),
// so do_SMFFT_CT_DIT uses the lane form for every N <= 256.  Around an HBM-bound fill / call / drain the two forms are within a
// few percent of each other (the lane form ahead in upstream's 32-thread blocks, behind in 64-thread blocks), which is why the
// two-argument external kernels below choose per shape.
#ifndef SMFFT_QUARTER_LANES
#define SMFFT_QUARTER_LANES 1          // 0: every length through LDS (A/B)
#endif
constexpr bool quarter_lanes_default(int n) { return SMFFT_QUARTER_LANES != 0 && n <= 256; }
// slot bit SLOT_BIT of the thread's four elements <-> lane bit LANE_BIT (ROW_SELECTS: lane bits 2 / 3 through the selects of
// swap_bit_select instead of the bank-masked moves of swap_bit_dpp_dword -- faster except in the blocks of N >= 1024)
template <int LANE_BIT, bool ROW_SELECTS>
__device__ __forceinline__ void lane_slot_swap(float2& A, float2& B) {
    using X = Engine<1024, 0, 1>;                // (the transposes are static members; the length is immaterial)
    if constexpr (LANE_BIT >= 4) {
        X::template swap_bit<LANE_BIT>(A, B);
    } else if constexpr (LANE_BIT <= 1 || ROW_SELECTS) {
        X::template swap_bit_select<LANE_BIT>(A, B);
    } else {
        X::template swap_bit_dpp_dword<LANE_BIT>(A.x, B.x);
        X::template swap_bit_dpp_dword<LANE_BIT>(A.y, B.y);
    }
}
template <int SLOT_BIT, int LANE_BIT, bool ROW_SELECTS>
__device__ __forceinline__ void slots_swap(float2 (&e)[4]) {
    if constexpr (SLOT_BIT == 0) {
        lane_slot_swap<LANE_BIT, ROW_SELECTS>(e[0], e[1]);
        lane_slot_swap<LANE_BIT, ROW_SELECTS>(e[2], e[3]);
    } else {
        lane_slot_swap<LANE_BIT, ROW_SELECTS>(e[0], e[2]);
        lane_slot_swap<LANE_BIT, ROW_SELECTS>(e[1], e[3]);
    }
}
template <int N, int DIR, int REORDER>
struct QuarterLanes {
    using R = QuarterTwiddleRows<N>;
    static constexpr int Q = N / 4, T_BITS = ilog2c(Q), N_BITS = ilog2c(N);
    static_assert(N >= 32 && N <= 256, "the transform's N / 4 threads must be lanes of one wave");
    // the lane bit that holds index bit 2 + b at the start
    static constexpr int lane_bit_of(int b) { return REORDER ? T_BITS - 1 - b : b; }
    // result element of slot i, relative to base: (N/4) * sigma(i)
    static constexpr int out_offset(int i) { return Q * (R::kOdd ? (((i & 1) << 1) | (i >> 1)) : i); }
    __device__ static __forceinline__ int base_of(int t) { return REORDER ? (int)(__brev((unsigned)t) >> (32 - T_BITS)) : t; }

    // (TW: the QuarterTwiddles of this length or of a longer one -- the rows W_4P^k depend on P only)
    template <int P_INDEX, class TW>
    __device__ static __forceinline__ void passes(float2 (&e)[4], const TW& tw) {
        if constexpr (P_INDEX < R::kPasses) {
            slots_swap<0, lane_bit_of(2 * P_INDEX - 2), TW::kRowSelects>(e);
            slots_swap<1, lane_bit_of(2 * P_INDEX - 1), TW::kRowSelects>(e);
            // slots: e[1] = element k + 2P, e[2] = element k + P of the two-stage form (t1 = w1 e[1], u2 = w2 (e[2] + w1 e[3]))
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(P_INDEX), TW::late(P_INDEX));
            passes<P_INDEX + 1>(e, tw);
        }
    }
    // The ladder from pass 1 on for a thread whose slots ALREADY hold index bits (2, 3) -- the natural-order transforms of N >= 256 read
    // pass 0's scattered results back that way (quarter_fft, round 6): lane bits 0, 1 = index bits 0, 1, lane bits 2 ... 5 = index bits
    // 4 ... 7, so pass 1 starts at once and the exchanges of the passes 2, 3 are the ones passes<2> makes anyway (with lane bits 2, 3 and
    // 4, 5; REORDER = 0 numbering): sixteen DPP-fed selects per thread and transform fewer than reading four neighbours and transposing.
    // k = lane mod P in every pass, and the thread ends with elements lane + 64 i as after run().
    // (The remaining exchange of lane bits 2, 3 through the wave's block of LDS as well -- run_exchanged's way -- measured 10 ... 29 % SLOWER
    //  here: the natural-order transform already makes three trips through LDS, a fourth makes the LDS unit the bound;
    //  profiles/r06_contract_phases.txt.)
    template <class TW>
    __device__ static __forceinline__ void passes_from_slots23(float2 (&e)[4], const TW& tw) {
        static_assert(N == 256 && !REORDER, "the wave-local ladder of an aligned block of 256 elements");
        quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), TW::late(1));
        passes<2>(e, tw);
    }
    // run() with the exchange between the passes 1 and 2 THROUGH LDS instead of through the row's DPP network (round 6; no reorder, the
    // wave's aligned block of 256 elements as scratch).  A transpose of the lane bits 2, 3 is sixteen DPP-fed selects per thread --
    // about 120 vector cycles of a wave -- and the vector unit is what bounds the no-reorder ladder (N = 1024: 600 vector cycles per
    // wave and transform against 224 LDS cycles per BLOCK); four ds_write_b64 + four ds_read_b64 of the wave are 32 cycles of an LDS
    // unit that is mostly idle here, no barrier (one wave), and the re-read deals the lanes afresh: slots = index bits (4, 5), lane
    // bits 0 ... 3 = index bits 0 ... 3, lane bits 4, 5 = index bits 6, 7 -- so the last exchange is still the cheap one
    // (v_permlane16/32_swap) and the thread ends with elements lane + 64 i as after run().  Image of the block: element p at
    // p ^ ((p >> 2) & 28) -- sixteen contiguous lanes of a store differ in index bits 0, 1, 4, 5, the 32 lanes of a read in bits
    // 0 ... 3 and 6: both conflict free (tools/quarter_phases_model.py).  k = lane mod P in every pass, as in run().
    __host__ __device__ static constexpr int exchange_image(int p) { return p ^ ((p >> 2) & 28); }
    template <class TW>
    __device__ static __forceinline__ void run_exchanged(float2 (&e)[4], const TW& tw, float2* block, int lane, bool loads_precede) {
        static_assert(N == 256 && !REORDER, "the wave-local ladder of an aligned block of 256 elements");
        {   // pass 0: twiddles 1, 1, -+i
            const float2 s0 = cadd(e[0], e[1]), d0 = csub(e[0], e[1]), s1 = cadd(e[2], e[3]), d1 = csub(e[2], e[3]);
            const float2 jd1 = DIR ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
            e[0] = cadd(s0, s1), e[1] = cadd(d0, jd1), e[2] = csub(s0, s1), e[3] = csub(d0, jd1);
        }
        slots_swap<0, 0, TW::kRowSelects>(e);
        slots_swap<1, 1, TW::kRowSelects>(e);
        quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), TW::late(1));
        if (loads_precede) fft_sync<false>();          // the wave's loads of the block precede its stores
        // slots = index bits (2, 3); lane bits 0, 1 = bits 0, 1; lane bits 2 ... 5 = bits 4 ... 7
        const int p0 = exchange_image((lane & 3) + 16 * (lane >> 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) block[p0 ^ (4 * j)] = e[j];           // (the image is linear and leaves 4 j alone)
        fft_sync<false>();
        const int q0 = exchange_image((lane & 15) + 64 * (lane >> 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = block[q0 ^ exchange_image(16 * j)];
        quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(2), TW::late(2));
        passes<3>(e, tw);
    }
    // the thread's twiddles: k = base mod P in every pass
    __device__ static __forceinline__ QuarterTwiddles<N, DIR, REORDER> twiddles_of(int t) {
        QuarterTwiddles<N, DIR, REORDER> tw;
        tw.load(base_of(t), base_of(t));
        return tw;
    }
    // e[slot]: natural order e[rev2(m)] = x[t + m N/4], no reorder e[i] = x[4 t + i]; on return e[i] = result element base + out_offset(i)
    template <class TW>
    __device__ static __forceinline__ void run(float2 (&e)[4], const TW& tw) {
        {   // pass 0: twiddles 1, 1, -+i
            const float2 s0 = cadd(e[0], e[1]), d0 = csub(e[0], e[1]), s1 = cadd(e[2], e[3]), d1 = csub(e[2], e[3]);
            const float2 jd1 = DIR ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
            e[0] = cadd(s0, s1), e[1] = cadd(d0, jd1), e[2] = csub(s0, s1), e[3] = csub(d0, jd1);
        }
        passes<1>(e, tw);
        if constexpr (R::kOdd) {
            slots_swap<0, lane_bit_of(N_BITS - 3), TW::kRowSelects>(e);
            const float2 w = tw.wr;
            const float2 t1 = cmul(e[1], w), v3 = cmul(e[3], w);
            const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
            const float2 x0 = e[0], x2 = e[2];
            e[0] = cadd(x0, t1), e[1] = csub(x0, t1), e[2] = cadd(x2, t3), e[3] = csub(x2, t3);
        }
    }
    // the contract's form: data in s[region_offset .. + N) natural order in and out (IN_REGS: the inputs come in x[] instead)
    template <bool IN_REGS>
    __device__ static __forceinline__ void lds_to_lds(float2 (&x)[4], float2* s, int t, int region_offset) {
        const QuarterTwiddles<N, DIR, REORDER> tw = twiddles_of(t);
        float2* sf = s + region_offset;
        float2 e[4];
        if constexpr (REORDER) {
#pragma unroll
            for (int m = 0; m < 4; ++m) e[((m & 1) << 1) | (m >> 1)] = IN_REGS ? x[m] : sf[t + m * Q];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = IN_REGS ? x[i] : sf[4 * t + i];
        }
        run(e, tw);
        if constexpr (!IN_REGS) fft_sync<false>();                  // every load of the transform's lanes precedes the stores (one wave)
        const int base = base_of(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) sf[base + out_offset(i)] = e[i];
    }
};

// ------------------------------------------------------------------------------------------------
// N = 64, 128 (N = 32 is written out as well, and loses: below), natural order (round 6): the road N = 256 takes (quarter_fft's phases) for the lengths of which a block holds SEVERAL
// transforms -- upstream's 32 threads = 128 / N of them (CT:586-595), the _wave64 classes' 64 threads = 256 / N; all in one wave.
// Pass 0 on x[t + m N/4], ONE trip through the block's image -- results 4 rev(t) + i scattered, read back with slots = index bits (2, 3),
// lane bits 0, 1 = bits 0, 1 of the thread's index in its transform, the bits above = index bits 4 ... -- then pass 1 at once, the exchange
// with lane bits 2, 3 (selects) and pass 2 (N >= 64), and the radix-2 pass of N = 128 (slot bit 0 <-> lane bit 4: v_permlane16_swap) / N = 32
// (slot bit 0 <-> lane bit 2).  Against QuarterLanes::lds_to_lds (every exchange on lanes): sixteen DPP-fed selects per thread and
// transform fewer (N = 128: 32 + 4 swaps -> 16 + 4; N = 64: 32 -> 16; N = 32: 24 -> 8) for four ds_write_b64 + four ds_read_b64 on an
// LDS unit these kernels leave idle, and a thread ends with elements u + (N/4) i: a lane-linear natural store where the ladder stored
// bit-reversed lanes.  Images (tools/quarter_phases_model.py searched them; both accesses conflict free for 32- and 64-thread blocks):
// address bit b of the BLOCK-level index q = f N + p takes the parity of q & mask[b] in -- GF(2)-linear, triangular (a bijection of every
// aligned group of 32 elements), the identity on 0 ... 3.
template <int N>
__host__ __device__ constexpr int quarter_small_image(int q) {
    constexpr int m0 = N == 32 ? 144 : 16, m1 = N == 128 ? 64 : 4, m2 = N == 128 ? 96 : N == 64 ? 72 : 32, m3 = N == 32 ? 64 : 32;
    return q ^ (__builtin_popcount(q & m0) & 1) ^ ((__builtin_popcount(q & m1) & 1) << 1) ^ ((__builtin_popcount(q & m2) & 1) << 2) ^ ((__builtin_popcount(q & m3) & 1) << 3);
}
template <int N, int DIR>
__device__ __forceinline__ void quarter_small_natural(float2* s, int t, int region_offset) {
    static_assert(N == 32 || N == 64 || N == 128, "N = 256 and above: quarter_fft");
    constexpr int Q = N / 4, T_BITS = ilog2c(Q);
    QuarterTwiddles<N, DIR, 1> tw;
    tw.load(t, t);                                           // k = t mod P in every pass: lane bits 0 ... hold index bits 0 ... by the time a pass needs them
    float2* sf = s + region_offset;
    float2 e[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) e[((m & 1) << 1) | (m >> 1)] = sf[t + m * Q];
    const int a = 4 * (int)(__brev((unsigned)t) >> (32 - T_BITS));
    fft_sync<false>();                                       // the wave's loads precede its scattered stores (every transform of the block is in this wave)
    {
        const float2 s0 = cadd(e[0], e[1]), d0 = csub(e[0], e[1]), s1 = cadd(e[2], e[3]), d1 = csub(e[2], e[3]);
        const float2 jd1 = DIR ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
        const int a0 = quarter_small_image<N>(region_offset + a);
        s[a0] = cadd(s0, s1), s[a0 ^ 2] = csub(s0, s1), s[a0 ^ 1] = cadd(d0, jd1), s[a0 ^ 3] = csub(d0, jd1);
    }
    fft_sync<false>();
    const int p1 = N == 128 ? ((t & 3) | (((t >> 2) & 3) << 4) | ((t >> 4) << 6)) : ((t & 3) | ((t >> 2) << 4));
    const int b0 = quarter_small_image<N>(region_offset + p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = s[b0 ^ quarter_small_image<N>(4 * j)];
    quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), tw.late(1));
    if constexpr (N >= 64) {
        slots_swap<0, 2, true>(e);
        slots_swap<1, 3, true>(e);
        quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(2), tw.late(2));
    }
    fft_sync<false>();                                       // the wave's loads of the image precede its natural stores
    if constexpr (N == 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sf[t + 16 * i] = e[i];
    } else {
        slots_swap<0, N == 128 ? 4 : 2, true>(e);            // slots = (the top index bit, the one below): elements t, t + N/2, t + N/4, t + 3N/4
        const float2 w = tw.wr;
        const float2 t1 = cmul(e[1], w), v3 = cmul(e[3], w);
        const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
        sf[t] = cadd(e[0], t1);
        sf[t + N / 2] = csub(e[0], t1);
        sf[t + Q] = cadd(e[2], t3);
        sf[t + 3 * Q] = csub(e[2], t3);
    }
}

// ... and without reorder (N = 64, 128): the ladder of QuarterLanes with its MIDDLE exchange -- lane bits 2, 3, sixteen DPP-fed selects --
// through the block's image instead (the block-level form of QuarterLanes<256>::run_exchanged: same image q ^ ((q >> 2) & 28), which also
// keeps the two transforms that share a 32-lane read group of N = 64 apart).  The re-read deals the lanes afresh: slots = index bits (4, 5),
// lane bits 0 ... 3 = bits 0 ... 3, lane bit 4 = bit 6 (N = 128: its radix-2 pass behind a v_permlane16_swap).
// (N = 32, either ordering: the trip through LDS measured 21 % SLOWER than its three exchanges on lanes -- eight lanes per transform leave
//  the natural accesses 4-way conflicted and the vector work is small; it keeps QuarterLanes.  profiles/r06_contract_small.txt)
template <int N, int DIR>
__device__ __forceinline__ void quarter_small_noreorder(float2* s, int t, int region_offset) {
    static_assert(N == 64 || N == 128, "N = 256 and above: quarter_fft; N = 32: QuarterLanes");
    constexpr int Q = N / 4;
    QuarterTwiddles<N, DIR, 0> tw;
    tw.load(t, t);
    float2* sf = s + region_offset;
    float2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = sf[4 * t + i];
    {   // pass 0: twiddles 1, 1, -+i
        const float2 s0 = cadd(e[0], e[1]), d0 = csub(e[0], e[1]), s1 = cadd(e[2], e[3]), d1 = csub(e[2], e[3]);
        const float2 jd1 = DIR ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
        e[0] = cadd(s0, s1), e[1] = cadd(d0, jd1), e[2] = csub(s0, s1), e[3] = csub(d0, jd1);
    }
    slots_swap<0, 0, true>(e);
    slots_swap<1, 1, true>(e);
    quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), tw.late(1));
    fft_sync<false>();                                       // the wave's loads precede its stores into the block
    // slots = index bits (2, 3); lane bits 0, 1 = bits 0, 1; the lane bits above = bits 4 ...
    const int p0 = QuarterLanes<256, DIR, 0>::exchange_image(region_offset + (t & 3) + 16 * (t >> 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) s[p0 ^ (4 * j)] = e[j];
    fft_sync<false>();
    const int q0 = QuarterLanes<256, DIR, 0>::exchange_image(region_offset + (t & 15) + 64 * (t >> 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = s[q0 ^ QuarterLanes<256, DIR, 0>::exchange_image(16 * j)];
    quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(2), tw.late(2));
    fft_sync<false>();
    if constexpr (N == 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sf[t + 16 * i] = e[i];
    } else {
        slots_swap<0, 4, true>(e);                           // slots = (bit 6, bit 5): elements t, t + 64, t + 32, t + 96
        const float2 w = tw.wr;
        const float2 t1 = cmul(e[1], w), v3 = cmul(e[3], w);
        const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
        sf[t] = cadd(e[0], t1);
        sf[t + N / 2] = csub(e[0], t1);
        sf[t + Q] = cadd(e[2], t3);
        sf[t + 3 * Q] = csub(e[2], t3);
    }
}

// BLOCK_THREADS: threads the caller's block has (what decides between a wave-level fence and a workgroup barrier);
// s: the block's LDS region, region_offset: where this thread's transform starts in it (f * N).
// IN_REGS: the first pass takes its four inputs from x[] instead of loading them from s -- x[m] = element t + m N/4 (natural
// order) or 4 t + m (no reorder) of the transform: exactly what the pass would load -- and s only has to be FREE at the
// call; OUT_REGS: the last pass leaves its four results, elements t + m N/4, in x[] instead of storing them.  Both:
// the transform of a thread block's registers with s as scratch -- one LDS round trip and one synchronisation less at
// either end (the kernels in the reference's launch shape below use it for N >= 256).
// ENGINE: 0 = the default of the length (quarter_lanes_default), 1 = the LDS form, 2 = the lane form (N <= 256)
//
// PHASES (round 6; N >= 256, tools/quarter_phases_model.py replays them against numpy.fft and counts their LDS cycles).  A wave holds
// 256 elements: six index bits in its lanes, two in the four slots of a thread, and runs every pass over the bits it holds on lanes
// and registers (QuarterLanes).  The transform is cut where the owners of the bits change, and only there does it go through LDS:
//   natural order  phase 0  thread t loads x[t + m N/4] and runs pass 0, which carries the bit reversal: results = positions 4 rev(t) + i,
//                           scattered into the swizzled image (N = 256: quarter_image256);
//                  phase 1  read back with SLOTS = POSITION BITS (2, 3) (lane bits 0, 1 = bits 0, 1; lane bits 2 ... 5 = bits 4 ... 7):
//                           pass 1 starts at once -- round 5 read four neighbours and transposed twice first (sixteen DPP-fed selects) --
//                           passes 1 ... 3 on the wave's aligned block of 256 positions; N = 256 ends here: slots = bits (6, 7),
//                           lane = bits 0 ... 5, a conflict-free natural store (round 5's all-register form of this length stored
//                           bit-reversed lanes: 54 % of its LDS cycles were bank conflicts);
//   no reorder     phase 1  thread t loads x[4 t + i], passes 0 ... 3 on the wave's aligned block (as in round 4);
//   N = 512 / 1024 the last pass (radix 2 on bit 8 / radix 4 on bits 8, 9) in the thread, from the swizzled image, natural store;
//   N = 2048 / 4096 LAST PHASE: the block of 256 goes to the image in NATURAL layout; behind ONE barrier the wave owns position bits
//                  8 ... n-1 and the low 16 - n bits (slots = bits 8, 9; the other lane bits = bits 10, 11): pass 4, slots <-> lane bits 4, 5
//                  (v_permlane16/32_swap) and pass 5 (N = 2048: slot bit 0 <-> lane bit 5, the radix-2 pass), natural store.  Both
//                  cross-wave passes between one pair of barriers -- round 5 made two trips through LDS with a barrier each (N = 4096:
//                  43 % of the wave cycles waiting).  The natural layout is what lets every wave store into the very words it read
//                  (no barrier in front of the stores); its price is a 2-way conflict on the four reads of N = 4096.
//                  (OUT_REGS keeps the round-5 form there: its results must end as elements t + m N/4.)
#ifndef SMFFT_QUARTER_PHASES
#define SMFFT_QUARTER_PHASES 1         // 0: the round-5 form (A/B)
#endif
// Natural order in PAIRS of passes (round 6, last day; N = 512, 1024 with results to LDS): a phase is TWO passes with one exchange between
// them, and that exchange is always the one of the lane bits 4, 5 (v_permlane16/32_swap) -- never the sixteen DPP-fed selects of the lane
// bits 2, 3 that a phase of three passes needs; the trips through LDS stay as many as before and conflict free (tools/quarter_phases_model.py,
// transform_pairs):
//   pass (0,1) in the thread, scattered as before; [pass (2,3) | <-> 4, 5 | pass (4,5)] on the wave's aligned block of 256
//   (slots = bits 2, 3; lane bits 0 ... 5 = bits 0, 1, 6, 7, 4, 5); swizzled image;
//   N = 1024: [pass (6,7) | <-> 4, 5 | pass (8,9)] with wave = bits 4, 5: a barrier in front of the natural stores;
//   N = 512:  [pass (6,7) | slot bit 0 <-> lane bit 5 | radix 2 on bit 8] with wave = bit 5: the wave owns whole aligned groups
//             of 32, which the swizzle permutes -- it stores into the words it read, no barrier.
// Measured in the in-LDS loop against the three-pass phase (one process): N = 512 +3 %, N = 1024 +4 % -- a third of what the instruction
// count promised (a select costs the vector unit less, a v_permlane*_swap more, than the 7.4 and 9 cycles measured in isolation); the same at
// N = 256 ([pass 0 | <-> 5, 4 | pass 1] image [pass 2 | <-> 4, 5 | pass 3]: sixteen swaps for sixteen selects + eight swaps) measured 2 % SLOWER
// and was taken out again (profiles/r06_contract_pairs.txt).
// 0: slots = bits (2, 3) read-back and three passes per phase at these lengths too (the form of the rest of round 6; A/B)
#ifndef SMFFT_QUARTER_PAIRS
#define SMFFT_QUARTER_PAIRS 1
#endif
// ... and at N = 2048 (inputs from LDS): [pass 0 | <-> 5, 4 | pass 1] in front of the scattered store -- the thread loads
// x[(lane & 15) + 16 wave + 128 (lane >> 4) + m N/4]: its lane bits 4, 5 hold the input bits under the slots' (position bits 3, 2); a 2-way conflicted
// read of the caller's natural layout -- then [pass 2 | <-> 4, 5 | pass 3] on the wave's aligned block (run_exchanged's read-back), then the
// last phase as before: 2 + 2 + 1.5 passes instead of 1 + 3 + 1.5, +7.5 ... 8.3 % in the in-LDS loop.  The same at N = 4096 (2 + 2 + 2): nothing
// (-0.7 ... +0.2 %; sixteen waves per block, two blocks per CU: the block's own latency chain bounds it, and that chain is as long) -- not taken.
// OUT_PHASED (with OUT_REGS; round 6): the results stay in x[] WHERE THE LAST PHASE LEAVES THEM -- x[i] = element quarter_phased_element<N>(t, i)
// of the transform -- so that N = 2048 / 4096 can take the last phase with registers out as well (for N <= 1024 that is t + i N/4, as without it).
template <int N>
__device__ __forceinline__ int quarter_phased_element(int t, int i) {
    constexpr int N_BITS = ilog2c(N), LOW = 16 - N_BITS;
    if constexpr (N < 2048 || SMFFT_QUARTER_LANES == 0 || SMFFT_QUARTER_PHASES == 0) {
        return t + i * (N / 4);
    } else {
        const int lane = t & 63, wave = t >> 6;
        const int k_last = (lane & ((1 << LOW) - 1)) | (wave << LOW) | ((lane >> LOW) << 8);
        return N_BITS == 12 ? k_last + 1024 * i : k_last + 1024 * (i & 1) + 512 * (i >> 1);
    }
}
template <int N, int DIR, int REORDER, int BLOCK_THREADS, bool IN_REGS = false, bool OUT_REGS = false, int ENGINE = 0, bool OUT_PHASED = false>
__device__ __forceinline__ void quarter_fft(float2 (&x)[4], float2* s, int t, int region_offset = 0) {
    constexpr bool kLanesOn = ENGINE == 2 || (ENGINE == 0 && SMFFT_QUARTER_LANES != 0);
    constexpr bool kPhases = kLanesOn && SMFFT_QUARTER_PHASES != 0;
    constexpr bool kOneWaveNatural = kPhases && N == 256 && REORDER && BLOCK_THREADS <= 64;      // phase 0 + phase 1 inside one wave
    if constexpr (kPhases && (N == 64 || N == 128) && !IN_REGS && !OUT_REGS && BLOCK_THREADS <= 64) {      // one trip through the block's image instead of an exchange (two) on lanes
        if constexpr (REORDER) quarter_small_natural<N, DIR>(s, t, region_offset);
        else quarter_small_noreorder<N, DIR>(s, t, region_offset);
        return;
    }
    if constexpr ((ENGINE == 2 || (ENGINE == 0 && quarter_lanes_default(N))) && N <= 256 && !OUT_REGS && !kOneWaveNatural && !(kPhases && N == 256 && !REORDER && BLOCK_THREADS <= 64)) {   // the ladder on lanes and registers (QuarterLanes above)
        QuarterLanes<N, DIR, REORDER>::template lds_to_lds<IN_REGS>(x, s, t, region_offset);
        return;
    }
    using R = QuarterTwiddleRows<N>;
    constexpr int Q = N / 4;
    constexpr bool kBarrier = BLOCK_THREADS > 64;
    // A workgroup barrier is needed only where data crosses waves.  Thread t's elements of a pass with P <= 64 lie in the
    // aligned block of 256 elements number t >> 6 -- its wave's -- and so do their swizzled places (the swizzle permutes
    // aligned groups of 32) and the elements of the first (no-reorder) and of the last pass (t + m N/4: thread and element
    // share their offset inside a group of 32): between those a wave-level fence orders everything.  What does cross waves:
    // the natural-order first pass (loads t + m N/4, stores 4 rev(t) + m), the passes with P >= 256 and the radix-2 pass.
    constexpr int T_BITS = ilog2c(Q), N_BITS = ilog2c(N);
    constexpr int kLastQuad = R::kOdd ? -1 : R::kPasses - 1;    // the pass whose results leave in natural order (none: the radix-2 pass is last)
    constexpr bool kLanes512 = kLanesOn && N >= 512;
    constexpr bool kOneWaveHead = kPhases && N == 256 && !REORDER && BLOCK_THREADS <= 64;
    constexpr bool kLanesHead = (kLanes512 && !REORDER) || kOneWaveHead;
    constexpr bool kLanesMiddle = (kLanes512 && REORDER) || kOneWaveNatural;
    constexpr bool kSlots23 = kLanesMiddle && kPhases;          // phase 1 reads slots = position bits (2, 3)
    static_assert(!OUT_PHASED || OUT_REGS, "OUT_PHASED qualifies OUT_REGS");
    constexpr bool kLastPhase = kPhases && N >= 2048 && (!OUT_REGS || OUT_PHASED) && (kLanesHead || kLanesMiddle);
    constexpr int kFirstLdsPass = (kLanesHead || kLanesMiddle) ? 4 : 1;
    constexpr bool kPairs = kPhases && SMFFT_QUARTER_PAIRS != 0 && REORDER && (N == 512 || N == 1024) && kLanesMiddle && !OUT_REGS;
    constexpr bool kPairsHead = kPhases && SMFFT_QUARTER_PAIRS != 0 && REORDER && N == 2048 && kLastPhase && kLanesMiddle && !IN_REGS;
    constexpr int W_BITS = N_BITS - 8;                   // waves of the block (N >= 512)
    // the thread of the last phase: lanes 0 ... LOW-1 = position bits 0 ... LOW-1, wave = the bits up to 7, slots = bits (8, 9), the other
    // lane bits = bits 10 (, 11); after the exchange in front of the last pass those lane bits hold bits 8 (, 9)
    constexpr int LOW = 16 - N_BITS;
    const int lane = t & 63, wave = t >> 6;
    const int k_last = kLastPhase ? ((lane & ((1 << LOW) - 1)) | (wave << LOW) | ((lane >> LOW) << 8)) : t;
    // the twiddles first (QuarterTwiddles): k = t mod P in every pass of this form -- and in the wave-local ladder below, whose
    // butterfly index is the lane = t mod 64 with P <= 64; the last phase: k_last
    // (pairs: the numbering of the threads changes at the second image -- the passes 1, 2 k = lane bits 0, 1, 4, 5, then k = the low bits of the
    //  thread's last-phase element k_pairs)
    const int k_pairs = N == 1024 ? ((lane & 15) | (wave << 4) | ((lane >> 4) << 6)) : ((lane & 31) | (wave << 5) | ((lane >> 5) << 6));
    QuarterTwiddles<N, DIR, REORDER, kLastPhase, kPairs ? 3 : 4> tw;
    if constexpr (kPairs) tw.load((lane & 3) | ((lane >> 4) << 2), k_pairs, k_pairs);
    else if constexpr (kPairsHead) tw.load(t, k_last, k_last, ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1));
    else tw.load(t, k_last, k_last);
    float2* sf = s + region_offset;
    float2 e[4];
    // N >= 512, no reorder: the first FOUR passes (index bits 0 ... 7) never leave the wave -- thread t = lane + 64 w starts with
    // elements 4 t + i, i.e. its wave owns the aligned block of 256 elements number w -- so they run on lanes and registers as
    // a 256-point ladder (QuarterLanes<256>: same twiddles W_4P^k, k = lane mod P); its results, elements 256 w + lane + 64 i, go
    // to the image, and only the passes that cross waves (P >= 256 and the radix-2 pass) follow behind a barrier.
    // Natural order: pass 0 carries the bit reversal (scattered stores into the swizzled image, across waves); after it the
    // wave again owns the aligned block of 256 elements number w, and the passes 1 ... 3 (P = 4, 16, 64: wave-local) run on lanes as well.
    if constexpr (kLanesHead) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = IN_REGS ? x[i] : sf[4 * t + i];
        if constexpr (kPhases) {
            QuarterLanes<256, DIR, 0>::run_exchanged(e, tw, sf + 256 * wave, lane, !IN_REGS);
            fft_sync<false>();                                // the exchange's loads precede the stores into the same 256 elements
        } else {
            QuarterLanes<256, DIR, 0>::run(e, tw);
            if constexpr (!IN_REGS) fft_sync<false>();        // the wave's own loads precede its stores into the same 256 elements
        }
    } else {
    // ---- pass 0 (P = 1): twiddles 1, 1, -+i -----------------------------------------------------------------------
    int a;
    if constexpr (REORDER) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {                                                   // e[i] = x[bitrev(4j + i)], i = rev2(m)
            if constexpr (IN_REGS) e[((m & 1) << 1) | (m >> 1)] = x[m];
            else if constexpr (kPairsHead) e[((m & 1) << 1) | (m >> 1)] = sf[(lane & 15) + 16 * wave + ((lane >> 4) << (4 + W_BITS)) + m * Q];
            else if constexpr (kPhases && N <= 2048) e[((m & 1) << 1) | (m >> 1)] = lds_load_single(sf + t + m * Q);
            else e[((m & 1) << 1) | (m >> 1)] = sf[t + m * Q];
        }
        a = 4 * (int)(T_BITS ? __brev((unsigned)t) >> (32 - (T_BITS ? T_BITS : 1)) : 0);
        if constexpr (!IN_REGS) fft_sync<kBarrier>();                                   // every load precedes the scattered stores
        // (a timing-only build WITHOUT this barrier is within +-1.5 % at every length, N = 4096 5 % slower: an image that lets every wave scatter
        //  into the words it has just read would buy nothing; profiles/r06_contract_phases.txt (4))
    } else {
        a = 4 * t;
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = IN_REGS ? x[i] : sf[a + i];
        if constexpr (!IN_REGS && kLastQuad != 0) fft_sync<false>();  // ... and the swizzled ones (same wave's)
    }
    {
        const float2 s0 = cadd(e[0], e[1]), d0 = csub(e[0], e[1]), s1 = cadd(e[2], e[3]), d1 = csub(e[2], e[3]);
        const float2 jd1 = DIR ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);   // -+i * d1
        if constexpr (kLastQuad == 0) {
            sf[a + 0] = cadd(s0, s1), sf[a + 2] = csub(s0, s1), sf[a + 1] = cadd(d0, jd1), sf[a + 3] = csub(d0, jd1);
        } else if constexpr (kPairsHead) {
            e[0] = cadd(s0, s1), e[1] = cadd(d0, jd1), e[2] = csub(s0, s1), e[3] = csub(d0, jd1);
            slots_swap<0, 5, true>(e);                             // lane bits 5, 4 hold position bits 2, 3
            slots_swap<1, 4, true>(e);
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), tw.late(1));
            // slots = position bits (2, 3); lane bits 5, 4 = bits 0, 1; wave = bits 4 ... (reversed); lane bits 0 ... 3 = the four highest (reversed)
            const int pw = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | ((int)(__brev((unsigned)wave) >> (32 - W_BITS)) << 4) |
                           ((int)(__brev((unsigned)(lane & 15)) >> 28) << (4 + W_BITS));
            const int a0 = quarter_swizzle(region_offset + pw);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[a0 ^ (4 * j)] = e[j];
        } else {
            const int a0 = kOneWaveNatural ? region_offset + quarter_image256(a) : quarter_swizzle(region_offset + a);   // a0 ^ m: the same aligned group of four
            s[a0] = cadd(s0, s1), s[a0 ^ 2] = csub(s0, s1), s[a0 ^ 1] = cadd(d0, jd1), s[a0 ^ 3] = csub(d0, jd1);
        }
    }
    if constexpr (kLanesMiddle) {
        fft_sync<kBarrier>();                                      // every wave's scattered stores precede the loads
        if constexpr (kPairs) {
            // ---- [pass 1 | <-> 4, 5 | pass 2] on the wave's aligned block: slots = position bits (2, 3); lane bits 0, 1 = bits 0, 1; 2, 3 = bits 6, 7; 4, 5 = bits 4, 5
            const int pb = (lane & 3) | ((lane >> 4) << 4) | (((lane >> 2) & 3) << 6) | (wave << 8);
            const int b0 = quarter_swizzle(region_offset + pb);
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = s[b0 ^ (4 * j)];
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(1), tw.late(1));
            slots_swap<0, 4, true>(e);
            slots_swap<1, 5, true>(e);
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(2), tw.late(2));
            fft_sync<false>();                                     // the wave's loads precede its stores into the same 256 elements
            // slots = bits (4, 5); lane bits 4, 5 = bits 2, 3
            const int c0 = quarter_swizzle(region_offset + ((lane & 3) | ((lane >> 4) << 2) | (((lane >> 2) & 3) << 6) | (wave << 8)));
#pragma unroll
            for (int j = 0; j < 4; ++j) s[c0 ^ quarter_swizzle(16 * j)] = e[j];
            fft_sync<kBarrier>();                                  // the owners change: every wave's block is in the image
            if constexpr (N == 1024) {
                // ---- [pass 3 | <-> 4, 5 | pass 4]: slots = bits (6, 7); lane bits 0 ... 3 = bits 0 ... 3; wave = bits 4, 5; lane bits 4, 5 = bits 8, 9
                const int d0 = quarter_swizzle(region_offset + ((lane & 15) | (wave << 4) | ((lane >> 4) << 8)));
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = s[d0 ^ quarter_swizzle(64 * j)];
                quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(3), tw.late(3));
                slots_swap<0, 4, true>(e);
                slots_swap<1, 5, true>(e);
                quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(4), tw.late(4));
                fft_sync<kBarrier>();                              // every wave's loads of the image precede the natural stores (a wave owns half of every group of 32)
#pragma unroll
                for (int i = 0; i < 4; ++i) sf[k_pairs + 256 * i] = e[i];
            } else {
                // ---- [pass 3 | slot bit 0 <-> lane bit 5 | radix 2 on bit 8]: slots = bits (6, 7); lane bits 0 ... 4 = bits 0 ... 4; wave = bit 5; lane bit 5 = bit 8
                const int d0 = quarter_swizzle(region_offset + ((lane & 31) | (wave << 5) | ((lane >> 5) << 8)));
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = s[d0 ^ quarter_swizzle(64 * j)];
                quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(3), tw.late(3));
                slots_swap<0, 5, true>(e);                         // slots = (bit 8, bit 7): elements k, k + 256, k + 128, k + 384
                const float2 w = tw.wr;
                const float2 t1 = cmul(e[1], w), v3 = cmul(e[3], w);
                const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
                fft_sync<false>();                                 // the wave's loads precede its stores into the same groups of 32
                sf[k_pairs] = cadd(e[0], t1);
                sf[k_pairs + N / 2] = csub(e[0], t1);
                sf[k_pairs + Q] = cadd(e[2], t3);
                sf[k_pairs + 3 * Q] = csub(e[2], t3);
            }
            return;
        } else if constexpr (kPairsHead) {
            // ---- [pass 2 | <-> 4, 5 | pass 3] on the wave's aligned block: slots = position bits (4, 5); lane bits 0 ... 3 = bits 0 ... 3; lane bits 4, 5 = bits 6, 7
            const int b0 = quarter_swizzle(region_offset + ((lane & 15) | ((lane >> 4) << 6) | (wave << 8)));
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = s[b0 ^ quarter_swizzle(16 * j)];
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(2), tw.late(2));
            QuarterLanes<256, DIR, 0>::template passes<3>(e, tw);
        } else if constexpr (kSlots23) {
            // slots = position bits (2, 3): elements p1 + 4 j, p1 = position bits 0, 1 from lane bits 0, 1 and bits 4 ... 7 from lane bits 2 ... 5
            const int p1 = (lane & 3) + 16 * (lane >> 2) + 256 * wave;
            const int b0 = kOneWaveNatural ? region_offset + quarter_image256(p1) : quarter_swizzle(region_offset + p1);      // (both images leave 4 j alone)
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = s[b0 ^ (4 * j)];
            QuarterLanes<256, DIR, 0>::passes_from_slots23(e, tw);
        } else {
            const int b0 = quarter_swizzle(region_offset + 4 * t);     // elements 4 t + i: one aligned group of four
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = s[b0 ^ i];
            QuarterLanes<256, DIR, 0>::template passes<1>(e, tw);
        }
        fft_sync<false>();                                         // the wave's own loads precede its stores into the same 256 elements
    }
    }
    if constexpr (kLanesHead || kLanesMiddle) {
        // the thread holds elements 256 w + lane + 64 i of its wave's block
        if constexpr (N == 256) {                                  // (one wave, natural order: the results)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (OUT_REGS) x[i] = e[i];
                else sf[t + 64 * i] = e[i];
            }
            return;
        } else if constexpr (kLastPhase) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sf[t + 192 * wave + 64 * i] = e[i];      // natural layout: 256 w + lane + 64 i
        } else {
            const int j0 = region_offset + (t & ~63) * 4 + (t & 63);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[quarter_swizzle(j0 + 64 * i)] = e[i];
        }
    }
    if constexpr (kLastPhase) {
        // ---- the last phase (N = 2048 / 4096): passes 4 and 5 (the radix-2 pass) on lanes and registers --------------------------------
        fft_sync<true>();                                          // every wave's block is in the image
        const float2* src = sf + ((lane & ((1 << LOW) - 1)) | (wave << LOW) | ((lane >> LOW) << 10));
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = src[256 * j];
        quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(4), tw.late(4));
        if constexpr (!R::kOdd) {
            slots_swap<0, 4, true>(e);
            slots_swap<1, 5, true>(e);
            quad_butterfly<DIR>(e[0], e[1], e[2], e[3], tw.of(5), tw.late(5));
            if constexpr (OUT_REGS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = e[i];           // elements k_last + 1024 i (quarter_phased_element)
                return;
            }
            fft_sync<false>();                                     // the wave's loads precede its stores into the same words
#pragma unroll
            for (int i = 0; i < 4; ++i) sf[k_last + 1024 * i] = e[i];
        } else {
            slots_swap<0, 5, true>(e);                             // slots = (bit 10, bit 9): elements k, k + 1024, k + 512, k + 1536
            const float2 w = tw.wr;
            const float2 t1 = cmul(e[1], w), v3 = cmul(e[3], w);
            const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
            if constexpr (OUT_REGS) {                              // elements k_last, + 1024, + 512, + 1536
                x[0] = cadd(e[0], t1), x[1] = csub(e[0], t1), x[2] = cadd(e[2], t3), x[3] = csub(e[2], t3);
                return;
            }
            fft_sync<false>();
            sf[k_last] = cadd(e[0], t1);
            sf[k_last + N / 2] = csub(e[0], t1);
            sf[k_last + Q] = cadd(e[2], t3);
            sf[k_last + 3 * Q] = csub(e[2], t3);
        }
        return;
    }
    // ---- passes 1 .. (P = 4, 16, ...) -----------------------------------------------------------------------------
    if constexpr (R::kPasses > kFirstLdsPass) {
        int P = 1 << (2 * kFirstLdsPass);
#pragma unroll
        for (int p = kFirstLdsPass; p < R::kPasses; ++p, P *= 4) {
            if (P > 64 || (REORDER && p == 1)) fft_sync<kBarrier>();
            else fft_sync<false>();
            const int k = t & (P - 1);
            const int base = ((t - k) << 2) + k;
            const int a0 = quarter_swizzle(region_offset + base);
            const int a1 = a0 ^ quarter_swizzle(P), a2 = a0 ^ quarter_swizzle(2 * P), a3 = a0 ^ quarter_swizzle(3 * P);
            // elements base, base + P, base + 2P, base + 3P; in the butterfly's order (w1 on the second, w2 on the third): x0, x1, x2, x3
            float2 x0 = s[a0], x1 = s[a1], x2 = s[a2], x3 = s[a3];
            quad_butterfly<DIR>(x0, x1, x2, x3, tw.of(p), tw.late(p));             // results: x0 -> base, x1 -> base + P, x2 -> base + 2P, x3 -> base + 3P
            if (p == kLastQuad) {
                if constexpr (OUT_REGS) {                                                // base = t, P = N / 4
                    x[0] = x0, x[2] = x2, x[1] = x1, x[3] = x3;
                } else {
                    fft_sync<false>();                                                   // every swizzled load precedes the natural stores (same wave's)
                    sf[base] = x0, sf[base + 2 * P] = x2, sf[base + P] = x1, sf[base + 3 * P] = x3;
                }
            } else {
                s[a0] = x0, s[a2] = x2, s[a1] = x1, s[a3] = x3;
            }
        }
    }
    // ---- odd log2 N: the last radix-2 stage (span N/2), two butterflies per thread ----------------------------------
    if constexpr (R::kOdd) {
        fft_sync<kBarrier>();
        const float2 w = tw.wr;
        const int a0 = quarter_swizzle(region_offset + t);
        const float2 x0 = s[a0], x1 = s[a0 ^ quarter_swizzle(N / 2)], x2 = s[a0 ^ quarter_swizzle(Q)], x3 = s[a0 ^ quarter_swizzle(3 * Q)];
        const float2 t1 = cmul(x1, w), v3 = cmul(x3, w);
        const float2 t3 = DIR ? make_float2(-v3.y, v3.x) : make_float2(v3.y, -v3.x);
        if constexpr (OUT_REGS) {
            x[0] = cadd(x0, t1), x[2] = csub(x0, t1), x[1] = cadd(x2, t3), x[3] = csub(x2, t3);
        } else {
            fft_sync<false>();                                                           // every swizzled load precedes the natural stores (same wave's)
            sf[t] = cadd(x0, t1);
            sf[t + N / 2] = csub(x0, t1);
            sf[t + Q] = cadd(x2, t3);
            sf[t + 3 * Q] = csub(x2, t3);
        }
    }
}
// in place on the LDS region (the reference's contract)
template <int N, int DIR, int REORDER, int BLOCK_THREADS, int ENGINE = 0>
__device__ __forceinline__ void quarter_fft_inplace(float2* s, int t, int region_offset = 0) {
    float2 unused[4];
    quarter_fft<N, DIR, REORDER, BLOCK_THREADS, false, false, ENGINE>(unused, s, t, region_offset);
}

// Hermitian split / merge on the reference's thread shape (L/4 threads, two pairs each: i = t + 1 and t + 1 + L/4, RC:289-328)
// (w: hermitian_twiddles_quarter(t), fetched by the caller in front of everything else for the reason QuarterTwiddles gives)
template <int L, int DIR>
struct HermitianTwiddles { float2 w[2]; };
template <int L, int DIR>
__device__ __forceinline__ HermitianTwiddles<L, DIR> hermitian_twiddles_quarter(int t) {
    constexpr float ohx = DIR ? -0.5f : 0.5f;
    HermitianTwiddles<L, DIR> h;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float2 W = twiddle<DIR>((1 + t + k * (L / 4)) * (4096 / (2 * L)));
        h.w[k] = make_float2(ohx * W.x, ohx * W.y);
    }
    return h;
}
template <int L, int DIR>
__device__ __forceinline__ void hermitian_pass_quarter(float2* sf, int t, const HermitianTwiddles<L, DIR>& tw) {
    if (DIR && t == 0) {
        const float2 z = sf[0];
        sf[0] = make_float2(0.5f * (z.x + z.y), 0.5f * (z.x - z.y));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = 1 + t + h * (L / 4);
        const float2 A = sf[i], B = sf[L - i];
        const float2 Wh = tw.w[h];
        const float2 S = make_float2(A.x + B.x, A.y - B.y);
        const float2 D = make_float2(A.y + B.y, A.x - B.x);
        const float2 WH = make_float2(fmaf(Wh.x, D.x, Wh.y * D.y), fmaf(Wh.y, D.x, -Wh.x * D.y));
        sf[i] = make_float2(fmaf(0.5f, S.x, WH.x), fmaf(0.5f, S.y, WH.y));
        sf[L - i] = make_float2(fmaf(0.5f, S.x, -WH.x), fmaf(-0.5f, S.y, WH.y));   // i == L/2 (t = L/4 - 1, h = 1): this value stays (RC:308)
    }
    if (!DIR && t == 0) {
        const float2 z = sf[0];
        sf[0] = make_float2(z.x + z.y, z.x - z.y);
    }
}

namespace tiled {

// the tiled contract is 256 threads per workgroup: anything else would index other FFTs' regions
__device__ __forceinline__ void require_tiled_block() {
    if (blockDim.x != 256) __builtin_trap();
}

template <class const_params>
__device__ void do_SMFFT_CT_DIT(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_size, const_params::fft_direction, const_params::fft_reorder> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
// Stockham C2C program: un-normalised INVERSE (+i) transform, natural order (ST:76, :429).
template <class const_params>
__device__ void do_FFT_Stockham_mk6(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, 1, 1> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, const_direction::fft_direction, 1> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_R2C_C2R(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, const_direction::fft_direction, 1> eng;
    eng.init(threadIdx.x);
    r2c_c2r_lds_inplace<const_params::fft_length, const_direction::fft_direction>(s_input, eng);
}

}  // namespace tiled
}  // namespace smfft

// =================================================================================================
// (1) the reference's contract
// =================================================================================================
template <class const_params>
__device__ void do_SMFFT_CT_DIT(float2* s_input) {
    constexpr int N = const_params::fft_size;
    constexpr int kBlock = const_params::fft_length / 4;                 // 32 threads hold 128 / N transforms for N <= 128 (CT:586-595)
    const int f = threadIdx.x / (N / 4), t = threadIdx.x % (N / 4);
    smfft::quarter_fft_inplace<N, const_params::fft_direction, const_params::fft_reorder, kBlock>(s_input, t, f * N);
}

// The same transform on the block's REGISTERS, in the reference's thread shape (an extension; N >= 256, blockDim.x = N / 4):
// x[m] = element threadIdx.x + m N/4 of the block's transform going in (no-reorder classes: element 4 threadIdx.x + m --
// what the first pass consumes) and element threadIdx.x + m N/4 of the result coming out; s_scratch = P::fft_sm_required
// float2 of LDS that nobody is still reading.  For kernels that load from and store to global memory anyway: one LDS round
// trip and one synchronisation less at either end than do_SMFFT_CT_DIT between a fill and a drain of s.
template <class const_params>
__device__ void do_SMFFT_CT_DIT_registers(float2 (&x)[4], float2* s_scratch) {
    constexpr int N = const_params::fft_size;
    static_assert(N >= 256 && const_params::fft_length == N, "one transform per block of N / 4 threads");
    smfft::quarter_fft<N, const_params::fft_direction, const_params::fft_reorder, N / 4, true, true>(x, s_scratch, threadIdx.x);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C_registers(float2 (&x)[4], float2* s_scratch) {
    constexpr int N = const_params::fft_length;
    smfft::quarter_fft<N, const_direction::fft_direction, 1, N / 4, true, true>(x, s_scratch, threadIdx.x);
}
// ... and with the results left where the transform's last phase puts them (round 6): x[i] = element element[i] of the result on return --
// threadIdx.x + i N/4 for N <= 1024; for N = 2048 / 4096 the sixteen-element runs of smfft::quarter_phased_element, which spare those lengths
// a trip through LDS and a barrier (a kernel that stores its results to global memory does not care which thread stores which element;
// a store instruction of a wave still writes four 128-byte runs).  Inputs as for do_SMFFT_CT_DIT_registers.
template <class const_params>
__device__ void do_SMFFT_CT_DIT_registers_out(float2 (&x)[4], float2* s_scratch, int (&element)[4]) {
    constexpr int N = const_params::fft_size;
    static_assert(N >= 256 && const_params::fft_length == N, "one transform per block of N / 4 threads");
    smfft::quarter_fft<N, const_params::fft_direction, const_params::fft_reorder, N / 4, true, true, 0, true>(x, s_scratch, threadIdx.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) element[i] = smfft::quarter_phased_element<N>(threadIdx.x, i);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C_registers_out(float2 (&x)[4], float2* s_scratch, int (&element)[4]) {
    constexpr int N = const_params::fft_length;
    smfft::quarter_fft<N, const_direction::fft_direction, 1, N / 4, true, true, 0, true>(x, s_scratch, threadIdx.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) element[i] = smfft::quarter_phased_element<N>(threadIdx.x, i);
}

template <class const_params>
__device__ void do_FFT_Stockham_mk6(float2* s_input) {
    constexpr int N = const_params::fft_length;
    smfft::quarter_fft_inplace<N, 1, 1, N / 4>(s_input, threadIdx.x);
    __syncthreads();   // upstream's function ends with a barrier (ST:239) and its kernels store right after the call (ST:253)
}

template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C(float2* s_input) {
    constexpr int N = const_params::fft_length;
    smfft::quarter_fft_inplace<N, const_direction::fft_direction, 1, N / 4>(s_input, threadIdx.x);
    __syncthreads();   // upstream's function ends with a barrier (RC:265) and its callers rely on it (RC:360-361)
}

template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_R2C_C2R(float2* s_input) {
    constexpr int L = const_params::fft_length;
    constexpr int D = const_direction::fft_direction;
    const auto herm = smfft::hermitian_twiddles_quarter<L, D>(threadIdx.x);
    if constexpr (D == 0) {
        smfft::quarter_fft_inplace<L, 0, 1, L / 4>(s_input, threadIdx.x);
        __syncthreads();
        smfft::hermitian_pass_quarter<L, 0>(s_input, threadIdx.x, herm);
    } else {
        smfft::hermitian_pass_quarter<L, 1>(s_input, threadIdx.x, herm);
        __syncthreads();
        smfft::quarter_fft_inplace<L, 1, 1, L / 4>(s_input, threadIdx.x);
    }
    __syncthreads();   // as upstream: the forward branch ends behind a barrier (RC:330), the inverse one in do_FFT_Stockham_C2C
}

// ---- kernels in the reference's launch shape (own text; CT:534-572) ----
// (N >= 256: the block's transform on registers -- the four loads of a thread are what the first pass consumes and the last
//  pass's results are what it stores; SMFFT_CONTRACT_FUSED_IO=0 keeps the fill / call / drain form of CT:534-551, which is also
//  what a user's kernel around do_SMFFT_CT_DIT looks like: examples/reference_shape_kernel.hip)
#ifndef SMFFT_CONTRACT_FUSED_IO
#define SMFFT_CONTRACT_FUSED_IO 1
#endif
#ifndef SMFFT_CONTRACT_ROTATE_PRIORITY
// log2 of the rotation period of the two-argument `multiple` kernels' wave priorities in shader clocks; 0 (default): off.  In the
// reference's launch shape a launch is several rounds of short-lived blocks, where the arbiter's oldest-first order does no harm:
// with 15 the in-LDS contract path measured 1-4 % SLOWER (the library's own persistent schedule gains 10-14 % from it).
#define SMFFT_CONTRACT_ROTATE_PRIORITY 0
#endif
// In upstream's 32-thread blocks of N <= 128 the two-argument external kernel loads a thread's four
// elements from global memory straight into the lane-and-register ladder (QuarterLanes) and stores its four results from there:
// no LDS at all (0.84 -> 0.91 of the tiled rate at N = 32 / 64, 0.71 -> 0.82 at N = 128).  Natural order: the lanes of a
// transform store bit-reversed positions of a contiguous run of N/4 elements.  The 64-thread blocks of the _wave64 classes keep
// fill / transform in LDS / drain on the LDS form of the ladder, which measured 3-6 % faster there (profiles/r04_contract_lanes.txt).
template <class const_params>
__global__ void SMFFT_DIT_external(float2* d_input, float2* d_output) {
    // (N = 128 natural order in upstream's 32-thread blocks: fill / do_SMFFT_CT_DIT / drain instead -- since round 6 that function makes one trip through
    //  the block's image where this LDS-free ladder pays two more exchanges on lanes: 1.92 -> 1.84 ms on the config-2 batch)
    constexpr bool kStagedSmall = SMFFT_QUARTER_PHASES != 0 && const_params::fft_size == 128 && const_params::fft_reorder != 0;
    if constexpr (SMFFT_CONTRACT_FUSED_IO && SMFFT_QUARTER_LANES != 0 && const_params::fft_size <= 128 && const_params::fft_length_quarter == 32 && !kStagedSmall) {
        constexpr int N = const_params::fft_size, Q = N / 4;
        using L = smfft::QuarterLanes<N, const_params::fft_direction, const_params::fft_reorder>;
        const int t = threadIdx.x % Q;
        const float2* in = d_input + (size_t)blockIdx.x * const_params::fft_length + (threadIdx.x / Q) * N;
        float2* out = d_output + (size_t)blockIdx.x * const_params::fft_length + (threadIdx.x / Q) * N;
        const auto tw = L::twiddles_of(t);
        float2 e[4];
        if constexpr (const_params::fft_reorder) {
#pragma unroll
            for (int m = 0; m < 4; ++m) e[((m & 1) << 1) | (m >> 1)] = in[t + m * Q];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = in[4 * t + i];
        }
        L::run(e, tw);
        const int base = L::base_of(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) out[base + L::out_offset(i)] = e[i];
        return;
    }
    __shared__ float2 s_input[const_params::fft_sm_required];
    if constexpr (SMFFT_CONTRACT_FUSED_IO && const_params::fft_size >= 256) {
        const int block = blockIdx.x * const_params::fft_length;
        float2 x[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            x[m] = const_params::fft_reorder ? d_input[block + threadIdx.x + m * const_params::fft_length_quarter] : d_input[block + 4 * threadIdx.x + m];
        int element[4];
        do_SMFFT_CT_DIT_registers_out<const_params>(x, s_input, element);
#pragma unroll
        for (int m = 0; m < 4; ++m) d_output[block + element[m]] = x[m];
        return;
    }
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    s_input[threadIdx.x] = d_input[base];
    s_input[threadIdx.x + const_params::fft_length_quarter] = d_input[base + const_params::fft_length_quarter];
    s_input[threadIdx.x + const_params::fft_length_half] = d_input[base + const_params::fft_length_half];
    s_input[threadIdx.x + const_params::fft_length_three_quarters] = d_input[base + const_params::fft_length_three_quarters];
    __syncthreads();
    if constexpr (const_params::fft_size <= 128 && !(kStagedSmall && const_params::fft_length_quarter == 32)) {     // the LDS form of the ladder between fill and drain (3-6 % faster there than do_SMFFT_CT_DIT's lane form)
        constexpr int N = const_params::fft_size;
        smfft::quarter_fft_inplace<N, const_params::fft_direction, const_params::fft_reorder, const_params::fft_length / 4, 1>(s_input, threadIdx.x % (N / 4), (threadIdx.x / (N / 4)) * N);
    } else {
        do_SMFFT_CT_DIT<const_params>(s_input);
    }
    __syncthreads();
    d_output[base] = s_input[threadIdx.x];
    d_output[base + const_params::fft_length_quarter] = s_input[threadIdx.x + const_params::fft_length_quarter];
    d_output[base + const_params::fft_length_half] = s_input[threadIdx.x + const_params::fft_length_half];
    d_output[base + const_params::fft_length_three_quarters] = s_input[threadIdx.x + const_params::fft_length_three_quarters];
}

template <class const_params>
__global__ void SMFFT_DIT_multiple(float2* d_input, float2* d_output) {
    __shared__ float2 s_input[const_params::fft_sm_required];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    s_input[threadIdx.x] = d_input[base];
    s_input[threadIdx.x + const_params::fft_length_quarter] = d_input[base + const_params::fft_length_quarter];
    s_input[threadIdx.x + const_params::fft_length_half] = d_input[base + const_params::fft_length_half];
    s_input[threadIdx.x + const_params::fft_length_three_quarters] = d_input[base + const_params::fft_length_three_quarters];
    __syncthreads();
    // (SMFFT_CONTRACT_ROTATE_PRIORITY: the blocks that share a SIMD advance at the same average rate instead of oldest first --
    //  smfft_engine.hpp, WavePriority; a user's own loop around the device function can do the same)
    const smfft::WavePriority priority(SMFFT_CONTRACT_ROTATE_PRIORITY);
    for (int f = 0; f < NREUSES; f++) {
        priority.at_application(f);
        do_SMFFT_CT_DIT<const_params>(s_input);
        __syncthreads();   // the reference has none here (latent race, CT:563-565)
    }
    d_output[base] = s_input[threadIdx.x];
    d_output[base + const_params::fft_length_quarter] = s_input[threadIdx.x + const_params::fft_length_quarter];
    d_output[base + const_params::fft_length_half] = s_input[threadIdx.x + const_params::fft_length_half];
    d_output[base + const_params::fft_length_three_quarters] = s_input[threadIdx.x + const_params::fft_length_three_quarters];
}

// Stockham C2C program, ST:243-278: dynamic LDS of FFT_size * 8 bytes (ST:319), blockDim.x = N / 4
template <class const_params>
__global__ void FFT_GPU_external(float2* d_input, float2* d_output) {
    extern __shared__ float2 s_input_dynamic[];
    float2* s_input = s_input_dynamic;
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    if constexpr (SMFFT_CONTRACT_FUSED_IO) {
        float2 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = d_input[base + k * const_params::fft_quarter];
        int element[4];
        do_FFT_Stockham_C2C_registers_out<const_params, FFT_inverse>(x, s_input, element);
#pragma unroll
        for (int k = 0; k < 4; ++k) d_output[base - threadIdx.x + element[k]] = x[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    do_FFT_Stockham_mk6<const_params>(s_input);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

template <class const_params>
__global__ void FFT_GPU_multiple(float2* d_input, float2* d_output) {   // ST:260-278
    extern __shared__ float2 s_input_dynamic[];
    float2* s_input = s_input_dynamic;
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    const smfft::WavePriority priority(SMFFT_CONTRACT_ROTATE_PRIORITY);
    for (int f = 0; f < NREUSES; f++) {
        priority.at_application(f);
        do_FFT_Stockham_mk6<const_params>(s_input);   // the function ends with a barrier
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

// R2C / C2R program, RC:349-365: L + 1 float2 of LDS, blockDim.x = L / 4, L = const_params::fft_length = real length / 2
template <class const_params, class const_direction>
__global__ void FFT_GPU_R2C_C2R_external(float2* d_input, float2* d_output) {
    __shared__ float2 s_input[const_params::fft_length + 1];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    if constexpr (SMFFT_CONTRACT_FUSED_IO) {
        // the complex transform takes its inputs from (R2C) / leaves its results in (C2R) the thread's registers; the split /
        // merge works on the natural layout in LDS as upstream (RC:269-344)
        constexpr int L = const_params::fft_length;
        float2 x[4];
        if constexpr (const_direction::fft_direction == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = d_input[base + k * const_params::fft_quarter];
            const auto herm = smfft::hermitian_twiddles_quarter<L, 0>(threadIdx.x);
            smfft::quarter_fft<L, 0, 1, L / 4, true, false>(x, s_input, threadIdx.x);
            __syncthreads();
            smfft::hermitian_pass_quarter<L, 0>(s_input, threadIdx.x, herm);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
            const auto herm = smfft::hermitian_twiddles_quarter<L, 1>(threadIdx.x);
            __syncthreads();
            smfft::hermitian_pass_quarter<L, 1>(s_input, threadIdx.x, herm);
            __syncthreads();
            smfft::quarter_fft<L, 1, 1, L / 4, false, true, 0, true>(x, s_input, threadIdx.x);
#pragma unroll
            for (int k = 0; k < 4; ++k) d_output[base - threadIdx.x + smfft::quarter_phased_element<L>(threadIdx.x, k)] = x[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    do_FFT_Stockham_R2C_C2R<const_params, const_direction>(s_input);
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

template <class const_params, class const_direction>
__global__ void FFT_GPU_R2C_C2R_multiple(float2* d_input, float2* d_output) {   // RC:367-384
    __shared__ float2 s_input[const_params::fft_length + 1];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    const smfft::WavePriority priority(SMFFT_CONTRACT_ROTATE_PRIORITY);
    for (int f = 0; f < NREUSES; f++) {
        priority.at_application(f);
        do_FFT_Stockham_R2C_C2R<const_params, const_direction>(s_input);   // ends with a barrier
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}
