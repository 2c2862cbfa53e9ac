// smfft_planar.hpp -- the engine of the in-LDS (`multiple`) kernels: the same transform as smfft_engine.hpp (same radix
// plan N = R1 * RM * 16, same small DFTs, same twiddle table), with every LDS image kept as TWO PLANES OF DWORDS
// (real parts, imaginary parts) instead of an array of float2.
//
// Why (measured on MI355X, tools/microbench/lds_forms.hip, profiles/r03_lds_forms.txt; decomposition of the round-2
// kernel in profiles/r03_decompose_1024.txt):
//   * A float2 store (ds_write_b64) costs ~6 LDS cycles per wave instruction because the address and data VGPRs travel
//     to the LDS at 2 cycles per dword; ds_write_addtid_b32 has no address VGPR (address = M0 + offset + 4 * lane) and
//     costs 2 -- a float2 as two of them 4.  With 256 multiply-adds beside one exchange of a wave's 1024 elements the
//     float2 form runs 0.556 us per exchange per SIMD, the planar form 0.45-0.47 (the arithmetic alone: 0.44).
//   * addtid stores are lane linear (register j of thread tid -> row j, dword tid), so all freedom is in WHO READS WHAT:
//     every exchange is arranged so that a reader's sixteen values are runs of contiguous dwords of few rows
//     (ds_read_b128 / ds_read_b64: the full 256 B/clk) and the rows are shifted against each other so that those reads
//     are bank-conflict free (tools/soa_model.py states every read pattern, counts conflicts with the gfx950 lane-group
//     rules and searched the shifts; tests/test_planar_layout_model.py ties this header's tables to it).
//   * With exchanges this cheap, exchange 1 of N = 1024 goes through LDS as well: the 32 v_permlane*_swap of the
//     register form were 124 ns of the 718 ns an N = 1024 application took per SIMD.  (N = 512 keeps its 16 swaps.)
//
// Thread positions and roles.  A compact workgroup has TW = max(64, T) threads, T = N / 16 per FFT; thread tid is at
// position v = tid % T of FFT tid / T, and every store puts register j at dword tid of row j.
//   three-pass sizes (N >= 512, T = 16 * RM):
//     pass 1      role t1 = t2 + 16 * r2 with v = RM * t2 + r2       (so that the middle pass reads RM contiguous dwords)
//     middle      role (t2, a) with v = 16 * a + t2                   (so that the last pass reads 16 contiguous dwords)
//     last        output index klow = v (no reorder: the stored result is lane linear in klow, which the bit-reversed
//                 read of the next application needs) or klow = pass-1 role of v (reorder: the registers a thread ends
//                 with are the ones it starts the next application with -- forwarded, never re-loaded)
//     exceptions  N = 512: exchange 1 in registers, roles = positions; N = 2048 no reorder: r2 bit-reversed; N = 4096: rotated
//                 exchange-1 reads and klow = role in both orderings (PlanarEngine below says why)
//   two-pass sizes (N = 64, 128, 256): pass-1 role t1 = v, last pass q1 = v.
#pragma once
#include "smfft_engine.hpp"

namespace smfft {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// ds_write_addtid_b32 x 8: four elements (re, im).  M0 holds the wave's base (plane 0 + 4 * 64 * wave) for the duration of the
// block; an SALU write of M0 needs one wait state before an LDS "add-TID" instruction reads it (the assembler does not see into
// inline assembly).  LLVM treats M0 as reserved (a clobber would only draw a warning) and may share ONE M0 set-up between several
// of its own uses (LDS-DMA loads, GWS, s_movrel register indexing, s_sendmsg), so the block SAVES M0 and RESTORES it: a user
// kernel may combine this engine with any of those (tests/test_gpu_parity.py::test_planar_stores_preserve_m0 runs LDS-DMA loads
// around planar stores).  The DS instructions read M0 when they issue, in order, so the restore may follow them directly.
// What remains the caller's business in a MULTI-WAVE kernel: the compiler does not count these stores, so consumers in other
// waves must be ordered by planar_sync (s_waitcnt lgkmcnt(0) + barrier), never by a bare __syncthreads().
template <int O0, int O1, int O2, int O3, int P>
__device__ __forceinline__ void addtid_store4(unsigned m0, float2 a, float2 b, float2 c, float2 d) {
    unsigned saved;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %1 offset:%10\n\tds_write_addtid_b32 %2 offset:%11\n\t"
        "ds_write_addtid_b32 %3 offset:%12\n\tds_write_addtid_b32 %4 offset:%13\n\t"
        "ds_write_addtid_b32 %5 offset:%14\n\tds_write_addtid_b32 %6 offset:%15\n\t"
        "ds_write_addtid_b32 %7 offset:%16\n\tds_write_addtid_b32 %8 offset:%17\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(saved)
        : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(c.x), "v"(c.y), "v"(d.x), "v"(d.y), "s"(m0),
          "n"(4 * O0), "n"(4 * (O0 + P)), "n"(4 * O1), "n"(4 * (O1 + P)), "n"(4 * O2), "n"(4 * (O2 + P)), "n"(4 * O3), "n"(4 * (O3 + P))
        : "memory");
}

// Row placement.  A row holds TW dwords (one per thread of the workgroup); row j of an exchange starts at
// TW * rank(j) + 4 * q[j] dwords, rank = the row's place in the order of the residues q (so rows never overlap and a plane
// is at most 16 * TW + 60 dwords).  Only q[j] (the row's base / 4 mod 16) matters for bank conflicts; the residues per
// length and exchange are the ones tools/soa_model.py found conflict free.
struct RowTable {
    int base[16];
    int span;
};
constexpr RowTable place_rows(int tw, int q0, int q1, int q2, int q3, int q4, int q5, int q6, int q7, int q8, int q9, int q10, int q11, int q12, int q13, int q14, int q15) {
    const int q[16] = {q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11, q12, q13, q14, q15};
    RowTable t{};
    int span = 0;
    for (int j = 0; j < 16; ++j) {
        int rank = 0;
        for (int k = 0; k < 16; ++k) rank += (q[k] < q[j] || (q[k] == q[j] && k < j)) ? 1 : 0;
        t.base[j] = tw * rank + 4 * q[j];
        span = t.base[j] + tw > span ? t.base[j] + tw : span;
    }
    t.span = span;
    return t;
}
constexpr int bit_of(int j, int b) { return (j >> b) & 1; }
// residue of row j as a function of (j's bits): the tables of tools/soa_model.py in closed form
enum class RowKind { image, x1, x2 };
template <int N, int REORDER>
constexpr int row_residue(RowKind kind, int j) {
    switch (kind) {
        case RowKind::image:    // read by the bit-reversed loads of the no-reorder variants (lane linear in the reorder variants)
            return N == 32 ? 8 * bit_of(j, 3) : N == 64 ? 4 * (j >> 2) : N == 128 ? bit_of(j, 2) + 8 * bit_of(j, 3) : N == 256 ? j : N == 512 ? bit_of(j, 0) + 2 * bit_of(j, 2) + 8 * bit_of(j, 3) : N == 4096 ? 4 * bit_of(j, 3) : (j >> 2);
        case RowKind::x1:       // after pass 1 (three-pass lengths)
            return N == 2048 ? bit_of(j, 1) : N == 4096 ? 2 * bit_of(j, 0) : 0;  // (N = 512 exchanges in registers)
        default:                // in front of the last pass
            if (N == 32) return 8 * bit_of(j, 0);
            if (N == 64) return 4 * (j & 3);
            if (N == 128) return bit_of(j, 0) + 8 * bit_of(j, 1);
            if (N == 256) return (j & 3) + 8 * bit_of(j, 3);
            if ((REORDER && N != 512) || N == 4096) return (j & 3) + 8 * bit_of(j, 3);   // klow = pass-1 role
            return N == 512 ? bit_of(j, 0) + 2 * bit_of(j, 1) + 8 * bit_of(j, 2) : N == 1024 ? (j >> 2) : bit_of(j, 0) + 2 * bit_of(j, 3);
    }
}
template <int N, int REORDER>
constexpr RowTable make_rows(RowKind k) {
    constexpr int tw = Geometry<N>::kCompactThreads;
    return place_rows(tw, row_residue<N, REORDER>(k, 0), row_residue<N, REORDER>(k, 1), row_residue<N, REORDER>(k, 2), row_residue<N, REORDER>(k, 3),
                      row_residue<N, REORDER>(k, 4), row_residue<N, REORDER>(k, 5), row_residue<N, REORDER>(k, 6), row_residue<N, REORDER>(k, 7),
                      row_residue<N, REORDER>(k, 8), row_residue<N, REORDER>(k, 9), row_residue<N, REORDER>(k, 10), row_residue<N, REORDER>(k, 11),
                      row_residue<N, REORDER>(k, 12), row_residue<N, REORDER>(k, 13), row_residue<N, REORDER>(k, 14), row_residue<N, REORDER>(k, 15));
}
// Synchronisation between the threads of an FFT.  The compiler does not count the stores issued from inline assembly, so a
// multi-wave FFT drains them itself in front of the workgroup barrier; inside one wave DS operations execute in order.
template <bool MULTI_WAVE>
__device__ __forceinline__ void planar_sync() {
    if (MULTI_WAVE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fft_sync<MULTI_WAVE>();
}

template <int N, int REORDER>
struct PlanarGeometry {
    using G = Geometry<N>;
    static constexpr int T = G::T, RM = G::RM, BM = G::BM, R1 = G::R1, B1 = G::B1;
    static constexpr int TW = G::kCompactThreads;
    static constexpr int F = TW / T;
    static constexpr RowTable kImage = make_rows<N, REORDER>(RowKind::image), kX1 = make_rows<N, REORDER>(RowKind::x1), kX2 = make_rows<N, REORDER>(RowKind::x2);
    static constexpr int image_row(int j) { return kImage.base[j]; }
    // A row's base for a row that is known at run time only (the tile copies; a thread's offsets at set-up), without a table in memory -- sixteen dependent
    // constant-memory loads per copy were 2-3 of the 3.5 us a tile copy took (round 5 traces): a row's base is TW * rank + 4 * residue
    // (place_rows), and the sixteen ranks and residues are four bits each: two 64-bit constants, one shift and mask per look-up.
    static constexpr const RowTable& table_of(RowKind kind) { return kind == RowKind::image ? kImage : kind == RowKind::x1 ? kX1 : kX2; }
    static constexpr unsigned long long pack_rows(RowKind kind, bool ranks) {
        unsigned long long packed = 0;
        for (int j = 0; j < 16; ++j) {
            const int q = row_residue<N, REORDER>(kind, j);
            const int rank = (table_of(kind).base[j] - 4 * q) / TW;
            packed |= (unsigned long long)((ranks ? rank : q) & 15) << (4 * j);
        }
        return packed;
    }
    static constexpr bool packing_is_exact(RowKind kind) {
        for (int j = 0; j < 16; ++j) {
            const int q = row_residue<N, REORDER>(kind, j);
            if (q < 0 || q > 15 || (table_of(kind).base[j] - 4 * q) % TW != 0 || (table_of(kind).base[j] - 4 * q) / TW > 15) return false;
        }
        return true;
    }
    template <RowKind KIND>
    __device__ static __forceinline__ int row_at(int j) {
        static_assert(packing_is_exact(KIND), "a row's base is TW * rank + 4 * residue with both below 16");
        constexpr unsigned long long ranks = pack_rows(KIND, true), residues = pack_rows(KIND, false);
        return TW * (int)((ranks >> (4 * j)) & 15) + 4 * (int)((residues >> (4 * j)) & 15);
    }
    __device__ static __forceinline__ int image_row_at(int j) { return row_at<RowKind::image>(j); }
    static constexpr int x1_row(int j) { return kX1.base[j]; }
    static constexpr int x2_row(int j) { return kX2.base[j]; }
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int kPlane = max3(kImage.span, RM > 1 ? kX1.span : 0, kX2.span);   // dwords per plane (a multiple of 4)
    // (N = 4096 with its fifteen middle-pass twiddles read from a table in LDS instead of 30 registers measured 7 % slower:
    //  profiles/r03_ab_planar_c.txt)
    static constexpr int kLdsFloats = 2 * kPlane;
};

template <int N, int DIR, int REORDER>
struct PlanarEngine {
    using G = Geometry<N>;
    using P = PlanarGeometry<N, REORDER>;
    static constexpr int T = G::T, RM = G::RM, BM = G::BM, R1 = G::R1, B1 = G::B1, TW = P::TW;
    static constexpr int T_BITS = ilog2c(T), R1_BITS = ilog2c(R1), B1_BITS = ilog2c(B1), RM_BITS = ilog2c(RM);
    static constexpr bool kThreePass = RM > 1;
    // N = 512 (RM = 2): exchange 1 moves data between two threads only -- sixteen v_permlane16_swap (Engine::exchange1_registers)
    // measured faster than a third trip through LDS (profiles/r03_ab_planar_all.txt, r04_ab_512_lds_x1.txt); its roles are the
    // register engine's (t1 = v)
    static constexpr bool kRegisterX1 = (RM == 2);
    // N = 4096 (RM = 16: sixteen consecutive threads share t2, so eight lanes of a ds_read_b128 group would read blocks of ONE
    // row -- 2-way conflicts no row shift can undo; round 3's first form had them on three reads, 384 of 1541 LDS cycles per FFT):
    //  * exchange 1: a lane reads the four quads of its run in an order rotated by rot = t2 >> 3.  Its registers then hold the
    //    radix-16 input cyclically shifted by 4*rot, i.e. the output times (-+i)^(q2*rot) -- folded into the middle twiddle at set-up;
    //  * the last pass computes klow = the thread's pass-1 role in BOTH orderings (every lane of a read group then reads a
    //    different row); the no-reorder image read, whose sixteen elements are then 16 dwords apart, is sixteen ds_read_b32 per
    //    plane: conflict free at half the rate = what the conflicted ds_read_b128 cost.
    static constexpr bool kRotatedX1 = (RM == 16);
    static constexpr bool kRoleOutputs = REORDER || RM == 16;   // the last-pass thread computes klow = its own pass-1 role
    static constexpr bool kForward = kRoleOutputs;
    // no reorder, N = 2048: the RM threads that share t2 take r2 in bit-reversed order, so that the blocks of a row their
    // bit-reversed loads touch are neighbours (conflict free)
    static constexpr bool kReverseR2 = !REORDER && RM == 8;
    static constexpr int r2_of(int m) {
        if (!kReverseR2) return m;
        int r = 0;
        for (int i = 0; i < RM_BITS; ++i) r |= ((m >> i) & 1) << (RM_BITS - 1 - i);
        return r;
    }

    int v, fft;          // position inside the FFT, FFT inside the workgroup
    int t1;              // pass-1 role
    int t2, a;           // middle role (three-pass lengths)
    int klow;            // last-pass output index: r[q3] = X[klow + T * q3]
    unsigned m0;         // LDS byte address of (plane 0, dword 64 * wave)
    // dword offsets of the thread's runs (set-up, loop invariant)
    static constexpr int kImageRuns = T >= 16 ? 1 : 16 / T;
    int off_image[kImageRuns], off_x1, off_x2[B1];
    Twiddles<N, DIR> tw;

    static constexpr bool x1_rows_regular() {
        for (int aa = 0; aa < RM; ++aa)
            for (int c = 0; c < BM; ++c)
                if (P::x1_row(aa * BM + c) - P::x1_row(aa * BM) != P::x1_row(c) - P::x1_row(0)) return false;
        return true;
    }
    __device__ static __forceinline__ int pass1_role(int pos) {
        if constexpr (!kThreePass || kRegisterX1) return pos;
        else return (pos / RM) + 16 * (kReverseR2 ? (int)(__brev((unsigned)(pos % RM)) >> (32 - RM_BITS)) : pos % RM);
    }
    __device__ static __forceinline__ int position_of_role(int role) {
        if constexpr (!kThreePass || kRegisterX1) return role;
        else return RM * (role % 16) + (kReverseR2 ? (int)(__brev((unsigned)(role / 16)) >> (32 - RM_BITS)) : role / 16);
    }

    __device__ __forceinline__ void init(int tid, const float* planes) {
        v = tid % T;
        fft = tid / T;
        t1 = pass1_role(v);
        t2 = v % 16;
        a = v / 16;
        klow = (kThreePass && kRoleOutputs) ? t1 : v;
        typedef __attribute__((address_space(3))) const float lds_float;
        m0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_float*)planes + 4u * (tid & ~63));
        tw.init(t1, t2);
        if constexpr (kRotatedX1) {
            const bool rot = (t2 >> 3) & 1;
#pragma unroll
            for (int q2 = 1; q2 < RM; ++q2) {
                const float2 w = tw.wm[q2];
                // w * (-+i)^q2   (forward: -i, inverse: +i)
                const float2 r1 = DIR ? make_float2(-w.y, w.x) : make_float2(w.y, -w.x), r3 = make_float2(-r1.x, -r1.y);
                const float2 turned = (q2 & 3) == 0 ? w : (q2 & 3) == 1 ? r1 : (q2 & 3) == 2 ? make_float2(-w.x, -w.y) : r3;
                tw.wm[q2] = rot ? turned : w;
            }
        }
        // bit-reversed load: the sixteen contiguous elements p = 16 * rho + i of the natural image, rho = rev_T(t1):
        // element p is dword fft * T + p % T of row p / T
        const int rho = (int)(__brev((unsigned)t1) >> (32 - T_BITS));
        if constexpr (kRotatedX1) {
            off_image[0] = P::image_row_at(rho / 16) + fft * T + (rho % 16);   // element 16 * (rho % 16) + i sits at the dword of role ... + i: 16 * i + rho % 16
        } else if constexpr (T >= 16) {
            off_image[0] = P::image_row_at(rho / (T / 16)) + fft * T + 16 * (rho % (T / 16));
        } else {
#pragma unroll
            for (int h = 0; h < kImageRuns; ++h) off_image[h] = P::image_row_at(kImageRuns * rho + h) + fft * T;   // T dwords of each row
        }
        if constexpr (kThreePass) {
            static_assert(x1_rows_regular(), "x1_load takes the rows a*BM + c at compile-time distances from row a*BM");
            off_x1 = kRegisterX1 ? 0 : P::template row_at<RowKind::x1>(a * BM) + fft * T + RM * t2;
            const int c = klow % BM, aa = (klow % 16) / BM, q2 = klow / 16;
            off_x2[0] = P::template row_at<RowKind::x2>(c * RM + q2) + fft * T + 16 * aa;
        } else {
            off_x1 = 0;
#pragma unroll
            for (int b = 0; b < B1; ++b) off_x2[b] = P::template row_at<RowKind::x2>(b * R1 + v) + fft * T;
        }
    }

    // ---- sixteen registers -> rows ROW(j), dword tid, both planes -------------------------------------------------
    template <int (*ROW)(int)>
    __device__ __forceinline__ void store_rows(const float2 (&r)[16]) const {
        addtid_store4<ROW(0), ROW(1), ROW(2), ROW(3), P::kPlane>(m0, r[0], r[1], r[2], r[3]);
        addtid_store4<ROW(4), ROW(5), ROW(6), ROW(7), P::kPlane>(m0, r[4], r[5], r[6], r[7]);
        addtid_store4<ROW(8), ROW(9), ROW(10), ROW(11), P::kPlane>(m0, r[8], r[9], r[10], r[11]);
        addtid_store4<ROW(12), ROW(13), ROW(14), ROW(15), P::kPlane>(m0, r[12], r[13], r[14], r[15]);
    }
    // a run of RUN (4, 8, 16: ds_read_b128; 2: ds_read_b64) contiguous dwords of both planes -> out[0 .. RUN)
    template <int RUN>
    __device__ static __forceinline__ void load_run(float2* out, const float* p) {
        if constexpr (RUN == 2) {
            const f2v re = *reinterpret_cast<const f2v*>(p);
            const f2v im = *reinterpret_cast<const f2v*>(p + P::kPlane);
            out[0] = make_float2(re[0], im[0]);
            out[1] = make_float2(re[1], im[1]);
        } else {
#pragma unroll
            for (int k = 0; k < RUN / 4; ++k) {
                const f4v re = *reinterpret_cast<const f4v*>(p + 4 * k);
                const f4v im = *reinterpret_cast<const f4v*>(p + P::kPlane + 4 * k);
#pragma unroll
                for (int m = 0; m < 4; ++m) out[4 * k + m] = make_float2(re[m], im[m]);
            }
        }
    }

    // ---- the image between applications: row c, dword (fft, position of the thread whose natural index is u) holds x[u + T*c]
    __device__ __forceinline__ void image_store(const float2 (&r)[16]) const { store_rows<P::image_row>(r); }

    // natural registers r[c] = x[role + T*c] of the thread itself (reorder: first application of a tile only)
    __device__ __forceinline__ void image_load_own(float2 (&r)[16], const float* planes) const {
        const float* p = planes + fft * T + v;
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = make_float2(p[P::image_row(c)], p[P::kPlane + P::image_row(c)]);
    }

    // pass-1 slot b*R1 + r1 holds x'[t1 + T*b + T1*r1]; without reorder x' = x o bitrev, which makes it element
    // i = rev_B1(b) * R1 + rev_R1(r1) of the thread's sixteen contiguous ones
    static constexpr int slot_of_element(int i) {
        const int hb = i / R1, lr = i % R1;
        int b = 0, r1 = 0;
        for (int k = 0; k < B1_BITS; ++k) b |= ((hb >> k) & 1) << (B1_BITS - 1 - k);
        for (int k = 0; k < R1_BITS; ++k) r1 |= ((lr >> k) & 1) << (R1_BITS - 1 - k);
        return b * R1 + r1;
    }
    __device__ __forceinline__ void image_load_bitrev(float2 (&r)[16], const float* planes) const {
        float2 e[16];
        if constexpr (kRotatedX1) {
            const float* p = planes + off_image[0];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = make_float2(p[16 * i], p[P::kPlane + 16 * i]);
        } else if constexpr (T >= 16) {
            load_run<16>(e, planes + off_image[0]);
        } else {
#pragma unroll
            for (int h = 0; h < kImageRuns; ++h) load_run<T>(&e[T * h], planes + off_image[h]);   // T = 8: rows 2 * rho, 2 * rho + 1
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r[slot_of_element(i)] = e[i];
    }
    // reorder: natural registers r[c] = x[t1 + T*c] -> pass-1 slots r[b*R1 + r1] = x[t1 + T*b + T1*r1] (c = b + B1*r1): a renaming
    __device__ static __forceinline__ void natural_to_slots(float2 (&r)[16]) {
        if constexpr (B1 > 1) {
            float2 t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = r[c];
#pragma unroll
            for (int b = 0; b < B1; ++b)
#pragma unroll
                for (int r1 = 0; r1 < R1; ++r1) r[b * R1 + r1] = t[b + B1 * r1];
        }
    }

    // ---- pass 1: B1 radix-R1 butterflies, then W_N^{(t1 + T*b) * q1} -----------------------------------------------
    __device__ __forceinline__ void pass1(float2 (&r)[16]) const {
#pragma unroll
        for (int b = 0; b < B1; ++b) {
            float2 y[R1];
            SmallDft<R1, 1, DIR>::run(&r[b * R1], y);
            r[b * R1] = y[0];
#pragma unroll
            for (int q1 = 1; q1 < R1; ++q1) r[b * R1 + q1] = cmul(y[q1], tw.w1[b * R1 + q1]);
        }
    }

    // ---- exchange 1 (three-pass): (t1, q1) in row q1 at the dword of the thread with role t1 -------------------------
    __device__ __forceinline__ void x1_store(const float2 (&r)[16]) const { store_rows<P::x1_row>(r); }
    // middle thread (t2, a): r[c*RM + r2] = element (t2 + 16*r2, q1 = a*BM + c) = row a*BM + c, dwords RM*t2 + m, r2 = r2_of(m)
    __device__ __forceinline__ void x1_load(float2 (&r)[16], const float* planes) const {
        const float* p = planes + off_x1;
        if constexpr (kRotatedX1) {
            // quad (k + rot) % 4 of the run at step k: p0 + 4k for k < 3, and for k = 3 the quad that is left
            const int rot = (t2 >> 3) & 1;
            const float* p0 = p + 4 * rot;
            const float* p3 = p + (rot ? 0 : 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* q = (k < 3) ? p0 + 4 * k : p3;
                const f4v re = *reinterpret_cast<const f4v*>(q);
                const f4v im = *reinterpret_cast<const f4v*>(q + P::kPlane);
#pragma unroll
                for (int m = 0; m < 4; ++m) r[4 * k + m] = make_float2(re[m], im[m]);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < BM; ++c) {
            float2 e[RM];
            // rows a*BM + c for c < BM are consecutive in the placement (same residue class order), so their distance is constant
            load_run<RM>(e, p + (P::x1_row(c) - P::x1_row(0)));
#pragma unroll
            for (int m = 0; m < RM; ++m) r[c * RM + r2_of(m)] = e[m];
        }
    }

    // ---- middle pass: BM radix-RM butterflies over r2, then W_{T1}^{t2*q2} -----------------------------------------
    __device__ __forceinline__ void middle(float2 (&r)[16]) const {
#pragma unroll
        for (int c = 0; c < BM; ++c) {
            float2 y[RM];
            SmallDft<RM, 1, DIR>::run(&r[c * RM], y);
            r[c * RM] = y[0];
#pragma unroll
            for (int q2 = 1; q2 < RM; ++q2) r[c * RM + q2] = cmul(y[q2], tw.wm[q2]);
        }
    }

    // ---- exchange in front of the last pass ---------------------------------------------------------------------------
    // three-pass: register j = c*RM + q2 of thread (t2, a) is element (t2, klow = a*BM + c + 16*q2); the last-pass thread klow
    //             reads t2 = 0..15 = dwords 16*a + t2 of row c*RM + q2  (c = klow % BM, a = (klow % 16) / BM, q2 = klow / 16)
    // two-pass:   register b*R1 + q1 of thread t1 is element (n1 = t1 + T*b, q1); the last-pass thread q1 = v reads
    //             n1 = 0..15 = dwords t1 of rows b*R1 + v
    __device__ __forceinline__ void x2_store(const float2 (&r)[16]) const { store_rows<P::x2_row>(r); }
    __device__ __forceinline__ void x2_load(float2 (&x)[16], const float* planes) const {
        if constexpr (kThreePass || B1 == 1) {
            load_run<16>(x, planes + off_x2[0]);
        } else {
#pragma unroll
            for (int b = 0; b < B1; ++b) load_run<T>(&x[T * b], planes + off_x2[b]);
        }
    }

    // ---- one application: pass-1 slots in r -> natural result r[q3] = X[klow + T*q3] --------------------------------
    __device__ __forceinline__ void transform_from_pass1_slots(float2 (&r)[16], float* planes) const {
        pass1(r);
        planar_sync<G::kMultiWave>();       // every read of the previous image is done -- and every STORE of it: the rows of the image and of exchange 1
                                            // start at different residues, so near a wave boundary this wave's x1 dwords are another wave's image dwords
                                            // (dropping this in the fused chain, where nobody reads the image, broke N = 2048: a late image store landed on x1 data)
        if constexpr (kRegisterX1) {
            Engine<N, DIR, 1>::exchange1_registers_static(r);
            middle(r);
        } else if constexpr (kThreePass) {
            x1_store(r);
            planar_sync<G::kMultiWave>();
            x1_load(r, planes);
            middle(r);
            planar_sync<G::kMultiWave>();
        }
        x2_store(r);
        planar_sync<G::kMultiWave>();
        float2 x[16];
        x2_load(x, planes);
        SmallDft<16, 1, DIR>::run(x, r);
    }

    // ---- R2C / C2R (real length 2L through this complex engine of length L = N; reorder roles) ----------------------
    // The split (after the forward transform) / merge (in front of the inverse one) of RC:269-344 on the natural REGISTERS
    // r[q] = x[i], i = klow + T*q:   out[i] = S/2 + V * D,  S = A + conj(B), D = A - conj(B), A = x[i], B = x[L - i],
    // V = (-+i/2) * W_2L^i = herm_w * W_32^q  (HermitianRegisters::apply in smfft_kernels.hpp states the algebra).  The
    // partner x[L - i] = x[(T - klow) + T*(15 - q)] is register 15 - q of the thread whose role is T - klow: it is read
    // from the stored image (row 15 - q, that thread's dword); role 0 pairs with its own registers, and its element 0
    // packs DC and Nyquist (RC:280-286, 332-339).  Precondition: the image holds the registers of every thread.
    float2 herm_w;
    int off_partner;
    bool herm_first;
    __device__ __forceinline__ void init_hermitian() {
        static_assert(REORDER, "the real transforms are natural order");
        herm_first = (klow == 0);
        off_partner = fft * T + position_of_role((T - klow) % T);
        const float2 w = twiddle<DIR>(klow * (4096 / (2 * N)));
        herm_w = DIR ? make_float2(-0.5f * w.y, 0.5f * w.x) : make_float2(0.5f * w.y, -0.5f * w.x);
    }
    // PAIR-WISE: out[i] = S/2 + V*D and out[L - i] = conj(S - out[i]) come from ONE S, D and V: 16 instructions per pair (two separate
    // evaluations took 28: round 3's form, profiles/r04_ab_rc_pairs.txt).
    // A thread takes the pairs of its registers q = 0..7 (i = klow + T*q); the partner x[L - i] is register 15 - q of the
    // thread with role T - klow, so the registers 8..15 of every thread are somebody else's second halves:
    //   precondition  rows 8..15 of the image hold every thread's registers 8..15 (image_store_upper, or the loaded tile);
    //   1  fetch the eight partners (row 15 - q at the partner's dword), combine: r[q] = out[i]; other[q] = out[L - i];
    //   2  store other[15 - j] lane-linearly into row j (j = 8..15), sync, read row j back at the PARTNER's dword into r[j].
    // Role 0 is its own partner: its register q pairs with its OWN register 16 - q (q = 1..7), register 0 packs DC and
    // Nyquist and register 8 (i = L/2, its own partner) becomes its conjugate -- so that thread takes B from its registers
    // and stores row j <- other[16 - j] (row 8 <- conj(r[8])): thirty-two selects, no second address.
    // LDS traffic per application: 16 + 16 dword stores, 16 + 16 dword reads -- what image_store + hermitian_apply move.
    __device__ __forceinline__ void image_store_upper(const float2 (&r)[16]) const {
        addtid_store4<P::image_row(8), P::image_row(9), P::image_row(10), P::image_row(11), P::kPlane>(m0, r[8], r[9], r[10], r[11]);
        addtid_store4<P::image_row(12), P::image_row(13), P::image_row(14), P::image_row(15), P::kPlane>(m0, r[12], r[13], r[14], r[15]);
    }
    __device__ __forceinline__ void hermitian_apply_pairs(float2 (&r)[16], const float* planes) const {
        constexpr float c32[8] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f,
                                  0.38268343236508984f, 0.19509032201612833f};
        constexpr float s32[8] = {0.f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f,
                                  0.92387953251128674f, 0.98078528040323043f};
        const float* p = planes + off_partner;
        float2 w = herm_w;
        // The seven products W_2L^i = herm_w * W_32^q are loop invariant: L <= 512 keeps them (14 registers: 108 / 112 -> 122 / 126 of
        // the 128 that four waves per SIMD allow; R2C / C2R of real N = 512 / 1024 -2.5...-3 %, profiles/r05_ab_rc_keep.txt), the longer
        // lengths have no room and recompute them per application behind an opaque copy of herm_w.
        if constexpr (N > 512) asm volatile("" : "+v"(w.x), "+v"(w.y));
        float2 other[8];
        const float2 r8 = r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if ((q & 1) == 0) asm volatile("" ::: "memory");   // two partners in flight at a time (the kernels live on their occupancy: all at once took 139-157 registers = 3 waves per SIMD)
            const float2 fetched = make_float2(p[P::image_row(15 - q)], p[P::kPlane + P::image_row(15 - q)]);
            const float2 own = q == 0 ? r8 : r[16 - q];
            const float2 A = r[q], B = herm_first ? own : fetched;
            const float2 S = make_float2(A.x + B.x, A.y - B.y);
            const float2 D = make_float2(A.x - B.x, A.y + B.y);
            const float2 V = (q == 0) ? w : cmul(w, make_float2(c32[q], DIR ? s32[q] : -s32[q]));
            const float2 out = make_float2(fmaf(V.x, D.x, fmaf(-V.y, D.y, 0.5f * S.x)), fmaf(V.x, D.y, fmaf(V.y, D.x, 0.5f * S.y)));
            other[q] = make_float2(S.x - out.x, out.y - S.y);          // conj(S - out)
            if (q == 0) {
                const float2 packed = DIR ? make_float2(0.5f * (A.x + A.y), 0.5f * (A.x - A.y)) : make_float2(A.x + A.y, A.x - A.y);
                r[0] = herm_first ? packed : out;
                other[0] = herm_first ? make_float2(r8.x, -r8.y) : other[0];
            } else {
                r[q] = out;
            }
        }
        planar_sync<G::kMultiWave>();                // every partner is fetched before the rows are written over
        float2 st[8];
#pragma unroll
        for (int j = 8; j < 16; ++j) st[j - 8] = herm_first ? other[j == 8 ? 0 : 16 - j] : other[15 - j];
        addtid_store4<P::image_row(8), P::image_row(9), P::image_row(10), P::image_row(11), P::kPlane>(m0, st[0], st[1], st[2], st[3]);
        addtid_store4<P::image_row(12), P::image_row(13), P::image_row(14), P::image_row(15), P::kPlane>(m0, st[4], st[5], st[6], st[7]);
        planar_sync<G::kMultiWave>();
#pragma unroll
        for (int j = 8; j < 16; ++j) r[j] = make_float2(p[P::image_row(j)], p[P::kPlane + P::image_row(j)]);
    }
};

// tile <-> planar image (once per tile): element e = fft * N + n of the tile, n = u + T*c, lies in row c at dword
// fft * T + position(u); position(u) = u, or the position of the thread whose pass-1 role is u (reorder)
// (g is not __restrict__: a resumed piece of a cut chain reads what another workgroup of the same launch stored)
template <int N, int DIR, int REORDER>
__device__ __forceinline__ float* plane_word_of(float* planes, int e) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    const int f = e / N, n = e % N, u = n % E::T, row = n / E::T;
    const int pos = E::kForward ? E::position_of_role(u) : u;
    return planes + P::image_row_at(row) + f * E::T + pos;
}
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void tile_to_planes(const float2* g, float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int TW = E::TW;
    // (opaque copy of the thread index: the sixteen addresses below are computed where they are used, once per chain, instead
    //  of being hoisted out of the loop over chains and kept in registers across the applications)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    // (N = 4096: eight elements at a time.  The kernel sits at the 128-register cap of four waves per SIMD with 60 registers of
    //  twiddles alive across this copy, and sixteen values in flight spilled 44-52 bytes per lane to scratch -- outside the application
    //  loop, but a private segment all the same; a copy is once per hundred applications)
    constexpr int kBatch = N >= 2048 ? 8 : 16;
    const bool full = first_fft + P::F <= limit_fft;
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += kBatch) {
        float2 val[kBatch];
#pragma unroll
        for (int c = 0; c < kBatch; ++c) {
            const int e = tid + TW * (c0 + c);
            const bool ok = full || (first_fft + e / N < limit_fft);
            const float2 t = g[ok ? e : 0];
            val[c] = ok ? t : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < kBatch; ++c) {
            float* p = plane_word_of<N, DIR, REORDER>(planes, tid + TW * (c0 + c));
            p[0] = val[c].x;
            p[P::kPlane] = val[c].y;
        }
        if (kBatch < 16) asm volatile("" ::: "memory");      // the batches stay apart
    }
}
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void planes_to_tile(float2* g, const float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int TW = E::TW;
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const bool full = first_fft + P::F <= limit_fft;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = tid + TW * c;
        const float* p = plane_word_of<N, DIR, REORDER>(const_cast<float*>(planes), e);
        const float2 t = make_float2(p[0], p[P::kPlane]);
        if (full || first_fft + e / N < limit_fft) g[e] = t;
    }
}
// The same two copies for a tile that changes hands inside the launch (SharedTile: write-through stores, sc1 loads, 16 bytes per
// lane: elements 2 * tid, 2 * tid + 1 of every 2 * TW)
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void shared_tile_to_planes(const float2* g, float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int TW = E::TW;
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const long ffts = limit_fft - first_fft < P::F ? limit_fft - first_fft : P::F;
    const SharedTile tile(g, ffts * N * 8);
    float2 val[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) tile.load2(2 * tid + 2 * TW * c, val[2 * c], val[2 * c + 1]);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float* p = plane_word_of<N, DIR, REORDER>(planes, 2 * tid + (c & 1) + 2 * TW * (c >> 1));
        p[0] = val[c].x;
        p[P::kPlane] = val[c].y;
    }
}
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void planes_to_shared_tile(float2* g, const float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int TW = E::TW;
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const long ffts = limit_fft - first_fft < P::F ? limit_fft - first_fft : P::F;
    const SharedTile tile(g, ffts * N * 8);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int e = 2 * tid + 2 * TW * c;
        const float* p = plane_word_of<N, DIR, REORDER>(const_cast<float*>(planes), e);
        const float* q = plane_word_of<N, DIR, REORDER>(const_cast<float*>(planes), e + 1);
        tile.store2(e, make_float2(p[0], p[P::kPlane]), make_float2(q[0], q[P::kPlane]));
    }
}

}  // namespace smfft
