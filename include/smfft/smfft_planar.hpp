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
//     are bank-conflict free (tools/soa_model.py emulates the choreography and searches the shifts).
//   * With exchanges this cheap, exchange 1 of N = 512 / 1024 goes through LDS as well: the 32 v_permlane*_swap of the
//     register form were 124 ns of the 718 ns an N = 1024 application took per SIMD.
//
// Thread positions and roles.  A compact workgroup has TW = max(64, T) threads, T = N / 16 per FFT; thread tid is at
// position v = tid % T of FFT tid / T, and every store puts register j at dword tid of row j.
//   three-pass sizes (N >= 512, T = 16 * RM):
//     pass 1      role t1 = t2 + 16 * r2 with v = RM * t2 + r2       (so that the middle pass reads RM contiguous dwords)
//     middle      role (t2, a) with v = 16 * a + t2                   (so that the last pass reads 16 contiguous dwords)
//     last        output index klow = v (no reorder: the stored result is lane linear in klow, which the bit-reversed
//                 read of the next application needs) or klow = pass-1 role of v (reorder: the registers a thread ends
//                 with are the ones it starts the next application with -- forwarded, never re-loaded)
//   two-pass sizes (N = 128, 256): pass-1 role t1 = v, last pass q1 = v.
#pragma once
#include "smfft_engine.hpp"

namespace smfft {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// ds_write_addtid_b32 x 8: four elements (re, im).  M0 holds the wave's base (plane 0 + 4 * 64 * wave); an SALU write of
// M0 needs one wait state before an LDS "add-TID" instruction reads it (the assembler does not see into inline assembly).
template <int O0, int O1, int O2, int O3, int P>
__device__ __forceinline__ void addtid_store4(unsigned m0, float2 a, float2 b, float2 c, float2 d) {
    asm volatile(
        "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %0 offset:%9\n\tds_write_addtid_b32 %1 offset:%10\n\t"
        "ds_write_addtid_b32 %2 offset:%11\n\tds_write_addtid_b32 %3 offset:%12\n\t"
        "ds_write_addtid_b32 %4 offset:%13\n\tds_write_addtid_b32 %5 offset:%14\n\t"
        "ds_write_addtid_b32 %6 offset:%15\n\tds_write_addtid_b32 %7 offset:%16"
        :
        : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(c.x), "v"(c.y), "v"(d.x), "v"(d.y), "s"(m0),
          "n"(4 * O0), "n"(4 * (O0 + P)), "n"(4 * O1), "n"(4 * (O1 + P)), "n"(4 * O2), "n"(4 * (O2 + P)), "n"(4 * O3), "n"(4 * (O3 + P))
        : "memory");
}

template <int N, int REORDER>
struct PlanarGeometry {
    using G = Geometry<N>;
    static constexpr int T = G::T, RM = G::RM, BM = G::BM, R1 = G::R1, B1 = G::B1;
    static constexpr int TW = G::kCompactThreads;
    static constexpr int F = TW / T;
    // Row bases (dwords inside a plane; closed forms found by tools/soa_model.py, every read below conflict free).
    // Rows are TW dwords long; the shifts are multiples of 4 dwords (ds_read_b128 needs 16-byte alignment).
    static constexpr int image_row(int j) { return TW * j + 4 * ((j >> 1) & 3) + 16 * (j >> 3); }
    static constexpr int x1_row(int j) { return TW * j; }
    // no reorder: klow = position; reorder: klow = pass-1 role
    static constexpr int x2_row(int j) { return REORDER ? TW * j + 4 * (j & 3) + 32 * (j >> 2) : TW * j + 4 * (j >> 2); }
    static constexpr int span(int (*row)(int)) {
        int m = 0;
        for (int j = 0; j < 16; ++j) m = row(j) + TW > m ? row(j) + TW : m;
        return m;
    }
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int kPlane = max3(span(image_row), span(x1_row), span(x2_row));   // dwords per plane (a multiple of 4)
    static constexpr int kLdsFloats = 2 * kPlane;
};

template <int N, int DIR, int REORDER>
struct PlanarEngine {
    using G = Geometry<N>;
    using P = PlanarGeometry<N, REORDER>;
    static constexpr int T = G::T, RM = G::RM, BM = G::BM, R1 = G::R1, B1 = G::B1, TW = P::TW;
    static constexpr int T_BITS = ilog2c(T);
    static_assert(RM > 1, "three-pass sizes only (so far)");
    static constexpr bool kForward = REORDER;   // last-pass thread v computes klow = pass-1 role of v

    int v, fft;          // position inside the FFT, FFT inside the workgroup
    int t1;              // pass-1 role
    int t2, a;           // middle role
    int klow;            // last-pass output index: r[q3] = X[klow + T * q3]
    unsigned m0;         // LDS byte address of (plane 0, dword 64 * wave)
    Twiddles<N, DIR> tw;

    __device__ static __forceinline__ int pass1_role(int pos) { return (pos / RM) + 16 * (pos % RM); }
    __device__ static __forceinline__ int position_of_role(int role) { return RM * (role % 16) + role / 16; }

    __device__ __forceinline__ void init(int tid, const float* planes) {
        v = tid % T;
        fft = tid / T;
        t1 = pass1_role(v);
        t2 = v % 16;
        a = v / 16;
        klow = kForward ? t1 : v;
        typedef __attribute__((address_space(3))) const float lds_float;
        m0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_float*)planes + 4u * (tid & ~63));
        tw.init(t1, t2);
    }

    // ---- sixteen registers -> rows ROW(j), dword tid, both planes -------------------------------------------------
    template <int (*ROW)(int)>
    __device__ __forceinline__ void store_rows(const float2 (&r)[16]) const {
        addtid_store4<ROW(0), ROW(1), ROW(2), ROW(3), P::kPlane>(m0, r[0], r[1], r[2], r[3]);
        addtid_store4<ROW(4), ROW(5), ROW(6), ROW(7), P::kPlane>(m0, r[4], r[5], r[6], r[7]);
        addtid_store4<ROW(8), ROW(9), ROW(10), ROW(11), P::kPlane>(m0, r[8], r[9], r[10], r[11]);
        addtid_store4<ROW(12), ROW(13), ROW(14), ROW(15), P::kPlane>(m0, r[12], r[13], r[14], r[15]);
    }

    // ---- the image between applications: row c, dword (fft, position of the thread whose natural index is u) holds x[u + T*c]
    __device__ __forceinline__ void image_store(const float2 (&r)[16]) const { store_rows<P::image_row>(r); }

    // natural registers r[c] = x[role + T*c] of the thread itself (reorder: first application of a tile only)
    __device__ __forceinline__ void image_load_own(float2 (&r)[16], const float* planes) const {
        const float* p = planes + fft * T + v;
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = make_float2(p[P::image_row(c)], p[P::kPlane + P::image_row(c)]);
    }

    // no reorder: the thread with pass-1 role t1 needs x[bitrev(t1 + T1*r1)] = element 16 * rho + rev4(r1), rho = rev_T(t1):
    // sixteen contiguous elements p = 16 * rho + i of the natural image = dwords 16 * (rho % (T/16)) + i of row rho / (T/16)
    __device__ __forceinline__ void image_load_bitrev(float2 (&r)[16], const float* planes) const {
        static_assert(T >= 16, "");
        const int rho = (int)(__brev((unsigned)t1) >> (32 - T_BITS));
        const int row = rho / (T / 16);
        const float* p = planes + P::image_row(row) + fft * T + 16 * (rho % (T / 16));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f4v re = *reinterpret_cast<const f4v*>(p + 4 * k);
            const f4v im = *reinterpret_cast<const f4v*>(p + P::kPlane + 4 * k);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                r[rev4(4 * k + m)] = make_float2(re[m], im[m]);
            }
        }
    }
    static constexpr int rev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

    // ---- pass 1 (registers hold the pass-1 slots r[r1] = x'[t1 + T1*r1]) --------------------------------------------
    __device__ __forceinline__ void pass1(float2 (&r)[16]) const {
        float2 y[16];
        SmallDft<16, 1, DIR>::run(r, y);
        r[0] = y[0];
#pragma unroll
        for (int q1 = 1; q1 < 16; ++q1) r[q1] = cmul(y[q1], tw.w1[q1]);
    }

    // ---- exchange 1: (t1, q1) in row q1 at the dword of the thread with role t1 ------------------------------------
    __device__ __forceinline__ void x1_store(const float2 (&r)[16]) const { store_rows<P::x1_row>(r); }
    // middle thread (t2, a): r[c*RM + r2] = element (t2 + 16*r2, q1 = a*BM + c) = row a*BM + c, dwords RM*t2 + r2
    __device__ __forceinline__ void x1_load(float2 (&r)[16], const float* planes) const {
        const float* p = planes + P::x1_row(a * BM) + fft * T + RM * t2;     // x1_row is linear in the row index
        static_assert(P::x1_row(5) - P::x1_row(4) == TW && P::x1_row(0) == 0, "");
#pragma unroll
        for (int c = 0; c < BM; ++c) {
            if constexpr (RM == 2) {
                const f2v re = *reinterpret_cast<const f2v*>(p + TW * c);
                const f2v im = *reinterpret_cast<const f2v*>(p + P::kPlane + TW * c);
                r[c * RM + 0] = make_float2(re[0], im[0]);
                r[c * RM + 1] = make_float2(re[1], im[1]);
            } else {
#pragma unroll
                for (int k = 0; k < RM / 4; ++k) {
                    const f4v re = *reinterpret_cast<const f4v*>(p + TW * c + 4 * k);
                    const f4v im = *reinterpret_cast<const f4v*>(p + P::kPlane + TW * c + 4 * k);
#pragma unroll
                    for (int m = 0; m < 4; ++m) r[c * RM + 4 * k + m] = make_float2(re[m], im[m]);
                }
            }
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

    // ---- exchange 2: register j = c*RM + q2 of thread (t2, a) is element (t2, klow = a*BM + c + 16*q2) --------------
    __device__ __forceinline__ void x2_store(const float2 (&r)[16]) const {
        store_rows<P::x2_row>(r);
    }
    // last-pass thread klow reads t2 = 0..15: row c*RM + q2, dwords 16*a + t2  (c = klow % BM, a = (klow % 16) / BM, q2 = klow / 16)
    __device__ __forceinline__ void x2_load(float2 (&x)[16], const float* planes) const {
        const int c = klow % BM, aa = (klow % 16) / BM, q2 = klow / 16;
        const int j = c * RM + q2;
        const float* p = planes + P::x2_row(j) + fft * T + 16 * aa;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f4v re = *reinterpret_cast<const f4v*>(p + 4 * k);
            const f4v im = *reinterpret_cast<const f4v*>(p + P::kPlane + 4 * k);
#pragma unroll
            for (int m = 0; m < 4; ++m) x[4 * k + m] = make_float2(re[m], im[m]);
        }
    }

    // ---- one application: pass-1 slots in r -> natural result r[q3] = X[klow + T*q3] --------------------------------
    __device__ __forceinline__ void transform_from_pass1_slots(float2 (&r)[16], float* planes) const {
        pass1(r);
        fft_sync<G::kMultiWave>();       // every read of the previous image is done
        x1_store(r);
        fft_sync<G::kMultiWave>();
        x1_load(r, planes);
        middle(r);
        fft_sync<G::kMultiWave>();
        x2_store(r);
        fft_sync<G::kMultiWave>();
        float2 x[16];
        x2_load(x, planes);
        SmallDft<16, 1, DIR>::run(x, r);
    }
};

// tile <-> planar image (once per tile): element e = fft * N + n of the tile, n = u + T*c, lies in row c at dword
// fft * T + position(u); position(u) = u, or the position of the thread whose pass-1 role is u (reorder)
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void tile_to_planes(const float2* __restrict__ g, float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int T = E::T, TW = E::TW;
    float2 val[16];
    const bool full = first_fft + P::F <= limit_fft;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = threadIdx.x + TW * c;
        const bool ok = full || (first_fft + e / N < limit_fft);
        const float2 t = g[ok ? e : 0];
        val[c] = ok ? t : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = threadIdx.x + TW * c;
        const int f = e / N, n = e % N, u = n % T, row = n / T;
        const int pos = E::kForward ? E::position_of_role(u) : u;
        float* p = planes + P::image_row(row) + f * T + pos;
        p[0] = val[c].x;
        p[P::kPlane] = val[c].y;
    }
}
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void planes_to_tile(float2* __restrict__ g, const float* planes, long first_fft, long limit_fft) {
    using E = PlanarEngine<N, DIR, REORDER>;
    using P = PlanarGeometry<N, REORDER>;
    constexpr int T = E::T, TW = E::TW;
    const bool full = first_fft + P::F <= limit_fft;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = threadIdx.x + TW * c;
        const int f = e / N, n = e % N, u = n % T, row = n / T;
        const int pos = E::kForward ? E::position_of_role(u) : u;
        const float* p = planes + P::image_row(row) + f * T + pos;
        const float2 t = make_float2(p[0], p[P::kPlane]);
        if (full || first_fft + f < limit_fft) g[e] = t;
    }
}

}  // namespace smfft
