// smfft_engine.hpp -- the gfx950 shared-memory (LDS) FFT engine and the device-function surface.
//
// What it replaces (reference = KAdamek/SMFFT, CUDA):
//   do_SMFFT_CT_DIT<P>           SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu:334-532  (+ reorder_* :54-329)
//   do_FFT_Stockham_mk6<P>       SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu:97-240
//   do_FFT_Stockham_C2C<P,D>     SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:106-266
//   do_FFT_Stockham_R2C_C2R<P,D> SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:269-344
// Same names, same results (semantics S1..S6 of SURVEY.md 8(a)); the algorithm is NOT the
// reference's 4-elements-per-thread radix-2 ladder.  It is designed for the 64-lane wave:
//
//   * 16 elements per thread, N/16 threads per FFT: an FFT of N <= 1024 lives inside ONE wave,
//     so its LDS exchanges need no s_barrier at all (a wave's DS instructions execute in order);
//     N = 2048 / 4096 use 2 / 4 waves and workgroup barriers.
//   * N = R1 * RM * 16.  Pass 1 (radix R1) and the last pass (radix 16) run entirely in
//     registers, N >= 512 adds a middle radix-RM pass: 2 LDS exchanges for N >= 512, 1 below,
//     against the reference's log2(N)-5 shared-memory stages + a 3-sync reorder.
//   * The exchanges are a Stockham autosort: the digit reversal is folded into the LDS
//     addressing, so "reorder" costs nothing.  Layouts are padded so that every ds_write_b64 is
//     bank-conflict free (16-lane groups, float2 index mod 16) and every ds_read_b64 too
//     (32-lane groups, mod 32); each FFT owns a region of 17N/16 float2 (the "33-stride" idea of
//     the reference, re-derived for 64 banks / 64 lanes; tools/plan_model.py checks it).
//   * Twiddles W_N^m come from a 4096-entry table rounded once from fp64 (<= 0.5 ulp); a thread
//     needs only 15 + (RM-1) of them, loaded once and kept in VGPRs across a persistent loop.
//
// This header is the engine only.  The device functions under the reference's names (do_SMFFT_CT_DIT,
// do_FFT_Stockham_*), in the reference's own launch shape and in the engine's tiled shape, are in
// smfft_device_functions.hpp; include/smfft_device.hpp pulls in everything a user kernel needs.
//
// LDS layout the engine works on: every FFT owns a REGION of Geometry::SF = 17N/16 float2; its data sit at
// region[0 .. N) in natural order before and after a transform, the rest of the region is exchange space.
#pragma once
#include <hip/hip_runtime.h>
#include "SM_FFT_parameters.hpp"

namespace smfft {

static __device__ const float2 twiddle_4096[4096] = {
#include "smfft_twiddles.inc"
};
// the same values for compile-time use (TwiddleRows below)
struct alignas(8) TwiddleValue { float x, y; };
static constexpr TwiddleValue twiddle_values[4096] = {
#include "smfft_twiddles.inc"
};

// ------------------------------------------------------------------------------------------------
template <int N>
struct Geometry {
    static_assert(N >= 32 && N <= 4096 && (N & (N - 1)) == 0, "N must be a power of two in [32, 4096]");
    static constexpr int T = N / 16;                      // threads per FFT
    static constexpr int R1 = (N <= 256) ? N / 16 : 16;   // first-pass radix
    static constexpr int RM = (N <= 256) ? 1 : N / 256;   // middle-pass radix (1 = no middle pass)
    static constexpr int T1 = N / R1;                     // butterflies in pass 1
    static constexpr int B1 = 16 / R1;                    // pass-1 butterflies per thread
    static constexpr int BM = 16 / RM;                    // middle butterflies per thread
    static constexpr int S1 = T1 + T1 / 16;             // row stride of exchange 1 (q1-major)
    static constexpr int S2 = T + 1;                      // row stride of the last layout (t-major)
    static constexpr int S0 = 17;                          // row stride of the two-pass sizes' only layout
    static constexpr int SF = 17 * (N / 16);               // LDS region of one FFT (float2)
    static constexpr bool kMultiWave = (T > 64);
    // N = 512 / 1024: exchange 1 is a transpose between the lane's row bits (lane >> 4) and the top
    // register-index bits, done in registers with v_permlane16_swap / v_permlane32_swap: no LDS.
    static constexpr bool kRegExchange1 = (RM == 2 || RM == 4);
    // N <= 64 (two passes, an FFT is 2 or 4 lanes of one 16-lane row): the exchange between the passes -- and
    // the bit-reversal transposition of the no-reorder variants -- are transposes between the FFT's lane bits
    // and as many register-index bits, done in registers with DPP row operations: no LDS at all inside the
    // transform (measured on the in-LDS path: DESIGN.md section 5).
    static constexpr bool kRegTwoPass = (RM == 1) && (N <= 64);
    static constexpr int kFftsPerBlock = 4096 / N;        // tiled kernels: 256 threads own 4096 elements
    // compact kernels (the in-LDS `multiple` path): the smallest workgroup that holds whole FFTs --
    // one wave and 1024 elements for N <= 1024, N / 16 threads and one FFT above
    static constexpr int kCompactThreads = T < 64 ? 64 : T;
    static constexpr int kCompactTile = N < 1024 ? 1024 : N;
    static constexpr int kCompactFfts = kCompactTile / N;
    static constexpr int kCompactLds = (kCompactTile / N) * SF;
    // no-reorder variants keep their data in LDS with one pad per 2^kPadShift elements (p -> p + (p >> kPadShift)),
    // the layout from which a thread can read its 16 bit-reversal-contiguous elements without bank conflicts
    // (N = 1024: roles must equal lanes for the register exchange 1, so the rows a read group touches are the
    // even or the odd ones: one pad per 32 elements makes those 32 rows distinct mod 32 as well)
    static constexpr int kPadShift = (N == 1024) ? 5 : 4;
};

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
// the same product with its rounding WRITTEN DOWN (one rounded product, one fused multiply-add per component): two inlined copies of
// cmul need not round alike -- the compiler contracts a*b - c*d either way round -- and the lane engines below run the same source
// in several copies whose results must agree to the bit (their radix-16 transform is the tangent form of SmallDft, which is explicit
// fused multiply-adds throughout; SmallDft<..., false, FIXED> is the same remedy for the plain form)
__device__ __forceinline__ float2 cmul_fixed(float2 a, float2 w) {
    return make_float2(__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Global-memory accessors of the streaming kernels.  SMFFT_NT=1 marks them non-temporal (`nt`):
// every input byte is read once and every output byte written once, so nothing is worth keeping
// in L2/MALL (copy ceiling on MI355X at 4 GiB: 5.36 TB/s plain vs 5.61 TB/s nt, tools/microbench).
#ifndef SMFFT_NT
#define SMFFT_NT 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 gload(const float2* p) {
#if SMFFT_NT
    v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void gstore(float2* p, float2 a) {
#if SMFFT_NT
    v2f v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(p));
#else
    *p = a;
#endif
}

// Tile accesses of the in-LDS (`multiple`) kernels.  A tile that is handed from one workgroup of the launch to another (the
// balanced schedule parks a cut chain in its slot of d_output) is stored WRITE-THROUGH (`sc1`: the bytes leave the storing XCD's
// L2, whose lines no other XCD can see) and loaded with `sc1` loads (never served by the loading CU's L1), so the hand-over needs
// no cache-wide write-back or invalidate -- a workgroup's `buffer_wbl2` cost the README launches 10-90 us
// (profiles/r05_handover_forms.txt; MI355X_MICROARCH.md, inter-workgroup visibility: valid forms).
typedef __attribute__((address_space(1))) unsigned long long global_u64;
typedef __attribute__((address_space(1))) unsigned global_u32;
// Such a tile as a raw buffer: 16 bytes per lane and instruction with the `sc1` bit (aux = 16) -- an 8-byte sc1 store is a fabric
// write of its own and costs 2.7 times the bytes of a 16-byte one (MI355X_MICROARCH.md).  num_records = the bytes of the tile that
// belong to the batch: accesses beyond them are dropped (stores) or return zero (loads), which is exactly what a ragged last tile
// needs.  `tile` must be wave-uniform.
typedef unsigned tile_quad __attribute__((ext_vector_type(4)));
struct SharedTile {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ SharedTile(const void* tile, long valid_bytes) : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile), 0, (int)valid_bytes, 0x00020000)) {}
    // elements e, e + 1 of the tile (e even)
    __device__ __forceinline__ void load2(int e, float2& a, float2& b) const {
        const tile_quad w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 8 * e, 0, 16);
        a = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
        b = make_float2(__uint_as_float(w[2]), __uint_as_float(w[3]));
    }
    __device__ __forceinline__ void store2(int e, float2 a, float2 b) const {
        const tile_quad w = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y)};
        __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, 8 * e, 0, 16);
    }
};

// W_N^m (forward sign) or its conjugate (inverse), from the fp64-rounded table.
// (v_cos_f32 / v_sin_f32 on the exact fraction m/4096 instead of the table was tried in round 2: no memory access, but a
//  maximum error of 1.2e-7 instead of 0.5 ulp -- profiles/HISTORY.md)
template <int DIR>
__device__ __forceinline__ float2 twiddle(int m_times_4096_over_N) {
    float2 w = twiddle_4096[m_times_4096_over_N & 4095];
    if (DIR) w.y = -w.y;
    return w;
}

// Synchronisation between the threads that share one FFT: nothing but a compiler fence when the
// FFT lives in one wave (DS operations of a wave are executed in issue order), a workgroup
// barrier otherwise.
template <bool MULTI_WAVE>
__device__ __forceinline__ void fft_sync() {
    if (MULTI_WAVE) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The SIMD's arbiter serves the OLDEST wave first.  Co-resident chains therefore do not share a SIMD evenly: on a CU that holds
// four N = 4096 workgroups the oldest finishes its 100 applications after 428 k cycles, the others after 471 k, 613 k and 745 k
// (profiles/r04_workgroup_trace.txt) -- every round of resident chains ends in a tail of its own making in which three, two
// and finally one workgroup are left on the CU, and a persistent schedule inherits the same staircase over its whole length.
// s_setprio overrides the age: each wave takes priority (slot + clock / 2^rotate) mod 4, slot = its wave slot in the SIMD, so
// that at any moment the waves of a SIMD still run in a strict order (which is what overlaps one wave's LDS phase with
// another's arithmetic) but over four periods every wave has had every rank, and co-resident chains end together.
// For any kernel that keeps several long-running waves on a SIMD (the in-LDS `multiple` kernels; a user kernel that calls the
// device functions in a loop): construct once, call at_application() once per iteration.  rotate = log2 of the period in shader
// clocks (15 = 14 us measured best, profiles/r04_priority_rotation.txt); 0: leave the arbiter alone.
struct WavePriority {
    int slot, shift;
    __device__ __forceinline__ explicit WavePriority(int rotate = 15) : shift(rotate) {
        slot = (int)(__builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((4 - 1) << 11)) & 3u);   // wave slot in the SIMD
    }
    // Outside the applications -- a wave that is copying a tile between device memory and LDS, or handing a chain over -- the wave
    // takes the TOP rank: those phases are short and mostly waiting, and a wave that runs them at whatever rank it happens to hold
    // (a newly started one: the lowest) is starved by the three that are computing: per-workgroup traces of round 5 showed the
    // fourth wave of every SIMD taking 15-35 us over its first tile (the others: 6), and it is the one the launch then ends with.
    __device__ __forceinline__ void between_applications() const {
        if (shift > 0) __builtin_amdgcn_s_setprio(3);
    }
    __device__ __forceinline__ void leaving() const {
        if (shift > 0) __builtin_amdgcn_s_setprio(0);
    }
    // called once per application: the rank follows the CU's clock, so at any moment the waves of a SIMD hold a permutation of
    // the ranks (their slots differ) and every wave holds every rank for the same share of the time
    __device__ __forceinline__ void at_application(int = 0) const {
        if (shift <= 0) return;
        const unsigned now = (unsigned)(__builtin_readcyclecounter() >> shift);
        switch ((slot + (int)now) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
};

// Sixteen single ds_read_b64 off ONE address register with compile-time float2 offsets OFF(k), issued in the order the
// caller names (k = 0 .. 15 -> v[k]).  For the layouts where a thread reads CONTIGUOUS elements of a padded row (the
// bit-reversed rows of the no-reorder variants): merged into ds_read2_b64, as hipcc does with plain C++ loads, those are served
// in 16-lane groups over 32 banks, where the 4-dword footprints of neighbouring rows overlap (2-way conflicts); single b64 reads
// over 64 banks are conflict free.  They are `volatile` 64-bit LDS loads: hipcc neither merges them nor reorders them, and --
// unlike inline assembly -- COUNTS them, so it places a stepped s_waitcnt lgkmcnt(n) in front of the first use of each value:
// the first butterflies start while the later reads are still in flight (an inline-assembly block that ended in a blanket
// s_waitcnt lgkmcnt(0) was rounds 1-2's form: profiles/r03_ab_step1.txt).
template <class OFF>
__device__ __forceinline__ void ds_read16_single(float2 (&v)[16], const float2* base) {
    typedef __attribute__((address_space(3))) const float2 lds_float2;
    typedef __attribute__((address_space(3))) const volatile unsigned long long lds_u64;
    lds_u64* p = (lds_u64*)(lds_float2*)base;
    unsigned long long w[16];   // 64-bit integers, not <2 x float>: vector-typed results invite v_pk_add_f32 (half rate) downstream
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = p[OFF::at(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = make_float2(__uint_as_float((unsigned)w[k]), __uint_as_float((unsigned)(w[k] >> 32)));
}

// issue order of sixteen reads that feed B butterflies of radix R = 16 / B (decimation in time: the innermost radix-2
// pairs are (j, j + R/2), then (j + R/4, ...)): bit-reversed within each butterfly, so that every group of reads
// completes the inputs of whole sub-butterflies
template <int R>
constexpr int dit_issue_slot(int k) {
    const int b = k / R, j = k % R;
    int rev = 0;
    for (int m = 1, w = R >> 1; w >= 1; m <<= 1, w >>= 1) rev |= (j & m) ? w : 0;
    return b * R + rev;
}

// Sixteen reads base[STRIDE * i]: plain C++, which hipcc merges pairwise into ds_read2_b64 -- half the instructions, and
// measured faster wherever the merged accesses are conflict free (the natural-order loads and the t-major last layout,
// profiles/r02_ab_mult.txt).
template <int STRIDE>
__device__ __forceinline__ void lds_read16(float2 (&r)[16], const float2* base) {
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = base[STRIDE * i];
}

// ------------------------------------------------------------------------------------------------
// Small in-register DFTs (R = 2, 4, 8, 16), decimation in time on compile-time indices.
// in[k*STRIDE], k < R  ->  out[q], q < R (natural order).  DIR = 0: e^{-2 pi i/R}, 1: e^{+}.
// ------------------------------------------------------------------------------------------------
template <int IDX16, int DIR, bool FIXED = false>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    // multiplies by W_16^IDX16 (IDX16 in [0,8)); conjugate for DIR = 1
    constexpr float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, h = 0.70710678118654752440f;
    if constexpr (IDX16 == 0) return a;
    else if constexpr (IDX16 == 4) return DIR ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    else if constexpr (IDX16 == 2) return DIR ? make_float2(h * (a.x - a.y), h * (a.x + a.y)) : make_float2(h * (a.x + a.y), h * (a.y - a.x));
    else if constexpr (IDX16 == 6) return DIR ? make_float2(-h * (a.x + a.y), h * (a.x - a.y)) : make_float2(h * (a.y - a.x), -h * (a.x + a.y));
    else {
        constexpr float wr = (IDX16 == 1) ? c1 : (IDX16 == 3) ? s1 : (IDX16 == 5) ? -s1 : -c1;
        constexpr float wi_f = (IDX16 == 1) ? -s1 : (IDX16 == 3) ? -c1 : (IDX16 == 5) ? -c1 : -s1;
        constexpr float wi = DIR ? -wi_f : wi_f;
        if constexpr (FIXED) return make_float2(__builtin_fmaf(a.x, wr, -(a.y * wi)), __builtin_fmaf(a.x, wi, a.y * wr));     // (cmul_fixed's reason)
        else return make_float2(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
    }
}

template <int R, int STRIDE, int DIR, bool TAN = true, bool FIXED = false>
struct SmallDft {
    static_assert(!(TAN && FIXED), "the tangent form is fused multiply-adds throughout: nothing to fix");
    __device__ static __forceinline__ void run(const float2* in, float2* out) {
        float2 e[R / 2], o[R / 2];
        SmallDft<R / 2, 2 * STRIDE, DIR, TAN, FIXED>::run(in, e);
        SmallDft<R / 2, 2 * STRIDE, DIR, TAN, FIXED>::run(in + STRIDE, o);
        combine<0>(e, o, out);
    }
    template <int K>
    __device__ static __forceinline__ void combine(const float2* e, const float2* o, float2* out) {
        if constexpr (K < R / 2) {
            constexpr int IDX = K * (16 / R);
            if constexpr (TAN && (IDX & 1)) {
                // e +- W*o with W = wr * (1 + i*tn): two multiply-adds for u = o * (1 + i*tn), four for e +- wr*u -- six
                // instructions where product, sum and difference take eight (the odd powers of W_16 only: the others are
                // cheaper still as they are).  16 instructions fewer per N = 1024 FFT; in-LDS path N >= 256 +0.5-2.5 %,
                // N = 32 / 64 1-7 % SLOWER on the general engine (their register-transposed kernels schedule worse with it): TAN = false
                // there (profiles/r02_ab_tan.txt); the lane engines of round 5 take it: +2...+4 % (profiles/r05_quad64.txt)
                constexpr float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f;
                constexpr float wr = (IDX == 1) ? c1 : (IDX == 3) ? s1 : (IDX == 5) ? -s1 : -c1;
                constexpr float wi_f = (IDX == 1) ? -s1 : (IDX == 3) ? -c1 : (IDX == 5) ? -c1 : -s1;
                constexpr float tn = (DIR ? -wi_f : wi_f) / wr;
                const float ux = fmaf(-tn, o[K].y, o[K].x), uy = fmaf(tn, o[K].x, o[K].y);
                out[K] = make_float2(fmaf(wr, ux, e[K].x), fmaf(wr, uy, e[K].y));
                out[K + R / 2] = make_float2(fmaf(-wr, ux, e[K].x), fmaf(-wr, uy, e[K].y));
            } else {
                float2 t = mul_w16<IDX, DIR, FIXED>(o[K]);
                out[K] = cadd(e[K], t);
                out[K + R / 2] = csub(e[K], t);
            }
            combine<K + 1>(e, o, out);
        }
    }
};
template <int STRIDE, int DIR, bool TAN, bool FIXED>
struct SmallDft<1, STRIDE, DIR, TAN, FIXED> {
    __device__ static __forceinline__ void run(const float2* in, float2* out) { out[0] = in[0]; }
};

// ------------------------------------------------------------------------------------------------
// Twiddles a thread keeps in registers.  u = thread index inside its FFT (0 .. T-1).
// ------------------------------------------------------------------------------------------------
// The register twiddles of Engine<N> as ROWS: one row of T values per register slot, so that the threads of an FFT
// read consecutive addresses.  Picked straight out of twiddle_4096, slot s of thread u is element u * q1 * 4096/N: a
// gather that touches 30-60 cache lines per load instruction -- ten times the requests of the tile a wave then moves,
// on the path every global access takes.  With 12288 workgroups per launch (12 generations per workgroup slot) that
// was a FIXED 45-55 us per launch of the external kernels, whatever the batch (N = 1024, 4 / 2 / 1 GiB each way:
// 0.80 / 0.78 / 0.71 of the HBM peak before, 0.815 / 0.82 / 0.82 after; round 2's tools/size_effect.py (git history)).
// Same values, same rounding: the rows are built at compile time from the table.
template <int N>
struct TwiddleRows {
    using G = Geometry<N>;
    TwiddleValue w1[16 * G::T];                        // [b*R1 + q1][u] = W_N^{(u + T*b) * q1}
    TwiddleValue wm[(G::RM > 1 ? G::RM : 1) * 16];     // [q2][t2]       = W_{T1}^{t2 * q2}
    constexpr TwiddleRows() : w1{}, wm{} {
        for (int b = 0; b < G::B1; ++b)
            for (int q1 = 0; q1 < G::R1; ++q1)
                for (int u = 0; u < G::T; ++u) w1[(b * G::R1 + q1) * G::T + u] = twiddle_values[((u + G::T * b) * q1 * (4096 / N)) & 4095];
        for (int q2 = 0; q2 < (G::RM > 1 ? G::RM : 1); ++q2)
            for (int t2 = 0; t2 < 16; ++t2) wm[q2 * 16 + t2] = twiddle_values[(t2 * q2 * (4096 / G::T1)) & 4095];
    }
};
template <int N>
static __device__ const TwiddleRows<N> twiddle_rows = TwiddleRows<N>();

template <int N, int DIR>
struct Twiddles {
    using G = Geometry<N>;
    float2 w1[16];                         // [b*R1 + q1] = W_N^{(u + T*b) * q1}
    float2 wm[G::RM > 1 ? G::RM : 1];      // [q2] = W_{T1}^{t2 * q2}
    __device__ static __forceinline__ float2 from_row(const TwiddleValue* p) {
        const TwiddleValue v = *p;
        return make_float2(v.x, DIR ? -v.y : v.y);
    }
    __device__ __forceinline__ void init(int u, int t2) {
#pragma unroll
        for (int b = 0; b < G::B1; ++b)
#pragma unroll
            for (int q1 = 1; q1 < G::R1; ++q1) w1[b * G::R1 + q1] = from_row(&twiddle_rows<N>.w1[(b * G::R1 + q1) * G::T + u]);
        if constexpr (G::RM > 1) {
#pragma unroll
            for (int q2 = 1; q2 < G::RM; ++q2) wm[q2] = from_row(&twiddle_rows<N>.wm[q2 * 16 + t2]);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// The engine.  One instance per thread; tid = threadIdx.x in a 256-thread workgroup.
// ------------------------------------------------------------------------------------------------
template <int N, int DIR, int REORDER>
struct Engine {
    using G = Geometry<N>;
    static constexpr int T = G::T, R1 = G::R1, RM = G::RM, T1 = G::T1, B1 = G::B1, BM = G::BM;
    static constexpr int S1 = G::S1, S2 = G::S2, S0 = G::S0, SF = G::SF;
    static constexpr int E_BITS = ilog2c(N), T_BITS = ilog2c(T), R1_BITS = ilog2c(R1), B1_BITS = ilog2c(B1);

    // physical lane bit of thread bit i of a register-two-pass FFT: i + kLaneShift -- the top log2(T) bits of the position inside a
    // 16-lane row (row-DPP transposes).  (These transposes serve the external kernels of N = 32 / 64 and the no-reorder bit reversal;
    // the in-LDS path of N = 32, whose cost they were, runs on PairEngine32 below since round 5.  What was measured on them there --
    // neighbouring rows + v_permlane16_swap -8...-10 %, ds_swizzle + selects -20 % -- is in profiles/HISTORY.md.)
    static constexpr int kLaneShift = !G::kRegTwoPass ? 0 : 4 - T_BITS;

    int u;        // thread inside the FFT
    int fft;      // FFT inside the workgroup
    int t1;       // pass-1 role (which butterfly t1 + T*b this thread computes)
    int t2, a;    // middle / last pass roles: v = t2 + 16*a
    Twiddles<N, DIR> tw;

    __device__ __forceinline__ void init(int tid) {
        u = tid % T;
        fft = tid / T;
        if constexpr (G::kRegTwoPass) {
            // the FFT's threads are the TOP log2(T) bits of the position inside a 16-lane row, so that the
            // DPP transposes select their lanes with the bank mask (lane bits 2, 3) wherever possible;
            // every LDS instruction still touches the same set of addresses per lane group
            const int lane = tid & 63;
            u = (lane >> kLaneShift) & (T - 1);
            if constexpr (kLaneShift == 4) fft = (tid >> 6) * 32 + (lane >> 5) * 16 + (lane & 15);
            else fft = (tid >> 6) * (64 / T) + (lane >> 4) * (16 / T) + (lane & ((1 << kLaneShift) - 1));
        }
        t2 = u & 15;
        a = u >> 4;
        // REORDER: role = lane.  No reorder: the thread with role t1 reads row rev_T(t1) of the
        // padded transposition image (to_pass1_layout); with role = lane the 32 lanes of a DS read
        // group would hit only 32 / 2^(log2(T)-5) different banks (T = 64/128/256: 2/4/8-way).
        // XOR-ing the top log2(T)-5 role bits with the reversed low lane bits makes the rows of a
        // read group distinct mod 32, while the low 4 role bits stay the lane's (exchange-1 writes
        // remain conflict free).
        t1 = u;
        if constexpr (!REORDER && G::kRegTwoPass) {
            // register transposition: lane u ends up with row u of the bit-reversed image, i.e. role rev_T(u)
            t1 = (int)(__brev((unsigned)u) >> (32 - T_BITS));
        }
        if constexpr (!REORDER && T_BITS > 5 && !G::kRegExchange1) {
            constexpr int m = T_BITS - 5;
            t1 = u ^ ((int)(__brev((unsigned)(u & ((1 << m) - 1))) >> (32 - m)) << 5);
        }
        tw.init(t1, t2);
    }

    // ---- inputs: natural order, r[c] = x[u + T*c] (consecutive threads -> consecutive elements) ----
    // (threads of an out-of-range FFT are pointed at FFT 0 by the caller instead of being predicated
    // per element: 16 unconditional back-to-back loads; their result is never stored)
    __device__ __forceinline__ void load_global(float2 (&r)[16], const float2* __restrict__ g) const {
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = gload(g + u + T * c);
    }
    __device__ __forceinline__ void load_lds(float2 (&r)[16], const float2* sf) const { lds_read16<T>(r, sf + u); }

    // ---- natural registers -> pass-1 slots r[b*R1 + r1] = x'[t1 + T1*r1], t1 = u + T*b ------------
    // REORDER: x' = x, and t1 + T1*r1 = u + T*(b + B1*r1): a compile-time renaming of registers.
    // No reorder: x'[n] = x[bitrev(n)].  bitrev(t1 + T1*r1) = 16*rev_T(u) + rev_B1(b)*R1 + rev_R1(r1):
    // every thread needs 16 CONTIGUOUS elements, the transpose of what coalesced/conflict-free
    // accesses deliver.  The transposition goes through the FFT's LDS region with one pad per 16
    // elements (position p at p + p/16), which makes both the write (consecutive lanes ->
    // consecutive p) and the read (lane -> its own row of 17) bank-conflict free.
    // Precondition: the region is free (earlier accesses ordered by fft_sync).
    __device__ __forceinline__ void to_pass1_layout(float2 (&r)[16], float2* sf) const {
        if constexpr (REORDER) {
            if constexpr (B1 > 1) {
                float2 t[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) t[c] = r[c];
#pragma unroll
                for (int b = 0; b < B1; ++b)
#pragma unroll
                    for (int r1 = 0; r1 < R1; ++r1) r[b * R1 + r1] = t[b + B1 * r1];
            }
        } else if constexpr (G::kRegTwoPass) {
            // r[c] = x[u + T*c] = x[16*rho + j] with rho = c >> B1_BITS and j = ((c & (B1-1)) << T_BITS) | u.
            // Swapping lane bit i with register bit B1_BITS + i gives lane = rho, register c'' = x[16*lane + j],
            // j = ((c'' & (B1-1)) << T_BITS) | (c'' >> B1_BITS); the thread's role is t1 = rev_T(lane) (init).
#pragma unroll
            for (int i = 0; i < T_BITS; ++i) swap_lane_bit_with_register_bit(r, i, B1_BITS + i);
            float2 t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = r[c];
#pragma unroll
            for (int b = 0; b < B1; ++b)
#pragma unroll
                for (int r1 = 0; r1 < R1; ++r1) {
                    const int rb = (B1_BITS > 0) ? (int)(__brev((unsigned)b) >> (32 - (B1_BITS > 0 ? B1_BITS : 1))) : 0;
                    const int rr = (R1_BITS > 0) ? (int)(__brev((unsigned)r1) >> (32 - (R1_BITS > 0 ? R1_BITS : 1))) : 0;
                    const int j = rb * R1 + rr;
                    r[b * R1 + r1] = t[((j & (T - 1)) << B1_BITS) | (j >> T_BITS)];
                }
        } else {
            bitrev_write(r, sf);
            fft_sync<G::kMultiWave>();
            bitrev_read(r, sf);
            fft_sync<G::kMultiWave>();
        }
    }

    // The two halves of the LDS transposition of the no-reorder variants (N >= 128).
    // bitrev_write: natural registers (r[c] = x[u + T*c]; also the layout of a transform's RESULT) -> the padded
    // image, element p at p + (p >> kPadShift): consecutive lanes write consecutive p, conflict free.
    __device__ __forceinline__ void bitrev_write(const float2 (&r)[16], float2* sf) const {
        constexpr int PS = G::kPadShift;
        // (u + T*c) >> PS == (u >> PS) + (T*c >> PS): T*c is a multiple of 2^PS (T >= 2^PS), or u + (T*c mod 2^PS) < 2^PS
        // (T < 2^PS, where u >> PS = 0): one base address and sixteen compile-time offsets
        const int base = u + (u >> PS);
#pragma unroll
        for (int c = 0; c < 16; ++c) sf[base + T * c + ((T * c) >> PS)] = r[c];
    }
    // bitrev_read: padded image -> pass-1 slots; the thread with role t1 reads the 16 contiguous elements of row
    // rev_T(t1), in bit-reversed order within the row.  Sixteen single ds_read_b64 off ONE address register with
    // immediate offsets.  Written as inline assembly because hipcc merges neighbouring reads into ds_read2_b64, which
    // is served in 16-lane groups over 32 banks -- half the rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table) and,
    // here, with the rows of bit-reversed neighbours colliding (measured 0.14 conflict cycles per LDS cycle in round 1).
    // The compiler does not count inline-assembly DS operations, so the block ends with its own s_waitcnt.
    __device__ __forceinline__ void bitrev_read(float2 (&r)[16], const float2* sf) const {
        constexpr int PS = G::kPadShift;
        const int g16 = 16 * (int)(__brev((unsigned)t1) >> (32 - T_BITS));
        const float2* row = sf + g16 + (g16 >> PS);
        float2 v[16];
        ds_read16_single<BitrevOffsets>(v, row);
#pragma unroll
        for (int k = 0; k < 16; ++k) r[dit_issue_slot<R1>(k)] = v[k];
    }
    struct BitrevOffsets {
        static constexpr int at(int k) { return slot_source(dit_issue_slot<R1>(k)); }
    };
    // pass-1 slot i = b*R1 + r1 takes element rev_B1(b)*R1 + rev_R1(r1) of the thread's row
    static constexpr int slot_source(int i) {
        const int b = i / R1, r1 = i % R1;
        int rb = 0, rr = 0;
        for (int k = 0; k < B1_BITS; ++k) rb |= ((b >> k) & 1) << (B1_BITS - 1 - k);
        for (int k = 0; k < R1_BITS; ++k) rr |= ((r1 >> k) & 1) << (R1_BITS - 1 - k);
        return rb * R1 + rr;
    }
    // position of element n of an FFT in the padded image
    __device__ static __forceinline__ int padded_index(int n) { return n + (n >> G::kPadShift); }

    // ---- pass 1: B1 radix-R1 butterflies, then W_N^{t1*q1} -------------------------------------
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

    // ---- exchange after pass 1 -----------------------------------------------------------------
    // One-bit transposes between a lane bit and a register-index bit.  v_permlane16_swap(A, B) swaps
    // the odd 16-lane rows of A with the even rows of B, v_permlane32_swap(A, B) the upper 32 lanes
    // of A with the lower 32 of B: exactly "element (lane bit 1, reg bit 0) <-> (lane bit 0, reg bit 1)".
    template <int LANE_BIT>
    __device__ static __forceinline__ void swap_bit(float2& A, float2& B) {
        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
        uint2v x, y;
        if constexpr (LANE_BIT == 4) {
            x = __builtin_amdgcn_permlane16_swap(__float_as_uint(A.x), __float_as_uint(B.x), false, false);
            y = __builtin_amdgcn_permlane16_swap(__float_as_uint(A.y), __float_as_uint(B.y), false, false);
        } else {
            x = __builtin_amdgcn_permlane32_swap(__float_as_uint(A.x), __float_as_uint(B.x), false, false);
            y = __builtin_amdgcn_permlane32_swap(__float_as_uint(A.y), __float_as_uint(B.y), false, false);
        }
        A = make_float2(__uint_as_float(x[0]), __uint_as_float(y[0]));
        B = make_float2(__uint_as_float(x[1]), __uint_as_float(y[1]));
    }

    // N = 512 / 1024.  After pass 1 lane (t2, row r2) holds element (t1 = t2 + 16*r2, q1) in r[q1];
    // the middle pass wants lane (t2, row a) to hold (t2 + 16*r2, q1 = a*BM + c) in r[c*RM + r2]:
    // transpose the row bits with the top log2(RM) bits of q1, then rename registers.
    __device__ __forceinline__ void exchange1_registers(float2 (&r)[16]) const { exchange1_registers_static(r); }
    __device__ static __forceinline__ void exchange1_registers_static(float2 (&r)[16]) {
        if constexpr (RM == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((q & 4) == 0) swap_bit<4>(r[q], r[q + 4]);
#pragma unroll
            for (int q = 0; q < 8; ++q) swap_bit<5>(r[q], r[q + 8]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) swap_bit<4>(r[q], r[q + 8]);
        }
        // now r[BM*r2 + c] = (t2 + 16*r2, a*BM + c)  ->  r[c*RM + r2]
        float2 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = r[i];
#pragma unroll
        for (int c = 0; c < BM; ++c)
#pragma unroll
            for (int r2 = 0; r2 < RM; ++r2) r[c * RM + r2] = t[BM * r2 + c];
    }

    // The same one-bit transpose for the lane bits 0 ... 3 (partner inside the 16-lane row) on a float2 pair in FOUR instructions:
    // v_cndmask_b32 takes its first source through DPP, so "keep mine or take the partner's" is ONE instruction per dword and side --
    // the lanes with the bit clear keep A and take the partner's A into B, the others keep B and take the partner's B into A;
    // partner = quad_perm (bits 0 / 1) or a rotation of the row (bits 2 / 3: the lanes with the bit set read lane - 2^bit =
    // row_ror:2^bit, the others lane + 2^bit = row_ror:16 - 2^bit).  hipcc does not fold a DPP move into the select (it emits
    // v_mov_b32_dpp + v_cndmask_b32), hence the inline assembly; the lane masks are constants because every transform's lanes start
    // at a multiple of their count.  s_nop 1: the two wait states a DPP read needs after a VALU write of the same register (the
    // assembler does not see into the block).  Measured (profiles/r05_select_row_swap.txt): eight selects per block, or the blocks
    // without `volatile` (free to be scheduled), are slower than this.
#define SMFFT_SELECT(CTL_A, CTL_B)                                                                          \
    "s_nop 1\n\ts_mov_b64 vcc, %8\n\t"                                                                     \
    "v_cndmask_b32_dpp %0, %6, %4, vcc " CTL_A " row_mask:0xf bank_mask:0xf\n\t"                            \
    "v_cndmask_b32_dpp %1, %7, %5, vcc " CTL_A " row_mask:0xf bank_mask:0xf\n\t"                            \
    "s_not_b64 vcc, vcc\n\t"                                                                               \
    "v_cndmask_b32_dpp %2, %4, %6, vcc " CTL_B " row_mask:0xf bank_mask:0xf\n\t"                            \
    "v_cndmask_b32_dpp %3, %5, %7, vcc " CTL_B " row_mask:0xf bank_mask:0xf"
    template <int LANE_BIT>
    __device__ static __forceinline__ void swap_bit_select(float2& A, float2& B) {
        static_assert(LANE_BIT >= 0 && LANE_BIT <= 3, "inside a 16-lane row");
        constexpr unsigned long long lo = LANE_BIT == 0 ? 0x5555555555555555ull : LANE_BIT == 1 ? 0x3333333333333333ull
                                        : LANE_BIT == 2 ? 0x0F0F0F0F0F0F0F0Full : 0x00FF00FF00FF00FFull;     // lanes with the bit clear
        float nax, nay, nbx, nby;
#define SMFFT_SELECT_OPERANDS : "=&v"(nax), "=&v"(nay), "=&v"(nbx), "=&v"(nby) : "v"(A.x), "v"(A.y), "v"(B.x), "v"(B.y), "s"(lo) : "vcc", "scc"   /* (s_not_b64 writes SCC) */
        if constexpr (LANE_BIT == 0) asm volatile(SMFFT_SELECT("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]") SMFFT_SELECT_OPERANDS);
        else if constexpr (LANE_BIT == 1) asm volatile(SMFFT_SELECT("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]") SMFFT_SELECT_OPERANDS);
        else if constexpr (LANE_BIT == 2) asm volatile(SMFFT_SELECT("row_ror:4", "row_ror:12") SMFFT_SELECT_OPERANDS);
        else asm volatile(SMFFT_SELECT("row_ror:8", "row_ror:8") SMFFT_SELECT_OPERANDS);
#undef SMFFT_SELECT_OPERANDS
        A = make_float2(nax, nay);     // lanes with the bit clear keep A; the others take the partner's B
        B = make_float2(nbx, nby);     // lanes with the bit set keep B; the others take the partner's A
    }
#undef SMFFT_SELECT
    // Lane bits 2 / 3 with two bank-masked v_mov_b32_dpp per dword pair, in place (a bank = 4 lanes of a row): the second move reads
    // what the first overwrote, so the compiler copies one register first -- three instructions per dword pair against the selects'
    // two, but instructions the compiler schedules.  The reference-contract ladder of N >= 1024 (4 ... 16 waves per block between
    // barriers) measured 2 ... 8 % FASTER with this form, every other user 3 ... 13 % slower (profiles/r05_select_row_swap.txt).
    template <int LANE_BIT>
    __device__ static __forceinline__ void swap_bit_dpp_dword(float& A, float& B) {
        static_assert(LANE_BIT == 2 || LANE_BIT == 3, "bits 0 / 1: swap_bit_select");
        const int a = __float_as_int(A), b = __float_as_int(B);
        int na, nb;
        if constexpr (LANE_BIT == 2) {
            na = __builtin_amdgcn_update_dpp(a, b, 0x114 /* row_shr:4 */, 0xF, 0xA, false);   // banks 1,3 <- B of lane-4
            nb = __builtin_amdgcn_update_dpp(b, a, 0x104 /* row_shl:4 */, 0xF, 0x5, false);   // banks 0,2 <- A of lane+4
        } else {
            na = __builtin_amdgcn_update_dpp(a, b, 0x128 /* row_ror:8 */, 0xF, 0xC, false);   // lanes 8..15 <- B of lane^8
            nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);                    // lanes 0..7  <- A of lane^8
        }
        A = __int_as_float(na);
        B = __int_as_float(nb);
    }
    // all 8 register pairs (c, c | 1 << reg_bit), c with that bit clear
    __device__ __forceinline__ void swap_lane_bit_with_register_bit(float2 (&r)[16], int lane_bit, int reg_bit) const {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if ((c >> reg_bit) & 1) continue;
            float2& A = r[c];
            float2& B = r[c | (1 << reg_bit)];
            switch (lane_bit + kLaneShift) {
                case 0: swap_bit_select<0>(A, B); break;
                case 1: swap_bit_select<1>(A, B); break;
                case 2: swap_bit_select<2>(A, B); break;
                default: swap_bit_select<3>(A, B); break;
            }
        }
    }

    // Two-pass sizes in registers.  After pass 1 lane u holds element (n1 = t1 + T*b, q1) in r[b*R1 + q1].
    // Swapping lane bit i with register bit i (i < log2 T) leaves lane w with q1 = w and register
    // b*R1 + v = the element of the lane v it came from, whose role was t1 = v (reorder) or rev_T(v):
    // x[t + T*b] of the last pass = r[b*R1 + (t or rev_T(t))].
    __device__ __forceinline__ void exchange_last_registers(float2 (&r)[16], float2 (&x)[16]) const {
#pragma unroll
        for (int i = 0; i < T_BITS; ++i) swap_lane_bit_with_register_bit(r, i, i);
#pragma unroll
        for (int b = 0; b < B1; ++b)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int v = REORDER ? t : (int)(__brev((unsigned)t) >> (32 - T_BITS));
                x[t + T * b] = r[b * R1 + v];
            }
    }

    __device__ __forceinline__ void exchange1_write(const float2 (&r)[16], float2* sf) const {
        if constexpr (RM > 1) {
            // q1-major rows of S1: element (t1, q1) at q1*S1 + t1   (B1 == 1)
#pragma unroll
            for (int q1 = 0; q1 < 16; ++q1) sf[q1 * S1 + t1] = r[q1];
        } else {
            // two-pass sizes go straight to the last layout: element (t1, q1) at q1*S0 + t1
#pragma unroll
            for (int b = 0; b < B1; ++b)
#pragma unroll
                for (int q1 = 0; q1 < R1; ++q1) sf[q1 * S0 + t1 + T * b] = r[b * R1 + q1];
        }
    }

    // ---- middle pass (N >= 512): BM radix-RM butterflies over r2, then W_{T1}^{t2*q2} -----------
    __device__ __forceinline__ void middle(float2 (&r)[16], float2* sf) const {
        if constexpr (RM > 1) {
            if constexpr (!G::kRegExchange1) {
#pragma unroll
                for (int c = 0; c < BM; ++c)
#pragma unroll
                    for (int r2 = 0; r2 < RM; ++r2) r[c * RM + r2] = sf[(a * BM + c) * S1 + t2 + 16 * r2];
                fft_sync<G::kMultiWave>();
            }
#pragma unroll
            for (int c = 0; c < BM; ++c) {
                float2 y[RM];
                SmallDft<RM, 1, DIR>::run(&r[c * RM], y);
                r[c * RM] = y[0];
#pragma unroll
                for (int q2 = 1; q2 < RM; ++q2) r[c * RM + q2] = cmul(y[q2], tw.wm[q2]);
            }
            // element (t2, klow = q1 + 16*q2) at t2*S2 + klow
#pragma unroll
            for (int c = 0; c < BM; ++c)
#pragma unroll
                for (int q2 = 0; q2 < RM; ++q2) sf[t2 * S2 + (a * BM + c) + 16 * q2] = r[c * RM + q2];
            fft_sync<G::kMultiWave>();
        }
    }

    // ---- last pass: one radix-16 butterfly per thread; r[q3] = X[u + T*q3] -----------------------
    __device__ __forceinline__ void last(float2 (&r)[16], const float2* sf) const {
        float2 x[16];
        if constexpr (RM > 1) lds_read16<S2>(x, sf + u);
        else lds_read16<1>(x, sf + u * S0);
        SmallDft<16, 1, DIR>::run(x, r);
    }

    // ---- outputs -------------------------------------------------------------------------------
    __device__ __forceinline__ void store_global(const float2 (&r)[16], float2* __restrict__ g, bool active) const {
        if (active) {
#pragma unroll
            for (int q3 = 0; q3 < 16; ++q3) gstore(g + u + T * q3, r[q3]);
        }
    }
    __device__ __forceinline__ void store_lds(const float2 (&r)[16], float2* sf) const {
#pragma unroll
        for (int q3 = 0; q3 < 16; ++q3) sf[u + T * q3] = r[q3];
    }

    // registers (natural order, r[c] = x[u + T*c]) -> registers (r[q] = X[u + T*q]) through the
    // FFT's LDS region.
    // Precondition: every earlier LDS access of this FFT's region has been ordered by fft_sync.
    __device__ __forceinline__ void transform(float2 (&r)[16], float2* sf) const {
        to_pass1_layout(r, sf);
        transform_from_pass1_slots(r, sf);
    }
    // the transform of registers that already hold the pass-1 slots (after to_pass1_layout or bitrev_read)
    __device__ __forceinline__ void transform_from_pass1_slots(float2 (&r)[16], float2* sf) const {
        pass1(r);
        if constexpr (G::kRegTwoPass) {
            float2 x[16];
            exchange_last_registers(r, x);
            SmallDft<16, 1, DIR, false>::run(x, r);
            return;
        }
        if constexpr (G::kRegExchange1) {
            exchange1_registers(r);
        } else {
            exchange1_write(r, sf);
            fft_sync<G::kMultiWave>();
        }
        middle(r, sf);
        last(r, sf);
    }
};

// ------------------------------------------------------------------------------------------------
// In place on one FFT's LDS region (natural order in, natural order out): the building block of the
// device functions in smfft_device_functions.hpp.  `stride` = float2 distance between the FFTs the
// workgroup holds (Geometry::SF in the tiled kernels, N where FFTs are packed contiguously).
// ------------------------------------------------------------------------------------------------
template <int N, int DIR, int REORDER>
__device__ __forceinline__ void fft_lds_inplace(float2* s, const Engine<N, DIR, REORDER>& eng, int stride = Geometry<N>::SF) {
    using G = Geometry<N>;
    float2* sf = s + eng.fft * stride;
    float2 r[16];
    eng.load_lds(r, sf);
    fft_sync<G::kMultiWave>();          // all inputs are in registers before the region is reused
    eng.transform(r, sf);
    fft_sync<G::kMultiWave>();          // all exchange reads done before the results overwrite them
    eng.store_lds(r, sf);
}

// ------------------------------------------------------------------------------------------------
// N = 32 in the in-LDS path: the FFT is a PAIR of lanes (u = 0 / 1, eight lanes apart: row_ror:8) with sixteen registers each,
// and the radix-2 stage that crosses the pair is one v_fmac_f32 per dword whose first source comes through DPP:
//     own <- own + s * partner's own                 (s = +-1 per lane, a register)
// -- exchange and butterfly in ONE instruction, where the general engine spends a select per dword for the exchange and an
// addition per dword for the butterfly (32 + 32 of its 252 instructions per application).  What makes it possible is to let the
// LAYOUT alternate instead of restoring it:
//     layout A: r[c] = x[u + 2c]         layout B: r[n] = x[n + 16u]
//     dit:  A -> B    Z_u = DFT16(r) . W_32^(u q),      lane p ends with X[q + 16p] = Z_0[q] + (-1)^p Z_1[q]
//     dif:  B -> A    y_0 = x[n] + x[n+16], y_1 = (x[n] - x[n+16]) W_32^n,   lane u ends with X[u + 2k] = DFT16(y_u)[k]
// Natural order: applications alternate dit, dif (the output of one IS the input layout of the other).  No reorder (the
// transform of x o bitrev, natural output): bitrev(u + 2c) = 16u + rev4(c), so layout A of x o bitrev is layout B of x with the
// registers renamed -- dit every time.  Signs: with s = (+1, -1) lane 1 ends a dit with -X[q + 16]; the next stage takes
// s = (-1, +1) and comes out plain (both dit and dif are linear in lane 1's registers), so nothing is ever negated in the
// loop -- only where a piece of a chain starts or ends on an odd application (load / store flip lane 1's sign bits there).  Which form an
// application takes depends on its index in the CHAIN only, so a chain cut between two workgroups computes the same bits.
// Per application: 140 (radix 16, tangent form) + 60 (fifteen twiddles) + 32 for the stage across the pair, no LDS memory access.
// ------------------------------------------------------------------------------------------------
// W_N^m, m < N, as ONE row (N = 32 / 64: 256 / 512 bytes): picked out of twiddle_4096 the lane engines' fifteen twiddles per lane lie
// 128 / 64 entries apart -- a cache line each, 15 ... 45 lines per wave at the start of every workgroup (the quad engine's workgroups
// reached their first tile 3 us later than the planar kernel's: profiles/r05_trace_summary.txt (c)).  Same values, same rounding.
template <int N>
struct LaneTwiddleRow {
    TwiddleValue w[N];
    constexpr LaneTwiddleRow() : w{} {
        for (int m = 0; m < N; ++m) w[m] = twiddle_values[m * (4096 / N)];
    }
};
template <int N>
static __device__ const LaneTwiddleRow<N> lane_twiddle_row = LaneTwiddleRow<N>();
template <int N, int DIR>
__device__ __forceinline__ float2 lane_twiddle(int m) {
    const TwiddleValue v = lane_twiddle_row<N>.w[m & (N - 1)];
    return make_float2(v.x, DIR ? -v.y : v.y);
}

template <int DIR, int REORDER>
struct PairEngine32 {
    static constexpr int N = 32;
    using G = Geometry<N>;
    int u, fft;
    float s_plain;            // +1 (lane 0) / -1 (lane 1): the stage whose inputs are plain; -s_plain: lane 1's are negated
    float2 tw[16];            // W_32^(u n): 1 in lane 0

    __device__ __forceinline__ void init(int tid) {
        const int lane = tid & 63;
        u = (lane >> 3) & 1;
        fft = (tid >> 6) * 32 + (lane >> 4) * 8 + (lane & 7);
        s_plain = u ? -1.f : 1.f;
#pragma unroll
        for (int n = 1; n < 16; ++n) {
            const float2 w = lane_twiddle<N, DIR>(n);
            tw[n] = u ? w : make_float2(1.f, 0.f);
        }
    }
    // r[i] <- r[i] + s * (partner's r[i]), i = 0 .. 15 (both dwords): two blocks of sixteen v_fmac_f32_dpp
    // (s_nop 1: the two wait states a DPP read needs after a VALU write of the same register)
    // The last SWZ of the sixteen values take the other road to the partner: ds_swizzle (lane ^ 8 through the LDS crossbar, no memory
    // access) + a plain v_fmac_f32 -- 2.3 vector cycles per dword instead of the fused form's 4.5, paid for on the otherwise idle LDS
    // pipe and in registers for the values in flight.  Measured (profiles/r05_lane_swizzle.txt): eight of sixteen in the dif form
    // +6 % (all sixteen +5 %, with spills), in the no-reorder dit +4 %; in the natural-order dit 0...+2 %: none there.
    template <int SWZ>
    __device__ static __forceinline__ void cross(float2 (&r)[16], float s) {
#pragma unroll
        for (int q = 16 - SWZ; q < 16; ++q) {
            const float px = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(r[q].x), 0x201F));     // bit mode: and 0x1f, or 0, xor 8
            const float py = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(r[q].y), 0x201F));
            r[q] = make_float2(__builtin_fmaf(px, s, r[q].x), __builtin_fmaf(py, s, r[q].y));
        }
#define SMFFT_FUSED8(B)                                                                                                  \
        asm volatile("s_nop 1\n\t"                                                                                       \
                     "v_fmac_f32_dpp %0, %0, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %1, %1, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %2, %2, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %3, %3, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %4, %4, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %5, %5, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %6, %6, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %7, %7, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %8, %8, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %9, %9, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_fmac_f32_dpp %10, %10, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_fmac_f32_dpp %11, %11, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_fmac_f32_dpp %12, %12, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_fmac_f32_dpp %13, %13, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_fmac_f32_dpp %14, %14, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                             \
                     "v_fmac_f32_dpp %15, %15, %16 row_ror:8 row_mask:0xf bank_mask:0xf"                                  \
                     : "+v"(r[B].x), "+v"(r[B].y), "+v"(r[B + 1].x), "+v"(r[B + 1].y), "+v"(r[B + 2].x), "+v"(r[B + 2].y),          \
                       "+v"(r[B + 3].x), "+v"(r[B + 3].y), "+v"(r[B + 4].x), "+v"(r[B + 4].y), "+v"(r[B + 5].x), "+v"(r[B + 5].y),  \
                       "+v"(r[B + 6].x), "+v"(r[B + 6].y), "+v"(r[B + 7].x), "+v"(r[B + 7].y)                                       \
                     : "v"(s))
        SMFFT_FUSED8(0);
        if constexpr (SWZ < 8) SMFFT_FUSED8(8);
#undef SMFFT_FUSED8
    }
    // layout A -> layout B: r[c] = x[u + 2c] -> r[q] = X[q + 16u]   (s: s_plain for plain inputs, -s_plain for a negated lane 1;
    // lane 1's results are negated in the first case, plain in the second)
    __device__ __forceinline__ void dit(float2 (&r)[16], float s) const {
        float2 y[16];
        SmallDft<16, 1, DIR, true>::run(r, y);
#pragma unroll
        for (int q = 1; q < 16; ++q) y[q] = cmul_fixed(y[q], tw[q]);
        cross<REORDER ? 0 : 8>(y, s);
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = y[q];
    }
    // layout B -> layout A: r[n] = x[n + 16u] -> r[k] = X[u + 2k]   (same convention for s; the results are plain when lane 1's inputs were negated)
    __device__ __forceinline__ void dif(float2 (&r)[16], float s) const {
        cross<8>(r, s);
#pragma unroll
        for (int n = 1; n < 16; ++n) r[n] = cmul_fixed(r[n], tw[n]);
        float2 y[16];
        SmallDft<16, 1, DIR, true>::run(r, y);
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = y[k];
    }
    // one application, number f of its chain (see the head of the struct); `f` odd: lane 1's registers come in negated
    __device__ __forceinline__ void apply(float2 (&r)[16], bool odd) const {
        const float s = odd ? -s_plain : s_plain;
        if constexpr (REORDER) {
            if (!odd) dit(r, s);
            else dif(r, s);
        } else {
            float2 x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = r[((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3)];     // layout A of x o bitrev
            dit(x, s);
#pragma unroll
            for (int q = 0; q < 16; ++q) r[q] = x[q];
        }
    }
    // the registers of a piece that starts at application f0 / ends before application f1 of its chain <-> the FFT's region of the
    // LDS image (natural order; PADDED: element p at p + (p >> 4), the no-reorder image of the tile copies)
    // (lane 1's negation where a piece starts or ends on an odd application is a flip of the SIGN BIT, not a product with -1: the
    //  README benchmark's 100 un-normalised applications overflow to infinities and NaNs half way, and a cut chain must still
    //  end with the bits of an uncut one -- flipping twice restores any bit pattern, multiplying a NaN twice need not)
    __device__ static __forceinline__ float negate_bits(float v, unsigned flip) { return __uint_as_float(__float_as_uint(v) ^ flip); }
    static constexpr bool kPadded = !REORDER;
    __device__ static __forceinline__ int image(int p) { return p + (kPadded ? (p >> G::kPadShift) : 0); }
    __device__ __forceinline__ void load(float2 (&r)[16], const float2* sf, int f0) const {
        const bool odd = f0 & 1;
        if (REORDER && !odd) {
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = sf[image(u + 2 * c)];
        } else {
            const unsigned flip = (odd && u) ? 0x80000000u : 0u;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const float2 v = sf[image(n + 16 * u)];
                r[n] = make_float2(negate_bits(v.x, flip), negate_bits(v.y, flip));
            }
        }
    }
    __device__ __forceinline__ void store(const float2 (&r)[16], float2* sf, int f1) const {
        const bool odd = f1 & 1;
        if (REORDER && !odd) {
#pragma unroll
            for (int c = 0; c < 16; ++c) sf[image(u + 2 * c)] = r[c];
        } else {
            const unsigned flip = (odd && u) ? 0x80000000u : 0u;
#pragma unroll
            for (int n = 0; n < 16; ++n) sf[image(n + 16 * u)] = make_float2(negate_bits(r[n].x, flip), negate_bits(r[n].y, flip));
        }
    }
};

// ------------------------------------------------------------------------------------------------
// N = 64 WITHOUT reorder in the in-LDS path, the same way: the FFT is a QUAD of lanes (j = 0 ... 3, a DPP quad) with sixteen registers
// each, and the radix-4 stage across the quad is two fused stages -- own <- own + s * partner's own, partner j ^ 2 through the LDS
// crossbar (ds_swizzle + v_fmac_f32), then partner j ^ 1 through DPP (v_fmac_f32_dpp) -- with a turn by -+i in lane 3 between them.  No reorder only: S2 transforms x o bitrev, and bitrev(j + 4c) = 16 rev2(j) + rev4(c) makes the input of lane
// j (role t1 = j: x'[j + 4c]) the contiguous block rev2(j) of the stored array with its registers renamed -- while the in-place radix-2
// network over the two lane bits leaves output block k in the lane rev2(k): lane j holds block rev2(j) before and after, for ever.
//     Z_j = DFT16(x'[j + 4c]) . W_64^(j q)
//     stage 1 (j ^ 2):  a = Z_0 + Z_2 (lane 0), b = Z_0 - Z_2 (lane 2), c = Z_1 + Z_3 (lane 1), d = Z_1 - Z_3 (lane 3);  lane 3: d <- -+i d
//     stage 2 (j ^ 1):  X[q] = a + c (lane 0), X[q + 32] = a - c (lane 1), X[q + 16] = b + d (lane 2), X[q + 48] = b - d (lane 3)
// Signs as in PairEngine32: own + s * partner's own cannot negate `own`, so a lane that wants "partner - own" ends negated.  From plain
// inputs, s1 = (+, +, -, -) and s2 = (+, -, +, -) leave the lanes (+, -, -, +); from inputs in that state the NEGATED sign vectors leave
// them plain: the state alternates with the application's number in the chain, nothing is negated in the loop, and a piece that starts
// or ends on an odd application flips the sign bits of lanes 1 and 2 where it loads / stores.  The natural-order variant stays on the
// planar engine (one LDS exchange per application there; this form would take a dit / dif pair like N = 32 and prices the same);
// the no-reorder one spent its time on the LDS unit (two exchanges per application, 89 % busy: DESIGN.md 5.2) and has none left.
// ------------------------------------------------------------------------------------------------
template <int DIR>
struct QuadEngine64 {
    static constexpr int N = 64;
    using G = Geometry<N>;
    int j, fft;
    float s1, s2;             // the sign vectors of an application whose inputs are plain; negated for one whose lanes 1 and 2 are
    bool turns;               // lane 3
    unsigned flip;            // the sign bit, in lanes 1 and 2
    float2 tw[16];            // W_64^(j q)

    __device__ __forceinline__ void init(int tid) {
        const int lane = tid & 63;
        j = lane & 3;
        fft = (tid >> 6) * 16 + (lane >> 2);
        s1 = (j & 2) ? -1.f : 1.f;
        s2 = (j & 1) ? -1.f : 1.f;
        turns = j == 3;
        flip = (j == 1 || j == 2) ? 0x80000000u : 0u;
#pragma unroll
        for (int q = 1; q < 16; ++q) tw[q] = lane_twiddle<N, DIR>(j * q);
    }
    // stage 2: r[i] <- r[i] + s * (lane ^ 1's r[i]), both dwords of sixteen registers, as DPP-fed v_fmac_f32 (stage 1 goes through the
    // LDS crossbar: apply)
    __device__ static __forceinline__ void cross_neighbour(float2 (&r)[16], float s) {
#define SMFFT_QUAD8(B, CTL)                                                                                              \
        asm volatile("s_nop 1\n\t"                                                                                       \
                     "v_fmac_f32_dpp %0, %0, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %1, %1, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %2, %2, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %3, %3, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %4, %4, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %5, %5, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %6, %6, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %7, %7, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %8, %8, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %9, %9, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                  \
                     "v_fmac_f32_dpp %10, %10, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                \
                     "v_fmac_f32_dpp %11, %11, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                \
                     "v_fmac_f32_dpp %12, %12, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                \
                     "v_fmac_f32_dpp %13, %13, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                \
                     "v_fmac_f32_dpp %14, %14, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t"                                \
                     "v_fmac_f32_dpp %15, %15, %16 " CTL " row_mask:0xf bank_mask:0xf"                                     \
                     : "+v"(r[B].x), "+v"(r[B].y), "+v"(r[B + 1].x), "+v"(r[B + 1].y), "+v"(r[B + 2].x), "+v"(r[B + 2].y),          \
                       "+v"(r[B + 3].x), "+v"(r[B + 3].y), "+v"(r[B + 4].x), "+v"(r[B + 4].y), "+v"(r[B + 5].x), "+v"(r[B + 5].y),  \
                       "+v"(r[B + 6].x), "+v"(r[B + 6].y), "+v"(r[B + 7].x), "+v"(r[B + 7].y)                                       \
                     : "v"(s))
        SMFFT_QUAD8(0, "quad_perm:[1,0,3,2]");
        SMFFT_QUAD8(8, "quad_perm:[1,0,3,2]");
#undef SMFFT_QUAD8
    }
    // lane 3: d <- -+i d (forward: (d.y, -d.x); inverse: (-d.y, d.x)), the other lanes unchanged: two selects per value (hipcc: v_cndmask_b32_e64
    // on an SGPR mask with the negation as a source modifier).  The same with v_swap_b32 under an EXEC mask of the lanes 3 and the sign
    // as a product measured 10 % SLOWER per launch (profiles/r05_quad64.txt).  That the turn cannot be folded into twiddles, signs or a
    // swapped / conjugated representation of some lanes is a parity argument: DESIGN.md 2.1a.
    __device__ __forceinline__ void turn(float2 (&y)[16]) const {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 d = y[q];
            y[q] = turns ? (DIR ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x)) : d;
        }
    }
    // one application, number f of its chain; r[i] = (the lane's sign) * stored element 16 rev2(j) + i, before and after
    __device__ __forceinline__ void apply(float2 (&r)[16], bool odd) const {
        float2 x[16], y[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = r[((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3)];     // x'[j + 4c]
        SmallDft<16, 1, DIR, true>::run(x, y);
#pragma unroll
        for (int q = 1; q < 16; ++q) y[q] = cmul_fixed(y[q], tw[q]);
        {   // stage 1 through the LDS crossbar (ds_swizzle, quad_perm [2,3,0,1]) + plain v_fmac_f32: it follows the twiddle products, which
            // cover its latency; +6 % against the fused form there.  Stage 2 the same way +4 %, both +1.5 %: the crossbar carries one.
            const float s = odd ? -s1 : s1;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float px = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(y[q].x), 0x804E));
                const float py = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(y[q].y), 0x804E));
                y[q] = make_float2(__builtin_fmaf(px, s, y[q].x), __builtin_fmaf(py, s, y[q].y));
            }
        }
        turn(y);
        cross_neighbour(y, odd ? -s2 : s2);
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = y[q];
    }
    // the registers of a piece that starts at application f0 / ends before application f1 of its chain <-> the FFT's region of the
    // padded LDS image of the tile copies (element p at p + (p >> 4))
    __device__ __forceinline__ int block_base() const { return 17 * (((j & 1) << 1) | (j >> 1)); }      // 16 rev2(j), padded
    __device__ __forceinline__ void load(float2 (&r)[16], const float2* sf, int f0) const {
        const unsigned m = (f0 & 1) ? flip : 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 v = sf[block_base() + i];
            r[i] = make_float2(__uint_as_float(__float_as_uint(v.x) ^ m), __uint_as_float(__float_as_uint(v.y) ^ m));
        }
    }
    __device__ __forceinline__ void store(const float2 (&r)[16], float2* sf, int f1) const {
        const unsigned m = (f1 & 1) ? flip : 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) sf[block_base() + i] = make_float2(__uint_as_float(__float_as_uint(r[i].x) ^ m), __uint_as_float(__float_as_uint(r[i].y) ^ m));
    }
};

}  // namespace smfft
