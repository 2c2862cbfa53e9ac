/*
 * smfft_oracle.c -- CPU restatement of the KAdamek/SMFFT device algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: the shipped
 * library (smfft_amd/csrc -> libsmfft_amd.so) never links, loads or calls this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" against reference-held vectors -- the reference ships no
 * golden vectors, known-answer tests or fixtures (its only self-check is a run-time comparison
 * with cuFFT, SMFFT_CooleyTukey_C2C/FFT.c:148-160), and it is CUDA, so it cannot be built or
 * run in this image.  What this oracle IS pinned against: numpy.fft in complex128 on seeded
 * inputs (the fp64 reference north_star names; fixtures in tests/golden/), analytic known-answer
 * tests, and the S1..S6 identities of SURVEY.md section 8(a) (tests/test_oracle.py).
 *
 * The file is compiled twice (see oracle/Makefile): REAL=float gives the arithmetic of the
 * reference (fp32 data, fp32 twiddles from sincosf like --use_fast_math code would), REAL=double
 * gives an accuracy oracle with the same control flow.  Complex data is interleaved (re,im),
 * batches are contiguous: FFT f occupies elements [f*N, (f+1)*N).
 *
 * Reference files followed (paths relative to the reference checkout):
 *   CT = SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu, ST = SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu,
 *   RC = SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUF
#define SUF _f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef struct { REAL x, y; } cplx;

/* W_N^m = exp(-/+ 2*pi*i*m/N).  CT:18-28 (Get_W_value / Get_W_value_inverse): the reference
 * evaluates sincosf(-/+6.283185308f * m/N) in fp32; the double build uses the exact angle. */
static cplx twiddle(int N, int m, int inverse) {
    cplx w;
    if (sizeof(REAL) == sizeof(float)) {
        float a = (inverse ? 6.283185308f : -6.283185308f) * ((float)m / (float)N);
        w.x = (REAL)cosf(a);
        w.y = (REAL)sinf(a);
    } else {
        double a = (inverse ? 2.0 : -2.0) * M_PI * (double)m / (double)N;
        w.x = (REAL)cos(a);
        w.y = (REAL)sin(a);
    }
    return w;
}

static unsigned bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

static int ilog2(int n) { int e = 0; while ((1 << e) < n) ++e; return e; }

/* ---------------------------------------------------------------------------------------------
 * Cooley-Tukey radix-2 DIT, one FFT in place.  CT:334-532 (do_SMFFT_CT_DIT).
 *  - reorder != 0: the block's data is first permuted to bit-reversed order (the net effect of
 *    reorder_32..reorder_4096, CT:126-329), so the result is the DFT
 *    of the input in natural order (semantics S1).
 *  - reorder == 0: the butterfly network runs on the natural-order input, which equals
 *    DFT(in[bitrev(n)]) (semantics S2).
 *  Stage q (span P = 2^q): (a, b) <- (a + W b, a - W b) with W = W_{2P}^{m}, m = index mod P.
 *  Register stages CT:364-411, shared-memory stages CT:456-490, last stage CT:493-531 (its second
 *  butterfly uses W^{m+N/4} = -/+ i W^m, which is the same twiddle value).  Un-normalised both ways.
 * ------------------------------------------------------------------------------------------- */
static void ct_dit_one(cplx* a, int N, int inverse, int reorder) {
    const int e = ilog2(N);
    if (reorder) {
        for (unsigned i = 0; i < (unsigned)N; ++i) {
            unsigned j = bitrev(i, e);
            if (j > i) { cplx t = a[i]; a[i] = a[j]; a[j] = t; }
        }
    }
    for (int q = 0; q < e; ++q) {
        const int P = 1 << q;
        for (int base = 0; base < N; base += 2 * P) {
            for (int m = 0; m < P; ++m) {
                cplx W = twiddle(2 * P, m, inverse);
                cplx A = a[base + m], B = a[base + m + P];
                REAL tx = W.x * B.x - W.y * B.y;
                REAL ty = W.x * B.y + W.y * B.x;
                a[base + m].x = A.x + tx;     a[base + m].y = A.y + ty;
                a[base + m + P].x = A.x - tx; a[base + m + P].y = A.y - ty;
            }
        }
    }
}

/* Batched CT C2C.  Mirrors SMFFT_DIT_external (CT:534-551): out-of-place, input untouched. */
void FN(oracle_ct_c2c)(const REAL* in, REAL* out, int N, long nFFTs, int inverse, int reorder) {
    memcpy(out, in, (size_t)nFFTs * N * sizeof(cplx));
#pragma omp parallel for schedule(static)
    for (long f = 0; f < nFFTs; ++f) ct_dit_one((cplx*)out + f * N, N, inverse, reorder);
}

/* ---------------------------------------------------------------------------------------------
 * Stockham autosort radix-2, one FFT.  ST:97-240 (do_FFT_Stockham_mk6, sign +) and RC:106-266
 * (do_FFT_Stockham_C2C, sign by direction).  Stage r = 1..log2 N, butterfly i in [0, N/2)
 * (the reference's thread t handles i = t and i = t + N/4):
 *     j = i >> (r-1), k = i & (2^(r-1) - 1), W = W_{2^r}^{k}
 *     out[j*2^r + k]           = in[i] + W * in[i + N/2]
 *     out[j*2^r + k + 2^(r-1)] = in[i] - W * in[i + N/2]
 * The reference reads everything, __syncthreads, then writes (ST:141-146), which a sequential
 * restatement reproduces with a scratch copy per stage.  Natural order in and out.
 * ------------------------------------------------------------------------------------------- */
static void stockham_one(cplx* a, cplx* tmp, int N, int inverse) {
    const int e = ilog2(N);
    const int half = N / 2;
    for (int r = 1; r <= e; ++r) {
        const int PoT = 1 << r, PoTm1 = PoT >> 1;
        memcpy(tmp, a, (size_t)N * sizeof(cplx));
        for (int i = 0; i < half; ++i) {
            int j = i >> (r - 1), k = i & (PoTm1 - 1);
            cplx W = twiddle(PoT, k, inverse);
            cplx A = tmp[i], B = tmp[i + half];
            REAL tx = W.x * B.x - W.y * B.y;
            REAL ty = W.x * B.y + W.y * B.x;
            a[j * PoT + k].x = A.x + tx;         a[j * PoT + k].y = A.y + ty;
            a[j * PoT + k + PoTm1].x = A.x - tx; a[j * PoT + k + PoTm1].y = A.y - ty;
        }
    }
}

/* Batched Stockham C2C.  inverse=1 is the ST program's only mode (ST:76, compared against
 * CUFFT_INVERSE at ST:429); inverse=0 is RC's FFT_forward helper. */
void FN(oracle_st_c2c)(const REAL* in, REAL* out, int N, long nFFTs, int inverse) {
    memcpy(out, in, (size_t)nFFTs * N * sizeof(cplx));
#pragma omp parallel
    {
        cplx* tmp = (cplx*)malloc((size_t)N * sizeof(cplx));
#pragma omp for schedule(static)
        for (long f = 0; f < nFFTs; ++f) stockham_one((cplx*)out + f * N, tmp, N, inverse);
        free(tmp);
    }
}

/* ---------------------------------------------------------------------------------------------
 * R2C / C2R of real length N through a complex FFT of length L = N/2.  RC:269-344
 * (do_FFT_Stockham_R2C_C2R).  s holds L complex values; forward: s = (x[2j], x[2j+1]) pairs
 * (pointer cast at RC:406), result packed as s[0] = (X[0].re, X[N/2].re), s[k] = X[k] (S5).
 * Inverse takes that packed layout and leaves (N/2) * x as interleaved reals (S6).
 * ------------------------------------------------------------------------------------------- */
static void r2c_c2r_one(cplx* s, cplx* tmp, int L, int inverse) {
    REAL ohx, ohy;
    if (!inverse) {
        ohx = (REAL)0.5; ohy = (REAL)-0.5;                  /* RC:272-275 */
        stockham_one(s, tmp, L, 0);
    } else {
        ohx = (REAL)-0.5; ohy = (REAL)0.5;                  /* RC:277-286 */
        cplx L0 = s[0];
        s[0].x = (REAL)0.5 * (L0.x + L0.y);
        s[0].y = (REAL)0.5 * (L0.x - L0.y);
    }
    /* RC:289-328: thread t and off in {0, L/4} cover i = t+1+off = 1 .. L/2.  For i = L/2 both
     * operands are the same element and F2 is the value that stays (written last, RC:308). */
    for (int i = 1; i <= L / 2; ++i) {
        cplx A = s[i], B = s[L - i];
        REAL h1x = (REAL)0.5 * (A.x + B.x);
        REAL h1y = (REAL)0.5 * (A.y - B.y);
        REAL h2x = ohx * (A.y + B.y);
        REAL h2y = ohy * (A.x - B.x);
        cplx W = twiddle(2 * L, i, inverse);
        cplx F1, F2;
        F1.x = h1x + W.x * h2x - W.y * h2y;
        F1.y = h1y + W.x * h2y + W.y * h2x;
        F2.x = h1x - W.x * h2x + W.y * h2y;
        F2.y = -h1y + W.x * h2y + W.y * h2x;
        s[i] = F1;
        s[L - i] = F2;
    }
    if (!inverse) {
        cplx L0 = s[0];                                      /* RC:332-339 */
        s[0].x = L0.x + L0.y;
        s[0].y = L0.x - L0.y;
    } else {
        stockham_one(s, tmp, L, 1);                          /* RC:342 */
    }
}

/* Batched R2C (inverse=0: in = nFFTs*N reals, out = nFFTs*N/2 packed complex) or
 * C2R (inverse=1: in = packed complex, out = reals, un-normalised: (N/2)*x).
 * Mirrors FFT_GPU_R2C_C2R_external (RC:349-365): N/2 float2 in, N/2 float2 out. */
void FN(oracle_r2c_c2r)(const REAL* in, REAL* out, int N, long nFFTs, int inverse) {
    const int L = N / 2;
    memcpy(out, in, (size_t)nFFTs * L * sizeof(cplx));
#pragma omp parallel
    {
        cplx* tmp = (cplx*)malloc((size_t)L * sizeof(cplx));
#pragma omp for schedule(static)
        for (long f = 0; f < nFFTs; ++f) r2c_c2r_one((cplx*)out + f * L, tmp, L, inverse);
        free(tmp);
    }
}

#ifdef ORACLE_WITH_COMPARATORS
/* ---------------------------------------------------------------------------------------------
 * The harness's comparison metric, restated so the tests can report it next to the fp64 gate.
 * CT/FFT.c:23-49 (get_error), :52-77 (Compare_data); RC/FFT.c:67-95, :126-185.
 * ------------------------------------------------------------------------------------------- */
float oracle_get_error(float A, float B) {
    float div_error, order;
    int power;
    if (A < 0) A = -A;
    if (B < 0) B = -B;
    if (A > B) {
        div_error = A - B;
        if (B > 10) { power = (int)log10(B); order = (float)pow(10, power); div_error = div_error / order; }
    } else {
        div_error = B - A;
        if (A > 10) { power = (int)log10(A); order = (float)pow(10, power); div_error = div_error / order; }
    }
    return div_error < 10000.0f ? div_error : 10000.0f;
}

/* Compare_data: counts elements whose max(re-error, im-error) exceeds max_error. */
long oracle_compare_data(const float* ref, const float* got, int N, long nFFTs, double max_error,
                         double* cumulative_error, double* mean_error) {
    long nErrors = 0;
    double acc = 0;
    for (long p = 0; p < nFFTs * (long)N; ++p) {
        float er = oracle_get_error(ref[2 * p], got[2 * p]);
        float ei = oracle_get_error(ref[2 * p + 1], got[2 * p + 1]);
        double e = er >= ei ? er : ei;
        if (e > max_error) ++nErrors;
        acc += e;
    }
    if (cumulative_error) *cumulative_error = acc;
    if (mean_error) *mean_error = acc / (double)((long)N * nFFTs);
    return nErrors;
}

/* RC/FFT.c:67-95: the float2 overload compares only max(x, y) of each operand. */
static float get_error_f2(float ax, float ay, float bx, float by) {
    float A = ax > ay ? ax : ay, B = bx > by ? bx : by;
    return oracle_get_error(A, B);
}

/* Compare_R2C_output (RC/FFT.c:126-159): got = packed N/2 per FFT, ref = N/2+1 per FFT. */
long oracle_compare_r2c(const float* got, const float* ref, int N, long nFFTs, double max_error) {
    long nErrors = 0;
    const int cs = N / 2 + 1, ks = N / 2;
    for (long f = 0; f < nFFTs; ++f) {
        const float* k = got + 2 * f * ks;
        const float* c = ref + 2 * f * cs;
        if (get_error_f2(k[0], k[1], c[0], c[2 * (cs - 1)]) > max_error) ++nErrors;
        for (int i = 1; i < ks; ++i)
            if (get_error_f2(k[2 * i], k[2 * i + 1], c[2 * i], c[2 * i + 1]) > max_error) ++nErrors;
    }
    return nErrors;
}

/* Compare_C2R_output (RC/FFT.c:161-185): got/(N/2) against ref/N. */
long oracle_compare_c2r(const float* got, const float* ref, int N, long nFFTs, double max_error) {
    long nErrors = 0;
    for (long p = 0; p < nFFTs * (long)N; ++p)
        if (oracle_get_error(got[p] / (float)(N >> 1), ref[p] / (float)N) > max_error) ++nErrors;
    return nErrors;
}
#endif
