// smfft_vendor.hip -- the vendor-library comparator of the harness (libsmfft_vendor.so).
//
// The reference times cuFFT next to smFFT and uses it as its only correctness oracle
// (GPU_cuFFT CT:758-825, ST:389-454; GPU_cuFFT_R2C / _C2R RC:471-567).  The AMD counterpart is
// hipFFT (rocFFT underneath): same plan shapes (hipfftPlan1d(N, C2C|R2C|C2R, batch)), one timed
// exec for C2C (the reference ignores nRuns there, CT:798-802), nRuns execs inside one timing for
// R2C/C2R (RC:501-506).  Kept in its own library so libsmfft_amd.so has no vendor dependency.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>

#include <dlfcn.h>
#include <cstdlib>

#include "../../include/smfft_reference_api.h"
#include "smfft_host_util.hpp"

// Both libraries of a harness run are timed on the same KIND of buffers -- and, sizes permitting, on the SAME pair: the
// comparator takes its two buffers from smfft_malloc_pair_for_wrapper like the smFFT wrappers (the pair it releases is
// the one the smFFT wrapper then gets), sized for the larger of its two sides (R2C / C2R: (N/2 + 1) * nFFTs complex against
// N * nFFTs real); two plain allocations for both with SMFFT_WRAPPER_PLACEMENT=0.  libsmfft_amd.so is looked up at run time
// so this library keeps no link dependency; if it is not visible (loaded RTLD_LOCAL) the buffers are plain and the
// printed line says so.
struct VendorPair {
    void *in = nullptr, *out = nullptr;
    bool placed = false;
};
static VendorPair vendor_alloc(size_t in_bytes, size_t out_bytes) {
    VendorPair p;
    const char* e = getenv("SMFFT_WRAPPER_PLACEMENT");
    if (!(e && atoi(e) == 0)) {
        typedef int (*pair_fn)(unsigned long long, void**, void**);
        pair_fn f = (pair_fn)dlsym(RTLD_DEFAULT, "smfft_malloc_pair_for_wrapper");
        if (f && f(in_bytes > out_bytes ? in_bytes : out_bytes, &p.in, &p.out) == 0) { p.placed = true; return p; }
        printf("  hipFFT comparator: libsmfft_amd.so's allocator is not visible, timing on two plain allocations\n");
    }
    checkHipErrors(hipMalloc(&p.in, in_bytes));
    checkHipErrors(hipMalloc(&p.out, out_bytes));
    return p;
}
static void vendor_free(VendorPair& p) {
    if (p.placed) {
        typedef int (*free_fn)(void*);
        free_fn f = (free_fn)dlsym(RTLD_DEFAULT, "smfft_free_pair");
        if (f) { f(p.in); return; }
    }
    checkHipErrors(hipFree(p.in));
    checkHipErrors(hipFree(p.out));
}

static int vendor_c2c(float2* h_input, float2* h_output, int FFT_size, int nFFTs, bool inverse, double* single_ex_time) {
    size_t free_mem, total_mem;
    checkHipErrors(hipMemGetInfo(&free_mem, &total_mem));
    const size_t bytes = (size_t)FFT_size * nFFTs * sizeof(float2);
    if (2 * bytes > free_mem) {
        printf("Error: Not enough memory! Input data are too big for the device.\n");
        return 1;
    }
    VendorPair pair = vendor_alloc(bytes, bytes);
    float2 *d_input = (float2*)pair.in, *d_output = (float2*)pair.out;
    checkHipErrors(hipMemcpy(d_input, h_input, bytes, hipMemcpyHostToDevice));
    double time_vendor = 0;
    GpuTimer timer;
    hipfftHandle plan;
    hipfftResult error = hipfftPlan1d(&plan, FFT_size, HIPFFT_C2C, nFFTs);
    if (HIPFFT_SUCCESS != error) printf("HIPFFT error: %d", error);
    // one untimed exec first: rocFFT compiles its kernel on first use, which cuFFT does not
    hipfftExecC2C(plan, (hipfftComplex*)d_input, (hipfftComplex*)d_output, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD);
    checkHipErrors(hipDeviceSynchronize());
    timer.Start();
    hipfftExecC2C(plan, (hipfftComplex*)d_input, (hipfftComplex*)d_output, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD);
    timer.Stop();
    time_vendor += timer.Elapsed();
    hipfftDestroy(plan);
    if (single_ex_time) *single_ex_time = time_vendor;
    printf("  FFT size: %d; cuFFT time = %0.3f ms;\n", FFT_size, time_vendor);   // (hipFFT; label kept for scripts that parse it)
    checkHipErrors(hipDeviceSynchronize());
    checkHipErrors(hipMemcpy(h_output, d_output, bytes, hipMemcpyDeviceToHost));
    checkHipErrors(hipGetLastError());
    vendor_free(pair);
    return 0;
}

int GPU_cuFFT(float2* h_input, float2* h_output, int FFT_size, int nFFTs, bool inverse, int /*nRuns*/, double* single_ex_time) {
    return vendor_c2c(h_input, h_output, FFT_size, nFFTs, inverse, single_ex_time);
}
// Stockham program: compared against the INVERSE vendor transform (ST:429)
int GPU_cuFFT(float2* h_input, float2* h_output, int FFT_size, int nFFTs, int /*nRuns*/, double* single_ex_time) {
    return vendor_c2c(h_input, h_output, FFT_size, nFFTs, true, single_ex_time);
}

int GPU_cuFFT_R2C(float2* h_output, float* h_input, int FFT_size, int nFFTs, int nRuns) {
    const size_t in_bytes = (size_t)FFT_size * nFFTs * sizeof(float);
    const size_t out_bytes = (size_t)((FFT_size >> 1) + 1) * nFFTs * sizeof(float2);
    float* d_input;
    float2* d_output;
    VendorPair pair = vendor_alloc(in_bytes, out_bytes);
    d_input = (decltype(d_input))pair.in;
    d_output = (decltype(d_output))pair.out;
    checkHipErrors(hipMemcpy(d_input, h_input, in_bytes, hipMemcpyHostToDevice));
    hipfftHandle plan;
    hipfftResult error = hipfftPlan1d(&plan, FFT_size, HIPFFT_R2C, nFFTs);
    if (HIPFFT_SUCCESS != error) printf("HIPFFT error: %d", error);
    hipfftExecR2C(plan, d_input, (hipfftComplex*)d_output);
    checkHipErrors(hipDeviceSynchronize());
    GpuTimer timer;
    timer.Start();
    for (int f = 0; f < nRuns; f++) hipfftExecR2C(plan, d_input, (hipfftComplex*)d_output);
    timer.Stop();
    printf("  cuFFT R2C time: %0.3f ms\n", timer.Elapsed() / nRuns);
    hipfftDestroy(plan);
    checkHipErrors(hipMemcpy(h_output, d_output, out_bytes, hipMemcpyDeviceToHost));
    vendor_free(pair);
    return 0;
}

int GPU_cuFFT_C2R(float* h_output, float2* h_input, int FFT_size, int nFFTs, int nRuns) {
    const size_t in_bytes = (size_t)((FFT_size >> 1) + 1) * nFFTs * sizeof(float2);
    const size_t out_bytes = (size_t)FFT_size * nFFTs * sizeof(float);
    float2* d_input;
    float* d_output;
    VendorPair pair = vendor_alloc(in_bytes, out_bytes);
    d_input = (decltype(d_input))pair.in;
    d_output = (decltype(d_output))pair.out;
    hipfftHandle plan;
    hipfftResult error = hipfftPlan1d(&plan, FFT_size, HIPFFT_C2R, nFFTs);
    if (HIPFFT_SUCCESS != error) printf("HIPFFT error: %d", error);
    // C2R may overwrite its input: re-upload before every exec
    checkHipErrors(hipMemcpy(d_input, h_input, in_bytes, hipMemcpyHostToDevice));
    hipfftExecC2R(plan, (hipfftComplex*)d_input, d_output);
    checkHipErrors(hipDeviceSynchronize());
    double total = 0;
    for (int f = 0; f < nRuns; f++) {
        checkHipErrors(hipMemcpy(d_input, h_input, in_bytes, hipMemcpyHostToDevice));
        GpuTimer timer;
        timer.Start();
        hipfftExecC2R(plan, (hipfftComplex*)d_input, d_output);
        timer.Stop();
        total += timer.Elapsed();
    }
    printf("  cuFFT C2R time: %0.3f ms\n", total / nRuns);
    hipfftDestroy(plan);
    checkHipErrors(hipMemcpy(h_output, d_output, out_bytes, hipMemcpyDeviceToHost));
    vendor_free(pair);
    return 0;
}
