// smfft_reference_api.h -- the reference's own host prototypes (C++ linkage), exported verbatim by
// libsmfft_amd.so so that the reference's FFT.c harnesses (compiled by g++) link unchanged.
//
// The three upstream programs each declare their prototypes inside FFT.c
// (SMFFT_CooleyTukey_C2C/FFT.c:80-81, SMFFT_Stockham_C2C/FFT.c:79-81,
// SMFFT_Stockham_R2C_C2R/FFT.c:188-191) and define FFT_init / FFT_*_benchmark in the .cu files
// (CT:576,583,666; ST:299,306,348; RC:388,396,435).  All parameter lists are distinct, so the three
// families coexist as overloads in one library.  float2 is HIP's float2 (g++ and hipcc mangle it
// identically: P15HIP_vector_typeIfLj2EE).
#pragma once
#include <hip/hip_vector_types.h>

// ---- L2: device-pointer launch API ---------------------------------------------------------------
void FFT_init();
// Cooley-Tukey C2C (CT:583, :666)
int FFT_external_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, bool inverse, bool reorder, double* FFT_time);
int FFT_multiple_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, bool inverse, bool reorder, double* FFT_time);
// Stockham C2C (ST:306, :348)
void FFT_external_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, double* FFT_time);
void FFT_multiple_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, double* FFT_time);
// Stockham R2C/C2R (RC:396, :435)
void FFT_external_benchmark(float* d_input, float* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time);
void FFT_multiple_benchmark(float* d_input, float* d_output, int FFT_size, int nFFTs, double* FFT_time);

// ---- L3: host-pointer wrappers ---------------------------------------------------------------------
int GPU_smFFT_4elements(float2* h_input, float2* h_output, int FFT_size, int nFFTs, bool inverse, bool reorder, int nRuns, double* single_ex_time, double* multi_ex_time);  // CT:827
int GPU_FFT_C2C_Stockham(float2* h_input, float2* h_smFFT_output, int FFT_size, int nFFTs, int nRuns, double* single_ex_time, double* multi_ex_time);                    // ST:457
int GPU_smFFT_R2C(float2* h_output, float* h_input, int FFT_size, int nFFTs, int nRuns);  // RC:572
int GPU_smFFT_C2R(float* h_output, float2* h_input, int FFT_size, int nFFTs, int nRuns);  // RC:652

// ---- vendor comparator (the reference's cuFFT calls, here hipFFT/rocFFT; libsmfft_vendor.so) ---------
int GPU_cuFFT(float2* h_input, float2* h_output, int FFT_size, int nFFTs, bool inverse, int nRuns, double* single_ex_time);  // CT:758
int GPU_cuFFT(float2* h_input, float2* h_output, int FFT_size, int nFFTs, int nRuns, double* single_ex_time);               // ST:389 (inverse)
int GPU_cuFFT_R2C(float2* h_output, float* h_input, int FFT_size, int nFFTs, int nRuns);  // RC:471
int GPU_cuFFT_C2R(float* h_output, float2* h_input, int FFT_size, int nFFTs, int nRuns);  // RC:520
