/* cuda_runtime.h -- shim for compiling the reference's host harness (the FFT.c of each SMFFT program, compiled by
 * g++) against libsmfft_amd.so on ROCm.  FFT.c needs exactly two things from the CUDA headers: the float2 type of
 * its prototypes (FFT.c:80-81; HIP's float2 = HIP_vector_type<float, 2u> mangles identically in g++ and hipcc) and
 * cudaDeviceReset() at the end of main.  Put this directory first on the include path and define
 * __HIP_PLATFORM_AMD__. */
#ifndef SMFFT_SHIM_CUDA_RUNTIME_H_
#define SMFFT_SHIM_CUDA_RUNTIME_H_
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#define cudaDeviceReset hipDeviceReset
#endif
