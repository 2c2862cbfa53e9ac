// utils_hip.h -- abort-on-error macro with the behaviour of the reference's checkCudaErrors
// (SMFFT_CooleyTukey_C2C/utils_cuda.h:12-22): print file:line + the runtime's message, exit(1).
#ifndef SMFFT_UTILS_HIP_H__
#define SMFFT_UTILS_HIP_H__
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>

#define checkHipErrors(val) smfft_check((val), #val, __FILE__, __LINE__)

static inline void smfft_check(hipError_t err, const char* const func, const char* const file, const int line) {
	if (err != hipSuccess) {
		fprintf(stderr, "HIP error at: %s:%d\n%s %s\n", file, line, hipGetErrorString(err), func);
		exit(1);
	}
}
#endif
