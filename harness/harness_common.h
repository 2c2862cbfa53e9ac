// harness_common.h -- host-side helpers shared by the three harness programs (own code; behaviour
// follows the reference's FFT.c files, cited per function).
#ifndef SMFFT_HARNESS_COMMON_H_
#define SMFFT_HARNESS_COMMON_H_
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef DEBUG
#define DEBUG true
#endif

static double max_error = 1.0e-4;   // CT/FFT.c:12

// Upstream seeds with time(NULL) (CT/FFT.c:139); SMFFT_SEED makes runs reproducible.
static inline void harness_seed(void) {
	const char *e = getenv("SMFFT_SEED");
	srand(e ? (unsigned) strtoul(e, NULL, 10) : (unsigned) time(NULL));
}

// Error metric of the reference (CT/FFT.c:23-49): |A|,|B|, difference, divided by the decade of
// the smaller magnitude when that exceeds 10.
static inline float get_error(float A, float B) {
	float lo, hi;
	A = fabsf(A); B = fabsf(B);
	if (A > B) { hi = A; lo = B; } else { hi = B; lo = A; }
	float diff = hi - lo;
	if (lo > 10) diff = diff / (float) pow(10, (int) log10(lo));
	return diff < 10000.0f ? diff : 10000.0f;
}

// RC/FFT.c:67-95: the float2 flavour compares max(x,y) of each operand only.
static inline float get_error(float2 A, float2 B) {
	return get_error(A.x > A.y ? A.x : A.y, B.x > B.y ? B.x : B.y);
}

// CT/FFT.c:52-77
static inline int Compare_data(float2 *vendor_result, float2 *smFFT_result, int FFT_size, int nFFTs, double *cumulative_error, double *mean_error) {
	int nErrors = 0;
	double sum = 0;
	for (size_t pos = 0; pos < (size_t) FFT_size*nFFTs; pos++) {
		float er = get_error(vendor_result[pos].x, smFFT_result[pos].x);
		float ei = get_error(vendor_result[pos].y, smFFT_result[pos].y);
		double e = (er >= ei ? er : ei);
		if (e > max_error) nErrors++;
		sum += e;
	}
	*cumulative_error = sum;
	*mean_error = sum/((double) FFT_size*nFFTs);
	return nErrors;
}

static inline void print_verdict(int nErrors) {   // CT/FFT.c:158-159
	if (nErrors == 0) printf("  FFT test:\033[1;32mPASSED\033[0m\n");
	else printf("  FFT test:\033[1;31mFAILED\033[0m\n");
}
#endif
