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
#include <atomic>
#include <thread>
#include <vector>

#ifndef DEBUG
#define DEBUG true
#endif

static double max_error = 1.0e-4;   // CT/FFT.c:12

// Upstream seeds with time(NULL) (CT/FFT.c:139); SMFFT_SEED makes runs reproducible.
static unsigned long long harness_seed_value = 0;
static bool harness_libc_rand = false;
static inline void harness_seed(void) {
	const char *e = getenv("SMFFT_SEED");
	unsigned seed = e ? (unsigned) strtoul(e, NULL, 10) : (unsigned) time(NULL);
	srand(seed);
	harness_seed_value = seed;
	e = getenv("SMFFT_LIBC_RAND");
	harness_libc_rand = e && atoi(e) != 0;
}

// Host data path: at the README batch upstream spends 10-20 s in 2^30 serial rand()
// calls and another 5-10 s in the serial comparison, against 1.4 ms of transform.  Element i of a fill is
// U[0,1) from a counter-based generator (splitmix64 of seed and index), so the data do not depend on the
// number of threads, and fills and comparisons run on all host cores.  SMFFT_LIBC_RAND=1 restores the
// serial libc rand() stream of the reference.
static inline float harness_uniform(unsigned long long i) {
	unsigned long long z = (harness_seed_value + 1) * 0xD1342543DE82EF95ull + (i + 1) * 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	z ^= z >> 31;
	return (float) (z >> 40) * (1.0f / 16777216.0f);
}
static inline float harness_next_uniform(unsigned long long i) {   // i is ignored by the libc stream
	return harness_libc_rand ? rand()/(float) RAND_MAX : harness_uniform(i);
}

// f(first, last, chunk) over [0, n) cut into a FIXED number of chunks (results that are reduced per chunk do
// not depend on the thread count), chunks dealt to min(hardware threads, 32) std::threads.
#define HARNESS_CHUNKS 256
template <class F>
static inline void harness_parallel_chunks(size_t n, F f) {
	unsigned nthreads = std::thread::hardware_concurrency();
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 32) nthreads = 32;
	if (n < (1u << 20)) nthreads = 1;
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (int c = next++; c < HARNESS_CHUNKS; c = next++) {
			size_t a = n / HARNESS_CHUNKS * c + (n % HARNESS_CHUNKS < (size_t) c ? n % HARNESS_CHUNKS : (size_t) c);
			size_t b = a + n / HARNESS_CHUNKS + ((size_t) c < n % HARNESS_CHUNKS ? 1 : 0);
			if (b > a) f(a, b, c);
		}
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(worker);
	worker();
	for (auto &t : pool) t.join();
}

// n floats of U[0,1) (for float2 arrays: .x and .y of element k are floats 2k and 2k+1)
static inline void harness_fill_uniform(float *dst, size_t n) {
	if (harness_libc_rand) {
		for (size_t i = 0; i + 1 < n; i += 2) {   // .y before .x, as CT/FFT.c:141-142
			dst[i + 1] = rand()/(float) RAND_MAX;
			dst[i] = rand()/(float) RAND_MAX;
		}
		if (n & 1) dst[n - 1] = rand()/(float) RAND_MAX;
		return;
	}
	harness_parallel_chunks(n, [&](size_t a, size_t b, int) { for (size_t i = a; i < b; i++) dst[i] = harness_uniform(i); });
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

// CT/FFT.c:52-77 (same metric and counts; evaluated per chunk on all cores, partial sums added in chunk order)
static inline int Compare_data(float2 *vendor_result, float2 *smFFT_result, int FFT_size, int nFFTs, double *cumulative_error, double *mean_error) {
	static int errors[HARNESS_CHUNKS];
	static double sums[HARNESS_CHUNKS];
	for (int c = 0; c < HARNESS_CHUNKS; c++) { errors[c] = 0; sums[c] = 0; }
	harness_parallel_chunks((size_t) FFT_size*nFFTs, [&](size_t a, size_t b, int c) {
		int nErrors = 0;
		double sum = 0;
		for (size_t pos = a; pos < b; pos++) {
			float er = get_error(vendor_result[pos].x, smFFT_result[pos].x);
			float ei = get_error(vendor_result[pos].y, smFFT_result[pos].y);
			double e = (er >= ei ? er : ei);
			if (e > max_error) nErrors++;
			sum += e;
		}
		errors[c] = nErrors;
		sums[c] = sum;
	});
	int nErrors = 0;
	double sum = 0;
	for (int c = 0; c < HARNESS_CHUNKS; c++) { nErrors += errors[c]; sum += sums[c]; }
	*cumulative_error = sum;
	*mean_error = sum/((double) FFT_size*nFFTs);
	return nErrors;
}

// positional integer arguments; prints `usage` and returns false unless exactly n are given
static inline bool harness_parse_ints(int argc, char **argv, int n, long *out, const char *usage) {
	if (argc != n + 1) { fputs(usage, stdout); return false; }
	for (int i = 0; i < n; i++) out[i] = strtol(argv[i + 1], NULL, 10);
	return true;
}

static inline void print_verdict(int nErrors) {   // CT/FFT.c:158-159
	if (nErrors == 0) printf("  FFT test:\033[1;32mPASSED\033[0m\n");
	else printf("  FFT test:\033[1;31mFAILED\033[0m\n");
}
#endif
