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

// CT/FFT.c:52-77 (same metric and counts; evaluated per chunk on all cores, partial sums added in chunk order).
// The element with the largest error is remembered (harness_worst_pos) for harness_attribute below.
static size_t harness_worst_pos = 0;
static double harness_worst_error = -1;
static inline int Compare_data(float2 *vendor_result, float2 *smFFT_result, int FFT_size, int nFFTs, double *cumulative_error, double *mean_error) {
	static int errors[HARNESS_CHUNKS];
	static double sums[HARNESS_CHUNKS], worst[HARNESS_CHUNKS];
	static size_t worst_at[HARNESS_CHUNKS];
	for (int c = 0; c < HARNESS_CHUNKS; c++) { errors[c] = 0; sums[c] = 0; worst[c] = -1; worst_at[c] = 0; }
	harness_parallel_chunks((size_t) FFT_size*nFFTs, [&](size_t a, size_t b, int c) {
		int nErrors = 0;
		double sum = 0, w = -1;
		size_t wa = a;
		for (size_t pos = a; pos < b; pos++) {
			float er = get_error(vendor_result[pos].x, smFFT_result[pos].x);
			float ei = get_error(vendor_result[pos].y, smFFT_result[pos].y);
			double e = (er >= ei ? er : ei);
			if (e > max_error) nErrors++;
			if (e > w) { w = e; wa = pos; }
			sum += e;
		}
		errors[c] = nErrors;
		sums[c] = sum;
		worst[c] = w;
		worst_at[c] = wa;
	});
	int nErrors = 0;
	double sum = 0;
	harness_worst_error = -1;
	for (int c = 0; c < HARNESS_CHUNKS; c++) {
		nErrors += errors[c];
		sum += sums[c];
		if (worst[c] > harness_worst_error) { harness_worst_error = worst[c]; harness_worst_pos = worst_at[c]; }
	}
	*cumulative_error = sum;
	*mean_error = sum/((double) FFT_size*nFFTs);
	return nErrors;
}

// WHO IS OFF when the metric above reports errors?  Upstream's check is a comparison of two fp32 results under an ABSOLUTE
// bound (max_error = 1e-4 on U[0,1) data, CT/FFT.c:12,23-49): at N >= 2048 the bins are large enough (DC ~ N/2) for the fp32
// round-off of either side to exceed it, and the verdict says FAILED without saying whose.  This prints, for the worst element,
// both values against the un-normalised DFT of the same input evaluated on the host in fp64 (one bin: N terms), and each
// side's distance from it relative to the largest bin of that FFT -- the measure this library's own tests gate on (1e-6).
// sign: -1 forward, +1 inverse; bitrev: the transform of the bit-reversed input (reorder = 0).
static inline void harness_dft_bin(const float2 *x, int N, int k, int sign, bool bitrev, double *re, double *im) {
	int bits = 0;
	while ((1 << bits) < N) bits++;
	const double w = sign*2.0*M_PI/N;
	double sr = 0, si = 0;
	for (int n = 0; n < N; n++) {
		int src = n;
		if (bitrev) { src = 0; for (int b = 0; b < bits; b++) src |= ((n >> b) & 1) << (bits - 1 - b); }
		const double c = cos(w*(double) ((long) n*k % N)), s = sin(w*(double) ((long) n*k % N));
		sr += x[src].x*c - x[src].y*s;
		si += x[src].x*s + x[src].y*c;
	}
	*re = sr; *im = si;
}
static inline void harness_attribute(const float2 *h_input, const float2 *vendor, const float2 *smFFT, int FFT_size, int sign, bool bitrev) {
	const size_t f = harness_worst_pos/FFT_size;
	const int k = (int) (harness_worst_pos % FFT_size);
	const float2 *x = h_input + f*FFT_size;
	double re, im, dre, dim;
	harness_dft_bin(x, FFT_size, k, sign, bitrev, &re, &im);
	harness_dft_bin(x, FFT_size, 0, sign, bitrev, &dre, &dim);     // U[0,1) data: the DC bin is the largest
	double largest = sqrt(dre*dre + dim*dim);
	const double here = sqrt(re*re + im*im);
	if (here > largest) largest = here;
	const float2 a = smFFT[harness_worst_pos], b = vendor[harness_worst_pos];
	const double ea = sqrt((a.x - re)*(a.x - re) + (a.y - im)*(a.y - im)), eb = sqrt((b.x - re)*(b.x - re) + (b.y - im)*(b.y - im));
	printf("  Worst element (error %g by the metric above): FFT %zu, bin %d: smFFT (%.9g, %.9g), vendor FFT (%.9g, %.9g), fp64 DFT of the same input (%.12g, %.12g)\n",
	       harness_worst_error, f, k, a.x, a.y, b.x, b.y, re, im);
	printf("  Distance from the fp64 DFT, relative to the largest bin of that FFT (%.6g): smFFT %.3e, vendor FFT %.3e  (fp32 round-off of a %d-point transform is ~1e-7; this library's tests gate on 1e-6)\n",
	       largest, ea/largest, eb/largest, FFT_size);
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
