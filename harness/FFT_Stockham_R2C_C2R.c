// Harness of the R2C/C2R program: FFT.exe <real FFT length> <nFFTs> <nRuns>
// Behaviour of SMFFT_Stockham_R2C_C2R/FFT.c:193-330 (prototypes FFT.c:188-191): R2C against the
// vendor R2C (N/2+1 bins, smFFT packs the Nyquist bin into element 0's imaginary part), then C2R
// on a random Hermitian-packable spectrum against the vendor C2R (smFFT/(N/2) vs vendor/N).
#include "harness_common.h"

int GPU_cuFFT_R2C(float2 *h_output, float *h_input, int FFT_size, int nFFTs, int nRuns);
int GPU_cuFFT_C2R(float *h_output, float2 *h_input, int FFT_size, int nFFTs, int nRuns);
int GPU_smFFT_R2C(float2 *h_output, float *h_input, int FFT_size, int nFFTs, int nRuns);
int GPU_smFFT_C2R(float *h_output, float2 *h_input, int FFT_size, int nFFTs, int nRuns);

// RC/FFT.c:126-159: element 0 of the packed smFFT output holds (DC, Nyquist)
static int Compare_R2C_output(float2 *kFFT, float2 *vendor, int FFT_size, int nFFTs) {
	std::atomic<int> nErrors(0), printed(0);
	const int vs = (FFT_size >> 1) + 1, ks = (FFT_size >> 1);
	harness_parallel_chunks((size_t) nFFTs, [&](size_t fa, size_t fb, int) {
		for (size_t f = fa; f < fb; f++) {
			float2 v0 = make_float2(vendor[f*vs].x, vendor[(f + 1)*vs - 1].x);
			for (int i = 0; i < ks; i++) {
				float2 k = kFFT[f*ks + i];
				float2 v = (i == 0 ? v0 : vendor[f*vs + i]);
				float error = get_error(k, v);
				if (error > max_error) {
					if (printed++ < 20) printf("FFT: %zu; element: %d; Error is [%f] value is [%f,%f] while it should be [%f,%f]\n", f, i, error, k.x, k.y, v.x, v.y);
					nErrors++;
				}
			}
		}
	});
	return nErrors;
}

// RC/FFT.c:161-185: smFFT/(N/2) against vendor/N
static int Compare_C2R_output(float *kFFT, float *vendor, int FFT_size, int nFFTs) {
	std::atomic<int> nErrors(0), printed(0);
	harness_parallel_chunks((size_t) nFFTs*FFT_size, [&](size_t a, size_t b, int) {
		for (size_t pos = a; pos < b; pos++) {
			float k = kFFT[pos]/(FFT_size >> 1), v = vendor[pos]/FFT_size;
			float error = get_error(k, v);
			if (error > max_error) {
				if (printed++ < 20) printf("element: %zu; Error is [%f] kFFT value is [%f] while it should be [%f]\n", pos, error, k, v);
				nErrors++;
			}
		}
	});
	return nErrors;
}

static const char *usage =   // the reference's text (RC/FFT.c:194-201)
	"Argument error!\n"
	" 1) FFT length\n"
	" 2) number of FFTs\n"
	" 3) the number of kernel executions\n"
	"For example: FFT.exe 1024 100000 20\n";

int main(int argc, char **argv) {
	long arg[3];
	if (!harness_parse_ints(argc, argv, 3, arg, usage)) return 1;
	const int FFT_size = (int) arg[0], nFFTs = (int) arg[1], nRuns = (int) arg[2];
	const int vs = (FFT_size >> 1) + 1, ks = (FFT_size >> 1);
	if (DEBUG) printf("FFT size: %d; Number of FFTs: %d; input size = %0.3f MB; output size = %0.3f MB\n", FFT_size, nFFTs, (size_t) nFFTs*FFT_size*sizeof(float)/(1024.0*1024.0), (size_t) nFFTs*vs*sizeof(float2)/(1024.0*1024.0));
	if (FFT_size < 128) { printf("This FFT works for N>=128.\n"); return 1; }

	float  *h_input_R2C      = (float *)  calloc((size_t) nFFTs*FFT_size, sizeof(float));
	float2 *h_input_C2R      = (float2 *) calloc((size_t) nFFTs*vs, sizeof(float2));
	float2 *h_input_C2R_kFFT = (float2 *) calloc((size_t) nFFTs*vs, sizeof(float2));
	float2 *h_kFFT_output          = (float2 *) calloc((size_t) nFFTs*vs, sizeof(float2));
	float  *h_kFFT_output_inverse  = (float *)  calloc((size_t) nFFTs*FFT_size, sizeof(float));
	float2 *h_cuFFT_output         = (float2 *) calloc((size_t) nFFTs*vs, sizeof(float2));
	float  *h_cuFFT_output_inverse = (float *)  calloc((size_t) nFFTs*FFT_size, sizeof(float));

	harness_seed();
	harness_fill_uniform(h_input_R2C, (size_t) nFFTs*FFT_size);
	// C2R inputs (FFT.c:264-283): the vendor layout has N/2+1 bins with real DC and Nyquist; the
	// smFFT layout packs the Nyquist value into element 0's imaginary part.
	const unsigned long long c2r_base = (unsigned long long) nFFTs*FFT_size;   // counter offset of this second data set
	auto fill_c2r = [&](size_t fa, size_t fb, int) {
		for (size_t f = fa; f < fb; f++) {
			const unsigned long long i0 = c2r_base + (unsigned long long) f*FFT_size;
			float nyquist = harness_next_uniform(i0);
			float dc = harness_next_uniform(i0 + 1);
			h_input_C2R[f*vs] = make_float2(dc, 0);
			h_input_C2R_kFFT[f*ks] = make_float2(dc, nyquist);
			for (int s = 1; s < ks; s++) {
				float re = harness_next_uniform(i0 + 2*s);
				float im = harness_next_uniform(i0 + 2*s + 1);
				h_input_C2R[f*vs + s] = make_float2(re, im);
				h_input_C2R_kFFT[f*ks + s] = make_float2(re, im);
			}
			h_input_C2R[f*vs + ks].x = nyquist;
		}
	};
	if (harness_libc_rand) fill_c2r(0, (size_t) nFFTs, 0);   // the serial libc stream keeps its order
	else harness_parallel_chunks((size_t) nFFTs, fill_c2r);

	GPU_cuFFT_R2C(h_cuFFT_output, h_input_R2C, FFT_size, nFFTs, nRuns);
	GPU_smFFT_R2C(h_kFFT_output,  h_input_R2C, FFT_size, nFFTs, nRuns);
	print_verdict(Compare_R2C_output(h_kFFT_output, h_cuFFT_output, FFT_size, nFFTs));

	GPU_cuFFT_C2R(h_cuFFT_output_inverse, h_input_C2R,      FFT_size, nFFTs, nRuns);
	GPU_smFFT_C2R(h_kFFT_output_inverse,  h_input_C2R_kFFT, FFT_size, nFFTs, nRuns);
	print_verdict(Compare_C2R_output(h_kFFT_output_inverse, h_cuFFT_output_inverse, FFT_size, nFFTs));

	free(h_input_R2C); free(h_input_C2R); free(h_input_C2R_kFFT);
	free(h_kFFT_output); free(h_kFFT_output_inverse); free(h_cuFFT_output); free(h_cuFFT_output_inverse);
	(void) hipDeviceReset();
	if (DEBUG) printf("\nFinished!\n");
	return 0;
}
