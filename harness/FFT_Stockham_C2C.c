// Harness of the Stockham C2C program: FFT.exe <FFT length> <nFFTs> <nRuns>
// Behaviour of SMFFT_Stockham_C2C/FFT.c:84-153 (prototypes FFT.c:79-81): inverse-sign transform
// checked against the vendor library's inverse C2C.
#include "harness_common.h"

int GPU_cuFFT(float2 *h_input, float2 *h_output, int FFT_size, int nFFTs, int nRuns, double *single_ex_time);
int GPU_FFT_C2C_Stockham(float2 *h_input, float2 *h_smFFT_output, int FFT_size, int nFFTs, int nRuns, double *single_ex_time, double *multi_ex_time);

static const char *usage =   // the reference's text (ST/FFT.c:85-92)
	"Argument error!\n"
	" 1) FFT length\n"
	" 2) number of FFTs\n"
	" 3) the number of kernel executions\n"
	"For example: FFT.exe 1024 100000 20\n";

int main(int argc, char **argv) {
	long arg[3];
	if (!harness_parse_ints(argc, argv, 3, arg, usage)) return 1;
	const int FFT_size = (int) arg[0], nFFTs = (int) arg[1], nRuns = (int) arg[2];
	size_t count = (size_t) nFFTs*FFT_size;
	if (DEBUG) printf("FFT size: %d; Number of FFTs: %d; input size: %zu elements = %0.3f MB; output size: %zu elements = %0.3f\n", FFT_size, nFFTs, count, count*sizeof(float2)/(1024.0*1024.0), count, count*sizeof(float2)/(1024.0*1024.0));
	if (FFT_size < 128) { printf("This FFT implementation works for N>=128.\n"); return 1; }

	float2 *h_input        = (float2 *) calloc(count, sizeof(float2));
	float2 *h_smFFT_output = (float2 *) calloc(count, sizeof(float2));
	float2 *h_cuFFT_output = (float2 *) calloc(count, sizeof(float2));
	if (!h_input || !h_smFFT_output || !h_cuFFT_output) { printf("Host memory allocation failed.\n"); return 1; }

	harness_seed();
	harness_fill_uniform((float *) h_input, 2*count);   // U[0,1) re and im (FFT.c:141-142)

	double cuFFT_execution_time, smFFT_execution_time, smFFT_multiple_execution_time;
	GPU_cuFFT(h_input, h_cuFFT_output, FFT_size, nFFTs, nRuns, &cuFFT_execution_time);
	GPU_FFT_C2C_Stockham(h_input, h_smFFT_output, FFT_size, nFFTs, nRuns, &smFFT_execution_time, &smFFT_multiple_execution_time);

	double cumulative_error, mean_error;
	const int nErrors = Compare_data(h_cuFFT_output, h_smFFT_output, FFT_size, nFFTs, &cumulative_error, &mean_error);
	print_verdict(nErrors);
	if (nErrors > 0) harness_attribute(h_input, h_cuFFT_output, h_smFFT_output, FFT_size, 1, false);   // (the Stockham program is the + sign transform, ST:76; an extension: upstream stops at the verdict)

	free(h_input); free(h_smFFT_output); free(h_cuFFT_output);   // (upstream delete[]s malloc'ed memory, FFT.c:146-148)
	(void) hipDeviceReset();
	return 0;
}
