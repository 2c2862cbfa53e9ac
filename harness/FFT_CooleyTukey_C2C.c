// Harness of the Cooley-Tukey C2C program: FFT.exe <FFT length> <nFFTs> <nRuns> <inverse 0|1> <reorder 0|1>
// Same command line, data distribution, checks and printed lines as SMFFT_CooleyTukey_C2C/FFT.c:83-172;
// it binds the same two prototypes (FFT.c:80-81), here provided by libsmfft_amd.so / libsmfft_vendor.so.
#include "harness_common.h"

int GPU_smFFT_4elements(float2 *h_input, float2 *h_output, int FFT_size, int nFFTs, bool inverse, bool reorder, int nRuns, double *single_ex_time, double *multi_ex_time);
int GPU_cuFFT(float2 *h_input, float2 *h_output, int FFT_size, int nFFTs, bool inverse, int nRuns, double *single_ex_time);

int main(int argc, char* argv[]) {
	if (argc != 6) {
		printf("Argument error!\n");
		printf(" 1) FFT length\n");
		printf(" 2) number of FFTs\n");
		printf(" 3) the number of kernel executions\n");
		printf(" 4) do inverse FFT 1=yes 0=no\n");
		printf(" 5) reorder elements to correct order 1=yes 0=no\n");
		printf("For example: FFT.exe 1024 100000 20 0 1\n");
		return 1;
	}
	int FFT_size = (int) strtol(argv[1], NULL, 10);
	int nFFTs    = (int) strtol(argv[2], NULL, 10);
	int nRuns    = (int) strtol(argv[3], NULL, 10);
	bool inverse = strtol(argv[4], NULL, 10) == 1;
	bool reorder = strtol(argv[5], NULL, 10) == 1;

	// FFT.c:105-116 keeps the batch a multiple of the FFTs-per-block of the CUDA kernels
	if (FFT_size == 32) {
		printf("FFT length is 32 making sure that the number of FFTs is divisible by 4. ");
		nFFTs = ((nFFTs + 3)/4)*4;
		printf("New number of FFTs is %d.\n", nFFTs);
	}
	if (FFT_size == 64) {
		printf("FFT length is 64 making sure that the number of FFTs is divisible by 2. ");
		nFFTs = ((nFFTs + 1)/2)*2;
		printf("New number of FFTs is %d.\n", nFFTs);
	}

	size_t count = (size_t) nFFTs*FFT_size;
	float2 *h_input        = (float2 *) calloc(count, sizeof(float2));
	float2 *h_output_smFFT = (float2 *) calloc(count, sizeof(float2));
	float2 *h_output_cuFFT = (float2 *) calloc(count, sizeof(float2));
	if (!h_input || !h_output_smFFT || !h_output_cuFFT) { printf("Host memory allocation failed.\n"); return 1; }

	if (DEBUG) printf("Initializing data with random numbers...\t");
	harness_seed();
	harness_fill_uniform((float *) h_input, 2*count);   // U[0,1) re and im (FFT.c:141-142)
	if (DEBUG) printf("done.\n");

	double cuFFT_execution_time, smFFT_execution_time, smFFT_multiple_execution_time;
	GPU_cuFFT(h_input, h_output_cuFFT, FFT_size, nFFTs, inverse, nRuns, &cuFFT_execution_time);
	GPU_smFFT_4elements(h_input, h_output_smFFT, FFT_size, nFFTs, inverse, reorder, nRuns, &smFFT_execution_time, &smFFT_multiple_execution_time);

	if (reorder) {
		double cumulative_error, mean_error;
		print_verdict(Compare_data(h_output_cuFFT, h_output_smFFT, FFT_size, nFFTs, &cumulative_error, &mean_error));
	}
	else {
		printf("  There is no verification of the results if FFT are not reordered.\n");
	}

	free(h_input); free(h_output_smFFT); free(h_output_cuFFT);
	(void) hipDeviceReset();
	return 0;
}
