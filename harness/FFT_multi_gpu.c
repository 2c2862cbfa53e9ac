// FFT_multi_gpu.c -- config 5 of BASELINE.json at the C level: a batch of N-point C2C FFTs sharded
// across the GPUs of one node.  Usage:
//     FFT_multi_gpu.exe <FFT length> <FFTs per GPU> <nRuns> <inverse 0|1> <reorder 0|1> [nGPUs] [exchange 0|1] [virtual shards G]
//
// The path shards embarrassingly: GPU g owns the contiguous slab
// [g*B, (g+1)*B) of the batch, generates it on its own host thread, uploads it, and runs the
// identical kernel through the same L2 call (FFT_external_benchmark) as the single-GPU harness;
// there is NO data-path collective.  RCCL (over xGMI between GPUs) is used only for what
// north_star calls the trivial part: one communicator per GPU (ncclCommInitAll) and an all-reduce
// of the per-GPU statistics -- MAX of the kernel time, SUM of the error counts -- so every rank
// ends with the job-level numbers.  One host thread per GPU (HIP's current device is per thread).
//
// exchange = 1 adds, OUTSIDE the timed transform and reported separately (a 4 GiB slab over one
// xGMI link takes ~28 ms against 1.4 ms of compute), the two optional payload movements of a single-buffer
// workflow: an ncclAllGather of the output slabs into one buffer on every GPU, and a scatter of that buffer's
// slabs from GPU 0 back to their owners with a grouped ncclSend / ncclRecv; both are verified bit for bit
// against the GPU's own output.  With exchange = 1 the slabs live in plain hipMalloc memory: the allocator's built outputs are
// virtual-memory ranges with access granted to their own device only, which RCCL's peer-to-peer paths need not accept.
//
// virtual shards G > 0 ("test without 8 GPUs"): the second argument is then the TOTAL batch, which is cut
// into G contiguous slabs exactly as the real run cuts it over G GPUs -- ragged if G does not divide it: the first
// (total mod G) slabs hold one FFT more -- and the slabs are transformed one after the other on device 0; the per-slab
// statistics are reduced on the host the way the all-reduce does (MAX of the times, SUM of the errors) and the concatenated
// output is compared bit for bit with ONE launch over the whole batch.
#include "harness_common.h"
#include <pthread.h>
#include <rccl/rccl.h>

int FFT_external_benchmark(float2 *d_input, float2 *d_output, int FFT_size, int nFFTs, bool inverse, bool reorder, double *FFT_time);
void FFT_init();
// the library's placement-aware allocator (include/smfft.h): input and output in different memory regions
extern "C" int smfft_malloc_pair(unsigned long long bytes, void **d_read, void **d_written);
extern "C" int smfft_free_pair(void *d_read);
extern "C" void *smfft_malloc(unsigned long long bytes);
extern "C" int smfft_free(void *d_ptr);

typedef struct {
	int gpu, nGPUs, FFT_size, nFFTs, nRuns;
	bool inverse, reorder, exchange;
	double gather_ms, scatter_ms;   // MAX over GPUs (after the all-reduce); 0 without exchange
	ncclComm_t comm;
	double kernel_ms;       // this GPU's mean launch time
	double job_ms;          // MAX over GPUs (after the all-reduce)
	double job_errors;      // SUM over GPUs
	int status;
} worker_t;

// A worker that fails before the collectives must not leave the others waiting inside them: every worker reports its
// local status, all meet at a host barrier, and the RCCL calls are made only if NOBODY failed.
static pthread_barrier_t g_rendezvous;
static int g_failures = 0;

static void *worker(void *arg) {
	worker_t *w = (worker_t *) arg;
	w->status = 1;
	const size_t count = (size_t) w->FFT_size*w->nFFTs, bytes = count*sizeof(float2);
	float2 *h_in = NULL, *h_out = NULL, *d_in = NULL, *d_out = NULL, *d_all = NULL, *d_back = NULL;
	float *d_stats = NULL;
	hipStream_t stream = NULL;
	hipEvent_t e0 = NULL, e1 = NULL, e2 = NULL;
	double errors = 0;
	bool ok = hipSetDevice(w->gpu) == hipSuccess;
	if (ok) {
		h_in = (float2 *) malloc(bytes);
		h_out = (float2 *) malloc(bytes);
		if (w->exchange) {   // buffers that go into collectives: plain allocations (see the header comment)
			d_in = (float2 *) smfft_malloc(bytes);
			d_out = (float2 *) smfft_malloc(bytes);
		}
		ok = h_in && h_out && (w->exchange ? (d_in && d_out) : smfft_malloc_pair(bytes, (void **) &d_in, (void **) &d_out) == 0)
		     && hipMalloc((void **) &d_stats, 4*sizeof(float)) == hipSuccess && hipStreamCreate(&stream) == hipSuccess;
		if (ok && w->exchange)
			ok = hipMalloc((void **) &d_all, bytes*w->nGPUs) == hipSuccess && hipMalloc((void **) &d_back, bytes) == hipSuccess
			     && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventCreate(&e2) == hipSuccess;
		if (!ok) printf("GPU %d: allocation failed\n", w->gpu);
	}
	if (ok) {
		// per-slab reproducible data, U[0,1) like FFT.c:141-142: element i of slab g is counter g*2*count + i of the
		// harness's counter-based generator (seed fixed in main), so the batch does not depend on the number of GPUs
		for (size_t i = 0; i < 2*count; i++) ((float *) h_in)[i] = harness_uniform((unsigned long long) w->gpu*2*count + i);
		ok = hipMemcpy(d_in, h_in, bytes, hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		FFT_init();
		double total = 0;
		FFT_external_benchmark(d_in, d_out, w->FFT_size, w->nFFTs, w->inverse, w->reorder, &total);   // warm-up
		total = 0;
		for (int r = 0; r < w->nRuns; r++) FFT_external_benchmark(d_in, d_out, w->FFT_size, w->nFFTs, w->inverse, w->reorder, &total);
		w->kernel_ms = total/w->nRuns;
		ok = hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (ok) {
		// slab self-check without a second library: Parseval, sum|X|^2 = N * sum|x|^2 per FFT (first 64 FFTs)
		const int nCheck = w->nFFTs < 64 ? w->nFFTs : 64;
		for (int f = 0; f < nCheck; f++) {
			double ein = 0, eout = 0;
			for (int i = 0; i < w->FFT_size; i++) {
				float2 a = h_in[(size_t) f*w->FFT_size + i], b = h_out[(size_t) f*w->FFT_size + i];
				ein += (double) a.x*a.x + (double) a.y*a.y;
				eout += (double) b.x*b.x + (double) b.y*b.y;
			}
			if (fabs(eout/(w->FFT_size*ein) - 1.0) > 1e-5) { printf("GPU %d: FFT %d fails the Parseval check\n", w->gpu, f); errors += 1; }
		}
	}
	if (!ok) __sync_fetch_and_add(&g_failures, 1);
	pthread_barrier_wait(&g_rendezvous);
	const bool everybody_ok = __sync_fetch_and_add(&g_failures, 0) == 0;

	// optional payload movement over xGMI, never inside the timed transform
	float gather_ms = 0, scatter_ms = 0;
	if (everybody_ok && w->exchange) {
		(void) hipMemsetAsync(d_back, 0, bytes, stream);
		(void) hipEventRecord(e0, stream);
		ncclAllGather(d_out, d_all, 2*count, ncclFloat, w->comm, stream);
		(void) hipEventRecord(e1, stream);
		// point-to-point pieces of at most 1 GiB: a single 2 GiB ncclSend/ncclRecv arrived without its tail on
		// this RCCL (2^31-byte message), the all-gather has no such limit
		const size_t piece = (size_t) 1 << 27;   // float2 elements = 1 GiB
		for (size_t off = 0; off < count; off += piece) {
			const size_t n = (count - off < piece ? count - off : piece);
			ncclGroupStart();
			if (w->gpu == 0) for (int r = 0; r < w->nGPUs; r++) ncclSend(d_all + (size_t) r*count + off, 2*n, ncclFloat, r, w->comm, stream);
			ncclRecv(d_back + off, 2*n, ncclFloat, 0, w->comm, stream);
			ncclGroupEnd();
		}
		(void) hipEventRecord(e2, stream);
		(void) hipStreamSynchronize(stream);
		(void) hipEventElapsedTime(&gather_ms, e0, e1);
		(void) hipEventElapsedTime(&scatter_ms, e1, e2);
		// own segment of the gathered buffer and the slab scattered back == own output, bit for bit (first and last MiB)
		const size_t probe = bytes < (1u << 20) ? bytes : (1u << 20);
		char *t = (char *) malloc(probe);
		const char *srcs[2] = {(const char *) (d_all + (size_t) w->gpu*count), (const char *) d_back};
		for (int k = 0; k < 2; k++) {
			const size_t offs[2] = {0, bytes - probe};
			for (int o = 0; o < 2; o++) {
				if (!t || hipMemcpy(t, srcs[k] + offs[o], probe, hipMemcpyDeviceToHost) != hipSuccess || memcmp(t, (const char *) h_out + offs[o], probe) != 0) {
					printf("GPU %d: %s differs from the GPU's own output at byte offset %zu\n", w->gpu, k == 0 ? "all-gather segment" : "scattered slab", offs[o]);
					errors += 1;
				}
			}
		}
		free(t);
	}

	// the only communication of the timed path: job-level statistics over RCCL
	if (everybody_ok) {
		float h_stats[4] = {(float) w->kernel_ms, gather_ms, scatter_ms, (float) errors};
		(void) hipMemcpy(d_stats, h_stats, sizeof(h_stats), hipMemcpyHostToDevice);
		ncclAllReduce(d_stats, d_stats, 3, ncclFloat, ncclMax, w->comm, stream);
		ncclAllReduce(d_stats + 3, d_stats + 3, 1, ncclFloat, ncclSum, w->comm, stream);
		(void) hipStreamSynchronize(stream);
		(void) hipMemcpy(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost);
		w->job_ms = h_stats[0];
		w->gather_ms = h_stats[1];
		w->scatter_ms = h_stats[2];
		w->job_errors = h_stats[3];
	}

	// every path releases what it got
	if (w->exchange) { if (d_in) (void) smfft_free(d_in); if (d_out) (void) smfft_free(d_out); }
	else if (d_in) (void) smfft_free_pair(d_in);
	if (d_stats) (void) hipFree(d_stats);
	if (d_all) (void) hipFree(d_all);
	if (d_back) (void) hipFree(d_back);
	if (e0) (void) hipEventDestroy(e0);
	if (e1) (void) hipEventDestroy(e1);
	if (e2) (void) hipEventDestroy(e2);
	if (stream) (void) hipStreamDestroy(stream);
	free(h_in); free(h_out);
	w->status = ok ? 0 : 1;
	return NULL;
}

// contiguous, balanced split (smfft_amd/sharding.py shard_range): the first total % G slabs get one FFT more
static void shard_range(long total, int g, int G, long *first, long *count) {
	const long base = total/G, extra = total % G;
	*first = g*base + (g < extra ? g : extra);
	*count = base + (g < extra ? 1 : 0);
}

static int run_virtual_shards(int FFT_size, long total, int nRuns, bool inverse, bool reorder, int G) {
	const size_t count = (size_t) FFT_size*total, bytes = count*sizeof(float2);
	float2 *h_in = (float2 *) malloc(bytes), *h_whole = (float2 *) malloc(bytes), *h_cat = (float2 *) malloc(bytes);
	float2 *d_in = (float2 *) smfft_malloc(bytes), *d_whole = (float2 *) smfft_malloc(bytes), *d_cat = (float2 *) smfft_malloc(bytes);
	if (!h_in || !h_whole || !h_cat || !d_in || !d_whole || !d_cat) { printf("allocation failed\n"); return 1; }
	for (size_t i = 0; i < 2*count; i++) ((float *) h_in)[i] = harness_uniform((unsigned long long) i);
	bool ok = hipMemcpy(d_in, h_in, bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_cat, 0xFF, bytes) == hipSuccess;
	FFT_init();
	double whole_ms = 0, job_ms = 0, errors = 0;
	FFT_external_benchmark(d_in, d_whole, FFT_size, (int) total, inverse, reorder, &whole_ms);
	long covered = 0;
	for (int g = 0; g < G && ok; g++) {
		long first, n;
		shard_range(total, g, G, &first, &n);
		if (first != covered) { printf("shard %d does not start where shard %d ended\n", g, g - 1); errors += 1; }
		covered = first + n;
		double ms = 0;
		if (n > 0) for (int r = 0; r < nRuns; r++) FFT_external_benchmark(d_in + (size_t) first*FFT_size, d_cat + (size_t) first*FFT_size, FFT_size, (int) n, inverse, reorder, &ms);
		ms /= (nRuns > 0 ? nRuns : 1);
		printf("  shard %d of %d: FFTs [%ld, %ld): SH FFT normal = %0.4f ms\n", g, G, first, first + n, ms);
		if (ms > job_ms) job_ms = ms;            // what ncclAllReduce(MAX) leaves on every rank
	}
	if (covered != total) { printf("the shards cover %ld of %ld FFTs\n", covered, total); errors += 1; }
	ok = ok && hipMemcpy(h_whole, d_whole, bytes, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(h_cat, d_cat, bytes, hipMemcpyDeviceToHost) == hipSuccess;
	const bool identical = ok && memcmp(h_whole, h_cat, bytes) == 0;
	if (!identical) errors += 1;
	printf("  %d virtual shard(s) on device 0, %ld FFTs of %d: job time (MAX over shards) = %0.4f ms; one launch over the whole batch = %0.4f ms\n", G, total, FFT_size, job_ms, whole_ms);
	printf("  concatenated shard outputs vs one launch over the whole batch: %s\n", identical ? "bit-identical" : "DIFFERENT");
	smfft_free(d_in); smfft_free(d_whole); smfft_free(d_cat);
	free(h_in); free(h_whole); free(h_cat);
	print_verdict((int) errors);
	return ok && errors == 0 ? 0 : 1;
}

int main(int argc, char *argv[]) {
	if (argc < 6 || argc > 9) {
		printf("Argument error!\n 1) FFT length\n 2) number of FFTs per GPU\n 3) the number of kernel executions\n 4) do inverse FFT 1=yes 0=no\n 5) reorder 1=yes 0=no\n 6) [number of GPUs, default (or 0) all]\n 7) [exchange 1 = also time an all-gather of the outputs and a scatter from GPU 0]\n 8) [virtual shards G: argument 2 is the TOTAL batch, cut into G slabs run one after the other on device 0]\n");
		printf("For example: FFT_multi_gpu.exe 1024 524288 20 0 1\n");
		return 1;
	}
	int FFT_size = (int) strtol(argv[1], NULL, 10), nFFTs = (int) strtol(argv[2], NULL, 10), nRuns = (int) strtol(argv[3], NULL, 10);
	bool inverse = strtol(argv[4], NULL, 10) == 1, reorder = strtol(argv[5], NULL, 10) == 1;
	int devCount = 0;
	if (hipGetDeviceCount(&devCount) != hipSuccess || devCount < 1) { printf("No HIP device.\n"); return 1; }
	int nGPUs = (argc >= 7) ? (int) strtol(argv[6], NULL, 10) : devCount;
	const bool exchange = (argc >= 8) && strtol(argv[7], NULL, 10) == 1;
	if (nGPUs < 1 || nGPUs > devCount) nGPUs = devCount;
	harness_seed_value = getenv("SMFFT_SEED") ? strtoull(getenv("SMFFT_SEED"), NULL, 10) : 20200720ull;
	if (argc == 9 && strtol(argv[8], NULL, 10) > 0) {
		if (hipSetDevice(0) != hipSuccess) { printf("No HIP device.\n"); return 1; }
		return run_virtual_shards(FFT_size, nFFTs, nRuns, inverse, reorder, (int) strtol(argv[8], NULL, 10));
	}

	ncclComm_t *comms = (ncclComm_t *) malloc(nGPUs*sizeof(ncclComm_t));
	int *devs = (int *) malloc(nGPUs*sizeof(int));
	for (int g = 0; g < nGPUs; g++) devs[g] = g;
	if (ncclCommInitAll(comms, nGPUs, devs) != ncclSuccess) { printf("ncclCommInitAll failed\n"); return 1; }

	worker_t *w = (worker_t *) calloc(nGPUs, sizeof(worker_t));
	pthread_t *th = (pthread_t *) malloc(nGPUs*sizeof(pthread_t));
	pthread_barrier_init(&g_rendezvous, NULL, nGPUs);
	for (int g = 0; g < nGPUs; g++) {
		w[g].gpu = g; w[g].nGPUs = nGPUs; w[g].FFT_size = FFT_size; w[g].nFFTs = nFFTs; w[g].nRuns = nRuns;
		w[g].inverse = inverse; w[g].reorder = reorder; w[g].exchange = exchange; w[g].comm = comms[g];
		pthread_create(&th[g], NULL, worker, &w[g]);
	}
	int failed = 0;
	for (int g = 0; g < nGPUs; g++) { pthread_join(th[g], NULL); failed += w[g].status; }
	for (int g = 0; g < nGPUs; g++) ncclCommDestroy(comms[g]);
	if (failed) { printf("  %d GPU worker(s) failed\n", failed); return 1; }

	const double bytes_per_gpu = 2.0*FFT_size*(double) nFFTs*sizeof(float2);
	for (int g = 0; g < nGPUs; g++) printf("  GPU %d: SH FFT normal = %0.3f ms (%0.1f GB/s)\n", g, w[g].kernel_ms, bytes_per_gpu/w[g].kernel_ms/1e6);
	printf("  %d GPU(s), %d FFTs of %d each: job time = %0.3f ms; %0.4g FFT/s; %0.1f GB/s aggregate\n", nGPUs, nFFTs, FFT_size, w[0].job_ms,
	       (double) nFFTs*nGPUs/(w[0].job_ms*1e-3), nGPUs*bytes_per_gpu/w[0].job_ms/1e6);
	if (exchange) {
		const double slab_gb = bytes_per_gpu/2/1e9;
		printf("  exchange (not part of the job time): all-gather of the %d output slabs = %0.3f ms (%0.1f GB/s received per GPU); scatter from GPU 0 = %0.3f ms (%0.1f GB/s sent by GPU 0)\n",
		       nGPUs, w[0].gather_ms, (nGPUs - 1)*slab_gb/(w[0].gather_ms*1e-3 + 1e-12), w[0].scatter_ms, (nGPUs - 1)*slab_gb/(w[0].scatter_ms*1e-3 + 1e-12));
	}
	print_verdict((int) w[0].job_errors);
	return 0;
}
