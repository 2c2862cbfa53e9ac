/*
 * smfft.h -- C ABI of libsmfft_amd.so, the MI355X (gfx950) shared-memory FFT library.
 *
 * This is the drop-in boundary for the hot path of KAdamek/SMFFT: plain pointers and sizes, no
 * C++ or torch types.  Every entry point names the reference interface it replaces (paths are
 * relative to the reference checkout; CT = SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu,
 * ST = SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu, RC = SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu).
 * The same library also exports the reference's own C++-linkage symbols (FFT_init,
 * FFT_external_benchmark, FFT_multiple_benchmark, GPU_smFFT_4elements, ...; see
 * include/smfft_reference_api.h) so the reference's FFT.c harnesses link against it unchanged.
 *
 * Conventions shared with the reference:
 *   - complex data is interleaved float (re, im) = float2; FFT f of a batch is at element f*N;
 *   - d_* pointers are DEVICE pointers owned by the caller; transforms are out of place and leave
 *     the input untouched; all transforms are un-normalised;
 *   - the *_benchmark calls time exactly one kernel launch with events and ADD the milliseconds to
 *     *FFT_time (CT:598,660-662); they are synchronous;
 *   - return value 0 = ok.  Unsupported lengths print "Error wrong FFT length!" and return 0 with
 *     nothing launched, as upstream (CT:656-658).
 */
#ifndef SMFFT_H_
#define SMFFT_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SMFFT_NREUSES 100 /* FFTs per load/store in the `multiple` path (CT:10) */

/* FFT_init (CT:576-581).  The CUDA cache/bank configuration calls have no gfx950 meaning; this
 * selects the device (SMFFT_DEVICE or the one already current) and reads the tuning environment
 * (SMFFT_GRID_CAP: maximum workgroups per launch, default 12288; 0 = one per 4096-element tile). */
void smfft_init(void);

/* ---- Cooley-Tukey C2C family, N = 32 .. 4096 ------------------------------------------------ */
/* FFT_external_benchmark (CT:583-664): d_output[f] = FFT(d_input[f]), f < nFFTs.
 * reorder != 0: natural-order DFT; reorder == 0: DFT of the bit-reversed-index input (the DIT
 * butterfly network applied to natural-order data), exactly as fft_reorder = 0 upstream. */
int smfft_ct_external_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs,
                                int inverse, int reorder, double* FFT_time);
/* FFT_multiple_benchmark (CT:666-752): the first nFFTs/100 FFTs are each transformed 100 times in
 * LDS.  Returns 1 and sets *FFT_time = -1 when nFFTs/100 == 0 (CT:669-673). */
int smfft_ct_multiple_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs,
                                int inverse, int reorder, double* FFT_time);
/* The same benchmark on the natural-order compact kernel WITHOUT cross-application fusion: every one of the NREUSES applications
 * reads its input from the LDS image and leaves its output there -- what one call of do_SMFFT_CT_DIT costs a kernel whose data
 * live in LDS (CT:553-572). */
int smfft_ct_multiple_unfused_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time);
/* ... and for either ordering (round 6).  The kernels that keep a chain in registers across its applications -- the natural-order
 * planar kernels (N >= 64) and the lane engines of N = 32 and of N = 64 without reorder, which touch no LDS memory between a chain's
 * first and last application -- run with one image load and one image store per application, the shape of upstream's own loop
 * (CT:553-572); the planar no-reorder kernels (N >= 128) re-read the image in every application as they are, so for them this IS
 * smfft_ct_multiple_benchmark. */
int smfft_ct_multiple_percall_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, int reorder, double* FFT_time);

/* ---- Stockham C2C family (un-normalised INVERSE transform, ST:76), N = 32 .. 4096 ------------
 * (upstream: 256 .. 4096; the smaller lengths are an extension) */
/* FFT_external_benchmark / FFT_multiple_benchmark (ST:306-346, :348-384). */
int smfft_st_external_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, double* FFT_time);
int smfft_st_multiple_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, double* FFT_time);

/* Forward-direction counterpart of the Stockham external call.  Upstream's Stockham C2C program only has
 * the + sign (ST:76); SURVEY.md 8(f) item 3 lists the forward direction as a gap.  inverse != 0 is
 * smfft_st_external_benchmark; inverse == 0 computes the un-normalised FORWARD transform, natural order. */
int smfft_st_external_benchmark_dir(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time);

/* ---- R2C / C2R family, real FFT_size = 512 .. 4096 -------------------------------------------- */
/* FFT_external_benchmark (RC:396-432).  inverse == 0: d_input = nFFTs*FFT_size reals,
 * d_output = nFFTs*FFT_size/2 float2 with element 0 = (X[0].re, X[N/2].re).  inverse != 0: the
 * packed layout in, (FFT_size/2) * x out as reals. */
int smfft_rc_external_benchmark(const float* d_input, float* d_output, int FFT_size, int nFFTs,
                                int inverse, double* FFT_time);
/* FFT_multiple_benchmark (RC:435-467), forward only as upstream. */
int smfft_rc_multiple_benchmark(const float* d_input, float* d_output, int FFT_size, int nFFTs, double* FFT_time);

/* ---- launch-only forms (no events, no synchronisation) on a caller-provided hipStream_t --------
 * family: 0 = CT, 1 = ST, 2 = RC.  path: 0 = external, 1 = multiple.  For family 2, FFT_size is
 * the REAL length.  family 1 (the Stockham program, + sign only upstream, ST:76): inverse != 0 is the
 * program's transform, inverse == 0 the forward extension; reorder is ignored (natural order).  Used by
 * bench.py and smfft_host_transform; same kernels as the *_benchmark calls (capturable into a hipGraph by the caller).
 * Returns 0, a hipError_t, or -1 for an unsupported (family, FFT_size). */
int smfft_launch(int family, int path, const void* d_input, void* d_output, int FFT_size, int nFFTs,
                 int inverse, int reorder, void* hip_stream);

/* Calibration: streams n_float2 elements (a multiple of 4096) from d_input to d_output with exactly
 * the external kernels' global access shape and grid, no FFT: the same-run copy ceiling. */
int smfft_copy_launch(const void* d_input, void* d_output, long long n_float2, void* hip_stream);

/* ---- L3 wrappers: host buffers in, host buffers out (alloc, H2D, nRuns launches, D2H, free) ---- */
/* GPU_smFFT_4elements (CT:827-908). */
int smfft_gpu_ct(const void* h_input, void* h_output, int FFT_size, int nFFTs, int inverse, int reorder,
                 int nRuns, double* single_ex_time, double* multi_ex_time);
/* GPU_FFT_C2C_Stockham (ST:457-530). */
int smfft_gpu_st(const void* h_input, void* h_output, int FFT_size, int nFFTs, int nRuns,
                 double* single_ex_time, double* multi_ex_time);
/* GPU_smFFT_R2C (RC:572-650) / GPU_smFFT_C2R (RC:652-688). */
int smfft_gpu_r2c(void* h_output, const float* h_input, int FFT_size, int nFFTs, int nRuns);
int smfft_gpu_c2r(float* h_output, const void* h_input, int FFT_size, int nFFTs, int nRuns);

/* ---- host-resident batches (no upstream counterpart) ---------------------
 * smfft_host_transform streams nFFTs transforms whose input AND output live in HOST memory through the
 * current device: the batch is cut into slabs of slab_ffts FFTs (<= 0: 32 MiB worth, SMFFT_HOST_SLAB_MIB)
 * and `lanes` (<= 0: 8 for pageable, 2 for pinned memory; SMFFT_HOST_LANES) host threads, each with its own HIP stream and two device slab
 * pairs, run H2D -> FFT -> D2H for interleaved slabs, so both PCIe directions and the kernels overlap and
 * the device never holds more than 2 * lanes slab pairs (the batch may exceed device memory).  Pageable
 * buffers go through per-lane pinned bounce buffers, copied by the lane threads in parallel.  PINNED buffers
 * (smfft_host_malloc, or hipHostRegister'ed memory) are not copied at all: the transform's kernel reads and
 * writes them directly over PCIe, both directions at once (config-2 batch: 88 ms = 97 GB/s in + out against
 * 118 ms through the slabs; SMFFT_HOST_ZERO_COPY=0 keeps the slab pipeline, slab_ffts / lanes then apply).  family / FFT_size / inverse / reorder as for smfft_launch (family 2: real length; input and
 * output are FFT_size * 4 bytes per FFT either way).  *elapsed_ms = wall-clock time of the whole call
 * (end to end, PCIe included); the first call also builds the cached pipeline (smfft_host_pipeline_release
 * frees it).  Returns 0, -1 for an unsupported (family, FFT_size), -4 when the pipeline cannot be allocated,
 * or a hipError_t. */
int smfft_host_transform(int family, const void* h_input, void* h_output, int FFT_size, long long nFFTs,
                         int inverse, int reorder, long long slab_ffts, int lanes, double* elapsed_ms);
void* smfft_host_malloc(unsigned long long bytes); /* pinned host memory (hipHostMalloc) */
int smfft_host_free(void* h_ptr);
void smfft_host_pipeline_release(void);

/* ---- tuning / introspection -------------------------------------------------------------------
 * The setters act on the CALLING HOST THREAD (upstream: process globals and one thread, CT:15): device, grid cap,
 * applications per slot and pacing belong to the thread, so that one thread per GPU can drive the unchanged prototypes
 * concurrently; a value a thread has not set falls back to the process default (SMFFT_DEVICE, SMFFT_GRID_CAP, SMFFT_PACING,
 * read once).  The lanes of smfft_host_transform inherit the state of the thread that called it. */
void smfft_set_grid_cap(int max_workgroups); /* default 12288 (persistent, grid-strided); 0 = one workgroup per tile */
int smfft_get_grid_cap(void);
/* Applications of the transform per slot in the `multiple` kernels.  The benchmark value is
 * SMFFT_NREUSES (100, CT:10); the knob exists so the in-LDS device functions can be verified with
 * 1, 2 or 4 applications (100 un-normalised FFTs overflow fp32, as they do upstream). n <= 0 resets. */
void smfft_set_nreuses(int n);
int smfft_get_nreuses(void);
int smfft_device_count(void);
int smfft_set_device(int device);
const char* smfft_version(void);

/* ---- plain device-memory helpers so a C / ctypes caller needs no other HIP binding ----------- */
void* smfft_malloc(unsigned long long bytes);
/* Two buffers of `bytes` each for a kernel that READS the first and WRITES the second.  On MI355X the rate of such a
 * kernel depends on which physical memory the two buffers are (profiles/r02_placement_*,
 * profiles/r02_vmm_mixed_assembly.txt, profiles/r02_vmm_interleave.txt): physical memory comes in classes, and about one
 * physical GiB in seven is MIXED (pure writes 20 % faster, pure reads 7 % slower than ordinary memory).  Input and output
 * ordinary and in the same class -- what two hipMalloc calls in a row give -- move the 4 GiB + 4 GiB N=1024 batch in
 * 1.55-1.60 ms (0.69 of the HBM peak), in different classes in 1.48-1.53 ms, with the output in mixed memory in
 * 1.30-1.31 ms (0.82) -- and in 1.32 ms with an output whose 8 MiB pieces alternate between ordinary memory of two classes.
 * This call takes the input from hipMalloc and BUILDS the output with the virtual-memory API: physical memory is created
 * in 8 MiB handles, 1 GiB at a time, and each GiB is classified by two write-only passes (mixed or not; same or other
 * class than the first ordinary GiB).  When mixed memory plus equal parts of two classes cover the output (and six GiB
 * further), the candidate outputs -- mixed memory first, interleaved classes only -- are each timed as the target of a
 * copy from the real input over the whole pair and the best is kept; while it is not good (a pass into it beating the same
 * pass into ordinary memory of the same scan by the margin mixed memory shows on this device) eight more GiB are scanned and the candidates tried again, up to four times and inside the budgets -- a quarter
 * of the free memory (SMFFT_PAIR_BUDGET_FRAC; never more than what is free after the pair itself less 1 GiB), 2 s
 * (SMFFT_PAIR_BUDGET_MS); whatever is missing then comes from ordinary chunks.  Typically 10-25 GiB and 80-550 ms for a 4 GiB output (on the system runtime up to 1-2 s).  The
 * chosen handles are blended evenly into one virtual range (an ordinary device pointer for the caller), the rest is
 * released at once.  Buffers are exactly `bytes` long (the output's range is rounded up to 8 MiB).
 * SMFFT_PAIR_POLICY=candidates: whole hipMalloc blocks timed as copy targets inside the same budgets
 * (also the fallback where the virtual-memory API is unavailable); =plain: two plain allocations.  Nothing is kept after
 * smfft_free_pair unless SMFFT_PAIR_CACHE=1 (pairs taken through smfft_malloc_pair_for_wrapper are kept for the next such call).
 * Every virtual address range the allocator has used stays reserved and unmapped afterwards (smfft_va_window).  Requests below 256 MiB are served plainly.
 * The L3 wrappers take their two buffers from this call; SMFFT_WRAPPER_PLACEMENT=0: two plain allocations as upstream (CT:850-853).
 * Release with smfft_free_pair(d_read) (an error for a pointer this call did not return). */
int smfft_malloc_pair(unsigned long long bytes, void** d_read, void** d_written);
/* the same with explicit budgets for this call (a negative value = the default / the environment's): for a process that owns
 * the device and prefers a longer scan to an output that is only partly fast (bench.py's second attempt) */
int smfft_malloc_pair_budget(unsigned long long bytes, void** d_read, void** d_written, double budget_frac, double budget_ms);
int smfft_free_pair(void* d_read);
/* Only the WRITTEN buffer, built the same way, for a caller whose input exists already (allocated elsewhere, produced by
 * another library): changing the allocation of the output is then the one-line way to the rates above.  Release with
 * smfft_free_written. */
int smfft_malloc_written(unsigned long long bytes, void** d_written);
/* the same with the caller's input buffer (at least `bytes` long; only read): the candidate outputs are then judged by
 * timed copies from it, as in smfft_malloc_pair, instead of by write passes alone */
int smfft_malloc_written_for(const void* d_read, unsigned long long bytes, void** d_written);
int smfft_free_written(void* d_written);
/* gives back the pair SMFFT_PAIR_CACHE=1 keeps */
int smfft_pair_cache_release(void);
/* the pair the L3 wrappers (and the harness's hipFFT comparator) take: as smfft_malloc_pair unless SMFFT_WRAPPER_PLACEMENT=0
   (two plain allocations, as upstream), and the pair released last is kept for the next wrapper call of the same size */
int smfft_malloc_pair_for_wrapper(unsigned long long bytes, void** d_read, void** d_written);
/* what the last smfft_malloc_pair of this process did (telemetry for bench.py and the tests) */
typedef struct SmfftPairInfo {
    unsigned long long bytes;            /* size of each buffer */
    unsigned long long candidate_bytes;  /* physical memory the scan held at its end (<= max(byte budget + 1 GiB, bytes)) */
    int candidates;                      /* mixed policy: GiB chunks scanned (+1 if a remainder was created unprobed); candidates policy: blocks probed; 0: plain */
    int chosen;                          /* mixed policy: GiB of mixed memory in the output; candidates policy: index of the block kept */
    int good_enough;                     /* 1: the whole-pair copy from the input into the output takes at most 2.31 x the pure read pass over the input (good outputs: 2.18-2.29 x, ordinary memory of the input's class: 2.49-2.61 x); without an input: a write pass into it beats the scan's ordinary chunks by the margin its mixed chunks show */
    float read_ms, copy_ms, first_copy_ms;   /* over min(bytes, 1 GiB): pure read of the input; copy into the output; copy into the first chunk / block seen */
    double search_ms;
    unsigned long long mixed_bytes;        /* mixed policy: bytes of the output that are mixed memory ... */
    unsigned long long interleaved_bytes;  /* ... and bytes that are ordinary memory of two classes interleaved in 8 MiB handles */
    float first_ordinary_copy_ms;        /* over min(bytes, 1 GiB): copy into the first clearly ordinary chunk of the scan (reported; good_enough is judged against the input's read pass) */
    int classification;                  /* 1: the scan's write times split into a fast (mixed) and a slow (ordinary) cluster; 0: inconclusive -- nothing was called mixed */
} SmfftPairInfo;
int smfft_last_pair_info(SmfftPairInfo* out);
/* the window of virtual addresses the allocator has handed out and retired so far: [*first, *next); returns the number of
   retired ranges it holds re-reserved (never mapped), so that no other reservation of the process can land in them */
int smfft_va_window(unsigned long long* first, unsigned long long* next);
/* K >= 0: the external kernels of THIS host thread run their rate limiter with K loads whatever the output buffer; < 0: automatic */
void smfft_set_pacing(int k);
/* The multiple paths' schedule for THIS host thread: 1 (default; SMFFT_MULT_BALANCE) = when a launch holds more chains than fit on
   the chip at once, a persistent grid of the co-resident workgroups shares the launch's applications evenly -- a chain that straddles
   two workgroups is parked once in its own output slot and resumed (same bits); 0 = one chain per workgroup,
   grid-strided, as in rounds 1-3; n >= 2 (tests): balanced over n workgroups, as if the chip held no more; < 0: back to the process default */
void smfft_set_multiple_balance(int on);
int smfft_get_multiple_balance(void);
/* The multiple paths' wave priorities for THIS host thread: k > 0 = every wave's scheduling priority rotates every 2^k shader clocks
   (default 15; SMFFT_PRIO_ROTATE), so that the chains sharing a SIMD advance at the same average rate and end together; 0 = the
   hardware's oldest-wave-first order (rounds 1-3); < 0: back to the process default */
void smfft_set_multiple_rotation(int log2_clocks);
int smfft_get_multiple_rotation(void);
/* The balanced schedule hands a cut chain from one workgroup to the next through the chain's slot of d_output and a word of device
   memory.  It does not depend on the two workgroups being resident together: a workgroup that has waited `microseconds` (default 1000;
   SMFFT_HANDOFF_WAIT_US; THIS host thread) for data nobody has committed to parking runs the whole chain itself from d_input -- same
   bits -- and the late owner skips it.  So a launch finishes whatever else occupies the device; < 0: back to the process default. */
void smfft_set_handoff_wait_us(int microseconds);
/* (fault injection for the tests of that path: include/smfft_debug.h)
   Introspection: buffers of hand-over words the library holds (one per balanced launch in flight, recycled by event; at most 32,
   256 KiB each -- 65536 chains x one word) and, in *in_flight, how many of them a launch is still using. */
int smfft_schedule_buffers(int* in_flight);
/* How many workgroups of a multiple kernel the device holds at once, COUNTED by a calibration launch over scratch buffers (family 0
   CT / 1 Stockham, path 1 or 2); *assumed = what the balanced schedule computes from the kernel's registers and LDS.  < 0: error. */
int smfft_measure_multiple_residency(int family, int FFT_size, int inverse, int reorder, int path, int* assumed);
/* the K an external launch of `family` (0 CT, 1 Stockham, 2 R2C/C2R: FFT_size = the real length) and length FFT_size would run
   with for this output buffer right now (introspection for the tests: what pacing_for chooses per launch) */
int smfft_pacing_for_output(const void* d_output, int family, int FFT_size);
int smfft_free(void* d_ptr);
int smfft_memcpy_h2d(void* d_dst, const void* h_src, unsigned long long bytes);
int smfft_memcpy_d2h(void* h_dst, const void* d_src, unsigned long long bytes);
int smfft_memcpy_d2d(void* d_dst, const void* d_src, unsigned long long bytes);
int smfft_memset(void* d_ptr, int value, unsigned long long bytes);
int smfft_synchronize(void);
/* free and total device memory of the current device as hipMemGetInfo reports them (what FFT_init prints, CT/FFT-GPU-32bit.cu:766, 839) */
int smfft_mem_info(unsigned long long* free_bytes, unsigned long long* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SMFFT_H_ */
