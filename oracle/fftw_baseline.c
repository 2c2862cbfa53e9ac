/*
 * fftw_baseline.c -- CPU speed baseline for bench.py's "cpu_baseline" leg (test/measurement
 * infrastructure only, never part of the product path).
 *
 * north_star asks for "FFTW batched on the host cores of the GPU box in the same run".  Real FFTW
 * is not installed in this image; Intel MKL's FFTW3 interface (libmkl_rt.so) is, so the library is
 * resolved at run time:  libfftw3f.so.3  ->  libmkl_rt.so (FFTW3 wrappers)  ->  none (the caller
 * then times oracle_ct_c2c_f32 from smfft_oracle.c instead).  The plan is the FFTW-API batched
 * out-of-place C2C plan: fftwf_plan_many_dft(1,&N,howmany, in,NULL,1,N, out,NULL,1,N, sign, ESTIMATE).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>

typedef void* (*plan_many_t)(int, const int*, int, void*, const int*, int, int, void*, const int*, int, int, int, unsigned);
typedef void (*execute_t)(void*);
typedef void (*destroy_t)(void*);
typedef int (*init_threads_t)(void);
typedef void (*plan_threads_t)(int);

static void* lib;
static plan_many_t p_plan;
static execute_t p_exec;
static destroy_t p_destroy;
static char backend[64] = "none";

/* returns 1 if an FFTW3-API provider was found; may be called again to change the thread count */
int fftw_baseline_init(int threads) {
    const char* cands[] = {"libfftw3f_omp.so.3", "libfftw3f_threads.so.3", "libfftw3f.so.3", "libmkl_rt.so",
                           "/opt/conda/lib/libmkl_rt.so", "libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.1"};
    for (unsigned i = 0; i < sizeof(cands) / sizeof(cands[0]) && !lib; ++i) {
        void* h = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        p_plan = (plan_many_t)dlsym(h, "fftwf_plan_many_dft");
        p_exec = (execute_t)dlsym(h, "fftwf_execute");
        p_destroy = (destroy_t)dlsym(h, "fftwf_destroy_plan");
        if (p_plan && p_exec && p_destroy) {
            lib = h;
            snprintf(backend, sizeof backend, "%s", strstr(cands[i], "mkl") ? "mkl-fftw3-api" : "fftw3f");
            init_threads_t it = (init_threads_t)dlsym(h, "fftwf_init_threads");
            if (it) it();
        } else {
            dlclose(h);
        }
    }
    if (lib && threads > 0) {
        plan_threads_t pt = (plan_threads_t)dlsym(lib, "fftwf_plan_with_nthreads");
        if (pt) pt(threads);
        void (*mkl_set)(int) = (void (*)(int))dlsym(lib, "MKL_Set_Num_Threads");
        if (mkl_set) mkl_set(threads);
        void (*mkl_dyn)(int) = (void (*)(int))dlsym(lib, "MKL_Set_Dynamic");
        if (mkl_dyn) mkl_dyn(0);
    }
    return lib != 0;
}

const char* fftw_baseline_backend(void) { return backend; }

/* Executes the batched plan `reps` times; returns the best wall time in seconds, or -1. */
double fftw_baseline_c2c(const float* in, float* out, int N, int nFFTs, int inverse, int reps) {
    if (!lib) return -1.0;
    void* plan = p_plan(1, &N, nFFTs, (void*)in, 0, 1, N, (void*)out, 0, 1, N, inverse ? +1 : -1, 1u << 6 /*FFTW_ESTIMATE*/);
    if (!plan) return -1.0;
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        p_exec(plan);
        clock_gettime(CLOCK_MONOTONIC, &b);
        double t = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        if (t < best) best = t;
    }
    p_destroy(plan);
    return best;
}

/* Batch-parallel variant: the batch is cut into `nthreads` contiguous slices, each with its own
 * single-threaded FFTW-API plan (plans are created serially: the planner is not thread-safe;
 * fftwf_execute on distinct plans is), executed concurrently by pthreads.  This is how a batched
 * FFT is normally run on a many-core host, and it does not depend on the library's own threading
 * layer (MKL's OpenMP layer measured SLOWER with more threads on the 2 x 64-core GPU host).
 * Returns the best wall time in seconds over `reps` rounds, or -1. */
typedef struct { void* plan; pthread_barrier_t* bar; int reps; } slice_t;

static void* slice_main(void* arg) {
    slice_t* s = (slice_t*)arg;
    /* a caller whose main thread was pinned (OpenMP / torch affinity settings) would hand the same
     * one-CPU mask to every worker: ask for every CPU, the kernel intersects with the cpuset */
    cpu_set_t all;
    CPU_ZERO(&all);
    for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &all);
    pthread_setaffinity_np(pthread_self(), sizeof all, &all);
    for (int r = 0; r < s->reps; ++r) {
        pthread_barrier_wait(s->bar);
        p_exec(s->plan);
        pthread_barrier_wait(s->bar);
    }
    return 0;
}

double fftw_baseline_c2c_sliced(const float* in, float* out, int N, int nFFTs, int inverse, int reps, int nthreads) {
    if (!lib || nthreads < 1) return -1.0;
    if (nthreads > nFFTs) nthreads = nFFTs;
    slice_t* sl = (slice_t*)calloc((size_t)nthreads, sizeof(slice_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, (unsigned)nthreads + 1);
    int ok = 1;
    for (int t = 0; t < nthreads; ++t) {
        long f0 = (long)nFFTs * t / nthreads, f1 = (long)nFFTs * (t + 1) / nthreads;
        sl[t].plan = p_plan(1, &N, (int)(f1 - f0), (void*)(in + 2 * f0 * N), 0, 1, N, (void*)(out + 2 * f0 * N), 0, 1, N, inverse ? +1 : -1, 1u << 6);
        sl[t].bar = &bar;
        sl[t].reps = reps;
        if (!sl[t].plan) ok = 0;
    }
    double best = 1e30;
    if (ok) {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, slice_main, &sl[t]);
        for (int r = 0; r < reps; ++r) {
            struct timespec a, b;
            pthread_barrier_wait(&bar);
            clock_gettime(CLOCK_MONOTONIC, &a);
            pthread_barrier_wait(&bar);
            clock_gettime(CLOCK_MONOTONIC, &b);
            double t = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
            if (t < best) best = t;
        }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
    }
    for (int t = 0; t < nthreads; ++t) if (sl[t].plan) p_destroy(sl[t].plan);
    pthread_barrier_destroy(&bar);
    free(sl); free(th);
    return ok ? best : -1.0;
}
