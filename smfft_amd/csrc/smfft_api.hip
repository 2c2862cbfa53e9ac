// smfft_api.hip -- host side of libsmfft_amd.so: the C ABI of include/smfft.h and the reference's
// own C++-linkage entry points (include/smfft_reference_api.h).
//
// Mirrors, in behaviour (return codes, printed lines, timing convention), the host code of the
// three reference programs: CT:576-752 + :827-908, ST:299-384 + :457-530, RC:388-467 + :572-688.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/smfft.h"
#include "../../include/smfft_reference_api.h"
#include "smfft_host_util.hpp"
#include "smfft_launch.hpp"

namespace {

int g_device = 0;       // the reference's global `int device = 0` (CT:15)
int g_grid_cap = 12288; // workgroups per launch (grid-stride over tiles; 12288 = 12 or 16 rounds of the 4 or 3
                        // resident workgroups per CU; measured sweet spot); <= 0: one per tile
int g_nreuses = SMFFT_NREUSES;  // applications per slot in the `multiple` kernels (tests lower it)
std::once_flag g_env_once;

// launches may come from several host threads (per-GPU threads of a multi-GPU driver, the lanes of
// smfft_host_transform): the environment is read exactly once
void read_env() {
    std::call_once(g_env_once, [] {
        if (const char* e = getenv("SMFFT_GRID_CAP")) g_grid_cap = atoi(e);
        if (const char* e = getenv("SMFFT_DEVICE")) g_device = atoi(e);
    });
}

// count of FFT slots the `multiple` path touches (CT:669-683; ST:351; RC:438)
int ct_multiple_slots(int FFT_size, int nFFTs) {
    if (FFT_size == 32) return (nFFTs / (4 * SMFFT_NREUSES)) * 4;
    if (FFT_size == 64) return (nFFTs / (2 * SMFFT_NREUSES)) * 2;
    return nFFTs / SMFFT_NREUSES;
}

using smfft::launch_ct;
using smfft::launch_rc;
using smfft::launch_st;

// returns -1 for an unsupported length (nothing launched), else the launch status
int dispatch_ct(const float2* in, float2* out, int N, int count, int inverse, int reorder, int path, hipStream_t st) {
    switch (N) {
        case 32:   return launch_ct<32>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 64:   return launch_ct<64>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 128:  return launch_ct<128>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 256:  return launch_ct<256>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 512:  return launch_ct<512>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 1024: return launch_ct<1024>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 2048: return launch_ct<2048>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        case 4096: return launch_ct<4096>(in, out, count, inverse, reorder, path, g_grid_cap, g_nreuses, st);
        default:   return -1;
    }
}
int dispatch_st(const float2* in, float2* out, int N, int count, int path, hipStream_t st) {
    switch (N) {
        case 32:   return launch_st<32>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 64:   return launch_st<64>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 128:  return launch_st<128>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 256:  return launch_st<256>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 512:  return launch_st<512>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 1024: return launch_st<1024>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 2048: return launch_st<2048>(in, out, count, path, g_grid_cap, g_nreuses, st);
        case 4096: return launch_st<4096>(in, out, count, path, g_grid_cap, g_nreuses, st);
        default:   return -1;
    }
}
// FFT_size is the REAL length; the kernels are instantiated on the complex length L = FFT_size/2 (RC:404-428)
int dispatch_rc(const float2* in, float2* out, int FFT_size, int count, int inverse, int path, hipStream_t st) {
    switch (FFT_size) {
        case 512:  return launch_rc<256>(in, out, count, inverse, path, g_grid_cap, g_nreuses, st);
        case 1024: return launch_rc<512>(in, out, count, inverse, path, g_grid_cap, g_nreuses, st);
        case 2048: return launch_rc<1024>(in, out, count, inverse, path, g_grid_cap, g_nreuses, st);
        case 4096: return launch_rc<2048>(in, out, count, inverse, path, g_grid_cap, g_nreuses, st);
        default:   return -1;
    }
}

// One event-timed launch on stream 0, elapsed ms ADDED to *FFT_time (CT:598,660-662).
template <class F>
int timed(F&& launch, double* FFT_time) {
    read_env();
    GpuTimer timer;
    timer.Start();
    int rc = launch();
    timer.Stop();
    if (rc == -1) printf("Error wrong FFT length!\n");
    else if (rc != 0) checkHipErrors((hipError_t)rc);
    if (FFT_time) *FFT_time += timer.Elapsed();
    return 0;
}

// ---- paired allocation ----------------------------------------------------------------------------
// Measured on MI355X (profiles/r01_chunk_map.txt, tools/chunk_map.py): the streaming rate of a kernel that
// reads one buffer and writes another depends on WHICH physical memory the two buffers are.  Separately
// allocated chunks fall into a few classes; input and output in the same class run the 4 GiB + 4 GiB N=1024
// batch in 1.51-1.55 ms, in different classes in 1.41-1.46 ms, and on most boxes one region (often the memory
// allocated last) is faster still as a WRITE target: 1.31-1.39 ms with any input -- while reading from it is slow
// (1.49-1.53 ms).  None of this is visible in the virtual addresses, so nothing is assumed: smfft_malloc_pair
// allocates as many buffer-sized chunks as the device has room for, times the external kernels' own access shape
// (the stream-copy kernel) from a reference chunk into every other chunk, then from every chunk into the best
// output, keeps the fastest (input, output) and frees the rest.  With 288 GB of HBM that is about 66 candidates
// for 4 GiB buffers and 4-5 s, nearly all of it hipMalloc / hipFree time.
struct PairRec { void* a; void* b; size_t searched; int device; bool pool_b = false; };   // searched: candidate size of a
                                                                                     // placement search, 0 = none; pool_b: b is from hipMallocAsync
PairRec g_pairs[64];
PairRec g_pair_cache = {nullptr, nullptr, 0, -1};   // the last searched pair that was released (see free_pair)
std::mutex g_pairs_mutex;   // the table is shared by the per-GPU host threads of a multi-GPU driver

// mean ms of a few stream-copy launches (the external kernels' access shape) over the whole buffers
float probe_copy_ms(const void* in, void* out, size_t bytes, int launches) {
    const long n = (long)(bytes / 8 / 4096 * 4096);
    if (n <= 0) return 0.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0.f;
    smfft::launch_stream_copy((const float2*)in, (float2*)out, n, 12288, 0);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) smfft::launch_stream_copy((const float2*)in, (float2*)out, n, 12288, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms / launches;
}

void set_pair(int slot, PairRec rec) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_pairs_mutex);
    g_pairs[slot] = rec;
}

int alloc_pair(size_t bytes, void** d_a, void** d_b) {
    size_t free_mem = 0, total_mem = 0;
    *d_a = *d_b = nullptr;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lock(g_pairs_mutex);
        for (int i = 0; i < 64; ++i) if (!g_pairs[i].a) { slot = i; g_pairs[i].a = (void*)&g_pairs[i]; break; }   // reserved
    }
    const bool want_search = getenv("SMFFT_NO_PAIR_PLACEMENT") == nullptr;
    int device = -1;
    (void)hipGetDevice(&device);
    if (want_search && slot >= 0) {   // a searched pair released earlier on this device that is large enough: no new search
        std::lock_guard<std::mutex> lock(g_pairs_mutex);
        if (g_pair_cache.a && g_pair_cache.device == device && g_pair_cache.searched >= bytes) {
            g_pairs[slot] = g_pair_cache;
            *d_a = g_pair_cache.a;
            *d_b = g_pair_cache.b;
            g_pair_cache = {nullptr, nullptr, 0, -1};
            return 0;
        }
    }
    // Shortcut before the search: on most boxes the memory the stream-ordered allocator (hipMallocAsync) hands out at
    // this point IS the fast write region (tools/microbench/alloc_kinds.hip, tools/async_pool_probe.py: hipMalloc input +
    // pool output 1.33-1.34 ms on two boxes of three, 1.50 ms on the third, where the search still found 1.33).  One copy
    // probe decides: at 6.25 TB/s or more the pair is in the class the search would end in, and it cost 0.3 s, not 4-5.
    if (want_search && slot >= 0 && bytes >= (1ull << 30) && bytes <= (16ull << 30) && getenv("SMFFT_NO_POOL_SHORTCUT") == nullptr) {
        void *in = nullptr, *out = nullptr;
        if (hipMalloc(&in, bytes) == hipSuccess) {
            if (hipMallocAsync(&out, bytes, 0) == hipSuccess && hipStreamSynchronize(0) == hipSuccess) {
                const float ms = probe_copy_ms(in, out, bytes, 3);
                if (ms > 0.f && 2.0 * (double)bytes / (ms * 1e-3) >= 6.25e12) {   // fast write region: 6.4-6.5; best ordinary class: <= 6.17
                    *d_a = in;
                    *d_b = out;
                    PairRec rec = {in, out, 0, device};
                    rec.pool_b = true;
                    set_pair(slot, rec);
                    return 0;
                }
                (void)hipFreeAsync(out, 0);
                (void)hipStreamSynchronize(0);
            }
            (void)hipFree(in);
        }
        (void)hipGetLastError();
    }
    if (want_search && slot >= 0 && bytes >= (1ull << 30) && bytes <= (16ull << 30) && hipMemGetInfo(&free_mem, &total_mem) == hipSuccess) {
        // candidates of at least 4 GiB so that about 70 of them cover the whole memory (hipMalloc + hipFree cost
        // about 15 ms per GiB whatever the chunk size: ~4 s for 288 GB; SMFFT_PAIR_SEARCH_CHUNKS=k stops after k
        // candidates, e.g. 12 = 0.6 s, which still separates the two common classes but rarely reaches the fast
        // write region)
        constexpr int kMaxChunks = 72;
        void* chunk[kMaxChunks];
        int n = 0, limit = kMaxChunks;
        if (const char* e = getenv("SMFFT_PAIR_SEARCH_CHUNKS")) limit = atoi(e) < 2 ? 2 : (atoi(e) > kMaxChunks ? kMaxChunks : atoi(e));
        const size_t reserve = 6ull << 30;   // left to the rest of the application while the search runs
        const size_t chunk_bytes = bytes > (4ull << 30) ? bytes : (4ull << 30);
        while (n < limit && hipMemGetInfo(&free_mem, &total_mem) == hipSuccess && free_mem > chunk_bytes + reserve
               && hipMalloc(&chunk[n], chunk_bytes) == hipSuccess) ++n;
        (void)hipGetLastError();
        if (n >= 2) {
            int best_in = 0, best_out = 1;
            if (n > 2) {
                const int ref = n / 2;
                float best = 1e30f;
                for (int j = 0; j < n; ++j) {          // best write target for a reference input
                    if (j == ref) continue;
                    const float ms = probe_copy_ms(chunk[ref], chunk[j], bytes, 3);
                    if (ms > 0.f && ms < best) { best = ms; best_out = j; }
                }
                best = 1e30f;
                for (int i = 0; i < n; ++i) {          // best input for that target
                    if (i == best_out) continue;
                    const float ms = probe_copy_ms(chunk[i], chunk[best_out], bytes, 3);
                    if (ms > 0.f && ms < best) { best = ms; best_in = i; }
                }
            }
            for (int i = 0; i < n; ++i)
                if (i != best_in && i != best_out) (void)hipFree(chunk[i]);
            *d_a = chunk[best_in];
            *d_b = chunk[best_out];
            set_pair(slot, {*d_a, *d_b, n > 2 ? chunk_bytes : 0, device});
            return 0;
        }
        if (n == 1) (void)hipFree(chunk[0]);
    }
    if (hipMalloc(d_a, bytes) != hipSuccess) { set_pair(slot, {nullptr, nullptr, 0, -1}); return 1; }
    if (hipMalloc(d_b, bytes) != hipSuccess) { (void)hipFree(*d_a); *d_a = nullptr; set_pair(slot, {nullptr, nullptr, 0, -1}); return 1; }
    set_pair(slot, {*d_a, *d_b, 0, device});
    return 0;
}

// A searched pair costs seconds to find, so the most recently released one is kept (8 GiB or more of device memory)
// for the next smfft_malloc_pair of this device that fits into it -- the L3 wrappers are typically called several
// times in a row -- until smfft_pair_cache_release() or a newer searched pair replaces it.
int free_pair(void* d_a) {
    PairRec rec = {nullptr, nullptr, 0, -1}, evicted = {nullptr, nullptr, 0, -1};
    {
        std::lock_guard<std::mutex> lock(g_pairs_mutex);
        for (int i = 0; i < 64; ++i) {
            if (g_pairs[i].a == d_a && d_a) {
                rec = g_pairs[i];
                g_pairs[i] = {nullptr, nullptr, 0, -1};
                break;
            }
        }
        if (rec.a && rec.searched && getenv("SMFFT_NO_PAIR_CACHE") == nullptr) {
            evicted = g_pair_cache;
            g_pair_cache = rec;
            rec = evicted;          // free the previous occupant (possibly nothing) instead
            if (!rec.a) return 0;
        }
    }
    if (!rec.a) return (int)hipFree(d_a);
    if (rec.pool_b) {
        int rc = (int)hipFree(rec.a) | (int)hipFreeAsync(rec.b, 0);
        return rc | (int)hipStreamSynchronize(0);
    }
    return (int)hipFree(rec.a) | (int)hipFree(rec.b);
}

int release_pair_cache() {
    PairRec rec;
    {
        std::lock_guard<std::mutex> lock(g_pairs_mutex);
        rec = g_pair_cache;
        g_pair_cache = {nullptr, nullptr, 0, -1};
    }
    if (!rec.a) return 0;
    return (int)hipFree(rec.a) | (int)hipFree(rec.b);
}

int select_device() {
    read_env();
    int devCount = 0;
    checkHipErrors(hipGetDeviceCount(&devCount));
    if (devCount > g_device) checkHipErrors(hipSetDevice(g_device));
    return devCount;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

void smfft_init(void) {
    read_env();
    // cudaDeviceSetCacheConfig / cudaDeviceSetSharedMemConfig (CT:579-580) have no CDNA meaning.
}

int smfft_ct_external_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, int reorder, double* FFT_time) {
    return timed([&] { return dispatch_ct((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs, inverse != 0, reorder != 0, 0, 0); }, FFT_time);
}

int smfft_ct_multiple_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, int reorder, double* FFT_time) {
    if (nFFTs / SMFFT_NREUSES == 0) {
        if (FFT_time) *FFT_time = -1;
        return 1;
    }
    const int slots = ct_multiple_slots(FFT_size, nFFTs);
    return timed([&] { return dispatch_ct((const float2*)d_input, (float2*)d_output, FFT_size, slots, inverse != 0, reorder != 0, 1, 0); }, FFT_time);
}

int smfft_st_external_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    return timed([&] { return dispatch_st((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs, 0, 0); }, FFT_time);
}
int smfft_st_external_benchmark_dir(const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time) {
    if (inverse) return smfft_st_external_benchmark(d_input, d_output, FFT_size, nFFTs, FFT_time);
    // the engine is the same autosort Stockham plan for both signs: forward natural order = Engine<N, 0, 1>
    return timed([&] { return dispatch_ct((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs, false, true, 0, 0); }, FFT_time);
}
int smfft_st_multiple_benchmark(const void* d_input, void* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    return timed([&] { return dispatch_st((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs / SMFFT_NREUSES, 1, 0); }, FFT_time);
}

int smfft_rc_external_benchmark(const float* d_input, float* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time) {
    return timed([&] { return dispatch_rc((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs, inverse != 0, 0, 0); }, FFT_time);
}
int smfft_rc_multiple_benchmark(const float* d_input, float* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    return timed([&] { return dispatch_rc((const float2*)d_input, (float2*)d_output, FFT_size, nFFTs / SMFFT_NREUSES, 0, 1, 0); }, FFT_time);
}

int smfft_launch(int family, int path, const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, int reorder, void* hip_stream) {
    read_env();
    hipStream_t st = (hipStream_t)hip_stream;
    const float2* in = (const float2*)d_input;
    float2* out = (float2*)d_output;
    if (family == 0) {
        int count = path ? ct_multiple_slots(FFT_size, nFFTs) : nFFTs;
        return dispatch_ct(in, out, FFT_size, count, inverse != 0, reorder != 0, path, st);
    }
    if (family == 1) {
        // the Stockham program is the + sign transform (ST:76); inverse = 0 asks for the forward extension, which is
        // the same autosort engine with the other sign = the natural-order CT variant (external and multiple alike)
        if (!inverse) return dispatch_ct(in, out, FFT_size, path ? nFFTs / SMFFT_NREUSES : nFFTs, false, true, path, st);
        return dispatch_st(in, out, FFT_size, path ? nFFTs / SMFFT_NREUSES : nFFTs, path, st);
    }
    if (family == 2) return dispatch_rc(in, out, FFT_size, path ? nFFTs / SMFFT_NREUSES : nFFTs, inverse != 0, path, st);
    return -1;
}

int smfft_copy_launch(const void* d_input, void* d_output, long long n_float2, void* hip_stream) {
    read_env();
    return smfft::launch_stream_copy((const float2*)d_input, (float2*)d_output, (long)n_float2, g_grid_cap, (hipStream_t)hip_stream);
}

// ---- L3 wrappers ---------------------------------------------------------------------------------
int smfft_gpu_ct(const void* h_input, void* h_output, int FFT_size, int nFFTs, int inverse, int reorder, int nRuns, double* single_ex_time, double* multi_ex_time) {
    select_device();
    // edge cases the reference rejects (CT:835-836)
    if (FFT_size == 32 && (nFFTs % 4) != 0) return 1;
    if (FFT_size == 64 && (nFFTs % 2) != 0) return 1;

    size_t free_mem, total_mem;
    checkHipErrors(hipMemGetInfo(&free_mem, &total_mem));
    if (DEBUG) printf("\n  Device has %0.3f MB of total memory, which %0.3f MB is available.\n", (float)total_mem / (1024.0 * 1024.0), (float)free_mem / (1024.0 * 1024.0));
    const size_t bytes = (size_t)FFT_size * nFFTs * sizeof(float2);
    if (2 * bytes > free_mem) {
        printf("Error: Not enough memory! Input data is too big for the device.\n");
        return 1;
    }
    float2 *d_input, *d_output;
    if (alloc_pair(bytes, (void**)&d_input, (void**)&d_output)) checkHipErrors(hipErrorOutOfMemory);

    double time_FFT_external = 0, time_FFT_multiple = 0;
    // The reference re-uploads h_input before every one of the 2*nRuns launches (CT:868,884), outside
    // the timed region.  The kernels never modify d_input, so one upload is equivalent and saves
    // (2*nRuns - 1) pageable 4 GiB copies at the README batch.
    checkHipErrors(hipMemcpy(d_input, h_input, bytes, hipMemcpyHostToDevice));
    if (MULTIPLE) {
        if (DEBUG) printf("  Running shared memory FFT (Cooley-Tukey) 100 times per GPU kernel (eliminates device memory)... ");
        smfft_init();
        double total = 0;
        for (int f = 0; f < nRuns; f++) {
            smfft_ct_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, inverse, reorder, &total);
        }
        time_FFT_multiple = total / nRuns;
        if (DEBUG) printf("done in %g ms.\n", time_FFT_multiple);
        if (multi_ex_time) *multi_ex_time = time_FFT_multiple;
    }
    checkHipErrors(hipGetLastError());
    if (EXTERNAL) {
        if (DEBUG) printf("  Running shared memory FFT (Cooley-Tukey)... ");
        smfft_init();
        double total = 0;
        for (int f = 0; f < nRuns; f++) {
            smfft_ct_external_benchmark(d_input, d_output, FFT_size, nFFTs, inverse, reorder, &total);
        }
        time_FFT_external = total / nRuns;
        if (DEBUG) printf("done in %g ms.\n", time_FFT_external);
        if (single_ex_time) *single_ex_time = time_FFT_external;
    }
    checkHipErrors(hipGetLastError());
    printf("  SH FFT normal = %0.3f ms; SM FFT multiple times = %0.3f ms\n", time_FFT_external, time_FFT_multiple);
    // MULTIPLE runs first, so d_output holds the EXTERNAL result (CT:862-890,898)
    checkHipErrors(hipMemcpy(h_output, d_output, bytes, hipMemcpyDeviceToHost));
    checkHipErrors(hipGetLastError());
    checkHipErrors((hipError_t)free_pair(d_input));
    return 0;
}

int smfft_gpu_st(const void* h_input, void* h_output, int FFT_size, int nFFTs, int nRuns, double* single_ex_time, double* multi_ex_time) {
    select_device();
    size_t free_mem, total_mem;
    checkHipErrors(hipMemGetInfo(&free_mem, &total_mem));
    const size_t bytes = (size_t)FFT_size * nFFTs * sizeof(float2);
    if (2 * bytes > free_mem) {
        printf("Error: Not enough memory! Input data is too big for the device.\n");
        return 1;
    }
    float2 *d_input, *d_output;
    if (alloc_pair(bytes, (void**)&d_input, (void**)&d_output)) checkHipErrors(hipErrorOutOfMemory);
    double time_FFT_external = 0, time_FFT_multiple = 0;
    checkHipErrors(hipMemcpy(d_input, h_input, bytes, hipMemcpyHostToDevice));   // once (upstream: per run, ST:496,508)
    if (MULTIPLE) {
        smfft_init();
        double total = 0;
        for (int f = 0; f < nRuns; f++) {
            smfft_st_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, &total);
        }
        time_FFT_multiple = total / nRuns;
        if (multi_ex_time) *multi_ex_time = time_FFT_multiple;
    }
    if (EXTERNAL) {
        smfft_init();
        double total = 0;
        for (int f = 0; f < nRuns; f++) {
            smfft_st_external_benchmark(d_input, d_output, FFT_size, nFFTs, &total);
        }
        time_FFT_external = total / nRuns;
        if (single_ex_time) *single_ex_time = time_FFT_external;
    }
    checkHipErrors(hipMemcpy(h_output, d_output, bytes, hipMemcpyDeviceToHost));
    checkHipErrors(hipGetLastError());
    checkHipErrors((hipError_t)free_pair(d_input));
    printf("  SH FFT normal = %0.3f ms; SM FFT multiple times = %0.3f ms\n", time_FFT_external, time_FFT_multiple);
    return 0;
}

int smfft_gpu_r2c(void* h_output, const float* h_input, int FFT_size, int nFFTs, int nRuns) {
    select_device();
    size_t free_memory, total_memory;
    checkHipErrors(hipMemGetInfo(&free_memory, &total_memory));
    double FFT_external_time = 0, FFT_multiple_time = 0;
    const size_t input_size_bytes = (size_t)FFT_size * nFFTs * sizeof(float);
    const size_t output_size_bytes = (size_t)(FFT_size >> 1) * nFFTs * sizeof(float2);
    if ((input_size_bytes + output_size_bytes) > free_memory) {
        printf("Error not enough free memory!\n");
        return 1;
    }
    float* d_input;
    float2* d_output;
    if (alloc_pair(input_size_bytes, (void**)&d_input, (void**)&d_output)) checkHipErrors(hipErrorOutOfMemory);   // both sides are N*nFFTs*4 bytes
    checkHipErrors(hipMemcpy(d_input, h_input, input_size_bytes, hipMemcpyHostToDevice));
    if (MULTIPLE) {
        for (int r = 0; r < nRuns; r++) {
            smfft_init();
            smfft_rc_multiple_benchmark(d_input, (float*)d_output, FFT_size, nFFTs, &FFT_multiple_time);
        }
    }
    checkHipErrors(hipMemset(d_output, 0, output_size_bytes));
    if (EXTERNAL) {
        for (int r = 0; r < nRuns; r++) {
            smfft_init();
            smfft_rc_external_benchmark(d_input, (float*)d_output, FFT_size, nFFTs, 0, &FFT_external_time);
        }
    }
    printf("  smFFT R2C time: ex: %0.3f ms; mul: %0.3f ms\n", FFT_external_time / nRuns, FFT_multiple_time / nRuns);
    checkHipErrors(hipMemcpy(h_output, d_output, output_size_bytes, hipMemcpyDeviceToHost));
    checkHipErrors((hipError_t)free_pair(d_input));
    return 0;
}

int smfft_gpu_c2r(float* h_output, const void* h_input, int FFT_size, int nFFTs, int nRuns) {
    select_device();
    size_t free_memory, total_memory;
    checkHipErrors(hipMemGetInfo(&free_memory, &total_memory));
    double FFT_external_time = 0;
    const size_t input_size_bytes = (size_t)(FFT_size >> 1) * nFFTs * sizeof(float2);
    const size_t output_size_bytes = (size_t)FFT_size * nFFTs * sizeof(float);
    if ((input_size_bytes + output_size_bytes) > free_memory) {
        printf("Error not enough free memory!\n");
        return 1;
    }
    float2* d_input;
    float* d_output;
    if (alloc_pair(input_size_bytes, (void**)&d_input, (void**)&d_output)) checkHipErrors(hipErrorOutOfMemory);
    checkHipErrors(hipMemcpy(d_input, h_input, input_size_bytes, hipMemcpyHostToDevice));
    if (EXTERNAL) {
        for (int r = 0; r < nRuns; r++) {
            smfft_init();
            smfft_rc_external_benchmark((const float*)d_input, d_output, FFT_size, nFFTs, 1, &FFT_external_time);
        }
    }
    printf("  smFFT C2R time: ex: %0.3f ms;\n", FFT_external_time / nRuns);
    checkHipErrors(hipMemcpy(h_output, d_output, output_size_bytes, hipMemcpyDeviceToHost));
    checkHipErrors((hipError_t)free_pair(d_input));
    return 0;
}

// ---- tuning / introspection ------------------------------------------------------------------------
void smfft_set_grid_cap(int max_workgroups) { read_env(); g_grid_cap = max_workgroups; }
void smfft_set_nreuses(int n) { g_nreuses = n > 0 ? n : SMFFT_NREUSES; }
int smfft_get_nreuses(void) { return g_nreuses; }
int smfft_get_grid_cap(void) { read_env(); return g_grid_cap; }
int smfft_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int smfft_set_device(int device) { read_env(); g_device = device; return (int)hipSetDevice(device); }
const char* smfft_version(void) { return "smfft_amd 0.1 (gfx950)"; }

int smfft_malloc_pair(unsigned long long bytes, void** d_read, void** d_written) { read_env(); return alloc_pair((size_t)bytes, d_read, d_written); }
int smfft_free_pair(void* d_read) { return free_pair(d_read); }
int smfft_pair_cache_release(void) { return release_pair_cache(); }
void* smfft_malloc(unsigned long long bytes) { void* p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
int smfft_free(void* d_ptr) { return (int)hipFree(d_ptr); }
int smfft_memcpy_h2d(void* d_dst, const void* h_src, unsigned long long bytes) { return (int)hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice); }
int smfft_memcpy_d2h(void* h_dst, const void* d_src, unsigned long long bytes) { return (int)hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost); }
int smfft_memcpy_d2d(void* d_dst, const void* d_src, unsigned long long bytes) { return (int)hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice); }
int smfft_memset(void* d_ptr, int value, unsigned long long bytes) { return (int)hipMemset(d_ptr, value, bytes); }
int smfft_synchronize(void) { return (int)hipDeviceSynchronize(); }

}  // extern "C"

// =================================================================================================
// The reference's C++-linkage names (what its FFT.c harnesses bind; include/smfft_reference_api.h)
// =================================================================================================
void FFT_init() { smfft_init(); }

int FFT_external_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, bool inverse, bool reorder, double* FFT_time) {
    return smfft_ct_external_benchmark(d_input, d_output, FFT_size, nFFTs, inverse, reorder, FFT_time);
}
int FFT_multiple_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, bool inverse, bool reorder, double* FFT_time) {
    return smfft_ct_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, inverse, reorder, FFT_time);
}
void FFT_external_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    smfft_st_external_benchmark(d_input, d_output, FFT_size, nFFTs, FFT_time);
}
void FFT_multiple_benchmark(float2* d_input, float2* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    smfft_st_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, FFT_time);
}
void FFT_external_benchmark(float* d_input, float* d_output, int FFT_size, int nFFTs, int inverse, double* FFT_time) {
    smfft_rc_external_benchmark(d_input, d_output, FFT_size, nFFTs, inverse, FFT_time);
}
void FFT_multiple_benchmark(float* d_input, float* d_output, int FFT_size, int nFFTs, double* FFT_time) {
    smfft_rc_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, FFT_time);
}

int GPU_smFFT_4elements(float2* h_input, float2* h_output, int FFT_size, int nFFTs, bool inverse, bool reorder, int nRuns, double* single_ex_time, double* multi_ex_time) {
    return smfft_gpu_ct(h_input, h_output, FFT_size, nFFTs, inverse, reorder, nRuns, single_ex_time, multi_ex_time);
}
int GPU_FFT_C2C_Stockham(float2* h_input, float2* h_smFFT_output, int FFT_size, int nFFTs, int nRuns, double* single_ex_time, double* multi_ex_time) {
    return smfft_gpu_st(h_input, h_smFFT_output, FFT_size, nFFTs, nRuns, single_ex_time, multi_ex_time);
}
int GPU_smFFT_R2C(float2* h_output, float* h_input, int FFT_size, int nFFTs, int nRuns) { return smfft_gpu_r2c(h_output, h_input, FFT_size, nFFTs, nRuns); }
int GPU_smFFT_C2R(float* h_output, float2* h_input, int FFT_size, int nFFTs, int nRuns) { return smfft_gpu_c2r(h_output, h_input, FFT_size, nFFTs, nRuns); }
