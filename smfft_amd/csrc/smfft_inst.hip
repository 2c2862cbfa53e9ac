// smfft_inst.hip -- instantiates every kernel of ONE transform length; compiled once per length
// with -DSMFFT_N=<32..4096> (see Makefile).
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "smfft_kernels.hpp"
#include "smfft_launch.hpp"

#ifndef SMFFT_N
#error "compile with -DSMFFT_N=<transform length>"
#endif
// SMFFT_INST_PART: 0 (default: tools that compile this file by itself) = everything of the length in one object; the Makefile builds two
// objects per length -- 1 = the external (HBM-bound) kernels and the dispatch, 2 = the in-LDS (`multiple`) kernels -- because the two
// kinds want different code generation: the in-LDS kernels of N >= 128 run 2 ... 5 % faster without LLVM's post-RA machine scheduler
// (-mllvm -enable-post-misched=0), the external kernels 1.5 ... 2.7 % SLOWER, N = 32's lane engine 15 % slower
// (profiles/r06_post_misched.txt); the flags per length are the Makefile's MULT_FLAGS_<N> (tools/inst_flags.py reads them for the ISA tests).
#ifndef SMFFT_INST_PART
#define SMFFT_INST_PART 0
#endif

#define SMFFT_PASTE3_(a, b, c) a##b##c
#define SMFFT_PASTE3(a, b, c) SMFFT_PASTE3_(a, b, c)
#define CT_CLASS(suffix) SMFFT_PASTE3(FFT_, SMFFT_N, suffix)

namespace smfft {

// the in-LDS half of launch_ct / launch_st / launch_rc (path 1 or 2), defined in part 2
template <int N>
int launch_ct_multiple(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, const LaunchOptions& opt, hipStream_t stream);
template <int N>
int launch_st_multiple(const float2* d_input, float2* d_output, int count, const LaunchOptions& opt, hipStream_t stream);
template <int L>
int launch_rc_multiple(const float2* d_input, float2* d_output, int count, int inverse, const LaunchOptions& opt, hipStream_t stream);
template <>
int launch_ct_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, const LaunchOptions& opt, hipStream_t stream);
template <>
int launch_st_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, const LaunchOptions& opt, hipStream_t stream);
#if SMFFT_N >= 256 && SMFFT_N <= 2048
template <>
int launch_rc_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, const LaunchOptions& opt, hipStream_t stream);
#endif
#define ST_CLASS SMFFT_PASTE3(FFT_, SMFFT_N, )

#if SMFFT_INST_PART != 1
// One launch of a compact (in-LDS) kernel over `count` FFT slots.  Unbalanced: one chain per workgroup, grid-strided under the
// grid cap.  Balanced (the default when the batch is more chains than the chip holds at once): a persistent grid of the
// co-resident workgroups, each owning an equal share of the launch's ntiles * nreuses applications (MultipleSchedule).
using CompactKernel = void (*)(const float2*, float2*, int, int, MultipleSchedule);
static int launch_compact(CompactKernel kernel, const float2* d_input, float2* d_output, int count, const LaunchOptions& opt, hipStream_t stream) {
    using G = Geometry<SMFFT_N>;
    constexpr int threads = G::kCompactThreads;
    const int ntiles = (count + G::kCompactFfts - 1) / G::kCompactFfts;
    const int nreuses = opt.nreuses, balance = opt.balance;
    // (rotate: the waves' scheduling priority rotates every 2^15 shader clocks = 14 us by default; smfft_engine.hpp, WavePriority;
    //  sweep of the period: profiles/r04_priority_rotation.txt)
    MultipleSchedule sch = {};
    sch.rotate = opt.rotate;
    sch.delay_chain = -1;
    int grid = grid_for(count, G::kCompactFfts, opt.grid_cap);
    // Which schedule.  More chains than the chip holds workgroups: the balanced persistent grid and
    // rotating priorities -- no tail, co-resident chains end together: +12-24 % on the README batches (1.28 rounds' worth of
    // chains), still +2-5 % at 8.4 rounds (profiles/r05_schedule_crossover.txt; with round 4's hand-over, whose cache write-backs
    // grew with the launch, it lost beyond four rounds).  At most one round: one chain per workgroup, priorities rotating.
    // Beyond what a buffer of hand-over words holds (kScheduleMaxChains chains): one chain per workgroup, grid-strided, the
    // arbiter's own order.  balance >= 2 (tests): that many workgroups, as if the chip held no more.
    // A caller who caps the grid below what the chip holds (to leave CUs to other work) keeps that cap: no persistent grid then.
    const int resident = resident_workgroups((const void*)kernel, threads);
    const int slots = balance >= 2 ? balance : resident;
    const bool short_launch = slots <= 0 || balance >= 2 || ntiles <= kScheduleMaxChains;
    const bool capped = balance < 2 && opt.grid_cap > 0 && opt.grid_cap < slots;
    if (!short_launch) sch.rotate = 0;
    // a launch that is being CAPTURED into a graph keeps one chain per workgroup: the balanced grid's hand-over words carry the
    // epoch of one launch (a replayed graph would find them set)
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    int ticket = -1;
    unsigned base = 0;
    unsigned* flags = nullptr;
    const bool balanced = balance && nreuses > 1 && short_launch && !capped && capture == hipStreamCaptureStatusNone && slots > 0 && ntiles > slots && ntiles <= kScheduleMaxChains;
    if (balanced) {
        flags = schedule_acquire(ntiles, stream, &base, &ticket);
        const long total = (long)ntiles * nreuses;
        const long per_wg = (total + slots - 1) / slots;            // > nreuses, so a chain straddles at most two workgroups
        if (flags && per_wg < (1l << 30)) {
            sch.per_wg = (int)per_wg;
            sch.base = base;
            sch.flags = flags;
            sch.wait_ticks = (opt.handoff_wait_us < 40000000u ? opt.handoff_wait_us : 40000000u) * 100u;      // 100 MHz ticks in 32 bits: at most 40 s
            sch.delay_chain = opt.delay_chain;
            sch.delay_ticks = opt.delay_ms * 100000u;
            sch.delay_after_commit = opt.delay_after_commit;
            grid = (int)((total + per_wg - 1) / per_wg);
        }
    }
    sch.residency = residency_probe();
    if (sch.residency) note_resident_workgroups(resident);          // non-null only inside smfft_measure_multiple_residency
    static const bool debug = getenv("SMFFT_SCHEDULE_DEBUG") != nullptr;
    if (debug) printf("smfft multiple N=%d: %d chains x %d applications, %d co-resident workgroups (from the kernel's registers and LDS), grid %d, %d applications per workgroup%s\n", SMFFT_N, ntiles, nreuses,
                      resident, grid, sch.per_wg, sch.per_wg ? "" : " (one chain at a time)");
    static const char* trace_file = getenv("SMFFT_SCHEDULE_TRACE");          // experiments: one line per workgroup of the LAST launch
    if (trace_file && (hipMalloc((void**)&sch.trace, (size_t)grid * kTraceWords * 8) != hipSuccess || hipMemset(sch.trace, 0, (size_t)grid * kTraceWords * 8) != hipSuccess)) sch.trace = nullptr;
    kernel<<<dim3(grid), dim3(threads), 0, stream>>>(d_input, d_output, count, nreuses, sch);
    const int rc = (int)hipGetLastError();
    if (ticket >= 0) schedule_release(ticket, stream);
    if (sch.trace) {
        std::vector<unsigned long long> host((size_t)grid * kTraceWords);
        (void)hipMemcpy(host.data(), sch.trace, host.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(sch.trace);
        if (FILE* f = fopen(trace_file, "w")) {
            fprintf(f, "# N=%d chains=%d nreuses=%d grid=%d per_wg=%d : block start end hw_id xcc_id start_100MHz end_100MHz, then per piece (up to four): tile_in_LDS applications_done tile_stored (100 MHz)\n", SMFFT_N, ntiles, nreuses, grid, sch.per_wg);
            for (int i = 0; i < grid; ++i) {
                const unsigned long long* w = &host[(size_t)i * kTraceWords];
                fprintf(f, "%d %llu %llu %llx %llx %llu %llu", i, w[0], w[1], w[2], w[3], w[4], w[5]);
                for (int k = 6; k < kTraceWords; ++k) fprintf(f, " %llu", w[k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return rc;
}

template <>
int launch_ct_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, const LaunchOptions& opt, hipStream_t stream) {
    // in-LDS path: compact workgroups (one wave per 1024 elements for N <= 1024, one FFT per workgroup above)
#if SMFFT_N >= 64
    if (path == 2 && reorder) {     // no cross-application fusion (what one call of the device function costs)
        if (!inverse) return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_forward)>, d_input, d_output, count, opt, stream);
        return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_inverse)>, d_input, d_output, count, opt, stream);
    }
#endif
#if SMFFT_N == 32
    if (path == 2 && reorder) {     // the pair engine with an image load and an image store per application
        if (!inverse) return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_forward)>, d_input, d_output, count, opt, stream);
        return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_inverse)>, d_input, d_output, count, opt, stream);
    }
#endif
#if SMFFT_N <= 64
    if (path == 2 && !reorder) {    // the lane engines' no-reorder kernels, unfused the same way (N >= 128: the planar kernels re-read the image as they are)
        if (!inverse) return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_forward_noreorder)>, d_input, d_output, count, opt, stream);
        return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_inverse_noreorder)>, d_input, d_output, count, opt, stream);
    }
#endif
    if (!inverse && reorder)  return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_forward)>, d_input, d_output, count, opt, stream);
    if (!inverse && !reorder) return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_forward_noreorder)>, d_input, d_output, count, opt, stream);
    if (inverse && reorder)   return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_inverse)>, d_input, d_output, count, opt, stream);
    return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_inverse_noreorder)>, d_input, d_output, count, opt, stream);
}

template <>
int launch_st_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, const LaunchOptions& opt, hipStream_t stream) {
    return launch_compact(FFT_GPU_multiple<ST_CLASS>, d_input, d_output, count, opt, stream);
}
#if SMFFT_N >= 256 && SMFFT_N <= 2048
template <>
int launch_rc_multiple<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, const LaunchOptions& opt, hipStream_t stream) {
    if (!inverse) return launch_compact(FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_forward>, d_input, d_output, count, opt, stream);
    return launch_compact(FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_inverse>, d_input, d_output, count, opt, stream);
}
#endif
#endif  // SMFFT_INST_PART != 1

#if SMFFT_INST_PART != 2
template <>
int launch_ct<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, const LaunchOptions& opt, hipStream_t stream) {
    if (count <= 0) return 0;
    dim3 grid(grid_for(count, 4096 / SMFFT_N, opt.grid_cap)), block(256);
    const int pace = opt.pace;
    if (path == 0) {
#if SMFFT_N == 4096
#define SMFFT_EXTERNAL_REORDER_KERNEL SMFFT_DIT_external_occ3
#else
#define SMFFT_EXTERNAL_REORDER_KERNEL SMFFT_DIT_external
#endif
        if (!inverse && reorder)  SMFFT_EXTERNAL_REORDER_KERNEL<CT_CLASS(_forward)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
        if (!inverse && !reorder) SMFFT_DIT_external<CT_CLASS(_forward_noreorder)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
        if (inverse && reorder)   SMFFT_EXTERNAL_REORDER_KERNEL<CT_CLASS(_inverse)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
        if (inverse && !reorder)  SMFFT_DIT_external<CT_CLASS(_inverse_noreorder)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
        return (int)hipGetLastError();
    }
    // in-LDS path: compact workgroups (one wave per 1024 elements for N <= 1024, one FFT per workgroup above): part 2
    return launch_ct_multiple<SMFFT_N>(d_input, d_output, count, inverse, reorder, path, opt, stream);
}

#if SMFFT_N == 1024
int launch_stream_copy(const float2* d_input, float2* d_output, long n_float2, int grid_cap, int pace, hipStream_t stream) {
    long ntiles = n_float2 / 4096;
    if (ntiles <= 0) return 0;
    long g = (grid_cap > 0 && ntiles > grid_cap) ? grid_cap : ntiles;
    SMFFT_stream_copy<0><<<dim3((unsigned)g), dim3(256), 0, stream>>>(d_input, d_output, ntiles, pace);
    return (int)hipGetLastError();
}
int launch_stream_write(float2* d_output, long n_float2, int grid_cap, hipStream_t stream) {
    long ntiles = n_float2 / 4096;
    if (ntiles <= 0) return 0;
    long g = (grid_cap > 0 && ntiles > grid_cap) ? grid_cap : ntiles;
    SMFFT_stream_write<0><<<dim3((unsigned)g), dim3(256), 0, stream>>>(d_output, ntiles);
    return (int)hipGetLastError();
}
// the read-only pass never stores (its guard cannot be true), but the kernel needs somewhere a store COULD go that is not the
// caller's input buffer: 2 KiB per device, allocated at first use and kept
static float2* stream_read_sink() {
    static std::mutex mutex;
    static float2* sinks[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mutex);
    if (!sinks[dev] && hipMalloc((void**)&sinks[dev], 256 * sizeof(float2)) != hipSuccess) { (void)hipGetLastError(); sinks[dev] = nullptr; }
    return sinks[dev];
}
int launch_stream_read(const float2* d_input, long n_float2, int grid_cap, hipStream_t stream) {
    long ntiles = n_float2 / 4096;
    if (ntiles <= 0) return 0;
    float2* sink = stream_read_sink();
    if (!sink) return (int)hipErrorOutOfMemory;
    long g = (grid_cap > 0 && ntiles > grid_cap) ? grid_cap : ntiles;
    SMFFT_stream_read<0><<<dim3((unsigned)g), dim3(256), 0, stream>>>(d_input, sink, ntiles);
    return (int)hipGetLastError();
}
#endif

template <>
int launch_st<SMFFT_N>(const float2* d_input, float2* d_output, int count, int path, const LaunchOptions& opt, hipStream_t stream) {
    if (count <= 0) return 0;
    if (path != 0) return launch_st_multiple<SMFFT_N>(d_input, d_output, count, opt, stream);
    dim3 grid(grid_for(count, 4096 / SMFFT_N, opt.grid_cap)), block(256);
    const int pace = opt.pace;
#if SMFFT_N == 4096
    // same transform (Engine<4096, inverse, reorder>) through the occupancy-3 build, see SMFFT_DIT_external_occ3
    SMFFT_DIT_external_occ3<CT_CLASS(_inverse)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
#else
    FFT_GPU_external<ST_CLASS><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
#endif
    return (int)hipGetLastError();
}

#if SMFFT_N >= 256 && SMFFT_N <= 2048
template <>
int launch_rc<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int path, const LaunchOptions& opt, hipStream_t stream) {
    if (count <= 0) return 0;
    if (path != 0) return launch_rc_multiple<SMFFT_N>(d_input, d_output, count, inverse, opt, stream);
    dim3 grid(grid_for(count, 4096 / SMFFT_N, opt.grid_cap)), block(256);
    const int pace = opt.pace;
    if (!inverse) FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_forward><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
    else          FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_inverse><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
    return (int)hipGetLastError();
}
#endif

#endif  // SMFFT_INST_PART != 2

}  // namespace smfft
