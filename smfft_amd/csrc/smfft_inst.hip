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

#define SMFFT_PASTE3_(a, b, c) a##b##c
#define SMFFT_PASTE3(a, b, c) SMFFT_PASTE3_(a, b, c)
#define CT_CLASS(suffix) SMFFT_PASTE3(FFT_, SMFFT_N, suffix)

namespace smfft {

// One launch of a compact (in-LDS) kernel over `count` FFT slots.  Unbalanced: one chain per workgroup, grid-strided under the
// grid cap.  Balanced (the default when the batch is more chains than the chip holds at once): a persistent grid of the
// co-resident workgroups, each owning an equal share of the launch's ntiles * nreuses applications (MultipleSchedule).
using CompactKernel = void (*)(const float2*, float2*, int, int, MultipleSchedule);
static int launch_compact(CompactKernel kernel, const float2* d_input, float2* d_output, int count, int grid_cap, int nreuses, int balance, int rotate, hipStream_t stream,
                          int threads = Geometry<SMFFT_N>::kCompactThreads) {
    using G = Geometry<SMFFT_N>;
    const int ntiles = (count + G::kCompactFfts - 1) / G::kCompactFfts;
    // (rotate: the waves' scheduling priority rotates every 2^15 shader clocks = 14 us by default; smfft_kernels.hpp, WavePriority;
    //  sweep of the period: profiles/r04_priority_rotation.txt)
    MultipleSchedule sch = {0, 0u, nullptr, rotate, nullptr, nullptr};
    int grid = grid_for(count, G::kCompactFfts, grid_cap);
    // Which schedule.  Up to four rounds' worth of chains: the balanced persistent grid (when there is
    // more than one round) and rotating priorities -- no tail, co-resident chains end together: +14-24 % on the README batches.
    // A long launch (more than four rounds) is in a steady state of its own: workgroups start whenever an older one ends, the
    // tail is a few percent, and there the persistent grid measured 0-7 % SLOWER (N = 2048 most): one chain per workgroup,
    // grid-strided, the arbiter's own order.  balance >= 2 (tests): that many workgroups, as if the chip held no more.
    const int slots = balance >= 2 ? balance : resident_workgroups((const void*)kernel, threads);
    const bool short_launch = slots <= 0 || balance >= 2 || (long)ntiles <= 4l * slots;
    if (!short_launch) sch.rotate = 0;
    // a launch that is being CAPTURED into a graph keeps one chain per workgroup: the balanced grid's hand-off flags carry the
    // epoch of one launch (a replayed graph would find them set), and their buffer may have to be allocated here
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    if (balance && nreuses > 1 && short_launch && capture == hipStreamCaptureStatusNone) {
        if (slots > 0 && ntiles > slots) {
            const long total = (long)ntiles * nreuses;
            const long per_wg = (total + slots - 1) / slots;            // > nreuses, so a chain straddles at most two workgroups
            unsigned epoch = 0;
            unsigned* flags = schedule_flags(ntiles, stream, &epoch);
            if (flags && per_wg < (1l << 30)) {
                sch.per_wg = (int)per_wg;
                sch.epoch = epoch;
                sch.flags = flags;
                grid = (int)((total + per_wg - 1) / per_wg);
            }
        }
    }
    sch.residency = residency_probe();
    if (sch.residency) note_resident_workgroups(resident_workgroups((const void*)kernel, threads));          // non-null only inside smfft_measure_multiple_residency
    static const bool debug = getenv("SMFFT_SCHEDULE_DEBUG") != nullptr;
    if (debug) printf("smfft multiple N=%d: %d chains x %d applications, %d co-resident workgroups (from the kernel's registers and LDS), grid %d, %d applications per workgroup%s\n", SMFFT_N, ntiles, nreuses,
                      resident_workgroups((const void*)kernel, threads), grid, sch.per_wg, sch.per_wg ? "" : " (one chain at a time)");
    static const char* trace_file = getenv("SMFFT_SCHEDULE_TRACE");          // experiments: one line per workgroup of the LAST launch
    if (trace_file && hipMalloc((void**)&sch.trace, (size_t)grid * 32) != hipSuccess) sch.trace = nullptr;
    kernel<<<dim3(grid), dim3(threads), 0, stream>>>(d_input, d_output, count, nreuses, sch);
    if (sch.trace) {
        std::vector<unsigned long long> host((size_t)grid * 4);
        (void)hipMemcpy(host.data(), sch.trace, host.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(sch.trace);
        if (FILE* f = fopen(trace_file, "w")) {
            fprintf(f, "# N=%d chains=%d nreuses=%d grid=%d per_wg=%d : block start end hw_id xcc_id\n", SMFFT_N, ntiles, nreuses, grid, sch.per_wg);
            for (int i = 0; i < grid; ++i) fprintf(f, "%d %llu %llu %llx %llx\n", i, host[4 * i], host[4 * i + 1], host[4 * i + 2], host[4 * i + 3]);
            fclose(f);
        }
    }
    return (int)hipGetLastError();
}

template <>
int launch_ct<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream) {
    if (count <= 0) return 0;
    dim3 grid(grid_for(count, 4096 / SMFFT_N, grid_cap)), block(256);
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
    // in-LDS path: compact workgroups (one wave per 1024 elements for N <= 1024, one FFT per workgroup above)
#if SMFFT_PLANAR_SIZES(SMFFT_N)
    if (path == 2 && reorder) {     // no cross-application fusion (what one call of the device function costs)
        if (!inverse) return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_forward)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
        return launch_compact(SMFFT_DIT_multiple_unfused<CT_CLASS(_inverse)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    }
#endif
#if SMFFT_X2_SIZES(SMFFT_N)
    if (path == 1) {                // two virtual threads per lane: N = 2048 in one wave, N = 4096 in two
        constexpr int kThreads = Geometry<SMFFT_N>::kCompactThreads / 2;
        if (!inverse && reorder)  return launch_compact(SMFFT_DIT_multiple_x2<CT_CLASS(_forward)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream, kThreads);
        if (!inverse && !reorder) return launch_compact(SMFFT_DIT_multiple_x2<CT_CLASS(_forward_noreorder)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream, kThreads);
        if (inverse && reorder)   return launch_compact(SMFFT_DIT_multiple_x2<CT_CLASS(_inverse)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream, kThreads);
        return launch_compact(SMFFT_DIT_multiple_x2<CT_CLASS(_inverse_noreorder)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream, kThreads);
    }
#endif
    if (!inverse && reorder)  return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_forward)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    if (!inverse && !reorder) return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_forward_noreorder)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    if (inverse && reorder)   return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_inverse)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    return launch_compact(SMFFT_DIT_multiple<CT_CLASS(_inverse_noreorder)>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
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

#define ST_CLASS SMFFT_PASTE3(FFT_, SMFFT_N, )
template <>
int launch_st<SMFFT_N>(const float2* d_input, float2* d_output, int count, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream) {
    if (count <= 0) return 0;
    if (path != 0) return launch_compact(FFT_GPU_multiple<ST_CLASS>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    dim3 grid(grid_for(count, 4096 / SMFFT_N, grid_cap)), block(256);
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
int launch_rc<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream) {
    if (count <= 0) return 0;
    if (path != 0) {
        if (!inverse) return launch_compact(FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_forward>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
        return launch_compact(FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_inverse>, d_input, d_output, count, grid_cap, nreuses, balance, rotate, stream);
    }
    dim3 grid(grid_for(count, 4096 / SMFFT_N, grid_cap)), block(256);
    if (!inverse) FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_forward><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
    else          FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_inverse><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
    return (int)hipGetLastError();
}
#endif

}  // namespace smfft
