#include <cstdlib>
#include <mutex>
// smfft_inst.hip -- instantiates every kernel of ONE transform length; compiled once per length
// with -DSMFFT_N=<32..4096> (see Makefile).
#include "smfft_kernels.hpp"
#include "smfft_launch.hpp"

#ifndef SMFFT_N
#error "compile with -DSMFFT_N=<transform length>"
#endif

#define SMFFT_PASTE3_(a, b, c) a##b##c
#define SMFFT_PASTE3(a, b, c) SMFFT_PASTE3_(a, b, c)
#define CT_CLASS(suffix) SMFFT_PASTE3(FFT_, SMFFT_N, suffix)

namespace smfft {

template <>
int launch_ct<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, int grid_cap, int nreuses, int pace, hipStream_t stream) {
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
    } else {
        // in-LDS path: compact workgroups (one wave per 1024 elements for N <= 1024, one FFT per workgroup above)
        grid = dim3(grid_for(count, Geometry<SMFFT_N>::kCompactFfts, grid_cap));
        block = dim3(Geometry<SMFFT_N>::kCompactThreads);
        if (!inverse && reorder)  SMFFT_DIT_multiple<CT_CLASS(_forward)><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
        if (!inverse && !reorder) SMFFT_DIT_multiple<CT_CLASS(_forward_noreorder)><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
        if (inverse && reorder)   SMFFT_DIT_multiple<CT_CLASS(_inverse)><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
        if (inverse && !reorder)  SMFFT_DIT_multiple<CT_CLASS(_inverse_noreorder)><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
    }
    return (int)hipGetLastError();
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
int launch_st<SMFFT_N>(const float2* d_input, float2* d_output, int count, int path, int grid_cap, int nreuses, int pace, hipStream_t stream) {
    if (count <= 0) return 0;
    dim3 grid(grid_for(count, 4096 / SMFFT_N, grid_cap)), block(256);
#if SMFFT_N == 4096
    // same transform (Engine<4096, inverse, reorder>) through the occupancy-3 build, see SMFFT_DIT_external_occ3
    if (path == 0) SMFFT_DIT_external_occ3<CT_CLASS(_inverse)><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
#else
    if (path == 0) FFT_GPU_external<ST_CLASS><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
#endif
    else           FFT_GPU_multiple<ST_CLASS><<<dim3(grid_for(count, Geometry<SMFFT_N>::kCompactFfts, grid_cap)), dim3(Geometry<SMFFT_N>::kCompactThreads), 0, stream>>>(d_input, d_output, count, nreuses);
    return (int)hipGetLastError();
}

#if SMFFT_N >= 256 && SMFFT_N <= 2048
template <>
int launch_rc<SMFFT_N>(const float2* d_input, float2* d_output, int count, int inverse, int path, int grid_cap, int nreuses, int pace, hipStream_t stream) {
    if (count <= 0) return 0;
    dim3 grid(grid_for(count, 4096 / SMFFT_N, grid_cap)), block(256);
    if (path == 0) {
        if (!inverse) FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_forward><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
        else          FFT_GPU_R2C_C2R_external<ST_CLASS, FFT_inverse><<<grid, block, 0, stream>>>(d_input, d_output, count, pace);
    } else {
        grid = dim3(grid_for(count, Geometry<SMFFT_N>::kCompactFfts, grid_cap));
        block = dim3(Geometry<SMFFT_N>::kCompactThreads);
        if (!inverse) FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_forward><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
        else          FFT_GPU_R2C_C2R_multiple<ST_CLASS, FFT_inverse><<<grid, block, 0, stream>>>(d_input, d_output, count, nreuses);
    }
    return (int)hipGetLastError();
}
#endif

}  // namespace smfft
