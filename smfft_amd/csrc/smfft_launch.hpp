// smfft_launch.hpp -- host-side launch entry points, one translation unit per transform length so
// the 80-odd kernel instantiations compile in parallel (and do not share a register-allocation
// context more than necessary).
#pragma once
#include <hip/hip_runtime.h>

namespace smfft {

// path: 0 = external (count = number of FFTs), 1 = multiple (count = number of FFT slots, each
// transformed nreuses times in LDS; the benchmark entry points pass NREUSES = 100), 2 = multiple without cross-application
// fusion (natural-order variants of the planar lengths; everything else runs path 1).  grid_cap <= 0: one workgroup per 4096-element tile.
// balance != 0 (multiple paths): when the batch is more chains than fit on the chip at once, the launch is a persistent grid of the
// co-resident workgroups with the applications spread evenly over them (smfft_kernels.hpp, MultipleSchedule); 0: one chain per workgroup, grid-strided.
// rotate = k > 0 (multiple paths): the waves' scheduling priority rotates every 2^k shader clocks (WavePriority); 0: the arbiter's oldest-first order.
// pace = K > 0: the external kernels run their rate limiter with K serialised loads (smfft_kernels.hpp, vmem_throttle); the host API decides it per launch.
// Returns hipSuccess (0) or the launch error.
template <int N>
int launch_ct(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream);
// Stockham C2C program (inverse sign), N = 256..4096.
template <int N>
int launch_st(const float2* d_input, float2* d_output, int count, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream);
// R2C (inverse = 0) / C2R (inverse = 1) of real length 2L, L = 256..2048.
template <int L>
int launch_rc(const float2* d_input, float2* d_output, int count, int inverse, int path, int grid_cap, int nreuses, int pace, int balance, int rotate, hipStream_t stream);

// calibration copy of n_float2 elements (multiple of 4096) with the external kernels' access shape
int launch_stream_copy(const float2* d_input, float2* d_output, long n_float2, int grid_cap, int pace, hipStream_t stream);
// the same access shape, writes only / reads only
int launch_stream_write(float2* d_output, long n_float2, int grid_cap, hipStream_t stream);
int launch_stream_read(const float2* d_input, long n_float2, int grid_cap, hipStream_t stream);

// ---- the balanced schedule of the multiple paths (host side: smfft_api.hip) ----------------------------------------------
// workgroups of `kernel` (block of `threads`, static LDS only) that are co-resident on the current device; 0 if unknown
int resident_workgroups(const void* kernel, int threads);
// device counters for the launches of the calling thread, or nullptr (the normal case): see smfft_measure_multiple_residency
unsigned* residency_probe();
void note_resident_workgroups(int slots);     // what the last launch_compact of this thread assumed (for the probe's caller)
int last_noted_slots();
// a zero-initialised flag per chain for launches on `stream` of the current device, and the launch's own epoch (flags hold the
// epoch of the launch that set them, so nothing is reset between launches); nullptr: no memory -- launch unbalanced
unsigned* schedule_flags(int nchains, hipStream_t stream, unsigned* epoch);

inline int grid_for(int count, int ffts_per_block, int grid_cap) {
    int ntiles = (count + ffts_per_block - 1) / ffts_per_block;
    if (grid_cap > 0 && ntiles > grid_cap) return grid_cap;
    return ntiles;
}

}  // namespace smfft
