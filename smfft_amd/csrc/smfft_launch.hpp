// smfft_launch.hpp -- host-side launch entry points, one translation unit per transform length so
// the 80-odd kernel instantiations compile in parallel (and do not share a register-allocation
// context more than necessary).
#pragma once
#include <hip/hip_runtime.h>

namespace smfft {

// What a launch takes from the host API's per-thread state (smfft_api.hip).
//   grid_cap <= 0: one workgroup per 4096-element tile.
//   nreuses: applications per chain of the multiple paths (the benchmark entry points pass NREUSES = 100).
//   pace = K > 0: the external kernels run their rate limiter with K serialised loads (smfft_kernels.hpp, vmem_throttle).
//   balance != 0 (multiple paths): when the batch is more chains than fit on the chip at once, the launch is a persistent grid of the
//     co-resident workgroups with the applications spread evenly over them (smfft_kernels.hpp, MultipleSchedule); 0: one chain per
//     workgroup, grid-strided; n >= 2 (tests): balanced over n workgroups.
//   rotate = k > 0 (multiple paths): the waves' scheduling priority rotates every 2^k shader clocks (WavePriority); 0: the arbiter's own order.
//   handoff_wait_us: balanced schedule -- how long the workgroup that resumes a cut chain waits for data nobody has committed to
//     parking before it takes the whole chain over (MultipleSchedule).
//   delay_*: fault injection for the tests of that path (smfft_debug_delay_parking).
struct LaunchOptions {
    int grid_cap, nreuses, pace, balance, rotate;
    unsigned handoff_wait_us;
    int delay_chain;
    unsigned delay_ms;
    int delay_after_commit;
};

// path: 0 = external (count = number of FFTs), 1 = multiple (count = number of FFT slots, each transformed nreuses times in LDS),
// 2 = multiple without cross-application fusion (natural-order variants of the planar lengths; everything else runs path 1).
// Returns hipSuccess (0) or the launch error.
template <int N>
int launch_ct(const float2* d_input, float2* d_output, int count, int inverse, int reorder, int path, const LaunchOptions& opt, hipStream_t stream);
// Stockham C2C program (inverse sign), N = 32..4096.
template <int N>
int launch_st(const float2* d_input, float2* d_output, int count, int path, const LaunchOptions& opt, hipStream_t stream);
// R2C (inverse = 0) / C2R (inverse = 1) of real length 2L, L = 256..2048.
template <int L>
int launch_rc(const float2* d_input, float2* d_output, int count, int inverse, int path, const LaunchOptions& opt, hipStream_t stream);

// calibration copy of n_float2 elements (multiple of 4096) with the external kernels' access shape
int launch_stream_copy(const float2* d_input, float2* d_output, long n_float2, int grid_cap, int pace, hipStream_t stream);
// the same access shape, writes only / reads only
int launch_stream_write(float2* d_output, long n_float2, int grid_cap, hipStream_t stream);
int launch_stream_read(const float2* d_input, long n_float2, int grid_cap, hipStream_t stream);

// ---- the balanced schedule of the multiple paths (host side: smfft_api.hip) ----------------------------------------------
// workgroups of `kernel` (block of `threads`, static LDS only) that are co-resident on the current device; 0 if unknown
// (a device this library has no register-file figures for: the multiple paths then run one chain per workgroup)
int resident_workgroups(const void* kernel, int threads);
// device counters for the launches of the calling thread, or nullptr (the normal case): see smfft_measure_multiple_residency
unsigned* residency_probe();
void note_resident_workgroups(int slots);     // what the last launch_compact of this thread assumed (for the probe's caller)
int last_noted_slots();
// The hand-over words of ONE balanced launch: a buffer of at least `nchains` words that no other launch in flight uses, and the
// launch's `base` (4 * its epoch in that buffer: words hold the state of the launch that wrote them, so nothing is reset between
// launches).  nullptr: none to be had (no memory, or too many balanced launches in flight) -- launch unbalanced.  After the
// kernel has been enqueued the launcher calls schedule_release(ticket, stream): the buffer goes back to the pool when the stream
// has passed that point (an event; nothing in the launch path frees memory or waits for the device).
unsigned* schedule_acquire(int nchains, hipStream_t stream, unsigned* base, int* ticket);
void schedule_release(int ticket, hipStream_t stream);
constexpr int kScheduleMaxChains = 65536;      // words per buffer (256 KiB): the most chains a launch may have to be balanced

inline int grid_for(int count, int ffts_per_block, int grid_cap) {
    int ntiles = (count + ffts_per_block - 1) / ffts_per_block;
    if (grid_cap > 0 && ntiles > grid_cap) return grid_cap;
    return ntiles;
}

}  // namespace smfft
