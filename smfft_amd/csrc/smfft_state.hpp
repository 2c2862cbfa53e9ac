// smfft_state.hpp -- the per-host-thread launch state of libsmfft_amd.so (smfft_api.hip) and the hook through which the
// lanes of smfft_host_transform (smfft_stream.hip) inherit it from the thread that called them.
#pragma once
#include <climits>

namespace smfft {

constexpr int kUnsetGridCap = INT_MIN;
struct LaunchState {
    int device;      // -1: the process default (SMFFT_DEVICE, else 0)
    int grid_cap;    // kUnsetGridCap: the process default (SMFFT_GRID_CAP, else 12288); <= 0: one workgroup per tile
    int nreuses;     // 0: NREUSES = 100
    int pacing;      // -2: the process default; -1: chosen per launch from the output buffer; K >= 0: K loads
    int balance;     // -1: the process default (SMFFT_MULT_BALANCE, else 1); 0 / 1: the multiple paths' balanced schedule off / on
    int rotate;      // -1: the process default (SMFFT_PRIO_ROTATE, else 15); 0: the arbiter's oldest-first order; k: priorities rotate every 2^k clocks
    int handoff_wait_us;   // -1: the process default (SMFFT_HANDOFF_WAIT_US, else 1000): balanced schedule, how long a resumer waits before it takes a chain over
    int delay_chain, delay_ms, delay_after_commit;   // fault injection (smfft_debug_delay_parking); delay_chain < 0: none
};
LaunchState get_thread_state();
void set_thread_state(const LaunchState& s);

}  // namespace smfft
