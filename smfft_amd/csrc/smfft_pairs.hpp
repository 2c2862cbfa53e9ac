// smfft_pairs.hpp -- the paired-buffer allocator of libsmfft_amd.so (smfft_pairs.hip), as the host API (smfft_api.hip) sees it.
// FROZEN since round 3: the policy, thresholds and budgets are those measured in rounds 2 and 3; round 4
// only moved it into a translation unit of its own and split the scan into its four steps.  The reference has no counterpart:
// its wrapper takes two plain cudaMalloc blocks (CT:850-853), which SMFFT_WRAPPER_PLACEMENT=0 / SMFFT_PAIR_POLICY=plain restore.
#pragma once
#include <cstddef>

#include "../../include/smfft.h"

namespace smfft {
namespace pairs {

// K serialised loads the external kernels pace their stores with for this output buffer: `forced` >= 0 wins (smfft_set_pacing /
// SMFFT_PACING); otherwise k_mixed when d_output lies inside an output this allocator built from mixed / interleaved memory (or
// one that took the timed copy like such an output), k_ordinary for everything else.  Lock-free (sorted immutable snapshot).
int pacing_for(const void* d_output, int k_ordinary, int k_mixed, int forced);

// with_input = false (smfft_malloc_written[_for]): only the written buffer, the record kept under ITS address; caller_input (then
// only): the caller's own input of at least `bytes`, read -- never written -- by the timed copies that judge the candidates.
// budget_frac / budget_ms < 0: SMFFT_PAIR_BUDGET_FRAC / _MS, else a quarter of the free memory / 2 s.  0 = ok, 1 = out of memory.
int alloc_pair(size_t bytes, void** d_a, void** d_b, bool allow_search, double budget_frac = -1.0, double budget_ms = -1.0, bool with_input = true,
               const void* caller_input = nullptr, bool for_wrapper = false);
// the pair the L3 wrappers take: searched unless SMFFT_WRAPPER_PLACEMENT=0, kept (per device) for the next wrapper call of the same size
int alloc_pair_for_wrapper(size_t bytes, void** d_a, void** d_b);
int free_pair(void* d_a);
// gives back what the wrappers' cache holds: on `device` only (>= 0) or on every device (-1)
int release_pair_cache(int device = -1);
// device memory the cache holds on `device` right now (counts as available to the next wrapper call: it is either re-used or released)
size_t cached_bytes(int device);
// true: the next alloc_pair_for_wrapper(bytes) on `device` is served from the cache (no allocation at all)
bool cache_would_serve(size_t bytes, int device);
// The most device memory alloc_pair_for_wrapper(bytes) holds at any moment, given free_mem bytes free before the call: the two
// buffers plus -- while the search runs -- the scanned chunks, which are bounded by the byte budget and by what is free after the
// pair less 1 GiB of head room (the search is skipped when that is not even one chunk more than the output).
size_t wrapper_peak_bytes(size_t bytes, size_t free_mem);
void last_pair_info(SmfftPairInfo* out);
int va_window(unsigned long long* first, unsigned long long* next);

}  // namespace pairs
}  // namespace smfft
