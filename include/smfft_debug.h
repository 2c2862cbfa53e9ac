/* smfft_debug.h -- test-only entry points of libsmfft_amd.so.  Not part of the drop-in boundary (include/smfft.h): nothing a caller of the
 * reference's API needs, and their settings persist in the calling host thread's launch state until they are switched off again. */
#ifndef SMFFT_DEBUG_H_
#define SMFFT_DEBUG_H_
#ifdef __cplusplus
extern "C" {
#endif

/* Fault injection for the tests of the balanced schedule's hand-over (THIS host thread; smfft.h, smfft_set_handoff_wait_us): in the next
   balanced launches the workgroup that parks chain `chain` sleeps `milliseconds` before it commits to parking (after_commit = 0: the
   resumer takes the chain over) or between its commit and the parked word (1: the resumer waits for the store it has been promised).
   milliseconds <= 0: off.  Switch it off again: smfft_host_transform's worker lanes inherit the calling thread's launch state. */
void smfft_debug_delay_parking(int chain, int milliseconds, int after_commit);

#ifdef __cplusplus
}
#endif
#endif /* SMFFT_DEBUG_H_ */
