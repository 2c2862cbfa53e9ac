// timer.h -- hipEvent stopwatch with the interface of the reference's GpuTimer
// (SMFFT_CooleyTukey_C2C/timer.h:6-40): Start()/Stop() record on a stream, Elapsed() waits for the
// stop event and returns milliseconds.
#ifndef SMFFT_GPU_TIMER_H__
#define SMFFT_GPU_TIMER_H__
#include <hip/hip_runtime_api.h>

struct GpuTimer {
	hipEvent_t start;
	hipEvent_t stop;
	hipStream_t stream;

	explicit GpuTimer(hipStream_t s = 0) : stream(s) {
		(void)hipEventCreate(&start);
		(void)hipEventCreate(&stop);
	}
	~GpuTimer() {
		(void)hipEventDestroy(start);
		(void)hipEventDestroy(stop);
	}
	void Start() { (void)hipEventRecord(start, stream); }
	void Stop() { (void)hipEventRecord(stop, stream); }
	float Elapsed() {
		float elapsed = 0.f;
		(void)hipEventSynchronize(stop);
		(void)hipEventElapsedTime(&elapsed, start, stop);
		return elapsed;
	}
};
#endif
