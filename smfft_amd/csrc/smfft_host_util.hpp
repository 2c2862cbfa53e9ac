// smfft_host_util.hpp -- host-side helpers of libsmfft_amd.so and libsmfft_vendor.so, in one place:
//   * the compile-time switches of the L3 wrappers, under the macro names the reference's debug.h files
//     use (DEBUG, TESTING, CUFFT, EXTERNAL, MULTIPLE; SMFFT_*/debug.h);
//   * checkHipErrors(call): a failing HIP call prints where it happened and ends the process, the error
//     convention of the reference's checkCudaErrors (SMFFT_CooleyTukey_C2C/utils_cuda.h:12-22);
//   * GpuTimer: the event stopwatch the launch API times its single launch with (interface of the reference's
//     GpuTimer, SMFFT_CooleyTukey_C2C/timer.h:6-40: Start, Stop, Elapsed in milliseconds).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>

// ---- switches ------------------------------------------------------------------------------------
#ifndef DEBUG
#define DEBUG false      // progress lines of the L3 wrappers
#endif
#define TESTING          // Stockham program: compare with the vendor library
#define CUFFT true       // run the vendor comparator (hipFFT here)
#define EXTERNAL true    // run the global -> FFT -> global benchmark
#define MULTIPLE true    // run the NREUSES-FFTs-per-load benchmark

// ---- error convention ----------------------------------------------------------------------------
namespace smfft_host {
inline void require_success(hipError_t status, const char* expression, const char* file, int line) {
    if (status == hipSuccess) return;
    fprintf(stderr, "HIP error at: %s:%d\n%s %s\n", file, line, hipGetErrorString(status), expression);
    exit(1);
}

// ---- stopwatch -----------------------------------------------------------------------------------
// Two events on one stream; the events are made on first use so that a timer can be declared before the
// device is selected.
class EventStopwatch {
public:
    explicit EventStopwatch(hipStream_t stream = nullptr) : stream_(stream) {}
    EventStopwatch(const EventStopwatch&) = delete;
    EventStopwatch& operator=(const EventStopwatch&) = delete;
    ~EventStopwatch() {
        for (hipEvent_t e : marks_)
            if (e) (void)hipEventDestroy(e);
    }
    void Start() { mark(0); }
    void Stop() { mark(1); }
    // waits for the stop mark; milliseconds between the two marks
    float Elapsed() {
        float ms = 0.f;
        if (marks_[0] && marks_[1] && hipEventSynchronize(marks_[1]) == hipSuccess) (void)hipEventElapsedTime(&ms, marks_[0], marks_[1]);
        return ms;
    }

private:
    void mark(int which) {
        if (!marks_[which] && hipEventCreate(&marks_[which]) != hipSuccess) { marks_[which] = nullptr; return; }
        (void)hipEventRecord(marks_[which], stream_);
    }
    hipStream_t stream_;
    hipEvent_t marks_[2] = {nullptr, nullptr};
};
}  // namespace smfft_host

#define checkHipErrors(call) smfft_host::require_success((call), #call, __FILE__, __LINE__)
using GpuTimer = smfft_host::EventStopwatch;
