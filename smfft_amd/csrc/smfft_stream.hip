// smfft_stream.hip -- host-resident batches streamed through the GPU.
//
// The reference's L3 wrappers (GPU_smFFT_4elements, CT/FFT-GPU-32bit.cu:827-908) move the whole
// batch with one pageable cudaMemcpy each way around the timed launches; at 4 GiB per direction the
// copies cost hundreds of ms against 1.4 ms of transform.  smfft_host_transform() is the form a
// caller with data in HOST memory actually wants: the batch is cut into slabs and every slab goes
// H2D -> FFT -> D2H on the stream of one of several independent LANES.  A lane is a host thread with
// its own HIP stream, two device slab pairs and (for pageable memory) two pinned bounce-buffer pairs;
// lane j owns slabs j, j+L, j+2L, ...  Lanes need no synchronisation with each other, their DMA
// transfers in the two directions overlap on the two PCIe directions, and the lanes' host-side
// memcpys into / out of the bounce buffers run in parallel, which is what makes pageable memory as
// fast as pinned used to be.  Pinned memory (smfft_host_malloc, or anything hipHostRegister'ed) needs none of this: the
// transform's kernel reads and writes the caller's buffers directly over PCIe (zero copy; the slab pipeline without bounce
// buffers stays available with SMFFT_HOST_ZERO_COPY=0).
//
// The batch may be larger than device memory: the device only ever holds 2 slab pairs per lane.
#include <hip/hip_runtime.h>
#include "smfft_state.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/smfft.h"

namespace {

constexpr int kSlots = 2;   // slab pairs in flight per lane

struct Lane {
    hipStream_t stream = nullptr;
    char* d_in[kSlots] = {};
    char* d_out[kSlots] = {};
    char* b_in[kSlots] = {};    // pinned bounce buffers (pageable callers only)
    char* b_out[kSlots] = {};
    hipEvent_t done[kSlots] = {};
};

struct Pipe {
    int device = -1;
    size_t slab_bytes = 0;      // capacity of every buffer below
    bool bounce = false;
    std::vector<Lane> lanes;
};

std::mutex g_pipe_mutex;        // one host transform at a time per process (the pipe is cached)
Pipe g_pipe;

void destroy_pipe(Pipe& p) {
    for (Lane& l : p.lanes) {
        for (int k = 0; k < kSlots; ++k) {
            if (l.d_in[k]) (void)hipFree(l.d_in[k]);
            if (l.d_out[k]) (void)hipFree(l.d_out[k]);
            if (l.b_in[k]) (void)hipHostFree(l.b_in[k]);
            if (l.b_out[k]) (void)hipHostFree(l.b_out[k]);
            if (l.done[k]) (void)hipEventDestroy(l.done[k]);
        }
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    p = Pipe();
}

// (re)builds the cached pipe when the geometry grows; returns false when an allocation fails
bool ensure_pipe(int device, int nlanes, size_t slab_bytes, bool bounce) {
    Pipe& p = g_pipe;
    if (p.device == device && (int)p.lanes.size() == nlanes && p.slab_bytes >= slab_bytes && (p.bounce || !bounce)) return true;
    destroy_pipe(p);
    p.device = device;
    p.slab_bytes = slab_bytes;
    p.bounce = bounce;
    p.lanes.resize(nlanes);
    for (Lane& l : p.lanes) {
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) return false;
        for (int k = 0; k < kSlots; ++k) {
            if (hipMalloc(&l.d_in[k], slab_bytes) != hipSuccess) return false;
            if (hipMalloc(&l.d_out[k], slab_bytes) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&l.done[k], hipEventDisableTiming) != hipSuccess) return false;
            if (bounce) {
                if (hipHostMalloc(&l.b_in[k], slab_bytes, hipHostMallocDefault) != hipSuccess) return false;
                if (hipHostMalloc(&l.b_out[k], slab_bytes, hipHostMallocDefault) != hipSuccess) return false;
            }
        }
    }
    return true;
}

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // pageable memory is "invalid value" to the runtime: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

struct Job {
    smfft::LaunchState state;   // of the thread that called smfft_host_transform: its lanes launch with the same settings
    int family, FFT_size, inverse, reorder, device;
    const char* h_in;
    char* h_out;
    long long nFFTs, slab_ffts, nslabs;
    size_t fft_bytes;
    bool bounce;
    std::atomic<int> status{0};
};

void lane_main(Job* job, int lane_index, int nlanes) {
    if (hipSetDevice(job->device) != hipSuccess) { job->status = -2; return; }
    smfft::set_thread_state(job->state);
    Lane& l = g_pipe.lanes[lane_index];
    struct Pending { long long first = -1, count = 0; } pending[kSlots];
    auto drain = [&](int k) {   // slab previously issued on slot k: wait for its D2H, hand the result to the caller
        if (pending[k].first < 0) return;
        if (hipEventSynchronize(l.done[k]) != hipSuccess) job->status = -3;
        if (job->bounce) memcpy(job->h_out + pending[k].first * job->fft_bytes, l.b_out[k], pending[k].count * job->fft_bytes);
        pending[k].first = -1;
    };
    long long local = 0;
    for (long long s = lane_index; s < job->nslabs && job->status == 0; s += nlanes, ++local) {
        const int k = (int)(local % kSlots);
        drain(k);
        const long long first = s * job->slab_ffts;
        const long long count = std::min(job->slab_ffts, job->nFFTs - first);
        const size_t bytes = (size_t)count * job->fft_bytes;
        const char* src = job->h_in + first * job->fft_bytes;
        char* dst = job->h_out + first * job->fft_bytes;
        if (job->bounce) {
            memcpy(l.b_in[k], src, bytes);
            src = l.b_in[k];
            dst = l.b_out[k];
        }
        int rc = (int)hipMemcpyAsync(l.d_in[k], src, bytes, hipMemcpyHostToDevice, l.stream);
        if (rc == 0) rc = smfft_launch(job->family, 0, l.d_in[k], l.d_out[k], job->FFT_size, (int)count, job->inverse, job->reorder, l.stream);
        if (rc == 0) rc = (int)hipMemcpyAsync(dst, l.d_out[k], bytes, hipMemcpyDeviceToHost, l.stream);
        if (rc == 0) rc = (int)hipEventRecord(l.done[k], l.stream);
        if (rc != 0) { job->status = rc; break; }
        pending[k].first = first;
        pending[k].count = count;
    }
    // oldest first
    for (int i = 0; i < kSlots; ++i) drain((int)((local + i) % kSlots));
}

}  // namespace

extern "C" {

void* smfft_host_malloc(unsigned long long bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
int smfft_host_free(void* h_ptr) { return h_ptr ? (int)hipHostFree(h_ptr) : 0; }

void smfft_host_pipeline_release(void) {
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    destroy_pipe(g_pipe);
}

int smfft_host_transform(int family, const void* h_input, void* h_output, int FFT_size, long long nFFTs, int inverse, int reorder,
                         long long slab_ffts, int lanes, double* elapsed_ms) {
    const auto t0 = std::chrono::steady_clock::now();
    if (family < 0 || family > 2 || FFT_size < 32 || FFT_size > 4096 || (FFT_size & (FFT_size - 1))) {
        printf("Error wrong FFT length!\n");
        return -1;
    }
    if (family == 2 && FFT_size < 512) { printf("Error wrong FFT length!\n"); return -1; }
    if (nFFTs <= 0) { if (elapsed_ms) *elapsed_ms = 0; return 0; }
    smfft_init();
    Job job;
    job.family = family; job.FFT_size = FFT_size; job.inverse = inverse; job.reorder = reorder;
    job.h_in = (const char*)h_input; job.h_out = (char*)h_output; job.nFFTs = nFFTs;
    job.fft_bytes = (size_t)FFT_size * (family == 2 ? 4 : 8);
    if (hipGetDevice(&job.device) != hipSuccess) return -2;
    job.state = smfft::get_thread_state();
    // defaults: 32 MiB slabs (1024 workgroup tiles: one full wave of the persistent grid), 8 lanes
    const char* e = getenv("SMFFT_HOST_SLAB_MIB");
    const long long slab_default = std::max<long long>(1, ((e ? atoll(e) : 32) << 20) / (long long)job.fft_bytes);
    job.slab_ffts = slab_ffts > 0 ? slab_ffts : slab_default;
    job.slab_ffts = std::min<long long>(job.slab_ffts, std::min<long long>(nFFTs, (1LL << 31) / FFT_size));   // int count per launch
    job.nslabs = (nFFTs + job.slab_ffts - 1) / job.slab_ffts;
    job.bounce = !(is_pinned(h_input) && is_pinned(h_output));
    // Both buffers pinned: no copies at all.  The external kernel reads the host buffer and writes the host buffer
    // directly over PCIe -- loads and stores of thousands of waves in flight use the link in both directions at once:
    // 97-98 GB/s in + out for the config-2 batch (49 GB/s each way; H2D or D2H alone 56-57) against 68-75 GB/s for the slab
    // pipeline below, whose DMA copies do not overlap that well (round 2's tools/host_stream_probe.py (git history), profiles/r02_host_stream.txt).
    // SMFFT_HOST_ZERO_COPY=0 keeps the slab pipeline.  (For pageable memory the same kernel over the lanes' pinned bounce
    // buffers measured no better than the DMA pipeline -- 62-75 against 69 GB/s: the host-side memcpys bound it.)
    e = getenv("SMFFT_HOST_ZERO_COPY");
    if (!job.bounce && !(e && atoi(e) == 0)) {
        void *d_in = nullptr, *d_out = nullptr;
        if (hipHostGetDevicePointer(&d_in, const_cast<void*>(h_input), 0) == hipSuccess && hipHostGetDevicePointer(&d_out, h_output, 0) == hipSuccess) {
            const long long per_launch = std::max<long long>(1, ((1LL << 31) / FFT_size - 1) / 4096 * 4096);   // int count per launch, whole tiles
            int rc = 0;
            for (long long first = 0; first < nFFTs && rc == 0; first += per_launch) {
                const long long count = std::min(per_launch, nFFTs - first);
                rc = smfft_launch(family, 0, (const char*)d_in + (size_t)first * job.fft_bytes, (char*)d_out + (size_t)first * job.fft_bytes, FFT_size, (int)count, inverse, reorder, nullptr);
            }
            if (rc == 0) rc = (int)hipStreamSynchronize(nullptr);
            if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return rc == 0 ? 0 : -5;
        }
        (void)hipGetLastError();
    }
    // measured on the config-2 batch (round 2's tools/host_stream_probe.py (git history)): pageable memory needs the 8 lanes' parallel
    // memcpys (1 lane 328 ms, 2: 175, 4: 121, 8: 118); pinned memory is best with 2 (120 ms; 8: 130, 16: 289)
    e = getenv("SMFFT_HOST_LANES");
    int nlanes = lanes > 0 ? lanes : (e ? atoi(e) : (job.bounce ? 8 : 2));
    nlanes = (int)std::max<long long>(1, std::min<long long>(std::min(nlanes, 32), job.nslabs));

    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    if (!ensure_pipe(job.device, nlanes, (size_t)job.slab_ffts * job.fft_bytes, job.bounce)) {
        (void)hipGetLastError();
        destroy_pipe(g_pipe);
        return -4;
    }
    std::vector<std::thread> threads;
    for (int j = 1; j < nlanes; ++j) threads.emplace_back(lane_main, &job, j, nlanes);
    lane_main(&job, 0, nlanes);
    for (auto& t : threads) t.join();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return job.status;
}

}  // extern "C"
