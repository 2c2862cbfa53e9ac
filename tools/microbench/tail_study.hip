// tail_study.hip -- the fixed cost of one launch of an HBM-bound tile kernel (ramp + tail) and what tile scheduling does to it.
// t(size) = c + k * size, measured at 1, 2 and 4 GiB per direction on a smfft_malloc_pair pair, for
//   static  G : G workgroups, grid-stride over 32 KiB tiles (the external kernels' shape; G = 12288 is the library default)
//   dynamic G : G persistent workgroups that take tiles from an atomic counter (next index fetched while the current tile is
//               in flight); the last workgroup to finish resets the counters
// Every workgroup does the work of an N = 1024 FFT tile's worth of VALU (16 FMAs per float) and one LDS round trip.
//
// Build: hipcc -O3 --offload-arch=gfx950 tail_study.hip -o tail_study -L../../smfft_amd -lsmfft_amd -Wl,-rpath,'$ORIGIN/../../smfft_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
extern "C" int smfft_malloc_pair(unsigned long long bytes, void** a, void** b);
extern "C" int smfft_free_pair(void* a);
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_work(const v2f* __restrict__ in, v2f* __restrict__ out, long tile, v2f* s, float k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v2f* g = in + tile * 4096 + wave * 1024 + lane;
    v2f* o = out + tile * 4096 + wave * 1024 + lane;
    v2f r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            r[c].x = __builtin_fmaf(r[c].x, k, r[(c + 1) & 15].y);
            r[c].y = __builtin_fmaf(r[c].y, k, r[(c + 5) & 15].x);
        }
    v2f* q = s + wave * 1088 + lane;
#pragma unroll
    for (int c = 0; c < 16; ++c) q[64 * c] = r[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = q[64 * (15 - c) + ((63 - lane) - lane)];
#pragma unroll
    for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
}

__global__ void __launch_bounds__(256) copy_static(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles, float k) {
    __shared__ v2f s[4352];
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) tile_work(in, out, tile, s, k);
}

// single-wave workgroups: every wave is its own workgroup and strides over 8 KiB rows
__global__ void __launch_bounds__(64) copy_static_wave(const v2f* __restrict__ in, v2f* __restrict__ out, long nrows, float k) {
    __shared__ v2f s[1088];
    const int lane = threadIdx.x;
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const v2f* g = in + row * 1024 + lane;
        v2f* o = out + row * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                r[c].x = __builtin_fmaf(r[c].x, k, r[(c + 1) & 15].y);
                r[c].y = __builtin_fmaf(r[c].y, k, r[(c + 5) & 15].x);
            }
        v2f* q = s + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) q[64 * c] = r[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = q[64 * (15 - c) + ((63 - lane) - lane)];
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

// counters[0] = next tile, counters[1] = workgroups done
__global__ void __launch_bounds__(256) copy_dynamic(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles, float k, unsigned* counters) {
    __shared__ v2f s[4352];
    __shared__ unsigned next_tile[2];
    // the first gridDim.x tiles are taken by index; later ones from the counter, fetched one tile ahead
    unsigned tile = blockIdx.x;
    int slot = 0;
    if (threadIdx.x == 0) next_tile[0] = gridDim.x + atomicAdd(&counters[0], 1u);
    while (tile < ntiles) {
        tile_work(in, out, tile, s, k);
        __syncthreads();
        tile = next_tile[slot];
        slot ^= 1;
        if (threadIdx.x == 0 && tile < ntiles) next_tile[slot] = gridDim.x + atomicAdd(&counters[0], 1u);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&counters[1], 1u) == gridDim.x - 1) { counters[0] = 0; counters[1] = 0; __threadfence(); }
    }
}

// wave-granular: every wave takes its own 8 KiB rows from the counter (no workgroup barrier at all)
__global__ void __launch_bounds__(256) copy_dynamic_wave(const v2f* __restrict__ in, v2f* __restrict__ out, long nrows, float k, unsigned* counters) {
    __shared__ v2f s[4352];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned nwaves = gridDim.x * 4;
    unsigned row = blockIdx.x * 4 + wave;
    unsigned nxt = 0;
    if (lane == 0) nxt = nwaves + atomicAdd(&counters[0], 1u);
    while (row < nrows) {
        const v2f* g = in + (long)row * 1024 + lane;
        v2f* o = out + (long)row * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                r[c].x = __builtin_fmaf(r[c].x, k, r[(c + 1) & 15].y);
                r[c].y = __builtin_fmaf(r[c].y, k, r[(c + 5) & 15].x);
            }
        v2f* q = s + wave * 1088 + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) q[64 * c] = r[c];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = q[64 * (15 - c) + ((63 - lane) - lane)];
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
        row = __builtin_amdgcn_readfirstlane(nxt);
        if (lane == 0 && row < nrows) nxt = nwaves + atomicAdd(&counters[0], 1u);
    }
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(&counters[1], 1u) == nwaves - 1) { counters[0] = 0; counters[1] = 0; __threadfence(); }
    }
}

static hipEvent_t e0, e1;
template <class F>
static float median_ms(F&& launch) {
    std::vector<float> t;
    for (int i = 0; i < 13; ++i) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t bytes = 4ull << 30;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    void *pa, *pb;
    unsigned* counters;
    if (smfft_malloc_pair(bytes, &pa, &pb)) { printf("smfft_malloc_pair failed\n"); return 1; }
    CK(hipMalloc(&counters, 256));
    CK(hipMemset(counters, 0, 256));
    CK(hipMemset(pa, 0, bytes));
    const v2f* in = (const v2f*)pa;
    v2f* out = (v2f*)pb;
    struct Cfg { const char* name; int kind; int grid; };
    const Cfg cfgs[] = {{"static 12288", 0, 12288}, {"static 4096", 0, 4096}, {"static 1024", 0, 1024}, {"static 24576", 0, 24576},
                        {"dynamic 1024", 1, 1024}, {"static-wave 24576", 3, 24576}, {"static-wave 49152", 3, 49152}, {"static-wave 98304", 3, 98304}, {"static-wave 196608", 3, 196608}};
    for (int round = 0; round < 2; ++round)
        for (const Cfg& c : cfgs) {
            float t[3];
            for (int s = 0; s < 3; ++s) {
                const long ntiles = (long)(1 << s) * (1 << 30) / 32768;
                t[s] = median_ms([&] {
                    if (c.kind == 0) copy_static<<<c.grid, 256>>>(in, out, ntiles, 0.999f);
                    else if (c.kind == 1) copy_dynamic<<<c.grid, 256>>>(in, out, ntiles, 0.999f, counters);
                    else if (c.kind == 2) copy_dynamic_wave<<<c.grid, 256>>>(in, out, ntiles * 4, 0.999f, counters);
                    else copy_static_wave<<<c.grid, 64>>>(in, out, ntiles * 4, 0.999f);
                });
            }
            const float k = (t[2] - t[1]) / 2.f, fixed = t[1] - 2.f * k;
            printf("%-18s 1 GiB %.4f  2 GiB %.4f  4 GiB %.4f ms | per GiB %.4f ms (%.3f of peak)  fixed %.1f us | 4 GiB frac %.3f  2 GiB frac %.3f\n", c.name, t[0], t[1], t[2], k,
                   2.147483648 / k / 8.0, fixed * 1e3, 8.589934592 / t[2] / 8.0, 4.294967296 / t[1] / 8.0);
        }
    unsigned h[2];
    CK(hipMemcpy(h, counters, 8, hipMemcpyDeviceToHost));
    printf("counters after the last launch: %u %u\n", h[0], h[1]);
    smfft_free_pair(pa);
    return 0;
}
