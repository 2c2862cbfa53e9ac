// phase_study.hip -- chip-wide read / write PHASES for a copy between two ordinary buffers, gated by the shared
// real-time counter (s_memrealtime, 100 MHz, the same value on every CU): loads may only be issued while bit k of the
// counter is set, stores while it is clear.  Ordinary memory serves pure reads at 7.2 TB/s and pure writes at 5.6 TB/s
// (serial sum for 4 GiB + 4 GiB: 1.365 ms) but a mixed stream in 1.48-1.60 ms; do coarse phases get closer to the sum?
// Buffers: A, B allocated back to back (normally the same memory class), C after a 96 GiB filler (normally another).
//
// Build: hipcc -O3 --offload-arch=gfx950 phase_study.hip -o phase_study
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wait_phase(int k, unsigned want) {
    while (((unsigned)(__builtin_amdgcn_s_memrealtime() >> k) & 1u) != want) __builtin_amdgcn_s_sleep(1);
}

// MODE 0: plain; 1: K serialised LDS loads between loads and stores; 2: phases (loads in odd windows, stores in even ones);
// 3: phases in two groups (workgroups with odd index use the opposite windows for their LOADS only: reads stay together?  no:
//    group g loads in window g and stores in window 2 + g of a 4-window cycle: read, read, write, write)
template <int MODE>
__global__ void __launch_bounds__(256) copyk(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles, int k, float f) {
    __shared__ v2f s[4352];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v2f* g = in + tile * 4096 + wave * 1024 + lane;
        v2f* o = out + tile * 4096 + wave * 1024 + lane;
        v2f r[16];
        if (MODE == 2) wait_phase(k, 1u);
        if (MODE == 3) { const unsigned grp = blockIdx.x & 1; while ((((unsigned)(__builtin_amdgcn_s_memrealtime() >> k)) & 3u) != grp) __builtin_amdgcn_s_sleep(1); }
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(r[c]));
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                r[c].x = __builtin_fmaf(r[c].x, f, r[(c + 1) & 15].y);
                r[c].y = __builtin_fmaf(r[c].y, f, r[(c + 5) & 15].x);
            }
        if (MODE == 1) {
            for (int c = 0; c < k; ++c) {
                const v2f* q = s + wave * 1088 + lane + 64 * (c & 15);
                v2f d;
                asm volatile("flat_load_dwordx2 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(d) : "v"(q) : "memory");
            }
        }
        if (MODE == 2) wait_phase(k, 0u);
        if (MODE == 3) { const unsigned grp = 2u + (blockIdx.x & 1); while ((((unsigned)(__builtin_amdgcn_s_memrealtime() >> k)) & 3u) != grp) __builtin_amdgcn_s_sleep(1); }
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

static hipEvent_t e0, e1;
template <class F>
static float median_ms(F&& launch) {
    std::vector<float> t;
    for (int i = 0; i < 9; ++i) {
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
    const long ntiles = bytes / 32768;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    void *A, *B, *C, *filler;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&B, bytes));
    CK(hipMalloc(&filler, 96ull << 30));
    CK(hipMalloc(&C, bytes));
    CK(hipMemset(A, 0, bytes));
    CK(hipMemset(B, 0, bytes));
    CK(hipMemset(C, 0, bytes));
    const int G = 12288;
    for (int round = 0; round < 2; ++round)
        for (int target = 0; target < 2; ++target) {
            const v2f* in = (const v2f*)A;
            v2f* out = (v2f*)(target ? C : B);
            const char* name = target ? "A->C (far)" : "A->B (adjacent)";
            printf("%s  plain %.4f", name, median_ms([&] { copyk<0><<<G, 256>>>(in, out, ntiles, 0, 0.999f); }));
            for (int k : {8, 12, 16}) printf(" | K=%d %.4f", k, median_ms([&] { copyk<1><<<G, 256>>>(in, out, ntiles, k, 0.999f); }));
            printf("\n%s  phases(2 windows)", name);
            for (int k : {4, 5, 6, 7, 8, 9}) printf(" | %d ns %.4f", 10 << k, median_ms([&] { copyk<2><<<G, 256>>>(in, out, ntiles, k, 0.999f); }));
            printf("\n%s  phases(4 windows, 2 groups)", name);
            for (int k : {4, 5, 6, 7, 8, 9}) printf(" | %d ns %.4f", 10 << k, median_ms([&] { copyk<3><<<G, 256>>>(in, out, ntiles, k, 0.999f); }));
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
