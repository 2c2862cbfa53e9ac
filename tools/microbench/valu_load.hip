// valu_load.hip -- how much of the copy ceiling does VALU / LDS work between a wave's loads and stores cost?
// The external kernels' access shape (256-thread workgroups, one 8 KiB row per wave, 16 x 8 B/lane non-temporal loads,
// 16 stores, 34816 B of LDS per workgroup) with F dependent FMAs per loaded float in between, and optionally one LDS
// round trip, at 4 and at 3 waves per SIMD.  Buffers: a smfft_malloc_pair pair and two plain hipMalloc blocks.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_load.hip -o valu_load -L../../smfft_amd -lsmfft_amd -Wl,-rpath,'$ORIGIN/../../smfft_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
extern "C" int smfft_malloc_pair(unsigned long long bytes, void** a, void** b);
extern "C" int smfft_free_pair(void* a);

typedef float v2f __attribute__((ext_vector_type(2)));

template <int F, int LDSRT, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
copy_with_work(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles, float k) {
    __shared__ v2f s[4352];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v2f* g = in + tile * 4096 + wave * 1024 + lane;
        v2f* o = out + tile * 4096 + wave * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                r[c].x = __builtin_fmaf(r[c].x, k, r[(c + 1) & 15].y);
                r[c].y = __builtin_fmaf(r[c].y, k, r[(c + 5) & 15].x);
            }
        if (LDSRT) {
            v2f* q = s + wave * 1088 + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) q[64 * c] = r[c];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = q[64 * (15 - c) + ((63 - lane) - lane)];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

static hipEvent_t e0, e1;
template <int F, int LDSRT, int WAVES>
static float run(const void* a, void* b, long ntiles, int reps) {
    auto launch = [&] { copy_with_work<F, LDSRT, WAVES><<<12288, 256>>>((const v2f*)a, (v2f*)b, ntiles, 0.999f); };
    launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int WAVES>
static void sweep(const char* what, const void* a, void* b, long ntiles, double bytes) {
    const int reps = 20;
    float t[8];
    t[0] = run<0, 0, WAVES>(a, b, ntiles, reps);
    t[1] = run<4, 0, WAVES>(a, b, ntiles, reps);
    t[2] = run<8, 0, WAVES>(a, b, ntiles, reps);
    t[3] = run<16, 0, WAVES>(a, b, ntiles, reps);
    t[4] = run<24, 0, WAVES>(a, b, ntiles, reps);
    t[5] = run<0, 1, WAVES>(a, b, ntiles, reps);
    t[6] = run<16, 1, WAVES>(a, b, ntiles, reps);
    t[7] = run<0, 0, WAVES>(a, b, ntiles, reps);
    const char* names[8] = {"F=0", "F=4 (128 VALU/wave-row)", "F=8 (256)", "F=16 (512)", "F=24 (768)", "F=0 + LDS round trip", "F=16 + LDS round trip", "F=0 again"};
    for (int i = 0; i < 8; ++i)
        printf("%-6s waves/SIMD %d  %-26s %.4f ms  %.3f TB/s  frac %.3f  vs F=0 %+.1f %%\n", what, WAVES, names[i], t[i], 2 * bytes / t[i] * 1e-9, 2 * bytes / t[i] * 1e-9 / 8.0, (t[i] / t[0] - 1) * 100);
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 4096) * (1ull << 20);
    const long ntiles = bytes / (4096 * 8);
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    void *pa, *pb, *qa, *qb;
    if (smfft_malloc_pair(bytes, &pa, &pb)) { printf("smfft_malloc_pair failed\n"); return 1; }
    CK(hipMalloc(&qa, bytes));
    CK(hipMalloc(&qb, bytes));
    CK(hipMemset(pa, 0, bytes));
    CK(hipMemset(qa, 0, bytes));
    for (int round = 0; round < 2; ++round) {
        sweep<4>("pair", pa, pb, ntiles, (double)bytes);
        sweep<3>("pair", pa, pb, ntiles, (double)bytes);
        sweep<4>("plain", qa, qb, ntiles, (double)bytes);
        sweep<3>("plain", qa, qb, ntiles, (double)bytes);
    }
    smfft_free_pair(pa);
    return 0;
}
