// clock_ratio.hip -- the EFFECTIVE shader clock under VALU-dense load, measured, not read from hwmon (which shows the PLL
// target: 2.39 GHz whatever runs, profiles/r02_clocks.txt).  Every wave executes the same issue-bound instruction stream:
// REPS x 16 independent multiply-add chains (a wave64 VALU instruction issues in 2 cycles on a SIMD-32, so one wave keeps
// its SIMD busy).  One wave on an otherwise idle chip runs at the full clock; W waves on every SIMD of the chip take W times
// as many SIMD cycles: effective clock / full clock = W * t(1 wave) / t(full chip).
// Build: hipcc -O3 --offload-arch=gfx950 clock_ratio.hip -o clock_ratio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) chains(float* out, int reps, float k) {
    float a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (float)(threadIdx.x + c);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = __builtin_fmaf(a[c], k, 1.0f);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += a[c];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 40000;                       // 40000 * 128 = 5.12e6 VALU instructions per wave
    auto run = [&](int blocks) {
        std::vector<float> t;
        for (int i = 0; i < 7; ++i) {
            hipEventRecord(e0, 0);
            chains<<<blocks, 64>>>(out, reps, 0.9999f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    const double insts = (double)reps * 128.0;
    const float t1 = run(1);
    printf("1 wave on the chip:            %.3f ms -> %.2f cycles per instruction at 2.4 GHz (2.00 = issue bound at the full clock)\n", t1, t1 * 1e-3 * 2.4e9 / insts);
    for (int w : {1, 2, 4, 5, 8}) {
        const float tn = run(1024 * w);          // 1024 SIMDs, w waves each
        printf("%d wave(s) on each of 1024 SIMDs: %.3f ms -> effective clock %.2f GHz (%.0f %% of the single-wave rate)\n", w, tn, w * insts * 2.0 / (tn * 1e-3) / 1e9, 100.0 * w * t1 / tn);
    }
    // a short burst: does the chip start at the full clock and come down?
    for (int r : {400, 4000, 40000, 200000}) {
        std::vector<float> t;
        for (int i = 0; i < 5; ++i) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            chains<<<4096, 64>>>(out, r, 0.9999f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("4 waves per SIMD, %6d reps: %.3f ms -> effective clock %.2f GHz\n", r, t[2], 4 * (double)r * 128.0 * 2.0 / (t[2] * 1e-3) / 1e9);
    }
    return 0;
}
