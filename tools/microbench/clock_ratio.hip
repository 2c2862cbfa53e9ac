// clock_ratio.hip -- does the chip slow its shader clock when every SIMD runs VALU-dense code?  (hwmon shows the PLL target,
// 2.39 GHz, whatever runs: profiles/r02_clocks.txt.)  Every wave executes the same issue-bound stream of REPS x 128 v_add_f32.
// One wave on an otherwise idle chip runs at the full clock; W waves on every SIMD of the chip take W times as many SIMD
// cycles; rate(full chip) / rate(one wave) = W * t(1 wave) / t(full chip) is the clock ratio -- independent of what one
// instruction costs (a lone wave issues one v_add_f32 per ~4 cycles, two or more waves per SIMD one per 2.1:
// tools/microbench/valu_forms.hip), so the comparison is made at equal waves per SIMD: 1 wave on ONE SIMD vs 1 wave on EVERY SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize clock_ratio.hip -o clock_ratio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) chains(float* out, int reps, float k) {
    float a[16], b[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { a[c] = (float)(threadIdx.x + c); b[c] = k * (float)(c + 1); }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = a[c] + b[(c + u) & 15];
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
    printf("1 wave on the chip:               %.3f ms = %.2f ns per instruction\n", t1, t1 * 1e6 / insts);
    for (int w : {1, 2, 4, 5, 8}) {
        const float tn = run(1024 * w);          // 1024 SIMDs, w waves each
        printf("%d wave(s) on each of 1024 SIMDs:  %.3f ms = %.2f ns per instruction per SIMD%s\n", w, tn, tn * 1e6 / (insts * w),
               w == 1 ? "   <- against the line above: the clock ratio full chip / idle chip" : "");
    }
    return 0;
}
