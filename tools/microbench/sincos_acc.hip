// Twiddle accuracy probe: how far are the candidate on-device twiddle sources from the
// correctly rounded fp32 value of W_N^m = (cos, -sin)(2*pi*m/N), N = 4096?
//   hw    : v_cos_f32 / v_sin_f32 on the exact fraction m/N (argument in revolutions)
//   fast  : __cosf/__sinf(2*pi*m/N)   (what --use_fast_math gives the reference)
//   ocml  : sincospif(2*m/N)
// Build: hipcc -O3 --offload-arch=gfx950 sincos_acc.hip -o sincos_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k(float2* hw, float2* fast, float2* ocml, int N) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    float f = (float)m / (float)N;  // exact for power-of-two N
    hw[m] = make_float2(__builtin_amdgcn_cosf(f), -__builtin_amdgcn_sinf(f));
    float a = -6.283185308f * f;
    fast[m] = make_float2(__cosf(a), __sinf(a));
    float s, c;
    sincospif(2.0f * f, &s, &c);
    ocml[m] = make_float2(c, -s);
}

int main() {
    const int N = 4096;
    float2 *d[3];
    for (auto& p : d) hipMalloc(&p, N * 8);
    k<<<N / 256, 256>>>(d[0], d[1], d[2], N);
    hipDeviceSynchronize();
    const char* names[3] = {"hw v_sin/v_cos", "fast __sinf/__cosf", "ocml sincospif"};
    for (int v = 0; v < 3; ++v) {
        std::vector<float2> h(N);
        hipMemcpy(h.data(), d[v], N * 8, hipMemcpyDeviceToHost);
        double maxabs = 0, sumsq = 0;
        int worst = 0;
        for (int m = 0; m < N; ++m) {
            double ang = -2.0 * M_PI * (double)m / N;
            double ex = cos(ang), ey = sin(ang);
            double e = hypot(h[m].x - ex, h[m].y - ey);
            sumsq += e * e;
            if (e > maxabs) { maxabs = e; worst = m; }
        }
        printf("%-20s max |err| = %.3e at m=%d   rms = %.3e\n", names[v], maxabs, worst, sqrt(sumsq / N));
    }
    return 0;
}
