// Cache-policy probe for the streaming copy: raw buffer loads/stores of 8 B/lane with every
// combination of the gfx950 cache bits (aux: 1 = sc0, 2 = nt, 16 = sc1) on the load and on the
// store side, same tile shape and grid as the FFT kernels (one wave per 8 KiB chunk, 12288 workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int LAUX, int SAUX>
__global__ void __launch_bounds__(256) copy_policy(const float2* __restrict__ in, float2* __restrict__ out, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float2* g = in + tile * 4096 + wave * 1024;
        float2* o = out + tile * 4096 + wave * 1024;
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 8192, 0x00020000);
        auto ws = __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 8192, 0x00020000);
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_amdgcn_raw_buffer_load_b64(rs, (lane + 64 * c) * 8, 0, LAUX);
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_amdgcn_raw_buffer_store_b64(r[c], ws, (lane + 64 * c) * 8, 0, SAUX);
    }
}

template <int L, int S>
static void run(const float2* in, float2* out, long ntiles) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) copy_policy<L, S><<<12288, 256>>>(in, out, ntiles);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 15; ++i) {
        CK(hipEventRecord(a, 0)); copy_policy<L, S><<<12288, 256>>>(in, out, ntiles); CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("load aux %2d store aux %2d: median %.4f ms %.1f GB/s  (min %.4f)\n", L, S, ms[7], ntiles * 65536.0 / ms[7] / 1e6, ms[0]);
}

int main() {
    const long n2 = 1L << 29; float2 *in, *out;
    CK(hipMalloc(&in, n2 * 8)); CK(hipMalloc(&out, n2 * 8));
    std::vector<float> h(1 << 22); unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    for (size_t off = 0; off < (size_t)n2 * 2; off += h.size()) CK(hipMemcpy((float*)in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const long nt = n2 / 4096;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(in, out, nt); run<2, 2>(in, out, nt); run<0, 2>(in, out, nt); run<2, 0>(in, out, nt);
        run<1, 1>(in, out, nt); run<16, 16>(in, out, nt); run<17, 17>(in, out, nt); run<2, 17>(in, out, nt);
        run<2, 16>(in, out, nt); run<3, 3>(in, out, nt); run<18, 18>(in, out, nt); run<2, 18>(in, out, nt); run<2, 19>(in, out, nt);
    }
    return 0;
}
