// Calibration microbenchmarks for the external (HBM-bound) path on gfx950.
// Measures what a pure copy of the config-2 buffers (4 GiB in + 4 GiB out) reaches with
// the access shapes the FFT kernels can use, so the FFT kernels are judged against a
// same-run ceiling and not only against the 8 TB/s datasheet number.
//
//   copy16      : 16 B/lane grid-stride float4 copy (the guide's 6.29 TB/s shape)
//   tile8       : one wave per 8 KiB tile, 16 x 8 B/lane loads at 512 B stride, 16 stores
//                 (the direct register I/O shape of the N=1024 engine)
//   tile16      : one wave per 8 KiB tile, 8 x 16 B/lane loads (1 KiB per instruction)
//   tile8_lds   : tile8 plus two LDS round trips (cost model of the two exchanges)
//
// Build: hipcc -O3 --offload-arch=gfx950 membw.hip -o membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) copy16(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = in[i];
}

// wave tile = 1024 float2 = 8 KiB; ntiles = n2/1024
__global__ void __launch_bounds__(256) tile8(const float2* __restrict__ in, float2* __restrict__ out, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t t = wave; t < ntiles; t += nwaves) {
        const float2* p = in + t * 1024 + lane;
        float2 r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = p[64 * k];
        float2* q = out + t * 1024 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) q[64 * k] = r[k];
    }
}

__global__ void __launch_bounds__(256) tile16(const float4* __restrict__ in, float4* __restrict__ out, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t t = wave; t < ntiles; t += nwaves) {
        const float4* p = in + t * 512 + lane;
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[64 * k];
        float4* q = out + t * 512 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) q[64 * k] = r[k];
    }
}

// tile8 with two padded LDS transposes in between (no arithmetic): prices the LDS traffic
// of the two exchanges of the 16x4x16 engine when overlapped with streaming.
__global__ void __launch_bounds__(256) tile8_lds(const float2* __restrict__ in, float2* __restrict__ out, size_t ntiles) {
    __shared__ float2 s[4 * 1088];
    const int lane = threadIdx.x & 63;
    float2* sw = s + (threadIdx.x >> 6) * 1088;
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t t = wave; t < ntiles; t += nwaves) {
        const float2* p = in + t * 1024 + lane;
        float2 r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = p[64 * k];
        // exchange 1: write q1*68 + lane ; read (4*(v>>4)+c)*68 + (v&15) + 16*r2
#pragma unroll
        for (int k = 0; k < 16; ++k) sw[k * 68 + lane] = r[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) r[c * 4 + r2] = sw[(4 * (lane >> 4) + c) * 68 + (lane & 15) + 16 * r2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // exchange 2: write (v&15)*65 + 4*(v>>4)+c + 16*q2 ; read t2*65 + lane
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) sw[(lane & 15) * 65 + 4 * (lane >> 4) + c + 16 * q2] = r[c * 4 + q2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = sw[k * 65 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float2* q = out + t * 1024 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) q[64 * k] = r[k];
    }
}


// ---- variants to find the streaming ceiling of this box ----
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy16_nt(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        v4f v = __builtin_nontemporal_load(&in[i]);
        __builtin_nontemporal_store(v, &out[i]);
    }
}
__global__ void __launch_bounds__(256) copy16_u4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 768 < n4; i += stride) {
        float4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
        out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d;
    }
}
__global__ void __launch_bounds__(256) copy16_u4_nt(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 768 < n4; i += stride) {
        v4f a = __builtin_nontemporal_load(&in[i]), b = __builtin_nontemporal_load(&in[i + 256]);
        v4f c = __builtin_nontemporal_load(&in[i + 512]), d = __builtin_nontemporal_load(&in[i + 768]);
        __builtin_nontemporal_store(a, &out[i]); __builtin_nontemporal_store(b, &out[i + 256]);
        __builtin_nontemporal_store(c, &out[i + 512]); __builtin_nontemporal_store(d, &out[i + 768]);
    }
}
__global__ void __launch_bounds__(256) read16(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (; i + 768 < n4; i += stride) {
        float4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    if (acc.x == 123.456f) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) write16(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    float4 v = make_float4(threadIdx.x, 1, 2, 3);
    for (; i + 768 < n4; i += stride) { out[i] = v; out[i + 256] = v; out[i + 512] = v; out[i + 768] = v; }
}

template <class F>
static void run(const char* name, F launch, size_t bytes_moved, int rounds) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < rounds; ++i) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    double med = ms[ms.size() / 2], mn = ms[0];
    printf("%-12s median %.4f ms (%.1f GB/s)  min %.4f ms (%.1f GB/s)\n", name, med, bytes_moved / med / 1e6, mn, bytes_moved / mn / 1e6);
}

int main(int argc, char** argv) {
    size_t n2 = (size_t)1 << 29;  // float2 elements: 4 GiB
    if (argc > 1) n2 = (size_t)atoll(argv[1]);
    int rounds = 20;
    float2 *in, *out;
    CK(hipMalloc(&in, n2 * 8)); CK(hipMalloc(&out, n2 * 8));
    // random-ish fill (guide rule 25: never bench on zeros)
    {
        std::vector<float> h(1 << 22);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        for (size_t off = 0; off < n2 * 2; off += h.size())
            CK(hipMemcpy((float*)in + off, h.data(), std::min(h.size(), n2 * 2 - off) * 4, hipMemcpyHostToDevice));
    }
    size_t bytes = n2 * 16;  // read + write
    size_t ntiles = n2 / 1024;
    for (int blocks_per_cu : {4, 8}) {
        int grid = 256 * blocks_per_cu;
        printf("-- persistent grid %d x 256\n", grid);
        run("copy16", [&] { copy16<<<grid, 256>>>((const float4*)in, (float4*)out, n2 / 2); }, bytes, rounds);
        run("tile8", [&] { tile8<<<grid, 256>>>(in, out, ntiles); }, bytes, rounds);
        run("tile16", [&] { tile16<<<grid, 256>>>((const float4*)in, (float4*)out, ntiles); }, bytes, rounds);
        if (blocks_per_cu == 4) run("tile8_lds", [&] { tile8_lds<<<grid, 256>>>(in, out, ntiles); }, bytes, rounds);
    }
    {
        int grid = (int)(ntiles / 4);
        printf("-- one tile per wave, grid %d x 256\n", grid);
        run("tile8", [&] { tile8<<<grid, 256>>>(in, out, ntiles); }, bytes, rounds);
        run("tile16", [&] { tile16<<<grid, 256>>>((const float4*)in, (float4*)out, ntiles); }, bytes, rounds);
        run("tile8_lds", [&] { tile8_lds<<<grid, 256>>>(in, out, ntiles); }, bytes, rounds);
    }

    for (size_t nn : {n2 / 4, n2}) {
        printf("== streaming variants, %zu MiB per buffer\n", nn * 8 >> 20);
        for (int grid : {1024, 2048, 4096, 16384}) {
            printf("-- grid %d\n", grid);
            run("copy16", [&] { copy16<<<grid, 256>>>((const float4*)in, (float4*)out, nn / 2); }, nn * 16, rounds);
            run("copy16_nt", [&] { copy16_nt<<<grid, 256>>>((const v4f*)in, (v4f*)out, nn / 2); }, nn * 16, rounds);
            run("copy16_u4", [&] { copy16_u4<<<grid, 256>>>((const float4*)in, (float4*)out, nn / 2); }, nn * 16, rounds);
            run("copy16_u4_nt", [&] { copy16_u4_nt<<<grid, 256>>>((const v4f*)in, (v4f*)out, nn / 2); }, nn * 16, rounds);
            run("read16", [&] { read16<<<grid, 256>>>((const float4*)in, (float4*)out, nn / 2); }, nn * 8, rounds);
            run("write16", [&] { write16<<<grid, 256>>>((const float4*)in, (float4*)out, nn / 2); }, nn * 8, rounds);
        }
    }
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}
