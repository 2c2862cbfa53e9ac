// fft_convolution.hip -- the library use case of SMFFT (reference README.md:10-16: "the code is
// expected to be called within a GPU kernel"): a user kernel that calls the device function.
//
// Batched circular convolution of length-1024 complex series with one filter whose spectrum H is
// given:  y = IFFT( FFT(x) .* H ) / N.  One 256-thread workgroup owns 4 series (a 4096-element
// tile); everything between the global load and the global store stays in LDS / registers:
//     global -> LDS | do_SMFFT_CT_DIT<FFT_1024_forward> | .* H | do_SMFFT_CT_DIT<FFT_1024_inverse> | -> global
// (the device functions in the engine's tiled contract, namespace smfft::tiled; the reference-shaped form of the same
//  functions -- blockDim.x = N/4, contiguous data -- is exercised by examples/reference_shape_kernel.hip)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared \
//        -I include examples/fft_convolution.hip -o smfft_amd/libsmfft_examples.so
#include <hip/hip_runtime.h>
#include <smfft_device.hpp>

template <class Fwd, class Inv>
__global__ void __launch_bounds__(256) convolve_kernel(const float2* __restrict__ x, const float2* __restrict__ H, float2* __restrict__ y, int nSeries) {
    constexpr int N = Fwd::fft_size;
    __shared__ float2 s[Fwd::tile_sm_required];
    const long first = (long)blockIdx.x * Fwd::fft_per_block;
    // a thread meets N / 256 different points of the spectrum (e mod N with e = threadIdx.x + 256 i): the filter's values for
    // them are fetched with the series, not between the transforms behind a barrier
    constexpr int kPoints = N > 256 ? N / 256 : 1;
    float2 h[kPoints];
    // natural order, series j of the workgroup at s[j*fft_region + n]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = threadIdx.x + 256 * i, j = e / N, n = e % N;
        s[j * Fwd::fft_region + n] = (first + j < nSeries) ? x[first * N + e] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kPoints; ++i) h[i] = H[(threadIdx.x + 256 * i) % N];
    __syncthreads();
    smfft::tiled::do_SMFFT_CT_DIT<Fwd>(s);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = threadIdx.x + 256 * i, j = e / N, k = e % N;
        const float2 a = s[j * Fwd::fft_region + k], g = h[i % kPoints];
        s[j * Fwd::fft_region + k] = make_float2(a.x * g.x - a.y * g.y, a.x * g.y + a.y * g.x);
    }
    __syncthreads();
    smfft::tiled::do_SMFFT_CT_DIT<Inv>(s);
    __syncthreads();
    const float scale = 1.0f / N;
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int j = e / N, n = e % N;
        if (first + j < nSeries) {
            const float2 v = s[j * Fwd::fft_region + n];
            y[first * N + e] = make_float2(v.x * scale, v.y * scale);
        }
    }
}

// The same convolution on the register-level interface of the engine (smfft::Engine): the spectrum
// never goes back to LDS in natural order -- thread u of an FFT holds X[u + T*q] after transform(),
// multiplies it by H[u + T*q] in registers and feeds the inverse transform directly; global memory is
// read and written straight from registers.  Per series: 2 LDS exchanges instead of 2 exchanges + 4
// natural-order round trips.
template <int N>
__global__ void __launch_bounds__(256) convolve_kernel_registers(const float2* __restrict__ x, const float2* __restrict__ H, float2* __restrict__ y, int nSeries) {
    using G = smfft::Geometry<N>;
    __shared__ float2 s[4352];
    smfft::Engine<N, 0, 1> fwd;
    smfft::Engine<N, 1, 1> inv;
    fwd.init(threadIdx.x);
    inv.init(threadIdx.x);
    float2* sf = s + fwd.fft * G::SF;
    float2 h[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {   // 1 / N folded into the filter: the inverse transform's result is stored as it is
        const float2 t = H[fwd.u + G::T * q];
        h[q] = make_float2(t.x * (1.0f / N), t.y * (1.0f / N));
    }
    const int ntiles = (nSeries + G::kFftsPerBlock - 1) / G::kFftsPerBlock;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f = (long)tile * G::kFftsPerBlock + fwd.fft;
        const bool active = f < nSeries;
        float2 r[16];
        fwd.load_global(r, x + (active ? f : 0) * N);
        if (G::kMultiWave) __syncthreads();
        fwd.transform(r, sf);
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = smfft::cmul(r[q], h[q]);
        smfft::fft_sync<G::kMultiWave>();          // the forward transform's last LDS reads are done
        inv.transform(r, sf);
        inv.store_global(r, y + f * N, active);
    }
}

extern "C" int smfft_example_convolve_1024_registers(const void* d_x, const void* d_H, void* d_y, int nSeries, void* stream) {
    if (nSeries <= 0) return 0;
    int grid = (nSeries + 3) / 4;
    if (grid > 12288) grid = 12288;
    convolve_kernel_registers<1024><<<grid, 256, 0, (hipStream_t)stream>>>((const float2*)d_x, (const float2*)d_H, (float2*)d_y, nSeries);
    return (int)hipGetLastError();
}

extern "C" int smfft_example_convolve_1024(const void* d_x, const void* d_H, void* d_y, int nSeries, void* stream) {
    if (nSeries <= 0) return 0;
    const int grid = (nSeries + FFT_1024_forward::fft_per_block - 1) / FFT_1024_forward::fft_per_block;
    convolve_kernel<FFT_1024_forward, FFT_1024_inverse><<<grid, 256, 0, (hipStream_t)stream>>>((const float2*)d_x, (const float2*)d_H, (float2*)d_y, nSeries);
    return (int)hipGetLastError();
}

extern "C" int smfft_example_convolve_256(const void* d_x, const void* d_H, void* d_y, int nSeries, void* stream) {
    if (nSeries <= 0) return 0;
    const int grid = (nSeries + FFT_256_forward::fft_per_block - 1) / FFT_256_forward::fft_per_block;
    convolve_kernel<FFT_256_forward, FFT_256_inverse><<<grid, 256, 0, (hipStream_t)stream>>>((const float2*)d_x, (const float2*)d_H, (float2*)d_y, nSeries);
    return (int)hipGetLastError();
}
