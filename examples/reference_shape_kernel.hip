// reference_shape_kernel.hip -- user kernels written the way a KAdamek/SMFFT user writes them (reference
// README.md:10-18, 48-60; CT/FFT-GPU-32bit.cu:534-551, 586-595): one thread block per fft_length elements,
// blockDim.x = fft_length / 4, four float2 per thread into `__shared__ float2 s[P::fft_sm_required]`,
// __syncthreads, do_SMFFT_CT_DIT<P>(s), __syncthreads, four float2 per thread out -- and launched with the
// reference's grid arithmetic.  Nothing here knows about 64-lane waves, tiles or regions: the only change against
// a CUDA build is the include line.  tests/test_gpu_parity.py runs every length and variant against the oracle.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include examples/reference_shape_kernel.hip
#include <hip/hip_runtime.h>
#include <smfft_device.hpp>

// a user's own kernel around the device function (the "expected to be called within a GPU kernel" use, README.md:10)
template <class const_params>
__global__ void user_fft_kernel(float2* d_input, float2* d_output) {
    __shared__ float2 s_data[const_params::fft_sm_required];
    const int offset = blockIdx.x * const_params::fft_length;
    for (int k = 0; k < 4; k++) s_data[threadIdx.x + k * const_params::fft_length_quarter] = d_input[offset + threadIdx.x + k * const_params::fft_length_quarter];
    __syncthreads();
    do_SMFFT_CT_DIT<const_params>(s_data);
    __syncthreads();
    for (int k = 0; k < 4; k++) d_output[offset + threadIdx.x + k * const_params::fft_length_quarter] = s_data[threadIdx.x + k * const_params::fft_length_quarter];
}

// which = 0: the user's kernel above; 1: the library's reference-shaped SMFFT_DIT_external<P>(in, out)
template <class P>
static int launch_ct(float2* in, float2* out, int nFFTs, int which, hipStream_t st) {
    // the reference's launch arithmetic (CT:586-595): nFFTs blocks of N/4 threads; N = 32 / 64: nFFTs/4, nFFTs/2 blocks of 32
    dim3 grid(nFFTs / (P::fft_length / P::fft_size)), block(P::fft_length / 4);
    if (which == 0) user_fft_kernel<P><<<grid, block, 0, st>>>(in, out);
    else SMFFT_DIT_external<P><<<grid, block, 0, st>>>(in, out);
    return (int)hipGetLastError();
}

// The wave64-full classes of N = 32 / 64 / 128 (FFT_<N>_..._wave64: blockDim.x = 64, 256 elements per block): whole blocks of
// 256 / N transforms run on them, the last nFFTs mod (256 / N) transforms -- upstream only requires a multiple of 128 / N,
// CT:835-836 -- on the upstream-shaped class.  which = 2: the user's kernel, 3: the library's two-argument kernel.
template <class P64, class P32>
static int launch_ct_wave64(float2* in, float2* out, int nFFTs, int which, hipStream_t st) {
    constexpr int per64 = P64::fft_length / P64::fft_size, per32 = P32::fft_length / P32::fft_size;
    const int full = nFFTs / per64, rest = (nFFTs - full * per64) / per32;
    if (full > 0) {
        if (which == 2) user_fft_kernel<P64><<<dim3(full), dim3(P64::fft_length / 4), 0, st>>>(in, out);
        else SMFFT_DIT_external<P64><<<dim3(full), dim3(P64::fft_length / 4), 0, st>>>(in, out);
    }
    if (rest > 0) {
        float2 *in_tail = in + (size_t)full * P64::fft_length, *out_tail = out + (size_t)full * P64::fft_length;
        if (which == 2) user_fft_kernel<P32><<<dim3(rest), dim3(P32::fft_length / 4), 0, st>>>(in_tail, out_tail);
        else SMFFT_DIT_external<P32><<<dim3(rest), dim3(P32::fft_length / 4), 0, st>>>(in_tail, out_tail);
    }
    return (int)hipGetLastError();
}
#define CT_CASE_SMALL(N)                                                                                                                                   \
    case N:                                                                                                                                                \
        if (which >= 2) {                                                                                                                                  \
            if (!inverse && reorder) return launch_ct_wave64<FFT_##N##_forward_wave64, FFT_##N##_forward>(in, out, nFFTs, which, st);                      \
            if (!inverse && !reorder) return launch_ct_wave64<FFT_##N##_forward_noreorder_wave64, FFT_##N##_forward_noreorder>(in, out, nFFTs, which, st); \
            if (inverse && reorder) return launch_ct_wave64<FFT_##N##_inverse_wave64, FFT_##N##_inverse>(in, out, nFFTs, which, st);                       \
            return launch_ct_wave64<FFT_##N##_inverse_noreorder_wave64, FFT_##N##_inverse_noreorder>(in, out, nFFTs, which, st);                           \
        }                                                                                                                                                  \
        if (!inverse && reorder) return launch_ct<FFT_##N##_forward>(in, out, nFFTs, which, st);                                                           \
        if (!inverse && !reorder) return launch_ct<FFT_##N##_forward_noreorder>(in, out, nFFTs, which, st);                                                \
        if (inverse && reorder) return launch_ct<FFT_##N##_inverse>(in, out, nFFTs, which, st);                                                            \
        return launch_ct<FFT_##N##_inverse_noreorder>(in, out, nFFTs, which, st);

#define CT_CASE(N)                                                                                              \
    case N:                                                                                                     \
        if (!inverse && reorder) return launch_ct<FFT_##N##_forward>(in, out, nFFTs, which, st);               \
        if (!inverse && !reorder) return launch_ct<FFT_##N##_forward_noreorder>(in, out, nFFTs, which, st);    \
        if (inverse && reorder) return launch_ct<FFT_##N##_inverse>(in, out, nFFTs, which, st);                \
        return launch_ct<FFT_##N##_inverse_noreorder>(in, out, nFFTs, which, st);

extern "C" int smfft_example_reference_shape_ct(void* d_in, void* d_out, int FFT_size, int nFFTs, int inverse, int reorder, int which, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    if (which >= 2 && FFT_size >= 256) which -= 2;      // (the wave64-full classes exist for N <= 128 only)
    switch (FFT_size) {
        CT_CASE_SMALL(32) CT_CASE_SMALL(64) CT_CASE_SMALL(128) CT_CASE(256) CT_CASE(512) CT_CASE(1024) CT_CASE(2048) CT_CASE(4096)
        default: return -1;
    }
}

// Stockham C2C program in the reference's shape: <<<nFFTs, N/4, N*8 bytes>>> (ST:309-319)
#define ST_CASE(N) case N: FFT_GPU_external<FFT_##N><<<dim3(nFFTs), dim3(N / 4), N * 8, st>>>(in, out); break;
extern "C" int smfft_example_reference_shape_st(void* d_in, void* d_out, int FFT_size, int nFFTs, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    switch (FFT_size) {
        ST_CASE(256) ST_CASE(512) ST_CASE(1024) ST_CASE(2048) ST_CASE(4096)
        default: return -1;
    }
    return (int)hipGetLastError();
}

// R2C / C2R program in the reference's shape: real length FFT_size -> FFT_<FFT_size/2>, <<<nFFTs, (FFT_size/2)/4>>> (RC:399-428)
#define RC_CASE(NREAL, L)                                                                                        \
    case NREAL:                                                                                                  \
        if (!inverse) FFT_GPU_R2C_C2R_external<FFT_##L, FFT_forward><<<dim3(nFFTs), dim3(L / 4), 0, st>>>(in, out); \
        else FFT_GPU_R2C_C2R_external<FFT_##L, FFT_inverse><<<dim3(nFFTs), dim3(L / 4), 0, st>>>(in, out);       \
        break;
extern "C" int smfft_example_reference_shape_rc(void* d_in, void* d_out, int FFT_size, int nFFTs, int inverse, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    switch (FFT_size) {
        RC_CASE(512, 256) RC_CASE(1024, 512) RC_CASE(2048, 1024) RC_CASE(4096, 2048)
        default: return -1;
    }
    return (int)hipGetLastError();
}

// the two `multiple` kernels of the Stockham programs in the reference's shape (compile check; NREUSES applications overflow
// fp32 by design, as upstream, so only the launch is exercised)
extern "C" int smfft_example_reference_shape_multiple(void* d_in, void* d_out, int nFFTs, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FFT_GPU_multiple<FFT_1024><<<dim3(nFFTs), dim3(256), 1024 * 8, st>>>((float2*)d_in, (float2*)d_out);
    FFT_GPU_R2C_C2R_multiple<FFT_1024, FFT_forward><<<dim3(nFFTs), dim3(256), 0, st>>>((float2*)d_in, (float2*)d_out);
    SMFFT_DIT_multiple<FFT_1024_forward><<<dim3(nFFTs), dim3(256), 0, st>>>((float2*)d_in, (float2*)d_out);
    return (int)hipGetLastError();
}

// one of them alone, for timing (tools/reference_contract.py, bench.py): which = 0 SMFFT_DIT_multiple<FFT_1024_forward>,
// 1 SMFFT_DIT_multiple<FFT_1024_forward_noreorder>, 2 FFT_GPU_multiple<FFT_1024>, 3 FFT_GPU_R2C_C2R_multiple<FFT_1024, FFT_forward>
extern "C" int smfft_example_reference_shape_multiple_one(void* d_in, void* d_out, int nBlocks, int which, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    switch (which) {
        case 0: SMFFT_DIT_multiple<FFT_1024_forward><<<dim3(nBlocks), dim3(256), 0, st>>>(in, out); break;
        case 1: SMFFT_DIT_multiple<FFT_1024_forward_noreorder><<<dim3(nBlocks), dim3(256), 0, st>>>(in, out); break;
        case 2: FFT_GPU_multiple<FFT_1024><<<dim3(nBlocks), dim3(256), 1024 * 8, st>>>(in, out); break;
        case 3: FFT_GPU_R2C_C2R_multiple<FFT_1024, FFT_forward><<<dim3(nBlocks), dim3(256), 0, st>>>(in, out); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// SMFFT_DIT_multiple<P> at any length in the reference's launch shape (CT:669-683: blocks of fft_length / 4 threads, each
// holding fft_length / N transforms), forward; for timing the in-LDS contract path per length (tools/reference_contract.py)
template <class P>
static int launch_ct_multiple(float2* in, float2* out, int nBlocks, hipStream_t st) {
    SMFFT_DIT_multiple<P><<<dim3(nBlocks), dim3(P::fft_length / 4), 0, st>>>(in, out);
    return (int)hipGetLastError();
}
#define CTM_CASE(N) case N: return reorder ? launch_ct_multiple<FFT_##N##_forward>(in, out, nBlocks, st) : launch_ct_multiple<FFT_##N##_forward_noreorder>(in, out, nBlocks, st);
extern "C" int smfft_example_reference_shape_ct_multiple(void* d_in, void* d_out, int FFT_size, int nBlocks, int reorder, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    switch (FFT_size) {
        CTM_CASE(32) CTM_CASE(64) CTM_CASE(128) CTM_CASE(256) CTM_CASE(512) CTM_CASE(1024) CTM_CASE(2048) CTM_CASE(4096)
        default: return -1;
    }
}
// the same with the wave64-full classes of N = 32 / 64 / 128: nBlocks blocks of 64 threads, 256 elements each
#define CTM64_CASE(N) case N: return reorder ? launch_ct_multiple<FFT_##N##_forward_wave64>(in, out, nBlocks, st) : launch_ct_multiple<FFT_##N##_forward_noreorder_wave64>(in, out, nBlocks, st);
extern "C" int smfft_example_reference_shape_ct_multiple_wave64(void* d_in, void* d_out, int FFT_size, int nBlocks, int reorder, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    switch (FFT_size) {
        CTM64_CASE(32) CTM64_CASE(64) CTM64_CASE(128)
        default: return -1;
    }
}

// The library use case in the reference's shape: a batched circular convolution y = IFFT(FFT(x) .* H) / N written by a user
// who knows nothing but the reference's contract -- one block per series, blockDim.x = N / 4, do_SMFFT_CT_DIT<forward>,
// a pointwise product in shared memory, do_SMFFT_CT_DIT<inverse> (README.md:10-16).  examples/fft_convolution.hip has the
// same pipeline on the engine's tiled contract and on its register-level interface.
// Two lines of ordinary kernel hygiene, both the user's own (round 6; tools/ab_convolution.py, profiles/r06_convolution_user_kernel.txt):
// the filter's four values are fetched WITH the series -- fetched between the two transforms, behind a barrier, their L2 latency
// was exposed once per block: 2.05 -> 1.88 ms on the config-2 batch -- and the kernel states its block size (-> 1.79 ms).
template <class Fwd, class Inv>
__global__ void __launch_bounds__(Fwd::fft_length_quarter) user_convolution_kernel(const float2* d_x, const float2* d_H, float2* d_y) {
    __shared__ float2 s_data[Fwd::fft_sm_required];
    constexpr int N = Fwd::fft_length, Q = Fwd::fft_length_quarter;
    const int offset = blockIdx.x * N;
    float2 h[4];
    for (int k = 0; k < 4; k++) s_data[threadIdx.x + k * Q] = d_x[offset + threadIdx.x + k * Q];
    for (int k = 0; k < 4; k++) h[k] = d_H[threadIdx.x + k * Q];
    __syncthreads();
    do_SMFFT_CT_DIT<Fwd>(s_data);
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        const float2 a = s_data[threadIdx.x + k * Q];
        s_data[threadIdx.x + k * Q] = make_float2(a.x * h[k].x - a.y * h[k].y, a.x * h[k].y + a.y * h[k].x);
    }
    __syncthreads();
    do_SMFFT_CT_DIT<Inv>(s_data);
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        const float2 v = s_data[threadIdx.x + k * Q];
        d_y[offset + threadIdx.x + k * Q] = make_float2(v.x * (1.0f / N), v.y * (1.0f / N));
    }
}
// ... and with the register form of the same functions (do_SMFFT_CT_DIT_registers: a thread's four elements
// threadIdx.x + m N/4 in and out, the shared array as scratch): the series never lies in shared memory in natural order.
template <class Fwd, class Inv>
__global__ void __launch_bounds__(Fwd::fft_length_quarter) user_convolution_kernel_registers(const float2* d_x, const float2* d_H, float2* d_y) {
    __shared__ float2 s_scratch[Fwd::fft_sm_required];
    constexpr int N = Fwd::fft_length, Q = Fwd::fft_length_quarter;
    const int offset = blockIdx.x * N;
    float2 x[4], h[4];
    for (int k = 0; k < 4; k++) x[k] = d_x[offset + threadIdx.x + k * Q];
    for (int k = 0; k < 4; k++) h[k] = d_H[threadIdx.x + k * Q];
    do_SMFFT_CT_DIT_registers<Fwd>(x, s_scratch);
    for (int k = 0; k < 4; k++) {
        const float2 a = x[k];
        x[k] = make_float2((a.x * h[k].x - a.y * h[k].y) * (1.0f / N), (a.x * h[k].y + a.y * h[k].x) * (1.0f / N));
    }
    __syncthreads();   // the forward transform's last reads of s_scratch are done before the inverse one writes it
    do_SMFFT_CT_DIT_registers<Inv>(x, s_scratch);
    for (int k = 0; k < 4; k++) d_y[offset + threadIdx.x + k * Q] = x[k];
}
// which = 0: shared-memory form, 1: register form
extern "C" int smfft_example_reference_shape_convolve_1024(const void* d_x, const void* d_H, void* d_y, int nSeries, void* stream) {
    if (nSeries <= 0) return 0;
    user_convolution_kernel<FFT_1024_forward, FFT_1024_inverse><<<dim3(nSeries), dim3(256), 0, (hipStream_t)stream>>>((const float2*)d_x, (const float2*)d_H, (float2*)d_y);
    return (int)hipGetLastError();
}
extern "C" int smfft_example_reference_shape_convolve_1024_registers(const void* d_x, const void* d_H, void* d_y, int nSeries, void* stream) {
    if (nSeries <= 0) return 0;
    user_convolution_kernel_registers<FFT_1024_forward, FFT_1024_inverse><<<dim3(nSeries), dim3(256), 0, (hipStream_t)stream>>>((const float2*)d_x, (const float2*)d_H, (float2*)d_y);
    return (int)hipGetLastError();
}

// SMFFT_DIT_multiple<P>'s call pattern with two applications instead of NREUSES (whose 100 un-normalised transforms overflow
// fp32 by design, CT:563-565): load, (do_SMFFT_CT_DIT<P>, barrier) x 2, store -- so that the back-to-back calls can be
// checked: forward twice in natural order is N * x[(-n) mod N].
template <class const_params>
__global__ void user_fft_twice_kernel(float2* d_input, float2* d_output) {
    __shared__ float2 s_data[const_params::fft_sm_required];
    const int offset = blockIdx.x * const_params::fft_length;
    for (int k = 0; k < 4; k++) s_data[threadIdx.x + k * const_params::fft_length_quarter] = d_input[offset + threadIdx.x + k * const_params::fft_length_quarter];
    __syncthreads();
    for (int f = 0; f < 2; f++) {
        do_SMFFT_CT_DIT<const_params>(s_data);
        __syncthreads();
    }
    for (int k = 0; k < 4; k++) d_output[offset + threadIdx.x + k * const_params::fft_length_quarter] = s_data[threadIdx.x + k * const_params::fft_length_quarter];
}
// the same with a RUN-TIME number of applications (a loop the compiler cannot unroll: whatever the device function does to the
// wave's scalar state -- its inline assembly writes VCC and SCC -- must leave the loop's own control intact)
template <class const_params>
__global__ void user_fft_times_kernel(float2* d_input, float2* d_output, int times) {
    __shared__ float2 s_data[const_params::fft_sm_required];
    const int offset = blockIdx.x * const_params::fft_length;
    for (int k = 0; k < 4; k++) s_data[threadIdx.x + k * const_params::fft_length_quarter] = d_input[offset + threadIdx.x + k * const_params::fft_length_quarter];
    __syncthreads();
    for (int f = 0; f < times; f++) {
        do_SMFFT_CT_DIT<const_params>(s_data);
        __syncthreads();
    }
    for (int k = 0; k < 4; k++) d_output[offset + threadIdx.x + k * const_params::fft_length_quarter] = s_data[threadIdx.x + k * const_params::fft_length_quarter];
}
template <class P>
static int launch_ct_times(float2* in, float2* out, int nFFTs, int times, hipStream_t st) {
    user_fft_times_kernel<P><<<dim3(nFFTs / (P::fft_length / P::fft_size)), dim3(P::fft_length / 4), 0, st>>>(in, out, times);
    return (int)hipGetLastError();
}
#define CTT_CASE(N) case N: return reorder ? launch_ct_times<FFT_##N##_forward>(in, out, nFFTs, times, st) : launch_ct_times<FFT_##N##_forward_noreorder>(in, out, nFFTs, times, st);
#define CTT64_CASE(N) case N: return reorder ? launch_ct_times<FFT_##N##_forward_wave64>(in, out, nFFTs, times, st) : launch_ct_times<FFT_##N##_forward_noreorder_wave64>(in, out, nFFTs, times, st);
// wave64 != 0: the 64-thread classes of N <= 128 (nFFTs a multiple of 256 / N)
extern "C" int smfft_example_reference_shape_ct_times(void* d_in, void* d_out, int FFT_size, int nFFTs, int reorder, int times, int wave64, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    if (wave64 && FFT_size <= 128) {
        switch (FFT_size) {
            CTT64_CASE(32) CTT64_CASE(64) CTT64_CASE(128)
            default: return -1;
        }
    }
    switch (FFT_size) {
        CTT_CASE(32) CTT_CASE(64) CTT_CASE(128) CTT_CASE(256) CTT_CASE(512) CTT_CASE(1024) CTT_CASE(2048) CTT_CASE(4096)
        default: return -1;
    }
}

template <class P>
static int launch_ct_twice(float2* in, float2* out, int nFFTs, hipStream_t st) {
    user_fft_twice_kernel<P><<<dim3(nFFTs / (P::fft_length / P::fft_size)), dim3(P::fft_length / 4), 0, st>>>(in, out);
    return (int)hipGetLastError();
}
#define CT2_CASE(N) case N: return reorder ? launch_ct_twice<FFT_##N##_forward>(in, out, nFFTs, st) : launch_ct_twice<FFT_##N##_forward_noreorder>(in, out, nFFTs, st);
extern "C" int smfft_example_reference_shape_ct_twice(void* d_in, void* d_out, int FFT_size, int nFFTs, int reorder, void* stream) {
    float2 *in = (float2*)d_in, *out = (float2*)d_out;
    hipStream_t st = (hipStream_t)stream;
    switch (FFT_size) {
        CT2_CASE(32) CT2_CASE(64) CT2_CASE(128) CT2_CASE(256) CT2_CASE(512) CT2_CASE(1024) CT2_CASE(2048) CT2_CASE(4096)
        default: return -1;
    }
}
