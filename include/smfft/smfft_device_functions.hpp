// smfft_device_functions.hpp -- the device-function surface under the reference's names.
//
// Two forms of every function:
//
// (1) THE REFERENCE'S CONTRACT (global namespace; what a kernel written for KAdamek/SMFFT calls):
//       do_SMFFT_CT_DIT<P>(s)              CT/FFT-GPU-32bit.cu:334-532, README.md:10-18
//       do_FFT_Stockham_mk6<P>(s)          ST/FFT-GPU-32bit-Stockham.cu:97-240
//       do_FFT_Stockham_C2C<P,D>(s)        RC/FFT-GPU-32bit-Stockham.cu:106-266
//       do_FFT_Stockham_R2C_C2R<P,D>(s)    RC/FFT-GPU-32bit-Stockham.cu:269-344
//     and the kernels in the reference's launch shape
//       SMFFT_DIT_external<P>(in, out), SMFFT_DIT_multiple<P>(in, out)        CT:534-572, <<<nFFTs*N/fft_length, fft_length/4>>>
//       FFT_GPU_external<P>(in, out), FFT_GPU_multiple<P>(in, out)            ST:243-278, <<<nFFTs, N/4, N*8>>>
//       FFT_GPU_R2C_C2R_external<P,D>(in, out), FFT_GPU_R2C_C2R_multiple<P,D> RC:349-384, <<<nFFTs, L/4>>>
//     Same contract as upstream: blockDim.x = fft_length / 4 (CT: 32 for N <= 128; Stockham: N / 4), the data in
//     s[0 .. fft_length) contiguous and in natural order before and after, in place, EVERY thread of the block calls,
//     the caller barriers before the call (CT also after it).  LDS the caller provides: CT P::fft_sm_required
//     (= 17 * fft_length / 16 here); Stockham exactly N float2, R2C/C2R L + 1 (as upstream: ST:319, RC:351).
//     How it runs on a 64-lane wave: the engine needs N / 16 threads per FFT, a quarter of what the contract
//     launches, so for N <= 1024 the block's first wave does the work (lanes beyond N / 16 repeat it, which the
//     lane-exchange instructions need) and the other waves return at once; for N = 2048 / 4096 every wave runs
//     the transform of thread (threadIdx.x mod N/16), so that all waves meet at the same workgroup barriers.
//     That is a compatibility path -- a quarter of the lanes do useful work; kernels that want the engine's speed
//     use form (2) or the Engine directly (examples/fft_convolution.hip).
//
// (2) THE ENGINE'S TILED CONTRACT (namespace smfft::tiled; what this library's own kernels are built on):
//     256-thread workgroups own 4096 float2 = P::fft_per_block FFTs; `s` is an LDS array of P::tile_sm_required
//     (4352) float2; FFT j of the workgroup occupies s[j * P::fft_region + n], n in [0, N), natural order before
//     and after; all 256 threads call; callers barrier between filling s and the call and between the call and
//     reading s.
#pragma once
#include "smfft_engine.hpp"
#include "SM_FFT_stockham_parameters.hpp"

#ifndef NREUSES
#define NREUSES 100
#endif

namespace smfft {

// ------------------------------------------------------------------------------------------------
// R2C / C2R (real length 2L through a complex FFT of length L).  RC:269-344.
// Hermitian split (forward, after the C2C) / merge (inverse, before the C2C) on the natural
// layout in LDS; thread u of an FFT handles the 8 index pairs i = 1 + u + T*j, (i, L - i).
// ------------------------------------------------------------------------------------------------
template <int L, int DIR>
__device__ __forceinline__ void hermitian_pass(float2* sf, int u) {
    constexpr int T = L / 16;
    constexpr float ohx = DIR ? -0.5f : 0.5f;   // upstream's (ohx, ohy) = (1/2, -1/2) forward, (-1/2, 1/2) inverse (RC:289-328)
    if (DIR) {
        if (u == 0) {
            float2 z = sf[0];
            sf[0] = make_float2(0.5f * (z.x + z.y), 0.5f * (z.x - z.y));
        }
    }
    // H1 = S/2 with S = (A.x + B.x, A.y - B.y); H2 = (ohx * D.x, ohy * D.y) with D = (A.y + B.y, A.x - B.x) and ohy = -ohx,
    // so W * H2 = ((ohx W).x D.x + (ohx W).y D.y, (ohx W).y D.x - (ohx W).x D.y): with ohx folded into the twiddle (loop
    // invariant across the applications of the in-LDS kernels) a pair costs 12 instructions instead of 16
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 1 + u + T * j;
        const float2 A = sf[i], B = sf[L - i];
        const float2 W = twiddle<DIR>(i * (4096 / (2 * L)));
        const float2 Wh = make_float2(ohx * W.x, ohx * W.y);
        const float2 S = make_float2(A.x + B.x, A.y - B.y);
        const float2 D = make_float2(A.y + B.y, A.x - B.x);
        const float2 WH = make_float2(fmaf(Wh.x, D.x, Wh.y * D.y), fmaf(Wh.y, D.x, -Wh.x * D.y));
        sf[i] = make_float2(fmaf(0.5f, S.x, WH.x), fmaf(0.5f, S.y, WH.y));
        sf[L - i] = make_float2(fmaf(0.5f, S.x, -WH.x), fmaf(-0.5f, S.y, WH.y));   // for i == L/2 this value stays (RC:308)
    }
    if (!DIR) {
        if (u == 0) {   // sf[0] is not touched by the pair loop (i >= 1, L - i >= L/2)
            float2 z = sf[0];
            sf[0] = make_float2(z.x + z.y, z.x - z.y);
        }
    }
}

// In place on LDS, natural layout (device-function form; RC:269-344).  `hermitian_thread`: this thread takes part in
// the split / merge pass (false for the repeated threads of the reference-shaped form: the pass reads and rewrites
// the same cells, so exactly one thread may own each pair).
template <int L, int DIR, bool PAD>
__device__ __forceinline__ void r2c_c2r_lds_inplace(float2* s, const Engine<L, DIR, 1, PAD>& eng, int stride = Geometry<L, PAD>::SF, bool hermitian_thread = true) {
    using G = Geometry<L, PAD>;
    float2* sf = s + eng.fft * stride;
    if (DIR == 0) {
        fft_lds_inplace(s, eng, stride);
        fft_sync<G::kMultiWave>();
        if (hermitian_thread) hermitian_pass<L, 0>(sf, eng.u);
    } else {
        if (hermitian_thread) hermitian_pass<L, 1>(sf, eng.u);
        fft_sync<G::kMultiWave>();
        fft_lds_inplace(s, eng, stride);
    }
}

// Engine set up for a thread of a reference-shaped block (see the header comment).  Returns false for the threads
// that have nothing to do (the waves after the first one, N <= 1024).  NF = FFTs the block holds, packed contiguously.
template <int N, int DIR, int REORDER, bool PAD, int NF>
__device__ __forceinline__ bool reference_shape_init(Engine<N, DIR, REORDER, PAD>& eng) {
    using G = Geometry<N, PAD>;
    if constexpr (G::kMultiWave) {
        eng.init((int)(threadIdx.x % G::T));
    } else {
        if (threadIdx.x >= 64) return false;     // wave-uniform
        eng.init((int)threadIdx.x);
        eng.fft %= NF;
    }
    return true;
}

namespace tiled {

// the tiled contract is 256 threads per workgroup: anything else would index other FFTs' regions
__device__ __forceinline__ void require_tiled_block() {
    if (blockDim.x != 256) __builtin_trap();
}

template <class const_params>
__device__ void do_SMFFT_CT_DIT(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_size, const_params::fft_direction, const_params::fft_reorder> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
// Stockham C2C program: un-normalised INVERSE (+i) transform, natural order (ST:76, :429).
template <class const_params>
__device__ void do_FFT_Stockham_mk6(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, 1, 1> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, const_direction::fft_direction, 1> eng;
    eng.init(threadIdx.x);
    fft_lds_inplace(s_input, eng);
}
template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_R2C_C2R(float2* s_input) {
    require_tiled_block();
    Engine<const_params::fft_length, const_direction::fft_direction, 1> eng;
    eng.init(threadIdx.x);
    r2c_c2r_lds_inplace<const_params::fft_length, const_direction::fft_direction, true>(s_input, eng);
}

}  // namespace tiled
}  // namespace smfft

// =================================================================================================
// (1) the reference's contract
// =================================================================================================
template <class const_params>
__device__ void do_SMFFT_CT_DIT(float2* s_input) {
    constexpr int N = const_params::fft_size;
    smfft::Engine<N, const_params::fft_direction, const_params::fft_reorder> eng;
    if (!smfft::reference_shape_init<N, const_params::fft_direction, const_params::fft_reorder, true, const_params::fft_length / N>(eng)) return;
    smfft::fft_lds_inplace(s_input, eng, N);
}

template <class const_params>
__device__ void do_FFT_Stockham_mk6(float2* s_input) {
    constexpr int N = const_params::fft_length;
    smfft::Engine<N, 1, 1, false> eng;
    if (smfft::reference_shape_init<N, 1, 1, false, 1>(eng)) smfft::fft_lds_inplace(s_input, eng, N);
    __syncthreads();   // upstream's function ends with a barrier (ST:239) and its kernels store right after the call (ST:253)
}

template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_C2C(float2* s_input) {
    constexpr int N = const_params::fft_length;
    smfft::Engine<N, const_direction::fft_direction, 1, false> eng;
    if (smfft::reference_shape_init<N, const_direction::fft_direction, 1, false, 1>(eng)) smfft::fft_lds_inplace(s_input, eng, N);
    __syncthreads();   // upstream's function ends with a barrier (RC:265) and its callers rely on it (RC:360-361)
}

template <class const_params, class const_direction>
__device__ void do_FFT_Stockham_R2C_C2R(float2* s_input) {
    constexpr int L = const_params::fft_length;
    smfft::Engine<L, const_direction::fft_direction, 1, false> eng;
    if (smfft::reference_shape_init<L, const_direction::fft_direction, 1, false, 1>(eng))
        smfft::r2c_c2r_lds_inplace<L, const_direction::fft_direction, false>(s_input, eng, L, threadIdx.x < L / 16);
    __syncthreads();   // as upstream: the forward branch ends behind a barrier (RC:330), the inverse one in do_FFT_Stockham_C2C
}

// ---- kernels in the reference's launch shape (own text; same loads, stores and barriers as CT:534-572) ----
template <class const_params>
__global__ void SMFFT_DIT_external(float2* d_input, float2* d_output) {
    __shared__ float2 s_input[const_params::fft_sm_required];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    s_input[threadIdx.x] = d_input[base];
    s_input[threadIdx.x + const_params::fft_length_quarter] = d_input[base + const_params::fft_length_quarter];
    s_input[threadIdx.x + const_params::fft_length_half] = d_input[base + const_params::fft_length_half];
    s_input[threadIdx.x + const_params::fft_length_three_quarters] = d_input[base + const_params::fft_length_three_quarters];
    __syncthreads();
    do_SMFFT_CT_DIT<const_params>(s_input);
    __syncthreads();
    d_output[base] = s_input[threadIdx.x];
    d_output[base + const_params::fft_length_quarter] = s_input[threadIdx.x + const_params::fft_length_quarter];
    d_output[base + const_params::fft_length_half] = s_input[threadIdx.x + const_params::fft_length_half];
    d_output[base + const_params::fft_length_three_quarters] = s_input[threadIdx.x + const_params::fft_length_three_quarters];
}

template <class const_params>
__global__ void SMFFT_DIT_multiple(float2* d_input, float2* d_output) {
    __shared__ float2 s_input[const_params::fft_sm_required];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
    s_input[threadIdx.x] = d_input[base];
    s_input[threadIdx.x + const_params::fft_length_quarter] = d_input[base + const_params::fft_length_quarter];
    s_input[threadIdx.x + const_params::fft_length_half] = d_input[base + const_params::fft_length_half];
    s_input[threadIdx.x + const_params::fft_length_three_quarters] = d_input[base + const_params::fft_length_three_quarters];
    __syncthreads();
    for (int f = 0; f < NREUSES; f++) {
        do_SMFFT_CT_DIT<const_params>(s_input);
        __syncthreads();   // the reference has none here (latent race, CT:563-565)
    }
    d_output[base] = s_input[threadIdx.x];
    d_output[base + const_params::fft_length_quarter] = s_input[threadIdx.x + const_params::fft_length_quarter];
    d_output[base + const_params::fft_length_half] = s_input[threadIdx.x + const_params::fft_length_half];
    d_output[base + const_params::fft_length_three_quarters] = s_input[threadIdx.x + const_params::fft_length_three_quarters];
}

// Stockham C2C program, ST:243-278: dynamic LDS of FFT_size * 8 bytes (ST:319), blockDim.x = N / 4
template <class const_params>
__global__ void FFT_GPU_external(float2* d_input, float2* d_output) {
    extern __shared__ float2 s_input_dynamic[];
    float2* s_input = s_input_dynamic;
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    do_FFT_Stockham_mk6<const_params>(s_input);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

template <class const_params>
__global__ void FFT_GPU_multiple(float2* d_input, float2* d_output) {   // ST:260-278
    extern __shared__ float2 s_input_dynamic[];
    float2* s_input = s_input_dynamic;
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    for (int f = 0; f < NREUSES; f++) do_FFT_Stockham_mk6<const_params>(s_input);   // the function ends with a barrier
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

// R2C / C2R program, RC:349-365: L + 1 float2 of LDS, blockDim.x = L / 4, L = const_params::fft_length = real length / 2
template <class const_params, class const_direction>
__global__ void FFT_GPU_R2C_C2R_external(float2* d_input, float2* d_output) {
    __shared__ float2 s_input[const_params::fft_length + 1];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    do_FFT_Stockham_R2C_C2R<const_params, const_direction>(s_input);
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}

template <class const_params, class const_direction>
__global__ void FFT_GPU_R2C_C2R_multiple(float2* d_input, float2* d_output) {   // RC:367-384
    __shared__ float2 s_input[const_params::fft_length + 1];
    const int base = threadIdx.x + blockIdx.x * const_params::fft_length;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_input[threadIdx.x + k * const_params::fft_quarter] = d_input[base + k * const_params::fft_quarter];
    __syncthreads();
    for (int f = 0; f < NREUSES; f++) do_FFT_Stockham_R2C_C2R<const_params, const_direction>(s_input);   // ends with a barrier
#pragma unroll
    for (int k = 0; k < 4; ++k) d_output[base + k * const_params::fft_quarter] = s_input[threadIdx.x + k * const_params::fft_quarter];
}
