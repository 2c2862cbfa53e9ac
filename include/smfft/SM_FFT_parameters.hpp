// SM_FFT_parameters.hpp -- compile-time descriptions of the Cooley-Tukey C2C transform variants.
//
// Same class names and members as the reference's parameter header
// (SMFFT_CooleyTukey_C2C/SM_FFT_parameters.cuh:1-390: FFT_Params and the 32 classes
// FFT_<N>_{forward,inverse}{,_noreorder}).  The members the reference defines keep the reference's
// MEANING, so a kernel written against them (README.md:48-60, CT:534-551) compiles unchanged:
//   fft_exp                   log2 of the transform length
//   fft_length                float2 elements one thread block of the reference-shaped kernels holds:
//                             max(N, 128) -- 4 x 32, 2 x 64 or one FFT, exactly as upstream (:8-18, :56-66)
//   fft_length_quarter/half/three_quarters   fractions of fft_length; blockDim.x = fft_length_quarter
//   fft_sm_required           float2 elements of LDS the caller declares for do_SMFFT_CT_DIT:
//                             17 * fft_length / 16 (the data in [0, fft_length), the rest is the engine's
//                             padded exchange space; upstream: (fft_length / 32) * 33, README.md:18)
//   fft_direction             0 forward (e^-), 1 inverse (e^+); both un-normalised
//   fft_reorder               1: out = DFT(in); 0: out = DFT(in[bitrev(n)]) (DIT network on natural input)
//   warp                      64: a CDNA wavefront (upstream 32)
// Members that describe the gfx950 engine (include/smfft/smfft_engine.hpp) and the library's own
// tiled kernels; none of them exists upstream:
//   fft_size                  the transform length N
//   fft_threads               threads that cooperate on one FFT = N / 16
//   tile_length               float2 elements one 256-thread workgroup of the tiled kernels owns = 4096
//   tile_sm_required          LDS float2 of such a workgroup = 4352
//   fft_per_block             FFTs in a tile = 4096 / N
//   fft_region                float2 stride between consecutive FFTs of a tile in LDS = 17 N / 16
// WAVE64-FULL SMALL LENGTHS (an extension).  Upstream gives N = 32 / 64 / 128 a block of 32 threads holding 128
// elements (SM_FFT_parameters.cuh:8-18, CT:586-595) -- one CUDA warp, but HALF a CDNA wavefront.  The classes
//   FFT_{32,64,128}_{forward,inverse}{,_noreorder}_wave64
// describe the same transforms with fft_length = 256: blockDim.x = fft_length_quarter = 64 = one full wave holding 8 x 32,
// 4 x 64 or 2 x 128 elements, everything else (member names, do_SMFFT_CT_DIT<P>(s), the two-argument kernels) unchanged.
// A batch whose size is not a multiple of 256 / N transforms runs its last (nFFTs mod 256/N) transforms on the upstream-shaped
// class (examples/reference_shape_kernel.hip, launch_ct_wave64).  -DSMFFT_WAVE64_SMALL=1 gives the UPSTREAM class names the
// 64-thread shape too (for a code base that derives its launch shape from the class, as CT:586-595 does); the default keeps
// upstream's values (tests/test_abi_and_host.py).
// Deviation, documented: the reference's FFT_4096_inverse_noreorder has fft_direction = 0
// (SM_FFT_parameters.cuh:388, a typo that silently computes the forward transform); here it is 1
// (tests/test_lane_emulation.py shows both behaviours).
#pragma once

class FFT_Params {
public:
	static const int fft_exp = -1;
	static const int fft_length = -1;
	static const int warp = 64;
};

#ifndef SMFFT_WAVE64_SMALL
#define SMFFT_WAVE64_SMALL 0
#endif
#define SMFFT_SMALL_BLOCK_LENGTH (SMFFT_WAVE64_SMALL ? 256 : 128)

template<int EXP, int DIRECTION, int REORDER, int MIN_BLOCK_LENGTH = SMFFT_SMALL_BLOCK_LENGTH>
class FFT_ParamsOf : public FFT_Params {
public:
	// ---- the reference's members ----
	static const int fft_exp = EXP;
	static const int fft_length = (1 << EXP) < MIN_BLOCK_LENGTH ? MIN_BLOCK_LENGTH : (1 << EXP);
	static const int fft_length_quarter = fft_length / 4;
	static const int fft_length_half = fft_length / 2;
	static const int fft_length_three_quarters = 3 * (fft_length / 4);
	static const int fft_sm_required = (fft_length / 16) * 17;
	static const int fft_direction = DIRECTION;
	static const int fft_reorder = REORDER;
	// ---- the engine's ----
	static const int fft_size = 1 << EXP;
	static const int fft_threads = (1 << EXP) / 16;
	static const int tile_length = 4096;
	static const int tile_sm_required = 4352;
	static const int fft_per_block = 4096 >> EXP;
	static const int fft_region = ((1 << EXP) / 16) * 17;
};

class FFT_32_forward : public FFT_ParamsOf<5, 0, 1> {};
class FFT_32_forward_noreorder : public FFT_ParamsOf<5, 0, 0> {};
class FFT_32_inverse : public FFT_ParamsOf<5, 1, 1> {};
class FFT_32_inverse_noreorder : public FFT_ParamsOf<5, 1, 0> {};

class FFT_64_forward : public FFT_ParamsOf<6, 0, 1> {};
class FFT_64_forward_noreorder : public FFT_ParamsOf<6, 0, 0> {};
class FFT_64_inverse : public FFT_ParamsOf<6, 1, 1> {};
class FFT_64_inverse_noreorder : public FFT_ParamsOf<6, 1, 0> {};

class FFT_128_forward : public FFT_ParamsOf<7, 0, 1> {};
class FFT_128_forward_noreorder : public FFT_ParamsOf<7, 0, 0> {};
class FFT_128_inverse : public FFT_ParamsOf<7, 1, 1> {};
class FFT_128_inverse_noreorder : public FFT_ParamsOf<7, 1, 0> {};

// one full 64-lane wave per block (see the header comment)
class FFT_32_forward_wave64 : public FFT_ParamsOf<5, 0, 1, 256> {};
class FFT_32_forward_noreorder_wave64 : public FFT_ParamsOf<5, 0, 0, 256> {};
class FFT_32_inverse_wave64 : public FFT_ParamsOf<5, 1, 1, 256> {};
class FFT_32_inverse_noreorder_wave64 : public FFT_ParamsOf<5, 1, 0, 256> {};
class FFT_64_forward_wave64 : public FFT_ParamsOf<6, 0, 1, 256> {};
class FFT_64_forward_noreorder_wave64 : public FFT_ParamsOf<6, 0, 0, 256> {};
class FFT_64_inverse_wave64 : public FFT_ParamsOf<6, 1, 1, 256> {};
class FFT_64_inverse_noreorder_wave64 : public FFT_ParamsOf<6, 1, 0, 256> {};
class FFT_128_forward_wave64 : public FFT_ParamsOf<7, 0, 1, 256> {};
class FFT_128_forward_noreorder_wave64 : public FFT_ParamsOf<7, 0, 0, 256> {};
class FFT_128_inverse_wave64 : public FFT_ParamsOf<7, 1, 1, 256> {};
class FFT_128_inverse_noreorder_wave64 : public FFT_ParamsOf<7, 1, 0, 256> {};

class FFT_256_forward : public FFT_ParamsOf<8, 0, 1> {};
class FFT_256_forward_noreorder : public FFT_ParamsOf<8, 0, 0> {};
class FFT_256_inverse : public FFT_ParamsOf<8, 1, 1> {};
class FFT_256_inverse_noreorder : public FFT_ParamsOf<8, 1, 0> {};

class FFT_512_forward : public FFT_ParamsOf<9, 0, 1> {};
class FFT_512_forward_noreorder : public FFT_ParamsOf<9, 0, 0> {};
class FFT_512_inverse : public FFT_ParamsOf<9, 1, 1> {};
class FFT_512_inverse_noreorder : public FFT_ParamsOf<9, 1, 0> {};

class FFT_1024_forward : public FFT_ParamsOf<10, 0, 1> {};
class FFT_1024_forward_noreorder : public FFT_ParamsOf<10, 0, 0> {};
class FFT_1024_inverse : public FFT_ParamsOf<10, 1, 1> {};
class FFT_1024_inverse_noreorder : public FFT_ParamsOf<10, 1, 0> {};

class FFT_2048_forward : public FFT_ParamsOf<11, 0, 1> {};
class FFT_2048_forward_noreorder : public FFT_ParamsOf<11, 0, 0> {};
class FFT_2048_inverse : public FFT_ParamsOf<11, 1, 1> {};
class FFT_2048_inverse_noreorder : public FFT_ParamsOf<11, 1, 0> {};

class FFT_4096_forward : public FFT_ParamsOf<12, 0, 1> {};
class FFT_4096_forward_noreorder : public FFT_ParamsOf<12, 0, 0> {};
class FFT_4096_inverse : public FFT_ParamsOf<12, 1, 1> {};
class FFT_4096_inverse_noreorder : public FFT_ParamsOf<12, 1, 0> {};
