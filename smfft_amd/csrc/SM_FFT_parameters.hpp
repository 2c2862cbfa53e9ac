// SM_FFT_parameters.hpp -- compile-time descriptions of the Cooley-Tukey C2C transform variants.
//
// Same class names and members as the reference's parameter header
// (SMFFT_CooleyTukey_C2C/SM_FFT_parameters.cuh:1-390: FFT_Params and the 32 classes
// FFT_<N>_{forward,inverse}{,_noreorder}), so code templated on them keeps compiling; the VALUES
// describe the gfx950 engine (smfft_engine.hpp), not the CUDA one:
//   warp                      64 (a CDNA wavefront; the reference has 32)
//   fft_exp                   log2 of the transform length
//   fft_size                  the transform length N                              (new)
//   fft_length                float2 elements one workgroup processes = 4096 for every N
//                             (the reference uses max(N,128): 4x32, 2x64 or one FFT per block)
//   fft_length_quarter/half/three_quarters   fractions of fft_length (kept for source parity)
//   fft_sm_required           float2 elements of LDS one workgroup needs = 4352 (= 4096*17/16:
//                             every FFT owns a region of 17N/16 float2 that holds its natural-order
//                             data and the padded exchange layouts; reference: (N/32)*33)
//   fft_direction             0 forward (e^-), 1 inverse (e^+); both un-normalised
//   fft_reorder               1: out = DFT(in); 0: out = DFT(in[bitrev(n)]) (DIT network on natural input)
//   fft_threads               threads that cooperate on one FFT = N/16                (new)
//   fft_per_block             FFTs one 256-thread workgroup holds = 4096/N            (new)
//   fft_region                float2 stride between consecutive FFTs of a block in LDS = 17N/16 (new)
// Deviation, documented: the reference's FFT_4096_inverse_noreorder has fft_direction = 0
// (SM_FFT_parameters.cuh:388, a typo that silently computes the forward transform); here it is 1.
#pragma once

class FFT_Params {
public:
	static const int fft_exp = -1;
	static const int fft_length = -1;
	static const int warp = 64;
};

template<int EXP, int DIRECTION, int REORDER>
class FFT_ParamsOf : public FFT_Params {
public:
	static const int fft_exp = EXP;
	static const int fft_size = 1 << EXP;
	static const int fft_length = 4096;
	static const int fft_length_quarter = 1024;
	static const int fft_length_half = 2048;
	static const int fft_length_three_quarters = 3072;
	static const int fft_sm_required = 4352;
	static const int fft_direction = DIRECTION;
	static const int fft_reorder = REORDER;
	static const int fft_threads = (1 << EXP) / 16;
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
