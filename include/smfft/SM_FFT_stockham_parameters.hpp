// SM_FFT_stockham_parameters.hpp -- parameter classes of the Stockham C2C and R2C/C2R programs.
// Same names and members as the reference (SMFFT_Stockham_C2C/FFT-GPU-32bit-Stockham.cu:15-67 and
// SMFFT_Stockham_R2C_C2R/FFT-GPU-32bit-Stockham.cu:15-81): FFT_ConstParams, FFT_256 ... FFT_4096,
// FFT_ConstDirection, FFT_forward, FFT_inverse.  `warp` is 64 on gfx950.  For the R2C/C2R
// functions fft_length is the COMPLEX length L = N_real/2, exactly as upstream
// (FFT_external_benchmark maps real 2048 -> FFT_1024, RC:420-423).
#pragma once

class FFT_ConstParams {
public:
	static const int fft_exp = -1;
	static const int fft_length = -1;
	static const int fft_half = -1;
	static const int warp = 64;
};

template<int EXP>
class FFT_ConstParamsOf : public FFT_ConstParams {
public:
	static const int fft_exp = EXP;
	static const int fft_length = 1 << EXP;
	static const int fft_quarter = (1 << EXP) / 4;
	static const int fft_half = (1 << EXP) / 2;
	static const int fft_threequarters = 3 * ((1 << EXP) / 4);
};

// FFT_32 ... FFT_128 do not exist upstream (its Stockham kernels start at 256, ST:316-338); the
// engine here covers them, so the Stockham API accepts N = 32 ... 4096.
class FFT_32   : public FFT_ConstParamsOf<5>  {};
class FFT_64   : public FFT_ConstParamsOf<6>  {};
class FFT_128  : public FFT_ConstParamsOf<7>  {};
class FFT_256  : public FFT_ConstParamsOf<8>  {};
class FFT_512  : public FFT_ConstParamsOf<9>  {};
class FFT_1024 : public FFT_ConstParamsOf<10> {};
class FFT_2048 : public FFT_ConstParamsOf<11> {};
class FFT_4096 : public FFT_ConstParamsOf<12> {};

class FFT_ConstDirection {
public:
	static const int fft_direction = -1;
};
class FFT_forward : public FFT_ConstDirection {
public:
	static const int fft_direction = 0;
};
class FFT_inverse : public FFT_ConstDirection {
public:
	static const int fft_direction = 1;
};
