// smfft_device.hpp -- the device-side surface of smfft_amd for user kernels (header only, gfx950).
//
//   #include <smfft_device.hpp>          hipcc --offload-arch=gfx950 -std=c++17 -I<repo>/include
//
// What it provides (details in the headers it includes):
//   smfft/SM_FFT_parameters.hpp           FFT_Params + the 32 FFT_<N>_{forward,inverse}{,_noreorder} classes (CT)
//   smfft/SM_FFT_stockham_parameters.hpp  FFT_<N>, FFT_forward, FFT_inverse (Stockham and R2C/C2R programs)
//   smfft/smfft_device_functions.hpp      do_SMFFT_CT_DIT<P>, do_FFT_Stockham_mk6<P>, do_FFT_Stockham_C2C<P,D>,
//                                         do_FFT_Stockham_R2C_C2R<P,D> in the reference's own contract
//                                         (blockDim.x = fft_length / 4, contiguous data), the reference-shaped kernels
//                                         SMFFT_DIT_external<P>(in, out) etc., and the same functions in the engine's
//                                         tiled contract (namespace smfft::tiled)
//   smfft/smfft_engine.hpp                smfft::Engine<N, DIR, REORDER>: registers in, registers out
// Reference: SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu:334-551, README.md:10-60.
#pragma once
#include "smfft/SM_FFT_parameters.hpp"
#include "smfft/SM_FFT_stockham_parameters.hpp"
#include "smfft/smfft_engine.hpp"
#include "smfft/smfft_device_functions.hpp"
