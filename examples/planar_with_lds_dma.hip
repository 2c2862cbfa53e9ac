// planar_with_lds_dma.hip -- a user kernel that mixes the planar engine's stores (ds_write_addtid_b32, which take their base
// from M0 inside inline assembly; include/smfft/smfft_planar.hpp, addtid_store4) with the compiler's OWN users of M0: LDS-DMA
// loads (global_load_lds_dword: LDS address = M0 base + instruction offset + 4 * lane).  hipcc may set M0 once for several
// LDS-DMA instructions; the planar store block in between writes M0 -- and restores it.  The kernel's result is only right if
// every instruction saw the M0 it was compiled for (tests/test_gpu_parity.py::test_planar_stores_preserve_m0).
//
//   out[0 .. 64)      = in[0 .. 64)       via LDS-DMA into row A                (before the planar stores)
//   out[64 .. 128)    = in[64 .. 128)     via LDS-DMA into row A + 64 dwords    (after them: same M0 base, other offset)
//   out[128 .. 640)   = the planar stores: four float2 per lane -> eight rows of 64 dwords (rows 0-3 real, 4-7 imaginary)
#include <hip/hip_runtime.h>
#include <smfft/smfft_planar.hpp>

__global__ void __launch_bounds__(64) planar_with_lds_dma_kernel(const float* in, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[128 + 512];
    typedef __attribute__((address_space(3))) float lds_float;
    typedef __attribute__((address_space(1))) const float global_float;
    const int lane = threadIdx.x;
    lds_float* dma_row = (lds_float*)lds;
    // first LDS-DMA load: M0 = base of dma_row
    __builtin_amdgcn_global_load_lds((global_float*)(in + lane), dma_row, 4, 0, 0);
    // planar stores of lane-dependent values into lds[128 ...): M0 = that base inside the block
    const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_float*)(lds + 128));
    const float f = (float)lane;
    smfft::addtid_store4<0, 64, 128, 192, 256>(m0, make_float2(f, -f), make_float2(f + 100.f, -f - 100.f), make_float2(f + 200.f, -f - 200.f),
                                                 make_float2(f + 300.f, -f - 300.f));
    // second LDS-DMA load, same base, instruction offset 256 bytes (applied to the global AND the LDS address): the compiler
    // re-uses its M0 set-up of the first one (gfx950 ISA of this file: one s_mov_b32 m0 of its own for both loads)
    __builtin_amdgcn_global_load_lds((global_float*)(in + lane), dma_row, 4, 256, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 10; ++k) out[lane + 64 * k] = lds[lane + 64 * k];
}

extern "C" int smfft_example_planar_with_lds_dma(const void* d_in, void* d_out, void* stream) {
    planar_with_lds_dma_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>((const float*)d_in, (float*)d_out);
    return (int)hipGetLastError();
}
