// valu_forms.hip -- SIMD cycles per wave64 VALU instruction by operand form, every SIMD of the chip loaded with W waves that
// execute nothing else (effective clock from tools/microbench/clock_ratio: 2.3 GHz):
//   fma1: a = a * s + 1.0          one VGPR source (the other two: SGPR, inline constant)
//   add2: a = a + b                two VGPR sources
//   mul2: a = a * b                two VGPR sources
//   fma2: a = a * s + b            two VGPR sources + SGPR
//   fma3: a = a * b + c            three VGPR sources
// Build: hipcc -O3 --offload-arch=gfx950 valu_forms.hip -o valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int FORM>
__global__ void __launch_bounds__(64) forms(float* out, int reps, float k) {
    float a[16], b[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { a[c] = (float)(threadIdx.x + c); b[c] = 1.0f + 1e-6f * (float)(threadIdx.x * 16 + c); }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (FORM == 0) a[c] = __builtin_fmaf(a[c], k, 1.0f);
                if (FORM == 1) a[c] = a[c] + b[(c + u) & 15];
                if (FORM == 2) a[c] = a[c] * b[(c + u) & 15];
                if (FORM == 3) a[c] = __builtin_fmaf(a[c], k, b[(c + u) & 15]);
                if (FORM == 4) a[c] = __builtin_fmaf(a[c], b[(c + u) & 15], b[(c + u + 7) & 15]);
                if (FORM == 5) a[c] = __builtin_fmaf(a[c], 0.92387953f, b[(c + u) & 15]);          // v_fmamk_f32: literal multiplier
                if (FORM == 6) a[c] = a[c] * 0.92387953f;                                            // v_mul_f32 with a literal
                if (FORM == 7) a[c] = __builtin_fmaf(a[c], b[(c + u) & 15], 0.92387953f);            // v_fmaak_f32: literal addend
                if (FORM == 8) a[c] = __builtin_fmaf(b[(c + u) & 15], b[(c + u + 3) & 15], a[c]);    // v_fmac_f32: accumulate in place
                if (FORM == 9) a[c] = a[c] * k;                                                      // v_mul_f32 with an SGPR
                if (FORM == 10) a[c] = a[c] * 0.5f + 0.0f * b[c];                                    // v_mul_f32 with an inline constant
            }
            if (FORM == 11) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    u2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c]), __float_as_uint(a[c + 8]), false, false);
                    a[c] = __uint_as_float(x[0]);
                    a[c + 8] = __uint_as_float(x[1]);
                }
            }
            if (FORM == 13) {        // the DPP lane-bit-2 swap of two registers (row_shr:4 / row_shl:4 with bank masks): 2 v_mov_b32_dpp
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int x = __float_as_int(a[c]), y = __float_as_int(a[c + 8]);
                    const int nx = __builtin_amdgcn_update_dpp(x, y, 0x114, 0xF, 0xA, false);
                    const int ny = __builtin_amdgcn_update_dpp(y, x, 0x104, 0xF, 0x5, false);
                    a[c] = __int_as_float(nx);
                    a[c + 8] = __int_as_float(ny);
                }
            }
            if (FORM == 12) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    u2 x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[c]), __float_as_uint(a[c + 8]), false, false);
                    a[c] = __uint_as_float(x[0]);
                    a[c + 8] = __uint_as_float(x[1]);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += a[c] + b[c];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 20000;
    const double insts = (double)reps * 128.0;
    const char* names[14] = {"fma1 a=a*s+1", "add2 a=a+b", "mul2 a=a*b", "fma2 a=a*s+b", "fma3 a=a*b+c", "fmamk a=a*K+b", "mul  a=a*K", "fmaak a=a*b+K", "fmac a+=b*c", "mul  a=a*s", "mul a=a*0.5", "permlane32_swap of a register pair", "permlane16_swap of a register pair", "dpp bit-2 swap of a register pair"};
    for (int form = 0; form < 14; ++form)
        for (int w : {1, 4}) {
            std::vector<float> t;
            for (int i = 0; i < 5; ++i) {
                hipEventRecord(e0, 0);
                if (form == 0) forms<0><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 1) forms<1><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 2) forms<2><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 3) forms<3><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 4) forms<4><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 5) forms<5><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 6) forms<6><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 7) forms<7><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 8) forms<8><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 9) forms<9><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 10) forms<10><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 11) forms<11><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 12) forms<12><<<1024 * w, 64>>>(out, reps, 0.9999f);
                if (form == 13) forms<13><<<1024 * w, 64>>>(out, reps, 0.9999f);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("%-22s %d wave(s) per SIMD: %.3f ms -> %.2f SIMD cycles per instruction at 2.3 GHz\n", names[form], w, t[2], t[2] * 1e-3 * 2.3e9 / ((form >= 11 ? insts / 2.0 : insts) * w));
        }
    return 0;
}
