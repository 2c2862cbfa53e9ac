// lds_forms.hip -- what one exchange of a wave's 1024 float2 through LDS costs, by instruction form, with and without
// arithmetic beside it.  One wave per workgroup with an 8.5 KiB image (18 workgroups per CU, as the in-LDS kernels).
//   write forms   W0: 16 ds_write_b64 (array of float2, s[lane + 64 c])                              6 cycles each (guide)
//                 W1: 32 ds_write_addtid_b32 (two planes of dwords, address = M0 + offset + 4 * lane)  2 cycles each
//   read forms    R0: 16 ds_read_b64, sixteen contiguous float2 of the lane's padded row (stride 17)   2 cycles each
//                 R1:  8 ds_read_b128, sixteen contiguous dwords of each plane                          4 cycles each
//                 R2: 32 ds_read_b32 from the planes (plane[lane + 64 c]: what a natural re-load would be)
//                 R3: 32 ds_read_addtid_b32
// Also checks that the addtid forms address what the guide says (the read-back of pattern data).
// Build: hipcc -O3 --offload-arch=gfx950 lds_forms.hip -o lds_forms
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) float lds_float;

template <int ROUNDS>
__device__ __forceinline__ void valu_block(float (&x)[32], float k) {
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j)
#pragma unroll
        for (int c = 0; c < 32; ++c) x[c] = __builtin_fmaf(x[c], k, x[(c + 3 + 2 * j) & 31]);
}

__device__ __forceinline__ unsigned rev_bits(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

#define PLANE 1088   // dwords per plane: 16 rows of 64 + 4 pad per row (68 c + lane)

template <int W, int R, int ROUNDS>
__global__ void __launch_bounds__(64) forms(float* out, int reps, float k, int check) {
    __shared__ float s[2 * PLANE];
    const int lane = threadIdx.x;
    const unsigned sbase = (unsigned)(unsigned long)(lds_float*)s;
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = (float)(lane + 64 * c);
    for (int it = 0; it < reps; ++it) {
        // ---- write ----
        if (W == 0) {
            const unsigned a = sbase + 8 * (lane + (lane >> 4));
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(*(double*)&x[2 * c]), "n"(8 * 64 * c + 32 * c) : "memory");   // p + p/16 pads
        } else {
            asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(sbase)) : "memory");
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[2 * c]), "n"(4 * 68 * c) : "memory");
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x[2 * c + 1]), "n"(4 * (PLANE + 68 * c)) : "memory");
            }
        }
        // ---- read ----
        if (R == 0) {
            const unsigned a = sbase + 8 * 17 * lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(double*)&x[2 * c]) : "v"(a), "n"(8 * c) : "memory");
        } else if (R == 1) {
            // row c' = rev4(lane & 15), dwords [16 * rev2(lane >> 4), +16): conflict free for the 16-lane groups of ds_read_b128
            const unsigned a = sbase + 4 * (68 * rev_bits(lane & 15, 4) + 16 * rev_bits(lane >> 4, 2));
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 re, im;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(re) : "v"(a), "n"(16 * q) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(im) : "v"(a), "n"(4 * PLANE + 16 * q) : "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) { x[2 * (4 * q + i)] = re[i]; x[2 * (4 * q + i) + 1] = im[i]; }
            }
        } else if (R == 2) {
            const unsigned a = sbase + 4 * lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[2 * c]) : "v"(a), "n"(4 * 68 * c) : "memory");
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[2 * c + 1]) : "v"(a), "n"(4 * (PLANE + 68 * c)) : "memory");
            }
        } else {
            asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(sbase)) : "memory");
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                asm volatile("ds_read_addtid_b32 %0 offset:%1" : "=v"(x[2 * c]) : "n"(4 * 68 * c) : "memory");
                asm volatile("ds_read_addtid_b32 %0 offset:%1" : "=v"(x[2 * c + 1]) : "n"(4 * (PLANE + 68 * c)) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (check && it == 0) {
#pragma unroll
            for (int c = 0; c < 32; ++c) out[(blockIdx.x * 32 + c) * 64 + lane] = x[c];
        }
        valu_block<ROUNDS>(x, k);
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) acc += x[c];
    if (acc == 1234.5f) out[lane] = acc;
}

static hipEvent_t e0, e1;
template <class K>
static float run(K kern, int blocks, float* out) {
    std::vector<float> t;
    for (int i = 0; i < 7; ++i) {
        hipEventRecord(e0, 0);
        kern<<<blocks, 64>>>(out, 100, 0.9999f, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int W, int R, int ROUNDS>
static void sweep(const char* name, float* out) {
    printf("%-66s", name);
    for (int k16 : {16, 32, 64, 72, 288}) {
        const float ms = run(forms<W, R, ROUNDS>, 64 * k16, out);
        printf(" %5.2f w/SIMD %.3f us", k16 / 16.0, ms * 1e3 / (100.0 * k16 / 16.0));
    }
    printf("   (per exchange per SIMD)\n");
    fflush(stdout);
}

// what lane `lane` should read back in register c (dword index 2 * element + part) after one write + read
template <int W, int R>
static void verify(float* out) {
    std::vector<float> h(32 * 64);
    forms<W, R, 0><<<1, 64>>>(out, 1, 1.0f, 1);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int c = 0; c < 16; ++c)
            for (int part = 0; part < 2; ++part) {
                // element written: (writer lane wl, register wc) holds value wl + 64 * (2 wc + part)
                int wl, wc;
                if (R == 0) { int p = 16 * lane + c; wl = p & 63; wc = p >> 6; }                     // row `lane` of the padded image
                else if (R == 1) {
                    auto rev = [](int v, int b) { int r = 0; for (int i = 0; i < b; ++i) r |= ((v >> i) & 1) << (b - 1 - i); return r; };
                    wc = rev(lane & 15, 4); wl = 16 * rev(lane >> 4, 2) + c;
                } else { wl = lane; wc = c; }
                const float want = (float)(wl + 64 * (2 * wc + part));
                if (h[(2 * c + part) * 64 + lane] != want) { if (bad < 8) printf("  lane %d c %d part %d: got %g want %g\n", lane, c, part, h[(2 * c + part) * 64 + lane], want); ++bad; }
            }
    printf("verify W%d R%d: %s (%d mismatches)\n", W, R, bad ? "MISMATCH" : "ok", bad);
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    verify<0, 0>(out);
    verify<1, 1>(out);
    verify<1, 2>(out);
    verify<1, 3>(out);
    sweep<0, 0, 0>("W0 16 ds_write_b64 + R0 16 ds_read_b64, no VALU", out);
    sweep<1, 1, 0>("W1 32 ds_write_addtid_b32 + R1 8 ds_read_b128, no VALU", out);
    sweep<1, 2, 0>("W1 32 ds_write_addtid_b32 + R2 32 ds_read_b32, no VALU", out);
    sweep<1, 3, 0>("W1 32 ds_write_addtid_b32 + R3 32 ds_read_addtid_b32, no VALU", out);
    sweep<0, 0, 8>("W0 + R0, 256 VALU", out);
    sweep<1, 1, 8>("W1 + R1, 256 VALU", out);
    sweep<1, 2, 8>("W1 + R2, 256 VALU", out);
    sweep<1, 3, 8>("W1 + R3, 256 VALU", out);
    return 0;
}
