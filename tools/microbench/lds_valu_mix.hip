// lds_valu_mix.hip -- a synthetic stand-in for one application of the in-LDS N = 1024 transform (one wave, 1024 float2 in an
// 8.5 KiB LDS image): 16 LDS reads, ~256 VALU, 16 LDS writes, 16 LDS reads, ~256 VALU, 16 LDS writes, 100 times -- to see what
// the SIMD / LDS pair sustains for this mix at 1 ... 4.5 waves per SIMD, and which changes of the STRUCTURE move it:
//   WAIT  0: every read phase waits for all sixteen values (s_waitcnt lgkmcnt(0)), as an inline-asm read block does
//         1: the compiler's counted waits (plain C++ reads)
//   R2    0: second read phase as sixteen ds_read_b64 (stride S2 apart: not mergeable); 1: as the compiler merges them
//   SKIPW 1: the last sixteen writes + first sixteen reads of the next application dropped (data forwarded in registers)
//   VALU  number of 32-instruction rounds per VALU block (8 = 256 instructions)
// Build: hipcc -O3 --offload-arch=gfx950 lds_valu_mix.hip -o lds_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int ROUNDS>
__device__ __forceinline__ void valu_block(v2f (&r)[16], float k) {
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            r[c].x = __builtin_fmaf(r[c].x, k, r[(c + 1 + j) & 15].y);
            r[c].y = __builtin_fmaf(r[c].y, k, r[(c + 5 + j) & 15].x);
        }
}

template <int WAIT, int R2, int SKIPW, int ROUNDS>
__global__ void __launch_bounds__(64) mix(v2f* out, int reps, float k) {
    __shared__ v2f s[1088];
    const int lane = threadIdx.x;
    v2f r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[lane + 64 * c] = v2f{(float)lane, (float)c};
    __builtin_amdgcn_s_waitcnt(0);
    if (SKIPW) {
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = s[lane + 64 * c];
    }
    for (int it = 0; it < reps; ++it) {
        if (!SKIPW) {
            // read phase 1: sixteen contiguous elements of a padded row (the bit-reversed read): lane * 17 + c
            const v2f* row = s + lane * 17;
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = row[c];
            if (WAIT == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        }
        valu_block<ROUNDS>(r, k);
        // write phase 1 (exchange): t-major rows of 65
#pragma unroll
        for (int c = 0; c < 16; ++c) s[(lane & 15) * 65 + (lane >> 4) * 4 + (c & 3) + 16 * (c >> 2)] = r[c];
        // read phase 2: s[lane + 65 * i]
        if (R2) {
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = s[lane + 65 * c];
        } else {
            const v2f* col = s + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) { v2f v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(unsigned long)(__attribute__((address_space(3))) const v2f*)col), "n"(65 * 8 * 0) ); r[c] = v; col += 65; }
        }
        if (WAIT == 0 || !R2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        valu_block<ROUNDS>(r, k);
        if (!SKIPW) {
            // write phase 2 (store into the padded image): p + p / 16
#pragma unroll
            for (int c = 0; c < 16; ++c) s[lane + 64 * c + 4 * c + (lane >> 4)] = r[c];
        }
    }
    v2f acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += r[c];
    if (acc.x == 1234.5f) out[lane] = acc;
}

static hipEvent_t e0, e1;
template <class K>
static float run(K kern, int blocks, v2f* out) {
    std::vector<float> t;
    for (int i = 0; i < 7; ++i) {
        hipEventRecord(e0, 0);
        kern<<<blocks, 64>>>(out, 100, 0.9999f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int WAIT, int R2, int SKIPW, int ROUNDS>
static void sweep(const char* name, v2f* out) {
    printf("%-58s", name);
    for (int k16 : {16, 32, 64, 72, 160}) {
        const float ms = run(mix<WAIT, R2, SKIPW, ROUNDS>, 64 * k16, out);
        printf(" %5.2f w/SIMD %.3f us", k16 / 16.0, ms * 1e3 / (100.0 * k16 / 16.0));
    }
    printf("   (per application per SIMD)\n");
    fflush(stdout);
}

int main() {
    v2f* out;
    hipMalloc(&out, 4096);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    sweep<0, 1, 0, 8>("wait-all, merged second reads, 2x256 VALU", out);
    sweep<1, 1, 0, 8>("counted waits, merged second reads, 2x256 VALU", out);
    sweep<0, 0, 0, 8>("wait-all, single second reads, 2x256 VALU", out);
    sweep<1, 1, 1, 8>("forwarded (no store / re-load), counted waits, 2x256 VALU", out);
    sweep<1, 1, 0, 0>("LDS only (no VALU), counted waits", out);
    sweep<1, 1, 1, 0>("LDS only, forwarded", out);
    sweep<1, 1, 0, 4>("counted waits, 2x128 VALU", out);
    sweep<1, 1, 0, 12>("counted waits, 2x384 VALU", out);
    return 0;
}
