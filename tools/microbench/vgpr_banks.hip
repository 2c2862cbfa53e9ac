// vgpr_banks.hip -- does the rate of a two-source VALU instruction depend on WHICH registers it names?  Every SIMD of the chip runs W
// waves of a loop of 64 v_add_f32 whose sources are (v[A + 4i], v[B + 4i]) -- same residue mod 4 or not -- and whose destinations
// are disjoint from the sources.  Prints SIMD cycles per instruction.
// Build: hipcc -O3 --offload-arch=gfx950 vgpr_banks.hip -o vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
template <int FORM>
__global__ void __launch_bounds__(64) k(float* out, int reps) {
    float acc = 0.f;
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n s_mov_b32 s20, 0x33333333\n s_mov_b32 s21, 0x33333333" ::: "vcc", "s20", "s21");
    for (int r = 0; r < reps; ++r) {
        if (FORM == 0)       // sources v8 / v12: same residue mod 4
            asm volatile(REP8("v_add_f32 v40, v8, v12\n v_add_f32 v41, v8, v12\n v_add_f32 v42, v8, v12\n v_add_f32 v43, v8, v12\n v_add_f32 v44, v8, v12\n v_add_f32 v45, v8, v12\n v_add_f32 v46, v8, v12\n v_add_f32 v47, v8, v12\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v12");
        if (FORM == 1)       // sources v8 / v13: different residues
            asm volatile(REP8("v_add_f32 v40, v8, v13\n v_add_f32 v41, v8, v13\n v_add_f32 v42, v8, v13\n v_add_f32 v43, v8, v13\n v_add_f32 v44, v8, v13\n v_add_f32 v45, v8, v13\n v_add_f32 v46, v8, v13\n v_add_f32 v47, v8, v13\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13");
        if (FORM == 2)       // fma with three sources of one residue
            asm volatile(REP8("v_fma_f32 v40, v8, v12, v16\n v_fma_f32 v41, v8, v12, v16\n v_fma_f32 v42, v8, v12, v16\n v_fma_f32 v43, v8, v12, v16\n v_fma_f32 v44, v8, v12, v16\n v_fma_f32 v45, v8, v12, v16\n v_fma_f32 v46, v8, v12, v16\n v_fma_f32 v47, v8, v12, v16\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v12", "v16");
        if (FORM == 3)       // fma with three sources of three residues
            asm volatile(REP8("v_fma_f32 v40, v8, v13, v18\n v_fma_f32 v41, v8, v13, v18\n v_fma_f32 v42, v8, v13, v18\n v_fma_f32 v43, v8, v13, v18\n v_fma_f32 v44, v8, v13, v18\n v_fma_f32 v45, v8, v13, v18\n v_fma_f32 v46, v8, v13, v18\n v_fma_f32 v47, v8, v13, v18\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13", "v18");
        if (FORM == 4)       // destination of the same residue as a source
            asm volatile(REP8("v_add_f32 v40, v8, v13\n v_add_f32 v44, v8, v13\n v_add_f32 v48, v8, v13\n v_add_f32 v52, v8, v13\n v_add_f32 v56, v8, v13\n v_add_f32 v60, v8, v13\n v_add_f32 v64, v8, v13\n v_add_f32 v68, v8, v13\n") ::: "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68", "v8", "v13");
        if (FORM == 5)       // a dependent chain: each instruction reads the one before
            asm volatile(REP8("v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v40, v40, v13\n") ::: "v40", "v13");
        if (FORM == 6)       // two interleaved dependent chains
            asm volatile(REP8("v_add_f32 v40, v40, v13\n v_add_f32 v41, v41, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v41, v41, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v41, v41, v13\n v_add_f32 v40, v40, v13\n v_add_f32 v41, v41, v13\n") ::: "v40", "v41", "v13");
        if (FORM == 7)       // v_fmac_f32_dpp row_ror:8, independent
            asm volatile(REP8("v_fmac_f32_dpp v40, v40, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v41, v41, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v42, v42, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v43, v43, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v44, v44, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v45, v45, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v46, v46, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v47, v47, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v13");
        if (FORM == 8)       // v_cndmask_b32_dpp quad_perm (the selects of the lane <-> register transposes)
            asm volatile(REP8("v_cndmask_b32_dpp v40, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v41, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v42, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v43, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v44, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v45, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v46, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v47, v8, v13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13", "vcc");
        if (FORM == 9)       // v_mov_b32_dpp
            asm volatile(REP8("v_mov_b32_dpp v40, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v41, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v42, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v43, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v44, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v45, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v46, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v47, v8 row_ror:8 row_mask:0xf bank_mask:0xf\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8");
        if (FORM == 10)      // v_add_f32_dpp: an addition with a DPP source
            asm volatile(REP8("v_add_f32_dpp v40, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v41, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v42, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v43, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v44, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v45, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v46, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp v47, v8, v13 row_ror:8 row_mask:0xf bank_mask:0xf\n") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13");
        if (FORM == 11) asm volatile(REP8("v_cndmask_b32 v40, v8, v13, vcc\n v_cndmask_b32 v41, v8, v13, vcc\n v_cndmask_b32 v42, v8, v13, vcc\n v_cndmask_b32 v43, v8, v13, vcc\n v_cndmask_b32 v44, v8, v13, vcc\n v_cndmask_b32 v45, v8, v13, vcc\n v_cndmask_b32 v46, v8, v13, vcc\n v_cndmask_b32 v47, v8, v13, vcc\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13", "vcc");
        if (FORM == 12) asm volatile(REP8("v_cndmask_b32_dpp v40, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v41, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v42, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v43, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v44, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v45, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v46, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v47, v8, v13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13", "vcc");
        if (FORM == 13) asm volatile(REP8("v_mov_b32_dpp v40, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v41, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v42, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v43, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v44, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v45, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v46, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v47, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8");
        if (FORM == 14) asm volatile(REP8("v_cndmask_b32_dpp v40, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v41, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v42, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v43, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v44, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v45, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v46, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v47, v8, v12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v12", "vcc");
        if (FORM == 15) asm volatile(REP8("v_fma_f32 v40, v8, v12, v17\n v_fma_f32 v41, v8, v12, v17\n v_fma_f32 v42, v8, v12, v17\n v_fma_f32 v43, v8, v12, v17\n v_fma_f32 v44, v8, v12, v17\n v_fma_f32 v45, v8, v12, v17\n v_fma_f32 v46, v8, v12, v17\n v_fma_f32 v47, v8, v12, v17\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v12", "v17");
        if (FORM == 16) asm volatile(REP8("v_fmac_f32 v40, v9, v14\n v_fmac_f32 v41, v9, v14\n v_fmac_f32 v42, v9, v14\n v_fmac_f32 v43, v9, v14\n v_fmac_f32 v44, v9, v14\n v_fmac_f32 v45, v9, v14\n v_fmac_f32 v46, v9, v14\n v_fmac_f32 v47, v9, v14\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v9", "v14");
        if (FORM == 17) asm volatile(REP8("v_cndmask_b32_e64 v40, v8, v13, s[20:21]\n v_cndmask_b32_e64 v41, v8, v13, s[20:21]\n v_cndmask_b32_e64 v42, v8, v13, s[20:21]\n v_cndmask_b32_e64 v43, v8, v13, s[20:21]\n v_cndmask_b32_e64 v44, v8, v13, s[20:21]\n v_cndmask_b32_e64 v45, v8, v13, s[20:21]\n v_cndmask_b32_e64 v46, v8, v13, s[20:21]\n v_cndmask_b32_e64 v47, v8, v13, s[20:21]\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v13");
        if (FORM == 18) asm volatile(REP8("v_bfi_b32 v40, v9, v14, v19\n v_bfi_b32 v41, v9, v14, v19\n v_bfi_b32 v42, v9, v14, v19\n v_bfi_b32 v43, v9, v14, v19\n v_bfi_b32 v44, v9, v14, v19\n v_bfi_b32 v45, v9, v14, v19\n v_bfi_b32 v46, v9, v14, v19\n v_bfi_b32 v47, v9, v14, v19\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v9", "v14", "v19");
        if (FORM == 19) asm volatile(REP8("v_and_or_b32 v40, v9, v14, v19\n v_and_or_b32 v41, v9, v14, v19\n v_and_or_b32 v42, v9, v14, v19\n v_and_or_b32 v43, v9, v14, v19\n v_and_or_b32 v44, v9, v14, v19\n v_and_or_b32 v45, v9, v14, v19\n v_and_or_b32 v46, v9, v14, v19\n v_and_or_b32 v47, v9, v14, v19\n ") ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v9", "v14", "v19");
        if (FORM == 20) asm volatile(REP8("s_nop 1\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_dpp v40, v10, v8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v41, v11, v9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_not_b64 vcc, vcc\n v_cndmask_b32_dpp v42, v8, v10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp v43, v9, v11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") ::: "v8", "v9", "v10", "v11", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc", "scc");
        if (FORM == 21) asm volatile(REP8(" v_mov_b32_dpp v44, v10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v45, v11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v46, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v47, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_e64 v40, v44, v8, s[20:21]\n v_cndmask_b32_e64 v41, v45, v9, s[20:21]\n v_cndmask_b32_e64 v42, v10, v46, s[20:21]\n v_cndmask_b32_e64 v43, v11, v47, s[20:21]\n") ::: "v8", "v9", "v10", "v11", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc", "scc");
        if (FORM == 22) asm volatile(REP8(" v_mov_b32 v44, v8\n v_mov_b32 v45, v9\n v_mov_b32_dpp v8, v10 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp v9, v11 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp v10, v44 row_shl:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp v11, v45 row_shl:4 row_mask:0xf bank_mask:0x5\n") ::: "v8", "v9", "v10", "v11", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc", "scc");
    }
    if (reps < 0) out[threadIdx.x] = acc;
}

template <int FORM>
static double run(int waves, int reps, float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * waves;
    k<FORM><<<blocks, 64>>>(d, 10);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(a);
        k<FORM><<<blocks, 64>>>(d, reps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    const int reps = 20000;
    const char* names[] = {"add, sources of one residue mod 4", "add, sources of two residues", "fma, three sources of one residue", "fma, three residues", "add, destination residue = a source's",
                           "add, dependent chain", "add, two interleaved chains", "v_fmac_f32_dpp row_ror:8", "v_cndmask_b32_dpp quad_perm", "v_mov_b32_dpp row_ror:8", "v_add_f32_dpp row_ror:8", "v_cndmask_b32 (no DPP), vcc", "v_cndmask_b32_dpp row_ror:8", "v_mov_b32_dpp quad_perm", "v_cndmask_b32_dpp quad_perm, sources v8 / v12 (one residue)", "fma, two sources of one residue", "v_fmac_f32 (dst v40.., sources of other residues)", "v_cndmask_b32_e64, mask in s[20:21]", "v_bfi_b32 (three VGPR sources of three residues)", "v_and_or_b32", "float2 pair swap: s_nop, s_mov vcc, 2 DPP selects, s_not, 2 DPP selects (per BLOCK / 8)", "float2 pair swap: 4 v_mov_b32_dpp + 4 v_cndmask_b32_e64 with an SGPR mask (per BLOCK / 8)", "float2 pair swap: 2 copies + 4 bank-masked v_mov_b32_dpp (per BLOCK / 8)"};
    for (int waves : {1, 4}) {
        double ms[23] = {run<0>(waves, reps, d), run<1>(waves, reps, d), run<2>(waves, reps, d), run<3>(waves, reps, d), run<4>(waves, reps, d), run<5>(waves, reps, d), run<6>(waves, reps, d),
                         run<7>(waves, reps, d), run<8>(waves, reps, d), run<9>(waves, reps, d), run<10>(waves, reps, d), run<11>(waves, reps, d), run<12>(waves, reps, d), run<13>(waves, reps, d), run<14>(waves, reps, d), run<15>(waves, reps, d), run<16>(waves, reps, d), run<17>(waves, reps, d), run<18>(waves, reps, d), run<19>(waves, reps, d), run<20>(waves, reps, d), run<21>(waves, reps, d), run<22>(waves, reps, d)};
        for (int f = 0; f < 23; ++f)
            printf("%-110s %d wave(s) per SIMD: %.3f ms -> %.2f SIMD cycles per instruction at 2.4 GHz\n", names[f], waves, ms[f], ms[f] * 1e-3 * 2.4e9 / ((double)reps * 64 * waves));
    }
    return 0;
}
