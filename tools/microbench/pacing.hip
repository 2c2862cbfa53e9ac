// Pacing microbenchmark: what does a streaming copy of 4 GiB -> 4 GiB reach on gfx950 as a function of
//   VEC   bytes per lane per instruction (8, 16)
//   NLD   load instructions per wave tile (tile = 64 * VEC * NLD bytes)
//   TRIP  0 none | 1 one LDS write + read-back of the tile | 2 one ds_bpermute (identity) per dword
//         | 3 the LDS trip AFTER issuing the next tile's loads is not modelled here
// and of the grid cap.  Motivation: one LDS round trip makes the plain copy 7 % faster.
// Build: hipcc -O3 --offload-arch=gfx950 pacing.hip -o pacing ; run: ./pacing [out_offset_GiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VEC> struct VecT;
template <> struct VecT<8> { using type = v2f; };
template <> struct VecT<16> { using type = v4f; };

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MAP: 0 grid stride (block b: tiles b, b+G, ...: with G % 8 == 0 every XCD keeps one residue class mod 8)
//      1 the residue class rotates with the iteration (every XCD visits all 8 classes)
//      2 XCD x (= blockIdx % 8) owns the x-th contiguous eighth of the buffers
template <int VEC, int NLD, int TRIP, int WAVES, int MAP = 0>
__global__ void __launch_bounds__(64 * WAVES) copyk(const char* __restrict__ in, char* __restrict__ out, long ntiles /* block tiles */) {
    using V = typename VecT<VEC>::type;
    constexpr int kWaveBytes = 64 * VEC * NLD;
    constexpr int kWaveElems = 64 * NLD;
    __shared__ V s[(TRIP == 1 || TRIP == 4 || TRIP == 5 || TRIP >= 200) ? WAVES * (kWaveElems + 64 * 8 / VEC) : (TRIP >= 100 ? 64 * WAVES : 1)];
    static_assert(TRIP < 600 || TRIP >= 700 || true, "");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long iter = 0;
    for (long t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x, ++iter) {
        long tile = t0;
        if (MAP == 1) tile = (t0 & ~7L) | ((t0 + iter) & 7);
        if (MAP == 2) tile = (long)(blockIdx.x & 7) * (ntiles / 8) + (blockIdx.x >> 3) + iter * (gridDim.x >> 3);
        const V* g = reinterpret_cast<const V*>(in + (tile * WAVES + wave) * kWaveBytes) + lane;
        V* o = reinterpret_cast<V*>(out + (tile * WAVES + wave) * kWaveBytes) + lane;
        V r[NLD];
#pragma unroll
        for (int c = 0; c < NLD; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
        if (TRIP == 1) {
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
#pragma unroll
            for (int c = 0; c < NLD; ++c) sw[lane + 64 * c] = r[c];
            wave_sync();
#pragma unroll
            for (int c = 0; c < NLD; ++c) r[c] = *reinterpret_cast<const volatile V*>(&sw[lane + 64 * c]);
            wave_sync();
        } else if (TRIP >= 800 && TRIP < 1000) {
            // TRIP = 800 + S: all loads complete, one s_sleep S (64 * S cycles), then the stores
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
            __builtin_amdgcn_s_sleep(TRIP >= 800 && TRIP < 1000 ? TRIP - 800 : 0);
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
        } else if (TRIP >= 700 && TRIP < 800) {
            // TRIP = 700 + K: as 600 + K but every flat load is waited for before the next one is issued
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
            constexpr int KS = (TRIP >= 700 && TRIP < 800) ? TRIP - 700 : 1;
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                const V* q = &sw[lane + 64 * (c % NLD)];
                V d;
                if (VEC == 8) asm volatile("flat_load_dwordx2 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(d) : "v"(q) : "memory");
                else asm volatile("flat_load_dwordx4 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(d) : "v"(q) : "memory");
            }
        } else if (TRIP >= 600 && TRIP < 700) {
            // TRIP = 600 + K: after all global loads have arrived, K FLAT loads of 8 B/lane from this wave's LDS rows
            // (whatever is there), issued back to back, results discarded, one wait; the stored data never touch LDS
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
            constexpr int KF = (TRIP >= 600 && TRIP < 700) ? TRIP - 600 : 1;
            V dummy[KF];
#pragma unroll
            for (int c = 0; c < KF; ++c) {
                const V* q = &sw[lane + 64 * (c % NLD)];
                if (VEC == 8) asm volatile("flat_load_dwordx2 %0, %1" : "=v"(dummy[c]) : "v"(q) : "memory");
                else asm volatile("flat_load_dwordx4 %0, %1" : "=v"(dummy[c]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < KF; ++c) asm volatile("" : : "v"(dummy[c]));
        } else if (TRIP == 4) {
            // LDS write, then ordinary ds_read read-back, each read waited for before the next is issued (serialised)
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
#pragma unroll
            for (int c = 0; c < NLD; ++c) sw[lane + 64 * c] = r[c];
            wave_sync();
#pragma unroll
            for (int c = 0; c < NLD; ++c) {
                r[c] = sw[lane + 64 * c];
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[c]) : : "memory");
            }
            wave_sync();
        } else if (TRIP == 5) {
            // LDS write, then FLAT loads of the same locations issued back to back, ONE wait at the end (not serialised)
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
#pragma unroll
            for (int c = 0; c < NLD; ++c) sw[lane + 64 * c] = r[c];
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < NLD; ++c) {
                const V* q = &sw[lane + 64 * c];
                if (VEC == 8) asm volatile("flat_load_dwordx2 %0, %1" : "=v"(r[c]) : "v"(q) : "memory");
                else asm volatile("flat_load_dwordx4 %0, %1" : "=v"(r[c]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
            wave_sync();
        } else if (TRIP >= 300) {
            // TRIP = 300 + S: all loads complete, then S x s_sleep 127 (8128 cycles each), then the stores
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
            for (int k = 0; k < TRIP - 300; ++k) __builtin_amdgcn_s_sleep(127);
#pragma unroll
            for (int c = 0; c < NLD; ++c) asm volatile("" : "+v"(r[c]));
        } else if (TRIP >= 200) {
            // TRIP = 200 + K: only the first K registers make the write + volatile (flat, serialised) read-back trip
            constexpr int K = TRIP - 200;
            V* sw = s + wave * (kWaveElems + 64 * 8 / VEC);
#pragma unroll
            for (int c = 0; c < K; ++c) sw[lane + 64 * c] = r[c];
            wave_sync();
#pragma unroll
            for (int c = 0; c < K; ++c) r[c] = *reinterpret_cast<const volatile V*>(&sw[lane + 64 * c]);
            wave_sync();
        } else if (TRIP >= 100) {
            // TRIP = 100 + K: K serialised FLAT loads from LDS (volatile generic-pointer reads: flat_load + s_waitcnt
            // vmcnt(0) each), nothing written: the part of the "LDS trip" that turned out to matter
            const volatile int* q = reinterpret_cast<const volatile int*>(s) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < TRIP - 100; ++k) (void)*q;
        } else if (TRIP == 2) {
#pragma unroll
            for (int c = 0; c < NLD; ++c)
#pragma unroll
                for (int k = 0; k < VEC / 4; ++k)
                    r[c][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(lane * 4, __float_as_int(r[c][k])));
        }
#pragma unroll
        for (int c = 0; c < NLD; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

struct Variant { const char* name; void (*launch)(const char*, char*, long, int); long block_bytes; };
template <int VEC, int NLD, int TRIP, int WAVES, int MAP = 0>
void launch(const char* in, char* out, long nbytes, int cap) {
    long ntiles = nbytes / (64L * VEC * NLD * WAVES);
    long g = (cap > 0 && ntiles > cap) ? cap : ntiles;
    g &= ~7L;   // MAP 1/2 need a multiple of 8 (ntiles is one)
    copyk<VEC, NLD, TRIP, WAVES, MAP><<<dim3((unsigned)g), dim3(64 * WAVES)>>>(in, out, ntiles);
}
#define V(VEC, NLD, TRIP, WAVES) {"vec" #VEC " nld" #NLD " trip" #TRIP " waves" #WAVES, launch<VEC, NLD, TRIP, WAVES>, 64L * VEC * NLD * WAVES}
#define VM(VEC, NLD, TRIP, WAVES, MAP) {"vec" #VEC " nld" #NLD " trip" #TRIP " waves" #WAVES " map" #MAP, launch<VEC, NLD, TRIP, WAVES, MAP>, 64L * VEC * NLD * WAVES}

int main(int argc, char** argv) {
    const long nbytes = 4L << 30;
    long out_off = (argc > 1 ? atol(argv[1]) : 40) << 30;
    const char* in;
    char* out;
    if (argc > 1 && !strcmp(argv[1], "pair")) {
        // buffers from the library's placement search (run from the repository root)
        void* lib = dlopen("smfft_amd/libsmfft_amd.so", RTLD_NOW);
        if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
        auto pair = (int (*)(unsigned long long, void**, void**))dlsym(lib, "smfft_malloc_pair");
        void *a, *b;
        if (!pair || pair((unsigned long long)nbytes, &a, &b) != 0) { printf("smfft_malloc_pair failed\n"); return 1; }
        in = (const char*)a; out = (char*)b; out_off = 0;
        CK(hipMemset(a, 1, nbytes));
    } else {
        char* arena;
        CK(hipMalloc(&arena, out_off + nbytes));
        CK(hipMemset(arena, 1, nbytes));
        in = arena;
        out = arena + out_off;
    }
    std::vector<Variant> vs = {
        V(8, 16, 0, 4), V(8, 16, 1, 4), V(8, 16, 708, 4), V(8, 16, 712, 4), V(8, 16, 716, 4), V(8, 16, 720, 4),
        V(8, 8, 0, 4), V(8, 8, 708, 4), V(16, 8, 0, 4), V(16, 8, 708, 4), V(16, 16, 0, 4), V(16, 16, 716, 4), V(8, 32, 0, 4), V(8, 32, 716, 4),
    };
    const int caps[] = {8192, 12288, 16384, 24576};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> res(vs.size() * 4);
    for (int round = 0; round < 5; ++round)
        for (size_t v = 0; v < vs.size(); ++v)
            for (int ci = 0; ci < 4; ++ci) {
                // same number of workgroup-slots worth of bytes per cap: scale the cap by the block size
                int cap = (int)(caps[ci] * (32768.0 / vs[v].block_bytes));
                vs[v].launch(in, out, nbytes, cap);
                CK(hipEventRecord(e0));
                for (int k = 0; k < 5; ++k) vs[v].launch(in, out, nbytes, cap);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                res[v * 4 + ci].push_back(ms / 5);
            }
    printf("out offset %ld GiB; GB/s (median of 5 rounds x 5 launches) at caps (scaled to 32 KiB blocks) 8192 12288 16384 24576\n", out_off >> 30);
    for (size_t v = 0; v < vs.size(); ++v) {
        printf("%-28s", vs[v].name);
        for (int ci = 0; ci < 4; ++ci) {
            auto& r = res[v * 4 + ci];
            std::sort(r.begin(), r.end());
            printf("  %6.0f", 2.0 * nbytes / r[r.size() / 2] / 1e6);
        }
        printf("\n");
    }
    return 0;
}
