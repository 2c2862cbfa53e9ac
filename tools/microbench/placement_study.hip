// placement_study -- why does the streaming rate of "read buffer A, write buffer B" depend on WHICH memory A and B are?
//
// Round 1 found (profiles/r01_chunk_map.txt) that separately allocated chunks fall into three classes (same class:
// 1.51-1.55 ms for the 4 GiB + 4 GiB N=1024 batch, different classes: 1.41-1.46 ms) plus a "fast write region"
// (1.31-1.39 ms as an output, slow as an input).  This tool produces the evidence round 2 needs:
//
//   placement_study map [chunk_GiB] [max_chunks]
//       allocates chunks in order, and per chunk j prints: pure read rate, pure write rate, copy 0 -> j, copy j -> 0,
//       copy (j-1) -> j; then the full (input, output) matrix.  Also dumps the KFD memory-bank properties.
//   placement_study pmc [chunk_GiB] [max_chunks]
//       the same allocation + classification against chunk 0, then a fixed sequence of TAGGED dispatches (the tag is a
//       template argument, so it is part of the kernel name a `rocprofv3 --pmc` csv reports):
//         study_copy<1>  0 -> S   S = a chunk of chunk 0's class (slowest output)
//         study_copy<2>  0 -> X   X = a chunk of another class
//         study_copy<3>  0 -> F   F = the best write target found
//         study_copy<4>  F -> 0
//         study_copy<5>  S -> 0
//         study_copy<6>  X -> 0
//         study_read<1..3> on 0 / S|X / F, study_write<1..3> likewise
//         study_copy_paced<1..3>  the same three outputs with the 16-flat-load "VMEM throttle" of the external kernels
//       Classification dispatches use tag 0.
//   placement_study vmm [handle_MiB] [max_GiB]
//       the hypothesis test: physical memory allocated through the VMM API in small handles (hipMemCreate), 1 GiB
//       chunks classified as in `map`, then 4 GiB test buffers ASSEMBLED from chunks of chosen classes -- whole, or
//       interleaved handle by handle -- and timed as copy targets: does an output interleaved over the classes
//       reproduce the "fast write region"?
//   placement_study vmm7
//       remap check: after hipMemUnmap, does hipMemMap of ANOTHER handle at the same virtual address take effect?
//       (ROCm 7.2 / MI355X: no -- the GPU keeps translating to the first handle; profiles/r02_vmm_remap_check.txt.  Every
//       mode of this tool, and the library's allocator, therefore use a virtual address for ONE mapping only.)
// Build: hipcc -O3 --offload-arch=gfx950 placement_study.hip -o placement_study
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// the external kernels' access shape: 256-thread workgroups, tile = 4096 float2, each wave moves 8 KiB with
// 16 x 8 B/lane non-temporal loads at 512 B stride, grid-stride over tiles
template <int TAG>
__global__ void __launch_bounds__(256) study_copy(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v2f* g = in + tile * 4096 + wave * 1024 + lane;
        v2f* o = out + tile * 4096 + wave * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

// same, with K serialised flat loads from LDS between the loads and the stores (smfft_kernels.hpp vmem_throttle)
template <int TAG, int K>
__global__ void __launch_bounds__(256) study_copy_paced(const v2f* __restrict__ in, v2f* __restrict__ out, long ntiles) {
    __shared__ v2f s[4352];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v2f* g = in + tile * 4096 + wave * 1024 + lane;
        v2f* o = out + tile * 4096 + wave * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(r[c]));
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const v2f* q = s + wave * 1088 + lane + 64 * (c & 15);
            v2f d;
            asm volatile("flat_load_dwordx2 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(d) : "v"(q) : "memory");
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(r[c], o + 64 * c);
    }
}

template <int TAG>
__global__ void __launch_bounds__(256) study_read(const v2f* __restrict__ in, v2f* __restrict__ sink, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f acc = {0.f, 0.f};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v2f* g = in + tile * 4096 + wave * 1024 + lane;
        v2f r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = __builtin_nontemporal_load(g + 64 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += r[c];
    }
    if (acc.x == 123.456f && acc.y == 654.321f) sink[threadIdx.x] = acc;   // never true for the data used
}

template <int TAG>
__global__ void __launch_bounds__(256) study_write(v2f* __restrict__ out, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v2f v = {(float)lane, (float)wave};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        v2f* o = out + tile * 4096 + wave * 1024 + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(v, o + 64 * c);
    }
}

static long g_ntiles;
static int g_grid = 12288;
static hipEvent_t g_e0, g_e1;

template <class F>
static float time_ms(F&& launch, int reps) {
    launch();
    CK(hipEventRecord(g_e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(g_e1, 0));
    CK(hipEventSynchronize(g_e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    return ms / reps;
}

template <int TAG> static float t_copy(const void* a, void* b, int reps) {
    return time_ms([&] { study_copy<TAG><<<g_grid, 256>>>((const v2f*)a, (v2f*)b, g_ntiles); }, reps);
}
template <int TAG> static float t_paced(const void* a, void* b, int reps) {
    return time_ms([&] { study_copy_paced<TAG, 16><<<g_grid, 256>>>((const v2f*)a, (v2f*)b, g_ntiles); }, reps);
}
template <int TAG> static float t_read(const void* a, void* sink, int reps) {
    return time_ms([&] { study_read<TAG><<<g_grid, 256>>>((const v2f*)a, (v2f*)sink, g_ntiles); }, reps);
}
template <int TAG> static float t_write(void* b, int reps) {
    return time_ms([&] { study_write<TAG><<<g_grid, 256>>>((v2f*)b, g_ntiles); }, reps);
}

// bytes the driver reports as used VRAM (first amdgpu card that exposes the file), -1 if unreadable
static long long vram_used() {
    for (int card = 0; card < 64; ++card) {
        char p[128];
        snprintf(p, sizeof p, "/sys/class/drm/card%d/device/mem_info_vram_used", card);
        FILE* f = fopen(p, "r");
        if (!f) continue;
        long long v = -1;
        if (fscanf(f, "%lld", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
    return -1;
}

static void dump_file(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return;
    char line[512];
    printf("--- %s\n", path);
    while (fgets(line, sizeof line, f)) printf("    %s", line);
    fclose(f);
}


static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int vmm_mode(size_t handle_mib, size_t max_gib) {
    const size_t G = 1ull << 30, H = handle_mib << 20, per_chunk = G / H;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    size_t nchunks = std::min<size_t>(max_gib, (free_b - (24ull << 30)) / G);
    printf("vmm: granularity %zu, handles of %zu MiB, %zu chunks of 1 GiB\n", gran, handle_mib, nchunks);
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, nchunks * G, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> handle(nchunks * per_chunk);
    double t0 = now_s();
    size_t made = 0;
    for (; made < handle.size(); ++made)
        if (hipMemCreate(&handle[made], H, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    nchunks = made / per_chunk;
    printf("hipMemCreate: %zu handles in %.3f s (%.1f us each, %.2f ms per GiB)\n", made, now_s() - t0, (now_s() - t0) / made * 1e6, (now_s() - t0) / (made * H / (double)G) * 1e3);
    t0 = now_s();
    for (size_t h = 0; h < nchunks * per_chunk; ++h) CK(hipMemMap(va + h * H, H, 0, handle[h], 0));
    CK(hipMemSetAccess(va, nchunks * G, &acc, 1));
    printf("hipMemMap + hipMemSetAccess: %.3f s\n", now_s() - t0);
    g_ntiles = (long)(G / 8 / 4096);      // probes over 1 GiB windows
    for (size_t j = 0; j < nchunks; ++j) study_write<0><<<g_grid, 256>>>((v2f*)(va + j * G), g_ntiles);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 300; ++k) study_copy<0><<<g_grid, 256>>>((const v2f*)va, (v2f*)(va + G), g_ntiles);
    CK(hipDeviceSynchronize());
    // classify against chunk 0: copy 0 -> j, and the pure write rate of j
    std::vector<float> out0(nchunks, 0.f), wr(nchunks), rd(nchunks);
    for (size_t j = 0; j < nchunks; ++j) {
        if (j) out0[j] = t_copy<0>(va, va + j * G, 4);
        wr[j] = t_write<0>(va + j * G, 4);
        rd[j] = t_read<0>(va + j * G, va + (j ? 0 : G), 4);
    }
    printf("chunk: copy 0->j ms (1 GiB + 1 GiB), write ms, read ms\n");
    for (size_t j = 0; j < nchunks; ++j) printf("%3zu %.3f %.3f %.3f\n", j, out0[j], wr[j], rd[j]);
    // mixed chunks: pure write clearly faster than the typical chunk (ordinary chunks of the three classes cannot be told
    // apart reliably on a 1 GiB window: 3-4 % against 2 % of noise; the 4 GiB `map` mode does that)
    std::vector<float> sorted_wr(wr);
    std::sort(sorted_wr.begin(), sorted_wr.end());
    const float wr_typ = sorted_wr[nchunks / 2];
    std::vector<size_t> mixed, ordinary;
    for (size_t j = 1; j < nchunks; ++j) (wr[j] < 0.92f * wr_typ ? mixed : ordinary).push_back(j);
    std::sort(mixed.begin(), mixed.end(), [&](size_t x, size_t y) { return wr[x] < wr[y]; });
    printf("typical write %.3f ms per GiB; %zu mixed chunks (fastest writes first):", wr_typ, mixed.size());
    for (size_t j : mixed) printf(" %zu", j);
    printf("\n");
    if (mixed.size() < 8 || ordinary.size() < 12) { printf("not enough mixed chunks inside the allocated range\n"); return 0; }

    // assemble 4 GiB test buffers in a second VA range out of the handles of chosen chunks (unmapped from `va` first)
    g_ntiles = (long)(4 * G / 8 / 4096);
    auto assemble = [&](std::vector<size_t> chunks, const char* what) -> char* {
        char* t = nullptr;
        CK(hipMemAddressReserve((void**)&t, 4 * G, 0, nullptr, 0));
        size_t mapped = 0;
        for (size_t ch : chunks) {
            CK(hipMemUnmap(va + ch * G, G));
            for (size_t h = 0; h < per_chunk; ++h) CK(hipMemMap(t + (mapped++) * H, H, 0, handle[ch * per_chunk + h], 0));
        }
        CK(hipMemSetAccess(t, 4 * G, &acc, 1));
        study_write<0><<<g_grid, 256>>>((v2f*)t, g_ntiles);
        CK(hipDeviceSynchronize());
        printf("assembled %s:", what);
        for (size_t ch : chunks) printf(" %zu", ch);
        printf("\n");
        return t;
    };
    auto pick = [&](std::vector<size_t>& from, size_t first) { return std::vector<size_t>(from.begin() + first, from.begin() + first + 4); };
    char* inA = assemble(pick(ordinary, 0), "inA (ordinary chunks)");
    char* outA = assemble(pick(ordinary, 4), "outA (ordinary chunks)");
    char* outFar = assemble(pick(ordinary, ordinary.size() - 4), "outFar (the last ordinary chunks allocated)");
    char* outM = assemble(pick(mixed, 0), "outM (the four most mixed chunks)");
    char* outM2 = assemble(pick(mixed, 4), "outM2 (the next four mixed chunks)");
    struct Case { const char* name; char* in; char* out; };
    std::vector<Case> cases = {{"ordinary -> ordinary (adjacent)", inA, outA}, {"ordinary -> ordinary (far)", inA, outFar}, {"ordinary -> mixed", inA, outM},
                               {"ordinary -> mixed (2nd set)", inA, outM2}, {"mixed -> ordinary", outM, outA}, {"mixed -> mixed", outM, outM2}};
    for (int rep = 0; rep < 2; ++rep)
        for (auto& c : cases)
            printf("%-32s copy %.3f ms | write-only into the output %.3f ms | read-only from the input %.3f ms\n", c.name, t_copy<0>(c.in, c.out, 6),
                   t_write<0>(c.out, 6), t_read<0>(c.in, c.out, 6));
    return 0;
}

// vmm_il: can ORDINARY memory be made to behave like mixed memory by interleaving it by hand?  120 GiB of physical memory in
// creation order, 1 GiB chunks classified (mixed / ordinary) by a write pass; 4 GiB buffers assembled (each in a fresh virtual
// range) from ordinary chunks taken near the start, the middle and the end of the range -- consecutive ("seq": one region) or
// handle by handle round-robin over the three regions ("il") -- and the copy matrix between them, with mixed targets as reference.
static int vmm_il_mode(size_t handle_mib, size_t max_gib) {
    const size_t G = 1ull << 30, H = handle_mib << 20, per_chunk = G / H;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    size_t nchunks = std::min<size_t>(max_gib, (free_b - (24ull << 30)) / G);
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, nchunks * G, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> handle(nchunks * per_chunk);
    size_t made = 0;
    for (; made < handle.size(); ++made)
        if (hipMemCreate(&handle[made], H, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    nchunks = made / per_chunk;
    for (size_t h = 0; h < nchunks * per_chunk; ++h) CK(hipMemMap(va + h * H, H, 0, handle[h], 0));
    CK(hipMemSetAccess(va, nchunks * G, &acc, 1));
    printf("vmm_il: %zu chunks of 1 GiB in handles of %zu MiB\n", nchunks, handle_mib);
    g_ntiles = (long)(G / 8 / 4096);
    for (size_t j = 0; j < nchunks; ++j) study_write<0><<<g_grid, 256>>>((v2f*)(va + j * G), g_ntiles);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 300; ++k) study_copy<0><<<g_grid, 256>>>((const v2f*)va, (v2f*)(va + G), g_ntiles);
    CK(hipDeviceSynchronize());
    std::vector<float> wr(nchunks);
    for (size_t j = 0; j < nchunks; ++j) wr[j] = t_write<0>(va + j * G, 4);
    std::vector<float> sorted_wr(wr);
    std::sort(sorted_wr.begin(), sorted_wr.end());
    const float wr_typ = sorted_wr[nchunks / 2];
    std::vector<size_t> mixed, ordinary;
    for (size_t j = 0; j < nchunks; ++j) {
        if (wr[j] < 0.90f * wr_typ) mixed.push_back(j);
        else if (wr[j] > 0.97f * wr_typ) ordinary.push_back(j);      // chunks in between are left out of the experiment
    }
    std::sort(mixed.begin(), mixed.end(), [&](size_t x, size_t y) { return wr[x] < wr[y]; });
    printf("write ms per GiB by chunk:");
    for (size_t j = 0; j < nchunks; ++j) printf(" %.3f", wr[j]);
    printf("\ntypical write %.3f ms per GiB; %zu mixed, %zu strictly ordinary chunks\n", wr_typ, mixed.size(), ordinary.size());
    if (mixed.size() < 4 || ordinary.size() < 75) { printf("not enough chunks\n"); return 0; }
    CK(hipMemUnmap(va, nchunks * G));
    // three regions of ordinary chunks: start, middle, end of the creation order; each gives its chunks away one after the other
    const size_t third = ordinary.size() / 3;
    size_t next_chunk[3] = {0, third, 2 * third};
    auto take_chunk = [&](int region) { return ordinary[next_chunk[region]++]; };
    g_ntiles = (long)(4 * G / 8 / 4096);
    auto map_handles = [&](const std::vector<hipMemGenericAllocationHandle_t>& hs, const char* what) -> char* {
        char* t = nullptr;
        CK(hipMemAddressReserve((void**)&t, 4 * G, 0, nullptr, 0));
        for (size_t k = 0; k < hs.size(); ++k) CK(hipMemMap(t + k * H, H, 0, hs[k], 0));
        CK(hipMemSetAccess(t, 4 * G, &acc, 1));
        study_write<0><<<g_grid, 256>>>((v2f*)t, g_ntiles);
        CK(hipDeviceSynchronize());
        printf("assembled %s\n", what);
        return t;
    };
    auto seq = [&](int region, const char* what) {
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (int c = 0; c < 4; ++c) { const size_t ch = take_chunk(region); for (size_t h = 0; h < per_chunk; ++h) hs.push_back(handle[ch * per_chunk + h]); }
        return map_handles(hs, what);
    };
    // `stripe` consecutive handles from one region, then from the next, ...: 4 chunks of every region make three buffers' worth;
    // one buffer takes 4/3 chunks of each (the spare handles stay unused)
    // regs: the regions the `ways` interleaved streams come from (a region may appear more than once: separate chunks of it)
    auto il_regions = [&](std::vector<int> regs, size_t stripe, const char* what) {
        const size_t ways = regs.size();
        std::vector<size_t> ch(ways), used(ways, 0);
        for (size_t w = 0; w < ways; ++w) ch[w] = take_chunk(regs[w]);
        std::vector<hipMemGenericAllocationHandle_t> hs;
        const size_t need = 4 * per_chunk;
        for (size_t k = 0; hs.size() < need; ++k) {
            const size_t w = k % ways;
            for (size_t i = 0; i < stripe && hs.size() < need; ++i) {
                if (used[w] == per_chunk) { ch[w] = take_chunk(regs[w]); used[w] = 0; }
                hs.push_back(handle[ch[w] * per_chunk + used[w]++]);
            }
        }
        return map_handles(hs, what);
    };
    auto il = [&](size_t stripe, const char* what) { return il_regions({0, 1, 2}, stripe, what); };
    auto from_mixed = [&](size_t first, const char* what) {
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t c = 0; c < 4; ++c) for (size_t h = 0; h < per_chunk; ++h) hs.push_back(handle[mixed[(first + c) % mixed.size()] * per_chunk + h]);
        return map_handles(hs, what);
    };
    char* inSeq = seq(0, "inSeq   (4 ordinary chunks, start of the range)");
    char* outSeq0 = seq(0, "outSeq0 (4 ordinary chunks, start of the range)");
    char* outSeq1 = seq(1, "outSeq1 (4 ordinary chunks, middle)");
    char* outSeq2 = seq(2, "outSeq2 (4 ordinary chunks, end)");
    char* inIl1 = il(1, "inIl1   (one handle from each region in turn)");
    char* outIl1 = il(1, "outIl1  (one handle from each region in turn)");
    char* inIl16 = il(16, "inIl16  (16 handles from each region in turn)");
    char* outIl16 = il(16, "outIl16 (16 handles from each region in turn)");
    char* outM = from_mixed(0, "outM    (the four most mixed chunks)");
    char* out000 = il_regions({0, 0, 0}, 1, "out000  (three chunks of the first region, handle by handle)");
    char* out01 = il_regions({0, 1}, 1, "out01   (first and middle region, handle by handle)");
    char* out12 = il_regions({1, 2}, 1, "out12   (middle and last region, handle by handle)");
    char* out02 = il_regions({0, 2}, 1, "out02   (first and last region, handle by handle)");
    char* out0122 = il_regions({0, 1, 2, 2, 1, 0}, 1, "out012210 (six streams over the three regions)");
    struct Case { const char* name; char* in; char* out; };
    std::vector<Case> cases = {{"seq -> seq (same region)", inSeq, outSeq0}, {"seq -> seq (middle)", inSeq, outSeq1}, {"seq -> seq (end)", inSeq, outSeq2},
                               {"seq -> il1", inSeq, outIl1}, {"il1 -> seq (same region)", inIl1, outSeq0}, {"il1 -> il1", inIl1, outIl1},
                               {"seq -> il16", inSeq, outIl16}, {"il16 -> il16", inIl16, outIl16}, {"il1 -> il16", inIl1, outIl16},
                               {"seq -> mixed", inSeq, outM}, {"il1 -> mixed", inIl1, outM}, {"il16 -> mixed", inIl16, outM},
                               {"seq -> il(0,0,0)", inSeq, out000}, {"seq -> il(0,1)", inSeq, out01}, {"seq -> il(1,2)", inSeq, out12}, {"seq -> il(0,2)", inSeq, out02},
                               {"seq -> il(0,1,2,2,1,0)", inSeq, out0122}};
    for (int rep = 0; rep < 2; ++rep)
        for (auto& c : cases)
            printf("%-28s copy %.3f ms | paced copy %.3f ms | write-only %.3f ms | read-only %.3f ms\n", c.name, t_copy<0>(c.in, c.out, 6), t_paced<0>(c.in, c.out, 6),
                   t_write<0>(c.out, 6), t_read<0>(c.in, c.out, 6));
    return 0;
}

// vmm_cls: the three classes told apart with the interleave probe (write pass over a 1 GiB range in which the handles of two
// chunks alternate: fast = different classes): A = the class of the first ordinary chunk, B = the class of the first chunk that
// differs from it, C = what differs from both.  Then: input from A; outputs from one class, and interleaved from two -- is an
// output that shares NO class with the input better still?
static int vmm_cls_mode(size_t handle_mib, size_t max_gib) {
    const size_t G = 1ull << 30, H = handle_mib << 20, per_chunk = G / H;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    size_t nchunks = std::min<size_t>(max_gib, (free_b - (24ull << 30)) / G);
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, nchunks * G, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> handle(nchunks * per_chunk);
    for (size_t h = 0; h < handle.size(); ++h) CK(hipMemCreate(&handle[h], H, &prop, 0));
    for (size_t h = 0; h < nchunks * per_chunk; ++h) CK(hipMemMap(va + h * H, H, 0, handle[h], 0));
    CK(hipMemSetAccess(va, nchunks * G, &acc, 1));
    g_ntiles = (long)(G / 8 / 4096);
    for (size_t j = 0; j < nchunks; ++j) study_write<0><<<g_grid, 256>>>((v2f*)(va + j * G), g_ntiles);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 300; ++k) study_copy<0><<<g_grid, 256>>>((const v2f*)va, (v2f*)(va + G), g_ntiles);
    CK(hipDeviceSynchronize());
    std::vector<float> wr(nchunks);
    for (size_t j = 0; j < nchunks; ++j) wr[j] = t_write<0>(va + j * G, 4);
    std::vector<float> sorted_wr(wr);
    std::sort(sorted_wr.begin(), sorted_wr.end());
    const float wr_typ = sorted_wr[nchunks / 2];
    CK(hipMemUnmap(va, nchunks * G));
    std::vector<size_t> ordinary;
    for (size_t j = 0; j < nchunks; ++j) if (wr[j] > 0.97f * wr_typ) ordinary.push_back(j);
    auto differs = [&](size_t x, size_t y) {
        char* t = nullptr;
        CK(hipMemAddressReserve((void**)&t, G, 0, nullptr, 0));
        for (size_t k = 0; k < per_chunk / 2; ++k) {
            CK(hipMemMap(t + (2 * k) * H, H, 0, handle[x * per_chunk + k], 0));
            CK(hipMemMap(t + (2 * k + 1) * H, H, 0, handle[y * per_chunk + k], 0));
        }
        CK(hipMemSetAccess(t, G, &acc, 1));
        g_ntiles = (long)(G / 8 / 4096);
        const float ms = t_write<0>(t, 4);
        CK(hipMemUnmap(t, G));
        return ms < 0.91f * wr_typ;
    };
    std::vector<size_t> cls[3];
    cls[0].push_back(ordinary[0]);
    for (size_t i = 1; i < ordinary.size(); ++i) {
        const size_t j = ordinary[i];
        if (!differs(cls[0][0], j)) cls[0].push_back(j);
        else if (cls[1].empty() || !differs(cls[1][0], j)) cls[1].push_back(j);
        else cls[2].push_back(j);
    }
    printf("vmm_cls: %zu chunks, %zu strictly ordinary: class A %zu, B %zu, C %zu chunks\nclass by chunk:", nchunks, ordinary.size(), cls[0].size(), cls[1].size(), cls[2].size());
    {
        std::vector<char> tag(nchunks, '.');
        for (int c = 0; c < 3; ++c) for (size_t j : cls[c]) tag[j] = (char)('A' + c);
        for (size_t j = 0; j < nchunks; ++j) printf("%c", tag[j]);
        printf("\n");
    }
    if (cls[0].size() < 12 || cls[1].size() < 8 || cls[2].size() < 8) { printf("not enough chunks of every class\n"); return 0; }
    size_t next[3] = {0, 0, 0};
    g_ntiles = (long)(4 * G / 8 / 4096);
    auto build = [&](std::vector<int> ways, const char* what) -> char* {     // 4 GiB, handles round-robin over the listed classes
        std::vector<size_t> ch(ways.size()), used(ways.size(), 0);
        for (size_t w = 0; w < ways.size(); ++w) ch[w] = cls[ways[w]][next[ways[w]]++];
        char* t = nullptr;
        CK(hipMemAddressReserve((void**)&t, 4 * G, 0, nullptr, 0));
        for (size_t k = 0; k < 4 * per_chunk; ++k) {
            const size_t w = k % ways.size();
            if (used[w] == per_chunk) { ch[w] = cls[ways[w]][next[ways[w]]++]; used[w] = 0; }
            CK(hipMemMap(t + k * H, H, 0, handle[ch[w] * per_chunk + used[w]++], 0));
        }
        CK(hipMemSetAccess(t, 4 * G, &acc, 1));
        study_write<0><<<g_grid, 256>>>((v2f*)t, g_ntiles);
        CK(hipDeviceSynchronize());
        printf("assembled %s\n", what);
        return t;
    };
    char* inA = build({0}, "inA");
    struct Case { const char* name; char* in; char* out; };
    std::vector<Case> cases = {{"A -> A", inA, build({0}, "outA")}, {"A -> B", inA, build({1}, "outB")}, {"A -> C", inA, build({2}, "outC")},
                               {"A -> il(A,B)", inA, build({0, 1}, "outAB")}, {"A -> il(A,C)", inA, build({0, 2}, "outAC")}, {"A -> il(B,C)", inA, build({1, 2}, "outBC")},
                               {"A -> il(A,B,C)", inA, build({0, 1, 2}, "outABC")}};
    char* inBC = build({1, 2}, "inBC (interleaved input)");
    cases.push_back({"il(B,C) -> A", inBC, cases[0].out});
    for (int rep = 0; rep < 2; ++rep)
        for (auto& c : cases)
            printf("%-16s copy %.3f ms | paced copy %.3f ms | write-only %.3f ms | read-only %.3f ms\n", c.name, t_copy<0>(c.in, c.out, 6), t_paced<0>(c.in, c.out, 6),
                   t_write<0>(c.out, 6), t_read<0>(c.in, c.out, 6));
    return 0;
}

// vmm_spacer: the cheap way to physical memory of ANOTHER class: 2 GiB of handles (A), one plain hipMalloc of X GiB as a
// spacer, 2 GiB of handles (B), spacer freed; output = A and B interleaved handle by handle.  For X = 0, 8, ... GiB: how long the
// steps take, the write-only time of the output and the copy time from a hipMalloc input.
static int vmm_spacer_mode(size_t handle_mib) {
    const size_t G = 1ull << 30, H = handle_mib << 20, half = 2 * G / H;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* in = nullptr;
    CK(hipMalloc(&in, 4 * G));
    g_ntiles = (long)(4 * G / 8 / 4096);
    study_write<0><<<g_grid, 256>>>((v2f*)in, g_ntiles);
    for (int k = 0; k < 200; ++k) study_read<0><<<g_grid, 256>>>((const v2f*)in, (v2f*)in, g_ntiles);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep)
        for (size_t spacer_gib : {0, 8, 16, 24, 32, 48, 64, 96}) {
            std::vector<hipMemGenericAllocationHandle_t> a(half), b(half);
            double t0 = now_s();
            for (auto& h : a) CK(hipMemCreate(&h, H, &prop, 0));
            const double t_a = now_s() - t0;
            t0 = now_s();
            void* spacer = nullptr;
            if (spacer_gib && hipMalloc(&spacer, spacer_gib * G) != hipSuccess) { (void)hipGetLastError(); printf("spacer %zu GiB: no memory\n", spacer_gib); for (auto h : a) CK(hipMemRelease(h)); continue; }
            const double t_s = now_s() - t0;
            for (auto& h : b) CK(hipMemCreate(&h, H, &prop, 0));
            t0 = now_s();
            if (spacer) CK(hipFree(spacer));
            const double t_f = now_s() - t0;
            char* out = nullptr;
            CK(hipMemAddressReserve((void**)&out, 4 * G, 0, nullptr, 0));
            t0 = now_s();
            for (size_t k = 0; k < half; ++k) {
                CK(hipMemMap(out + (2 * k) * H, H, 0, a[k], 0));
                CK(hipMemMap(out + (2 * k + 1) * H, H, 0, b[k], 0));
            }
            CK(hipMemSetAccess(out, 4 * G, &acc, 1));
            const double t_m = now_s() - t0;
            study_write<0><<<g_grid, 256>>>((v2f*)out, g_ntiles);
            CK(hipDeviceSynchronize());
            printf("spacer %2zu GiB: create 2 GiB %.0f ms, hipMalloc spacer %.1f ms, hipFree %.1f ms, map 4 GiB %.0f ms | write-only %.3f ms | copy from hipMalloc input %.3f ms\n",
                   spacer_gib, t_a * 1e3, t_s * 1e3, t_f * 1e3, t_m * 1e3, t_write<0>(out, 6), t_copy<0>(in, out, 6));
            CK(hipMemUnmap(out, 4 * G));
            for (auto h : a) CK(hipMemRelease(h));
            for (auto h : b) CK(hipMemRelease(h));
            // the virtual range is deliberately not re-used (stale translations, see vmm7)
        }
    return 0;
}

// vmm7: does re-using a virtual range for ANOTHER handle (hipMemUnmap, then hipMemMap) really redirect the accesses?
// X and Y are two physical 8 MiB handles.  X mapped at S, filled with 1.0; unmapped; Y mapped at S, filled with 2.0;
// then X and Y are mapped at two FRESH ranges and read back.  Correct: X holds 1.0, Y holds 2.0.
__global__ void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static int vmm7_mode() {
    const size_t H = 8ull << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t X, Y;
    CK(hipMemCreate(&X, H, &prop, 0));
    CK(hipMemCreate(&Y, H, &prop, 0));
    char *S = nullptr, *FX = nullptr, *FY = nullptr;
    CK(hipMemAddressReserve((void**)&S, H, 0, nullptr, 0));
    CK(hipMemAddressReserve((void**)&FX, H, 0, nullptr, 0));
    CK(hipMemAddressReserve((void**)&FY, H, 0, nullptr, 0));
    const size_t n = H / 4;
    for (int variant = 0; variant < 2; ++variant) {
        CK(hipMemMap(S, H, 0, X, 0));
        CK(hipMemSetAccess(S, H, &acc, 1));
        fill_kernel<<<1024, 256>>>((float*)S, 1.0f, n);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(S, H));
        if (variant == 1) CK(hipDeviceSynchronize());
        CK(hipMemMap(S, H, 0, Y, 0));
        CK(hipMemSetAccess(S, H, &acc, 1));
        fill_kernel<<<1024, 256>>>((float*)S, 2.0f, n);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(S, H));
        CK(hipMemMap(FX, H, 0, X, 0));
        CK(hipMemSetAccess(FX, H, &acc, 1));
        CK(hipMemMap(FY, H, 0, Y, 0));
        CK(hipMemSetAccess(FY, H, &acc, 1));
        float hx[4], hy[4];
        CK(hipMemcpy(hx, FX + H / 2, sizeof hx, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hy, FY + H / 2, sizeof hy, hipMemcpyDeviceToHost));
        printf("variant %d: X holds %.1f (expected 1.0), Y holds %.1f (expected 2.0)  -> %s\n", variant, hx[0], hy[0],
               (hx[0] == 1.0f && hy[0] == 2.0f) ? "remapping works" : "STALE TRANSLATION: the second fill went to the first handle");
        fill_kernel<<<1024, 256>>>((float*)FX, 0.0f, n);
        fill_kernel<<<1024, 256>>>((float*)FY, 0.0f, n);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(FX, H));
        CK(hipMemUnmap(FY, H));
    }
    return 0;
}

// vmm8: when does the physical memory of VMM handles come back?  1 GiB in 8 MiB handles is created, mapped, written, unmapped
// (one call over all mappings / one call per mapping) and released; the virtual range is kept or freed: hipMemGetInfo after
// each step.  And: does hipMemAddressFree + a new reservation of the SAME address cure the stale translation of vmm7?
static size_t free_mib() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f >> 20; }
static int vmm8_mode() {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t H = 8ull << 20, N = 128;
    const size_t f0 = free_mib();
    for (int variant = 0; variant < 4; ++variant) {
        std::vector<hipMemGenericAllocationHandle_t> hs(N);
        for (auto& h : hs) CK(hipMemCreate(&h, H, &prop, 0));
        char* va = nullptr;
        CK(hipMemAddressReserve((void**)&va, N * H, 0, nullptr, 0));
        for (size_t k = 0; k < N; ++k) CK(hipMemMap(va + k * H, H, 0, hs[k], 0));
        CK(hipMemSetAccess(va, N * H, &acc, 1));
        CK(hipMemset(va, 1, N * H));
        CK(hipDeviceSynchronize());
        const size_t fm = free_mib();
        if (variant == 0 || variant == 2) CK(hipMemUnmap(va, N * H));
        else for (size_t k = 0; k < N; ++k) CK(hipMemUnmap(va + k * H, H));
        for (auto& h : hs) CK(hipMemRelease(h));
        const size_t fr = free_mib();
        if (variant >= 2) CK(hipMemAddressFree(va, N * H));
        CK(hipDeviceSynchronize());
        printf("variant %d (%s unmap, virtual range %s): in use while mapped %zu MiB; after unmap + release %zu MiB; at the end %zu MiB\n", variant,
               (variant & 1) ? "per-mapping" : "single-call", variant >= 2 ? "freed" : "kept", f0 - fm, f0 - fr, f0 - free_mib());
    }
    // the stale translation of vmm7 across hipMemAddressFree + hipMemAddressReserve of the same address
    const size_t n = H / 4;
    hipMemGenericAllocationHandle_t X, Y;
    CK(hipMemCreate(&X, H, &prop, 0));
    CK(hipMemCreate(&Y, H, &prop, 0));
    char *S = nullptr, *S2 = nullptr, *FX = nullptr, *FY = nullptr;
    CK(hipMemAddressReserve((void**)&S, H, 0, nullptr, 0));
    CK(hipMemMap(S, H, 0, X, 0));
    CK(hipMemSetAccess(S, H, &acc, 1));
    fill_kernel<<<1024, 256>>>((float*)S, 1.0f, n);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(S, H));
    CK(hipMemAddressFree(S, H));
    CK(hipMemAddressReserve((void**)&S2, H, 0, S, 0));
    printf("a new reservation with the freed address as its hint comes back at %s address\n", S2 == S ? "the SAME" : "another");
    CK(hipMemMap(S2, H, 0, Y, 0));
    CK(hipMemSetAccess(S2, H, &acc, 1));
    fill_kernel<<<1024, 256>>>((float*)S2, 2.0f, n);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(S2, H));
    CK(hipMemAddressFree(S2, H));
    CK(hipMemAddressReserve((void**)&FX, H, 0, (void*)((size_t)S + (64ull << 30)), 0));
    CK(hipMemAddressReserve((void**)&FY, H, 0, (void*)((size_t)S + (128ull << 30)), 0));
    printf("reservations with hints 64 and 128 GiB above it: %s\n", ((size_t)FX == (size_t)S + (64ull << 30) && (size_t)FY == (size_t)S + (128ull << 30)) ? "exactly where asked" : "elsewhere");
    CK(hipMemMap(FX, H, 0, X, 0));
    CK(hipMemSetAccess(FX, H, &acc, 1));
    CK(hipMemMap(FY, H, 0, Y, 0));
    CK(hipMemSetAccess(FY, H, &acc, 1));
    float hx = 0.f, hy = 0.f;
    CK(hipMemcpy(&hx, FX + H / 2, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hy, FY + H / 2, 4, hipMemcpyDeviceToHost));
    printf("X holds %.1f (expected 1.0), Y holds %.1f (expected 2.0) -> %s\n", hx, hy,
           (hx == 1.0f && hy == 2.0f) ? "freeing and re-reserving the address cures it" : "STALE TRANSLATION survives hipMemAddressFree + hipMemAddressReserve of the same address");
    return 0;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "map";
    const size_t chunk_gib = argc > 2 ? atol(argv[2]) : 4;
    const int max_chunks = argc > 3 ? atoi(argv[3]) : 64;
    const size_t window = 4ull << 30;                 // every probe moves the first 4 GiB of a chunk
    const size_t chunk_bytes = std::max(chunk_gib << 30, window);
    g_ntiles = (long)(window / 8 / 4096);
    CK(hipSetDevice(0));
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    if (mode == "vmm7") return vmm7_mode();
    if (mode == "vmm8") return vmm8_mode();
    if (mode == "vmm_cls") return vmm_cls_mode(argc > 2 ? atol(argv[2]) : 8, argc > 3 ? atol(argv[3]) : 140);
    if (mode == "vmm_spacer") return vmm_spacer_mode(argc > 2 ? atol(argv[2]) : 8);
    if (mode == "vmm_il") return vmm_il_mode(argc > 2 ? atol(argv[2]) : 8, argc > 3 ? atol(argv[3]) : 120);
    if (mode == "vmm") return vmm_mode(argc > 2 ? atol(argv[2]) : 8, argc > 3 ? atol(argv[3]) : 140);
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("mode %s: chunk %zu GiB, free %.1f GiB of %.1f GiB\n", mode.c_str(), chunk_bytes >> 30, free_b / 1073741824.0, total_b / 1073741824.0);
    if (mode == "map") {
        for (int node = 0; node < 16; ++node) {
            char p[256];
            for (int bank = 0; bank < 4; ++bank) {
                snprintf(p, sizeof p, "/sys/class/kfd/kfd/topology/nodes/%d/mem_banks/%d/properties", node, bank);
                dump_file(p);
            }
        }
    }
    std::vector<void*> chunk;
    const size_t reserve = 6ull << 30;
    printf("vram_used at start: %.2f GiB\n", vram_used() / 1073741824.0);
    // a stream-ordered-pool allocation made FIRST (round 1: such memory was the fast write region on 2 boxes of 3);
    // it becomes the LAST entry of the chunk list
    void* pool_chunk = nullptr;
    if (hipMallocAsync(&pool_chunk, window, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess) { (void)hipGetLastError(); pool_chunk = nullptr; }
    printf("vram_used after the pool allocation: %.2f GiB\n", vram_used() / 1073741824.0);
    while ((int)chunk.size() < max_chunks) {
        CK(hipMemGetInfo(&free_b, &total_b));
        if (free_b < chunk_bytes + reserve) break;
        void* p = nullptr;
        if (hipMalloc(&p, chunk_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        chunk.push_back(p);
    }
    printf("vram_used after %zu hipMalloc chunks: %.2f GiB\n", chunk.size(), vram_used() / 1073741824.0);
    if (pool_chunk) chunk.push_back(pool_chunk);
    const int n = (int)chunk.size();
    if (pool_chunk) printf("chunk %d is the hipMallocAsync allocation made first\n", n - 1);
    printf("%d chunks allocated; virtual addresses (GiB):", n);
    for (int j = 0; j < n; ++j) printf(" %.1f", (double)(uintptr_t)chunk[j] / 1073741824.0);
    printf("\n");
    if (n < 3) return 1;
    // touch everything once (first-touch effects out of the way) and warm the clocks
    for (int j = 0; j < n; ++j) study_write<0><<<g_grid, 256>>>((v2f*)chunk[j], g_ntiles);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 200; ++k) study_copy<0><<<g_grid, 256>>>((const v2f*)chunk[0], (v2f*)chunk[1], g_ntiles);
    CK(hipDeviceSynchronize());

    std::vector<float> rd(n), wr(n), out0(n), in0(n), adj(n);
    void* sink = chunk[n - 2];
    for (int j = 0; j < n; ++j) {
        rd[j] = t_read<0>(chunk[j], sink, 3);
        wr[j] = t_write<0>(chunk[j], 3);
        out0[j] = j ? t_copy<0>(chunk[0], chunk[j], 3) : 0.f;
        in0[j] = j ? t_copy<0>(chunk[j], chunk[0], 3) : 0.f;
        adj[j] = j ? t_copy<0>(chunk[j - 1], chunk[j], 3) : 0.f;
    }
    printf("per chunk (ms for the 4 GiB window; copies move 4 GiB in + 4 GiB out):\n  j   read  write  0->j   j->0  (j-1)->j\n");
    for (int j = 0; j < n; ++j) printf("%3d  %.3f  %.3f  %.3f  %.3f  %.3f\n", j, rd[j], wr[j], out0[j], in0[j], adj[j]);

    if (mode == "map") {
        printf("matrix: rows = input chunk, columns = output chunk (ms x 100, same chunk = 0)\n");
        for (int i = 0; i < n; ++i) {
            printf("%3d", i);
            for (int j = 0; j < n; ++j) {
                const float ms = (i == j) ? 0.f : t_copy<0>(chunk[i], chunk[j], 2);
                printf(" %3d", (int)(ms * 100.f + 0.5f));
            }
            printf("\n");
            fflush(stdout);
        }
    } else {
        // S: slowest output for input 0 (same class), F: fastest, X: the output closest to the midpoint of the two
        int S = 1, F = 1, X = 1;
        for (int j = 1; j < n; ++j) {
            if (out0[j] > out0[S]) S = j;
            if (out0[j] < out0[F]) F = j;
        }
        // X: an ordinary other-class chunk = the median output time
        std::vector<int> order;
        for (int j = 1; j < n; ++j) order.push_back(j);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return out0[a] < out0[b]; });
        X = order[order.size() / 2];
        printf("tags: S=%d (0->S %.3f ms)  X=%d (0->X %.3f ms)  F=%d (0->F %.3f ms)\n", S, out0[S], X, out0[X], F, out0[F]);
        const int reps = 6;
        printf("copy<1> 0->S %.3f\n", t_copy<1>(chunk[0], chunk[S], reps));
        printf("copy<2> 0->X %.3f\n", t_copy<2>(chunk[0], chunk[X], reps));
        printf("copy<3> 0->F %.3f\n", t_copy<3>(chunk[0], chunk[F], reps));
        printf("copy<4> F->0 %.3f\n", t_copy<4>(chunk[F], chunk[0], reps));
        printf("copy<5> S->0 %.3f\n", t_copy<5>(chunk[S], chunk[0], reps));
        printf("copy<6> X->0 %.3f\n", t_copy<6>(chunk[X], chunk[0], reps));
        printf("read<1> 0 %.3f\n", t_read<1>(chunk[0], sink, reps));
        printf("read<2> X %.3f\n", t_read<2>(chunk[X], sink, reps));
        printf("read<3> F %.3f\n", t_read<3>(chunk[F], sink, reps));
        printf("write<1> S %.3f\n", t_write<1>(chunk[S], reps));
        printf("write<2> X %.3f\n", t_write<2>(chunk[X], reps));
        printf("write<3> F %.3f\n", t_write<3>(chunk[F], reps));
        printf("paced<1> 0->S %.3f\n", t_paced<1>(chunk[0], chunk[S], reps));
        printf("paced<2> 0->X %.3f\n", t_paced<2>(chunk[0], chunk[X], reps));
        printf("paced<3> 0->F %.3f\n", t_paced<3>(chunk[0], chunk[F], reps));
    }
    if (pool_chunk) { chunk.pop_back(); (void)hipFreeAsync(pool_chunk, 0); (void)hipStreamSynchronize(0); }
    printf("vram_used before freeing: %.2f GiB\n", vram_used() / 1073741824.0);
    for (void* p : chunk) (void)hipFree(p);
    printf("vram_used right after hipFree of everything: %.2f GiB\n", vram_used() / 1073741824.0);
    CK(hipDeviceSynchronize());
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, 0) == hipSuccess) {
        (void)hipMemPoolTrimTo(pool, 0);
        printf("vram_used after hipMemPoolTrimTo(0): %.2f GiB\n", vram_used() / 1073741824.0);
    }
    return 0;
}
