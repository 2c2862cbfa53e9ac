// class_pairs.hip -- which (input, output) combinations of physical-memory classes take the config-2 copy fastest.
//
// Round 3 found that "ordinary memory" is not one thing: whole chunks of ONE class took the whole-pair copy from a hipMalloc
// input anywhere between 1.30 and 1.55 ms, depending on the (unknown) class of that input.  Here BOTH buffers are built
// through the virtual-memory API from chunks of known classes: NCHUNKS chunks of 1 GiB (128 handles of 8 MiB) are created,
// each gets a write pass (mixed chunks: clearly faster) and the ordinary ones are sorted into classes A (the first ordinary
// chunk's), B, C by write passes over ranges in which the handles of two chunks alternate (two classes interleaved write like
// mixed memory; same class: no gain).  Then every combination of 4 GiB inputs and outputs made of whole chunks of one class --
// and of mixed chunks, and of A/B interleaved handle by handle -- is timed with the library's same-shape copy
// (smfft_copy_launch, pacing off) and with the N = 1024 external FFT.
//
// Build: hipcc -O3 --offload-arch=gfx950 class_pairs.hip -o class_pairs -L../../smfft_amd -lsmfft_amd -Wl,-rpath,'$ORIGIN/../../smfft_amd'
// Run:   ./class_pairs [NCHUNKS = 96]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

extern "C" int smfft_copy_launch(const void* d_input, void* d_output, long long n_float2, void* hip_stream);
extern "C" int smfft_launch(int family, int path, const void* d_input, void* d_output, int FFT_size, int nFFTs, int inverse, int reorder, void* hip_stream);
extern "C" void smfft_set_pacing(int k);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr size_t kHandle = 8ull << 20, kChunk = 1ull << 30, kPerChunk = kChunk / kHandle, kBuf = 4ull << 30;

__global__ void __launch_bounds__(256) write_pass(float2* out, size_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float2* o = out + t * 4096 + wave * 1024 + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(1.0f * c, &o[64 * c].x), __builtin_nontemporal_store(0.5f, &o[64 * c].y);
    }
}

struct Chunk { std::vector<hipMemGenericAllocationHandle_t> hs; float write_ms = 0.f; char cls = '?'; };

static char* g_va = nullptr;       // one big reservation, carved upwards: no address is ever used twice (ROCm 7.2 keeps stale translations)
static size_t g_va_used = 0, g_va_size = 0;
static hipMemAccessDesc g_acc;

static char* carve(size_t bytes) {
    if (g_va_used + bytes > g_va_size) { printf("out of reserved address space\n"); exit(1); }
    char* p = g_va + g_va_used;
    g_va_used += bytes;
    return p;
}
static char* map_handles(const std::vector<hipMemGenericAllocationHandle_t>& hs) {
    char* va = carve(hs.size() * kHandle);
    for (size_t i = 0; i < hs.size(); ++i) CK(hipMemMap(va + i * kHandle, kHandle, 0, hs[i], 0));
    CK(hipMemSetAccess(va, hs.size() * kHandle, &g_acc, 1));
    return va;
}
static void unmap(char* va, size_t n_handles) { CK(hipMemUnmap(va, n_handles * kHandle)); }

static float time_ms(int reps, const std::function<void()>& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
static float write_ms_of(char* va, size_t bytes) {
    const size_t ntiles = bytes / 8 / 4096;
    return time_ms(3, [&] { write_pass<<<12288, 256>>>((float2*)va, ntiles); });
}
// write pass over a range in which the first halves of two chunks alternate handle by handle
static float interleave_probe(const Chunk& x, const Chunk& y) {
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t k = 0; k < kPerChunk / 2; ++k) { hs.push_back(x.hs[k]); hs.push_back(y.hs[k]); }
    char* va = map_handles(hs);
    const float ms = write_ms_of(va, kChunk);
    unmap(va, hs.size());
    return ms;
}

int main(int argc, char** argv) {
    const int nchunks = argc > 1 ? atoi(argv[1]) : 96;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    g_acc.location = prop.location;
    g_acc.flags = hipMemAccessFlagsProtReadWrite;
    g_va_size = 4ull << 40;                                   // 4 TiB of addresses
    CK(hipMemAddressReserve((void**)&g_va, g_va_size, 1ull << 30, nullptr, 0));
    smfft_set_pacing(0);

    std::vector<Chunk> chunks(nchunks);
    for (auto& c : chunks) {
        for (size_t h = 0; h < kPerChunk; ++h) {
            hipMemGenericAllocationHandle_t handle;
            CK(hipMemCreate(&handle, kHandle, &prop, 0));
            c.hs.push_back(handle);
        }
        char* va = map_handles(c.hs);
        c.write_ms = write_ms_of(va, kChunk);
        unmap(va, c.hs.size());
    }
    std::vector<float> t;
    for (auto& c : chunks) t.push_back(c.write_ms);
    std::sort(t.begin(), t.end());
    const float median = t[t.size() / 2];
    printf("write ms per chunk (median %.3f):", median);
    for (auto& c : chunks) printf(" %.3f", c.write_ms);
    printf("\n");
    // classes: mixed below 0.92 of the median, clearly ordinary above 0.97; A = the first ordinary chunk's class
    int refA = -1, refB = -1;
    for (int i = 0; i < nchunks; ++i) {
        Chunk& c = chunks[i];
        if (c.write_ms < 0.92f * median) { c.cls = 'M'; continue; }
        if (c.write_ms < 0.97f * median) { c.cls = '-'; continue; }
        if (refA < 0) { refA = i; c.cls = 'A'; continue; }
        const float own = 0.5f * (c.write_ms + chunks[refA].write_ms);
        if (interleave_probe(chunks[refA], c) > 0.90f * own) { c.cls = 'A'; continue; }
        if (refB < 0) { refB = i; c.cls = 'B'; continue; }
        const float ownb = 0.5f * (c.write_ms + chunks[refB].write_ms);
        c.cls = interleave_probe(chunks[refB], c) > 0.90f * ownb ? 'B' : 'C';
    }
    printf("classes:");
    for (auto& c : chunks) printf(" %c", c.cls);
    printf("\n");
    auto pool = [&](char cls, size_t skip_chunks) {            // whole chunks of one class, in scan order
        std::vector<hipMemGenericAllocationHandle_t> hs;
        size_t skipped = 0;
        for (auto& c : chunks) {
            if (c.cls != cls) continue;
            if (skipped < skip_chunks) { ++skipped; continue; }
            for (auto h : c.hs) if (hs.size() < kBuf / kHandle) hs.push_back(h);
        }
        return hs;
    };
    auto interleaved = [&](char c1, char c2, size_t skip) {
        auto p1 = pool(c1, skip), p2 = pool(c2, skip);
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t k = 0; k < kBuf / kHandle / 2 && k < p1.size() && k < p2.size(); ++k) { hs.push_back(p1[k]); hs.push_back(p2[k]); }
        return hs;
    };
    struct Named { std::string name; std::vector<hipMemGenericAllocationHandle_t> hs; };
    // inputs take the first chunks of a class, outputs the chunks after them (skip 4) so that a class can meet itself
    std::vector<Named> ins, outs;
    for (char cls : {'A', 'B', 'C', 'M'}) {
        ins.push_back({std::string(1, cls), pool(cls, 0)});
        outs.push_back({std::string(1, cls), pool(cls, 4)});
    }
    ins.push_back({"A|B", interleaved('A', 'B', 0)});            // chunks 0, 1 of each
    outs.push_back({"A|B", interleaved('A', 'B', 4)});           // chunks 4, 5 of each: no handle is mapped into an input and an output at once
    outs.push_back({"A|C", interleaved('A', 'C', 4)});
    outs.push_back({"B|C", interleaved('B', 'C', 4)});
    void* plain = nullptr;
    CK(hipMalloc(&plain, kBuf));
    CK(hipMemset(plain, 0, kBuf));
    const size_t need = kBuf / kHandle;
    printf("copy ms / C2C N=1024 ms, rows = input, columns = output (4 GiB each; '-': not enough chunks of that class)\n%10s", "");
    for (auto& o : outs) printf(" %13s", o.name.c_str());
    printf("\n");
    auto run_row = [&](const std::string& name, const void* in) {
        printf("%10s", name.c_str());
        for (auto& o : outs) {
            if (o.hs.size() < need) { printf(" %13s", "-"); continue; }
            char* out = map_handles(o.hs);
            const float copy = time_ms(6, [&] { smfft_copy_launch(in, out, (long long)(kBuf / 8), nullptr); });
            const float fft = time_ms(6, [&] { smfft_launch(0, 0, in, out, 1024, (int)(kBuf / 8 / 1024), 0, 1, nullptr); });
            printf("  %.3f/%.3f ", copy, fft);
            unmap(out, o.hs.size());
        }
        printf("\n");
    };
    for (auto& i : ins) {
        if (i.hs.size() < need) { printf("%10s  (not enough chunks)\n", i.name.c_str()); continue; }
        char* in = map_handles(i.hs);
        CK(hipMemset(in, 0, kBuf));
        run_row(i.name, in);
        unmap(in, i.hs.size());
    }
    run_row("hipMalloc", plain);
    return 0;
}
