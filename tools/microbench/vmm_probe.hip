// Can the memory in front of the fast write region be skipped cheaply?  Reserves most of the free memory as
// PHYSICAL allocations only (hipMemCreate, never mapped), then hipMallocs a few 4 GiB candidates, which therefore come
// from the end of the allocation order, and times the library's stream copy between them.  Prints what each step costs.
// Build: hipcc -O3 --offload-arch=gfx950 vmm_probe.hip -o vmm_probe -ldl ; run from the repository root.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t G = 1ull << 30, bytes = 4 * G;
    const int ncand = argc > 1 ? atoi(argv[1]) : 10;
    void* lib = dlopen("smfft_amd/libsmfft_amd.so", RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    auto copy = (int (*)(const void*, void*, long long, void*))dlsym(lib, "smfft_copy_launch");
    CK(hipSetDevice(0));
    size_t free_mem, total;
    CK(hipMemGetInfo(&free_mem, &total));
    printf("free %.1f GiB of %.1f\n", free_mem / (double)G, total / (double)G);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    // ballast: physical handles of 16 GiB, leaving room for the candidates + 6 GiB
    std::vector<hipMemGenericAllocationHandle_t> ballast;
    const size_t keep = (size_t)ncand * bytes + 6 * G;
    double t0 = now();
    while (true) {
        CK(hipMemGetInfo(&free_mem, &total));
        if (free_mem < keep + 16 * G) break;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, 16 * G, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        ballast.push_back(h);
    }
    printf("ballast: %zu x 16 GiB physical handles in %.3f s\n", ballast.size(), now() - t0);
    t0 = now();
    std::vector<void*> cand;
    for (int i = 0; i < ncand; ++i) { void* p; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; } cand.push_back(p); }
    printf("%zu candidates hipMalloc'ed in %.3f s\n", cand.size(), now() - t0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto probe = [&](void* in, void* out) {
        copy(in, out, (long long)(bytes / 8), nullptr);
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) copy(in, out, (long long)(bytes / 8), nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
    };
    printf("copy ms, rows = input candidate, columns = output candidate (allocation order):\n");
    for (size_t i = 0; i < cand.size(); ++i) {
        for (size_t o = 0; o < cand.size(); ++o) printf(i == o ? "    -  " : " %6.3f", probe(cand[i], cand[o]));
        printf("\n");
    }
    t0 = now();
    for (auto h : ballast) CK(hipMemRelease(h));
    for (auto p : cand) CK(hipFree(p));
    printf("release: %.3f s\n", now() - t0);
    return 0;
}
