// Where do different allocation APIs put 4 GiB buffers?  Stream-copy time between buffers obtained from hipMalloc,
// hipMallocAsync (stream-ordered pool), hipMallocManaged (+prefetch to the device) and hipExtMallocWithFlags variants,
// all made at the start of a fresh process.  Run from the repository root (uses libsmfft_amd.so's copy kernel).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
int main() {
    const size_t G = 1ull << 30, bytes = 4 * G;
    void* lib = dlopen("smfft_amd/libsmfft_amd.so", RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    auto copy = (int (*)(const void*, void*, long long, void*))dlsym(lib, "smfft_copy_launch");
    CK(hipSetDevice(0));
    std::vector<void*> cand; std::vector<const char*> kind;
    auto add = [&](void* p, const char* k) { cand.push_back(p); kind.push_back(k); };
    void* p;
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&p, bytes)); add(p, "malloc"); }
    for (int i = 0; i < 2; ++i) { if (hipMallocAsync(&p, bytes, 0) == hipSuccess) add(p, "async"); else { printf("hipMallocAsync failed\n"); (void)hipGetLastError(); } }
    CK(hipStreamSynchronize(0));
    for (int i = 0; i < 2; ++i) {
        if (hipMallocManaged(&p, bytes) == hipSuccess && hipMemPrefetchAsync(p, bytes, 0, 0) == hipSuccess) add(p, "managed");
        else { printf("managed failed\n"); (void)hipGetLastError(); }
    }
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) { if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) add(p, "uncached"); else (void)hipGetLastError(); }
    for (int i = 0; i < 2; ++i) { if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) add(p, "finegr"); else (void)hipGetLastError(); }
    {   // explicit virtual-memory management: physical handle + reserved address range + mapping
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (int i = 0; i < 2; ++i) {
            hipMemGenericAllocationHandle_t h;
            void* va = nullptr;
            if (hipMemCreate(&h, bytes, &prop, 0) == hipSuccess && hipMemAddressReserve(&va, bytes, 0, nullptr, 0) == hipSuccess
                && hipMemMap(va, bytes, 0, h, 0) == hipSuccess && hipMemSetAccess(va, bytes, &acc, 1) == hipSuccess) add(va, "vmm");
            else { printf("vmm allocation failed: %s\n", hipGetErrorString(hipGetLastError())); }
        }
    }
    for (auto q : cand) CK(hipMemset(q, 0, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto probe = [&](void* in, void* out) {
        copy(in, out, (long long)(bytes / 8), nullptr);
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) copy(in, out, (long long)(bytes / 8), nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
    };
    printf("         "); for (auto k : kind) printf(" %8s", k); printf("   (columns = output)\n");
    for (size_t i = 0; i < cand.size(); ++i) {
        printf("%-9s", kind[i]);
        for (size_t o = 0; o < cand.size(); ++o) { if (i == o) printf("      -  "); else printf(" %8.3f", probe(cand[i], cand[o])); }
        printf("\n");
    }
    return 0;
}
