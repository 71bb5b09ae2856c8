// tools/exp_vmm.hip — why does the same store kernel run at 4.7 .. 5.9 TB/s depending on which allocation it writes?
// Times a 14.7 GB non-temporal store sweep on buffers obtained in different ways:
//   A  hipMalloc, 6 buffers back to back (the round-1 "placement probe" situation)
//   B  hipMemCreate + hipMemMap, ONE physical handle of the whole size (minimum / recommended granularity)
//   C  hipMemCreate per 1 GiB / 256 MiB / 32 MiB chunk, mapped back to back into one VA range
//   D  hipMalloc of one big arena first-thing, sub-buffers carved at different offsets
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp_vmm.hip -o tools/exp_vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_store_nt(ulonglong2* out, size_t n16, unsigned long long v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    for (; i < n16; i += stride) { v2u64 y = {v + i, v ^ i}; __builtin_nontemporal_store(y, reinterpret_cast<v2u64*>(&out[i])); }
}
static double sweep(void* p, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_store_nt, dim3(262144), dim3(256), 0, 0, (ulonglong2*)p, bytes / 16, 7ull);
    hipEventRecord(e0);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_store_nt, dim3(262144), dim3(256), 0, 0, (ulonglong2*)p, bytes / 16, 7ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return bytes / (ms / 4) / 1e6;   // GB/s
}
int main() {
    const size_t bytes = 14737571840ull / (2u << 20) * (2u << 20);      // one chunk's witness, rounded to 2 MiB
    int dev = 0; CK(hipSetDevice(dev));
    size_t fr, tot; hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    {   // A
        std::vector<void*> bufs;
        for (int i = 0; i < 6; i++) { void* p; CK(hipMalloc(&p, bytes)); bufs.push_back(p); printf("A hipMalloc #%d  va %p : %.0f GB/s\n", i, p, sweep(p, bytes)); }
        // re-time the first after the others exist
        printf("A hipMalloc #0 again        : %.0f GB/s\n", sweep(bufs[0], bytes));
        for (void* p : bufs) hipFree(p);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int rep = 0; rep < 3; rep++) {   // B
        hipMemGenericAllocationHandle_t h; void* va;
        const size_t sz = (bytes + grec - 1) / grec * grec;
        CK(hipMemCreate(&h, sz, &prop, 0));
        CK(hipMemAddressReserve(&va, sz, 1ull << 30, nullptr, 0));
        CK(hipMemMap(va, sz, 0, h, 0)); CK(hipMemSetAccess(va, sz, &acc, 1));
        printf("B one handle (VA 1 GiB aligned) #%d va %p : %.0f GB/s\n", rep, va, sweep(va, bytes));
        hipMemUnmap(va, sz); hipMemRelease(h); hipMemAddressFree(va, sz);
    }
    for (size_t chunk : {1ull << 30, 256ull << 20, 32ull << 20, 2ull << 20}) {   // C
        const size_t n = (bytes + chunk - 1) / chunk, sz = n * chunk;
        if (n > 8192) continue;
        void* va; CK(hipMemAddressReserve(&va, sz, 1ull << 30, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        for (size_t i = 0; i < n; i++) { CK(hipMemCreate(&hs[i], chunk, &prop, 0)); CK(hipMemMap((char*)va + i * chunk, chunk, 0, hs[i], 0)); }
        CK(hipMemSetAccess(va, sz, &acc, 1));
        printf("C %4zu MiB chunks x %zu : %.0f GB/s\n", chunk >> 20, n, sweep(va, bytes));
        hipMemUnmap(va, sz); for (auto h : hs) hipMemRelease(h); hipMemAddressFree(va, sz);
    }
    {   // D
        void* arena; const size_t asz = 8 * bytes;
        CK(hipMalloc(&arena, asz));
        for (int i = 0; i < 8; i++) printf("D arena slice %d : %.0f GB/s\n", i, sweep((char*)arena + (size_t)i * bytes, bytes));
        hipFree(arena);
    }
    return 0;
}
