// tools/h2d_bench.hip — what PCIe gives the coalescing front end: H2D rate of page-locked memory by copy size, by the number of
// streams copying concurrently, and for a kernel that reads the page-locked memory itself (zero copy).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_read(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t TOTAL = 256u << 20;
    uint8_t *h, *d;
    CK(hipHostMalloc((void**)&h, TOTAL, hipHostMallocDefault));
    CK(hipMalloc((void**)&d, TOTAL));
    memset(h, 1, TOTAL);
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t sz : {(size_t)64 << 10, (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
        for (int ns : {1, 2, 4}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                const double t0 = now();
                size_t off = 0; int k = 0;
                while (off + sz <= TOTAL) { CK(hipMemcpyAsync(d + off, h + off, sz, hipMemcpyHostToDevice, st[k % ns])); off += sz; k++; }
                for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(st[i]));
                const double t = now() - t0;
                if (t < best) best = t;
            }
            printf("copy %6zu KB x %d streams: %.1f GB/s\n", sz >> 10, ns, TOTAL / best / 1e9);
        }
    }
    for (int blocks : {64, 256, 1024, 4096}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st[0], (const uint4*)h, (uint4*)d, TOTAL / 16);
            CK(hipStreamSynchronize(st[0]));
            const double t = now() - t0;
            if (t < best) best = t;
        }
        printf("kernel zero-copy read, %d blocks: %.1f GB/s\n", blocks, TOTAL / best / 1e9);
    }
    // pageable source, as a caller's plain memory
    std::vector<uint8_t> pg(TOTAL, 2);
    for (size_t sz : {(size_t)1 << 20, (size_t)16 << 20}) {
        const double t0 = now();
        for (size_t off = 0; off + sz <= TOTAL; off += sz) CK(hipMemcpyAsync(d + off, pg.data() + off, sz, hipMemcpyHostToDevice, st[0]));
        CK(hipStreamSynchronize(st[0]));
        printf("pageable copy %6zu KB: %.1f GB/s\n", sz >> 10, TOTAL / (now() - t0) / 1e9);
    }
    // host memcpy into page-locked memory (the staging copy), one thread
    {
        const double t0 = now();
        memcpy(h, pg.data(), TOTAL);
        printf("host memcpy pageable -> page-locked, 1 thread: %.1f GB/s\n", TOTAL / (now() - t0) / 1e9);
    }
    return 0;
}
