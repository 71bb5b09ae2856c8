// Does gfx950 under ROCm serve misaligned global dword / dwordx4 accesses (2-byte aligned addresses)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const uint8_t* src, uint8_t* dst, int off_s, int off_d, int n16) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = *reinterpret_cast<const uint4*>(src + off_s + 16 * i);
    *reinterpret_cast<uint4*>(dst + off_d + 16 * i) = v;
}
__global__ void k4(const uint8_t* src, uint8_t* dst, int off_s, int off_d, int n4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const uint32_t v = *reinterpret_cast<const uint32_t*>(src + off_s + 4 * i);
    *reinterpret_cast<uint32_t*>(dst + off_d + 4 * i) = v;
}
int main() {
    const int N = 1 << 20;
    std::vector<uint8_t> h(N + 64), out(N + 64);
    for (int i = 0; i < N + 64; i++) h[i] = (uint8_t)(i * 131 + 7);
    uint8_t *ds, *dd;
    hipMalloc(&ds, N + 64); hipMalloc(&dd, N + 64);
    hipMemcpy(ds, h.data(), N + 64, hipMemcpyHostToDevice);
    int bad = 0;
    for (int os : {0, 2, 4, 6, 10}) for (int od : {0, 2, 6, 14, 1, 3}) {
        hipMemset(dd, 0, N + 64);
        k<<<N / 16 / 256, 256>>>(ds, dd, os, od, N / 16);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(out.data(), dd, N + 64, hipMemcpyDeviceToHost);
        bool ok = e == hipSuccess && memcmp(out.data() + od, h.data() + os, N) == 0;
        hipMemset(dd, 0, N + 64);
        k4<<<N / 4 / 256, 256>>>(ds, dd, os, od, N / 4);
        e = hipDeviceSynchronize();
        hipMemcpy(out.data(), dd, N + 64, hipMemcpyDeviceToHost);
        bool ok4 = e == hipSuccess && memcmp(out.data() + od, h.data() + os, N) == 0;
        printf("src+%d dst+%d: x4 %s, dword %s\n", os, od, ok ? "ok" : "FAIL", ok4 ? "ok" : "FAIL");
        bad += !ok + !ok4;
    }
    printf("%s\n", bad ? "UNALIGNED NOT SUPPORTED" : "unaligned global access works");
    return bad != 0;
}
