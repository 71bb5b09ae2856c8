// tools/gather_bench.hip — how fast can MI355X fetch RANDOM 128-byte lines (the access pattern of the fixed-key Ed25519 tables:
// one 128-byte affine entry per lane per step, picked by a scalar digit)?  Each lane does `steps` dependent-address-free loads of a
// full line (8 x 16 B) from a table of `mb` MB at pseudo-random line indices and xors them into a checksum.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_bench tools/gather_bench.hip && tools/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_gather(const uint4* __restrict__ tab, uint64_t n_lines, int steps, uint32_t seed, uint4* out, int dependent) {
    const uint64_t me = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = (me + 1) * 0x9E3779B97F4A7C15ull ^ seed;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < steps; s++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint64_t line = (x + (dependent ? acc.x : 0u)) % n_lines;
        const uint4* p = tab + line * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint4 v = p[k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    out[me] = acc;
}
int main() {
    const size_t sizes_mb[] = {64, 577, 2950};
    for (size_t mb : sizes_mb) {
        const uint64_t n_lines = mb * (1ull << 20) / 128;
        uint4 *tab, *out;
        hipMalloc(&tab, n_lines * 128);
        hipMemset(tab, 1, n_lines * 128);
        for (int lanes_k : {64, 205, 1049}) {             // thousands of lanes: 65,536 / 204,800 / 1,048,576
            const uint64_t lanes = (uint64_t)lanes_k * 1000 / 64 * 64;
            hipMalloc(&out, lanes * 16);
            for (int dep = 0; dep < 2; dep++) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                const int steps = 22;
                hipLaunchKernelGGL(k_gather, dim3(lanes / 64), dim3(64), 0, 0, tab, n_lines, steps, 1u, out, dep);
                hipEventRecord(e0);
                for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_gather, dim3(lanes / 64), dim3(64), 0, 0, tab, n_lines, steps, 7u + r, out, dep);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                printf("table %5zu MB  lanes %8llu  %s  %d lines/lane: %.3f ms  %.2f TB/s of 128-B lines  (%.1f G lines/s)\n", mb, (unsigned long long)lanes,
                       dep ? "dependent  " : "independent", steps, ms, lanes * steps * 128.0 / ms / 1e9, lanes * steps / ms / 1e6);
            }
            hipFree(out);
        }
        hipFree(tab);
    }
    return 0;
}
