// tools/microbench_alu.hip — measured integer-ALU ceilings of the box for the ALU-bound kernels (mode S, Poseidon):
// instruction issue rates (full-rate 32-bit op, v_mad_u32_u24, v_mad_u64_u32), then the arithmetic bodies the kernels are
// made of, alone at full occupancy: Goldilocks multiplications/s, Poseidon permutations/s, SHA-512 compressions/s,
// Curve25519 field multiplications/s.  `roofline.peak` of bench.py's mode-S and Poseidon legs comes from here
// (profiles/r2_microbench_alu.txt); not product code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_alu.hip -o tools/microbench_alu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../blobstreamx_amd/csrc/poseidon.h"
#include "../blobstreamx_amd/csrc/poseidon_consts.h"
#include "../blobstreamx_amd/csrc/sha512.h"
#include "../blobstreamx_amd/csrc/fe25519.h"
using namespace bsx;

__constant__ uint64_t RC[BSX_POSEIDON_TABLE_N] = {BSX_POSEIDON_TABLE};

// MODE 0: v_add_u32   1: v_mad_u32_u24   2: v_mad_u64_u32   3: v_mul_lo_u32   4: v_xor3 (bitop3)   5: v_lshl_add_u64
template <int MODE>
__global__ void k_issue(uint32_t* out, int iters) {
    uint32_t a[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; q[i] = a[i] * 0x100000001ull; }
    const uint32_t m = threadIdx.x | 3, k = blockIdx.x | 5;
    for (int r = 0; r < iters; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 1) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(k));
                if (MODE == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(m), "v"(k) : "vcc");
                if (MODE == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(m), "v"(k));
                if (MODE == 5) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (MODE == 6) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[i]));
                if (MODE == 7) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(k));
                if (MODE == 8) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(m), "v"(k));
                if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (MODE == 10) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc");
            }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// BODY 0: gl_mul (4 independent chains)  1: poseidon_permute  2: sha512_compress  3: fe_mul  4: poseidon_mds only  5: gl_pow7 x12
//      6: fe_sq
template <int BODY>
__global__ void k_body(uint32_t* out, int iters) {
    uint32_t x = 0;
    if (BODY == 0) {
        uint64_t v[4] = {threadIdx.x + 3ull, blockIdx.x * 0x9e3779b97f4a7c15ull + 1, threadIdx.x * 0x100000001ull + 7, 0xdeadbeefcafef00dull};
        for (int r = 0; r < iters; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = gl_mul(v[i], v[(i + 1) & 3] | 1);
        for (int i = 0; i < 4; i++) x ^= (uint32_t)v[i] ^ (uint32_t)(v[i] >> 32);
    } else if (BODY == 1 || BODY == 4 || BODY == 5) {
        uint64_t s[12];
        for (int i = 0; i < 12; i++) s[i] = threadIdx.x * 0x9e3779b97f4a7c15ull + i + blockIdx.x;
        for (int r = 0; r < iters; r++) {
            if (BODY == 1) poseidon_permute(s, RC);
            if (BODY == 4) poseidon_mds(s);
            if (BODY == 5)
#pragma unroll
                for (int i = 0; i < 12; i++) s[i] = gl_pow7(s[i]);
        }
        for (int i = 0; i < 12; i++) x ^= (uint32_t)s[i] ^ (uint32_t)(s[i] >> 32);
    } else if (BODY == 2) {
        uint64_t st[8], w[16];
        for (int i = 0; i < 8; i++) st[i] = threadIdx.x * 31 + i + blockIdx.x;
        for (int i = 0; i < 16; i++) w[i] = threadIdx.x * 17 + i;
        for (int r = 0; r < iters; r++) { uint64_t ww[16]; for (int i = 0; i < 16; i++) ww[i] = w[i] ^ st[i & 7]; sha512_compress(st, ww); }
        for (int i = 0; i < 8; i++) x ^= (uint32_t)st[i] ^ (uint32_t)(st[i] >> 32);
    } else {
        fe f, g;
        for (int i = 0; i < 10; i++) { f.v[i] = (int32_t)((threadIdx.x * 2654435761u + i * 40503u) & 0x1ffffff); g.v[i] = (int32_t)((blockIdx.x * 97u + i * 7919u + threadIdx.x) & 0x1ffffff); }
        for (int r = 0; r < iters; r++) {
            if (BODY == 3) { f = fe_mul(f, g); g = fe_mul(g, f); }
            else { f = fe_sq(f); g = fe_sq(g); }
        }
        for (int i = 0; i < 10; i++) x ^= (uint32_t)f.v[i] ^ (uint32_t)g.v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <typename F>
static float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
    uint32_t* out; hipMalloc(&out, 64u << 20);
    const char* names[] = {"v_add_u32", "v_mad_u32_u24", "v_mad_u64_u32", "v_mul_lo_u32", "v_bitop3_b32", "v_lshl_add_u64",
                           "v_alignbit_b32", "v_add3_u32", "v_bfi_b32", "v_cndmask_b32", "v_add_co_u32"};
    const int blocks = 256 * 8, threads = 256, iters = 200;          // 8 waves per SIMD
    auto issue = [&](int mode) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k_issue<0>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 1: hipLaunchKernelGGL(k_issue<1>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 2: hipLaunchKernelGGL(k_issue<2>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 3: hipLaunchKernelGGL(k_issue<3>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 4: hipLaunchKernelGGL(k_issue<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 5: hipLaunchKernelGGL(k_issue<5>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 6: hipLaunchKernelGGL(k_issue<6>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 7: hipLaunchKernelGGL(k_issue<7>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 8: hipLaunchKernelGGL(k_issue<8>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            case 9: hipLaunchKernelGGL(k_issue<9>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            default: hipLaunchKernelGGL(k_issue<10>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
        }
    };
    for (int mode = 0; mode < 11; mode++) {
        float ms = timeit([&] { issue(mode); }, 3);
        double ops = (double)blocks * threads * iters * 64;
        printf("issue %-15s : %.3f ms  %.2f T lane-ops/s\n", names[mode], ms, ops / ms / 1e9);
    }
    const char* bn[] = {"gl_mul (Goldilocks 64x64 mod p)", "poseidon_permute", "sha512_compress", "fe_mul (Curve25519, 10 limbs)", "poseidon_mds", "gl_pow7 x 12",
                        "fe_sq (Curve25519)"};
    const double per_iter[] = {4, 1, 1, 2, 1, 12, 2};
    const int it[] = {2000, 40, 200, 400, 1000, 200, 400};
    for (int body = 0; body < 7; body++)
        for (int wps : {1, 2, 4, 8}) {
            const int nb = 256 * wps;
            auto run = [&] {
                switch (body) {
                    case 0: hipLaunchKernelGGL(k_body<0>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    case 1: hipLaunchKernelGGL(k_body<1>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    case 2: hipLaunchKernelGGL(k_body<2>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    case 3: hipLaunchKernelGGL(k_body<3>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    case 4: hipLaunchKernelGGL(k_body<4>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    case 5: hipLaunchKernelGGL(k_body<5>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                    default: hipLaunchKernelGGL(k_body<6>, dim3(nb), dim3(256), 0, 0, out, it[body]); break;
                }
            };
            float ms = timeit(run, 3);
            double n = (double)nb * 256 * it[body] * per_iter[body];
            printf("%-34s waves/SIMD=%d : %.3f ms  %.3f G/s\n", bn[body], wps, ms, n / ms / 1e6);
        }
    return 0;
}
