// tools/glmul_asm_bench.hip — VERDICT r4 #3: a hand-written gfx950 body for the Goldilocks multiplication against the compiler's.
// gl_mul3_asm: three INDEPENDENT 64 x 64 -> mod p multiplications interleaved instruction by instruction in one asm block, so that
// every VALU write of an SGPR carry is two instructions away from its reader (the gfx940 VALU-writes-SGPR -> VALU-reads hazard needs no
// s_nop), the reduction's 64-bit adds are v_add_co / v_addc pairs whose carry is USED (the compiler recomputes it with v_cmp_*_u64 +
// v_cndmask) and hi_lo * EPS + t0 is ONE v_mad_u64_u32 with a 64-bit addend and carry-out.  19 VALU instructions per multiplication
// against the compiler's 21 + 2.5 s_nop.  Prints both rates and checks the results against each other.
//   hipcc --offload-arch=gfx950 -O3 -o tools/glmul_asm_bench tools/glmul_asm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../blobstreamx_amd/csrc/goldilocks.h"
using namespace bsx;

#define MUL_STEP(n, body) body
__device__ __forceinline__ void gl_mul3_asm(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t y0, uint64_t y1, uint64_t y2) {
    // per multiplication k: P (lo / t / result), Q, R, H, Z = {scratch, 0}, carry pair C
    uint64_t P0, Q0, R0, H0, Z0 = 0, C0, P1, Q1, R1, H1, Z1 = 0, C1, P2, Q2, R2, H2, Z2 = 0, C2;
    const uint32_t a00 = (uint32_t)x0, a01 = (uint32_t)(x0 >> 32), b00 = (uint32_t)y0, b01 = (uint32_t)(y0 >> 32);
    const uint32_t a10 = (uint32_t)x1, a11 = (uint32_t)(x1 >> 32), b10 = (uint32_t)y1, b11 = (uint32_t)(y1 >> 32);
    const uint32_t a20 = (uint32_t)x2, a21 = (uint32_t)(x2 >> 32), b20 = (uint32_t)y2, b21 = (uint32_t)(y2 >> 32);
    // sub-registers of a 64-bit operand cannot be named in inline asm, so the halves that are read separately travel through C++:
    // the block is split where a half is needed (the compiler only renames registers there, no instruction is emitted)
#define MAD0(P, C, a, b) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(P), "=s"(C) : "v"(a), "v"(b))
#define MADZ(D, C, a, b, Z) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(D), "=s"(C) : "v"(a), "v"(b), "v"(Z))
    MAD0(P0, C0, a00, b00); MAD0(P1, C1, a10, b10); MAD0(P2, C2, a20, b20);
    Z0 = (uint32_t)(P0 >> 32); Z1 = (uint32_t)(P1 >> 32); Z2 = (uint32_t)(P2 >> 32);
    MADZ(Q0, C0, a00, b01, Z0); MADZ(Q1, C1, a10, b11, Z1); MADZ(Q2, C2, a20, b21, Z2);
    Z0 = (uint32_t)Q0; Z1 = (uint32_t)Q1; Z2 = (uint32_t)Q2;
    MADZ(R0, C0, a01, b00, Z0); MADZ(R1, C1, a11, b10, Z1); MADZ(R2, C2, a21, b20, Z2);
    Z0 = (uint32_t)(Q0 >> 32); Z1 = (uint32_t)(Q1 >> 32); Z2 = (uint32_t)(Q2 >> 32);
    MADZ(H0, C0, a01, b01, Z0); MADZ(H1, C1, a11, b11, Z1); MADZ(H2, C2, a21, b21, Z2);
    H0 += (uint32_t)(R0 >> 32); H1 += (uint32_t)(R1 >> 32); H2 += (uint32_t)(R2 >> 32);
    uint32_t l00 = (uint32_t)P0, l01 = (uint32_t)R0, l10 = (uint32_t)P1, l11 = (uint32_t)R1, l20 = (uint32_t)P2, l21 = (uint32_t)R2;
    const uint32_t h00 = (uint32_t)H0, h01 = (uint32_t)(H0 >> 32), h10 = (uint32_t)H1, h11 = (uint32_t)(H1 >> 32), h20 = (uint32_t)H2, h21 = (uint32_t)(H2 >> 32);
    uint32_t m0, m1, m2;
    // t0 = lo - hi_hi; t0 -= EPS if it borrowed — three multiplications interleaved: a carry's reader is two instructions behind its writer
    asm volatile(
        "v_sub_co_u32 %0, %9, %0, %12\n\t"
        "v_sub_co_u32 %2, %10, %2, %13\n\t"
        "v_sub_co_u32 %4, %11, %4, %14\n\t"
        "v_subbrev_co_u32 %1, %9, 0, %1, %9\n\t"
        "v_subbrev_co_u32 %3, %10, 0, %3, %10\n\t"
        "v_subbrev_co_u32 %5, %11, 0, %5, %11\n\t"
        "v_cndmask_b32 %6, 0, -1, %9\n\t"
        "v_cndmask_b32 %7, 0, -1, %10\n\t"
        "v_cndmask_b32 %8, 0, -1, %11\n\t"
        "v_sub_co_u32 %0, %9, %0, %6\n\t"
        "v_sub_co_u32 %2, %10, %2, %7\n\t"
        "v_sub_co_u32 %4, %11, %4, %8\n\t"
        "v_subbrev_co_u32 %1, %9, 0, %1, %9\n\t"
        "v_subbrev_co_u32 %3, %10, 0, %3, %10\n\t"
        "v_subbrev_co_u32 %5, %11, 0, %5, %11\n\t"
        : "+v"(l00), "+v"(l01), "+v"(l10), "+v"(l11), "+v"(l20), "+v"(l21), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&s"(C0), "=&s"(C1), "=&s"(C2)
        : "v"(h01), "v"(h11), "v"(h21));
    uint64_t T0 = (uint64_t)l00 | ((uint64_t)l01 << 32), T1 = (uint64_t)l10 | ((uint64_t)l11 << 32), T2 = (uint64_t)l20 | ((uint64_t)l21 << 32);
    // U = hi_lo * (2^32 - 1) + t0 with the carry out of the multiply-add itself
    uint64_t U0, U1, U2;
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U0), "=s"(C0) : "v"(h00), "v"(T0));
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U1), "=s"(C1) : "v"(h10), "v"(T1));
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U2), "=s"(C2) : "v"(h20), "v"(T2));
    uint32_t u00 = (uint32_t)U0, u01 = (uint32_t)(U0 >> 32), u10 = (uint32_t)U1, u11 = (uint32_t)(U1 >> 32), u20 = (uint32_t)U2, u21 = (uint32_t)(U2 >> 32);
    asm volatile(
        "v_cndmask_b32 %6, 0, -1, %9\n\t"
        "v_cndmask_b32 %7, 0, -1, %10\n\t"
        "v_cndmask_b32 %8, 0, -1, %11\n\t"
        "v_add_co_u32 %0, %9, %0, %6\n\t"
        "v_add_co_u32 %2, %10, %2, %7\n\t"
        "v_add_co_u32 %4, %11, %4, %8\n\t"
        "v_addc_co_u32 %1, %9, 0, %1, %9\n\t"
        "v_addc_co_u32 %3, %10, 0, %3, %10\n\t"
        "v_addc_co_u32 %5, %11, 0, %5, %11\n\t"
        : "+v"(u00), "+v"(u01), "+v"(u10), "+v"(u11), "+v"(u20), "+v"(u21), "=&v"(m0), "=&v"(m1), "=&v"(m2), "+s"(C0), "+s"(C1), "+s"(C2));
    x0 = (uint64_t)u00 | ((uint64_t)u01 << 32);
    x1 = (uint64_t)u10 | ((uint64_t)u11 << 32);
    x2 = (uint64_t)u20 | ((uint64_t)u21 << 32);
}

template <int MODE>
__global__ void k(uint64_t* out, int iters) {
    uint64_t v[6];
    for (int i = 0; i < 6; i++) v[i] = (threadIdx.x + 3ull) * 0x9e3779b97f4a7c15ull + i * 0x100000001ull + blockIdx.x;
    for (int r = 0; r < iters; r++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = gl_mul(v[i], v[(i + 1) % 6] | 1);
        } else if (MODE == 1) {
            // the same six products: v[i] * (v[i+1] | 1) with the OLD v[i+1] for i = 0..4 sequentially dependent in MODE 0; here both halves use
            // the values as MODE 0 does: (0,1,2) then (3,4,5) — v[2] needs the old v[3], v[5] the NEW v[0]
            uint64_t y0 = v[1] | 1, y1 = v[2] | 1, y2 = v[3] | 1;
            // MODE 0 order: v0 = v0*v1; v1 = v1*v2; v2 = v2*v3 (all old right operands) ; then v3 = v3*v4, v4 = v4*v5, v5 = v5*v0(new)
            gl_mul3_asm(v[0], v[1], v[2], y0, y1, y2);
            uint64_t z0 = v[4] | 1, z1 = v[5] | 1, z2 = v[0] | 1;
            gl_mul3_asm(v[3], v[4], v[5], z0, z1, z2);
        } else if (MODE == 2) {      // the library's gl_mul3 (goldilocks.h: second pass, no zero-extended addends)
#if defined(__HIP_DEVICE_COMPILE__)
            uint64_t y0 = v[1] | 1, y1 = v[2] | 1, y2 = v[3] | 1;
            gl_mul3(v[0], v[1], v[2], y0, y1, y2);
            uint64_t z0 = v[4] | 1, z1 = v[5] | 1, z2 = v[0] | 1;
            gl_mul3(v[3], v[4], v[5], z0, z1, z2);
#endif
        }
    }
    uint64_t x = 0;
    for (int i = 0; i < 6; i++) x ^= gl_canonical(v[i]) * (i + 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    uint64_t *o0, *o1;
    hipMalloc(&o0, (size_t)blocks * threads * 8); hipMalloc(&o1, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    std::vector<uint64_t> h0((size_t)blocks * threads), h1(h0.size());
    int rc = 0;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, o0, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, o1, iters);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, o1, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        printf("%s: %.3f ms  %.1f G gl_mul/s\n", mode == 2 ? "gl_mul3 (library, 2nd pass) " : mode ? "gl_mul3_asm (first pass)    " : "gl_mul (compiler)           ", ms[mode], (double)blocks * threads * iters * 6 / ms[mode] / 1e6);
        if (mode == 0) hipMemcpy(h0.data(), o0, h0.size() * 8, hipMemcpyDeviceToHost);
        else {
            hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < h0.size(); i++) bad += h0[i] != h1[i];
            printf("  results %s (%zu of %zu differ)\n", bad ? "DIFFER" : "equal", bad, h0.size());
            rc |= bad != 0;
        }
    }
    return rc;
}
