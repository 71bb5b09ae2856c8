// tools/poseidon_mfma_bench.hip — experiment (round 5): Poseidon's MDS layer on the matrix cores.
//
// The MDS layer is a genuine 12 x 12 matrix-vector product with coefficients < 64 (poseidon.h), so far evaluated on the VALU in
// 22-bit limbs (frequency-domain form: ~300 shift/add operations per layer + limb split / recombination + carry normalisation in the
// partial rounds = 480 of a full round's 1,500 instructions, 410 of a partial round's 527).  v_mfma_i32_4x4x4_16b_i8 computes, for each of
// the wave's 64 lanes separately, D[i] += sum_k A[i][k] * B[k] (i, k < 4): 16 blocks of 4 lanes, the lane's OWN four B bytes against
// four rows of A held by the block's four lanes.  With the state's words cut into BYTES (the natural 8-bit limbs of a u64) the whole
// layer is 8 limb positions x 3 row groups x 3 column groups = 72 of these instructions per state-per-lane wave, running on the matrix
// pipe beside the VALU; what stays on the VALU is the 4 x 4 byte transposition of the operands (v_perm_b32), the signed-byte offset
// (x ^ 0x80, corrected through the accumulator's initial value) and the recombination of eight 17-bit output limbs per word.
//
// This tool checks the layout assumptions of the instruction on the device, the permutation built on it against the library's
// (poseidon.h), and times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o tools/poseidon_mfma_bench tools/poseidon_mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../blobstreamx_amd/csrc/poseidon.h"
#include "../blobstreamx_amd/csrc/poseidon_consts.h"
#include "poseidon_mfma.h"
using namespace bsx;

__constant__ uint64_t RC[BSX_POSEIDON_TABLE_N] = {BSX_POSEIDON_TABLE};
__constant__ uint64_t RCF[BSX_POSEIDON_FOLDED_N] = {BSX_POSEIDON_FOLDED_TABLE};

// MODE 0: library permutation; 1: MFMA MDS in every layer, original constants; 2: MFMA MDS + folded partial-round constants
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_perm(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int iters) {
    const uint64_t me = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canonical(in[me * 12 + i]);
#if defined(__HIP_DEVICE_COMPILE__)
    for (int r = 0; r < iters; r++) {
        if (MODE == 0) poseidon_permute(s, RC);
        else if (MODE == 1) poseidon_permute_mfma<false>(s, RC, RCF);
        else poseidon_permute_mfma<true>(s, RC, RCF);
    }
#endif
#pragma unroll
    for (int i = 0; i < 12; i++) out[me * 12 + i] = gl_canonical(s[i]);
}

// one MDS layer alone, both forms (layout check)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mds(const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint64_t me = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = in[me * 12 + i];
#if defined(__HIP_DEVICE_COMPILE__)
    if (MODE == 0) poseidon_mds(s);
    else { const MdsRows A = mds_rows(); poseidon_mds_mfma(s, A); }
#endif
#pragma unroll
    for (int i = 0; i < 12; i++) out[me * 12 + i] = gl_canonical(s[i]);
}

static uint64_t rnd(uint64_t& x) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }

int main() {
    const int blocks = 256 * 12, threads = 256;
    const size_t n = (size_t)blocks * threads;
    std::vector<uint64_t> h(n * 12);
    uint64_t seed = 0x1234567887654321ull;
    for (size_t i = 0; i < h.size(); i++) {
        const uint64_t v = rnd(seed);
        const int kind = (int)(i % 97);
        h[i] = kind == 0 ? 0 : kind == 1 ? ~0ull : kind == 2 ? GL_P - 1 : kind == 3 ? GL_P : kind == 4 ? 0x8080808080808080ull : kind == 5 ? 0x7f7f7f7f7f7f7f7full : v;
    }
    uint64_t *din, *d0, *d1;
    hipMalloc(&din, h.size() * 8); hipMalloc(&d0, h.size() * 8); hipMalloc(&d1, h.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint64_t> r0(h.size()), r1(h.size());
    int rc = 0;
    // 1. one MDS layer
    hipLaunchKernelGGL(k_mds<0>, dim3(blocks), dim3(threads), 0, 0, din, d0);
    hipLaunchKernelGGL(k_mds<1>, dim3(blocks), dim3(threads), 0, 0, din, d1);
    hipDeviceSynchronize();
    hipMemcpy(r0.data(), d0, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); i++) {
        if (r0[i] != r1[i]) { if (bad < 8) printf("  mds differs at state %zu word %zu: %016llx vs %016llx (in %016llx)\n", i / 12, i % 12, (unsigned long long)r0[i], (unsigned long long)r1[i], (unsigned long long)h[i]); bad++; }
    }
    printf("MDS layer: %s (%zu of %zu words differ)\n", bad ? "DIFFERS" : "equal", bad, h.size());
    rc |= bad != 0;
    // 2. permutations
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 8;
    float ms[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; mode++) {
        uint64_t* dst = mode == 0 ? d0 : d1;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_perm<0>, dim3(blocks), dim3(threads), 0, 0, din, dst, iters);
            else if (mode == 1) hipLaunchKernelGGL(k_perm<1>, dim3(blocks), dim3(threads), 0, 0, din, dst, iters);
            else hipLaunchKernelGGL(k_perm<2>, dim3(blocks), dim3(threads), 0, 0, din, dst, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        printf("mode %d (%s): %.3f ms  %.3f G permutations/s\n", mode, mode == 0 ? "library: VALU limb MDS" : mode == 1 ? "MFMA MDS" : "MFMA MDS + folded partial-round constants",
               ms[mode], (double)n * iters / ms[mode] / 1e6);
        if (mode > 0) {
            hipMemcpy(r0.data(), d0, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, h.size() * 8, hipMemcpyDeviceToHost);
            bad = 0;
            for (size_t i = 0; i < h.size(); i++) bad += r0[i] != r1[i];
            printf("  permutation x %d vs library: %s (%zu of %zu words differ)\n", iters, bad ? "DIFFERS" : "equal", bad, h.size());
            rc |= bad != 0;
        }
    }
    return rc;
}
