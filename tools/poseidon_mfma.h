// poseidon_mfma.h — Poseidon's MDS layer on the gfx950 matrix cores (round 5; device only).
//
// out[r] = sum_i M[r][i] * s[i] with M = circulant(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...) (poseidon.h) is a genuine
// 12 x 12 matrix-vector product per state with coefficients < 64.  v_mfma_i32_4x4x4_16b_i8 evaluates, for each of the wave's 64 lanes
// separately, D[i] += sum_{k<4} A[i][k] * B[k] (i < 4): sixteen independent 4 x 4 x 4 blocks of four lanes each, where a lane supplies
// its OWN four B bytes (one column) and the block's four lanes supply the four rows of A.  So with one state per lane (the layout of
// every Poseidon kernel here) and the state words cut into BYTES — the natural 8-bit limbs of a u64 —
//     limb j of out[4R + i] = sum_{g<3} sum_{k<4} M[4R + i][4g + k] * byte_j(s[4g + k])
// is three instructions per (limb position j, row group R): 72 per layer, on the matrix pipe, beside the VALU.  The rows of M are
// constants of the lane (lane l holds row l mod 4 of each 4 x 4 block; M is circulant, so only (g - R) mod 3 distinguishes blocks: three
// registers, a fourth for the block with the diagonal).  What stays on the VALU:
//   * operands: byte j of four words side by side = a 4 x 4 byte transposition of four dwords (8 v_perm_b32), 48 per layer;
//   * the instruction multiplies SIGNED bytes: every operand byte is offset by 128 (x ^ 0x80 = x - 128 as a signed byte, one v_xor per
//     operand register) and the accumulators start at 128 * (row sum of M) instead of 0, so the outputs are the true unsigned limbs;
//   * eight output limbs (< 2^17) per word recombined mod p: pairs by v_lshl_add (8-bit spacing), one v_mad_u64_u32 with a register-pair
//     addend for the 16-bit spacing, the top limb's overflow (weight 2^64 = 2^32 - 1) folded with a second multiply-add.
// Against the frequency-domain limb form (poseidon.h: 22-bit limbs, ~100 shift/adds per limb set x 3 + split + recombination, and in
// the partial rounds a carry normalisation of every lane): 252 instead of 480 (full round) / 410 (partial round) VALU instructions per layer.
//
// EXEC: the matrix instruction reads A from ALL four lanes of a block, also from lanes that are masked off.  Kernels that use this
// header compute MdsRows with every lane active (first statement of the kernel) and keep whole waves alive (no early return; stores
// are predicated instead).
#pragma once
#include "../blobstreamx_amd/csrc/poseidon.h"

#if defined(__HIP_DEVICE_COMPILE__)
namespace bsx {

typedef int bsx_v4i32 __attribute__((ext_vector_type(4)));

struct MdsRows { int a[3]; int a00; };

// row (lane mod 4) of the 4 x 4 block whose column group is d groups to the right of its row group: M[r][i] = C[(i - r) mod 12]
__device__ __forceinline__ MdsRows mds_rows() {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t row = __lane_id() & 3u;
    MdsRows A;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t packs[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            uint32_t p = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) p |= C[(4 * d + k - rr + 12) % 12] << (8 * k);
            packs[rr] = p;
        }
        uint32_t v = row == 0 ? packs[0] : row == 1 ? packs[1] : row == 2 ? packs[2] : packs[3];
        asm volatile("" : "+v"(v));          // opaque: never recomputed later under a partial EXEC mask
        A.a[d] = (int)v;
    }
    uint32_t v00 = (uint32_t)A.a[0] + (row == 0 ? 8u : 0u);       // the diagonal: M[0][0] = 17 + 8
    asm volatile("" : "+v"(v00));
    A.a00 = (int)v00;
    return A;
}

// eight limbs o[j] < 2^17 at 8-bit spacing -> sum_j o[j] 2^(8j) mod p (any u64 representative)
__device__ __forceinline__ uint64_t poseidon_recombine_bytes(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t o4, uint32_t o5,
                                                             uint32_t o6, uint32_t o7) {
    const uint32_t e0 = o0 + (o1 << 8), e1 = o2 + (o3 << 8), e2 = o4 + (o5 << 8), e3 = o6 + (o7 << 8);       // < 2^25.1
    const uint64_t X = (uint64_t)e1 * 65536u + ((uint64_t)e0 | ((uint64_t)e2 << 32));                        // < 2^58
    uint64_t Y, Z;
    const bool c1 = __builtin_add_overflow(X, (uint64_t)(e3 << 16) << 32, &Y);                               // e3's low 16 bits at 2^48
    const uint64_t top = (uint64_t)(e3 >> 16) + (c1 ? 1u : 0u);                                              // weight 2^64 = EPS; < 2^10
    const bool c2 = __builtin_add_overflow(Y, (top << 32) - top, &Z);
    return Z + gl_eps_if(c2);
}

__device__ __forceinline__ void poseidon_mds_mfma(uint64_t s[12], const MdsRows& A) {
    uint32_t y[2][3][4];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const uint32_t x0 = (uint32_t)(s[4 * g] >> (32 * h)), x1 = (uint32_t)(s[4 * g + 1] >> (32 * h));
            const uint32_t x2 = (uint32_t)(s[4 * g + 2] >> (32 * h)), x3 = (uint32_t)(s[4 * g + 3] >> (32 * h));
            // v_perm_b32 D, S0, S1, sel: selector 0-3 = bytes of S1, 4-7 = bytes of S0
            const uint32_t t01l = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t01h = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
            const uint32_t t23l = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t23h = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
            y[h][g][0] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u) ^ 0x80808080u;
            y[h][g][1] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u) ^ 0x80808080u;
            y[h][g][2] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u) ^ 0x80808080u;
            y[h][g][3] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u) ^ 0x80808080u;
        }
    constexpr int OFF = 128 * 256, OFF0 = 128 * 264;      // 128 x the row sums of M (the circulant row adds to 256; row 0 has the diagonal 8)
#pragma unroll
    for (int R = 0; R < 3; R++) {
        bsx_v4i32 acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = bsx_v4i32{R == 0 ? OFF0 : OFF, OFF, OFF, OFF};
#pragma unroll
            for (int g = 0; g < 3; g++) {
                const int a = (R == 0 && g == 0) ? A.a00 : A.a[(g - R + 3) % 3];
                acc[j] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, (int)y[j >> 2][g][j & 3], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            s[4 * R + i] = poseidon_recombine_bytes((uint32_t)acc[0][i], (uint32_t)acc[1][i], (uint32_t)acc[2][i], (uint32_t)acc[3][i],
                                                    (uint32_t)acc[4][i], (uint32_t)acc[5][i], (uint32_t)acc[6][i], (uint32_t)acc[7][i]);
    }
}

// rc: the 360 round constants; rcf: BSX_POSEIDON_FOLDED_TABLE (the partial rounds' constants folded forward, poseidon_consts.h)
template <bool FOLDED>
__device__ __forceinline__ void poseidon_permute_mfma(uint64_t s[12], const uint64_t* rc, const uint64_t* rcf) {
    const MdsRows A = mds_rows();
    int r = 0;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, rc + 12 * r);
        poseidon_mds_mfma(s, A);
    }
    for (int k = 0; k < POSEIDON_PARTIAL; k++, r++) {
        if (FOLDED) {
            s[0] = gl_pow7(gl_add_canon(s[0], rcf[k]));
        } else {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = gl_add_canon(s[i], rc[12 * r + i]);
            s[0] = gl_pow7(s[0]);
        }
        poseidon_mds_mfma(s, A);
    }
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, (FOLDED && k == 0) ? rcf + POSEIDON_PARTIAL : rc + 12 * r);
        poseidon_mds_mfma(s, A);
    }
}

}  // namespace bsx
#endif
