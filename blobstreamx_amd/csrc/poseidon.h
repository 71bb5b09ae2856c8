// poseidon.h — plonky2's Poseidon over Goldilocks for one lane.
//
// What the reference reaches through `PoseidonGoldilocksConfig` (DefaultParameters, bin/header_range_2048.rs:1-17;
// builder.build()/prove and the recursion inside mapreduce, circuits/builder.rs:301-302); the implementation lives in
// plonky2 53c5bc3e (Cargo.lock:3110-3112) [UPSTREAM], restated here from its public definition:
//   state width 12 (rate 8, capacity 4), 4 full + 22 partial + 4 full rounds, S-box x^7,
//   round: add constants -> S-box (all lanes / lane 0) -> MDS, MDS = circulant(17,15,41,16,2,28,13,13,39,18,34,20)
//   + diag(8,0,...,0): out[r] = sum_i s[(i + r) % 12] * C[i] + D[r] * s[r];
//   hash_n_to_hash_no_pad: overwrite-mode sponge, 8 elements per absorb, digest = state[0..4);
//   two_to_one(l, r) = permute(l ‖ r ‖ 0000)[0..4); hash_or_noop: <= 4 elements are the digest itself (zero padded).
// Round constants: poseidon_consts.h (regenerated, see tools/gen_poseidon_constants.py).  PARITY: nothing under
// /root/reference holds a Poseidon value; pinned to plonky2's public test vectors (tests/test_oracle_poseidon.py).
//
// MDS on gfx950: every state word is split into 22/22/20-bit limbs; a limb times a coefficient (< 2^6) summed over a
// row (coefficients add to 284 < 2^9) stays below 2^31, so each limb set is an exact 32-bit integer problem, solved
// without multiplications (poseidon_mds_limbs below); a row is recombined and reduced mod p once.
#pragma once
#include "goldilocks.h"
#include "goldilocks_sbox_asm.h"

namespace bsx {

constexpr int POSEIDON_WIDTH = 12, POSEIDON_RATE = 8, POSEIDON_FULL_HALF = 4, POSEIDON_PARTIAL = 22, POSEIDON_ROUNDS = 30;

// The circulant part is a length-12 cyclic convolution: out = s (*) c' with c'[m] = C[-m mod 12].  By the CRT over
// x^12 - 1 = prod_{w^4 = 1} (x^3 - w) it splits into a 4-point DFT over the stride-3 sub-sequences (twiddles 1, i, -1, -i:
// additions only), three 3 x 3 products modulo x^3 - w, and the inverse DFT.  For THIS matrix the transformed
// coefficients are (after moving the inverse transform's 1/4 into them)
//     w =  1: (16, 32, 16)        w = -1: (-1, -8, 2)        w = i: (2 + i, -4 - i, 16 - i)
// — all +-2^k, so one limb set costs ~100 shift/add/sub operations in 32-bit wrap-around arithmetic instead of 144
// multiply-adds (the idea of plonky2's mds_multiply_freq, re-derived here from the matrix; checked against the
// definition in tests/test_oracle_poseidon.py::test_mds_layer_vs_definition).  Exact because every true output is < 284 * 2^22 < 2^31.
BSX_HDI void poseidon_mds_limbs(const uint32_t l[12], uint32_t out[12]) {
    constexpr int32_t K1[3] = {16, 32, 16}, KM[3] = {-1, -8, 2}, KR[3] = {2, -4, 16}, KI[3] = {1, -1, -1};
    uint32_t f1[3], fm[3], fr[3], fi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t t0 = l[k] + l[k + 6], t1 = l[k + 3] + l[k + 9];
        f1[k] = t0 + t1; fm[k] = t0 - t1; fr[k] = l[k] - l[k + 6]; fi[k] = l[k + 3] - l[k + 9];
    }
    uint32_t o1[3] = {0, 0, 0}, om[3] = {0, 0, 0}, orr[3] = {0, 0, 0}, oi[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int k = (a + b) % 3;
            const bool wrap = a + b >= 3;                 // x^3 = w
            o1[k] += f1[a] * (uint32_t)K1[b];
            const uint32_t pm = fm[a] * (uint32_t)KM[b];
            om[k] += wrap ? 0u - pm : pm;
            const uint32_t pr = fr[a] * (uint32_t)KR[b] - fi[a] * (uint32_t)KI[b];
            const uint32_t pi = fr[a] * (uint32_t)KI[b] + fi[a] * (uint32_t)KR[b];
            orr[k] += wrap ? 0u - pi : pr;                // times i: (pr, pi) -> (-pi, pr)
            oi[k] += wrap ? pr : pi;
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t p = o1[k] + om[k], q = o1[k] - om[k];
        out[k] = p + orr[k]; out[k + 3] = q + oi[k]; out[k + 6] = p - orr[k]; out[k + 9] = q - oi[k];
    }
    out[0] += l[0] * 8u;                                  // diagonal
}

// a0 + 2^22 a1 + 2^44 a2 reduced to a u64 representative, for a0 >= 0, a0 + (a1 << 22) < 2^64 wrap-free in 32-bit pieces:
// a0, a1 < 2^31, a2 < 2^29 (every caller: MDS outputs of 22 / 22 / 20-bit limbs).  The part of a2 above bit 20 and the carry out of the
// upper dword have weight 2^64 = EPS.  Device: carries are used where they fall (v_add_co / v_addc), top * EPS + x is one
// v_mad_u64_u32 with its own carry-out: 12 issue slots against the 22 of the compiled 64-bit form.
BSX_HDI uint64_t poseidon_recombine(uint32_t a0, uint32_t a1, uint32_t a2) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo, hi, top, m;
    uint64_t C, U;
    asm volatile(
        "v_lshlrev_b32 %0, 22, %5\n\t"
        "v_lshrrev_b32 %1, 10, %5\n\t"
        "v_lshrrev_b32 %2, 20, %6\n\t"
        "v_add_co_u32 %0, %3, %0, %4\n\t"
        "v_lshlrev_b32 %4, 12, %6\n\t"              // a0's register is free from here on (early-clobber scratch below)
        "s_nop 0\n\t"
        "v_addc_co_u32 %1, %3, %1, %4, %3\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %2, %3, 0, %2, %3\n\t"
        : "=&v"(lo), "=&v"(hi), "=&v"(top), "=&s"(C), "+v"(a0)
        : "v"(a1), "v"(a2));
    const uint64_t X = (uint64_t)lo | ((uint64_t)hi << 32);
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U), "=s"(C) : "v"(top), "v"(X));
    uint32_t u0 = (uint32_t)U, u1 = (uint32_t)(U >> 32);
    asm volatile(
        "s_nop 0\n\t"
        "v_cndmask_b32 %2, 0, -1, %3\n\t"
        "v_add_co_u32 %0, %3, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
        : "+v"(u0), "+v"(u1), "=&v"(m), "+s"(C));
    return (uint64_t)u0 | ((uint64_t)u1 << 32);
#else
    const uint64_t x = (uint64_t)a0 + ((uint64_t)a1 << 22);
    const uint64_t y = (uint64_t)(a2 & 0xfffffu) << 44;
    uint64_t lo, t;
    const bool c1 = __builtin_add_overflow(x, y, &lo);
    const uint64_t hi = (uint64_t)(a2 >> 20) + (c1 ? 1u : 0u);                     // < 2^13
    const bool c2 = __builtin_add_overflow(lo, (hi << 32) - hi, &t);
    return t + gl_eps_if(c2);
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
// three words at once, interleaved instruction by instruction: every carry is read three instructions behind its writer (no s_nop)
__device__ __forceinline__ void poseidon_recombine3(uint64_t& w0, uint64_t& w1, uint64_t& w2, uint32_t a00, uint32_t a10, uint32_t a20,
                                                    uint32_t a01, uint32_t a11, uint32_t a21, uint32_t a02, uint32_t a12, uint32_t a22) {
    // a{limb}{word}
    uint32_t lo0, hi0, top0, lo1, hi1, top1, lo2, hi2, top2, m0, m1, m2;
    uint64_t C0, C1, C2, U0, U1, U2;
    asm volatile(
        "v_lshlrev_b32 %0, 22, %15\n\t"
        "v_lshlrev_b32 %3, 22, %17\n\t"
        "v_lshlrev_b32 %6, 22, %19\n\t"
        "v_add_co_u32 %0, %9, %0, %12\n\t"
        "v_add_co_u32 %3, %10, %3, %13\n\t"
        "v_add_co_u32 %6, %11, %6, %14\n\t"
        "v_lshrrev_b32 %1, 10, %15\n\t"
        "v_lshrrev_b32 %4, 10, %17\n\t"
        "v_lshrrev_b32 %7, 10, %19\n\t"
        "v_lshlrev_b32 %12, 12, %16\n\t"
        "v_lshlrev_b32 %13, 12, %18\n\t"
        "v_lshlrev_b32 %14, 12, %20\n\t"
        "v_addc_co_u32 %1, %9, %1, %12, %9\n\t"
        "v_addc_co_u32 %4, %10, %4, %13, %10\n\t"
        "v_addc_co_u32 %7, %11, %7, %14, %11\n\t"
        "v_lshrrev_b32 %2, 20, %16\n\t"
        "v_lshrrev_b32 %5, 20, %18\n\t"
        "v_lshrrev_b32 %8, 20, %20\n\t"
        "v_addc_co_u32 %2, %9, 0, %2, %9\n\t"
        "v_addc_co_u32 %5, %10, 0, %5, %10\n\t"
        "v_addc_co_u32 %8, %11, 0, %8, %11\n\t"
        : "=&v"(lo0), "=&v"(hi0), "=&v"(top0), "=&v"(lo1), "=&v"(hi1), "=&v"(top1), "=&v"(lo2), "=&v"(hi2), "=&v"(top2),
          "=&s"(C0), "=&s"(C1), "=&s"(C2), "+v"(a00), "+v"(a01), "+v"(a02)
        : "v"(a10), "v"(a20), "v"(a11), "v"(a21), "v"(a12), "v"(a22));
    const uint64_t X0 = (uint64_t)lo0 | ((uint64_t)hi0 << 32), X1 = (uint64_t)lo1 | ((uint64_t)hi1 << 32), X2 = (uint64_t)lo2 | ((uint64_t)hi2 << 32);
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U0), "=s"(C0) : "v"(top0), "v"(X0));
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U1), "=s"(C1) : "v"(top1), "v"(X1));
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(U2), "=s"(C2) : "v"(top2), "v"(X2));
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
    w0 = (uint64_t)u00 | ((uint64_t)u01 << 32);
    w1 = (uint64_t)u10 | ((uint64_t)u11 << 32);
    w2 = (uint64_t)u20 | ((uint64_t)u21 << 32);
}
#endif
// the same for limbs out of the partial rounds, whose lowest one may be slightly negative (see below): carry steps first
// (two: a1 + carry is negative when a1 = 0 and a0 < 0 — every l1 of the state zero — and the value sits in a2).  |a0|, a1 < 2^31 - 2^9:
// the MDS outputs of the partial rounds (|a0| < 264 * 2^22, a1 < 264 * (2^22 + 2^18.1))
BSX_HDI void poseidon_carry_signed(uint32_t& a0, uint32_t& a1, uint32_t& a2) {
    const uint32_t t1 = a1 + (uint32_t)((int32_t)a0 >> 22);
    a2 += (uint32_t)((int32_t)t1 >> 22);
    a1 = t1 & 0x3fffffu;
    a0 &= 0x3fffffu;
}
BSX_HDI uint64_t poseidon_recombine_signed(uint32_t a0, uint32_t a1, uint32_t a2) {
    poseidon_carry_signed(a0, a1, a2);
    return poseidon_recombine(a0, a1, a2);
}

BSX_HDI void poseidon_mds(uint64_t s[12]) {
    uint32_t l0[12], l1[12], l2[12], a0[12], a1[12], a2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        l0[i] = (uint32_t)s[i] & 0x3fffffu;
        l1[i] = (uint32_t)(s[i] >> 22) & 0x3fffffu;
        l2[i] = (uint32_t)(s[i] >> 44);
    }
    poseidon_mds_limbs(l0, a0);
    poseidon_mds_limbs(l1, a1);
    poseidon_mds_limbs(l2, a2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int r = 0; r < 12; r += 3) poseidon_recombine3(s[r], s[r + 1], s[r + 2], a0[r], a1[r], a2[r], a0[r + 1], a1[r + 1], a2[r + 1], a0[r + 2], a1[r + 2], a2[r + 2]);
#else
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = poseidon_recombine(a0[r], a1[r], a2[r]);
#endif
}

// The 22 partial rounds with the state kept in LIMB form.  Only lane 0 goes through the S-box; lanes 1..11 see nothing but
// "+ constant" and the (linear) MDS from round to round.
//   * Constants (round 5): the constant of lane i > 0 in round r commutes with the S-box layer and can be added behind that round's
//     MDS as M (0, c_r[1..11]) — i.e. merged into round r + 1's constants, whose lanes 1..11 move on in turn.  What is left is ONE
//     constant per partial round (lane 0) and a full vector in front of the first full round behind them: rcf[0..22) and rcf[22..34)
//     (BSX_POSEIDON_FOLDED_TABLE, derived and checked against the 360-constant definition by tools/gen_poseidon_constants.py).
//   * Lanes 1..11 are never recombined to u64 between rounds: the MDS output limbs are carry-normalised in place,
//         t1 = a1 + (a0 >> 22), t2 = a2 + (t1 >> 22), h = t2 >> 20 (weight 2^64 = 2^32 - 1: l1 += h << 10, l0 -= h)
//     with l0 = (a0 & M22) - h left SIGNED (round 5: no borrow from l1; arithmetic shifts) — 10 operations per lane and round.
//     Invariants on entry to an MDS: -265 <= l0 < 2^22, 0 <= l1 < 2^22 + 2^18.1, 0 <= l2 < 2^20 (h <= 265 because a2 < 264 * 2^20),
//     so every MDS output fits an int32 (|a_k| < 264 * 2^22.1 < 2^31) and poseidon_mds_limbs' wrap-around arithmetic stays exact; a
//     lane's value l0 + 2^22 l1 + 2^44 l2 is never negative (h > 0 puts h << 10 >= 1024 into l1), hence t2 >= 0.
BSX_HDI void poseidon_partial_rounds(uint64_t s[12], const uint64_t* rcf) {
    constexpr uint32_t M22 = 0x3fffffu, M20 = 0xfffffu;
    uint32_t l0[12], l1[12], l2[12], a0[12], a1[12], a2[12];
#pragma unroll
    for (int i = 1; i < 12; i++) {
        l0[i] = (uint32_t)s[i] & M22;
        l1[i] = (uint32_t)(s[i] >> 22) & M22;
        l2[i] = (uint32_t)(s[i] >> 44);
    }
    uint64_t u0 = s[0];
    for (int k = 0; k < POSEIDON_PARTIAL; k++) {
        u0 = gl_pow7(gl_add_canon(u0, rcf[k]));
        l0[0] = (uint32_t)u0 & M22; l1[0] = (uint32_t)(u0 >> 22) & M22; l2[0] = (uint32_t)(u0 >> 44);
        poseidon_mds_limbs(l0, a0);
        poseidon_mds_limbs(l1, a1);
        poseidon_mds_limbs(l2, a2);
        u0 = poseidon_recombine_signed(a0[0], a1[0], a2[0]);
        if (k + 1 < POSEIDON_PARTIAL) {
#pragma unroll
            for (int i = 1; i < 12; i++) {
                const uint32_t t1 = a1[i] + (uint32_t)((int32_t)a0[i] >> 22);
                const uint32_t t2 = a2[i] + (uint32_t)((int32_t)t1 >> 22);
                const uint32_t h = t2 >> 20;
                l0[i] = (a0[i] & M22) - h;
                l1[i] = (t1 & M22) + (h << 10);
                l2[i] = t2 & M20;
            }
        }
    }
    s[0] = u0;
#if defined(__HIP_DEVICE_COMPILE__)
    {
#pragma unroll
        for (int i = 1; i < 12; i++) poseidon_carry_signed(a0[i], a1[i], a2[i]);
#pragma unroll
        for (int i = 1; i < 10; i += 3) poseidon_recombine3(s[i], s[i + 1], s[i + 2], a0[i], a1[i], a2[i], a0[i + 1], a1[i + 1], a2[i + 1], a0[i + 2], a1[i + 2], a2[i + 2]);
        s[10] = poseidon_recombine(a0[10], a1[10], a2[10]);
        s[11] = poseidon_recombine(a0[11], a1[11], a2[11]);
    }
#else
#pragma unroll
    for (int i = 1; i < 12; i++) s[i] = poseidon_recombine_signed(a0[i], a1[i], a2[i]);
#endif
}

// add the round's constants, x^7 on all 12 lanes.  On the device three words at a time through ONE hand-scheduled asm block
// (goldilocks_sbox_asm.h, generated: the constant, x^2, x^3, x^4, x^7 of three independent chains interleaved instruction by
// instruction, 64-bit temporaries in fixed registers), four blocks per round
BSX_HDI void poseidon_full_sbox(uint64_t s[12], const uint64_t* rc) {
#ifdef BSX_GL_SBOX3_ASM
#pragma unroll
    for (int i = 0; i < 12; i += 3) gl_sbox3(s[i], s[i + 1], s[i + 2], rc[i], rc[i + 1], rc[i + 2]);
#else
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_canon(s[i], rc[i]));
#endif
}

// rc: BSX_POSEIDON_TABLE (poseidon_consts.h) = the 360 round constants, round-major, then the 34 folded ones of the partial rounds
BSX_HDI void poseidon_permute(uint64_t s[12], const uint64_t* rc) {
    const uint64_t* rcf = rc + 12 * POSEIDON_ROUNDS;
    int r = 0;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, rc + 12 * r);
        poseidon_mds(s);
    }
    poseidon_partial_rounds(s, rcf);
    r += POSEIDON_PARTIAL;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, k == 0 ? rcf + POSEIDON_PARTIAL : rc + 12 * r);      // round 26 takes what the folding left of lanes 1..11
        poseidon_mds(s);
    }
}

// hash_n_to_hash_no_pad over n elements produced by get(k), k = 0..n-1 (values are reduced to canonical form on the
// way in, as GoldilocksField::from_noncanonical_u64 would); out = canonical digest.
template <typename Get>
BSX_HDI void poseidon_hash_no_pad(Get get, uint64_t n, const uint64_t* rc, uint64_t out[4]) {
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (uint64_t k = 0; k < n; k += POSEIDON_RATE) {
#pragma unroll
        for (int i = 0; i < POSEIDON_RATE; i++)
            if (k + i < n) s[i] = gl_canonical(get(k + i));
        poseidon_permute(s, rc);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = gl_canonical(s[i]);
}

// PoseidonHash::hash_or_noop (Merkle leaf digest): at most 4 elements are their own digest
template <typename Get>
BSX_HDI void poseidon_hash_or_noop(Get get, uint64_t n, const uint64_t* rc, uint64_t out[4]) {
    if (n <= 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = (uint64_t)i < n ? gl_canonical(get(i)) : 0;
    } else {
        poseidon_hash_no_pad(get, n, rc, out);
    }
}

// PoseidonHash::two_to_one (Merkle inner node)
BSX_HDI void poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], const uint64_t* rc, uint64_t out[4]) {
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = gl_canonical(l[i]); s[4 + i] = gl_canonical(r[i]); s[8 + i] = 0; }
    poseidon_permute(s, rc);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = gl_canonical(s[i]);
}

}  // namespace bsx
