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

// a0 + 2^22 a1 + 2^44 a2 (a_k < 2^32) reduced to a u64 representative: the part of a2 above bit 20 has weight 2^64 = EPS
BSX_HDI uint64_t poseidon_recombine(uint32_t a0, uint32_t a1, uint32_t a2) {
    const uint64_t x = (uint64_t)a0 + ((uint64_t)a1 << 22);
    const uint64_t y = (uint64_t)(a2 & 0xfffffu) << 44;
    uint64_t lo, t;
    const bool c1 = __builtin_add_overflow(x, y, &lo);
    const uint64_t hi = (uint64_t)(a2 >> 20) + (c1 ? 1u : 0u);                     // < 2^13
    const bool c2 = __builtin_add_overflow(lo, (hi << 32) - hi, &t);
    return t + gl_eps_if(c2);
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
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = poseidon_recombine(a0[r], a1[r], a2[r]);
}

// The 22 partial rounds with the state kept in LIMB form.  Only lane 0 goes through the S-box; lanes 1..11 see nothing but
// "+ constant" and the (linear) MDS from round to round, so instead of recombining them to u64, adding the constant mod p
// and splitting again (~32 VALU issue slots per lane and round), the MDS output limbs take the next round's constant limb
// by limb and are carry-normalised in place (~14): t_k = a_k + c_k + carry; the overflow h of the top limb (weight 2^64 =
// 2^32 - 1) is folded back as l1 += h << 10, l0 -= h (borrowing 2^22 from l1 when l0 < h).  Invariants: l0 < 2^23,
// l1 < 2^23.4, l2 < 2^21 on entry to an MDS, so every true MDS output is < 284 * 2^23.4 < 2^32 and the wrap-around 32-bit
// arithmetic of poseidon_mds_limbs stays exact.
BSX_HDI void poseidon_partial_rounds(uint64_t s[12], const uint64_t* rc) {
    constexpr uint32_t M22 = 0x3fffffu, M20 = 0xfffffu;
    uint32_t l0[12], l1[12], l2[12], a0[12], a1[12], a2[12];
    const int r0 = POSEIDON_FULL_HALF;
#pragma unroll
    for (int i = 1; i < 12; i++) {                       // lanes 1..11: split + the first partial round's constants
        const uint64_t c = rc[12 * r0 + i];
        l0[i] = ((uint32_t)s[i] & M22) + ((uint32_t)c & M22);
        l1[i] = ((uint32_t)(s[i] >> 22) & M22) + ((uint32_t)(c >> 22) & M22);
        l2[i] = (uint32_t)(s[i] >> 44) + (uint32_t)(c >> 44);
    }
    uint64_t u0 = s[0];
    for (int k = 0; k < POSEIDON_PARTIAL; k++) {
        const int r = r0 + k;
        u0 = gl_pow7(gl_add_canon(u0, rc[12 * r]));
        l0[0] = (uint32_t)u0 & M22; l1[0] = (uint32_t)(u0 >> 22) & M22; l2[0] = (uint32_t)(u0 >> 44);
        poseidon_mds_limbs(l0, a0);
        poseidon_mds_limbs(l1, a1);
        poseidon_mds_limbs(l2, a2);
        u0 = poseidon_recombine(a0[0], a1[0], a2[0]);
        if (k + 1 < POSEIDON_PARTIAL) {
#pragma unroll
            for (int i = 1; i < 12; i++) {
                const uint64_t c = rc[12 * (r + 1) + i];                               // uniform: limb split on the scalar unit
                const uint32_t t0 = a0[i] + ((uint32_t)c & M22);
                const uint32_t t1 = a1[i] + ((uint32_t)(c >> 22) & M22) + (t0 >> 22);
                const uint32_t t2 = a2[i] + (uint32_t)(c >> 44) + (t1 >> 22);
                const uint32_t h = t2 >> 20, q0 = t0 & M22;
                const uint32_t b = q0 < h ? 1u : 0u;
                l0[i] = q0 - h + (b << 22);
                l1[i] = (t1 & M22) + (h << 10) - b;
                l2[i] = t2 & M20;
            }
        }
    }
    s[0] = u0;
#pragma unroll
    for (int i = 1; i < 12; i++) s[i] = poseidon_recombine(a0[i], a1[i], a2[i]);
}

// add the round's constants, x^7 on all 12 lanes.  On the device the twelve S-boxes go through the hand-written triple multiplication
// (goldilocks.h gl_mul3: three independent chains interleaved), four triples per round
BSX_HDI void poseidon_full_sbox(uint64_t s[12], const uint64_t* rc) {
#ifdef BSX_GL_MUL3_ASM
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canon(s[i], rc[i]);
#pragma unroll
    for (int i = 0; i < 12; i += 3) gl_pow7_3(s[i], s[i + 1], s[i + 2]);
#else
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_canon(s[i], rc[i]));
#endif
}

// rc: the 360 round constants, round-major
BSX_HDI void poseidon_permute(uint64_t s[12], const uint64_t* rc) {
    int r = 0;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, rc + 12 * r);
        poseidon_mds(s);
    }
    poseidon_partial_rounds(s, rc);
    r += POSEIDON_PARTIAL;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
        poseidon_full_sbox(s, rc + 12 * r);
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
