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
// row (coefficients add to 284 < 2^9) stays below 2^31, so the 3 x 144 products are full-rate 24-bit multiply-adds
// (v_mad_u32_u24) instead of quarter-rate 32 x 32 -> 64 ones, and a row is recombined and reduced once.
#pragma once
#include "goldilocks.h"

namespace bsx {

constexpr int POSEIDON_WIDTH = 12, POSEIDON_RATE = 8, POSEIDON_FULL_HALF = 4, POSEIDON_PARTIAL = 22, POSEIDON_ROUNDS = 30;

BSX_HDI void poseidon_mds(uint64_t s[12]) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t l0[12], l1[12], l2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        l0[i] = (uint32_t)s[i] & 0x3fffffu;
        l1[i] = (uint32_t)(s[i] >> 22) & 0x3fffffu;
        l2[i] = (uint32_t)(s[i] >> 44);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int j = (i + r) % 12;
            a0 += l0[j] * C[i]; a1 += l1[j] * C[i]; a2 += l2[j] * C[i];
        }
        if (r == 0) { a0 += l0[0] * 8u; a1 += l1[0] * 8u; a2 += l2[0] * 8u; }      // diagonal
        // a0 + 2^22 a1 + 2^44 a2, a_k < 2^31: the part of a2 above bit 20 has weight 2^64 = EPS
        const uint64_t x = (uint64_t)a0 + ((uint64_t)a1 << 22);
        const uint64_t y = (uint64_t)(a2 & 0xfffffu) << 44;
        const uint64_t lo = x + y;
        const uint64_t hi = (uint64_t)(a2 >> 20) + (lo < y ? 1u : 0u);                 // < 2^12
        const uint64_t t = lo + ((hi << 32) - hi);
        s[r] = t < lo ? t + GL_EPS : t;
    }
}

// rc: the 360 round constants, round-major
BSX_HDI void poseidon_permute(uint64_t s[12], const uint64_t* rc) {
    int r = 0;
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_canon(s[i], rc[12 * r + i]));
        poseidon_mds(s);
    }
    for (int k = 0; k < POSEIDON_PARTIAL; k++, r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add_canon(s[i], rc[12 * r + i]);
        s[0] = gl_pow7(s[0]);
        poseidon_mds(s);
    }
    for (int k = 0; k < POSEIDON_FULL_HALF; k++, r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_canon(s[i], rc[12 * r + i]));
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
