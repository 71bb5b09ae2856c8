// goldilocks.h — arithmetic in the Goldilocks field F_p, p = 2^64 - 2^32 + 1, for one lane.
//
// The field of every plonky2x variable on the header_range path (`L::Field = GoldilocksField` through
// DefaultParameters, bin/header_range_2048.rs:1-17; plonky2 pinned at Cargo.lock:3110-3112 [UPSTREAM]).  Elements are
// u64; like plonky2's GoldilocksField the intermediate representation is NOT canonical (any u64 stands for x mod p),
// gl_canonical() produces the representative in [0, p).  Identities used: 2^64 = 2^32 - 1 = EPS and 2^96 = -1 (mod p).
// 64 x 64 -> 128 products are four 32 x 32 + 64 multiply-adds (v_mad_u64_u32 on gfx950).
#pragma once
#include "bsx_common.h"

namespace bsx {

constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull;
constexpr uint64_t GL_EPS = 0xFFFFFFFFull;   // 2^64 mod p

BSX_HDI uint64_t gl_canonical(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

// a + b for arbitrary representatives
// EPS if `carry` else 0, without a 64-bit compare/select: the carry-out of the preceding add/sub becomes a 32-bit mask
BSX_HDI uint64_t gl_eps_if(bool carry) { return (uint64_t)(0u - (uint32_t)carry); }

// a + b for arbitrary representatives
BSX_HDI uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s, t;
    const bool c1 = __builtin_add_overflow(a, b, &s);                  // wrapped: 2^64 = EPS
    const bool c2 = __builtin_add_overflow(s, gl_eps_if(c1), &t);      // second wrap (only when both inputs were >= p)
    return t + gl_eps_if(c2);
}
// a + c with c canonical (round constants): one correction suffices
BSX_HDI uint64_t gl_add_canon(uint64_t a, uint64_t c) {
    uint64_t s;
    const bool cy = __builtin_add_overflow(a, c, &s);
    return s + gl_eps_if(cy);
}
BSX_HDI uint64_t gl_sub(uint64_t a, uint64_t b) {
    b = gl_canonical(b);
    uint64_t d;
    const bool bw = __builtin_sub_overflow(a, b, &d);
    return d - gl_eps_if(bw);    // borrowed 2^64 = p + EPS: give EPS back
}

// full 128-bit product from 32-bit halves
BSX_HDI void gl_mul128(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)al * bl;
    const uint64_t m1 = (uint64_t)al * bh + (p00 >> 32);            // < 2^64
    const uint64_t m2 = (uint64_t)ah * bl + (uint32_t)m1;           // < 2^64
    hi = (uint64_t)ah * bh + (m1 >> 32) + (m2 >> 32);
    lo = (m2 << 32) | (uint32_t)p00;
}

// (lo + 2^64 hi) mod p, result any u64 representative
BSX_HDI uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
    const uint64_t hi_hi = hi >> 32, hi_lo = (uint32_t)hi;
    uint64_t t0, t2;
    const bool bw = __builtin_sub_overflow(lo, hi_hi, &t0);      // 2^96 = -1
    t0 -= gl_eps_if(bw);
    const uint64_t t1 = (hi_lo << 32) - hi_lo;                   // hi_lo * EPS
    const bool cy = __builtin_add_overflow(t0, t1, &t2);
    return t2 + gl_eps_if(cy);
}

BSX_HDI uint64_t gl_mul(uint64_t a, uint64_t b) {
    uint64_t lo, hi;
    gl_mul128(a, b, lo, hi);
    return gl_reduce128(lo, hi);
}
BSX_HDI uint64_t gl_sq(uint64_t a) { return gl_mul(a, a); }

// x^7: the Poseidon S-box (4 multiplications)
BSX_HDI uint64_t gl_pow7(uint64_t x) {
    const uint64_t x2 = gl_sq(x), x3 = gl_mul(x2, x), x4 = gl_sq(x2);
    return gl_mul(x3, x4);
}

}  // namespace bsx
