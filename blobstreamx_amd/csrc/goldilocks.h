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
BSX_HDI uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) {                 // wrapped: 2^64 = EPS
        s += GL_EPS;
        if (s < GL_EPS) s += GL_EPS;   // second wrap (only when both inputs were >= p)
    }
    return s;
}
// a + c with c canonical (round constants): one correction suffices
BSX_HDI uint64_t gl_add_canon(uint64_t a, uint64_t c) {
    const uint64_t s = a + c;
    return s < a ? s + GL_EPS : s;
}
BSX_HDI uint64_t gl_sub(uint64_t a, uint64_t b) {
    b = gl_canonical(b);
    uint64_t d = a - b;
    if (a < b) d -= GL_EPS;      // borrowed 2^64 = p + EPS: give EPS back
    return d;
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
    uint64_t t0 = lo - hi_hi;                        // 2^96 = -1
    if (lo < hi_hi) t0 -= GL_EPS;
    const uint64_t t1 = (hi_lo << 32) - hi_lo;       // hi_lo * EPS
    uint64_t t2 = t0 + t1;
    if (t2 < t1) t2 += GL_EPS;
    return t2;
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
