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

// ---- hand-written gfx950 body (round 5): THREE independent multiplications per call, interleaved instruction by instruction.
// What it buys over the compiler's gl_mul (tools/glmul_asm_bench.hip, same results):
//   * the reduction's 64-bit adds / subs are v_add_co / v_addc pairs whose carry is USED — the compiler never keeps a carry-out: it
//     re-derives every one with v_cmp_*_u64 + v_cndmask (4 half-rate compares and 8 instructions per multiplication);
//   * hi_lo * (2^32 - 1) + t0 is ONE v_mad_u64_u32 with the 64-bit addend and its own carry-out (the compiler: a multiply-add
//     with addend 0, a 64-bit add and a compare);
//   * NO zero-extended addends (second pass): the schoolbook chain P = a0 b0, Q = a0 b1 + hi(P), R = a1 b0 + lo(Q), H = a1 b1 + hi(Q)
//     feeds every multiply-add a 32-bit half widened to a register PAIR — two v_mov per addend (7.6 moves per multiplication in
//     the compiled round).  Here the two middle products are summed as 64-bit values, Q = a1 b0 + (a0 b1) with the carry-out kept
//     (bit 64 of Q), and the 128-bit product is put together by ONE carry chain over the halves:
//         w1 = hi(P) + lo(Q);  w2 = lo(H) + hi(Q) + c;  w3 = hi(H) + c;  w3 += carry(Q)
//     4 multiply-adds + 4 adds, no move;
//   * every VALU write of an SGPR carry sits two instructions in front of its reader (the gfx940 VALU-writes-SGPR -> VALU-reads
//     hazard): no s_nop, where the compiler's schedule of ONE dependent chain pays 2-3 per multiplication.
// Sub-registers of a 64-bit operand cannot be named in inline asm, so the blocks are split where a half is read on its own (the
// compiler only renames registers there).  Device only; the host compiles the portable gl_mul (tests/hostcheck).
#if defined(__HIP_DEVICE_COMPILE__)
#define BSX_GL_MUL3_ASM 1
__device__ __forceinline__ void gl_mul3(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t y0, uint64_t y1, uint64_t y2) {
    uint64_t P0, R0, H0, K0, C0, P1, R1, H1, K1, C1, P2, R2, H2, K2, C2;
    const uint32_t a00 = (uint32_t)x0, a01 = (uint32_t)(x0 >> 32), b00 = (uint32_t)y0, b01 = (uint32_t)(y0 >> 32);
    const uint32_t a10 = (uint32_t)x1, a11 = (uint32_t)(x1 >> 32), b10 = (uint32_t)y1, b11 = (uint32_t)(y1 >> 32);
    const uint32_t a20 = (uint32_t)x2, a21 = (uint32_t)(x2 >> 32), b20 = (uint32_t)y2, b21 = (uint32_t)(y2 >> 32);
    // the four products of each multiplication in one block (a carry-out nobody reads goes to vcc): P = a0 b0, H = a1 b1,
    // R = a1 b0 + a0 b1 with its carry K
    asm volatile(
        "v_mad_u64_u32 %0, vcc, %12, %14, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %18, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %20, %22, 0\n\t"
        "v_mad_u64_u32 %3, vcc, %12, %15, 0\n\t"
        "v_mad_u64_u32 %4, vcc, %16, %19, 0\n\t"
        "v_mad_u64_u32 %5, vcc, %20, %23, 0\n\t"
        "v_mad_u64_u32 %6, vcc, %13, %15, 0\n\t"
        "v_mad_u64_u32 %7, vcc, %17, %19, 0\n\t"
        "v_mad_u64_u32 %8, vcc, %21, %23, 0\n\t"
        "v_mad_u64_u32 %3, %9, %13, %14, %3\n\t"
        "v_mad_u64_u32 %4, %10, %17, %18, %4\n\t"
        "v_mad_u64_u32 %5, %11, %21, %22, %5\n\t"
        : "=&v"(P0), "=&v"(P1), "=&v"(P2), "=&v"(R0), "=&v"(R1), "=&v"(R2), "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&s"(K0), "=&s"(K1), "=&s"(K2)
        : "v"(a00), "v"(a01), "v"(b00), "v"(b01), "v"(a10), "v"(a11), "v"(b10), "v"(b11), "v"(a20), "v"(a21), "v"(b20), "v"(b21)
        : "vcc");
    uint32_t l00 = (uint32_t)P0, l10 = (uint32_t)P1, l20 = (uint32_t)P2, l01, l11, l21, h00, h01, h10, h11, h20, h21;
    const uint32_t p01 = (uint32_t)(P0 >> 32), p11 = (uint32_t)(P1 >> 32), p21 = (uint32_t)(P2 >> 32);
    const uint32_t g00 = (uint32_t)H0, g01 = (uint32_t)(H0 >> 32), g10 = (uint32_t)H1, g11 = (uint32_t)(H1 >> 32), g20 = (uint32_t)H2, g21 = (uint32_t)(H2 >> 32);
    const uint32_t q00 = (uint32_t)R0, q01 = (uint32_t)(R0 >> 32), q10 = (uint32_t)R1, q11 = (uint32_t)(R1 >> 32), q20 = (uint32_t)R2, q21 = (uint32_t)(R2 >> 32);
    // words 1-3 of the product: one carry chain per multiplication (l_1 = w1, h_0 = w2 = hi_lo, h_1 = w3 = hi_hi); the results go to
    // fresh registers (an in-out operand on a half of a 64-bit asm result costs a copy)
    asm volatile(
        "v_add_co_u32 %0, %9, %12, %15\n\t"
        "v_add_co_u32 %3, %10, %18, %21\n\t"
        "v_add_co_u32 %6, %11, %24, %27\n\t"
        "v_addc_co_u32 %1, %9, %13, %16, %9\n\t"
        "v_addc_co_u32 %4, %10, %19, %22, %10\n\t"
        "v_addc_co_u32 %7, %11, %25, %28, %11\n\t"
        "v_addc_co_u32 %2, %9, 0, %14, %9\n\t"
        "v_addc_co_u32 %5, %10, 0, %20, %10\n\t"
        "v_addc_co_u32 %8, %11, 0, %26, %11\n\t"
        "v_addc_co_u32 %2, %9, 0, %2, %17\n\t"
        "v_addc_co_u32 %5, %10, 0, %5, %23\n\t"
        "v_addc_co_u32 %8, %11, 0, %8, %29\n\t"
        : "=&v"(l01), "=&v"(h00), "=&v"(h01), "=&v"(l11), "=&v"(h10), "=&v"(h11), "=&v"(l21), "=&v"(h20), "=&v"(h21), "=&s"(C0), "=&s"(C1), "=&s"(C2)
        : "v"(p01), "v"(g00), "v"(g01), "v"(q00), "v"(q01), "s"(K0),
          "v"(p11), "v"(g10), "v"(g11), "v"(q10), "v"(q11), "s"(K1),
          "v"(p21), "v"(g20), "v"(g21), "v"(q20), "v"(q21), "s"(K2));
    uint32_t m0, m1, m2;
    // t0 = lo - hi_hi (2^96 = -1); a borrow wrapped by 2^64 = p + EPS: give EPS back
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
    const uint64_t T0 = (uint64_t)l00 | ((uint64_t)l01 << 32), T1 = (uint64_t)l10 | ((uint64_t)l11 << 32), T2 = (uint64_t)l20 | ((uint64_t)l21 << 32);
    // U = hi_lo * EPS + t0, carry out of the multiply-add itself (hi_lo * EPS <= 2^64 - 2^33 + 1: one wrap at most)
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
// x^7 of three elements: x2 = x x, x3 = x2 x, x4 = x2 x2, x7 = x3 x4 (16 multiplications of a full round's 48 per call of four)
__device__ __forceinline__ void gl_pow7_3(uint64_t& a, uint64_t& b, uint64_t& c) {
    uint64_t a2 = a, b2 = b, c2 = c;
    gl_mul3(a2, b2, c2, a, b, c);
    uint64_t a3 = a2, b3 = b2, c3 = c2;
    gl_mul3(a3, b3, c3, a, b, c);
    uint64_t a4 = a2, b4 = b2, c4 = c2;
    gl_mul3(a4, b4, c4, a2, b2, c2);
    gl_mul3(a3, b3, c3, a4, b4, c4);
    a = a3; b = b3; c = c3;
}
#endif

}  // namespace bsx
