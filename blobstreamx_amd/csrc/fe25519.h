// fe25519.h — GF(2^255-19) and the Ed25519 group for one lane, radix 2^25.5 (10 signed 32-bit limbs,
// 64-bit accumulation = v_mad_i64_i32 chains on gfx950).  Device side of P7: the
// per-validator check [s]B == R + [h]A the reference enforces inside builder.skip / builder.step
// (circuits/header_range.rs:42-48, circuits/next_header.rs:32-36; [UPSTREAM] curta EdDSA gadget).
// Representation and formulas differ from the test oracle on purpose (oracle: 5 x 51-bit, Shamir
// double-and-add, decompresses R; here: 10 x 25.5-bit, fixed 4-bit signed windows, compares the encoding).
#pragma once
#include "bsx_common.h"

namespace bsx {

struct fe {
    int32_t v[10];
};

BSX_HDI fe fe_zero() { fe r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
BSX_HDI fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }
BSX_HDI fe fe_add(const fe& a, const fe& b) { fe r; for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
BSX_HDI fe fe_sub(const fe& a, const fe& b) { fe r; for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
BSX_HDI fe fe_neg(const fe& a) { fe r; for (int i = 0; i < 10; i++) r.v[i] = -a.v[i]; return r; }
BSX_HDI fe fe_select(bool c, const fe& a, const fe& b) {  // c ? a : b
    fe r;
    for (int i = 0; i < 10; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// carry chain shared by mul / sq (interleaved two-lane order keeps every limb inside its bound)
BSX_HDI fe fe_carry64(int64_t h[10]) {
    int64_t c;
    c = (h[0] + (1 << 25)) >> 26; h[1] += c; h[0] -= c << 26;
    c = (h[4] + (1 << 25)) >> 26; h[5] += c; h[4] -= c << 26;
    c = (h[1] + (1 << 24)) >> 25; h[2] += c; h[1] -= c << 25;
    c = (h[5] + (1 << 24)) >> 25; h[6] += c; h[5] -= c << 25;
    c = (h[2] + (1 << 25)) >> 26; h[3] += c; h[2] -= c << 26;
    c = (h[6] + (1 << 25)) >> 26; h[7] += c; h[6] -= c << 26;
    c = (h[3] + (1 << 24)) >> 25; h[4] += c; h[3] -= c << 25;
    c = (h[7] + (1 << 24)) >> 25; h[8] += c; h[7] -= c << 25;
    c = (h[4] + (1 << 25)) >> 26; h[5] += c; h[4] -= c << 26;
    c = (h[8] + (1 << 25)) >> 26; h[9] += c; h[8] -= c << 26;
    c = (h[9] + (1 << 24)) >> 25; h[0] += c * 19; h[9] -= c << 25;
    c = (h[0] + (1 << 25)) >> 26; h[1] += c; h[0] -= c << 26;
    fe r;
    for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
    return r;
}

// h = f * g.  Preconditions as in the classic 25.5-bit schoolbook: |f|,|g| limbs <= 1.65*2^26 (even) / 2^25 (odd).
// NOT inlined on the device: ~70 call sites share one ~2 KB body.  The operands travel as native vectors (8 + 2 limbs
// each): the AMDGPU calling convention gives aggregates at most 16 argument registers, so a second `fe` struct would
// be passed through scratch memory — a store/load round trip per multiplication that stalls behind the HBM-bound
// witness expansion sharing the CU.  Vector arguments are not aggregates; all 20 limbs arrive in VGPRs.
typedef int32_t i32x8 __attribute__((vector_size(32)));
typedef int32_t i32x2 __attribute__((vector_size(8)));
BSX_HDI fe fe_from_v(i32x8 a, i32x2 b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a[i];
    r.v[8] = b[0]; r.v[9] = b[1];
    return r;
}
BSX_HDI i32x8 fe_lo8(const fe& f) { return i32x8{f.v[0], f.v[1], f.v[2], f.v[3], f.v[4], f.v[5], f.v[6], f.v[7]}; }
BSX_HDI i32x2 fe_hi2(const fe& f) { return i32x2{f.v[8], f.v[9]}; }

// the schoolbook body; fe_mul_inl inlines it at the call site (used only by the one-lane-per-key doubling chain of the
// fixed-key table build, which is pure dependent-issue latency: inlined, the three / four independent field operations of
// a point doubling interleave instead of running back to back behind call boundaries)
BSX_HDI fe fe_mul_inl(const fe& f, const fe& g) {
    int32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19 * g.v[i]; f2[i] = 2 * f.v[i]; }
    int64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int k = i + j;
            const int32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const int32_t gj = (k >= 10) ? g19[j] : g.v[j];
            h[k >= 10 ? k - 10 : k] += (int64_t)fi * gj;
        }
    }
    return fe_carry64(h);
}
BSX_HD_NOINLINE fe fe_mul_v(i32x8 fa, i32x2 fb, i32x8 ga, i32x2 gb) { return fe_mul_inl(fe_from_v(fa, fb), fe_from_v(ga, gb)); }
BSX_HDI fe fe_mul(const fe& f, const fe& g) { return fe_mul_v(fe_lo8(f), fe_hi2(f), fe_lo8(g), fe_hi2(g)); }

// h = f^2 (DBL == false) or 2 f^2 (DBL == true)
template <bool DBL>
BSX_HDI fe fe_sq_impl(const fe& f) {
    int32_t f19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { f19[i] = 19 * f.v[i]; f2[i] = 2 * f.v[i]; }
    int64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = i; j < 10; j++) {
            const int k = i + j;
            // term f_i f_j: x2 when i != j (symmetric pair), x2 when both odd (radix 25.5), x19 when wrapping
            const int coef = ((i == j) ? 1 : 2) * (((i & 1) && (j & 1)) ? 2 : 1);
            const int32_t a = (coef == 1) ? f.v[i] : (coef == 2 ? f2[i] : 2 * f2[i]);
            const int32_t b = (k >= 10) ? f19[j] : f.v[j];
            h[k >= 10 ? k - 10 : k] += (int64_t)a * b;
        }
    }
    if (DBL) {
#pragma unroll
        for (int k = 0; k < 10; k++) h[k] *= 2;
    }
    return fe_carry64(h);
}
BSX_HD_NOINLINE fe fe_sq(fe f) { return fe_sq_impl<false>(f); }
BSX_HD_NOINLINE fe fe_sq2(fe f) { return fe_sq_impl<true>(f); }

BSX_HDI fe fe_sqn(fe x, int n) {
    for (int i = 0; i < n; i++) x = fe_sq(x);
    return x;
}

// z^(2^252 - 3) and z^(p-2): one shared ladder
BSX_HDI void fe_pow_ladder(const fe& z, fe& z_250_0, fe& z11, fe& z2) {
    z2 = fe_sq(z);                               // 2
    fe z8 = fe_sqn(z2, 2);                       // 8
    fe z9 = fe_mul(z, z8);                       // 9
    z11 = fe_mul(z2, z9);                        // 11
    fe z22 = fe_sq(z11);                         // 22
    fe z_5_0 = fe_mul(z9, z22);                  // 2^5 - 1
    fe z_10_0 = fe_mul(fe_sqn(z_5_0, 5), z_5_0);
    fe z_20_0 = fe_mul(fe_sqn(z_10_0, 10), z_10_0);
    fe z_40_0 = fe_mul(fe_sqn(z_20_0, 20), z_20_0);
    fe z_50_0 = fe_mul(fe_sqn(z_40_0, 10), z_10_0);
    fe z_100_0 = fe_mul(fe_sqn(z_50_0, 50), z_50_0);
    fe z_200_0 = fe_mul(fe_sqn(z_100_0, 100), z_100_0);
    z_250_0 = fe_mul(fe_sqn(z_200_0, 50), z_50_0);  // 2^250 - 1
}
BSX_HDI fe fe_invert(const fe& z) {
    fe a, z11, z2;
    fe_pow_ladder(z, a, z11, z2);
    return fe_mul(fe_sqn(a, 5), z11);            // 2^255 - 21
}
BSX_HDI fe fe_pow22523(const fe& z) {
    fe a, z11, z2;
    fe_pow_ladder(z, a, z11, z2);
    return fe_mul(fe_sqn(a, 2), z);              // 2^252 - 3
}

// canonical little-endian bytes as 8 dwords
BSX_HDI void fe_tobytes(uint32_t s[8], const fe& f) {
    int32_t h[10];
    for (int i = 0; i < 10; i++) h[i] = f.v[i];
    int32_t q = (19 * h[9] + (1 << 24)) >> 25;
    q = (h[0] + q) >> 26; q = (h[1] + q) >> 25; q = (h[2] + q) >> 26; q = (h[3] + q) >> 25; q = (h[4] + q) >> 26;
    q = (h[5] + q) >> 25; q = (h[6] + q) >> 26; q = (h[7] + q) >> 25; q = (h[8] + q) >> 26; q = (h[9] + q) >> 25;
    h[0] += 19 * q;
    int32_t c;
    c = h[0] >> 26; h[1] += c; h[0] -= c << 26;
    c = h[1] >> 25; h[2] += c; h[1] -= c << 25;
    c = h[2] >> 26; h[3] += c; h[2] -= c << 26;
    c = h[3] >> 25; h[4] += c; h[3] -= c << 25;
    c = h[4] >> 26; h[5] += c; h[4] -= c << 26;
    c = h[5] >> 25; h[6] += c; h[5] -= c << 25;
    c = h[6] >> 26; h[7] += c; h[6] -= c << 26;
    c = h[7] >> 25; h[8] += c; h[7] -= c << 25;
    c = h[8] >> 26; h[9] += c; h[8] -= c << 26;
    c = h[9] >> 25; h[9] -= c << 25;
    // limbs now in [0, 2^26) / [0, 2^25): pack at bit offsets 0,26,51,77,102,128,153,179,204,230
    const uint32_t u0 = (uint32_t)h[0], u1 = (uint32_t)h[1], u2 = (uint32_t)h[2], u3 = (uint32_t)h[3], u4 = (uint32_t)h[4];
    const uint32_t u5 = (uint32_t)h[5], u6 = (uint32_t)h[6], u7 = (uint32_t)h[7], u8 = (uint32_t)h[8], u9 = (uint32_t)h[9];
    s[0] = u0 | (u1 << 26);
    s[1] = (u1 >> 6) | (u2 << 19);
    s[2] = (u2 >> 13) | (u3 << 13);
    s[3] = (u3 >> 19) | (u4 << 6);
    s[4] = u5 | (u6 << 25);
    s[5] = (u6 >> 7) | (u7 << 19);
    s[6] = (u7 >> 13) | (u8 << 12);
    s[7] = (u8 >> 20) | (u9 << 6);
}

// from 8 little-endian dwords, bit 255 ignored
BSX_HDI fe fe_frombytes(const uint32_t s[8]) {
    int64_t h[10];
    h[0] = s[0] & 0x3ffffff;                                   // bits 0..25
    h[1] = ((s[0] >> 26) | (s[1] << 6)) & 0x1ffffff;           // 26..50
    h[2] = ((s[1] >> 19) | (s[2] << 13)) & 0x3ffffff;          // 51..76
    h[3] = ((s[2] >> 13) | (s[3] << 19)) & 0x1ffffff;          // 77..101
    h[4] = (s[3] >> 6) & 0x3ffffff;                            // 102..127
    h[5] = s[4] & 0x1ffffff;                                   // 128..152
    h[6] = ((s[4] >> 25) | (s[5] << 7)) & 0x3ffffff;           // 153..178
    h[7] = ((s[5] >> 19) | (s[6] << 13)) & 0x1ffffff;          // 179..203
    h[8] = ((s[6] >> 12) | (s[7] << 20)) & 0x3ffffff;          // 204..229
    h[9] = (s[7] >> 6) & 0x1ffffff;                            // 230..254
    fe r;
    for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
    return r;
}

BSX_HDI bool fe_isnonzero(const fe& f) {
    uint32_t s[8];
    fe_tobytes(s, f);
    return (s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7]) != 0;
}
BSX_HDI int fe_isnegative(const fe& f) {
    uint32_t s[8];
    fe_tobytes(s, f);
    return (int)(s[0] & 1);
}

// ---------------------------------------------------------------- group
struct ge_p2 { fe X, Y, Z; };
struct ge_p3 { fe X, Y, Z, T; };
struct ge_p1p1 { fe X, Y, Z, T; };
struct ge_precomp { fe yplusx, yminusx, xy2d; };        // affine, Z = 1
struct ge_cached { fe YplusX, YminusX, Z, T2d; };

#include "ed25519_consts.h"   // FE_D, FE_D2, FE_SQRTM1, GE_B_TABLE[8] (generated: tools/gen_ed25519_tables.py)

BSX_HDI ge_p2 p1p1_to_p2(const ge_p1p1& p) { return ge_p2{fe_mul(p.X, p.T), fe_mul(p.Y, p.Z), fe_mul(p.Z, p.T)}; }
BSX_HDI ge_p3 p1p1_to_p3(const ge_p1p1& p) {
    return ge_p3{fe_mul(p.X, p.T), fe_mul(p.Y, p.Z), fe_mul(p.Z, p.T), fe_mul(p.X, p.Y)};
}
BSX_HDI ge_p1p1 ge_dbl(const fe& X, const fe& Y, const fe& Z) {
    ge_p1p1 r;
    fe xx = fe_sq(X), yy = fe_sq(Y), zz2 = fe_sq2(Z);
    fe t0 = fe_sq(fe_add(X, Y));
    r.Y = fe_add(yy, xx);
    r.Z = fe_sub(yy, xx);
    r.X = fe_sub(t0, r.Y);
    r.T = fe_sub(zz2, r.Z);
    return r;
}
// inlined forms for latency-bound single-chain code (see fe_mul_inl)
BSX_HDI ge_p2 p1p1_to_p2_inl(const ge_p1p1& p) { return ge_p2{fe_mul_inl(p.X, p.T), fe_mul_inl(p.Y, p.Z), fe_mul_inl(p.Z, p.T)}; }
BSX_HDI ge_p3 p1p1_to_p3_inl(const ge_p1p1& p) {
    return ge_p3{fe_mul_inl(p.X, p.T), fe_mul_inl(p.Y, p.Z), fe_mul_inl(p.Z, p.T), fe_mul_inl(p.X, p.Y)};
}
BSX_HDI ge_p1p1 ge_dbl_inl(const fe& X, const fe& Y, const fe& Z) {
    ge_p1p1 r;
    fe xx = fe_sq_impl<false>(X), yy = fe_sq_impl<false>(Y), zz2 = fe_sq_impl<true>(Z);
    fe t0 = fe_sq_impl<false>(fe_add(X, Y));
    r.Y = fe_add(yy, xx);
    r.Z = fe_sub(yy, xx);
    r.X = fe_sub(t0, r.Y);
    r.T = fe_sub(zz2, r.Z);
    return r;
}
BSX_HDI ge_cached p3_to_cached(const ge_p3& p) {
    return ge_cached{fe_add(p.Y, p.X), fe_sub(p.Y, p.X), p.Z, fe_mul(p.T, fe_d2())};
}
// r = p + q
BSX_HDI ge_p1p1 ge_add(const ge_p3& p, const ge_cached& q) {
    ge_p1p1 r;
    fe a = fe_mul(fe_add(p.Y, p.X), q.YplusX);
    fe b = fe_mul(fe_sub(p.Y, p.X), q.YminusX);
    fe c = fe_mul(q.T2d, p.T);
    fe zz = fe_mul(p.Z, q.Z);
    fe d = fe_add(zz, zz);
    r.X = fe_sub(a, b);
    r.Y = fe_add(a, b);
    r.Z = fe_add(d, c);
    r.T = fe_sub(d, c);
    return r;
}
// r = p + q, q affine precomputed
BSX_HDI ge_p1p1 ge_madd(const ge_p3& p, const ge_precomp& q) {
    ge_p1p1 r;
    fe a = fe_mul(fe_add(p.Y, p.X), q.yplusx);
    fe b = fe_mul(fe_sub(p.Y, p.X), q.yminusx);
    fe c = fe_mul(q.xy2d, p.T);
    fe d = fe_add(p.Z, p.Z);
    r.X = fe_sub(a, b);
    r.Y = fe_add(a, b);
    r.Z = fe_add(d, c);
    r.T = fe_sub(d, c);
    return r;
}
// conditional negation of table entries (neg == true: -q)
BSX_HDI ge_cached cached_cneg(const ge_cached& q, bool neg) {
    return ge_cached{fe_select(neg, q.YminusX, q.YplusX), fe_select(neg, q.YplusX, q.YminusX), q.Z,
                     fe_select(neg, fe_neg(q.T2d), q.T2d)};
}
BSX_HDI ge_precomp precomp_cneg(const ge_precomp& q, bool neg) {
    return ge_precomp{fe_select(neg, q.yminusx, q.yplusx), fe_select(neg, q.yplusx, q.yminusx),
                      fe_select(neg, fe_neg(q.xy2d), q.xy2d)};
}
BSX_HDI ge_cached cached_identity() { return ge_cached{fe_one(), fe_one(), fe_one(), fe_zero()}; }
BSX_HDI ge_precomp precomp_identity() { return ge_precomp{fe_one(), fe_one(), fe_zero()}; }

// Decode the public key and NEGATE it (we evaluate [s]B + [h](-A)).  RFC 8032 §5.1.3 strictness:
// reject y >= p, off-curve, and x == 0 with the sign bit set.  s: 8 LE dwords.
BSX_HDI bool ge_frombytes_negate(ge_p3& h, const uint32_t s[8]) {
    // canonical: (s & (2^255-1)) < p = 2^255 - 19
    const uint32_t top = s[7] & 0x7fffffffu;
    const bool all_ones = (s[1] & s[2] & s[3] & s[4] & s[5] & s[6]) == 0xffffffffu && top == 0x7fffffffu;
    const bool canonical = !(all_ones && s[0] >= 0xffffffedu);
    h.Y = fe_frombytes(s);
    h.Z = fe_one();
    fe u = fe_sq(h.Y);
    fe v = fe_mul(u, fe_d());
    u = fe_sub(u, h.Z);  // y^2 - 1
    v = fe_add(v, h.Z);  // d y^2 + 1
    fe v3 = fe_mul(fe_sq(v), v);
    fe x = fe_mul(fe_mul(fe_sq(v3), v), u);  // u v^7
    x = fe_pow22523(x);
    x = fe_mul(fe_mul(x, v3), u);            // u v^3 (u v^7)^((p-5)/8)
    fe vxx = fe_mul(fe_sq(x), v);
    bool ok = true;
    if (fe_isnonzero(fe_sub(vxx, u))) {
        if (fe_isnonzero(fe_add(vxx, u))) ok = false;
        x = fe_mul(x, fe_sqrtm1());
    }
    const int sign = (int)(s[7] >> 31);
    const bool xzero = !fe_isnonzero(x);
    if (xzero && sign) ok = false;
    // negate: the decoded x has parity `sign`; we want -x, i.e. parity != sign (x != 0)
    if (fe_isnegative(x) == sign) x = fe_neg(x);
    h.X = x;
    h.T = fe_mul(h.X, h.Y);
    return ok && canonical;
}

// encoding of p (projective): 8 LE dwords
BSX_HDI void ge_tobytes(uint32_t s[8], const fe& X, const fe& Y, const fe& Z) {
    fe zi = fe_invert(Z);
    fe x = fe_mul(X, zi), y = fe_mul(Y, zi);
    fe_tobytes(s, y);
    s[7] ^= (uint32_t)fe_isnegative(x) << 31;
}

}  // namespace bsx
