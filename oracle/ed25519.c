/* oracle/ed25519.c — CPU ORACLE (test infrastructure only; see orc.h).
 * RFC 8032 Ed25519 verification, cofactorless: accept iff A and R decode canonically (y < p, on
 * curve, RFC 8032 §5.1.3), s < L, and [s]B == R + [h]A with h = SHA512(R ‖ A ‖ M) mod L.
 * This is the equation the reference's circuit enforces inside builder.skip/step
 * (circuits/header_range.rs:42-48; [UPSTREAM] curta EdDSA gadget) and what every fixture
 * signature satisfies (tests/golden/mocha4.json).  Field: 5 x 51-bit limbs, unsigned __int128
 * products.  Deliberately a different representation from the HIP kernel (10 x 25.5-bit). */
#include <string.h>

#include "orc.h"

typedef unsigned __int128 u128;
typedef uint64_t fe[5];
#define M51 ((1ULL << 51) - 1)

static const fe FE_D = {0x34dca135978a3, 0x1a8283b156ebd, 0x5e7a26001c029, 0x739c663a03cbb, 0x52036cee2b6ff};
static const fe FE_2D = {0x69b9426b2f159, 0x35050762add7a, 0x3cf44c0038052, 0x6738cc7407977, 0x2406d9dc56dff};
static const fe FE_SQRTM1 = {0x61b274a0ea0b0, 0xd5a5fc8f189d, 0x7ef5e9cbd0c60, 0x78595a6804c9e, 0x2b8324804fc1d};
static const fe FE_BX = {0x62d608f25d51a, 0x412a4b4f6592a, 0x75b7171a4b31d, 0x1ff60527118fe, 0x216936d3cd6e5};
static const fe FE_BY = {0x6666666666658, 0x4cccccccccccc, 0x1999999999999, 0x3333333333333, 0x6666666666666};
static const fe FE_BT = {0x68ab3a5b7dda3, 0xeea2a5eadbb, 0x2af8df483c27e, 0x332b375274732, 0x67875f0fd78b7};
static const uint64_t SC_L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};

static void fe_copy(fe o, const fe a) { memcpy(o, a, sizeof(fe)); }
static void fe_0(fe o) { memset(o, 0, sizeof(fe)); }
static void fe_1(fe o) { fe_0(o); o[0] = 1; }
static void fe_carry(fe o) {
    uint64_t c;
    for (int r = 0; r < 2; r++) {
        c = o[0] >> 51; o[0] &= M51; o[1] += c;
        c = o[1] >> 51; o[1] &= M51; o[2] += c;
        c = o[2] >> 51; o[2] &= M51; o[3] += c;
        c = o[3] >> 51; o[3] &= M51; o[4] += c;
        c = o[4] >> 51; o[4] &= M51; o[0] += 19 * c;
    }
}
static void fe_add(fe o, const fe a, const fe b) {
    for (int i = 0; i < 5; i++) o[i] = a[i] + b[i];
    fe_carry(o);
}
static void fe_sub(fe o, const fe a, const fe b) {
    /* a + 4p - b keeps limbs non-negative for b < 2^52 per limb */
    o[0] = a[0] + 0x1fffffffffffb4ULL - b[0];
    for (int i = 1; i < 5; i++) o[i] = a[i] + 0x1ffffffffffffcULL - b[i];
    fe_carry(o);
}
static void fe_mul(fe o, const fe a, const fe b) {
    u128 t[5];
    uint64_t b1 = 19 * b[1], b2 = 19 * b[2], b3 = 19 * b[3], b4 = 19 * b[4];
    t[0] = (u128)a[0] * b[0] + (u128)a[1] * b4 + (u128)a[2] * b3 + (u128)a[3] * b2 + (u128)a[4] * b1;
    t[1] = (u128)a[0] * b[1] + (u128)a[1] * b[0] + (u128)a[2] * b4 + (u128)a[3] * b3 + (u128)a[4] * b2;
    t[2] = (u128)a[0] * b[2] + (u128)a[1] * b[1] + (u128)a[2] * b[0] + (u128)a[3] * b4 + (u128)a[4] * b3;
    t[3] = (u128)a[0] * b[3] + (u128)a[1] * b[2] + (u128)a[2] * b[1] + (u128)a[3] * b[0] + (u128)a[4] * b4;
    t[4] = (u128)a[0] * b[4] + (u128)a[1] * b[3] + (u128)a[2] * b[2] + (u128)a[3] * b[1] + (u128)a[4] * b[0];
    uint64_t c;
    t[1] += (uint64_t)(t[0] >> 51); o[0] = (uint64_t)t[0] & M51;
    t[2] += (uint64_t)(t[1] >> 51); o[1] = (uint64_t)t[1] & M51;
    t[3] += (uint64_t)(t[2] >> 51); o[2] = (uint64_t)t[2] & M51;
    t[4] += (uint64_t)(t[3] >> 51); o[3] = (uint64_t)t[3] & M51;
    c = (uint64_t)(t[4] >> 51); o[4] = (uint64_t)t[4] & M51;
    o[0] += 19 * c;
    c = o[0] >> 51; o[0] &= M51; o[1] += c;
}
static void fe_sq(fe o, const fe a) { fe_mul(o, a, a); }
static void fe_neg(fe o, const fe a) { fe z; fe_0(z); fe_sub(o, z, a); }
static void fe_tobytes(uint8_t s[32], const fe a) {
    fe t;
    fe_copy(t, a);
    fe_carry(t);
    /* canonical reduce: add 19, carry, see whether it overflows 2^255 */
    uint64_t q = (t[0] + 19) >> 51;
    q = (t[1] + q) >> 51; q = (t[2] + q) >> 51; q = (t[3] + q) >> 51; q = (t[4] + q) >> 51;
    t[0] += 19 * q;
    uint64_t c;
    c = t[0] >> 51; t[0] &= M51; t[1] += c;
    c = t[1] >> 51; t[1] &= M51; t[2] += c;
    c = t[2] >> 51; t[2] &= M51; t[3] += c;
    c = t[3] >> 51; t[3] &= M51; t[4] += c;
    t[4] &= M51;
    uint64_t w0 = t[0] | t[1] << 51, w1 = t[1] >> 13 | t[2] << 38, w2 = t[2] >> 26 | t[3] << 25, w3 = t[3] >> 39 | t[4] << 12;
    uint64_t w[4] = {w0, w1, w2, w3};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
static void fe_frombytes(fe o, const uint8_t s[32]) {
    uint64_t w[4];
    for (int i = 0; i < 4; i++) {
        w[i] = 0;
        for (int j = 7; j >= 0; j--) w[i] = w[i] << 8 | s[8 * i + j];
    }
    o[0] = w[0] & M51;
    o[1] = (w[0] >> 51 | w[1] << 13) & M51;
    o[2] = (w[1] >> 38 | w[2] << 26) & M51;
    o[3] = (w[2] >> 25 | w[3] << 39) & M51;
    o[4] = (w[3] >> 12) & M51; /* drops bit 255 */
}
static int fe_iszero(const fe a) {
    uint8_t s[32];
    fe_tobytes(s, a);
    uint8_t r = 0;
    for (int i = 0; i < 32; i++) r |= s[i];
    return r == 0;
}
static int fe_isneg(const fe a) {
    uint8_t s[32];
    fe_tobytes(s, a);
    return s[0] & 1;
}
static int fe_eq(const fe a, const fe b) {
    fe t;
    fe_sub(t, a, b);
    return fe_iszero(t);
}
/* o = a^(2^252 - 3) = a^((p-5)/8) */
static void fe_pow22523(fe o, const fe z) {
    fe t0, t1, t2;
    int i;
    fe_sq(t0, z);
    fe_sq(t1, t0); fe_sq(t1, t1);
    fe_mul(t1, z, t1);
    fe_mul(t0, t0, t1);
    fe_sq(t0, t0);
    fe_mul(t0, t1, t0);
    fe_sq(t1, t0); for (i = 1; i < 5; i++) fe_sq(t1, t1);
    fe_mul(t0, t1, t0);
    fe_sq(t1, t0); for (i = 1; i < 10; i++) fe_sq(t1, t1);
    fe_mul(t1, t1, t0);
    fe_sq(t2, t1); for (i = 1; i < 20; i++) fe_sq(t2, t2);
    fe_mul(t1, t2, t1);
    fe_sq(t1, t1); for (i = 1; i < 10; i++) fe_sq(t1, t1);
    fe_mul(t0, t1, t0);
    fe_sq(t1, t0); for (i = 1; i < 50; i++) fe_sq(t1, t1);
    fe_mul(t1, t1, t0);
    fe_sq(t2, t1); for (i = 1; i < 100; i++) fe_sq(t2, t2);
    fe_mul(t1, t2, t1);
    fe_sq(t1, t1); for (i = 1; i < 50; i++) fe_sq(t1, t1);
    fe_mul(t0, t1, t0);
    fe_sq(t0, t0); fe_sq(t0, t0);
    fe_mul(o, t0, z);
}

typedef struct { fe X, Y, Z, T; } ge;

static void ge_identity(ge* p) { fe_0(p->X); fe_1(p->Y); fe_1(p->Z); fe_0(p->T); }
/* add-2008-hwcd-3 (a = -1), complete */
static void ge_add(ge* r, const ge* p, const ge* q) {
    fe a, b, c, d, e, f, g, h, t;
    fe_sub(a, p->Y, p->X); fe_sub(t, q->Y, q->X); fe_mul(a, a, t);
    fe_add(b, p->Y, p->X); fe_add(t, q->Y, q->X); fe_mul(b, b, t);
    fe_mul(c, p->T, q->T); fe_mul(c, c, FE_2D);
    fe_mul(d, p->Z, q->Z); fe_add(d, d, d);
    fe_sub(e, b, a); fe_sub(f, d, c); fe_add(g, d, c); fe_add(h, b, a);
    fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}
/* dbl-2008-hwcd (a = -1) */
static void ge_dbl(ge* r, const ge* p) {
    fe a, b, c, d, e, f, g, h, t;
    fe_sq(a, p->X); fe_sq(b, p->Y); fe_sq(c, p->Z); fe_add(c, c, c);
    fe_neg(d, a);
    fe_add(t, p->X, p->Y); fe_sq(t, t); fe_sub(e, t, a); fe_sub(e, e, b);
    fe_add(g, d, b); fe_sub(f, g, c); fe_sub(h, d, b);
    fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}
static void ge_neg(ge* r, const ge* p) {
    fe_neg(r->X, p->X); fe_copy(r->Y, p->Y); fe_copy(r->Z, p->Z); fe_neg(r->T, p->T);
}
/* RFC 8032 §5.1.3; returns 0 on failure */
static int ge_decompress(ge* p, const uint8_t s[32]) {
    /* canonical y: y < p */
    static const uint8_t PB[32] = {0xed, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                   0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x7f};
    int sign = s[31] >> 7;
    uint8_t y[32];
    memcpy(y, s, 32);
    y[31] &= 0x7f;
    int lt = 0;
    for (int i = 31; i >= 0; i--) {
        if (y[i] < PB[i]) { lt = 1; break; }
        if (y[i] > PB[i]) { lt = 0; break; }
    }
    if (!lt) return 0;
    fe u, v, v3, x, vxx, chk;
    fe_frombytes(p->Y, y);
    fe_1(p->Z);
    fe_sq(u, p->Y);
    fe_mul(v, u, FE_D);
    fe_sub(u, u, p->Z); /* y^2 - 1 */
    fe_add(v, v, p->Z); /* d y^2 + 1 */
    fe_sq(v3, v); fe_mul(v3, v3, v);
    fe_sq(x, v3); fe_mul(x, x, v); fe_mul(x, x, u); /* u v^7 */
    fe_pow22523(x, x);
    fe_mul(x, x, v3); fe_mul(x, x, u); /* u v^3 (u v^7)^((p-5)/8) */
    fe_sq(vxx, x); fe_mul(vxx, vxx, v);
    fe_sub(chk, vxx, u);
    if (!fe_iszero(chk)) {
        fe_add(chk, vxx, u);
        if (!fe_iszero(chk)) return 0;
        fe_mul(x, x, FE_SQRTM1);
    }
    if (fe_iszero(x) && sign) return 0;
    if (fe_isneg(x) != sign) fe_neg(x, x);
    fe_copy(p->X, x);
    fe_mul(p->T, p->X, p->Y);
    return 1;
}
static int ge_eq(const ge* p, const ge* q) {
    fe a, b;
    fe_mul(a, p->X, q->Z); fe_mul(b, q->X, p->Z);
    if (!fe_eq(a, b)) return 0;
    fe_mul(a, p->Y, q->Z); fe_mul(b, q->Y, p->Z);
    return fe_eq(a, b);
}

/* ---- scalars: bit-serial reduction, r = (2r + bit) mod L */
static int sc_geq_L(const uint64_t r[4]) {
    for (int i = 3; i >= 0; i--) {
        if (r[i] > SC_L[i]) return 1;
        if (r[i] < SC_L[i]) return 0;
    }
    return 1;
}
static void sc_sub_L(uint64_t r[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t t = r[i] - SC_L[i] - borrow;
        borrow = (r[i] < SC_L[i] + borrow) || (SC_L[i] + borrow < borrow);
        r[i] = t;
    }
}
void orc_sc_reduce64(const uint8_t in[64], uint8_t out[32]) {
    uint64_t r[4] = {0, 0, 0, 0};
    for (int bit = 511; bit >= 0; bit--) {
        uint64_t b = (in[bit >> 3] >> (bit & 7)) & 1;
        r[3] = r[3] << 1 | r[2] >> 63; r[2] = r[2] << 1 | r[1] >> 63; r[1] = r[1] << 1 | r[0] >> 63; r[0] = r[0] << 1 | b;
        if (sc_geq_L(r)) sc_sub_L(r);
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(r[i] >> (8 * j));
}
static int sc_is_canonical(const uint8_t s[32]) {
    uint64_t r[4];
    for (int i = 0; i < 4; i++) {
        r[i] = 0;
        for (int j = 7; j >= 0; j--) r[i] = r[i] << 8 | s[8 * i + j];
    }
    return !sc_geq_L(r);
}

int orc_ed25519_verify_h(const uint8_t pk[32], const uint8_t sig[64], const uint8_t h[32]) {
    ge A, R, nA, BnA, B, Q;
    if (!ge_decompress(&A, pk)) return 0;
    if (!ge_decompress(&R, sig)) return 0;
    const uint8_t* s = sig + 32;
    if (!sc_is_canonical(s)) return 0;
    fe_copy(B.X, FE_BX); fe_copy(B.Y, FE_BY); fe_1(B.Z); fe_copy(B.T, FE_BT);
    ge_neg(&nA, &A);
    ge_add(&BnA, &B, &nA);
    /* Q = [s]B + [h](-A) by Shamir's trick; accept iff Q == R */
    ge_identity(&Q);
    for (int bit = 255; bit >= 0; bit--) {
        ge_dbl(&Q, &Q);
        int sb = (s[bit >> 3] >> (bit & 7)) & 1, hb = (h[bit >> 3] >> (bit & 7)) & 1;
        if (sb && hb) ge_add(&Q, &Q, &BnA);
        else if (sb) ge_add(&Q, &Q, &B);
        else if (hb) ge_add(&Q, &Q, &nA);
    }
    return ge_eq(&Q, &R);
}

int orc_ed25519_verify(const uint8_t pk[32], const uint8_t* msg, size_t len, const uint8_t sig[64]) {
    uint8_t buf[64 + 1024], dig[64], h[32];
    if (len > 1024) return 0;
    memcpy(buf, sig, 32);
    memcpy(buf + 32, pk, 32);
    memcpy(buf + 64, msg, len);
    orc_sha512(buf, 64 + len, dig);
    orc_sc_reduce64(dig, h);
    return orc_ed25519_verify_h(pk, sig, h);
}
