// ed25519.h — scalar arithmetic mod L and the per-lane verification [s]B + [h](-A) == R.
// Device side of P6 (reduction of the SHA-512 challenge) and P7; reference call sites
// circuits/header_range.rs:42-48 (builder.skip) and circuits/next_header.rs:32-36 (builder.step).
// Accept set = RFC 8032 cofactorless verification with canonical A, R, s (same as the test oracle).
#pragma once
#include "fe25519.h"

namespace bsx {

// 512-bit little-endian integer (16 dwords) mod L -> 8 dwords.  21-bit signed limbs; 2^252 = -c (mod L) with
// -c = 666643 + 470296*2^21 + 654183*2^42 - 997805*2^63 + 136657*2^84 - 683901*2^105.
BSX_HDI void sc_fold(int64_t* s, int i) {
    const int64_t x = s[i];
    s[i - 12] += x * 666643;
    s[i - 11] += x * 470296;
    s[i - 10] += x * 654183;
    s[i - 9] -= x * 997805;
    s[i - 8] += x * 136657;
    s[i - 7] -= x * 683901;
    s[i] = 0;
}
BSX_HDI void sc_carry_round(int64_t* s, int i) {  // balanced
    const int64_t c = (s[i] + (1 << 20)) >> 21;
    s[i + 1] += c;
    s[i] -= c << 21;
}
BSX_HDI void sc_carry_floor(int64_t* s, int i) {
    const int64_t c = s[i] >> 21;
    s[i + 1] += c;
    s[i] -= c << 21;
}
BSX_HDI void sc_reduce64(const uint32_t in[16], uint32_t out[8]) {
    int64_t s[24];
#pragma unroll
    for (int i = 0; i < 23; i++) {
        const int o = 21 * i, wd = o >> 5, sh = o & 31;
        const uint64_t two = ((uint64_t)in[wd + 1] << 32) | in[wd];
        s[i] = (int64_t)((two >> sh) & 0x1fffff);
    }
    s[23] = (int64_t)(in[15] >> 3);
#pragma unroll
    for (int i = 23; i >= 18; i--) sc_fold(s, i);
#pragma unroll
    for (int i = 6; i <= 16; i += 2) sc_carry_round(s, i);
#pragma unroll
    for (int i = 7; i <= 15; i += 2) sc_carry_round(s, i);
#pragma unroll
    for (int i = 17; i >= 12; i--) sc_fold(s, i);
#pragma unroll
    for (int i = 0; i <= 10; i += 2) sc_carry_round(s, i);
#pragma unroll
    for (int i = 1; i <= 11; i += 2) sc_carry_round(s, i);
    sc_fold(s, 12);
#pragma unroll
    for (int i = 0; i <= 11; i++) sc_carry_floor(s, i);
    sc_fold(s, 12);
#pragma unroll
    for (int i = 0; i <= 10; i++) sc_carry_floor(s, i);
    // pack 12 limbs (21 bits each, the last one holds the remaining high bits) into 256 bits
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int o = 21 * i, wd = o >> 5, sh = o & 31;
        const uint64_t v = (uint64_t)s[i] << sh;
        out[wd] |= (uint32_t)v;
        if (wd + 1 < 8) out[wd + 1] |= (uint32_t)(v >> 32);
    }
}

// s < L ?  (8 LE dwords)
BSX_HDI bool sc_is_canonical(const uint32_t s[8]) {
    constexpr uint32_t Lw[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && s[i] != Lw[i]) { lt = s[i] < Lw[i]; decided = true; }
    }
    return lt;
}

// signed radix-16 recoding without a carry chain: r = s + 0x888...8; digit_i = nibble_i(r) - 8 in [-8, 7]
BSX_HDI void sc_recode(const uint32_t s[8], uint32_t r[8]) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)s[i] + 0x88888888u;
        r[i] = (uint32_t)c;
        c >>= 32;
    }
}
BSX_HDI uint32_t pick8(const uint32_t r[8], int w) {  // r[w] for a (wave-uniform) runtime w without private-memory indexing
    uint32_t v = r[0];
#pragma unroll
    for (int k = 1; k < 8; k++) v = (w == k) ? r[k] : v;
    return v;
}
BSX_HDI int sc_digit(const uint32_t r[8], int i) { return (int)((pick8(r, i >> 3) >> (4 * (i & 7))) & 15) - 8; }

BSX_HDI ge_precomp ge_b_entry(int k) {  // (k+1) * B
    ge_precomp e;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        e.yplusx.v[i] = ge_b_limb(k, i);
        e.yminusx.v[i] = ge_b_limb(k, 10 + i);
        e.xy2d.v[i] = ge_b_limb(k, 20 + i);
    }
    return e;
}

// true iff the signature (R, s) verifies for public key A with challenge h (already reduced mod L).
// pk, sig_r, sig_s, h: 8 LE dwords each.
BSX_HDI bool ed25519_verify_core(const uint32_t pk[8], const uint32_t sig_r[8], const uint32_t sig_s[8],
                                 const uint32_t h[8]) {
    bool ok = sc_is_canonical(sig_s);
    ge_p3 negA;
    ok = ge_frombytes_negate(negA, pk) && ok;

    // table k * (-A), k = 1..8, cached form (per-lane: private memory)
    ge_cached tab[8];
    tab[0] = p3_to_cached(negA);
    ge_p3 cur = negA;
    for (int k = 1; k < 8; k++) {
        cur = p1p1_to_p3(ge_add(cur, tab[0]));
        tab[k] = p3_to_cached(cur);
    }

    uint32_t hr[8], sr[8];
    sc_recode(h, hr);
    sc_recode(sig_s, sr);

    ge_p2 q{fe_zero(), fe_one(), fe_one()};
    for (int i = 63; i >= 0; i--) {
        ge_p1p1 t = ge_dbl(q.X, q.Y, q.Z);
        for (int d = 0; d < 3; d++) {
            q = p1p1_to_p2(t);
            t = ge_dbl(q.X, q.Y, q.Z);
        }
        ge_p3 p = p1p1_to_p3(t);

        const int da = sc_digit(hr, i);
        const int ia = da < 0 ? -da : da;
        ge_cached ca = tab[ia ? ia - 1 : 0];
        ca = cached_cneg(ca, da < 0);
        if (ia == 0) ca = cached_identity();
        p = p1p1_to_p3(ge_add(p, ca));

        const int db = sc_digit(sr, i);
        const int ib = db < 0 ? -db : db;
        ge_precomp pb = ge_b_entry(ib ? ib - 1 : 0);
        pb = precomp_cneg(pb, db < 0);
        if (ib == 0) pb = precomp_identity();
        q = p1p1_to_p2(ge_madd(p, pb));
    }
    uint32_t enc[8];
    ge_tobytes(enc, q.X, q.Y, q.Z);
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) diff |= enc[k] ^ sig_r[k];
    return ok && diff == 0;
}

// ------------------------------------------------------------------------------------------------ fixed-key path
// A validator set signs every commit of a range batch with the same keys, and B is everybody's key, so ALL the per-key
// work is hoisted out of the per-signature lane: per point P (a validator's -A, or B) a table holds j * 2^(W k) P for
// every digit position k and j = 1..2^(W-1) in AFFINE form (y + x, y - x, 2 d x y; 32 int32 = one 128-byte cache line per
// entry, 30 used).  The 253-bit scalars are recoded into signed radix-2^W digits (h: 22 of 12 bits, s: 16 of 16 bits), and
//     [s]B + [h](-A) = sum_k T_A[k][h_k] + sum_k T_B[k][s_k]
// is 22 + 16 mixed additions (7 multiplications each) and NO doubling, NO decompression.  History: 2 parts of 128 bits
// with cached (projective) entries = 128 doublings + 64 additions; 8 parts of 32 bits = 32 + 64 (round 2, 576 mul + 128
// sq); one radix-256 digit per part = 0 + 64 (447 mul); 16-bit digits for B = 0 + 48 (335 mul); 12-bit digits for the keys
// = 0 + 38 (265 mul).  A key's table is 22 x 2048 x 128 B = 5.8 MB, built once per key and kept while the key stays
// (kernels_ed.hip k_table_entries: batch inversion, 2 ms per 100 keys); the B table (16 x 32768 x 128 B = 64 MB) is built
// by the same code from the encoding of -B when a context is created.
constexpr int KT_ENTRY_I32 = 32;          // one affine entry, padded to a cache line
// Digit widths (bits): per-key tables (h) and the table of B (s).  A table has PARTS = ceil(254 / W) parts of 2^(W-1)
// entries (signed digits); a signature costs KT_PARTS + BT_PARTS additions.
// Per-key tables, measured (M verifies/s at 204,800 / 1,048,576 signatures, BSX_BT_W = 16): W = 8 (0.5 MB per key): 291 /
// 408, 10 (1.7 MB): 298 / 466, 12 (5.8 MB): 326 / 532, 13 (10.5 MB): 341 / 531.
#ifndef BSX_KT_W
#define BSX_KT_W 12
#endif
// The table of B is shared by every signature of every key, so it can afford wider digits than the per-key tables.  Measured
// (1,048,576 signatures, M verifies/s, BSX_KT_W = 8): BSX_BT_W = 8: 335, 10: 354, 11: 365, 12: 376, 13: 383, 16: 412 — the
// 64 MB table of W = 16 lives in the 256 MB Infinity Cache and its loads do not depend on the point arithmetic.
#ifndef BSX_BT_W
#define BSX_BT_W 16
#endif
constexpr int KT_W = BSX_KT_W, BT_W = BSX_BT_W;
constexpr int KT_PARTS = (253 + KT_W) / KT_W, BT_PARTS = (253 + BT_W) / BT_W;   // digits cover >= 254 bits: x + the recoding constant < 2^254
constexpr int KT_HALF_ENTRIES = 1 << (KT_W - 1), BT_HALF_ENTRIES = 1 << (BT_W - 1);   // j = 1..2^(W-1) per part
constexpr int KT_KEY_I32 = KT_PARTS * KT_HALF_ENTRIES * KT_ENTRY_I32;
constexpr int BT_I32 = BT_PARTS * BT_HALF_ENTRIES * KT_ENTRY_I32;
static_assert(KT_W >= 8 && KT_W <= 16 && KT_W * KT_PARTS >= 254 && KT_W * KT_PARTS <= 288, "key-table digit width");
static_assert(BT_W >= 8 && BT_W <= 16 && BT_W * BT_PARTS >= 254 && BT_W * BT_PARTS <= 288, "B-table digit width");
// Round 5: the digit width of a KEY table is a property of the table (KT_W is the default: request-driven paths, where a validator
// set may be new and its table is built on the spot).  A caller whose validator set is resident for many millions of signatures
// (mode S) asks for KT_W_WIDE-bit digits: 16 parts of 32,768 entries = 64 MB per key, 16 + 16 instead of 22 + 16 additions per
// signature (2048 x 100: verification 0.54 -> 0.49 ms; 6.4 GB of tables at V = 100).  Geometry of a w-bit table:
constexpr int KT_W_WIDE = 16;
BSX_HDI int kt_parts(int w) { return (253 + w) / w; }
BSX_HDI int kt_half(int w) { return 1 << (w - 1); }
BSX_HDI int64_t kt_key_i32(int w) { return (int64_t)kt_parts(w) * kt_half(w) * KT_ENTRY_I32; }
BSX_HDI bool kt_w_ok(int w) { return w == KT_W || w == KT_W_WIDE; }
// encoding of -B (B = (x, 4/5) with x even: the encoding of B is 0x58, 0x66 x 31; -B sets the sign bit of x)
constexpr uint32_t GE_NEG_B_ENC[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0xe6666666u};

// Signed radix-2^W recoding without a carry chain: r = x + sum_i 2^(W-1) 2^(W i) (9 dwords; x < 2^253);
// digit_i = ((r >> W i) mod 2^W) - 2^(W-1) in [-2^(W-1), 2^(W-1))
template <int W, int PARTS>
BSX_HDI void sc_recode_t(const uint32_t x[8], uint32_t r[9]) {
    uint32_t cst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PARTS; i++) {
        const int bit = W * i + W - 1;
        cst[bit >> 5] |= 1u << (bit & 31);
    }
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += (uint64_t)(i < 8 ? x[i] : 0u) + cst[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
}
BSX_HDI uint32_t pick9(const uint32_t r[9], int w) {  // r[w] for a runtime w without private-memory indexing
    uint32_t v = r[0];
#pragma unroll
    for (int k = 1; k < 9; k++) v = (w == k) ? r[k] : v;
    return v;
}
template <int W>
BSX_HDI int sc_digit_t(const uint32_t r[9], int i) {
    const int bit = W * i, wd = bit >> 5, sh = bit & 31;
    const uint64_t two = ((uint64_t)pick9(r, wd + 1 < 9 ? wd + 1 : 8) << 32) | pick9(r, wd);
    return (int)((uint32_t)(two >> sh) & ((1u << W) - 1)) - (1 << (W - 1));
}

// the same for a digit width known at run time (uniform over a launch): KT_W or KT_W_WIDE
BSX_HDI void sc_recode_rt(const uint32_t x[8], uint32_t r[9], int w) {
    if (w == KT_W_WIDE) sc_recode_t<KT_W_WIDE, (253 + KT_W_WIDE) / KT_W_WIDE>(x, r);
    else sc_recode_t<KT_W, KT_PARTS>(x, r);
}
BSX_HDI int sc_digit_rt(const uint32_t r[9], int i, int w) {
    const int bit = w * i, wd = bit >> 5, sh = bit & 31;
    const uint64_t two = ((uint64_t)pick9(r, wd + 1 < 9 ? wd + 1 : 8) << 32) | pick9(r, wd);
    return (int)((uint32_t)(two >> sh) & ((1u << w) - 1)) - (1 << (w - 1));
}

// base[k + 1] = 2^bits * base[k]
BSX_HDI ge_p3 ge_keytable_next_base(const ge_p3& prev, int bits = 8) {
    ge_p2 q{prev.X, prev.Y, prev.Z};
    ge_p1p1 t = ge_dbl_inl(q.X, q.Y, q.Z);
#pragma unroll 1
    for (int i = 1; i < bits; i++) {
        q = p1p1_to_p2_inl(t);
        t = ge_dbl_inl(q.X, q.Y, q.Z);
    }
    return p1p1_to_p3_inl(t);
}
// m * base, 1 <= m < 2^bits, by a bits-step double-and-add that is uniform across lanes (the addition is selected, not
// branched)
BSX_HDI ge_p3 ge_mul_small(const ge_p3& base, int m, int bits) {
    const ge_cached cb = p3_to_cached(base);
    ge_p3 acc{fe_zero(), fe_one(), fe_one(), fe_zero()};
#pragma unroll 1
    for (int bit = bits - 1; bit >= 0; bit--) {
        acc = p1p1_to_p3(ge_dbl(acc.X, acc.Y, acc.Z));
        const ge_p3 sum = p1p1_to_p3(ge_add(acc, cb));
        const bool take = ((m >> bit) & 1) != 0;
        acc.X = fe_select(take, sum.X, acc.X);
        acc.Y = fe_select(take, sum.Y, acc.Y);
        acc.Z = fe_select(take, sum.Z, acc.Z);
        acc.T = fe_select(take, sum.T, acc.T);
    }
    return acc;
}
// affine table entry of the point (X : Y : Z) given 1 / Z
BSX_HDI ge_precomp ge_to_precomp(const fe& X, const fe& Y, const fe& zinv) {
    const fe x = fe_mul(X, zinv), y = fe_mul(Y, zinv);
    return ge_precomp{fe_add(y, x), fe_sub(y, x), fe_mul(fe_mul(x, y), fe_d2())};
}
BSX_HDI void precomp_store(int32_t* dst, const ge_precomp& e) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        dst[i] = e.yplusx.v[i];
        dst[10 + i] = e.yminusx.v[i];
        dst[20 + i] = e.xy2d.v[i];
    }
    dst[30] = 0;
    dst[31] = 0;
}
BSX_HDI ge_precomp precomp_load(const int32_t* src_) {
    const int32_t* src = static_cast<const int32_t*>(__builtin_assume_aligned(src_, 16));
    ge_precomp e;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        e.yplusx.v[i] = src[i];
        e.yminusx.v[i] = src[10 + i];
        e.xy2d.v[i] = src[20 + i];
    }
    return e;
}
// entry |d| of one part's 128 entries, negated for d < 0, the identity for d == 0
BSX_HDI ge_precomp keytable_pick(const int32_t* part_tab, int d) {
    const int a = d < 0 ? -d : d;
    ge_precomp e = precomp_load(part_tab + (a ? a - 1 : 0) * KT_ENTRY_I32);
    e = precomp_cneg(e, d < 0);
    if (a == 0) e = precomp_identity();
    return e;
}

// Same accept set as ed25519_verify_core for a key whose table (key_tab: KT_KEY_I32 int32, [part][j-1][32]) was built from
// -A; b_tab: the table of B (BT_I32 int32, [part][j-1][32], BT_W-bit digits).  The caller has already established that the key decodes.
// DEFER: stop before the encoding (which costs a field inversion: 254 squarings + 11 multiplications, a third of a
// verification) and hand back the projective result; k_ed25519_finish then inverts the Z of several signatures per lane
// with ONE inversion (Montgomery's trick: 3 multiplications per extra element).
template <bool DEFER>
BSX_HDI bool ed25519_verify_keyed_core_t(const int32_t* key_tab, const int32_t* b_tab, const uint32_t sig_r[8], const uint32_t sig_s[8],
                                         const uint32_t h[8], ge_p2* out_q, int w = KT_W) {
    const bool ok = sc_is_canonical(sig_s);
    uint32_t hr[9], sr[9];
    sc_recode_rt(h, hr, w);
    sc_recode_t<BT_W, BT_PARTS>(sig_s, sr);
    ge_p3 p{fe_zero(), fe_one(), fe_one(), fe_zero()};
    const int parts = kt_parts(w);
    const int64_t part_i32 = (int64_t)kt_half(w) * KT_ENTRY_I32;
    // not unrolled on the device: entries prefetched several at a time would spill
#pragma unroll 1
    for (int k = 0; k < parts; k++)
        p = p1p1_to_p3(ge_madd(p, keytable_pick(key_tab + (int64_t)k * part_i32, sc_digit_rt(hr, k, w))));
#pragma unroll 1
    for (int k = 0; k < BT_PARTS - 1; k++)
        p = p1p1_to_p3(ge_madd(p, keytable_pick(b_tab + (int64_t)k * BT_HALF_ENTRIES * KT_ENTRY_I32, sc_digit_t<BT_W>(sr, k))));
    const ge_p2 q = p1p1_to_p2(ge_madd(p, keytable_pick(b_tab + (int64_t)(BT_PARTS - 1) * BT_HALF_ENTRIES * KT_ENTRY_I32, sc_digit_t<BT_W>(sr, BT_PARTS - 1))));
    if (DEFER) {
        *out_q = q;
        return ok;
    }
    uint32_t enc[8];
    ge_tobytes(enc, q.X, q.Y, q.Z);
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) diff |= enc[k] ^ sig_r[k];
    return ok && diff == 0;
}
BSX_HDI bool ed25519_verify_keyed_core(const int32_t* key_tab, const int32_t* b_tab, const uint32_t sig_r[8], const uint32_t sig_s[8],
                                       const uint32_t h[8], int w = KT_W) {
    return ed25519_verify_keyed_core_t<false>(key_tab, b_tab, sig_r, sig_s, h, nullptr, w);
}
// One of SPLIT partial sums of [s]B + [h](-A): the table parts k = part0 (mod SPLIT).  Without doublings the sum splits
// freely — SPLIT lanes per signature shorten the dependent chain from 48 to 48 / SPLIT additions (+ log2 SPLIT full
// additions to join, kernels_ed.hip) at 20 % more total work: the form for small batches, where latency is all there is.
template <int SPLIT>
BSX_HDI ge_p3 ed25519_keyed_partial(const int32_t* key_tab, const int32_t* b_tab, const uint32_t sig_s[8], const uint32_t h[8], int part0, int w = KT_W) {
    uint32_t hr[9], sr[9];
    sc_recode_rt(h, hr, w);
    sc_recode_t<BT_W, BT_PARTS>(sig_s, sr);
    ge_p3 p{fe_zero(), fe_one(), fe_one(), fe_zero()};
    const int parts = kt_parts(w);
    const int64_t part_i32 = (int64_t)kt_half(w) * KT_ENTRY_I32;
#pragma unroll 1
    for (int k = part0; k < parts; k += SPLIT)
        p = p1p1_to_p3(ge_madd(p, keytable_pick(key_tab + (int64_t)k * part_i32, sc_digit_rt(hr, k, w))));
#pragma unroll 1
    for (int k = part0; k < BT_PARTS; k += SPLIT)
        p = p1p1_to_p3(ge_madd(p, keytable_pick(b_tab + (int64_t)k * BT_HALF_ENTRIES * KT_ENTRY_I32, sc_digit_t<BT_W>(sr, k))));
    return p;
}

}  // namespace bsx
