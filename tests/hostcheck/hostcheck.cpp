// tests/hostcheck/hostcheck.cpp — TEST HARNESS: compiles the device arithmetic headers
// (blobstreamx_amd/csrc/*.h, written __host__ __device__) with g++ so the exact kernel source can be
// checked against the oracle and hashlib on a machine without a GPU.  Never loaded by the product.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

#include "../../blobstreamx_amd/csrc/sha256.h"
#include "../../blobstreamx_amd/csrc/sha512.h"
#include "../../blobstreamx_amd/csrc/ed25519.h"
#include "../../blobstreamx_amd/csrc/poseidon.h"
#include "../../blobstreamx_amd/csrc/keycache.h"
#include "../../blobstreamx_amd/csrc/poseidon_consts.h"

using namespace bsx;

static void load_le(uint32_t* d, const uint8_t* p, int nbytes, int ndw) {
    for (int i = 0; i < ndw; i++) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) {
            int idx = 4 * i + b;
            if (idx < nbytes) v |= (uint32_t)p[idx] << (8 * b);
        }
        d[i] = v;
    }
}
static void store_digest(uint8_t* out, const Digest& x) {
    for (int k = 0; k < 8; k++) { out[4 * k] = x.w[k] >> 24; out[4 * k + 1] = x.w[k] >> 16; out[4 * k + 2] = x.w[k] >> 8; out[4 * k + 3] = x.w[k]; }
}
static Digest load_digest(const uint8_t* p) {
    uint32_t d[8];
    load_le(d, p, 32, 8);
    return digest_from_le(d);
}

extern "C" {
// garbage: bytes beyond len inside the dword view are filled with 0xa5 to prove they are ignored
void hc_leaf_hash_var(const uint8_t* data, int len, uint8_t out[32]) {
    uint8_t buf[80];
    memset(buf, 0xa5, sizeof buf);
    memcpy(buf, data, len);
    uint32_t d[20];
    load_le(d, buf, 80, 20);
    Digest r = (len <= 54) ? leaf_hash_1block(d, len) : leaf_hash_2block(d, len);
    store_digest(out, r);
}
void hc_leaf_hash_34(const uint8_t* data, uint8_t out[32]) { uint32_t d[9]; load_le(d, data, 34, 9); d[8] |= 0xa5a50000u; store_digest(out, leaf_hash_34(d)); }
void hc_leaf_hash_72(const uint8_t* data, uint8_t out[32]) { uint32_t d[18]; load_le(d, data, 72, 18); store_digest(out, leaf_hash_72(d)); }
void hc_leaf_hash_tuple(const uint8_t* data, uint8_t out[32]) {
    uint32_t t[16];
    for (int k = 0; k < 16; k++) t[k] = (uint32_t)data[4 * k] << 24 | (uint32_t)data[4 * k + 1] << 16 | (uint32_t)data[4 * k + 2] << 8 | data[4 * k + 3];
    store_digest(out, leaf_hash_tuple(t));
}
void hc_inner_hash(const uint8_t* l, const uint8_t* r, uint8_t out[32]) { store_digest(out, inner_hash(load_digest(l), load_digest(r))); }
void hc_sha512_ram(const uint8_t* r, const uint8_t* a, const uint8_t* m, int len, uint8_t out[64]) {
    uint32_t rd[8], ad[8], md[31], o[16];
    uint8_t buf[124];
    memset(buf, 0x5a, sizeof buf);
    memcpy(buf, m, len);
    load_le(rd, r, 32, 8); load_le(ad, a, 32, 8); load_le(md, buf, 124, 31);
    sha512_ram(rd, ad, md, len, o);
    memcpy(out, o, 64);
}
void hc_sc_reduce64(const uint8_t* in, uint8_t out[32]) {
    uint32_t i[16], o[8];
    load_le(i, in, 64, 16);
    sc_reduce64(i, o);
    memcpy(out, o, 32);
}
int hc_sc_is_canonical(const uint8_t* s) { uint32_t d[8]; load_le(d, s, 32, 8); return sc_is_canonical(d); }
int hc_ed25519_verify(const uint8_t* pk, const uint8_t* sig, const uint8_t* h) {
    uint32_t p[8], r[8], s[8], hh[8];
    load_le(p, pk, 32, 8); load_le(r, sig, 32, 8); load_le(s, sig + 32, 32, 8); load_le(hh, h, 32, 8);
    return ed25519_verify_core(p, r, s, hh) ? 1 : 0;
}
void hc_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t out[32]) {
    uint32_t x[8], y[8], o[8];
    load_le(x, a, 32, 8); load_le(y, b, 32, 8);
    fe_tobytes(o, fe_mul(fe_frombytes(x), fe_frombytes(y)));
    memcpy(out, o, 32);
}
void hc_fe_sq(const uint8_t* a, int dbl, uint8_t out[32]) {
    uint32_t x[8], o[8];
    load_le(x, a, 32, 8);
    fe_tobytes(o, dbl ? fe_sq2(fe_frombytes(x)) : fe_sq(fe_frombytes(x)));
    memcpy(out, o, 32);
}
void hc_fe_invert(const uint8_t* a, uint8_t out[32]) {
    uint32_t x[8], o[8];
    load_le(x, a, 32, 8);
    fe_tobytes(o, fe_invert(fe_frombytes(x)));
    memcpy(out, o, 32);
}
// fixed-key path: the table of a point (W-bit digits: parts x 2^(W-1) affine entries) built on the host by repeated
// addition + one batch inversion per part — NOT the device's builder (double-and-add + an inversion per entry, a minute
// per wide table on one host core): an independently built table under the device's verification code; the device-built
// tables are checked by the GPU parity tests.  pk = encoding of the NEGATED point.
static bool hc_build_table(const uint32_t pk[8], int32_t* tab, int W, int parts, int half) {
    ge_p3 base;
    const bool ok = ge_frombytes_negate(base, pk);
    std::vector<ge_p3> pts(half);
    std::vector<fe> pre(half);
    for (int part = 0; part < parts; part++) {
        if (part) base = ge_keytable_next_base(base, W);
        const ge_cached cb = p3_to_cached(base);
        ge_p3 acc = base;
        fe prod = fe_one();
        for (int j = 0; j < half; j++) {          // pts[j] = (j + 1) * base
            pts[j] = acc;
            prod = fe_mul(prod, acc.Z);
            pre[j] = prod;
            acc = p1p1_to_p3(ge_add(acc, cb));
        }
        fe inv = fe_invert(prod);
        for (int j = half - 1; j >= 0; j--) {
            const fe zi = j ? fe_mul(inv, pre[j - 1]) : inv;
            inv = fe_mul(inv, pts[j].Z);
            const fe x = fe_mul(pts[j].X, zi), y = fe_mul(pts[j].Y, zi);
            precomp_store(tab + ((size_t)part * half + j) * KT_ENTRY_I32, ge_precomp{fe_add(y, x), fe_sub(y, x), fe_mul(fe_mul(x, y), fe_d2())});
        }
    }
    return ok;
}
int hc_ed25519_verify_keyed(const uint8_t* pk, const uint8_t* sig, const uint8_t* h) {
    uint32_t p[8], r[8], s[8], hh[8];
    load_le(p, pk, 32, 8); load_le(r, sig, 32, 8); load_le(s, sig + 32, 32, 8); load_le(hh, h, 32, 8);
    static int32_t* btab = static_cast<int32_t*>(aligned_alloc(128, (size_t)BT_I32 * 4));
    static const bool b_ok = hc_build_table(GE_NEG_B_ENC, btab, BT_W, BT_PARTS, BT_HALF_ENTRIES);
    // the table of the most recent key is kept (tables persist across calls on the device too; the tests mostly repeat a key)
    static thread_local int32_t* tab = static_cast<int32_t*>(aligned_alloc(128, (size_t)KT_KEY_I32 * 4));
    static thread_local uint32_t tab_pk[8];
    static thread_local int tab_state = -1;                   // -1 none, 0 key does not decode, 1 built
    if (tab_state < 0 || memcmp(tab_pk, p, 32) != 0) {
        tab_state = hc_build_table(p, tab, KT_W, KT_PARTS, KT_HALF_ENTRIES) ? 1 : 0;
        memcpy(tab_pk, p, 32);
    }
    if (!b_ok || tab_state != 1) return 0;
    return ed25519_verify_keyed_core(tab, btab, r, s, hh) ? 1 : 0;
}
// the same over a key table with KT_W_WIDE-bit digits (round 5: the digit width is a run-time property of a table; 64 MB per key)
int hc_ed25519_verify_keyed_wide(const uint8_t* pk, const uint8_t* sig, const uint8_t* h) {
    uint32_t p[8], r[8], s[8], hh[8];
    load_le(p, pk, 32, 8); load_le(r, sig, 32, 8); load_le(s, sig + 32, 32, 8); load_le(hh, h, 32, 8);
    static int32_t* btab = static_cast<int32_t*>(aligned_alloc(128, (size_t)BT_I32 * 4));
    static const bool b_ok = hc_build_table(GE_NEG_B_ENC, btab, BT_W, BT_PARTS, BT_HALF_ENTRIES);
    static thread_local int32_t* tab = static_cast<int32_t*>(aligned_alloc(128, (size_t)kt_key_i32(KT_W_WIDE) * 4));
    static thread_local uint32_t tab_pk[8];
    static thread_local int tab_state = -1;
    if (tab_state < 0 || memcmp(tab_pk, p, 32) != 0) {
        tab_state = hc_build_table(p, tab, KT_W_WIDE, kt_parts(KT_W_WIDE), kt_half(KT_W_WIDE)) ? 1 : 0;
        memcpy(tab_pk, p, 32);
    }
    if (!b_ok || tab_state != 1) return 0;
    return ed25519_verify_keyed_core(tab, btab, r, s, hh, KT_W_WIDE) ? 1 : 0;
}

// ---- Goldilocks / Poseidon (goldilocks.h, poseidon.h): the device source on the host
uint64_t hc_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t hc_gl_add_canon(uint64_t a, uint64_t c) { return gl_add_canon(a, c); }
uint64_t hc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t hc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t hc_gl_pow7(uint64_t a) { return gl_pow7(a); }
uint64_t hc_gl_reduce128(uint64_t lo, uint64_t hi) { return gl_reduce128(lo, hi); }
uint64_t hc_gl_canonical(uint64_t a) { return gl_canonical(a); }
static const uint64_t HC_RC[BSX_POSEIDON_TABLE_N] = {BSX_POSEIDON_TABLE};
const uint64_t* hc_poseidon_rc(void) { return HC_RC; }
void hc_poseidon_mds(uint64_t s[12]) { poseidon_mds(s); }
// limbs a0 (signed) + 2^22 a1 + 2^44 a2 -> canonical word (the partial rounds' hand-over, poseidon.h poseidon_recombine_signed)
uint64_t hc_poseidon_recombine_signed(int32_t a0, uint32_t a1, uint32_t a2) { return gl_canonical(poseidon_recombine_signed((uint32_t)a0, a1, a2)); }
void hc_poseidon_permute(uint64_t s[12]) { poseidon_permute(s, HC_RC); for (int i = 0; i < 12; i++) s[i] = gl_canonical(s[i]); }
void hc_poseidon_hash(const uint64_t* in, uint64_t n, int noop, uint64_t out[4]) {
    auto get = [&](uint64_t k) -> uint64_t { return in[k]; };
    if (noop) poseidon_hash_or_noop(get, n, HC_RC, out); else poseidon_hash_no_pad(get, n, HC_RC, out);
}
void hc_poseidon_two_to_one(const uint64_t* l, const uint64_t* r, uint64_t out[4]) { poseidon_two_to_one(l, r, HC_RC, out); }

// ---- keycache.h (host-side bookkeeping of the fixed-key tables' rows; round 5): a cache object driven batch by batch
void* hc_keycache_new(uint32_t V, uint32_t N) { auto* k = new bsx_keycache(); k->init(V, N); return k; }
void hc_keycache_free(void* k) { delete static_cast<bsx_keycache*>(k); }
// one batch: rows_out [R * V]; dirty_out [N] (first *n_dirty entries); returns 1 when the map is the identity
int hc_keycache_assign(void* k_, const bsx_validator* sets, uint32_t R, uint32_t* rows_out, uint32_t* dirty_out, uint32_t* n_dirty, uint64_t* deferred) {
    auto* k = static_cast<bsx_keycache*>(k_);
    std::vector<uint32_t> d;
    const bool id = k->assign(sets, R, rows_out, d, deferred);
    *n_dirty = (uint32_t)d.size();
    for (size_t i = 0; i < d.size(); i++) dirty_out[i] = d[i];
    return id ? 1 : 0;
}
const uint8_t* hc_keycache_keys(void* k_) { return static_cast<bsx_keycache*>(k_)->keys.data(); }
const uint8_t* hc_keycache_used(void* k_) { return static_cast<bsx_keycache*>(k_)->used.data(); }
}
