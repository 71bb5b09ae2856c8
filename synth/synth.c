/*
 * synth/synth.c — seeded SYNTHETIC INPUT generator (chains of Tendermint headers, validator sets,
 * signed commits).  It only manufactures INPUTS for tests, smoke() and bench.py (there is no
 * network and no dataset); it computes none of the hot path's outputs and is not part of the
 * product library.  Self-contained on purpose (own SHA-256 / SHA-512 / Ed25519 *signing* code in
 * 16 x 16-bit limbs), so that chains signed here and then verified by the HIP kernels and by the
 * test oracle exercise three independent Ed25519 implementations.
 *
 * Workload definition: SURVEY.md §8(d) "synthetic inputs" — chain_id "celestia", heights from S,
 * header time 1.7e9 + 12 s * i with random nanos, version {block 11, app 1},
 * last_block_id.parts.total = 1, all 32-byte hashes uniform random including data_hash, proposer
 * 20 random bytes, V Ed25519 validators with power in [1e6, 5e7], block_id_flag = Commit, round 0,
 * per-validator timestamp = header time + U[10,12] s.  Byte formats follow the five mocha-4
 * fixture blocks.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bsx.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef i64 gf[16];
#define FOR(i, n) for (i = 0; i < n; ++i)

/* ------------------------------------------------------------------ PRNG (splitmix64) */
typedef struct { u64 s; } rng_t;
static u64 rng_next(rng_t* r) {
    u64 z = (r->s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void rng_bytes(rng_t* r, u8* out, size_t n) {
    while (n) {
        u64 v = rng_next(r);
        size_t k = n < 8 ? n : 8;
        memcpy(out, &v, k);
        out += k;
        n -= k;
    }
}

/* ------------------------------------------------------------------ SHA-256 / SHA-512 (compact) */
static u32 R32(u32 x, int c) { return (x >> c) | (x << (32 - c)); }
static u64 R64(u64 x, int c) { return (x >> c) | (x << (64 - c)); }
static const u32 K256[64] = {0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const u64 K512[80] = {0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL, 0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL, 0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL, 0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
static int g_init = 0;
static void init_tables(void);

static void sha256_block(u32 st[8], const u8* p) {
    u32 w[64], v[8];
    for (int i = 0; i < 16; i++) w[i] = (u32)p[4 * i] << 24 | (u32)p[4 * i + 1] << 16 | (u32)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++)
        w[i] = w[i - 16] + (R32(w[i - 15], 7) ^ R32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
               (R32(w[i - 2], 17) ^ R32(w[i - 2], 19) ^ (w[i - 2] >> 10));
    memcpy(v, st, 32);
    u32 a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6], h = v[7];
    for (int i = 0; i < 64; i++) {
        u32 t1 = h + (R32(e, 6) ^ R32(e, 11) ^ R32(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        u32 t2 = (R32(a, 2) ^ R32(a, 13) ^ R32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
static void sha256(const u8* m, size_t n, u8 out[32]) {
    u32 st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    if (!g_init) init_tables();
    size_t full = n / 64, rem = n - full * 64;
    for (size_t i = 0; i < full; i++) sha256_block(st, m + 64 * i);
    u8 buf[128];
    memset(buf, 0, 128);
    memcpy(buf, m + 64 * full, rem);
    buf[rem] = 0x80;
    size_t tl = rem < 56 ? 64 : 128;
    u64 bits = (u64)n * 8;
    for (int i = 0; i < 8; i++) buf[tl - 1 - i] = (u8)(bits >> (8 * i));
    for (size_t i = 0; i < tl / 64; i++) sha256_block(st, buf + 64 * i);
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)(st[i] >> 24); out[4 * i + 1] = (u8)(st[i] >> 16); out[4 * i + 2] = (u8)(st[i] >> 8); out[4 * i + 3] = (u8)st[i]; }
}

static void sha512_block(u64 st[8], const u8* p) {
    u64 w[80], v[8];
    for (int i = 0; i < 16; i++) { u64 x = 0; for (int j = 0; j < 8; j++) x = x << 8 | p[8 * i + j]; w[i] = x; }
    for (int i = 16; i < 80; i++)
        w[i] = w[i - 16] + (R64(w[i - 15], 1) ^ R64(w[i - 15], 8) ^ (w[i - 15] >> 7)) + w[i - 7] +
               (R64(w[i - 2], 19) ^ R64(w[i - 2], 61) ^ (w[i - 2] >> 6));
    memcpy(v, st, 64);
    for (int i = 0; i < 80; i++) {
        u64 t1 = v[7] + (R64(v[4], 14) ^ R64(v[4], 18) ^ R64(v[4], 41)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K512[i] + w[i];
        u64 t2 = (R64(v[0], 28) ^ R64(v[0], 34) ^ R64(v[0], 39)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        memmove(v + 1, v, 56);
        v[4] += t1;
        v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) st[i] += v[i];
}
static void sha512(const u8* m, size_t n, u8 out[64]) {
    u64 st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    if (!g_init) init_tables();
    size_t full = n / 128, rem = n - full * 128;
    for (size_t i = 0; i < full; i++) sha512_block(st, m + 128 * i);
    u8 buf[256];
    memset(buf, 0, 256);
    memcpy(buf, m + 128 * full, rem);
    buf[rem] = 0x80;
    size_t tl = rem < 112 ? 128 : 256;
    u64 bits = (u64)n * 8;
    for (int i = 0; i < 8; i++) buf[tl - 1 - i] = (u8)(bits >> (8 * i));
    for (size_t i = 0; i < tl / 128; i++) sha512_block(st, buf + 128 * i);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (u8)(st[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------ Ed25519 signing, 16 x 16-bit limbs */
static const gf gf0 = {0}, gf1 = {1};
static const gf D2 = {0xf159, 0x26b2, 0x9b94, 0xebd6, 0xb156, 0x8283, 0x149a, 0x00e0, 0xd130, 0xeef3, 0x80f2, 0x198e, 0xfce7, 0x56df, 0xd9dc, 0x2406};
static const gf BX = {0xd51a, 0x8f25, 0x2d60, 0xc956, 0xa7b2, 0x9525, 0xc760, 0x692c, 0xdc5c, 0xfdd6, 0xe231, 0xc0a4, 0x53fe, 0xcd6e, 0x36d3, 0x2169};
static const gf BY = {0x6658, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666, 0x6666};

static void set25519(gf r, const gf a) { int i; FOR(i, 16) r[i] = a[i]; }
static void car25519(gf o) {
    int i;
    i64 c;
    FOR(i, 16) {
        o[i] += (1LL << 16);
        c = o[i] >> 16;
        o[(i + 1) * (i < 15)] += c - 1 + 37 * (c - 1) * (i == 15);
        o[i] -= c * 65536; /* not `c << 16`: c may be negative (UBSan) */
    }
}
static void sel25519(gf p, gf q, int b) {
    i64 t, i, c = ~(b - 1);
    FOR(i, 16) { t = c & (p[i] ^ q[i]); p[i] ^= t; q[i] ^= t; }
}
static void pack25519(u8* o, const gf n) {
    int i, j, b;
    gf m, t;
    FOR(i, 16) t[i] = n[i];
    car25519(t); car25519(t); car25519(t);
    FOR(j, 2) {
        m[0] = t[0] - 0xffed;
        for (i = 1; i < 15; i++) { m[i] = t[i] - 0xffff - ((m[i - 1] >> 16) & 1); m[i - 1] &= 0xffff; }
        m[15] = t[15] - 0x7fff - ((m[14] >> 16) & 1);
        b = (m[15] >> 16) & 1;
        m[14] &= 0xffff;
        sel25519(t, m, 1 - b);
    }
    FOR(i, 16) { o[2 * i] = t[i] & 0xff; o[2 * i + 1] = (u8)(t[i] >> 8); }
}
static void fA(gf o, const gf a, const gf b) { int i; FOR(i, 16) o[i] = a[i] + b[i]; }
static void fZ(gf o, const gf a, const gf b) { int i; FOR(i, 16) o[i] = a[i] - b[i]; }
static void fM(gf o, const gf a, const gf b) {
    i64 i, j, t[31];
    FOR(i, 31) t[i] = 0;
    FOR(i, 16) FOR(j, 16) t[i + j] += a[i] * b[j];
    FOR(i, 15) t[i] += 38 * t[i + 16];
    FOR(i, 16) o[i] = t[i];
    car25519(o);
    car25519(o);
}
static void inv25519(gf o, const gf i) {
    gf c;
    int a;
    FOR(a, 16) c[a] = i[a];
    for (a = 253; a >= 0; a--) { fM(c, c, c); if (a != 2 && a != 4) fM(c, c, i); }
    FOR(a, 16) o[a] = c[a];
}
static void padd(gf p[4], gf q[4]) {
    gf a, b, c, d, t, e, f, g, h;
    fZ(a, p[1], p[0]); fZ(t, q[1], q[0]); fM(a, a, t);
    fA(b, p[0], p[1]); fA(t, q[0], q[1]); fM(b, b, t);
    fM(c, p[3], q[3]); fM(c, c, D2);
    fM(d, p[2], q[2]); fA(d, d, d);
    fZ(e, b, a); fZ(f, d, c); fA(g, d, c); fA(h, b, a);
    fM(p[0], e, f); fM(p[1], h, g); fM(p[2], g, f); fM(p[3], e, h);
}
static void ppack(u8* r, gf p[4]) {
    gf tx, ty, zi;
    u8 xb[32];
    inv25519(zi, p[2]);
    fM(tx, p[0], zi);
    fM(ty, p[1], zi);
    pack25519(r, ty);
    pack25519(xb, tx);
    r[31] ^= (u8)((xb[0] & 1) << 7);
}
/* fixed-base table: TB[i][j] = (j+1) * 16^i * B */
static gf TB[64][15][4];
static void scalarbase(gf p[4], const u8* s) {
    int i, first = 1;
    FOR(i, 64) {
        int d = (s[i / 2] >> (4 * (i & 1))) & 15;
        if (!d) continue;
        if (first) { int k; FOR(k, 4) set25519(p[k], TB[i][d - 1][k]); first = 0; }
        else padd(p, TB[i][d - 1]);
    }
    if (first) { set25519(p[0], gf0); set25519(p[1], gf1); set25519(p[2], gf1); set25519(p[3], gf0); }
}
static const u64 LL[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
static void modL(u8* r, i64 x[64]) {
    i64 carry, i, j;
    for (i = 63; i >= 32; --i) {
        carry = 0;
        for (j = i - 32; j < i - 12; ++j) {
            x[j] += carry - 16 * x[i] * (i64)LL[j - (i - 32)];
            carry = (x[j] + 128) >> 8;
            x[j] -= carry * 256; /* carry may be negative: no `<<` (UBSan) */
        }
        x[j] += carry;
        x[i] = 0;
    }
    carry = 0;
    FOR(j, 32) {
        x[j] += carry - (x[31] >> 4) * (i64)LL[j];
        carry = x[j] >> 8;
        x[j] &= 255;
    }
    FOR(j, 32) x[j] -= carry * (i64)LL[j];
    FOR(i, 32) { x[i + 1] += x[i] >> 8; r[i] = (u8)(x[i] & 255); }
}
static void reduce64(u8* r) {
    i64 x[64], i;
    FOR(i, 64) x[i] = (u64)r[i];
    FOR(i, 64) r[i] = 0;
    modL(r, x);
}

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_tables_once(void) {
    g_init = 1;
    /* fixed-base table */
    gf base[4], cur[4];
    set25519(base[0], BX); set25519(base[1], BY); set25519(base[2], gf1); fM(base[3], BX, BY);
    for (int i = 0; i < 64; i++) {
        int k;
        FOR(k, 4) set25519(TB[i][0][k], base[k]);
        for (int j = 1; j < 15; j++) {
            FOR(k, 4) set25519(cur[k], TB[i][j - 1][k]);
            padd(cur, base);
            FOR(k, 4) set25519(TB[i][j][k], cur[k]);
        }
        /* base = 16 * base */
        FOR(k, 4) set25519(cur[k], TB[i][14][k]);
        padd(cur, base);
        FOR(k, 4) set25519(base[k], cur[k]);
    }
}
static void init_tables(void) { pthread_once(&g_once, init_tables_once); }

/* seed -> (clamped scalar a, prefix, public key) */
static void ed_keypair(const u8 seed[32], u8 pk[32], u8 d[64]) {
    gf p[4];
    init_tables();
    sha512(seed, 32, d);
    d[0] &= 248; d[31] &= 127; d[31] |= 64;
    scalarbase(p, d);
    ppack(pk, p);
}
static void ed_sign(const u8 d[64], const u8 pk[32], const u8* m, size_t n, u8 sig[64]) {
    u8 buf[64 + 256], r[64], h[64];
    i64 x[64], i, j;
    gf p[4];
    memcpy(buf, d + 32, 32);
    memcpy(buf + 32, m, n);
    sha512(buf, 32 + n, r);
    reduce64(r);
    scalarbase(p, r);
    ppack(sig, p);
    memcpy(buf, sig, 32);
    memcpy(buf + 32, pk, 32);
    memcpy(buf + 64, m, n);
    sha512(buf, 64 + n, h);
    reduce64(h);
    FOR(i, 64) x[i] = 0;
    FOR(i, 32) x[i] = (u64)r[i];
    FOR(i, 32) FOR(j, 32) x[i + j] += (i64)h[i] * (i64)d[j];
    modL(sig + 32, x);
}

/* ------------------------------------------------------------------ Tendermint encodings */
static int put_varint(u8* o, u64 v) {
    int n = 0;
    while (v >= 0x80) { o[n++] = (u8)(v | 0x80); v >>= 7; }
    o[n++] = (u8)v;
    return n;
}
static void leaf_hash(const u8* x, size_t n, u8 out[32]) {
    u8 b[1 + 128];
    b[0] = 0;
    memcpy(b + 1, x, n);
    sha256(b, n + 1, out);
}
static void inner_hash(const u8 l[32], const u8 r[32], u8 out[32]) {
    u8 b[65];
    b[0] = 1;
    memcpy(b + 1, l, 32);
    memcpy(b + 33, r, 32);
    sha256(b, 65, out);
}
static void tree_root(u8 (*hashes)[32], size_t n, u8 out[32]) {
    if (n == 0) { sha256((const u8*)"", 0, out); return; }
    if (n == 1) { memcpy(out, hashes[0], 32); return; }
    size_t k = 1;
    while (k * 2 < n) k *= 2;
    u8 l[32], r[32];
    tree_root(hashes, k, l);
    tree_root(hashes + k, n - k, r);
    inner_hash(l, r, out);
}
static void header_root(const bsx_header* h, u8 out[32]) {
    u8 lh[14][32];
    const u8* f[14] = {h->version, h->chain_id, h->height, h->time, h->last_block_id, h->hash[0], h->hash[1], h->hash[2],
                       h->hash[3], h->hash[4], h->hash[5], h->hash[6], h->hash[7], h->proposer};
    for (int i = 0; i < 14; i++) leaf_hash(f[i], h->len[i], lh[i]);
    tree_root(lh, 14, out);
}

/* ------------------------------------------------------------------ public API */
uint32_t synth_version(void) { return 1; }

/* V validators: 32-byte secret seeds (out), public keys, voting powers in [1e6, 5e7]; returns validators_hash */
int synth_validator_set(uint64_t seed, uint32_t v, uint8_t* sk_seeds, uint8_t* pubkeys, uint64_t* powers,
                        uint8_t validators_hash[32]) {
    rng_t r = {seed ^ 0x5e7d5e7d5e7dULL};
    u8(*lh)[32] = malloc((size_t)(v ? v : 1) * 32);
    for (uint32_t i = 0; i < v; i++) {
        u8 d[64], leaf[48];
        rng_bytes(&r, sk_seeds + 32 * i, 32);
        ed_keypair(sk_seeds + 32 * i, pubkeys + 32 * i, d);
        powers[i] = 1000000 + rng_next(&r) % 49000001ULL;
        int n = 0;
        leaf[n++] = 0x0a; leaf[n++] = 0x22; leaf[n++] = 0x0a; leaf[n++] = 0x20;
        memcpy(leaf + n, pubkeys + 32 * i, 32);
        n += 32;
        leaf[n++] = 0x10;
        n += put_varint(leaf + n, powers[i]);
        leaf_hash(leaf, (size_t)n, lh[i]);
    }
    tree_root(lh, v, validators_hash);
    free(lh);
    return 0;
}

/* A validator set derived from another: `n_replace` slots (chosen by `seed`) get a fresh key pair and power, the rest keep theirs —
 * how a chain's validator set drifts between the ranges of a workload (keys rotate, validators join and leave).  In / out arrays as
 * synth_validator_set; returns the new set's validators_hash. */
int synth_validator_set_rotate(uint64_t seed, uint32_t v, uint32_t n_replace, uint8_t* sk_seeds, uint8_t* pubkeys, uint64_t* powers,
                               uint8_t validators_hash[32]) {
    rng_t r = {seed ^ 0x0207a7e0207a7eULL};
    u8(*lh)[32] = malloc((size_t)(v ? v : 1) * 32);
    for (uint32_t k = 0; k < n_replace && v; k++) {
        const uint32_t i = (uint32_t)(rng_next(&r) % v);
        u8 d[64];
        rng_bytes(&r, sk_seeds + 32 * i, 32);
        ed_keypair(sk_seeds + 32 * i, pubkeys + 32 * i, d);
        powers[i] = 1000000 + rng_next(&r) % 49000001ULL;
    }
    for (uint32_t i = 0; i < v; i++) {
        u8 leaf[48];
        int n = 0;
        leaf[n++] = 0x0a; leaf[n++] = 0x22; leaf[n++] = 0x0a; leaf[n++] = 0x20;
        memcpy(leaf + n, pubkeys + 32 * i, 32);
        n += 32;
        leaf[n++] = 0x10;
        n += put_varint(leaf + n, powers[i]);
        leaf_hash(leaf, (size_t)n, lh[i]);
    }
    tree_root(lh, v, validators_hash);
    free(lh);
    return 0;
}

/* n linked headers at heights start_height..; header 0 gets a random last_block_id.  out_hashes: n x 32.
 * time_secs0: time of header 0 (header i is +12 s * i). */
int synth_chain(uint64_t seed, const char* chain_id, uint64_t start_height, uint64_t n, uint64_t time_secs0,
                const uint8_t validators_hash[32], bsx_header* out, uint8_t* out_hashes) {
    rng_t r = {seed};
    size_t cl = strlen(chain_id);
    if (cl > 50) return 1;
    u8 prev[32];
    rng_bytes(&r, prev, 32);
    for (uint64_t i = 0; i < n; i++) {
        bsx_header* h = &out[i];
        memset(h, 0, sizeof *h);
        int k = 0;
        h->version[k++] = 0x08; h->version[k++] = 11; h->version[k++] = 0x10; h->version[k++] = 1;
        h->len[0] = (u8)k;
        h->chain_id[0] = 0x0a; h->chain_id[1] = (u8)cl;
        memcpy(h->chain_id + 2, chain_id, cl);
        h->len[1] = (u8)(2 + cl);
        h->height[0] = 0x08;
        h->len[2] = (u8)(1 + put_varint(h->height + 1, start_height + i));
        k = 0;
        h->time[k++] = 0x08;
        k += put_varint(h->time + k, time_secs0 + 12 * i);
        u64 nanos = rng_next(&r) % 1000000000ULL;
        if (nanos) { h->time[k++] = 0x10; k += put_varint(h->time + k, nanos); }
        h->len[3] = (u8)k;
        u8* b = h->last_block_id;
        b[0] = 0x0a; b[1] = 0x20;
        memcpy(b + 2, prev, 32);
        b[34] = 0x12; b[35] = 0x24; b[36] = 0x08; b[37] = 0x01; b[38] = 0x12; b[39] = 0x20;
        rng_bytes(&r, b + 40, 32);
        h->len[4] = 72;
        for (int j = 0; j < 8; j++) {
            h->hash[j][0] = 0x0a; h->hash[j][1] = 0x20;
            if (j == 2 || j == 3) memcpy(h->hash[j] + 2, validators_hash, 32);
            else rng_bytes(&r, h->hash[j] + 2, 32);
            h->len[5 + j] = 34;
        }
        h->proposer[0] = 0x0a; h->proposer[1] = 0x14;
        rng_bytes(&r, h->proposer + 2, 20);
        h->len[13] = 22;
        header_root(h, prev);
        if (out_hashes) memcpy(out_hashes + 32 * i, prev, 32);
    }
    return 0;
}

typedef struct {
    uint64_t seed;
    const char* chain_id;
    uint32_t n_commits, v, v_max, tid, nthreads;
    const uint64_t* heights;
    const uint8_t* block_hashes;
    const uint64_t* time_secs;
    const uint8_t *sk_seeds, *pubkeys;
    const uint64_t* powers;
    uint32_t absent_permille;
    bsx_validator* out;
    uint64_t round;            /* commit round: != 0 adds the CanonicalVote round field (block hash then sits at offset 25) */
    uint32_t nil_permille;     /* validators that voted NIL (BlockIDFlagNil): a valid signature over a vote WITHOUT block id,
                                  is_signed = 0 — the slot carries bytes that must not be counted */
} cjob_t;

static void* commit_worker(void* arg) {
    cjob_t* j = arg;
    size_t cl = strlen(j->chain_id);
    u8(*dk)[64] = malloc((size_t)j->v * 64);
    for (uint32_t i = 0; i < j->v; i++) {
        sha512(j->sk_seeds + 32 * i, 32, dk[i]);
        dk[i][0] &= 248; dk[i][31] &= 127; dk[i][31] |= 64;
    }
    for (uint32_t c = j->tid; c < j->n_commits; c += j->nthreads) {
        rng_t r = {j->seed + 0x1000003ULL * (c + 1)};
        u8 parts_hash[32];
        rng_bytes(&r, parts_hash, 32);
        for (uint32_t i = 0; i < j->v_max; i++) {
            bsx_validator* o = &j->out[(size_t)c * j->v_max + i];
            memset(o, 0, sizeof *o);
            if (i >= j->v) continue;
            o->enabled = 1;
            o->present_on_trusted = 1;
            o->voting_power = j->powers[i];
            memcpy(o->pubkey, j->pubkeys + 32 * i, 32);
            int absent = (rng_next(&r) % 1000) < j->absent_permille;
            if (absent) continue;
            const int nil = j->nil_permille && (rng_next(&r) % 1000) < j->nil_permille;
            o->is_signed = nil ? 0 : 1;
            /* CanonicalVote sign-bytes */
            u8 body[160], ts[16];
            int n = 0, tn = 0;
            body[n++] = 0x08; body[n++] = 0x02;
            body[n++] = 0x11;
            for (int b = 0; b < 8; b++) body[n++] = (u8)(j->heights[c] >> (8 * b));
            if (j->round) {
                body[n++] = 0x19;
                for (int b = 0; b < 8; b++) body[n++] = (u8)(j->round >> (8 * b));
            }
            if (!nil) {
                body[n++] = 0x22; body[n++] = 0x48; body[n++] = 0x0a; body[n++] = 0x20;
                memcpy(body + n, j->block_hashes + 32 * c, 32); n += 32;
                body[n++] = 0x12; body[n++] = 0x24; body[n++] = 0x08; body[n++] = 0x01; body[n++] = 0x12; body[n++] = 0x20;
                memcpy(body + n, parts_hash, 32); n += 32;
            }
            u64 secs = j->time_secs[c] + 10 + rng_next(&r) % 2, nanos = rng_next(&r) % 1000000000ULL;
            ts[tn++] = 0x08; tn += put_varint(ts + tn, secs);
            if (nanos) { ts[tn++] = 0x10; tn += put_varint(ts + tn, nanos); }
            body[n++] = 0x2a; body[n++] = (u8)tn;
            memcpy(body + n, ts, (size_t)tn); n += tn;
            body[n++] = 0x32; body[n++] = (u8)cl;
            memcpy(body + n, j->chain_id, cl); n += (int)cl;
            o->message[0] = (u8)n; /* n < 128: one-byte varint length prefix */
            memcpy(o->message + 1, body, (size_t)n);
            o->message_len = (uint32_t)(n + 1);
            ed_sign(dk[i], o->pubkey, o->message, o->message_len, o->signature);
        }
    }
    free(dk);
    return NULL;
}

/* n_commits commits; commit c signs block_hashes[c] at heights[c]; every validator < v signs unless drawn absent
 * (absent_permille).  out: n_commits * v_max validators (slots >= v disabled, zero). */
int synth_commits(uint64_t seed, const char* chain_id, uint32_t n_commits, const uint64_t* heights,
                  const uint8_t* block_hashes, const uint64_t* time_secs, uint32_t v, uint32_t v_max,
                  const uint8_t* sk_seeds, const uint8_t* pubkeys, const uint64_t* powers, uint32_t absent_permille,
                  int n_threads, bsx_validator* out, uint64_t round, uint32_t nil_permille) {
    if (strlen(chain_id) > (round ? 31 : 40) || v > v_max) return 1; /* keeps the sign-bytes <= 124 */
    init_tables();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    cjob_t jobs[64];
    for (int t = 0; t < n_threads; t++) {
        cjob_t j = {seed, chain_id, n_commits, v, v_max, (uint32_t)t, (uint32_t)n_threads, heights, block_hashes, time_secs,
                    sk_seeds, pubkeys, powers, absent_permille, out, round, nil_permille};
        jobs[t] = j;
        pthread_create(&th[t], NULL, commit_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    return 0;
}

/* raw helpers, exported so tests can cross-check this file against hashlib and the oracle */
void synth_sha256(const uint8_t* m, size_t n, uint8_t out[32]) { sha256(m, n, out); }
void synth_sha512(const uint8_t* m, size_t n, uint8_t out[64]) { sha512(m, n, out); }
void synth_ed25519_keypair(const uint8_t seed[32], uint8_t pk[32]) { u8 d[64]; ed_keypair(seed, pk, d); }
void synth_ed25519_sign(const uint8_t seed[32], const uint8_t* m, size_t n, uint8_t sig[64]) {
    u8 d[64], pk[32];
    if (n > 256) return;
    ed_keypair(seed, pk, d);
    ed_sign(d, pk, m, n, sig);
}
