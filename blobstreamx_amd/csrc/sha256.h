// sha256.h — register-resident SHA-256 for one lane (FIPS 180-4), shaped for the Tendermint Merkle tree.
//
// Replaces (device side) what the reference reaches through plonky2x `curta_sha256` / `sha256`
// (circuits/builder.rs:144-147,189-199,357-364,429-433,442) and tendermint Header::hash -> sha2
// (circuits/input.rs:250-261).  One lane = one independent hash chain; a digest lives as 8 big-endian
// words in VGPRs and never touches memory between tree levels.  Per 64-byte block: 64 rounds of
// (3 v_alignbit + v_xor3) x2, v_bfi x2, v_add3 x3 and 48 schedule steps — ~1450 VALU ops, no LDS.
#pragma once
#include "bsx_common.h"

namespace bsx {

struct Digest {
    uint32_t w[8];  // big-endian words h0..h7
};

BSX_HDI uint32_t sha256_k(int i) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    return K[i];
}

BSX_HDI void sha256_init(uint32_t st[8]) {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
}

struct Block16 {
    uint32_t w[16];  // message block, big-endian words
};

typedef uint32_t u32x8 __attribute__((vector_size(32)));
typedef uint32_t u32x16 __attribute__((vector_size(64)));

// One compression, by value in / by value out, NOT inlined on the device: the kernels call it from ~40 sites and a
// fully unrolled body is ~13 KB of code; one shared copy keeps the instruction cache warm.  Native vector types so
// that the AMDGPU calling convention keeps all 24 argument dwords and the 8 result dwords in VGPRs (a 64-byte struct
// argument is passed byval through scratch memory instead).
BSX_HD_NOINLINE u32x8 sha256_compress_v(u32x8 stv, u32x16 wv) {
    uint32_t a = stv[0], b = stv[1], c = stv[2], d = stv[3], e = stv[4], f = stv[5], g = stv[6], h = stv[7];
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = wv[k];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        uint32_t ch = g ^ (e & (f ^ g));                 // v_bitop3_b32
        uint32_t t1 = h + S1 + ch + sha256_k(i) + w[i & 15];
        uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        uint32_t maj = b ^ ((a ^ b) & (c ^ b));          // v_bitop3_b32
        uint32_t t2 = S0 + maj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    u32x8 o;
    o[0] = stv[0] + a; o[1] = stv[1] + b; o[2] = stv[2] + c; o[3] = stv[3] + d;
    o[4] = stv[4] + e; o[5] = stv[5] + f; o[6] = stv[6] + g; o[7] = stv[7] + h;
    return o;
}
BSX_HDI Digest sha256_compress_fn(Digest st, Block16 blk) {
    u32x8 s;
    u32x16 b;
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = st.w[k];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = blk.w[k];
    s = sha256_compress_v(s, b);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = s[k];
    return o;
}
// array-style wrapper used by the message builders below
BSX_HDI void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    u32x8 s;
    u32x16 b;
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = st[k];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = w[k];
    s = sha256_compress_v(s, b);
#pragma unroll
    for (int k = 0; k < 8; k++) st[k] = s[k];
}

// inner_hash(l, r) = SHA256(0x01 ‖ l ‖ r): 65 bytes -> 2 blocks, built from the two register digests with
// 16 funnel shifts (no byte traffic).  Tendermint/RFC 6962 inner node.
BSX_HDI Digest inner_hash(const Digest& l, const Digest& r) {
    uint32_t st[8], w[16];
    sha256_init(st);
    w[0] = 0x01000000u | (l.w[0] >> 8);
#pragma unroll
    for (int k = 1; k < 8; k++) w[k] = funnel_r(l.w[k - 1], l.w[k], 8);
    w[8] = funnel_r(l.w[7], r.w[0], 8);
#pragma unroll
    for (int k = 9; k < 16; k++) w[k] = funnel_r(r.w[k - 9], r.w[k - 8], 8);
    sha256_compress(st, w);
    w[0] = (r.w[7] << 24) | 0x00800000u;
#pragma unroll
    for (int k = 1; k < 15; k++) w[k] = 0;
    w[15] = 65 * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// Message words of 0x00 ‖ data for a leaf whose bytes are given as little-endian dwords d[0..ND)
// (d[j] holds data bytes 4j..4j+3).  Big-endian message word k = bytes {data[4k-1], data[4k], data[4k+1], data[4k+2]}.
BSX_HDI uint32_t leaf_msg_word(uint32_t d_prev, uint32_t d_cur) {
    // bytes (MSB..LSB): d_prev.byte3, d_cur.byte0, d_cur.byte1, d_cur.byte2
    return (d_prev & 0xff000000u) | ((d_cur & 0xffu) << 16) | (d_cur & 0xff00u) | ((d_cur >> 16) & 0xffu);
}

// Keep the first `len` bytes of the LE dword stream, put 0x80 at byte `len`, zero the rest (dword j).
BSX_HDI uint32_t pad_dword(uint32_t d, int j, int len) {
    int r = len - 4 * j;               // bytes of this dword that are data
    if (r >= 4) return d;
    if (r < 0) return 0;
    uint32_t keep = (r == 0) ? 0u : (0xffffffffu >> (32 - 8 * r));
    return (d & keep) | (0x80u << (8 * r));
}

// leaf_hash(x) = SHA256(0x00 ‖ x) for len <= 54 (message <= 55 bytes: one block).
// d: 14 LE dwords of x (bytes beyond len are ignored).
BSX_HDI Digest leaf_hash_1block(const uint32_t* d, int len) {
    uint32_t st[8], w[16], p[14];
    sha256_init(st);
#pragma unroll
    for (int j = 0; j < 14; j++) p[j] = pad_dword(d[j], j, len);
    w[0] = leaf_msg_word(0, p[0]);
#pragma unroll
    for (int k = 1; k < 14; k++) w[k] = leaf_msg_word(p[k - 1], p[k]);
    w[14] = 0;
    w[15] = (uint32_t)(len + 1) * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// leaf_hash for 55 <= len <= 76, i.e. two blocks (the 72..76-byte last_block_id leaf). d: 19 LE dwords.
BSX_HDI Digest leaf_hash_2block(const uint32_t* d, int len) {
    uint32_t st[8], w[16], p[20];
    sha256_init(st);
#pragma unroll
    for (int j = 0; j < 19; j++) p[j] = pad_dword(d[j], j, len);
    p[19] = pad_dword(0, 19, len);
    w[0] = leaf_msg_word(0, p[0]);
#pragma unroll
    for (int k = 1; k < 16; k++) w[k] = leaf_msg_word(p[k - 1], p[k]);
    sha256_compress(st, w);
#pragma unroll
    for (int k = 16; k < 20; k++) w[k - 16] = leaf_msg_word(p[k - 1], p[k]);
#pragma unroll
    for (int k = 4; k < 15; k++) w[k] = 0;
    w[15] = (uint32_t)(len + 1) * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// Fixed-size leaves of the inclusion proofs (no masking needed: sizes are compile-time constants).
// 34-byte leaf (PROTOBUF_HASH_SIZE_BYTES, consts.rs:4): d[0..9) LE dwords, d[8] holds bytes 32,33.
BSX_HDI Digest leaf_hash_34(const uint32_t* d) {
    uint32_t st[8], w[16];
    sha256_init(st);
    w[0] = leaf_msg_word(0, d[0]);
#pragma unroll
    for (int k = 1; k < 8; k++) w[k] = leaf_msg_word(d[k - 1], d[k]);
    // word 8 = bytes data[31], data[32], data[33], 0x80
    w[8] = (d[7] & 0xff000000u) | ((d[8] & 0xffu) << 16) | (d[8] & 0xff00u) | 0x80u;
#pragma unroll
    for (int k = 9; k < 15; k++) w[k] = 0;
    w[15] = 35 * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// 72-byte leaf (PROTOBUF_BLOCK_ID_SIZE_BYTES, consts.rs:7): d[0..18) LE dwords. 73 bytes -> 2 blocks.
BSX_HDI Digest leaf_hash_72(const uint32_t* d) {
    uint32_t st[8], w[16];
    sha256_init(st);
    w[0] = leaf_msg_word(0, d[0]);
#pragma unroll
    for (int k = 1; k < 16; k++) w[k] = leaf_msg_word(d[k - 1], d[k]);
    sha256_compress(st, w);
    w[0] = leaf_msg_word(d[15], d[16]);
    w[1] = leaf_msg_word(d[16], d[17]);
    w[2] = (d[17] & 0xff000000u) | 0x00800000u;   // data[71], 0x80
#pragma unroll
    for (int k = 3; k < 15; k++) w[k] = 0;
    w[15] = 73 * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// 64-byte leaf (the data-root tuple, ENC_DATA_ROOT_TUPLE_SIZE_BYTES): given directly as 16 big-endian words t[].
// 65 bytes -> 2 blocks.  builder.rs:137,144-147 (leaf hash inside compute_root_from_leaves) and :442.
BSX_HDI Digest leaf_hash_tuple(const uint32_t t[16]) {
    uint32_t st[8], w[16];
    sha256_init(st);
    w[0] = t[0] >> 8;  // 0x00 prefix
#pragma unroll
    for (int k = 1; k < 16; k++) w[k] = funnel_r(t[k - 1], t[k], 8);
    sha256_compress(st, w);
    w[0] = (t[15] << 24) | 0x00800000u;
#pragma unroll
    for (int k = 1; k < 15; k++) w[k] = 0;
    w[15] = 65 * 8;
    sha256_compress(st, w);
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = st[k];
    return o;
}

// digest <-> bytes (little-endian dword view of the 32 digest bytes)
BSX_HDI uint32_t digest_le_dword(const Digest& x, int k) { return bswap32(x.w[k]); }
BSX_HDI Digest digest_from_le(const uint32_t* d) {
    Digest o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = bswap32(d[k]);
    return o;
}
BSX_HDI bool digest_eq(const Digest& a, const Digest& b) {
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) x |= a.w[k] ^ b.w[k];
    return x == 0;
}

}  // namespace bsx
