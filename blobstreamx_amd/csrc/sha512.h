// sha512.h — register-resident SHA-512 for one lane: the Ed25519 challenge hash SHA512(R ‖ A ‖ M).
// Device side of P6: inside builder.skip / builder.step (circuits/header_range.rs:42-48,
// circuits/next_header.rs:32-36) the reference reaches it through plonky2x curta_eddsa_verify_sigs_conditional
// -> curta SHA-512 [UPSTREAM].  64-bit words are register pairs; rotates lower to v_alignbit_b32 pairs.
#pragma once
#include "bsx_common.h"

namespace bsx {

BSX_HDI uint64_t sha512_k(int i) {
    constexpr uint64_t K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
        0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
        0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
        0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
        0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
        0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
        0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
        0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
        0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
        0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
        0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
        0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
        0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    return K[i];
}

BSX_HDI uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

BSX_HDI void sha512_init(uint64_t st[8]) {
    st[0] = 0x6a09e667f3bcc908ULL; st[1] = 0xbb67ae8584caa73bULL; st[2] = 0x3c6ef372fe94f82bULL; st[3] = 0xa54ff53a5f1d36f1ULL;
    st[4] = 0x510e527fade682d1ULL; st[5] = 0x9b05688c2b3e6c1fULL; st[6] = 0x1f83d9abfb41bd6bULL; st[7] = 0x5be0cd19137e2179ULL;
}

// w[16]: message block as big-endian 64-bit words; clobbered.
BSX_HDI void sha512_compress(uint64_t st[8], uint64_t w[16]) {
    uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    // 5 x 16 rounds: the inner 16 are unrolled (static w[] indices), the outer loop stays rolled to keep the code small
#pragma unroll 1
    for (int o = 0; o < 80; o += 16)
#pragma unroll
    for (int ii = 0; ii < 16; ii++) {
        const int i = o + ii;
        if (o > 0) {
            uint64_t w15 = w[(ii + 1) & 15], w2 = w[(ii + 14) & 15];
            uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
            uint64_t s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
            w[ii] = w[ii] + s0 + w[(ii + 9) & 15] + s1;
        }
        uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
        uint64_t ch = g ^ (e & (f ^ g));
        uint64_t t1 = h + S1 + ch + sha512_k(i) + w[ii];
        uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
        uint64_t maj = b ^ ((a ^ b) & (c ^ b));
        uint64_t t2 = S0 + maj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// SHA512(R(32) ‖ A(32) ‖ M(len <= 124)) -> 64-byte digest as 16 little-endian dwords (byte order of the digest).
// Inputs as little-endian dwords: r[8], a[8], m[31] (message bytes beyond len are ignored).
// 64 + len <= 188 bytes -> always exactly 2 blocks (64+len+17 <= 256, and 64+len+17 > 128 iff len >= 48;
// shorter messages take 1 block: handled).
// dword j of the M part of the padded byte stream: message bytes below len, the 0x80 terminator, zeros above
BSX_HDI uint32_t sha512_ram_m_dword(uint32_t d, int j, int len) {
    const int rr = len - 4 * j;
    const uint32_t keep = (rr >= 4) ? 0xffffffffu : (rr <= 0 ? 0u : (0xffffffffu >> (32 - 8 * rr)));
    uint32_t v = d & keep;
    if (rr >= 0 && rr < 4) v |= 0x80u << (8 * rr);
    return v;
}
BSX_HDI void sha512_ram(const uint32_t r[8], const uint32_t a[8], const uint32_t* m, int len, uint32_t out_le[16]) {
    // byte stream as LE dwords: 16 dwords of R‖A, then up to 31 of M, then padding
    uint32_t s[64];
#pragma unroll
    for (int i = 0; i < 8; i++) { s[i] = r[i]; s[8 + i] = a[i]; }
#pragma unroll
    for (int j = 0; j < 32; j++) s[16 + j] = sha512_ram_m_dword((j < 31) ? m[j] : 0u, j, len);
#pragma unroll
    for (int j = 48; j < 64; j++) s[j] = 0;
    const uint64_t bits = (uint64_t)(64 + len) * 8;
    const bool two = (64 + len) >= 112;   // 0x80 + 16-byte length no longer fit the first block
    uint64_t st[8], w[16];
    sha512_init(st);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = ((uint64_t)bswap32(s[2 * k]) << 32) | bswap32(s[2 * k + 1]);
    if (!two) w[15] = bits;
    sha512_compress(st, w);
    if (two) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = ((uint64_t)bswap32(s[32 + 2 * k]) << 32) | bswap32(s[32 + 2 * k + 1]);
        w[15] = bits;
        sha512_compress(st, w);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        out_le[2 * k] = bswap32((uint32_t)(st[k] >> 32));
        out_le[2 * k + 1] = bswap32((uint32_t)st[k]);
    }
}

}  // namespace bsx
