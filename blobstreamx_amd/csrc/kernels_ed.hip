// kernels_ed.hip — commit verification side of the header_range hot path for gfx950 (CDNA4, wave64):
// the per-validator loop inside builder.skip / builder.step (circuits/header_range.rs:42-48,
// circuits/next_header.rs:32-36; circuit body [UPSTREAM] tendermintx v1.0.0; host twin is_valid_skip,
// circuits/fetcher.rs:76-80).
//
//   k_sha512_challenge  P6  h = SHA512(R ‖ A ‖ M) mod L          one lane per validator slot, one message block resident at a time
//   k_ed25519_verify    P7  [s]B + [h](-A) == R                   one lane per validator slot, ALU bound (no byte roofline)
//   k_keytable_check / k_table_entries / k_ed25519_verify_keyed(_small) / k_ed25519_finish
//                       P7, fixed-key form: per-validator tables of j*(-2^(12k) A) (k = 0..21, j = 1..2048, affine), rows kept
//                           across calls and rebuilt only when their key changes, and a 16-bit-digit table of B
//                           (k_btable_bases, per context; both filled by k_table_entries); a signature is 22 + 16 mixed
//                           additions of table entries picked by the signed digits of h and s; optionally the point encodings go through a per-lane batch
//                           inversion (k_ed25519_finish); slots whose key is not the table row's key are deferred to
//                           k_ed25519_verify<true> (same accept set)
//   k_skip_eval         operator skip-target search (fetcher.rs:60-87): is_valid_skip of every candidate in one launch
//   k_commit_tally     P8+P9 validator-set hash (masked Merkle tree), voting-power sums, message checks;
//                           one workgroup per commit, wave-shuffle + LDS reductions
//   k_skip_check        skip conditions of CombinedSkipCircuit (header_range.rs:42-48): header/validator-hash links,
//                           2/3 of the target set, > 1/3 of the trusted set
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"
#include "ed25519.h"
#include "kernels.h"
#include "sha256.h"
#include "sha512.h"


namespace bsx {

// ------------------------------------------------------------------------------------------------ k_sha512_challenge
// One lane per validator slot, each reading its own 256-byte record (every fetched line is consumed by the lane that
// fetched it).  The two blocks of R ‖ A ‖ M are built one after the other — the second half of the message is loaded after the
// first compression — so that the lane holds one block of message at a time: 96 VGPRs instead of 144 + spills for the
// whole padded stream, and no LDS (the round-1 version staged the records through 31 KB of LDS per 128 lanes, which held the
// kernel to 2.5 waves per SIMD).
constexpr int CH_THREADS = 128;

// wit (COMMIT units, include/bsx_layout.h): slot me = validator me % v_max of commit me / v_max — the lane also leaves the digest, the
// reduced challenge and the hint's validator record (pubkey, signature, message: the first 220 bytes of bsx_validator as they are)
__global__ __launch_bounds__(CH_THREADS) void k_sha512_challenge(const bsx_validator* __restrict__ vals, uint64_t n,
                                                                 uint8_t* __restrict__ out_h, uint8_t* __restrict__ out_digest,
                                                                 uint32_t v_max, bsxk_unit_dst wit) {
    const uint64_t me = (uint64_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (me >= n) return;
    const uint4* rec = reinterpret_cast<const uint4*>(vals + me);       // pubkey 0, R 32, s 64, message 96..219, message_len 220
    int len = (int)rec[13].w;
    if (len > BSX_VALIDATOR_MSG_MAX) len = BSX_VALIDATOR_MSG_MAX;
    const uint64_t bits = (uint64_t)(64 + len) * 8;
    const bool two = (64 + len) >= 112;                                 // 0x80 + 16-byte length no longer fit the first block
    uint64_t st[8], w[16];
    sha512_init(st);
    {
        const uint4 a0 = rec[0], a1 = rec[1], r0 = rec[2], r1 = rec[3];
        const uint32_t ra[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = ((uint64_t)bswap32(ra[2 * k]) << 32) | bswap32(ra[2 * k + 1]);
        const uint4 m0 = rec[6], m1 = rec[7], m2 = rec[8], m3 = rec[9];
        const uint32_t m[16] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w, m3.x, m3.y, m3.z, m3.w};
#pragma unroll
        for (int k = 0; k < 8; k++)
            w[8 + k] = ((uint64_t)bswap32(sha512_ram_m_dword(m[2 * k], 2 * k, len)) << 32) | bswap32(sha512_ram_m_dword(m[2 * k + 1], 2 * k + 1, len));
        if (!two) w[15] = bits;
        sha512_compress(st, w);
    }
    if (two) {
        const uint4 m4 = rec[10], m5 = rec[11], m6 = rec[12], m7 = rec[13];
        const uint32_t m[16] = {m4.x, m4.y, m4.z, m4.w, m5.x, m5.y, m5.z, m5.w, m6.x, m6.y, m6.z, m6.w, m7.x, m7.y, m7.z, 0u};   // dword 31 is message_len
#pragma unroll
        for (int k = 0; k < 8; k++)
            w[k] = ((uint64_t)bswap32(sha512_ram_m_dword(m[2 * k], 16 + 2 * k, len)) << 32) | bswap32(sha512_ram_m_dword(m[2 * k + 1], 17 + 2 * k, len));
#pragma unroll
        for (int k = 8; k < 15; k++) w[k] = 0;
        w[15] = bits;
        sha512_compress(st, w);
    }
    uint32_t dig[16], h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        dig[2 * k] = bswap32((uint32_t)(st[k] >> 32));
        dig[2 * k + 1] = bswap32((uint32_t)st[k]);
    }
    sc_reduce64(dig, h);
    uint4* oh = reinterpret_cast<uint4*>(out_h + me * 32);
    oh[0] = make_uint4(h[0], h[1], h[2], h[3]);
    oh[1] = make_uint4(h[4], h[5], h[6], h[7]);
    if (out_digest) {
        uint4* od = reinterpret_cast<uint4*>(out_digest + me * 64);
#pragma unroll
        for (int k = 0; k < 4; k++) od[k] = make_uint4(dig[4 * k], dig[4 * k + 1], dig[4 * k + 2], dig[4 * k + 3]);
    }
    if (wit.base) {
        const uint32_t c = (uint32_t)(me / v_max), v = (uint32_t)(me % v_max);
        uint8_t* cw = wit.base + (uint64_t)c * wit.stride;
        uint4* od = reinterpret_cast<uint4*>(cw + bsx_cm_off_digest(v_max) + 64 * v);
#pragma unroll
        for (int k = 0; k < 4; k++) od[k] = make_uint4(dig[4 * k], dig[4 * k + 1], dig[4 * k + 2], dig[4 * k + 3]);
        uint4* oc = reinterpret_cast<uint4*>(cw + bsx_cm_off_challenge(v_max) + 32 * v);
        oc[0] = make_uint4(h[0], h[1], h[2], h[3]);
        oc[1] = make_uint4(h[4], h[5], h[6], h[7]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(cw + bsx_cm_off_validators(v_max) + BSX_CM_VAL_BYTES * v);   // 4-byte aligned
#pragma unroll
        for (int k = 0; k < 13; k++) {
            const uint4 q = rec[k];
            dst[4 * k] = q.x; dst[4 * k + 1] = q.y; dst[4 * k + 2] = q.z; dst[4 * k + 3] = q.w;
        }
        const uint4 q = rec[13];
        dst[52] = q.x; dst[53] = q.y; dst[54] = q.z;                  // dword 55 of the record is message_len
    }
}

// ------------------------------------------------------------------------------------------------ k_ed25519_verify
constexpr int ED_THREADS = 64;   // one wave per workgroup: a 100-signature commit spreads over 2 CUs, R commits over 2R
constexpr uint8_t ED_DEFERRED = 2;   // ok_out marker: the fixed-key kernel left this slot to the generic one
constexpr uint8_t ED_PENDING = 3;    // ok_out marker: projective result parked in the scratch slot, k_ed25519_finish decides
// ONLY_DEFERRED: second pass behind k_ed25519_verify_keyed — touch only the slots it marked (normally none: the
// waves read one byte per lane and retire)
template <bool ONLY_DEFERRED>
__global__ __launch_bounds__(ED_THREADS) void k_ed25519_verify(const bsx_validator* __restrict__ vals,
                                                               const uint8_t* __restrict__ hs, uint64_t n,
                                                               uint8_t* __restrict__ ok_out) {
    // ONLY_DEFERRED: a SMALL grid strides over the markers (the launcher caps it): normally nothing is deferred, and a launch of
    // one wave per 64 slots that only reads a byte each still has to be dispatched through a GPU full of hashing waves (0.17 ms
    // per 25,600 slots beside k_header_merkle); 64 workgroups scan the same bytes in one round
    for (uint64_t me = (uint64_t)blockIdx.x * ED_THREADS + threadIdx.x; me < n; me += (uint64_t)gridDim.x * ED_THREADS) {
    if (ONLY_DEFERRED && ok_out[me] != ED_DEFERRED) continue;
    const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
    const uint4 flags = rec[14];                    // bytes 224..239: voting_power (8), enabled, is_signed, present, pad
    const bool active = ((flags.z & 0xffu) != 0) && (((flags.z >> 8) & 0xffu) != 0);
    bool ok = false;
    if (active) {
        uint32_t pk[8], sr[8], ss[8], h[8];
        const uint4 p0 = rec[0], p1 = rec[1], r0 = rec[2], r1 = rec[3], s0 = rec[4], s1 = rec[5];
        pk[0] = p0.x; pk[1] = p0.y; pk[2] = p0.z; pk[3] = p0.w; pk[4] = p1.x; pk[5] = p1.y; pk[6] = p1.z; pk[7] = p1.w;
        sr[0] = r0.x; sr[1] = r0.y; sr[2] = r0.z; sr[3] = r0.w; sr[4] = r1.x; sr[5] = r1.y; sr[6] = r1.z; sr[7] = r1.w;
        ss[0] = s0.x; ss[1] = s0.y; ss[2] = s0.z; ss[3] = s0.w; ss[4] = s1.x; ss[5] = s1.y; ss[6] = s1.z; ss[7] = s1.w;
        const uint4* hp = reinterpret_cast<const uint4*>(hs + me * 32);
        const uint4 h0 = hp[0], h1 = hp[1];
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
        ok = ed25519_verify_core(pk, sr, ss, h);
    }
    ok_out[me] = ok ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ fixed-key tables
// Key table in HBM (bsx_ed25519_keytable_bytes): [n_keys x 64 B key records: pubkey, decodes flag]
//                                                [64 B table word: dword 0 = some row is dirty (k_keytable_check)]
//                                                [n_keys x KT_PARTS x 40 i32 base points -2^(W k) A (X, Y, Z, T)]
//                                                [n_keys x KT_PARTS x 2^(W-1) x 32 i32 affine multiples, one cache line each]
// W = the table's digit width (ed25519.h: KT_W = 12 by default, KT_W_WIDE = 16 for resident validator sets): a property of the table,
// uniform over a launch, part of the rows' layout tag.
constexpr uint64_t KT_REC_BYTES = 64;
__host__ __device__ inline uint64_t kt_base_i32(int w) { return 40ull * (uint64_t)kt_parts(w); }
__host__ __device__ inline uint64_t kt_flag_off(uint64_t n_keys) { return n_keys * KT_REC_BYTES; }
__host__ __device__ inline uint64_t kt_bases_off(uint64_t n_keys) { return (n_keys + 1) * KT_REC_BYTES; }
// entries start on a cache line (the table itself must: hipMalloc / torch / arena allocations are 256-byte aligned)
__host__ __device__ inline uint64_t kt_entries_off(uint64_t n_keys, int w) { return (kt_bases_off(n_keys) + n_keys * kt_base_i32(w) * 4 + 127) & ~127ull; }
__host__ __device__ inline uint64_t kt_bytes(uint64_t n_keys, int w) { return kt_entries_off(n_keys, w) + n_keys * (uint64_t)kt_key_i32(w) * 4; }

__device__ __forceinline__ void load_pk(const bsx_validator* v, uint32_t pk[8]) {
    const uint4* rec = reinterpret_cast<const uint4*>(v);
    const uint4 p0 = rec[0], p1 = rec[1];
    pk[0] = p0.x; pk[1] = p0.y; pk[2] = p0.z; pk[3] = p0.w; pk[4] = p1.x; pk[5] = p1.y; pk[6] = p1.z; pk[7] = p1.w;
}

__host__ __device__ inline uint32_t kt_magic(int w) { return 0x4b540000u | (uint32_t)w; }     // layout tag: the digit width is part of it
// decode, negate, and run the (KT_PARTS - 1) x KT_W doublings that give the base points of the upper digit positions; row k of `table`
__device__ __forceinline__ void keytable_build_bases(const uint32_t pk[8], uint32_t k, uint32_t n_keys, uint8_t* __restrict__ table, int w) {
    ge_p3 b;
    const bool ok = ge_frombytes_negate(b, pk);
    uint4* rec = reinterpret_cast<uint4*>(table + k * KT_REC_BYTES);
    rec[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    rec[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    rec[2] = make_uint4(ok ? 1u : 0u, 0u, 0u, 0u);
    rec[3] = make_uint4(kt_magic(w), n_keys, 1u, 0u);         // stays dirty for k_keytable_entries; the next check re-evaluates
    int32_t* dst = reinterpret_cast<int32_t*>(table + kt_bases_off(n_keys)) + (uint64_t)k * kt_base_i32(w);
    const int parts = kt_parts(w);
#pragma unroll 1
    for (int half = 0; half < parts; half++) {             // one base point live at a time (40 VGPRs), stored as it is produced
        if (half) b = ge_keytable_next_base(b, w);
#pragma unroll
        for (int i = 0; i < 10; i++) {
            dst[half * 40 + i] = b.X.v[i];
            dst[half * 40 + 10 + i] = b.Y.v[i];
            dst[half * 40 + 20 + i] = b.Z.v[i];
            dst[half * 40 + 30 + i] = b.T.v[i];
        }
    }
}
// Reuse across calls: a key record ends with {KT_MAGIC, n_keys, dirty, 0}.  k_keytable_check marks row k dirty when the
// table does not hold THIS key for THIS n_keys (fresh zeroed table, validator-set change, different layout); the two
// build kernels skip clean rows.  The reference's validator set is fixed per proof and changes on the chain's
// unbonding time scale (header_range.rs:42-48 takes it from the trusted/target headers), so in steady state a step pays
// one 100-lane compare instead of 192 serial point doublings per key — and a changed key costs exactly its own rebuild.
// ONE workgroup: it also leaves "some row is dirty" in the table word, which lets every wave of the two build launches
// leave after one (shared, cached) load in the steady state — their grids are sized for a full rebuild (4,400 waves at
// V = 100), and a per-row flag load per wave cost 0.17 ms beside an expansion.
constexpr int KC_THREADS = 256;
__global__ __launch_bounds__(KC_THREADS) void k_keytable_check(const bsx_validator* __restrict__ vals, uint32_t n_keys,
                                                               uint8_t* __restrict__ table, uint32_t force, int w) {
    int any = 0;
    for (uint32_t k = threadIdx.x; k < n_keys; k += KC_THREADS) {
        uint32_t pk[8];
        load_pk(vals + k, pk);
        uint4* rec = reinterpret_cast<uint4*>(table + k * KT_REC_BYTES);
        const uint4 k0 = rec[0], k1 = rec[1], tag = rec[3];
        const bool same = tag.x == kt_magic(w) && tag.y == n_keys && k0.x == pk[0] && k0.y == pk[1] && k0.z == pk[2] && k0.w == pk[3] &&
                          k1.x == pk[4] && k1.y == pk[5] && k1.z == pk[6] && k1.w == pk[7];
        const uint32_t dirty = (force || !same) ? 1u : 0u;
        reinterpret_cast<uint32_t*>(rec + 3)[2] = dirty;
        any |= (int)dirty;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(table + kt_flag_off(n_keys)) = any ? 1u : 0u;
    if (!any) return;
    // rebuild: the base points of the dirty rows, here (one lane per key; the rows of a validator set fit one or two passes
    // of this workgroup) — a launch of its own cost a proof request 10 us of host time for nothing in the steady state
    for (uint32_t k = threadIdx.x; k < n_keys; k += KC_THREADS) {
        if (reinterpret_cast<const uint32_t*>(table + k * KT_REC_BYTES)[14] == 0) continue;
        uint32_t pk[8];
        load_pk(vals + k, pk);
        keytable_build_bases(pk, k, n_keys, table, w);
    }
}

// The context's table of B (bt_bytes()): [BT_PARTS x 40 i32 base points 2^(W k) B][pad][BT_PARTS x 2^(W-1) x 32 i32 entries],
// built by the key-table arithmetic from the encoding of -B ("-A" = B).
__host__ __device__ inline uint64_t bt_entries_off() { return ((uint64_t)BT_PARTS * 160 + 127) & ~127ull; }
__host__ __device__ inline uint64_t bt_bytes() { return bt_entries_off() + (uint64_t)BT_I32 * 4; }
__global__ void k_btable_bases(uint8_t* __restrict__ table) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t pk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) pk[i] = GE_NEG_B_ENC[i];
    ge_p3 b;
    (void)ge_frombytes_negate(b, pk);
    int32_t* dst = reinterpret_cast<int32_t*>(table);
#pragma unroll 1
    for (int part = 0; part < BT_PARTS; part++) {
        if (part) b = ge_keytable_next_base(b, BT_W);
#pragma unroll
        for (int i = 0; i < 10; i++) {
            dst[part * 40 + i] = b.X.v[i];
            dst[part * 40 + 10 + i] = b.Y.v[i];
            dst[part * 40 + 20 + i] = b.Z.v[i];
            dst[part * 40 + 30 + i] = b.T.v[i];
        }
    }
}
// Table entries, for key tables and the table of B alike.  Row = one (point, part) pair with its base point 2^(W part) P
// in `bases`; its 2^(W-1) entries j * base are produced KB_G at a time per lane: the group's first multiple by a uniform
// double-and-add, the rest by repeated addition, and ONE field inversion per group turns them affine (Montgomery's trick:
// prefix products of the Z's — kept in LDS, one column per lane — then 1/Z_j = inv * prefix_(j-1), inv *= Z_j walking back).
// ~43 multiplications per entry instead of the ~420 of a double-and-add plus an inversion of its own per entry: that
// is what makes 12- and 16-bit digit tables (45,056 / 524,288 entries per point) cheap to build.
constexpr int KB_G = 16;
struct TableBuildArgs {
    const int32_t* bases;       // [n_rows][40]
    int32_t* entries;           // [n_rows][half][32]
    const uint8_t* recs;        // key records (dirty flag at dword 14), or null: build every row
    const uint32_t* any_dirty;  // with recs: the table word, 0 = no row is dirty
    uint32_t n_rows, parts, half, bits;
};
__global__ __launch_bounds__(ED_THREADS) void k_table_entries(TableBuildArgs a) {
    __shared__ int32_t pre[KB_G * 10 * ED_THREADS];
    const uint32_t tid = threadIdx.x;
    if (a.any_dirty && *a.any_dirty == 0) return;
    const uint32_t groups = a.half / KB_G;
    const uint64_t total = (uint64_t)a.n_rows * groups;
    // grid-strided: the launcher caps the grid at what can be resident (LDS: 4 workgroups per CU), so that the steady-state
    // launch — every workgroup leaves after the one load above — is 1,024 workgroups, not one per 64 x 16 entries
    for (uint64_t gl = (uint64_t)blockIdx.x * ED_THREADS + tid; gl < total; gl += (uint64_t)gridDim.x * ED_THREADS) {
    const uint64_t row = gl / groups;
    const uint32_t g = (uint32_t)(gl % groups);
    if (a.recs && reinterpret_cast<const uint32_t*>(a.recs + (row / a.parts) * KT_REC_BYTES)[14] == 0) continue;   // clean row
    const int32_t* src = a.bases + row * 40;
    ge_p3 base;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        base.X.v[i] = src[i];
        base.Y.v[i] = src[10 + i];
        base.Z.v[i] = src[20 + i];
        base.T.v[i] = src[30 + i];
    }
    int32_t* out = a.entries + (row * a.half + (uint64_t)g * KB_G) * KT_ENTRY_I32;
    ge_p3 acc = ge_mul_small(base, (int)(g * KB_G + 1), (int)a.bits);
    const ge_cached cb = p3_to_cached(base);
    fe prod = fe_one();
#pragma unroll 1
    for (int j = 0; j < KB_G; j++) {
        int32_t* d = out + j * KT_ENTRY_I32;
#pragma unroll
        for (int i = 0; i < 10; i++) { d[i] = acc.X.v[i]; d[10 + i] = acc.Y.v[i]; d[20 + i] = acc.Z.v[i]; }
        prod = fe_mul(prod, acc.Z);
#pragma unroll
        for (int i = 0; i < 10; i++) pre[(j * 10 + i) * ED_THREADS + tid] = prod.v[i];
        if (j + 1 < KB_G) acc = p1p1_to_p3(ge_add(acc, cb));
    }
    fe inv = fe_invert(prod);
#pragma unroll 1
    for (int j = KB_G - 1; j >= 0; j--) {
        int32_t* d = out + j * KT_ENTRY_I32;
        fe X, Y, Z, zi = inv;
#pragma unroll
        for (int i = 0; i < 10; i++) { X.v[i] = d[i]; Y.v[i] = d[10 + i]; Z.v[i] = d[20 + i]; }
        if (j > 0) {
            fe p;
#pragma unroll
            for (int i = 0; i < 10; i++) p.v[i] = pre[((j - 1) * 10 + i) * ED_THREADS + tid];
            zi = fe_mul(inv, p);
        }
        inv = fe_mul(inv, Z);
        precomp_store(d, ge_to_precomp(X, Y, zi));
    }
    }
}

// one lane per validator slot; slot (me % v_max) uses key table row (me % v_max) when the record's public key is the
// table's key, and falls back to the generic per-signature path otherwise (a validator-set change inside the batch)
// scratch slot of the deferred-encode form: X, Y, Z (10 limbs each), prefix product (10), = 160 bytes per signature
constexpr uint32_t ED_SLOT_I32 = 40;
// signatures per lane of k_ed25519_finish (one inversion amortised over K), measured with the 38-addition signature kernel
// (M verifies/s, K = 2 / 4 / 8 / 16 / 32): 204,800 signatures 269 / 296 / 325 / 310 / 280, 1,048,576: 406 / 462 / 470 / 476 / 492
__host__ __device__ inline uint32_t ed_fin_k(uint64_t n) { return n >= 800000 ? 32u : n >= 400000 ? 16u : 8u; }
// below this many signatures a verification is spread over 4 lanes (k_ed25519_verify_keyed SPLIT)
constexpr uint64_t ED_SPLIT_BELOW = 300000;
// Lane order.  Signature me = commit * v_max + slot.  With many commits the lanes of a wave take 64 COMMITS of ONE slot
// (BY_KEY): all of them walk the same key's table, part by part, so a wave's 64 lookups of a step fall into one 16 KB part
// (lines shared between lanes, L2-resident) instead of 64 different 512 KB tables; the workgroups of a key are
// consecutive, and the block index is remapped so that each XCD (workgroups are dealt to the 8 XCDs round-robin) owns
// a contiguous run of keys.  With few commits (a single proof: 1 x 100 signatures) lanes take consecutive signatures.
// SPLIT lanes per signature (1 or 4): each sums its share of the 48 table entries, the shares are joined by a butterfly of
// full additions through wave shuffles (ed25519.h ed25519_keyed_partial).  SPLIT = 4 is the small-batch form.
// (block, n_blocks): the workgroup's index and the size of the grid it belongs to — the launch's own, or a part of it
// (k_ed25519_verify_keyed_mixed)
template <bool DEFER, bool BY_KEY, int SPLIT>
__device__ __forceinline__ void verify_keyed_body(uint32_t block, uint32_t n_blocks, const bsx_validator* __restrict__ vals,
                                                  const uint8_t* __restrict__ hs, uint64_t n, uint32_t v_max,
                                                  const uint8_t* __restrict__ table, uint32_t n_keys, const int32_t* __restrict__ b_tab,
                                                  uint8_t* __restrict__ ok_out, int32_t* __restrict__ scratch, const uint32_t* __restrict__ rows, int w) {
    constexpr uint32_t SIGS = ED_THREADS / SPLIT;                              // signatures per workgroup
    const uint32_t sub = threadIdx.x / SPLIT, part0 = threadIdx.x % SPLIT;
    uint64_t me;
    if (BY_KEY) {
        const uint32_t per_xcd = n_blocks >> 3;                                // the launcher pads the grid to a multiple of 8
        const uint32_t blk = (block & 7) * per_xcd + (block >> 3);             // logical block: XCD x owns [x, x + 1) * per_xcd
        const uint64_t n_commits = (n + v_max - 1) / v_max;
        const uint32_t wpk = (uint32_t)((n_commits + SIGS - 1) / SIGS);        // workgroups per key
        const uint32_t slot_ = blk / wpk;
        const uint64_t commit = (uint64_t)(blk % wpk) * SIGS + sub;
        if (slot_ >= v_max || commit >= n_commits) return;
        me = commit * v_max + slot_;
    } else {
        me = (uint64_t)block * SIGS + sub;
    }
    if (me >= n) return;                        // whole groups of SPLIT lanes leave together (every test below is per signature)
    const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
    const uint4 flags = rec[14];
    const bool active = ((flags.z & 0xffu) != 0) && (((flags.z >> 8) & 0xffu) != 0);
    bool ok = false;
    if (active) {
        uint32_t pk[8], sr[8], ss[8], h[8];
        load_pk(vals + me, pk);
        const uint4 r0 = rec[2], r1 = rec[3], s0 = rec[4], s1 = rec[5];
        sr[0] = r0.x; sr[1] = r0.y; sr[2] = r0.z; sr[3] = r0.w; sr[4] = r1.x; sr[5] = r1.y; sr[6] = r1.z; sr[7] = r1.w;
        ss[0] = s0.x; ss[1] = s0.y; ss[2] = s0.z; ss[3] = s0.w; ss[4] = s1.x; ss[5] = s1.y; ss[6] = s1.z; ss[7] = s1.w;
        const uint4* hp = reinterpret_cast<const uint4*>(hs + me * 32);
        const uint4 h0 = hp[0], h1 = hp[1];
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;

        const uint32_t slot = rows ? rows[me] : (uint32_t)(me % v_max);    // rows: the caller's key -> table row map
        bool keyed = slot < n_keys;
        bool decodes = false;
        if (keyed) {
            const uint4* kr = reinterpret_cast<const uint4*>(table + (uint64_t)slot * KT_REC_BYTES);
            const uint4 k0 = kr[0], k1 = kr[1];
            keyed = k0.x == pk[0] && k0.y == pk[1] && k0.z == pk[2] && k0.w == pk[3] && k1.x == pk[4] && k1.y == pk[5] &&
                    k1.z == pk[6] && k1.w == pk[7] && kr[3].x == kt_magic(w);      // the row holds THIS key in THIS digit width
            decodes = kr[2].x != 0;
        }
        if (!keyed) {
            if (part0 == 0) ok_out[me] = ED_DEFERRED;   // left to k_ed25519_verify<true>, launched right behind on the same stream
            return;
        }
        const int32_t* kt = reinterpret_cast<const int32_t*>(table + kt_entries_off(n_keys, w)) + (uint64_t)slot * kt_key_i32(w);
        if (SPLIT == 1) {
            if (DEFER) {
                ge_p2 q;
                const bool pre = decodes && ed25519_verify_keyed_core_t<true>(kt, b_tab, sr, ss, h, &q, w);
                int32_t* d = scratch + me * ED_SLOT_I32;
#pragma unroll
                for (int i = 0; i < 10; i++) { d[i] = q.X.v[i]; d[10 + i] = q.Y.v[i]; d[20 + i] = q.Z.v[i]; }
                ok_out[me] = pre ? ED_PENDING : 0;          // k_ed25519_finish turns ED_PENDING into the verdict
                return;
            }
            ok = decodes && ed25519_verify_keyed_core(kt, b_tab, sr, ss, h, w);
        } else {
            ge_p3 p = ed25519_keyed_partial<SPLIT>(kt, b_tab, ss, h, (int)part0, w);
            // butterfly over the SPLIT lanes of the signature (adjacent lanes of one wave): afterwards every lane holds the sum
#pragma unroll
            for (int m = 1; m < SPLIT; m <<= 1) {
                ge_p3 o;
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    o.X.v[i] = __shfl_xor(p.X.v[i], m, 64);
                    o.Y.v[i] = __shfl_xor(p.Y.v[i], m, 64);
                    o.Z.v[i] = __shfl_xor(p.Z.v[i], m, 64);
                    o.T.v[i] = __shfl_xor(p.T.v[i], m, 64);
                }
                p = p1p1_to_p3(ge_add(p, p3_to_cached(o)));
            }
            if (part0 != 0) return;
            const bool pre = decodes && sc_is_canonical(ss);
            if (DEFER) {
                int32_t* d = scratch + me * ED_SLOT_I32;
#pragma unroll
                for (int i = 0; i < 10; i++) { d[i] = p.X.v[i]; d[10 + i] = p.Y.v[i]; d[20 + i] = p.Z.v[i]; }
                ok_out[me] = pre ? ED_PENDING : 0;
                return;
            }
            uint32_t enc[8];
            ge_tobytes(enc, p.X, p.Y, p.Z);
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) diff |= enc[k] ^ sr[k];
            ok = pre && diff == 0;
        }
    }
    if (part0 == 0) ok_out[me] = ok ? 1 : 0;
}
template <bool DEFER, bool BY_KEY, int SPLIT>
__global__ __launch_bounds__(ED_THREADS, 2) void k_ed25519_verify_keyed(const bsx_validator* __restrict__ vals,
                                                                     const uint8_t* __restrict__ hs, uint64_t n,
                                                                     uint32_t v_max, const uint8_t* __restrict__ table,
                                                                     uint32_t n_keys, const int32_t* __restrict__ b_tab,
                                                                     uint8_t* __restrict__ ok_out,
                                                                     int32_t* __restrict__ scratch, const uint32_t* __restrict__ rows, int w) {
    verify_keyed_body<DEFER, BY_KEY, SPLIT>(blockIdx.x, gridDim.x, vals, hs, n, v_max, table, n_keys, b_tab, ok_out, scratch, rows, w);
}
// A batch whose one-lane-per-signature waves come to a little MORE than a whole number per SIMD (2048 commits x 100 slots: 3200
// waves on 1024 SIMDs) takes as long as the SIMDs with the extra wave: 4 chains where the average is 3.125.  Here the first
// blocks_a workgroups take commits [0, commits_a) one lane per signature — a whole number of waves per SIMD — and the others the
// remaining commits on FOUR lanes each (quarter-length chains, 1.5x the work, spread over four times as many SIMDs).  Four waves per
// SIMD (128 registers, 16 dwords spilled) so that both kinds are resident from the start.  Same scratch records: one k_ed25519_finish.
__global__ __launch_bounds__(ED_THREADS, 4) void k_ed25519_verify_keyed_mixed(const bsx_validator* __restrict__ vals,
                                                                           const uint8_t* __restrict__ hs, uint64_t n, uint32_t v_max,
                                                                           const uint8_t* __restrict__ table, uint32_t n_keys,
                                                                           const int32_t* __restrict__ b_tab, uint8_t* __restrict__ ok_out,
                                                                           int32_t* __restrict__ scratch, uint32_t blocks_a, uint64_t commits_a,
                                                                           const uint32_t* __restrict__ rows, int w) {
    const uint64_t na = commits_a * v_max;
    if (blockIdx.x < blocks_a)
        verify_keyed_body<true, true, 1>(blockIdx.x, blocks_a, vals, hs, na, v_max, table, n_keys, b_tab, ok_out, scratch, rows, w);
    else
        verify_keyed_body<true, true, 4>(blockIdx.x - blocks_a, gridDim.x - blocks_a, vals + na, hs + na * 32, n - na, v_max, table, n_keys, b_tab,
                                         ok_out + na, scratch + na * ED_SLOT_I32, rows ? rows + na : nullptr, w);
}

// The small-batch form (a single proof: 100 signatures): latency is everything, and a third of a signature's dependent
// chain is the field inversion that encodes the result for the byte comparison with R.  Here a SECOND wave of the workgroup
// decodes R instead (RFC 8032 strict: canonical y, on the curve, x = 0 only with sign 0 — a square root chain of the same
// length as the inversion, but independent of the point arithmetic, so the two waves run side by side), and the first
// wave compares projectively: (X : Y : Z) == (x_R, y_R)  <=>  X = x_R Z and Y = y_R Z.  Same accept set: the encoding of a
// point is canonical, so "encode(P) == R bytes" holds exactly when R decodes strictly and decodes to P.
// 16 signatures per workgroup: wave 0 = 16 x 4 lanes of partial sums, wave 1 = 16 lanes of decoding.
constexpr int EL_SIGS = 16;
__global__ __launch_bounds__(128) void k_ed25519_verify_keyed_small(const bsx_validator* __restrict__ vals, const uint8_t* __restrict__ hs,
                                                                  uint64_t n, uint32_t v_max, const uint8_t* __restrict__ table,
                                                                  uint32_t n_keys, const int32_t* __restrict__ b_tab,
                                                                  uint8_t* __restrict__ ok_out, const uint32_t* __restrict__ rows, int w) {
    __shared__ int32_t rdec[EL_SIGS][21];                 // -x_R (10 limbs), y_R (10), decodes
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t sub = wave == 0 ? lane / 4 : lane, part0 = lane % 4;
    const uint64_t me = (uint64_t)blockIdx.x * EL_SIGS + sub;
    const bool in_range = sub < EL_SIGS && me < n;
    uint32_t sr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ss[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool active = false;
    if (in_range) {
        const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
        const uint4 flags = rec[14];
        active = ((flags.z & 0xffu) != 0) && (((flags.z >> 8) & 0xffu) != 0);
        const uint4 r0 = rec[2], r1 = rec[3], s0 = rec[4], s1 = rec[5];
        sr[0] = r0.x; sr[1] = r0.y; sr[2] = r0.z; sr[3] = r0.w; sr[4] = r1.x; sr[5] = r1.y; sr[6] = r1.z; sr[7] = r1.w;
        ss[0] = s0.x; ss[1] = s0.y; ss[2] = s0.z; ss[3] = s0.w; ss[4] = s1.x; ss[5] = s1.y; ss[6] = s1.z; ss[7] = s1.w;
    }
    bool keyed = false, decodes = false;
    ge_p3 p{fe_zero(), fe_one(), fe_one(), fe_zero()};
    if (wave == 1) {
        if (in_range && active) {
            ge_p3 nr;
            const bool rok = ge_frombytes_negate(nr, sr);
#pragma unroll
            for (int i = 0; i < 10; i++) { rdec[sub][i] = nr.X.v[i]; rdec[sub][10 + i] = nr.Y.v[i]; }
            rdec[sub][20] = rok ? 1 : 0;
        }
    } else if (in_range && active) {
        uint32_t pk[8], h[8];
        load_pk(vals + me, pk);
        const uint4* hp = reinterpret_cast<const uint4*>(hs + me * 32);
        const uint4 h0 = hp[0], h1 = hp[1];
        h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
        const uint32_t slot = rows ? rows[me] : (uint32_t)(me % v_max);    // rows: the caller's key -> table row map
        keyed = slot < n_keys;
        if (keyed) {
            const uint4* kr = reinterpret_cast<const uint4*>(table + (uint64_t)slot * KT_REC_BYTES);
            const uint4 k0 = kr[0], k1 = kr[1];
            keyed = k0.x == pk[0] && k0.y == pk[1] && k0.z == pk[2] && k0.w == pk[3] && k1.x == pk[4] && k1.y == pk[5] &&
                    k1.z == pk[6] && k1.w == pk[7] && kr[3].x == kt_magic(w);
            decodes = kr[2].x != 0;
        }
        if (keyed) {                                        // the 4 lanes of a signature agree on every test above
            const int32_t* kt = reinterpret_cast<const int32_t*>(table + kt_entries_off(n_keys, w)) + (uint64_t)slot * kt_key_i32(w);
            p = ed25519_keyed_partial<4>(kt, b_tab, ss, h, (int)part0, w);
#pragma unroll
            for (int m = 1; m < 4; m <<= 1) {
                ge_p3 o;
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    o.X.v[i] = __shfl_xor(p.X.v[i], m, 64);
                    o.Y.v[i] = __shfl_xor(p.Y.v[i], m, 64);
                    o.Z.v[i] = __shfl_xor(p.Z.v[i], m, 64);
                    o.T.v[i] = __shfl_xor(p.T.v[i], m, 64);
                }
                p = p1p1_to_p3(ge_add(p, p3_to_cached(o)));
            }
        }
    }
    __syncthreads();
    if (wave != 0 || part0 != 0 || !in_range) return;
    if (!active) { ok_out[me] = 0; return; }
    if (!keyed) { ok_out[me] = ED_DEFERRED; return; }        // left to k_ed25519_verify<true>, launched right behind
    fe nx, ry;
#pragma unroll
    for (int i = 0; i < 10; i++) { nx.v[i] = rdec[sub][i]; ry.v[i] = rdec[sub][10 + i]; }
    const bool rok = rdec[sub][20] != 0;
    const bool same_x = !fe_isnonzero(fe_add(p.X, fe_mul(nx, p.Z)));      // X - x_R Z, with nx = -x_R
    const bool same_y = !fe_isnonzero(fe_sub(p.Y, fe_mul(ry, p.Z)));
    ok_out[me] = (decodes && sc_is_canonical(ss) && rok && same_x && same_y) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ the latency form (mode F)
// One proof verifies ONE commit (<= 100 signatures), a pipelined chunk a few thousand: every signature lane sits on its own
// SIMD and the launch takes as long as ONE signature's dependent chain.  In the forms above that chain is 12-48 additions
// followed by a field inversion (254 squarings) for the encoding.  Here
//   * R is DECODED ahead of time (k_ed25519_decode_r: strict RFC 8032 decoding, a square-root chain as long as the inversion
//     but independent of h and of the point arithmetic) — the pipeline runs it with the challenges, beside the header hashing,
//     off the commit check's critical path;
//   * the 38 table entries of a signature are summed by SPLIT = 8 or 16 lanes (2-5 mixed additions each) and joined by a
//     log2(SPLIT)-level butterfly of full additions through wave shuffles;
//   * the comparison is projective: (X : Y : Z) == (x_R, y_R)  <=>  X = x_R Z and Y = y_R Z — no inversion anywhere.
// Same accept set: encodings are canonical, so "encode(P) == R bytes" holds exactly when R decodes strictly and decodes to P.
// Dependent chain per signature: 3 + 4 (SPLIT 16) or 5 + 3 (SPLIT 8) additions and two multiplications, instead of
// 14 additions + an inversion: 0.35 ms -> ~0.05 ms per pipelined chunk's 12,800 signatures.
constexpr uint32_t ED_RDEC_I32 = 24;     // per signature: -x_R (10 limbs), y_R (10), decodes flag, 3 pad = 96 bytes
__global__ __launch_bounds__(ED_THREADS) void k_ed25519_decode_r(const bsx_validator* __restrict__ vals, uint64_t n, int32_t* __restrict__ rdec) {
    const uint64_t me = (uint64_t)blockIdx.x * ED_THREADS + threadIdx.x;
    if (me >= n) return;
    const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
    const uint4 flags = rec[14];
    const bool active = ((flags.z & 0xffu) != 0) && (((flags.z >> 8) & 0xffu) != 0);
    int32_t* d = rdec + me * ED_RDEC_I32;
    if (!active) { d[20] = 0; return; }
    const uint4 r0 = rec[2], r1 = rec[3];
    const uint32_t sr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    ge_p3 nr;
    const bool rok = ge_frombytes_negate(nr, sr);
#pragma unroll
    for (int i = 0; i < 10; i++) { d[i] = nr.X.v[i]; d[10 + i] = nr.Y.v[i]; }
    d[20] = rok ? 1 : 0;
}

template <int SPLIT>
__global__ __launch_bounds__(ED_THREADS, 2) void k_ed25519_verify_keyed_proj(const bsx_validator* __restrict__ vals, const uint8_t* __restrict__ hs,
                                                                          uint64_t n, uint32_t v_max, const uint8_t* __restrict__ table,
                                                                          uint32_t n_keys, const int32_t* __restrict__ b_tab,
                                                                          const int32_t* __restrict__ rdec, uint8_t* __restrict__ ok_out,
                                                                          const uint32_t* __restrict__ rows, int w) {
    constexpr uint32_t SIGS = ED_THREADS / SPLIT;
    const uint32_t sub = threadIdx.x / SPLIT, part0 = threadIdx.x % SPLIT;
    const uint64_t me = (uint64_t)blockIdx.x * SIGS + sub;
    if (me >= n) return;                        // whole groups of SPLIT lanes leave together (every test below is per signature)
    const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
    const uint4 flags = rec[14];
    const bool active = ((flags.z & 0xffu) != 0) && (((flags.z >> 8) & 0xffu) != 0);
    if (!active) {
        if (part0 == 0) ok_out[me] = 0;
        return;
    }
    uint32_t pk[8], ss[8], h[8];
    load_pk(vals + me, pk);
    const uint4 s0 = rec[4], s1 = rec[5];
    ss[0] = s0.x; ss[1] = s0.y; ss[2] = s0.z; ss[3] = s0.w; ss[4] = s1.x; ss[5] = s1.y; ss[6] = s1.z; ss[7] = s1.w;
    const uint4* hp = reinterpret_cast<const uint4*>(hs + me * 32);
    const uint4 h0 = hp[0], h1 = hp[1];
    h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[6] = h1.z; h[7] = h1.w;
    const uint32_t slot = rows ? rows[me] : (uint32_t)(me % v_max);    // rows: the caller's key -> table row map
    bool keyed = slot < n_keys;
    bool decodes = false;
    if (keyed) {
        const uint4* kr = reinterpret_cast<const uint4*>(table + (uint64_t)slot * KT_REC_BYTES);
        const uint4 k0 = kr[0], k1 = kr[1];
        keyed = k0.x == pk[0] && k0.y == pk[1] && k0.z == pk[2] && k0.w == pk[3] && k1.x == pk[4] && k1.y == pk[5] &&
                k1.z == pk[6] && k1.w == pk[7] && kr[3].x == kt_magic(w);
        decodes = kr[2].x != 0;
    }
    if (!keyed) {
        if (part0 == 0) ok_out[me] = ED_DEFERRED;   // left to k_ed25519_verify<true>, launched right behind on the same stream
        return;
    }
    const int32_t* kt = reinterpret_cast<const int32_t*>(table + kt_entries_off(n_keys, w)) + (uint64_t)slot * kt_key_i32(w);
    ge_p3 p = ed25519_keyed_partial<SPLIT>(kt, b_tab, ss, h, (int)part0, w);
#pragma unroll
    for (int m = 1; m < SPLIT; m <<= 1) {
        ge_p3 o;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_xor(p.X.v[i], m, 64);
            o.Y.v[i] = __shfl_xor(p.Y.v[i], m, 64);
            o.Z.v[i] = __shfl_xor(p.Z.v[i], m, 64);
            o.T.v[i] = __shfl_xor(p.T.v[i], m, 64);
        }
        p = p1p1_to_p3(ge_add(p, p3_to_cached(o)));
    }
    if (part0 != 0) return;
    const int32_t* d = rdec + me * ED_RDEC_I32;
    fe nx, ry;
#pragma unroll
    for (int i = 0; i < 10; i++) { nx.v[i] = d[i]; ry.v[i] = d[10 + i]; }
    const bool rok = d[20] != 0;
    const bool same_x = !fe_isnonzero(fe_add(p.X, fe_mul(nx, p.Z)));      // X - x_R Z, with nx = -x_R
    const bool same_y = !fe_isnonzero(fe_sub(p.Y, fe_mul(ry, p.Z)));
    ok_out[me] = (decodes && sc_is_canonical(ss) && rok && same_x && same_y) ? 1 : 0;
}

// Encoding + comparison of the deferred results: lane j owns signatures [j*K, (j+1)*K).  Montgomery's trick: prefix
// products of the Z coordinates (stored in the slots), one inversion of the total, then backwards 1/Z_i = inv * prefix_{i-1},
// inv *= Z_i.  Z of a point produced by the complete twisted-Edwards formulas from points on the curve is never zero.
__global__ __launch_bounds__(ED_THREADS) void k_ed25519_finish(const bsx_validator* __restrict__ vals, uint64_t n,
                                                               uint8_t* __restrict__ ok_out, int32_t* __restrict__ scratch, uint32_t K) {
    const uint64_t lane = (uint64_t)blockIdx.x * ED_THREADS + threadIdx.x;
    const uint64_t first = lane * K;
    if (first >= n) return;
    const uint32_t cnt = (uint32_t)((n - first < K) ? (n - first) : K);
    fe acc = fe_one();
    for (uint32_t i = 0; i < cnt; i++) {
        int32_t* d = scratch + (first + i) * ED_SLOT_I32;
        if (ok_out[first + i] == ED_PENDING) {
            fe z;
#pragma unroll
            for (int k = 0; k < 10; k++) z.v[k] = d[20 + k];
            acc = fe_mul(acc, z);
        }
#pragma unroll
        for (int k = 0; k < 10; k++) d[30 + k] = acc.v[k];          // prefix product INCLUDING element i
    }
    fe inv = fe_invert(acc);
    for (uint32_t ii = cnt; ii-- > 0;) {
        const uint64_t me = first + ii;
        if (ok_out[me] != ED_PENDING) continue;
        const int32_t* d = scratch + me * ED_SLOT_I32;
        fe x, y, z, prev = fe_one();
#pragma unroll
        for (int k = 0; k < 10; k++) { x.v[k] = d[k]; y.v[k] = d[10 + k]; z.v[k] = d[20 + k]; }
        if (ii > 0) {
            const int32_t* dp = scratch + (me - 1) * ED_SLOT_I32;
#pragma unroll
            for (int k = 0; k < 10; k++) prev.v[k] = dp[30 + k];
        }
        const fe zi = fe_mul(inv, prev);            // 1 / Z_i
        inv = fe_mul(inv, z);                       // 1 / (Z_0 .. Z_{i-1})
        const fe ax = fe_mul(x, zi), ay = fe_mul(y, zi);
        uint32_t enc[8];
        fe_tobytes(enc, ay);
        enc[7] ^= (uint32_t)fe_isnegative(ax) << 31;
        const uint4* rec = reinterpret_cast<const uint4*>(vals + me);
        const uint4 r0 = rec[2], r1 = rec[3];
        const uint32_t diff = (enc[0] ^ r0.x) | (enc[1] ^ r0.y) | (enc[2] ^ r0.z) | (enc[3] ^ r0.w) | (enc[4] ^ r1.x) | (enc[5] ^ r1.y) |
                              (enc[6] ^ r1.z) | (enc[7] ^ r1.w);
        ok_out[me] = diff == 0 ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ k_commit_tally
constexpr int TL_THREADS = 256;
constexpr int TL_VMAX = 512;     // padded power of two of the validator slots one workgroup folds

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// SimpleValidator leaf: 0a 22 0a 20 pk32 [10 varint(power)] as LE dwords; returns the byte length
__device__ __forceinline__ int validator_leaf(const uint32_t pk[8], uint64_t power, uint32_t d[14]) {
    d[0] = 0x200a220au;
#pragma unroll
    for (int k = 0; k < 8; k++) d[1 + k] = pk[k];
    // field 2 (voting_power) is omitted when zero (proto3).  Static byte positions, no private-memory indexing:
    // s[0] = 0x10, s[1+k] = 7-bit group k with the continuation bit while higher groups are non-zero
    uint32_t s[12];
    int nvar = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const uint64_t rest = power >> (7 * k);
        const bool present = (k == 0) ? (power != 0) : (rest != 0);
        const bool more = (k < 9) && ((power >> (7 * (k + 1))) != 0);
        s[1 + k] = present ? (uint32_t)((rest & 0x7f) | (more ? 0x80 : 0)) : 0u;
        nvar += present ? 1 : 0;
    }
    s[0] = power ? 0x10u : 0u;
    s[11] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) d[9 + k] = s[4 * k] | (s[4 * k + 1] << 8) | (s[4 * k + 2] << 16) | (s[4 * k + 3] << 24);
    d[12] = 0; d[13] = 0;
    return 36 + (power ? 1 + nvar : 0);
}

// wit: where the validator set's variables go (include/bsx_layout.h) — mode 0: the COMMIT unit of commit c (every slot's words and
// bools, leaves, the masked tree, sums, verdicts); mode 1: the trusted set inside the SKIP unit of range c (pubkeys, leaves, tree,
// enabled bools, total power; signed_target / overlap are k_skip_check's).  Every byte of the groups named here is written on
// every launch, so a resident unit never needs clearing.
// G commits per workgroup (round 5).  One commit per workgroup left its second wave after the leaves and three quarters of the first
// idle through the narrow levels: 16 wave-compressions per commit of P = 128 leaves where the hashing needs 4.7.  With G = 4 commits
// side by side (256 threads) every level's G * width nodes are dealt densely to the lanes — 256 / 128 / 64 / 32 / 16 / 8 / 4 nodes on
// 4 / 2 / 1 / 1 ... waves: 7.5 wave-compressions per commit, the same 15-compression dependent chain.  n_commits: commits beyond it
// (the last workgroup's tail) are skipped.  G = 1 is the form for P > 128 and for single commits.
template <int G>
__global__ __launch_bounds__(G == 8 ? 2 * TL_THREADS : TL_THREADS) void k_commit_tally(const bsx_validator* __restrict__ vals, uint32_t v_max, uint32_t n_commits,
                                                             const uint8_t* __restrict__ header_hashes,
                                                             const uint8_t* __restrict__ ok_in,
                                                             bsx_commit_result* __restrict__ results, bsxk_unit_dst wit) {
    // dynamic LDS sized by the padded validator count P (launcher): nodes[2][G * P * 8] u32, then en[2][G * P] u8 — 8.4 KB per commit at
    // V = 100 instead of the 33 KB of the 512-slot maximum, so that 2048 commits are resident at once (mode S)
    extern __shared__ uint32_t tl_lds[];
    __shared__ unsigned long long s_total[G], s_signed[G], s_trusted[G], s_total_hi[G], s_total_lo[G];
    __shared__ uint32_t s_nen[G], s_nsig[G], s_nbad[G], s_firstbad[G], s_nbadmsg[G];
    const uint32_t c0 = blockIdx.x * G, tid = threadIdx.x;
    uint32_t P = 1;
    while (P < v_max) P *= 2;
    // two node buffers: [0] holds a level of up to G * P nodes (the leaves), [1] the level above (G * P / 2) — every later level fits the
    // buffer it lands in, so 1.5 x the leaves' size is enough (G = 8, P = 128: 48 KB + 1.5 KB of flags)
    uint32_t* const nodes0 = tl_lds;
    uint8_t* const en0 = reinterpret_cast<uint8_t*>(tl_lds + (G * P * 8 + G * (P / 2 ? P / 2 : 1) * 8));
#define nodes(b, idx) nodes0[(b) * G * P * 8 + (idx)]
#define en(b, idx) en0[(b) * G * P + (idx)]
    const uint32_t nthreads = blockDim.x;
    if (tid < G) { s_total[tid] = 0; s_signed[tid] = 0; s_trusted[tid] = 0; s_total_hi[tid] = 0; s_total_lo[tid] = 0; s_nen[tid] = 0; s_nsig[tid] = 0; s_nbad[tid] = 0; s_firstbad[tid] = 0xffffffffu; s_nbadmsg[tid] = 0; }
    __syncthreads();
    // witness destinations
    const bool w_on = wit.base != nullptr, w_commit = w_on && wit.mode == 0;
    const uint32_t o_leaf = w_commit ? bsx_cm_off_leaf(v_max) : bsx_sk_off_leaf(v_max);
    const uint32_t o_lh = w_commit ? bsx_cm_off_leaf_hash(v_max) : bsx_sk_off_leaf_hash(v_max);
    const uint32_t o_inner = w_commit ? bsx_cm_off_inner(v_max) : bsx_sk_off_inner(v_max);
    const uint32_t o_node = w_commit ? bsx_cm_off_node(v_max) : bsx_sk_off_node(v_max);
    const uint32_t o_root = w_commit ? bsx_cm_off_root(v_max) : bsx_sk_off_root(v_max);
    const uint32_t b_leaf_en = w_commit ? bsx_cm_b_leaf_enabled(v_max) : bsx_sk_b_leaf_enabled(v_max);
    const uint32_t b_node_en = w_commit ? bsx_cm_b_node_enabled(v_max) : bsx_sk_b_node_enabled(v_max);

    // leaves: leaf index li = g * P + v over the workgroup's G commits.  P is a multiple of 64 or below it: with G > 1 (P <= 128, a
    // power of two >= 64 is not required) the lanes of a wave may span commits, so the per-commit sums go through LDS atomics per lane
    // group: every lane adds its own contribution only when it has one (enabled slots), wave-reduced when the wave is uniform in g
    for (uint32_t li = tid; li < (uint32_t)G * P; li += nthreads) {
        const uint32_t g = G == 1 ? 0u : li / P, v = G == 1 ? li : li % P;
        const uint32_t c = c0 + g;
        const bool live = c < n_commits;
        const bsx_validator* cv = vals + (uint64_t)(live ? c : 0) * v_max;
        uint8_t* const cw = (w_on && live) ? wit.base + (uint64_t)c * wit.stride : nullptr;
        uint32_t* const WW = cw ? reinterpret_cast<uint32_t*>(cw + wit.off_words) : nullptr;
        uint8_t* const WB = cw ? cw + wit.off_bools : nullptr;
        uint32_t hh[8];
#pragma unroll
        for (int k = 0; k < 8; k++) hh[k] = (header_hashes && live) ? reinterpret_cast<const uint32_t*>(header_hashes + 32 * (uint64_t)c)[k] : 0u;
        if (w_commit && cw && v == 0) {
            uint4* hd = reinterpret_cast<uint4*>(cw + bsx_cm_off_header_hash());
            hd[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]); hd[1] = make_uint4(hh[4], hh[5], hh[6], hh[7]);
        }
        uint64_t total = 0, signedp = 0, trusted = 0;
        uint64_t total_hi = 0, total_lo = 0;     // exact sum in two halves: the u64 `total` may wrap
        uint32_t nen = 0, nsig = 0, nbad = 0, nbadmsg = 0;
        uint32_t pk[8];
        uint64_t power = 0;
        bool enabled = false;
        if (live && v < v_max) {
            const uint4* rec = reinterpret_cast<const uint4*>(cv + v);
            const uint4 p0 = rec[0], p1 = rec[1], fl = rec[14];
            pk[0] = p0.x; pk[1] = p0.y; pk[2] = p0.z; pk[3] = p0.w; pk[4] = p1.x; pk[5] = p1.y; pk[6] = p1.z; pk[7] = p1.w;
            power = (uint64_t)fl.x | ((uint64_t)fl.y << 32);
            enabled = (fl.z & 0xffu) != 0;
            const bool is_signed = ((fl.z >> 8) & 0xffu) != 0, present = ((fl.z >> 16) & 0xffu) != 0;
            bool sig = false, msg = false, has_round = false;
            uint32_t mlen = 0;
            if ((enabled && is_signed) || w_commit) {
                // the signed message must carry the header hash at offset 16 (25 with a round field); evaluated for every slot
                // when the witness is emitted (static circuit), only where it counts otherwise
                const uint32_t* mw = reinterpret_cast<const uint32_t*>(cv[v].message);
                mlen = cv[v].message_len;
                has_round = mlen > 12 && (mw[3] & 0xffu) == 0x19u;
                const uint32_t off = has_round ? 25u : 16u;
                msg = mlen <= BSX_VALIDATOR_MSG_MAX && mlen >= off + 32;
                uint32_t diff = 0;
                if (has_round) {
#pragma unroll
                    for (int k = 0; k < 8; k++) diff |= funnel_r(mw[7 + k], mw[6 + k], 8) ^ hh[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) diff |= mw[4 + k] ^ hh[k];
                }
                msg = msg && diff == 0;
            }
            if (enabled) {
                nen++;
                total += power;
                total_hi += power >> 32; total_lo += power & 0xffffffffull;
                if (is_signed) {
                    nsig++;
                    sig = ok_in ? (ok_in[(uint64_t)c * v_max + v] == 1) : false;     // a deferred / pending marker never counts as valid
                    if (!sig) { nbad++; atomicMin(&s_firstbad[g], v); }
                    if (!msg) nbadmsg++;
                    if (sig && msg) { signedp += power; if (present) trusted += power; }
                }
            }
            if (w_commit) {
                uint32_t* ws = WW + BSX_CM_SLOT_WORDS * v;
                ws[0] = mlen; ws[2] = (uint32_t)power; ws[3] = (uint32_t)(power >> 32);
                uint8_t* b = WB + BSX_CM_SLOT_BOOLS * v;
                b[0] = enabled; b[1] = is_signed; b[2] = present; b[3] = sig; b[4] = has_round; b[5] = msg;
                b[6] = enabled && is_signed && sig && msg;
            } else if (w_on) {
                uint32_t* ws = WW + BSX_SK_W_SLOTS + 3 * v;
                ws[1] = (uint32_t)power; ws[2] = (uint32_t)(power >> 32);
                WB[2 * v] = enabled;
                uint4* pkd = reinterpret_cast<uint4*>(cw + bsx_sk_off_pubkeys(v_max) + 32 * v);
                pkd[0] = p0; pkd[1] = p1;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) pk[k] = 0;
        }
        uint32_t d[14];
        const int len = validator_leaf(pk, power, d);
        {
            const Digest lh = leaf_hash_1block(d, len);
#pragma unroll
            for (int k = 0; k < 8; k++) nodes(0, li * 8 + k) = lh.w[k];
            en(0, li) = enabled ? 1 : 0;
            if (cw) {
                if (v < v_max) {
                    // the leaf's bytes beyond its length are zero by construction (validator_leaf fills static positions)
                    uint4* ld = reinterpret_cast<uint4*>(cw + o_leaf + 48 * v);
                    ld[0] = make_uint4(d[0], d[1], d[2], d[3]); ld[1] = make_uint4(d[4], d[5], d[6], d[7]); ld[2] = make_uint4(d[8], d[9], d[10], d[11]);
                    WW[w_commit ? BSX_CM_SLOT_WORDS * v + 1 : BSX_SK_W_SLOTS + 3 * v] = (uint32_t)len;
                }
                uint4* hd = reinterpret_cast<uint4*>(cw + o_lh + 32 * v);
                hd[0] = make_uint4(bswap32(lh.w[0]), bswap32(lh.w[1]), bswap32(lh.w[2]), bswap32(lh.w[3]));
                hd[1] = make_uint4(bswap32(lh.w[4]), bswap32(lh.w[5]), bswap32(lh.w[6]), bswap32(lh.w[7]));
                WB[b_leaf_en + v] = enabled ? 1 : 0;
            }
        }
        // the commit's sums: a wave whose 64 leaves belong to ONE commit (P >= 64: always) reduces with shuffles and issues one LDS
        // atomic per counter; smaller trees add per lane
        if (P >= 64) {
            total = wave_sum_u64(total); signedp = wave_sum_u64(signedp); trusted = wave_sum_u64(trusted);
            total_hi = wave_sum_u64(total_hi); total_lo = wave_sum_u64(total_lo);
            nen = wave_sum_u32(nen); nsig = wave_sum_u32(nsig); nbad = wave_sum_u32(nbad); nbadmsg = wave_sum_u32(nbadmsg);
            if ((tid & 63) == 0) {
                atomicAdd(&s_total[g], (unsigned long long)total); atomicAdd(&s_signed[g], (unsigned long long)signedp);
                atomicAdd(&s_trusted[g], (unsigned long long)trusted);
                atomicAdd(&s_total_hi[g], (unsigned long long)total_hi); atomicAdd(&s_total_lo[g], (unsigned long long)total_lo);
                atomicAdd(&s_nen[g], nen); atomicAdd(&s_nsig[g], nsig); atomicAdd(&s_nbad[g], nbad); atomicAdd(&s_nbadmsg[g], nbadmsg);
            }
        } else if (nen) {
            atomicAdd(&s_total[g], (unsigned long long)total); atomicAdd(&s_signed[g], (unsigned long long)signedp);
            atomicAdd(&s_trusted[g], (unsigned long long)trusted);
            atomicAdd(&s_total_hi[g], (unsigned long long)total_hi); atomicAdd(&s_total_lo[g], (unsigned long long)total_lo);
            atomicAdd(&s_nen[g], nen); atomicAdd(&s_nsig[g], nsig); atomicAdd(&s_nbad[g], nbad); atomicAdd(&s_nbadmsg[g], nbadmsg);
        }
    }
    __syncthreads();
    int cur = 0;
    uint32_t level_off = 0;
    // a wave beyond the level's nodes has none left on this or any later level: it ends (a barrier counts the surviving waves), and its
    // slot goes to the next workgroup instead of idling through the narrow levels.  Node ni of a level = commit ni / width, node ni % width
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(tid & ~63u);
    for (uint32_t width = P / 2; width >= 1; width /= 2) {
        if (wave_base && wave_base >= (uint32_t)G * width) return;
        for (uint32_t ni = tid; ni < (uint32_t)G * width; ni += nthreads) {
            const uint32_t g = G == 1 ? 0u : ni / width, t = G == 1 ? ni : ni % width;
            const uint32_t src = g * (2 * width) + 2 * t;          // this level's children live densely: commit g's 2 * width nodes first
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = nodes(cur, src * 8 + k); r.w[k] = nodes(cur, (src + 1) * 8 + k); }
            const bool el = en(cur, src) != 0, er = en(cur, src + 1) != 0;
            const Digest in = inner_hash(l, r);
            const Digest node = (el && er) ? in : l;
#pragma unroll
            for (int k = 0; k < 8; k++) nodes(cur ^ 1, ni * 8 + k) = node.w[k];
            en(cur ^ 1, ni) = (el || er) ? 1 : 0;
            if (w_on && c0 + g < n_commits) {
                uint8_t* const cw = wit.base + (uint64_t)(c0 + g) * wit.stride;
                uint4* di = reinterpret_cast<uint4*>(cw + o_inner + 32 * (level_off + t));
                di[0] = make_uint4(bswap32(in.w[0]), bswap32(in.w[1]), bswap32(in.w[2]), bswap32(in.w[3]));
                di[1] = make_uint4(bswap32(in.w[4]), bswap32(in.w[5]), bswap32(in.w[6]), bswap32(in.w[7]));
                uint4* dn = reinterpret_cast<uint4*>(cw + o_node + 32 * (level_off + t));
                dn[0] = make_uint4(bswap32(node.w[0]), bswap32(node.w[1]), bswap32(node.w[2]), bswap32(node.w[3]));
                dn[1] = make_uint4(bswap32(node.w[4]), bswap32(node.w[5]), bswap32(node.w[6]), bswap32(node.w[7]));
                (cw + wit.off_bools)[b_node_en + level_off + t] = (el || er) ? 1 : 0;
            }
        }
        __syncthreads();
        cur ^= 1;
        level_off += width;
    }
    // the roots: node g of the last level.  P == 1 (a single slot): the leaf itself
    if (tid < 8 * G) {
        const uint32_t g = tid >> 3, k = tid & 7;
        if (w_on && c0 + g < n_commits) reinterpret_cast<uint32_t*>(wit.base + (uint64_t)(c0 + g) * wit.stride + o_root)[k] = bswap32(nodes(cur, g * 8 + k));
    }
    if (tid < G && c0 + tid < n_commits) {
        const uint32_t g = tid;
        bsx_commit_result* o = results + c0 + g;
#pragma unroll
        for (int k = 0; k < 8; k++) reinterpret_cast<uint32_t*>(o->validators_hash)[k] = bswap32(nodes(cur, g * 8 + k));
        o->total_power = s_total[g]; o->signed_power = s_signed[g]; o->trusted_signed_power = s_trusted[g];
        o->n_enabled = s_nen[g]; o->n_signed = s_nsig[g]; o->n_bad_signature = s_nbad[g]; o->first_bad_signature = s_firstbad[g];
        o->n_bad_message = s_nbadmsg[g];
        const bool overflow = (((unsigned __int128)s_total_hi[g] << 32) + s_total_lo[g]) > (unsigned __int128)BSX_MAX_TOTAL_VOTING_POWER;
        const bool two_thirds = !overflow && (unsigned __int128)s_signed[g] * 3 > (unsigned __int128)s_total[g] * 2;
        o->two_thirds_ok = two_thirds ? 1u : 0u;
        o->power_overflow = overflow ? 1u : 0u;
        o->_pad[0] = o->_pad[1] = o->_pad[2] = 0;
        if (w_on) {
            uint8_t* const cw = wit.base + (uint64_t)(c0 + g) * wit.stride;
            uint32_t* const WW = reinterpret_cast<uint32_t*>(cw + wit.off_words);
            uint8_t* const WB = cw + wit.off_bools;
            if (w_commit) {
                uint32_t* wt = WW + bsx_cm_w_total(v_max);
                wt[0] = (uint32_t)s_total[g]; wt[1] = (uint32_t)(s_total[g] >> 32);
                wt[2] = (uint32_t)s_signed[g]; wt[3] = (uint32_t)(s_signed[g] >> 32);
                wt[4] = (uint32_t)s_trusted[g]; wt[5] = (uint32_t)(s_trusted[g] >> 32);
                uint8_t* t = WB + bsx_cm_b_tail(v_max);
                t[0] = two_thirds; t[1] = overflow; t[2] = (s_nbad[g] == 0 && s_nbadmsg[g] == 0);
            } else {
                uint32_t* wt = WW + bsx_sk_w_total(v_max);
                wt[0] = (uint32_t)s_total[g]; wt[1] = (uint32_t)(s_total[g] >> 32);
            }
        }
    }
#undef nodes
#undef en
}

// ------------------------------------------------------------------------------------------------ k_commit_sums
// The signature-dependent half of k_commit_tally on its own: per-slot verdicts (signature valid, message carries the header hash),
// the signed / trusted-signed power sums, the counters and the 2/3 rule — no hashing.  The host tier runs k_commit_tally EARLY with
// ok_in = nullptr (validator leaves, the masked tree, total power: nothing there depends on the signatures) beside the R decoding,
// and this kernel behind the signature check: the commit chain of a proof request loses the tree's ~50 us (round 4; the chain was
// verify -> tally -> skip conditions).  Completes the bsx_commit_result k_commit_tally began (validators_hash, total_power,
// n_enabled, power_overflow stay) and, with `wit`, the COMMIT unit's slot bools / sums / verdict bools.  One workgroup per commit.
__global__ __launch_bounds__(TL_THREADS) void k_commit_sums(const bsx_validator* __restrict__ vals, uint32_t v_max,
                                                            const uint8_t* __restrict__ header_hashes, const uint8_t* __restrict__ ok_in,
                                                            bsx_commit_result* __restrict__ results, bsxk_unit_dst wit) {
    __shared__ unsigned long long s_signed, s_trusted;
    __shared__ uint32_t s_nsig, s_nbad, s_firstbad, s_nbadmsg;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    const bsx_validator* cv = vals + (uint64_t)c * v_max;
    if (tid == 0) { s_signed = 0; s_trusted = 0; s_nsig = 0; s_nbad = 0; s_firstbad = 0xffffffffu; s_nbadmsg = 0; }
    __syncthreads();
    uint32_t hh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) hh[k] = reinterpret_cast<const uint32_t*>(header_hashes + 32 * (uint64_t)c)[k];
    const bool w_on = wit.base != nullptr;
    uint8_t* const cw = w_on ? wit.base + (uint64_t)c * wit.stride : nullptr;
    uint32_t* const WW = w_on ? reinterpret_cast<uint32_t*>(cw + wit.off_words) : nullptr;
    uint8_t* const WB = w_on ? cw + wit.off_bools : nullptr;
    if (w_on && tid < 8) reinterpret_cast<uint32_t*>(cw + bsx_cm_off_header_hash())[tid] = hh[tid];
    uint64_t signedp = 0, trusted = 0;
    uint32_t nsig = 0, nbad = 0, nbadmsg = 0;
    for (uint32_t v = tid; v < v_max; v += nthreads) {
        const uint4 fl = reinterpret_cast<const uint4*>(cv + v)[14];
        const uint64_t power = (uint64_t)fl.x | ((uint64_t)fl.y << 32);
        const bool enabled = (fl.z & 0xffu) != 0, is_signed = ((fl.z >> 8) & 0xffu) != 0, present = ((fl.z >> 16) & 0xffu) != 0;
        bool sig = false, msg = false, has_round = false;
        uint32_t mlen = 0;
        if ((enabled && is_signed) || w_on) {
            const uint32_t* mw = reinterpret_cast<const uint32_t*>(cv[v].message);
            mlen = cv[v].message_len;
            has_round = mlen > 12 && (mw[3] & 0xffu) == 0x19u;
            const uint32_t off = has_round ? 25u : 16u;
            msg = mlen <= BSX_VALIDATOR_MSG_MAX && mlen >= off + 32;
            uint32_t diff = 0;
            if (has_round) {
#pragma unroll
                for (int k = 0; k < 8; k++) diff |= funnel_r(mw[7 + k], mw[6 + k], 8) ^ hh[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) diff |= mw[4 + k] ^ hh[k];
            }
            msg = msg && diff == 0;
        }
        if (enabled && is_signed) {
            nsig++;
            sig = ok_in[(uint64_t)c * v_max + v] == 1;
            if (!sig) { nbad++; atomicMin(&s_firstbad, v); }
            if (!msg) nbadmsg++;
            if (sig && msg) { signedp += power; if (present) trusted += power; }
        }
        if (w_on) {
            WW[BSX_CM_SLOT_WORDS * v] = mlen;
            uint8_t* b = WB + BSX_CM_SLOT_BOOLS * v;
            b[3] = sig; b[4] = has_round; b[5] = msg; b[6] = enabled && is_signed && sig && msg;
        }
    }
    signedp = wave_sum_u64(signedp); trusted = wave_sum_u64(trusted);
    nsig = wave_sum_u32(nsig); nbad = wave_sum_u32(nbad); nbadmsg = wave_sum_u32(nbadmsg);
    if ((tid & 63) == 0) {
        atomicAdd(&s_signed, (unsigned long long)signedp); atomicAdd(&s_trusted, (unsigned long long)trusted);
        atomicAdd(&s_nsig, nsig); atomicAdd(&s_nbad, nbad); atomicAdd(&s_nbadmsg, nbadmsg);
    }
    __syncthreads();
    if (tid == 0) {
        bsx_commit_result* o = results + c;
        o->signed_power = s_signed; o->trusted_signed_power = s_trusted;
        o->n_signed = s_nsig; o->n_bad_signature = s_nbad; o->first_bad_signature = s_firstbad; o->n_bad_message = s_nbadmsg;
        const bool overflow = o->power_overflow != 0;
        const bool two_thirds = !overflow && (unsigned __int128)s_signed * 3 > (unsigned __int128)o->total_power * 2;
        o->two_thirds_ok = two_thirds ? 1u : 0u;
        if (w_on) {
            uint32_t* wt = WW + bsx_cm_w_total(v_max);
            wt[2] = (uint32_t)s_signed; wt[3] = (uint32_t)(s_signed >> 32);
            wt[4] = (uint32_t)s_trusted; wt[5] = (uint32_t)(s_trusted >> 32);
            uint8_t* t = WB + bsx_cm_b_tail(v_max);
            t[0] = two_thirds; t[2] = (s_nbad == 0 && s_nbadmsg == 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_skip_check
// One workgroup per range.  skip_status[r] = bsx_status of the skip part (BSX_OK when every condition holds).
struct SkipArgs {
    uint32_t n_ranges, v_max;
    const bsx_shared_ctx* ranges;
    const bsx_header* headers;          // range r: headers[r*hpr + (height - S_r)]
    uint64_t headers_per_range;
    const uint8_t* hashes;              // same indexing, 32 B each
    const bsx_validator* target;        // n_ranges * v_max
    const bsx_validator* trusted;       // n_ranges * v_max
    const uint8_t* target_ok;           // n_ranges * v_max (signature valid)
    bsx_commit_result* target_res;      // n_ranges (trusted_signed_power is overwritten with the trusted-set overlap)
    const bsx_commit_result* trusted_res;   // n_ranges (validators_hash + total_power of the trusted set)
    uint32_t* skip_status;              // n_ranges
    uint8_t* target_hashes;             // n_ranges * 32 (out): hash of the target header
    const uint32_t* target_idx;         // optional: index of the target header inside the range's header block (default E - S)
    uint32_t chain_id_len;              // C::CHAIN_ID_BYTES (header_range.rs:42-43): the target header's field 1 must be 0a len bytes
    uint8_t chain_id[52];
    bsxk_unit_dst wit;                  // SKIP units (include/bsx_layout.h): public inputs, target hash, signed_target bools, overlap, check bools
};
__global__ __launch_bounds__(256) void k_skip_check(SkipArgs a) {
    __shared__ uint32_t tpk[TL_VMAX * 8];
    __shared__ uint8_t tsig[TL_VMAX];
    __shared__ unsigned long long s_overlap;
    const uint32_t r = blockIdx.x, tid = threadIdx.x, V = a.v_max;
    const bsx_shared_ctx rg = a.ranges[r];
    const bsx_validator* tv = a.target + (uint64_t)r * V;
    const bsx_validator* rv = a.trusted + (uint64_t)r * V;
    if (tid == 0) s_overlap = 0;
    for (uint32_t k = tid; k < V; k += 256) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(tv[k].pubkey);
#pragma unroll
        for (int q = 0; q < 8; q++) tpk[k * 8 + q] = p[q];
        tsig[k] = (tv[k].enabled && tv[k].is_signed && a.target_ok[(uint64_t)r * V + k] == 1) ? 1 : 0;
    }
    __syncthreads();
    uint8_t* const scw = a.wit.base ? a.wit.base + (uint64_t)r * a.wit.stride : nullptr;
    uint32_t* const SW = scw ? reinterpret_cast<uint32_t*>(scw + a.wit.off_words) : nullptr;
    uint8_t* const SB = scw ? scw + a.wit.off_bools : nullptr;
    uint64_t ov = 0;
    for (uint32_t i = tid; i < V; i += 256) {
        if (SB) SB[2 * i + 1] = 0;
        if (!rv[i].enabled) continue;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(rv[i].pubkey);
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pk[q] = p[q];
        bool found = false;
        for (uint32_t k = 0; k < V && !found; k++) {
            if (!tsig[k]) continue;
            uint32_t d = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) d |= tpk[k * 8 + q] ^ pk[q];
            found = d == 0;
        }
        if (found) ov += rv[i].voting_power;
        if (SB && found) SB[2 * i + 1] = 1;
    }
    ov = wave_sum_u64(ov);
    if ((tid & 63) == 0) atomicAdd(&s_overlap, (unsigned long long)ov);
    __syncthreads();
    // Byte comparisons spread over the first wave (lane q owns byte q of every 32-byte pair) and reduced with ballots:
    // one memory round trip instead of ~150 dependent byte loads by a single lane (0.04 ms alone either way, but
    // 0.4-0.6 ms beside the expansion, at the very end of the commit chain).
    if (tid < 64) {
        const uint32_t q = tid & 31;
        const uint64_t S = rg.start_block, E = rg.end_block;
        const uint64_t ti = a.target_idx ? (uint64_t)a.target_idx[r] : (E - S);
        const bsx_header* th = a.headers + (uint64_t)r * a.headers_per_range + ti;
        const bsx_header* tr = a.headers + (uint64_t)r * a.headers_per_range;
        const uint8_t* thash = a.hashes + ((uint64_t)r * a.headers_per_range + ti) * 32;
        const uint8_t* trhash = a.hashes + ((uint64_t)r * a.headers_per_range) * 32;
        bsx_commit_result* cr = a.target_res + r;
        const bsx_commit_result* trc = a.trusted_res + r;
        // loads first
        const uint8_t b_trhash = trhash[q], b_thash = thash[q];
        const uint8_t b_tvh = th->hash[2][2 + q], b_rvh = tr->hash[2][2 + q];
        const uint8_t b_cvh = cr->validators_hash[q], b_tcvh = trc->validators_hash[q];
        const uint8_t b_height = th->height[q < 12 ? q : 0];
        const uint8_t b_chain = th->chain_id[tid < 52 ? tid : 0];          // all 64 lanes: the leaf is up to 52 bytes
        const uint8_t l_chain = th->len[1];
        const uint8_t l_height = th->len[BSX_BLOCK_HEIGHT_INDEX], l_tv = th->len[7], l_rv = tr->len[7];
        const uint32_t n_bad_sig = cr->n_bad_signature, n_bad_msg = cr->n_bad_message, two_thirds = cr->two_thirds_ok;
        const uint32_t overflow = cr->power_overflow | trc->power_overflow;
        const uint64_t ttotal64 = trc->total_power;
        // the target header's height leaf must encode the target block (varint): byte q of 08 varint(E)
        int hn = 1;
        uint8_t want = 0x08;
        {
            uint64_t hv = E;
            int pos = 1;
            for (;;) {
                const bool more = hv >= 0x80;
                const uint8_t byte = (uint8_t)(more ? (hv | 0x80) : hv);
                if ((int)q == pos) want = byte;
                pos++;
                if (!more) break;
                hv >>= 7;
            }
            hn = pos;
        }
        const bool eq_trusted = __ballot(b_trhash != rg.start_header_hash[q]) == 0;     // trusted header hash is the public input
        const bool heq = (l_height == hn) && __ballot((int)q < hn && b_height != want) == 0;
        // chain-id leaf = 0a len bytes
        const uint32_t cl = a.chain_id_len;
        const uint8_t want_c = tid == 0 ? 0x0a : tid == 1 ? (uint8_t)cl : a.chain_id[tid >= 2 && tid < 52 ? tid - 2 : 0];
        const bool ceq = (l_chain == cl + 2) && __ballot(tid < cl + 2 && b_chain != want_c) == 0;
        const bool veq = (l_tv == 34) && __ballot(b_tvh != b_cvh) == 0;
        const bool treq = (l_rv == 34) && __ballot(b_rvh != b_tcvh) == 0;
        if (a.target_hashes && tid < 32) a.target_hashes[32 * (uint64_t)r + q] = b_thash;
        if (scw && tid < 32) { scw[q] = rg.start_header_hash[q]; scw[32 + q] = b_thash; }      // header_range.rs:34, :57
        if (tid == 0) {
            uint32_t st = BSX_OK;
            if (overflow) st = BSX_ERR_BAD_ARG;          // voting powers beyond MaxTotalVotingPower: the tallies cannot be trusted
            if (!st && !eq_trusted) st = BSX_ERR_ASSERT;
            if (!st && !heq) st = BSX_ERR_ASSERT;
            if (!st && !ceq) st = BSX_ERR_ASSERT;
            if (!st && (n_bad_sig || n_bad_msg)) st = BSX_ERR_BAD_SIGNATURE;
            if (!st && !veq) st = BSX_ERR_ASSERT;
            if (!st && !treq) st = BSX_ERR_ASSERT;
            if (!st && !two_thirds) st = BSX_ERR_VOTING_POWER;
            const unsigned __int128 overlap = s_overlap, ttotal = ttotal64;
            if (!st && !(overlap * 3 > ttotal)) st = BSX_ERR_VOTING_POWER;
            cr->trusted_signed_power = s_overlap;
            a.skip_status[r] = st;
            if (scw) {
                SW[BSX_SK_W_TRUSTED_BLOCK] = (uint32_t)S; SW[BSX_SK_W_TRUSTED_BLOCK + 1] = (uint32_t)(S >> 32);
                SW[BSX_SK_W_TARGET_BLOCK] = (uint32_t)E; SW[BSX_SK_W_TARGET_BLOCK + 1] = (uint32_t)(E >> 32);
                uint32_t* wt = SW + bsx_sk_w_total(V);
                wt[2] = (uint32_t)s_overlap; wt[3] = (uint32_t)(s_overlap >> 32);
                uint8_t* c = SB + bsx_sk_b_checks(V);
                c[0] = eq_trusted; c[1] = heq; c[2] = ceq; c[3] = !(n_bad_sig || n_bad_msg); c[4] = veq; c[5] = treq; c[6] = two_thirds != 0;
                c[7] = overlap * 3 > ttotal; c[8] = overflow != 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_skip_eval
// Operator skip-target search (fetcher.rs:60-87): one workgroup per candidate target evaluates is_valid_skip — the
// start-set power of the validators that signed the candidate's commit against 1/3 of the start set ([UPSTREAM]
// predicate, see bsx.h).  Candidate keys are staged in LDS; each lane owns start validators and scans them.
__global__ __launch_bounds__(256) void k_skip_eval(const bsx_validator* __restrict__ start, const bsx_validator* __restrict__ cand,
                                                   uint32_t v_max, bsx_skip_eval* __restrict__ out) {
    __shared__ uint32_t tpk[TL_VMAX * 8];
    __shared__ uint8_t tsig[TL_VMAX];
    __shared__ unsigned long long acc[8];          // overlap, start total, signed, target total; exact halves of the two totals
    const uint32_t c = blockIdx.x, tid = threadIdx.x, V = v_max;
    const bsx_validator* tv = cand + (uint64_t)c * V;
    if (tid < 8) acc[tid] = 0;
    uint64_t signed_p = 0, target_total = 0, tt_hi = 0, tt_lo = 0, st_hi = 0, st_lo = 0;
    for (uint32_t k = tid; k < V; k += 256) {
        uint32_t pk[8];
        load_pk(tv + k, pk);
#pragma unroll
        for (int q = 0; q < 8; q++) tpk[k * 8 + q] = pk[q];
        const uint4 flags = reinterpret_cast<const uint4*>(tv + k)[14];      // voting_power (8), enabled, is_signed, ...
        const bool en = (flags.z & 0xffu) != 0, sg = ((flags.z >> 8) & 0xffu) != 0;
        const uint64_t power = (uint64_t)flags.x | ((uint64_t)flags.y << 32);
        tsig[k] = (en && sg) ? 1 : 0;
        if (en) { target_total += power; tt_hi += power >> 32; tt_lo += power & 0xffffffffull; }
        if (en && sg) signed_p += power;
    }
    __syncthreads();
    uint64_t ov = 0, start_total = 0;
    for (uint32_t i = tid; i < V; i += 256) {
        const uint4 flags = reinterpret_cast<const uint4*>(start + i)[14];
        if ((flags.z & 0xffu) == 0) continue;
        const uint64_t power = (uint64_t)flags.x | ((uint64_t)flags.y << 32);
        start_total += power;
        st_hi += power >> 32; st_lo += power & 0xffffffffull;
        uint32_t pk[8];
        load_pk(start + i, pk);
        bool found = false;
        for (uint32_t k = 0; k < V && !found; k++) {
            if (!tsig[k]) continue;
            uint32_t d = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) d |= tpk[k * 8 + q] ^ pk[q];
            found = d == 0;
        }
        if (found) ov += power;
    }
    ov = wave_sum_u64(ov); start_total = wave_sum_u64(start_total);
    signed_p = wave_sum_u64(signed_p); target_total = wave_sum_u64(target_total);
    tt_hi = wave_sum_u64(tt_hi); tt_lo = wave_sum_u64(tt_lo); st_hi = wave_sum_u64(st_hi); st_lo = wave_sum_u64(st_lo);
    if ((tid & 63) == 0) {
        atomicAdd(&acc[4], (unsigned long long)tt_hi); atomicAdd(&acc[5], (unsigned long long)tt_lo);
        atomicAdd(&acc[6], (unsigned long long)st_hi); atomicAdd(&acc[7], (unsigned long long)st_lo);
        atomicAdd(&acc[0], (unsigned long long)ov); atomicAdd(&acc[1], (unsigned long long)start_total);
        atomicAdd(&acc[2], (unsigned long long)signed_p); atomicAdd(&acc[3], (unsigned long long)target_total);
    }
    __syncthreads();
    if (tid == 0) {
        bsx_skip_eval e;
        e.overlap_power = acc[0]; e.start_total_power = acc[1]; e.signed_power = acc[2]; e.target_total_power = acc[3];
        const unsigned __int128 cap = BSX_MAX_TOTAL_VOTING_POWER;
        const bool overflow = (((unsigned __int128)acc[4] << 32) + acc[5]) > cap || (((unsigned __int128)acc[6] << 32) + acc[7]) > cap;
        e.valid = (!overflow && (unsigned __int128)acc[0] * 3 > (unsigned __int128)acc[1]) ? 1u : 0u;
        e.power_overflow = overflow ? 1u : 0u;
        out[c] = e;
    }
}

}  // namespace bsx

extern "C" {
using namespace bsx;
hipError_t bsxk_sha512_challenge(hipStream_t s, const bsx_validator* vals, uint64_t n, uint8_t* h, uint8_t* digest, uint32_t v_max,
                                 const bsxk_unit_dst* wit) {
    if (!n) return hipSuccess;
    const bsxk_unit_dst w = wit ? *wit : bsxk_unit_dst{nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_sha512_challenge, dim3((uint32_t)((n + CH_THREADS - 1) / CH_THREADS)), dim3(CH_THREADS), 0, s, vals, n, h, digest,
                       v_max ? v_max : 1u, w);
    return hipGetLastError();
}
hipError_t bsxk_ed25519_verify(hipStream_t s, const bsx_validator* vals, const uint8_t* h, uint64_t n, uint8_t* ok) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_ed25519_verify<false>, dim3((uint32_t)((n + ED_THREADS - 1) / ED_THREADS)), dim3(ED_THREADS), 0, s, vals, h, n, ok);
    return hipGetLastError();
}
int bsxk_keytable_default_bits(void) { return KT_W; }
int bsxk_keytable_bits_ok(int w) { return kt_w_ok(w) ? 1 : 0; }
uint64_t bsxk_keytable_bytes(uint32_t n_keys, int w) { return kt_bytes(n_keys, w); }
hipError_t bsxk_ed25519_keytable(hipStream_t s, const bsx_validator* vals, uint32_t n_keys, uint8_t* table, int w) {
    if (n_keys == 0) return hipSuccess;
    if (!kt_w_ok(w)) return hipErrorInvalidValue;
    // experiments build only: BSX_KEYTABLE_REUSE=0 forces a full rebuild on every call (cold-build measurements)
    static const uint32_t force = bsx_knob("BSX_KEYTABLE_REUSE", 1) == 0 ? 1u : 0u;
    hipLaunchKernelGGL(k_keytable_check, dim3(1), dim3(KC_THREADS), 0, s, vals, n_keys, table, force, w);
    TableBuildArgs a{reinterpret_cast<const int32_t*>(table + kt_bases_off(n_keys)), reinterpret_cast<int32_t*>(table + kt_entries_off(n_keys, w)), table,
                     reinterpret_cast<const uint32_t*>(table + kt_flag_off(n_keys)), n_keys * (uint32_t)kt_parts(w), (uint32_t)kt_parts(w), (uint32_t)kt_half(w), (uint32_t)w};
    const uint64_t lanes = (uint64_t)a.n_rows * (kt_half(w) / KB_G);
    const uint64_t wgs = (lanes + ED_THREADS - 1) / ED_THREADS;
    hipLaunchKernelGGL(k_table_entries, dim3((uint32_t)(wgs < 1024 ? wgs : 1024)), dim3(ED_THREADS), 0, s, a);
    return hipGetLastError();
}
// the B table of a context (bsxk_ed25519_btable_bytes() bytes, 128-byte aligned): built once, on `s`
uint64_t bsxk_ed25519_btable_bytes() { return bt_bytes(); }
hipError_t bsxk_ed25519_btable(hipStream_t s, uint8_t* table) {
    hipLaunchKernelGGL(k_btable_bases, dim3(1), dim3(64), 0, s, table);
    TableBuildArgs a{reinterpret_cast<const int32_t*>(table), reinterpret_cast<int32_t*>(table + bt_entries_off()), nullptr, nullptr,
                     (uint32_t)BT_PARTS, (uint32_t)BT_PARTS, (uint32_t)BT_HALF_ENTRIES, (uint32_t)BT_W};
    const uint64_t lanes = (uint64_t)a.n_rows * (BT_HALF_ENTRIES / KB_G);
    const uint64_t wgs = (lanes + ED_THREADS - 1) / ED_THREADS;
    hipLaunchKernelGGL(k_table_entries, dim3((uint32_t)(wgs < 1024 ? wgs : 1024)), dim3(ED_THREADS), 0, s, a);
    return hipGetLastError();
}
uint64_t bsxk_ed25519_scratch_bytes(uint64_t n) { return n * ED_SLOT_I32 * 4; }
// The pass behind a fixed-key kernel for the slots it deferred (key != table row).  n_deferred = how many the CALLER knows there
// are (it compared the keys on the host when the validators were uploaded): 0 -> no launch at all; a few -> a 64-workgroup scan
// (one round of waves that read a marker byte per slot); many -> one wave per 64 slots, or every deferred slot's 256 doublings
// would queue up on 4096 lanes; < 0 = unknown -> the scan.
static inline uint32_t deferred_grid(uint64_t n, int64_t n_deferred) {
    const uint64_t wgs = (n + ED_THREADS - 1) / ED_THREADS;
    if (n_deferred > 4096) return (uint32_t)wgs;
    return (uint32_t)(wgs < 64 ? wgs : 64);
}
#define BSX_LAUNCH_DEFERRED()                                                                                                                   \
    do {                                                                                                                                        \
        if (n_deferred != 0)                                                                                                                    \
            hipLaunchKernelGGL(k_ed25519_verify<true>, dim3(deferred_grid(n, n_deferred)), dim3(ED_THREADS), 0, s, vals, h, n, ok);              \
    } while (0)
uint64_t bsxk_ed25519_rdec_bytes(uint64_t n) { return n * ED_RDEC_I32 * 4; }
hipError_t bsxk_ed25519_decode_r(hipStream_t s, const bsx_validator* vals, uint64_t n, void* rdec) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_ed25519_decode_r, dim3((uint32_t)((n + ED_THREADS - 1) / ED_THREADS)), dim3(ED_THREADS), 0, s, vals, n, static_cast<int32_t*>(rdec));
    return hipGetLastError();
}
// rdec (optional, from bsxk_ed25519_decode_r over the same records): batches below ED_SPLIT_BELOW signatures without a
// batch-inversion scratch take the latency form (k_ed25519_verify_keyed_proj)
hipError_t bsxk_ed25519_verify_keyed(hipStream_t s, const bsx_validator* vals, const uint8_t* h, uint64_t n, uint32_t v_max,
                                     const uint8_t* table, uint32_t n_keys, const uint8_t* btable, uint8_t* ok, void* scratch, const void* rdec,
                                     int64_t n_deferred, const uint32_t* rows, int w) {
    if (n == 0) return hipSuccess;
    if (!kt_w_ok(w)) return hipErrorInvalidValue;
    const int32_t* b_tab = reinterpret_cast<const int32_t*>(btable + bt_entries_off());
    if (rdec && rdec != BSXK_ED_THROUGHPUT && !scratch && n < ED_SPLIT_BELOW) {
        // BSX_ED_PROJ_SPLIT (experiments): 8 / 16 lanes per signature; default 16 while the launch is a single wave round anyway
        static const long env_ps = bsx_knob("BSX_ED_PROJ_SPLIT", 0);
        const bool s16 = env_ps ? env_ps == 16 : n <= 8192;
        const int32_t* rd = static_cast<const int32_t*>(rdec);
        BSX_NOTE_FORM(BSX_FORM_ED, 0x100u | (s16 ? 16u : 8u));
        if (s16) hipLaunchKernelGGL(k_ed25519_verify_keyed_proj<16>, dim3((uint32_t)((n + 3) / 4)), dim3(ED_THREADS), 0, s, vals, h, n, v_max, table, n_keys, b_tab, rd, ok, rows, w);
        else hipLaunchKernelGGL(k_ed25519_verify_keyed_proj<8>, dim3((uint32_t)((n + 7) / 8)), dim3(ED_THREADS), 0, s, vals, h, n, v_max, table, n_keys, b_tab, rd, ok, rows, w);
        BSX_LAUNCH_DEFERRED();
        return hipGetLastError();
    }
    int32_t* scr = static_cast<int32_t*>(scratch);
    const uint64_t n_commits = (n + v_max - 1) / v_max;
    // BSX_ED_BY_KEY (experiments): 0 / 1 forces the lane order; default: by key from 32 commits on (waves at least half full)
    static const long env_by_key = bsx_knob("BSX_ED_BY_KEY", -1);
    // BSX_ED_SPLIT (experiments): 1 / 4 lanes per signature; default: 4 while the batch cannot fill the GPU's wave slots
    // anyway (latency is what counts) or fills them so barely that finer units balance better, 1 above (20 % less work)
    static const long env_split = bsx_knob("BSX_ED_SPLIT", -1);
    // rdec == BSXK_ED_THROUGHPUT (with a scratch): the caller has a whole step of slack and an ALU-bound GPU (the compact
    // pipeline) — one lane per signature + batch inversion is the form with the least total work (280 multiplications per
    // signature against 421 on four lanes and 761 in the latency form), whatever the batch size
    const bool throughput = rdec == BSXK_ED_THROUGHPUT && scratch;
    const bool split4 = env_split >= 0 ? env_split == 4 : (n < ED_SPLIT_BELOW && !throughput);
    // two lanes per signature (with the batch-inversion scratch, lanes by key): the middle form, experiments build only for now
    const bool split2 = env_split == 2 && scratch != nullptr;
    const uint32_t sigs = split4 ? ED_THREADS / 4 : split2 ? ED_THREADS / 2 : ED_THREADS;               // signatures per workgroup
    const bool by_key = env_by_key >= 0 ? env_by_key != 0 : n_commits >= sigs / 2;
    dim3 grid;
    if (by_key) {
        const uint64_t wpk = (n_commits + sigs - 1) / sigs;
        grid = dim3((uint32_t)((wpk * v_max + 7) / 8 * 8));                   // k_ed25519_verify_keyed: the XCD remap needs a multiple of 8
    } else {
        grid = dim3((uint32_t)((n + sigs - 1) / sigs));
    }
    // one lane per signature, lanes by key, with the batch-inversion scratch (mode S): peel the waves beyond a whole number per SIMD
    // off into the four-lane form (k_ed25519_verify_keyed_mixed).  BSX_ED_MIXED=0 (experiments): never
    static const bool mixed_on = bsx_knob("BSX_ED_MIXED", 1) != 0;
    if (mixed_on && env_split < 0 && env_by_key < 0 && n_commits >= ED_THREADS / 2 && scr && !throughput) {
        const uint64_t simds = (uint64_t)bsxk_compute_units() * 4;
        const uint64_t waves = (n_commits + ED_THREADS - 1) / ED_THREADS * v_max, per = waves / simds, extra = waves % simds;
        // worth it while the batch is a few rounds deep and the extra waves would leave most SIMDs waiting for a few
        if (per >= 1 && per <= 6 && extra && extra * 2 <= simds) {
            const uint64_t wpk_a = per * simds / v_max;                                   // one-lane workgroups per key
            const uint64_t commits_a = wpk_a * ED_THREADS;
            if (wpk_a && commits_a < n_commits) {
                const uint32_t blocks_a = (uint32_t)((wpk_a * v_max + 7) / 8 * 8);
                const uint64_t wpk_b = (n_commits - commits_a + ED_THREADS / 4 - 1) / (ED_THREADS / 4);
                const uint32_t blocks_b = (uint32_t)((wpk_b * v_max + 7) / 8 * 8);
                BSX_NOTE_FORM(BSX_FORM_ED, 0x200u);
                hipLaunchKernelGGL(k_ed25519_verify_keyed_mixed, dim3(blocks_a + blocks_b), dim3(ED_THREADS), 0, s, vals, h, n, v_max, table, n_keys,
                                   b_tab, ok, scr, blocks_a, commits_a, rows, w);
                static const long env_kf = bsx_knob("BSX_ED_FIN_K", 0);        // experiments: signatures per batch inversion
                const uint32_t K = env_kf > 0 ? (uint32_t)env_kf : ed_fin_k(n);
                const uint64_t lanes = (n + K - 1) / K;
                hipLaunchKernelGGL(k_ed25519_finish, dim3((uint32_t)((lanes + ED_THREADS - 1) / ED_THREADS)), dim3(ED_THREADS), 0, s, vals, n, ok, scr, K);
                BSX_LAUNCH_DEFERRED();
                return hipGetLastError();
            }
        }
    }
#define BSX_LAUNCH_KEYED(DEFER_, BYKEY_, SPLIT_)                                                                        \
    do {                                                                                                                \
        BSX_NOTE_FORM(BSX_FORM_ED, 0x400u | (uint32_t)(SPLIT_) | ((BYKEY_) ? 0x10u : 0u) | ((DEFER_) ? 0x20u : 0u));    \
        hipLaunchKernelGGL((k_ed25519_verify_keyed<DEFER_, BYKEY_, SPLIT_>), grid, dim3(ED_THREADS), 0, s, vals, h, n, v_max, table, n_keys, b_tab, ok, scr, rows, w); \
    } while (0)
    // BSX_ED_SMALL=0 (experiments): no decode-R form for small batches
    static const bool small_form = bsx_knob("BSX_ED_SMALL", 1) != 0;
    if (split4 && !scr && !by_key && small_form) {
        BSX_NOTE_FORM(BSX_FORM_ED, 0x300u);
        hipLaunchKernelGGL(k_ed25519_verify_keyed_small, dim3((uint32_t)((n + EL_SIGS - 1) / EL_SIGS)), dim3(128), 0, s, vals, h, n, v_max, table,
                           n_keys, b_tab, ok, rows, w);
#ifdef BSX_EXPERIMENTS
    } else if (split2) {
        if (by_key) BSX_LAUNCH_KEYED(true, true, 2); else BSX_LAUNCH_KEYED(true, false, 2);
#endif
    } else if (split4) {
        if (scr) { if (by_key) BSX_LAUNCH_KEYED(true, true, 4); else BSX_LAUNCH_KEYED(true, false, 4); }
        else     { if (by_key) BSX_LAUNCH_KEYED(false, true, 4); else BSX_LAUNCH_KEYED(false, false, 4); }
    } else {
        if (scr) { if (by_key) BSX_LAUNCH_KEYED(true, true, 1); else BSX_LAUNCH_KEYED(true, false, 1); }
        else     { if (by_key) BSX_LAUNCH_KEYED(false, true, 1); else BSX_LAUNCH_KEYED(false, false, 1); }
    }
#undef BSX_LAUNCH_KEYED
    if (scr) {
        static const long env_k = bsx_knob("BSX_ED_FIN_K", 0);
        const uint32_t K = env_k > 0 ? (uint32_t)env_k : ed_fin_k(n);
        const uint64_t lanes = (n + K - 1) / K;
        hipLaunchKernelGGL(k_ed25519_finish, dim3((uint32_t)((lanes + ED_THREADS - 1) / ED_THREADS)), dim3(ED_THREADS), 0, s, vals, n, ok, scr, K);
    }
    BSX_LAUNCH_DEFERRED();
    return hipGetLastError();
}
hipError_t bsxk_commit_tally(hipStream_t s, const bsx_validator* vals, uint32_t n_commits, uint32_t v_max, const uint8_t* header_hashes,
                             const uint8_t* ok, bsx_commit_result* results, const bsxk_unit_dst* wit) {
    if (!n_commits) return hipSuccess;
    uint32_t P = 1;
    while (P < v_max) P *= 2;
    const bsxk_unit_dst w = wit ? *wit : bsxk_unit_dst{nullptr, 0, 0, 0, 0};
    // up to 128 validator slots and at least four commits: four commits per 256-thread workgroup, the levels of their four trees dealt
    // densely to the lanes (7.5 wave-compressions per commit instead of 16); a commit of <= 128 slots alone is two waves' worth of
    // leaves: a 128-thread workgroup; larger sets: one commit per 256 threads.  The tree is a latency chain either way
    // dynamic LDS: the leaves' buffer + the half-size buffer of the level above, node flags likewise (k_commit_tally)
    auto lds_bytes = [P](uint32_t G) { const uint32_t h = P / 2 ? P / 2 : 1; return (size_t)G * (P * 32 + h * 32 + P + h); };
    static const long env_g = bsx_knob("BSX_TALLY_G", 0);                 // experiments: 4 / 8 forces the form
    // Round 6: from two 4-commit workgroups per compute unit on (2048 x 100: 512 workgroups on 256 CUs) EIGHT commits share a 512-thread
    // workgroup — one per CU, so a CU carries ONE serial tail of narrow levels instead of two that may land on the same SIMD
    const bool g8 = env_g ? env_g == 8 : (P <= 128 && n_commits >= 8u * (uint32_t)bsxk_compute_units());
    if (P <= 128 && n_commits >= 8 && g8) {
        constexpr int G = 8;
        hipLaunchKernelGGL(k_commit_tally<G>, dim3((n_commits + G - 1) / G), dim3(2 * TL_THREADS), lds_bytes(G), s, vals, v_max, n_commits,
                           header_hashes, ok, results, w);
    } else if (P <= 128 && n_commits >= 4) {
        constexpr int G = 4;
        hipLaunchKernelGGL(k_commit_tally<G>, dim3((n_commits + G - 1) / G), dim3(TL_THREADS), lds_bytes(G), s, vals, v_max, n_commits,
                           header_hashes, ok, results, w);
    } else {
        const uint32_t threads = P <= 128 ? 128 : TL_THREADS;
        hipLaunchKernelGGL(k_commit_tally<1>, dim3(n_commits), dim3(threads), lds_bytes(1), s, vals, v_max, n_commits, header_hashes, ok, results, w);
    }
    return hipGetLastError();
}
hipError_t bsxk_skip_check(hipStream_t s, uint32_t n_ranges, uint32_t v_max, const bsx_shared_ctx* ranges, const bsx_header* headers,
                           uint64_t hpr, const uint8_t* hashes, const bsx_validator* target, const bsx_validator* trusted,
                           const uint8_t* target_ok, bsx_commit_result* target_res, const bsx_commit_result* trusted_res,
                           uint32_t* skip_status, uint8_t* target_hashes, const uint32_t* target_idx, const uint8_t* chain_id,
                           uint32_t chain_id_len, const bsxk_unit_dst* wit) {
    if (!n_ranges) return hipSuccess;
    SkipArgs a{n_ranges, v_max, ranges, headers, hpr, hashes, target, trusted, target_ok, target_res, trusted_res, skip_status, target_hashes, target_idx,
               chain_id_len, {0}, wit ? *wit : bsxk_unit_dst{nullptr, 0, 0, 0, 0}};
    for (uint32_t i = 0; i < chain_id_len && i < 50; i++) a.chain_id[i] = chain_id[i];
    hipLaunchKernelGGL(k_skip_check, dim3(n_ranges), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t bsxk_commit_sums(hipStream_t s, const bsx_validator* vals, uint32_t n_commits, uint32_t v_max, const uint8_t* header_hashes, const uint8_t* ok,
                            bsx_commit_result* results, const bsxk_unit_dst* wit) {
    if (!n_commits) return hipSuccess;
    const bsxk_unit_dst w = wit ? *wit : bsxk_unit_dst{nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_commit_sums, dim3(n_commits), dim3(v_max <= 128 ? 128 : TL_THREADS), 0, s, vals, v_max, header_hashes, ok, results, w);
    return hipGetLastError();
}
int bsxk_tally_vmax(void) { return TL_VMAX; }
hipError_t bsxk_skip_eval(hipStream_t s, const bsx_validator* start, const bsx_validator* cand, uint32_t n_cand, uint32_t v_max, bsx_skip_eval* out) {
    if (!n_cand) return hipSuccess;
    hipLaunchKernelGGL(k_skip_eval, dim3(n_cand), dim3(256), 0, s, start, cand, v_max, out);
    return hipGetLastError();
}
}
