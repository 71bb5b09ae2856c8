/* oracle/poseidon.c — CPU ORACLE (test infrastructure only; see orc.h).
 *
 * plonky2's Poseidon over Goldilocks (PoseidonGoldilocksConfig — the hash config of every reference binary through
 * plonky2x DefaultParameters, bin/header_range_2048.rs:1-17; used by builder.build()/prove and the mapreduce
 * recursion, circuits/builder.rs:301-302).  The code lives in plonky2 53c5bc3e (Cargo.lock:3110-3112), a crate that
 * is NOT under /root/reference: restated from its public definition, in the plainest possible form (unsigned __int128
 * and % p everywhere, definition-level MDS) so that it shares nothing with the product's limb-split device code.
 *
 * PINNING: the reference tree holds no Poseidon constant or vector ("parity unpinned" by the build rules).  What pins
 * this file instead are plonky2's PUBLIC known answers, reproduced by tests/test_oracle_poseidon.py:
 *   - the round-constant generator of plonky2's generate_constants (ChaCha8Rng::seed_from_u64(0) +
 *     gen_range(0..ORDER)), re-implemented below in C independently of tools/gen_poseidon_constants.py, yields
 *     ALL_ROUND_CONSTANTS[0..3] = b585f766f2144405 7746a55f43921ad7 b2fb0d31cee799b4 0f6760a4803427d7;
 *   - poseidon([0;12]) = 3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 ... and poseidon([0..12)) =
 *     d64e1e3efc5b8e9e 53666633020aaa47 ... (plonky2 poseidon_goldilocks.rs test_vectors).
 */
#include <stdlib.h>
#include <string.h>

#include "orc.h"

typedef unsigned __int128 u128;
#define GL_P 0xFFFFFFFF00000001ull

/* ------------------------------------------------------------------ constants: ChaCha8Rng::seed_from_u64(0) */
static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) \
    a += b; d = rotl32(d ^ a, 16); c += d; b = rotl32(b ^ c, 12); a += b; d = rotl32(d ^ a, 8); c += d; b = rotl32(b ^ c, 7)

static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5],
                      key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
    uint32_t w[16];
    memcpy(w, s, sizeof w);
    for (int r = 0; r < 4; r++) { /* 8 rounds = 4 double rounds */
        QR(w[0], w[4], w[8], w[12]); QR(w[1], w[5], w[9], w[13]); QR(w[2], w[6], w[10], w[14]); QR(w[3], w[7], w[11], w[15]);
        QR(w[0], w[5], w[10], w[15]); QR(w[1], w[6], w[11], w[12]); QR(w[2], w[7], w[8], w[13]); QR(w[3], w[4], w[9], w[14]);
    }
    for (int i = 0; i < 16; i++) out[i] = w[i] + s[i];
}

static uint64_t g_rc[360];
static int g_rc_ready = 0;

static void gen_constants(void) {
    /* rand_core SeedableRng::seed_from_u64: PCG32 expands the u64 into the 32-byte key */
    uint64_t st = 0;
    uint32_t key[8];
    for (int i = 0; i < 8; i++) {
        st = st * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    uint32_t blk[16];
    uint64_t ctr = 0;
    int pos = 16;
    for (int k = 0; k < 360;) {
        if (pos == 16) { chacha8_block(key, ctr++, blk); pos = 0; }
        uint64_t v = (uint64_t)blk[pos] | ((uint64_t)blk[pos + 1] << 32);   /* next_u64: low word first */
        pos += 2;
        /* rand 0.8 UniformInt<u64>::sample_single(0, ORDER): range has no leading zeros, zone = range - 1 */
        u128 m = (u128)v * GL_P;
        if ((uint64_t)m <= GL_P - 1) g_rc[k++] = (uint64_t)(m >> 64);
    }
    g_rc_ready = 1;
}

const uint64_t* orc_poseidon_round_constants(void) {
    if (!g_rc_ready) gen_constants();
    return g_rc;
}

/* ------------------------------------------------------------------ field + permutation, definition level */
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % GL_P); }
static uint64_t pow7(uint64_t x) {
    uint64_t x2 = mulmod(x, x), x4 = mulmod(x2, x2);
    return mulmod(mulmod(x4, x2), x);
}

void orc_poseidon_permute(uint64_t s[12]) {
    static const uint64_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}, D[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t* rc = orc_poseidon_round_constants();
    for (int i = 0; i < 12; i++) s[i] %= GL_P;
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = (uint64_t)(((u128)s[i] + rc[12 * r + i]) % GL_P);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
        } else {
            s[0] = pow7(s[0]);
        }
        uint64_t t[12];
        for (int k = 0; k < 12; k++) {
            u128 acc = (u128)s[k] * D[k];
            for (int i = 0; i < 12; i++) acc += (u128)s[(i + k) % 12] * C[i];
            t[k] = (uint64_t)(acc % GL_P);
        }
        memcpy(s, t, sizeof t);
    }
}

/* hash_n_to_hash_no_pad: overwrite-mode sponge, rate 8 */
void orc_poseidon_hash_no_pad(const uint64_t* in, uint64_t n, uint64_t out[4]) {
    uint64_t s[12] = {0};
    for (uint64_t k = 0; k < n; k += 8) {
        for (int i = 0; i < 8 && k + i < n; i++) s[i] = in[k + i] % GL_P;
        orc_poseidon_permute(s);
    }
    memcpy(out, s, 32);
}

/* PoseidonHash::hash_or_noop */
void orc_poseidon_hash_or_noop(const uint64_t* in, uint64_t n, uint64_t out[4]) {
    if (n <= 4) {
        for (int i = 0; i < 4; i++) out[i] = (uint64_t)i < n ? in[i] % GL_P : 0;
    } else {
        orc_poseidon_hash_no_pad(in, n, out);
    }
}

void orc_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    orc_poseidon_permute(s);
    memcpy(out, s, 32);
}

/* MerkleTree::new(leaves, cap_height) over n_leaves (power of two) rows of leaf_len elements: every level bottom-up
 * into tree[] (n_leaves leaf digests, n_leaves/2 parents, ..., 2^cap_height cap digests: 2*n_leaves - 2^cap_height
 * digests of 4 elements).  rows beyond n_rows and elements beyond n_elements read as zero. */
int orc_poseidon_merkle_tree(const uint64_t* elements, uint64_t n_elements, uint32_t leaf_len, uint32_t n_leaves,
                             uint32_t cap_height, uint64_t* tree) {
    if (!leaf_len || !n_leaves || (n_leaves & (n_leaves - 1)) || (1u << cap_height) > n_leaves) return BSX_ERR_BAD_ARG;
    uint64_t* row = malloc((size_t)leaf_len * 8);
    for (uint32_t j = 0; j < n_leaves; j++) {
        for (uint32_t k = 0; k < leaf_len; k++) {
            uint64_t e = (uint64_t)j * leaf_len + k;
            row[k] = e < n_elements ? elements[e] : 0;
        }
        orc_poseidon_hash_or_noop(row, leaf_len, tree + 4 * (size_t)j);
    }
    free(row);
    uint64_t* level = tree;
    for (uint32_t w = n_leaves; w > (1u << cap_height); w /= 2) {
        uint64_t* up = level + 4 * (size_t)w;
        for (uint32_t t = 0; t < w / 2; t++) orc_poseidon_two_to_one(level + 8 * (size_t)t, level + 8 * (size_t)t + 4, up + 4 * (size_t)t);
        level = up;
    }
    return BSX_OK;
}


/* ------------------------------------------------------------------ cpu_baseline form of the permutation
 * Same function as orc_poseidon_permute, written the way a CPU implementation of plonky2's Poseidon is (no `%`): products and MDS
 * rows accumulate in 128 bits and are folded with 2^64 = 2^32 - 1, 2^96 = -1 (mod p) — one fold per output.  The definition-level
 * version above stays the reference of every parity test; tests/test_oracle_poseidon.py holds the two equal.  Used only by the
 * bench's `fused_commitment.cpu_baseline` leg, so that the CPU figure is not an artefact of 5,000 __umodti3 calls per permutation. */
#define GL_EPS 0xFFFFFFFFull
static inline uint64_t gl_reduce128(u128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & GL_EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;              /* borrow: add p = 2^64 - EPS, i.e. subtract EPS mod 2^64 */
    const uint64_t t1 = hl * GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS;                /* carry: 2^64 = EPS */
    return r;
}
static inline uint64_t gl_mul_f(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }
static inline uint64_t gl_pow7_f(uint64_t x) {
    const uint64_t x2 = gl_mul_f(x, x), x3 = gl_mul_f(x2, x), x4 = gl_mul_f(x2, x2);
    return gl_mul_f(x4, x3);
}
void orc_poseidon_permute_fast(uint64_t s[12]) {
    static const uint64_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint64_t* rc = orc_poseidon_round_constants();
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl_reduce128((u128)s[i] + rc[12 * r + i]);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) s[i] = gl_pow7_f(s[i]);
        } else {
            s[0] = gl_pow7_f(s[0]);
        }
        uint64_t t[12];
        for (int k = 0; k < 12; k++) {
            u128 acc = (u128)s[k] * (k == 0 ? 8u : 0u);
            for (int i = 0; i < 12; i++) acc += (u128)s[(i + k) % 12] * C[i];     /* < 2^64 * 264: no overflow */
            t[k] = gl_reduce128(acc);
        }
        memcpy(s, t, sizeof t);
    }
    for (int i = 0; i < 12; i++) if (s[i] >= GL_P) s[i] -= GL_P;
}

/* Merkle caps of n_jobs compact witnesses (layout L) the way bsx_dev_witness_leaf_hashes + bsx_dev_poseidon_merkle_caps produce
 * them: expand one job at a time, rows of leaf_len elements -> hash_or_noop, tree down to 2^cap_height; jobs dealt to n_threads
 * threads, `reps` passes (task t works on job t % n_jobs).  out_caps: n_jobs x 2^cap_height x 4. */
typedef struct {
    const bsx_witness_layout* L;
    uint32_t n_jobs, leaf_len, n_leaves, cap_height, reps;
    const uint8_t* compact;
    uint64_t* out_caps;
    int n_threads, tid;
} cjob_t;
static void fast_hash_or_noop(const uint64_t* in, uint32_t n, uint64_t out[4]) {
    if (n <= 4) { for (uint32_t i = 0; i < 4; i++) out[i] = i < n ? in[i] % GL_P : 0; return; }
    uint64_t s[12] = {0};
    for (uint32_t k = 0; k < n; k += 8) {
        for (uint32_t i = 0; i < 8 && k + i < n; i++) s[i] = in[k + i];
        orc_poseidon_permute_fast(s);
    }
    memcpy(out, s, 32);
}
#include <pthread.h>
static void* cworker(void* arg) {
    cjob_t* jb = arg;
    const bsx_witness_layout* L = jb->L;
    const uint32_t ncap = 1u << jb->cap_height;
    const size_t nd = 2 * (size_t)jb->n_leaves - ncap;
    uint64_t* wit = malloc((size_t)L->n_elements * 8);
    uint64_t* tree = malloc(nd * 32);
    uint64_t* row = malloc((size_t)jb->leaf_len * 8);
    const uint32_t n_tasks = jb->n_jobs * jb->reps;
    for (uint32_t t = (uint32_t)jb->tid; t < n_tasks; t += (uint32_t)jb->n_threads) {
        const uint32_t j = t % jb->n_jobs;
        orc_expand_witness(L, 1, jb->compact + (size_t)j * L->compact_stride, wit);
        for (uint32_t r = 0; r < jb->n_leaves; r++) {
            for (uint32_t k = 0; k < jb->leaf_len; k++) {
                const uint64_t e = (uint64_t)r * jb->leaf_len + k;
                row[k] = e < L->n_elements ? wit[e] : 0;
            }
            fast_hash_or_noop(row, jb->leaf_len, tree + 4 * (size_t)r);
        }
        uint64_t* level = tree;
        for (uint32_t w = jb->n_leaves; w > ncap; w /= 2) {
            uint64_t* up = level + 4 * (size_t)w;
            for (uint32_t q = 0; q < w / 2; q++) {
                uint64_t s[12] = {0};
                memcpy(s, level + 8 * (size_t)q, 64);
                orc_poseidon_permute_fast(s);
                memcpy(up + 4 * (size_t)q, s, 32);
            }
            level = up;
        }
        if (t < jb->n_jobs) memcpy(jb->out_caps + (size_t)j * ncap * 4, tree + (nd - ncap) * 4, (size_t)ncap * 32);
    }
    free(wit); free(tree); free(row);
    return NULL;
}
int orc_bench_witness_caps(const bsx_witness_layout* L, uint32_t n_jobs, uint32_t reps, const uint8_t* compact, uint32_t leaf_len,
                           uint32_t n_leaves, uint32_t cap_height, int n_threads, uint64_t* out_caps) {
    if (!leaf_len || !n_leaves || (n_leaves & (n_leaves - 1)) || (1u << cap_height) > n_leaves) return BSX_ERR_BAD_ARG;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if (reps < 1) reps = 1;
    cjob_t* jobs = calloc((size_t)n_threads, sizeof *jobs);
    pthread_t* th = calloc((size_t)n_threads, sizeof *th);
    for (int t = 0; t < n_threads; t++) {
        cjob_t j = {L, n_jobs, leaf_len, n_leaves, cap_height, reps, compact, out_caps, n_threads, t};
        jobs[t] = j;
        pthread_create(&th[t], NULL, cworker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return BSX_OK;
}
