/*
 * oracle/orc.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm for the header_range hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (blobstreamx_amd/, libbsx.so) never links, imports or calls it.
 *
 * Pinning status:
 *   PINNED against the reference's own fixtures (tests/golden/mocha4.json, derived from
 *   circuits/fixtures/mocha-4 by tests/golden/gen_golden.py): Tendermint header hash +
 *   inclusion proofs (5 blocks), validators_hash (5), Ed25519 verification of the commit
 *   signatures incl. SHA-512 challenge (10), chain links (4), data commitments (4),
 *   encode_data_root_tuple KAT (circuits/builder.rs:584-605).  Plus spec KATs: FIPS 180-4
 *   SHA-256/512 vectors, RFC 8032 §7.1 Ed25519 vectors, differential tests against hashlib.
 *   PARITY UNPINNED (no golden value exists anywhere in the reference; the arithmetic lives in
 *   un-vendored crates plonky2x v1.0.3 / tendermintx v1.0.0 that cannot be built here): the
 *   Goldilocks witness ordering (our own documented layout), the skip circuit's voting-power
 *   and overlap rules, commits with V > 2, round != 0, absent validators.
 *
 * The reference cannot be compiled in this image (no Rust toolchain, git dependencies not
 * vendored), so there is no oracle/_ref build; see DESIGN.md.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#include "../include/bsx.h"        /* POD layouts only (data formats), no product code */
#include "../include/bsx_layout.h" /* witness section offsets (data format) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashes (FIPS 180-4) */
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
void orc_sha512(const uint8_t* msg, size_t len, uint8_t out[64]);
int orc_sha256_has_shani(void);
void orc_sha256_force_portable(int on);

/* ---- Ed25519 (RFC 8032, cofactorless [s]B == R + [h]A, s < L, canonical encodings) */
void orc_sc_reduce64(const uint8_t in[64], uint8_t out[32]);          /* 512-bit LE mod L */
int orc_ed25519_verify_h(const uint8_t pk[32], const uint8_t sig[64], const uint8_t h[32]);
int orc_ed25519_verify(const uint8_t pk[32], const uint8_t* msg, size_t len, const uint8_t sig[64]);

/* ---- Tendermint simple Merkle tree (RFC 6962 style) */
void orc_leaf_hash(const uint8_t* leaf, size_t len, uint8_t out[32]);
void orc_inner_hash(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]);
/* root over n variable-length items (items[i], lens[i]); n == 0 -> SHA256("") */
void orc_merkle_root(const uint8_t* const* items, const size_t* lens, size_t n, uint8_t out[32]);
/* aunts bottom-up; returns depth */
int orc_merkle_proof(const uint8_t* const* items, const size_t* lens, size_t n, size_t idx, uint8_t aunts[][32]);
/* plonky2x get_root_from_merkle_proof [UPSTREAM, SURVEY Appendix B]; path bits LSB first;
 * path_digests (optional): leaf hash then each level's node = depth+1 digests */
void orc_root_from_proof(const uint8_t* leaf, size_t leaf_len, const uint8_t aunts[][32], const uint8_t* path_bits,
                         int depth, uint8_t out[32], uint8_t path_digests[][32]);

/* ---- header hash / proofs (P5) */
int orc_header_check(const bsx_header* h);
int orc_header_hash(const bsx_header* h, uint8_t out_hash[32], bsx_data_hash_proof* dh, bsx_last_block_id_proof* lb);

/* ---- circuits/builder.rs + circuits/input.rs restatement */
void orc_encode_data_root_tuple(const uint8_t data_hash[32], uint64_t height, uint8_t out[64]);
int orc_get_data_commitment(const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start_block, uint64_t end_block,
                            uint8_t out_root[32], uint32_t* assert_fail);
int orc_data_commitment_inputs(const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                               uint64_t latest_block, uint64_t start_block, uint64_t end_block, uint32_t max_leaves,
                               uint8_t out_start_header[32], uint8_t out_end_header[32], bsx_data_hash_proof* out_dh,
                               bsx_last_block_id_proof* out_lb, uint8_t out_expected[32]);
/* compact (optional): one map-job compact witness (bsx_map_layout(batch).compact_stride bytes) */
int orc_prove_subchain(uint32_t batch_size, const bsx_shared_ctx* range, const uint8_t start_header[32],
                       const uint8_t end_header[32], const bsx_data_hash_proof* dh, const bsx_last_block_id_proof* lb,
                       uint64_t batch_start_block, uint64_t batch_end_block, uint64_t global_end_block,
                       const uint8_t global_end_header_hash[32], bsx_subchain* out, uint8_t* compact);
void orc_reduce_pair(const bsx_subchain* left, const bsx_subchain* right, bsx_subchain* out, uint8_t* compact);
int orc_reduce(const bsx_subchain* records, uint32_t n, bsx_subchain* out, uint8_t* reduce_compact);
int orc_prove_data_commitment(uint32_t nb_map_jobs, uint32_t batch_size, const bsx_shared_ctx* range,
                              const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                              uint64_t latest_block, uint8_t out_commitment[32], bsx_subchain* out_result,
                              bsx_subchain* records, uint8_t* compact /* jobs then reduce nodes */,
                              uint32_t* status);
int orc_prove_next_header_data_commitment(uint64_t prev_block, const uint8_t prev_header_hash[32], uint64_t next_block,
                                          const bsx_header* header, uint64_t latest_block, uint8_t out[32]);
void orc_expand_witness(const bsx_witness_layout* layout, uint32_t n_jobs, const uint8_t* compact, uint64_t* out);

/* ---- commit verification (skip / step inner loop) */
void orc_sha512_challenge(const bsx_validator* v, uint8_t h[32], uint8_t digest[64]);
int orc_validator_leaf(const uint8_t pk[32], uint64_t power, uint8_t out[BSX_VALIDATOR_LEAF_MAX]);
void orc_verify_commit(const bsx_validator* vals, uint32_t v_max, const uint8_t header_hash[32],
                       bsx_commit_result* out, uint8_t* sig_ok);
/* same, also emitting the COMMIT unit's compact witness (bsx_commit_layout(v_max).compact_stride bytes; optional) */
void orc_verify_commit_w(const bsx_validator* vals, uint32_t v_max, const uint8_t header_hash[32],
                         bsx_commit_result* out, uint8_t* sig_ok, uint8_t* compact);
/* mode-S fold of a slice of commit results (checker of bsx_dev_verify_commits' d_fold; include/bsx.h bsx_commit_fold) */
void orc_commit_fold(const bsx_commit_result* res, uint32_t n, uint32_t first_index, bsx_commit_fold* out);
int orc_header_range(uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const bsx_header* headers,
                     uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                     const bsx_validator* target_validators, const bsx_validator* trusted_validators, uint32_t v_max,
                     const uint8_t* chain_id, uint32_t chain_id_len,
                     uint8_t output64[64], bsx_commit_result* out_commit,
                     uint8_t* compact /* optional: J map jobs, J - 1 reduce nodes, the COMMIT unit, the SKIP unit */);

/* CombinedStepCircuit::define (circuits/next_header.rs:25-46); builder.step is [UPSTREAM] (checks restated from
 * SURVEY App. B, same list as include/bsx.h bsx_next_header) */
int orc_next_header(const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                    uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                    uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit);

/* same, also emitting the compact witness: the COMMIT unit then the STEP unit (optional) */
int orc_next_header_w(const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                      uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                      uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit, uint8_t* compact);

/* ---- operator skip-target search (circuits/fetcher.rs:60-87 find_block_to_request).  The loop is
 * the reference's; the predicate is_valid_skip is [UPSTREAM] tendermintx v1.0.0 (not under /root/reference): restated
 * as the > 1/3 trusted-power overlap rule the circuit enforces -> PARITY UNPINNED for the predicate. */
typedef struct orc_skip_eval {
    uint64_t overlap_power, start_total_power, signed_power, target_total_power;
    uint32_t valid, power_overflow;
} orc_skip_eval;
void orc_is_valid_skip(const bsx_validator* start_validators, const bsx_validator* target_validators, uint32_t v_max,
                       orc_skip_eval* out);
int orc_find_block_to_request(uint64_t start_block, uint64_t max_end_block, const bsx_validator* start_validators,
                              uint32_t n_candidates, const uint64_t* candidate_heights,
                              const bsx_validator* candidate_validators, uint32_t v_max, uint64_t* out_block,
                              orc_skip_eval* out_evals /* optional, n_candidates */);

/* ---- Poseidon over Goldilocks (plonky2 PoseidonGoldilocksConfig [UPSTREAM, Cargo.lock:3110-3112]; oracle/poseidon.c).
 * Pinned to plonky2's public known answers only — the reference tree holds none. */
const uint64_t* orc_poseidon_round_constants(void);            /* 360, regenerated (ChaCha8Rng::seed_from_u64(0)) */
void orc_poseidon_permute(uint64_t state[12]);
void orc_poseidon_hash_no_pad(const uint64_t* in, uint64_t n, uint64_t out[4]);
void orc_poseidon_hash_or_noop(const uint64_t* in, uint64_t n, uint64_t out[4]);
void orc_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
int orc_poseidon_merkle_tree(const uint64_t* elements, uint64_t n_elements, uint32_t leaf_len, uint32_t n_leaves,
                             uint32_t cap_height, uint64_t* tree /* (2*n_leaves - 2^cap_height) * 4 */);

/* the same permutation in the form a CPU implementation takes (128-bit accumulation, 2^64 = 2^32 - 1 folds; no `%`): only for the
 * bench's Poseidon cpu_baseline; held equal to orc_poseidon_permute by tests/test_oracle_poseidon.py */
void orc_poseidon_permute_fast(uint64_t state[12]);
/* cpu_baseline of the witness commitment: Merkle caps of n_jobs compact witnesses on n_threads threads (expand, hash rows, tree) */
int orc_bench_witness_caps(const bsx_witness_layout* L, uint32_t n_jobs, uint32_t reps, const uint8_t* compact, uint32_t leaf_len,
                           uint32_t n_leaves, uint32_t cap_height, int n_threads, uint64_t* out_caps);

/* ---- batch drivers for the cpu_baseline leg (pthread pool, n_threads >= 1) */
int orc_bench_header_range(uint32_t n_ranges, uint32_t reps, uint32_t nb_map_jobs, uint32_t batch_size,
                           const bsx_shared_ctx* ranges, const bsx_header* headers, uint64_t headers_per_range,
                           const uint64_t* latest_block, const bsx_validator* target_validators,
                           const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id,
                           uint32_t chain_id_len, int with_witness, int n_threads, uint8_t* out64, uint64_t* checksum);

int orc_bench_verify_commits(uint32_t n_commits, uint32_t reps, uint32_t v_max, const bsx_validator* validators,
                             const uint8_t* header_hashes, int n_threads, bsx_commit_result* out_results, uint8_t* out_sig_ok);

#ifdef __cplusplus
}
#endif
#endif
