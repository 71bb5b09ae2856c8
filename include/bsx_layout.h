/*
 * bsx_layout.h — byte offsets of the COMPACT witness of one map job / one reduce node.
 *
 * Data-format header (no code paths): shared by the HIP kernels, the host library and the
 * test oracle, like the POD structs in bsx.h.  Usable from C, C++ and HIP device code.
 *
 * The reference materialises every circuit variable as Goldilocks field elements
 * [UPSTREAM plonky2x v1.0.3 vars]: ByteVariable = 8 BoolVariable MSB-first, Bytes32Variable =
 * 32 ByteVariable, U64Variable = two u32 limbs (limb 0 least significant, cf.
 * circuits/builder.rs:124-128), BoolVariable = 0/1.  The target indices plonky2x assigns are not
 * knowable without building the un-vendored crate, so the witness here is OUR documented layout:
 * the variables the cited builder lines create, grouped by type into three dense sections per
 * map job, in the order below.  expand(compact) = [bits of `bytes`, MSB first] ‖ words ‖ bools.
 *
 * Map job (prove_subchain<B>, circuits/builder.rs:150-271 + get_data_commitment :105-148):
 *  bytes section
 *    0     ctx.start_header_hash[32]                          builder.rs:14
 *    32    ctx.end_header_hash[32]                            builder.rs:16
 *    64    data_comm_proof.start_header[32]                   vars.rs:16
 *    96    data_comm_proof.end_header[32]                     vars.rs:17
 *    128   data_hash_proofs[B]      {aunts[4][32], leaf[34]}  vars.rs:18-21  (162 B each)
 *    ..    last_block_id_proofs[B]  {aunts[4][32], leaf[72]}  vars.rs:22-25  (200 B each)
 *    ..    slot[B] { dh_path[5][32]   leaf hash, then the 4 inner nodes; last = data_hash_proof_root   :189-193
 *                    lb_path[5][32]   same for last_block_id_proof_root                                :195-199
 *                    curr_header[32]  value after the select at                                        :223 }
 *    ..    tuple[B][64]             encode_data_root_tuple                                             :137
 *    ..    leaf_hash[B][32]         compute_root_from_leaves: leaf hashes   [UPSTREAM, SURVEY App. B]
 *    ..    inner[B-1][32]           inner_hash of every pair, level by level (B/2 nodes, B/4, ...)
 *    ..    node[B-1][32]            select(both_enabled, inner, left), same order; last = data_merkle_root
 *    ..    record.start_header[32], record.end_header[32], record.data_merkle_root[32]               :263-270
 *  words section (u32; a U64Variable is lo then hi)
 *    ctx.start_block, ctx.end_block, batch_start_block, batch_end_block, last_block_to_process (:177),
 *    curr_idx[B] (:182), temp_end_block_num (:236), end_block_num (:241), nb_blocks_in_batch (:119),
 *    block_height[B] (:134), record.start_block, record.end_block
 *  bools section (u8 0/1)
 *    is_batch_enabled (:174), slot[B]{curr_block_disabled :184, is_last_block :185,
 *    is_valid_prev_header :205, prev_header_check :206, is_data_hash_proof_valid :210, data_hash_check :211,
 *    root_matches_end_header :216, end_header_check :218, curr_block_enabled after :225},
 *    is_last_block_disabled :229, last_block_matches_end_header :230, end_header_check :231,
 *    is_batch_end_lt_global_end :235, is_end_block_lt_start :240, end_block_gte_start_block :113,
 *    leaf_enabled[B], node_enabled[B-1], record.is_enabled
 *
 * Reduce node (circuits/builder.rs:337-395):
 *  bytes: computed_data_merkle_root[32] (:364), out.start_header[32], out.end_header[32], out.data_merkle_root[32]
 *  words: out.start_block, out.end_block
 *  bools: is_right_subchain_disabled :344, subchains_headers_linked :348-349, subchains_blocks_linked :350,
 *         subchains_linked :351, link_check :352, out.is_enabled
 */
#ifndef BSX_LAYOUT_H
#define BSX_LAYOUT_H

#include <stdint.h>

#include "bsx.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define BSX_HD __host__ __device__ static inline
#else
#define BSX_HD static inline
#endif

#define BSX_DH_PROOF_SIZE 162u
#define BSX_LB_PROOF_SIZE 200u
#define BSX_SLOT_BYTES 352u       /* dh_path 160 + lb_path 160 + curr_header 32 */
#define BSX_SLOT_BOOLS 9u
#define BSX_MAP_FIXED_WORDS 20u   /* 10 U64Variables */

/* byte-section offsets */
BSX_HD uint32_t bsx_off_ctx_start_header(void) { return 0; }
BSX_HD uint32_t bsx_off_ctx_end_header(void) { return 32; }
BSX_HD uint32_t bsx_off_start_header(void) { return 64; }
BSX_HD uint32_t bsx_off_end_header(void) { return 96; }
BSX_HD uint32_t bsx_off_dh_proofs(uint32_t B) { (void)B; return 128; }
BSX_HD uint32_t bsx_off_lb_proofs(uint32_t B) { return 128 + BSX_DH_PROOF_SIZE * B; }
BSX_HD uint32_t bsx_off_slots(uint32_t B) { return 128 + (BSX_DH_PROOF_SIZE + BSX_LB_PROOF_SIZE) * B; }
BSX_HD uint32_t bsx_off_tuples(uint32_t B) { return bsx_off_slots(B) + BSX_SLOT_BYTES * B; }
BSX_HD uint32_t bsx_off_leaf_hashes(uint32_t B) { return bsx_off_tuples(B) + 64 * B; }
BSX_HD uint32_t bsx_off_inner(uint32_t B) { return bsx_off_leaf_hashes(B) + 32 * B; }
BSX_HD uint32_t bsx_off_nodes(uint32_t B) { return bsx_off_inner(B) + 32 * (B - 1); }
BSX_HD uint32_t bsx_off_record(uint32_t B) { return bsx_off_nodes(B) + 32 * (B - 1); }
BSX_HD uint32_t bsx_map_n_bytes(uint32_t B) { return bsx_off_record(B) + 96; }

/* word-section indices (in u32 units) */
#define BSX_W_CTX_START 0u
#define BSX_W_CTX_END 2u
#define BSX_W_BATCH_START 4u
#define BSX_W_BATCH_END 6u
#define BSX_W_LAST_TO_PROCESS 8u
#define BSX_W_CURR_IDX 10u /* [B] */
BSX_HD uint32_t bsx_w_temp_end(uint32_t B) { return 10 + 2 * B; }
BSX_HD uint32_t bsx_w_end_block_num(uint32_t B) { return 12 + 2 * B; }
BSX_HD uint32_t bsx_w_nb_blocks(uint32_t B) { return 14 + 2 * B; }
BSX_HD uint32_t bsx_w_block_height(uint32_t B) { return 16 + 2 * B; } /* [B] */
BSX_HD uint32_t bsx_w_rec_start(uint32_t B) { return 16 + 4 * B; }
BSX_HD uint32_t bsx_w_rec_end(uint32_t B) { return 18 + 4 * B; }
BSX_HD uint32_t bsx_map_n_words(uint32_t B) { return BSX_MAP_FIXED_WORDS + 4 * B; }

/* bool-section indices */
#define BSX_B_BATCH_ENABLED 0u
#define BSX_B_SLOTS 1u /* [B][9] */
BSX_HD uint32_t bsx_b_tail(uint32_t B) { return 1 + BSX_SLOT_BOOLS * B; }            /* 6 bools */
BSX_HD uint32_t bsx_b_leaf_enabled(uint32_t B) { return bsx_b_tail(B) + 6; }         /* [B] */
BSX_HD uint32_t bsx_b_node_enabled(uint32_t B) { return bsx_b_leaf_enabled(B) + B; } /* [B-1] */
BSX_HD uint32_t bsx_b_rec_enabled(uint32_t B) { return bsx_b_node_enabled(B) + (B - 1); }
BSX_HD uint32_t bsx_map_n_bools(uint32_t B) { return bsx_b_rec_enabled(B) + 1; }

BSX_HD uint32_t bsx_align16(uint32_t x) { return (x + 15u) & ~15u; }

BSX_HD bsx_witness_layout bsx_make_layout(uint32_t batch, uint32_t n_bytes, uint32_t n_words, uint32_t n_bools) {
    bsx_witness_layout l;
    l.batch_size = batch;
    l.n_bytes = n_bytes;
    l.n_words = n_words;
    l.n_bools = n_bools;
    l.off_words = bsx_align16(n_bytes);
    l.off_bools = l.off_words + bsx_align16(4 * n_words);
    l.compact_stride = l.off_bools + bsx_align16(n_bools);
    l._pad = 0;
    l.n_elements = 8ull * n_bytes + n_words + n_bools;
    return l;
}
BSX_HD bsx_witness_layout bsx_map_layout(uint32_t B) {
    return bsx_make_layout(B, bsx_map_n_bytes(B), bsx_map_n_words(B), bsx_map_n_bools(B));
}

/* reduce node */
#define BSX_RED_N_BYTES 128u
#define BSX_RED_N_WORDS 4u
#define BSX_RED_N_BOOLS 6u
BSX_HD bsx_witness_layout bsx_reduce_layout(void) {
    return bsx_make_layout(0, BSX_RED_N_BYTES, BSX_RED_N_WORDS, BSX_RED_N_BOOLS);
}

/* ====================================================================================================================
 * Commit / skip / step sections (round 4): the variables of builder.skip / builder.step — circuits/header_range.rs:42-48,
 * circuits/next_header.rs:32-36; the circuit bodies are [UPSTREAM] tendermintx v1.0.0 (verify_skip / verify_step; SURVEY
 * App. B lists the hint's outputs: validators {pubkey, signature, message, message_byte_length, voting_power,
 * validator_byte_length, enabled, signed, present_on_trusted_header}, header inclusion proofs for chain id / height /
 * validators_hash, the trusted validator-hash fields) — as the same three dense sections (bytes, u32 words, bools) per
 * unit, so that k_expand_witness / the fused Poseidon leaf kernel take them unchanged.  As with the map job the target
 * indices plonky2x would assign are unknowable here: the ORDER is this documented layout, the VALUES are what the
 * restated rules (oracle/commit.c) compute.  P = V rounded up to a power of two (padding leaves = the all-zero validator).
 *
 * COMMIT unit = verification of ONE commit of V validator slots against a header hash (P6-P9): bsx_commit_layout(V)
 *  bytes
 *    header_hash[32]                 the hash every signed message must carry
 *    sha512_digest[V][64]            SHA512(R ‖ A ‖ M)                                              (P6)
 *    challenge[V][32]                digest mod L, little endian                                    (P6)
 *    leaf[V][48]                     SimpleValidator bytes 0a 22 0a 20 pk [10 varint(power)], zero padded   (P8)
 *    leaf_hash[P][32]  inner[P-1][32]  node[P-1][32] (select(both enabled, inner, left); levels bottom-up)
 *    validators_hash[32]
 *    validator[V] {pubkey[32], signature[64] (R ‖ s), message[124]}                                 (hint output)
 *  words: slot[V]{message_byte_length, validator_byte_length, voting_power lo, hi}, total_power, signed_power,
 *         trusted_signed_power (U64: lo, hi)
 *  bools: slot[V]{enabled, signed, present_on_trusted_header, signature_valid (P7), message_has_round,
 *         message_carries_header_hash, counted = enabled & signed & signature_valid & message ok}, leaf_enabled[P],
 *         node_enabled[P-1], two_thirds_ok (P9), power_overflow, signatures_ok (no signed validator failed P7 / the message check)
 *
 * Header-field proof record (tendermintx *ProofVariable): aunts[4][32], path[5][32] (leaf hash, then the 4 nodes of
 * get_root_from_merkle_proof; last = the header hash), leaf[cap] (zero padded to the field's bsx_header capacity).
 *
 * SKIP unit = the rest of CombinedSkipCircuit::define per range: bsx_skip_layout(V)
 *  bytes
 *    trusted_header_hash[32] (public input, header_range.rs:34)  target_header_hash[32] (:57)  data_commitment[32] (:58)
 *    trusted_leaf[V][48]  trusted_leaf_hash[P][32]  trusted_inner[P-1][32]  trusted_node[P-1][32]  trusted_validators_hash[32]
 *    trusted_pubkey[V][32]
 *    proof[4]: target chain_id (field 1, cap 52), target height (2, cap 12), target validators_hash (7, cap 36),
 *              trusted validators_hash (7, cap 36)
 *  words: trusted_block, target_block (U64, :33,:35), proof leaf lengths[4], trusted slot[V]{validator_byte_length,
 *         voting_power lo, hi}, trusted_total_power, trusted_overlap_power (U64)
 *  bools: trusted slot[V]{enabled, signed_target (its key validly signed the target commit)}, trusted_leaf_enabled[P],
 *         trusted_node_enabled[P-1], checks: trusted_hash_ok, height_ok, chain_id_ok, signatures_ok, target_validators_hash_ok,
 *         trusted_validators_hash_ok, two_thirds_ok, one_third_ok, power_overflow
 *
 * STEP unit = the rest of CombinedStepCircuit::define (next_header.rs:25-46) incl. prove_next_header_data_commitment
 * (builder.rs:411-443): bsx_step_layout()
 *  bytes
 *    prev_header_hash[32] (public input, next_header.rs:27)  next_header_hash[32] (:45)  data_commitment[32] (:46)
 *    proof[6]: next chain_id (1), next height (2), next validators_hash (7), next last_block_id (4, cap 76),
 *              prev next_validators_hash (8), prev data_hash (6: data_hash_proofs[0] of the MAX_LEAVES = 1 hint, builder.rs:418-433;
 *              all zero when the hint clamps it away, input.rs:160-172,220-239)
 *    data_root_tuple[64] (builder.rs:436-439; its leaf hash is data_commitment, :442)
 *  words: prev_block, next_block (U64), proof leaf lengths[6]
 *  bools: prev_hash_ok, height_ok, chain_id_ok, signatures_ok, validators_hash_ok, next_validators_hash_ok, last_block_id_ok,
 *         two_thirds_ok, power_overflow, data_hash_root_ok (A10, builder.rs:434)
 */
BSX_HD uint32_t bsx_pow2_ceil(uint32_t v) { uint32_t p = 1; while (p < v) p *= 2; return p; }

#define BSX_CM_VAL_BYTES 220u
#define BSX_CM_SLOT_WORDS 4u
#define BSX_CM_SLOT_BOOLS 7u
#define BSX_CM_TAIL_BOOLS 3u
BSX_HD uint32_t bsx_cm_off_header_hash(void) { return 0; }
BSX_HD uint32_t bsx_cm_off_digest(uint32_t V) { (void)V; return 32; }
BSX_HD uint32_t bsx_cm_off_challenge(uint32_t V) { return 32 + 64 * V; }
BSX_HD uint32_t bsx_cm_off_leaf(uint32_t V) { return 32 + 96 * V; }
BSX_HD uint32_t bsx_cm_off_leaf_hash(uint32_t V) { return 32 + 144 * V; }
BSX_HD uint32_t bsx_cm_off_inner(uint32_t V) { return bsx_cm_off_leaf_hash(V) + 32 * bsx_pow2_ceil(V); }
BSX_HD uint32_t bsx_cm_off_node(uint32_t V) { return bsx_cm_off_inner(V) + 32 * (bsx_pow2_ceil(V) - 1); }
BSX_HD uint32_t bsx_cm_off_root(uint32_t V) { return bsx_cm_off_node(V) + 32 * (bsx_pow2_ceil(V) - 1); }
BSX_HD uint32_t bsx_cm_off_validators(uint32_t V) { return bsx_cm_off_root(V) + 32; }
BSX_HD uint32_t bsx_cm_n_bytes(uint32_t V) { return bsx_cm_off_validators(V) + BSX_CM_VAL_BYTES * V; }
BSX_HD uint32_t bsx_cm_w_total(uint32_t V) { return BSX_CM_SLOT_WORDS * V; }         /* then signed (+2), trusted_signed (+4) */
BSX_HD uint32_t bsx_cm_n_words(uint32_t V) { return BSX_CM_SLOT_WORDS * V + 6; }
BSX_HD uint32_t bsx_cm_b_leaf_enabled(uint32_t V) { return BSX_CM_SLOT_BOOLS * V; }
BSX_HD uint32_t bsx_cm_b_node_enabled(uint32_t V) { return BSX_CM_SLOT_BOOLS * V + bsx_pow2_ceil(V); }
BSX_HD uint32_t bsx_cm_b_tail(uint32_t V) { return BSX_CM_SLOT_BOOLS * V + 2 * bsx_pow2_ceil(V) - 1; }
BSX_HD uint32_t bsx_cm_n_bools(uint32_t V) { return bsx_cm_b_tail(V) + BSX_CM_TAIL_BOOLS; }
BSX_HD bsx_witness_layout bsx_commit_layout(uint32_t V) {
    return bsx_make_layout(V, bsx_cm_n_bytes(V), bsx_cm_n_words(V), bsx_cm_n_bools(V));   /* batch_size field = V */
}

/* header-field proof record: aunts 0, path 128, leaf 288 */
#define BSX_PROOF_FIXED 288u
BSX_HD uint32_t bsx_proof_bytes(uint32_t cap) { return BSX_PROOF_FIXED + cap; }

#define BSX_SK_N_PROOFS 4u
#define BSX_SK_CHECK_BOOLS 9u
BSX_HD uint32_t bsx_sk_proof_cap(uint32_t k) { return k == 0 ? 52u : k == 1 ? 12u : 36u; }
BSX_HD uint32_t bsx_sk_off_leaf(uint32_t V) { (void)V; return 96; }
BSX_HD uint32_t bsx_sk_off_leaf_hash(uint32_t V) { return bsx_sk_off_leaf(V) + 48 * V; }
BSX_HD uint32_t bsx_sk_off_inner(uint32_t V) { return bsx_sk_off_leaf_hash(V) + 32 * bsx_pow2_ceil(V); }
BSX_HD uint32_t bsx_sk_off_node(uint32_t V) { return bsx_sk_off_inner(V) + 32 * (bsx_pow2_ceil(V) - 1); }
BSX_HD uint32_t bsx_sk_off_root(uint32_t V) { return bsx_sk_off_node(V) + 32 * (bsx_pow2_ceil(V) - 1); }
BSX_HD uint32_t bsx_sk_off_pubkeys(uint32_t V) { return bsx_sk_off_root(V) + 32; }
BSX_HD uint32_t bsx_sk_off_proof(uint32_t V, uint32_t k) {
    uint32_t o = bsx_sk_off_pubkeys(V) + 32 * V;
    for (uint32_t i = 0; i < k; i++) o += bsx_proof_bytes(bsx_sk_proof_cap(i));
    return o;
}
BSX_HD uint32_t bsx_sk_n_bytes(uint32_t V) { return bsx_sk_off_proof(V, BSX_SK_N_PROOFS); }
#define BSX_SK_W_TRUSTED_BLOCK 0u
#define BSX_SK_W_TARGET_BLOCK 2u
#define BSX_SK_W_LEAF_LEN 4u   /* [4] */
#define BSX_SK_W_SLOTS 8u      /* [V][3] */
BSX_HD uint32_t bsx_sk_w_total(uint32_t V) { return 8 + 3 * V; }                     /* then overlap (+2) */
BSX_HD uint32_t bsx_sk_n_words(uint32_t V) { return 12 + 3 * V; }
BSX_HD uint32_t bsx_sk_b_leaf_enabled(uint32_t V) { return 2 * V; }
BSX_HD uint32_t bsx_sk_b_node_enabled(uint32_t V) { return 2 * V + bsx_pow2_ceil(V); }
BSX_HD uint32_t bsx_sk_b_checks(uint32_t V) { return 2 * V + 2 * bsx_pow2_ceil(V) - 1; }
BSX_HD uint32_t bsx_sk_n_bools(uint32_t V) { return bsx_sk_b_checks(V) + BSX_SK_CHECK_BOOLS; }
BSX_HD bsx_witness_layout bsx_skip_layout(uint32_t V) {
    return bsx_make_layout(V, bsx_sk_n_bytes(V), bsx_sk_n_words(V), bsx_sk_n_bools(V));
}

#define BSX_ST_N_PROOFS 6u
#define BSX_ST_CHECK_BOOLS 10u
BSX_HD uint32_t bsx_st_proof_cap(uint32_t k) { return k == 0 ? 52u : k == 1 ? 12u : k == 3 ? 76u : 36u; }
BSX_HD uint32_t bsx_st_off_proof(uint32_t k) {
    uint32_t o = 96;
    for (uint32_t i = 0; i < k; i++) o += bsx_proof_bytes(bsx_st_proof_cap(i));
    return o;
}
BSX_HD uint32_t bsx_st_off_tuple(void) { return bsx_st_off_proof(BSX_ST_N_PROOFS); }
BSX_HD uint32_t bsx_st_n_bytes(void) { return bsx_st_off_tuple() + 64; }
#define BSX_ST_W_PREV_BLOCK 0u
#define BSX_ST_W_NEXT_BLOCK 2u
#define BSX_ST_W_LEAF_LEN 4u   /* [6] */
#define BSX_ST_N_WORDS 10u
BSX_HD bsx_witness_layout bsx_step_layout(void) {
    return bsx_make_layout(0, bsx_st_n_bytes(), BSX_ST_N_WORDS, BSX_ST_CHECK_BOOLS);
}

#endif /* BSX_LAYOUT_H */
