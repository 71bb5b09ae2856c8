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

#endif /* BSX_LAYOUT_H */
