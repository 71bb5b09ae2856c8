// manifest.cpp — bsx_witness_manifest: which Goldilocks elements of a witness belong to which circuit variable.
//
// The witness a map job / reduce node emits is OUR documented layout (include/bsx_layout.h) of the variables the cited
// builder.rs lines create; a plonky2x `AsyncHint` / generator shim needs to hand each value to the matching `Variable`
// (DataCommitmentProofVariable / MapReduceSubchainVariable, circuits/vars.rs:13-36, and the intermediate variables of
// circuits/builder.rs:105-271,337-395).  This table is that mapping in machine-readable form: one entry per variable
// group with its element offset, so the shim never hard-codes offsets.  Host-side bookkeeping only (no GPU needed).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"

namespace {
struct Builder {
    std::vector<bsx_manifest_entry> v;
    uint64_t bits0 = 0, words0 = 0, bools0 = 0;   // first element of the three sections
    void add(const char* name, const char* ref, uint32_t kind, uint64_t off, uint64_t per, uint32_t repeat, uint64_t stride) {
        bsx_manifest_entry e;
        memset(&e, 0, sizeof e);
        snprintf(e.name, sizeof e.name, "%s", name);
        snprintf(e.reference, sizeof e.reference, "%s", ref);
        e.kind = kind;
        e.repeat = repeat;
        e.element_offset = off;
        e.elements_per_record = per;
        e.record_stride = repeat > 1 ? stride : per;
        v.push_back(e);
    }
    // byte-section group: `nbytes` bytes at byte offset `boff`, `repeat` records `bstride` bytes apart
    void bytes(const char* name, const char* ref, uint32_t boff, uint32_t nbytes, uint32_t repeat = 1, uint32_t bstride = 0) {
        add(name, ref, BSX_KIND_BYTES, bits0 + 8ull * boff, 8ull * nbytes, repeat, 8ull * bstride);
    }
    void words(const char* name, const char* ref, uint32_t widx, uint32_t n, uint32_t repeat = 1, uint32_t wstride = 0) {
        add(name, ref, BSX_KIND_U32, words0 + widx, n, repeat, wstride);
    }
    void bools(const char* name, const char* ref, uint32_t bidx, uint32_t n, uint32_t repeat = 1, uint32_t bstride = 0) {
        add(name, ref, BSX_KIND_BOOL, bools0 + bidx, n, repeat, bstride);
    }
};
}  // namespace

namespace {
// validator-set tree groups shared by the commit and the skip section
void tree_groups(Builder& m, const char* prefix, uint32_t V, uint32_t off_leaf, uint32_t off_leaf_hash, uint32_t off_inner, uint32_t off_node,
                 uint32_t off_root) {
    const uint32_t P = bsx_pow2_ceil(V);
    char nm[80];
    snprintf(nm, sizeof nm, "%sleaf[] (SimpleValidator bytes, zero padded)", prefix);
    m.bytes(nm, "tendermintx [UPSTREAM]", off_leaf, 48, V, 48);
    snprintf(nm, sizeof nm, "%sleaf_hash[] (P = pow2 >= V; padding = zero validator)", prefix);
    m.bytes(nm, "tendermintx [UPSTREAM]", off_leaf_hash, 32, P, 32);
    if (P > 1) {
        snprintf(nm, sizeof nm, "%stree.inner[] (levels bottom-up)", prefix);
        m.bytes(nm, "tendermintx [UPSTREAM]", off_inner, 32, P - 1, 32);
        snprintf(nm, sizeof nm, "%stree.node[] (select(both enabled, inner, left))", prefix);
        m.bytes(nm, "tendermintx [UPSTREAM]", off_node, 32, P - 1, 32);
    }
    snprintf(nm, sizeof nm, "%svalidators_hash", prefix);
    m.bytes(nm, "tendermintx [UPSTREAM]", off_root, 32);
}
void proof_groups(Builder& m, const char* name, const char* ref, uint32_t off, uint32_t cap) {
    char nm[80];
    snprintf(nm, sizeof nm, "%s.proof (4 aunts)", name);
    m.bytes(nm, ref, off, 128);
    snprintf(nm, sizeof nm, "%s.path (leaf hash, 4 nodes; last = header hash)", name);
    m.bytes(nm, ref, off + 128, 160);
    snprintf(nm, sizeof nm, "%s.leaf (zero padded to the field capacity)", name);
    m.bytes(nm, ref, off + BSX_PROOF_FIXED, cap);
}
}  // namespace

extern "C" int bsx_witness_manifest_section(uint32_t section, uint32_t param, bsx_manifest_entry* entries, uint32_t capacity, uint32_t* out_n) {
    if (!out_n) return BSX_ERR_BAD_ARG;
    Builder m;
    if (section == BSX_SECTION_COMMIT) {          // one commit of V = param validator slots (builder.skip / builder.step inner loop)
        const uint32_t V = param;
        if (V == 0 || V > 512) return BSX_ERR_BAD_ARG;
        const uint32_t P = bsx_pow2_ceil(V);
        const bsx_witness_layout L = bsx_commit_layout(V);
        const char* ref = "header_range.rs:42-48";
        m.words0 = 8ull * L.n_bytes;
        m.bools0 = m.words0 + L.n_words;
        m.bytes("header_hash", ref, bsx_cm_off_header_hash(), 32);
        m.bytes("validator[].sha512_digest (SHA512(R|A|M))", ref, bsx_cm_off_digest(V), 64, V, 64);
        m.bytes("validator[].challenge (digest mod L, little endian)", ref, bsx_cm_off_challenge(V), 32, V, 32);
        tree_groups(m, "", V, bsx_cm_off_leaf(V), bsx_cm_off_leaf_hash(V), bsx_cm_off_inner(V), bsx_cm_off_node(V), bsx_cm_off_root(V));
        m.bytes("validator[].pubkey", ref, bsx_cm_off_validators(V), 32, V, BSX_CM_VAL_BYTES);
        m.bytes("validator[].signature (R | s)", ref, bsx_cm_off_validators(V) + 32, 64, V, BSX_CM_VAL_BYTES);
        m.bytes("validator[].message (CanonicalVote sign bytes, zero padded)", ref, bsx_cm_off_validators(V) + 96, 124, V, BSX_CM_VAL_BYTES);
        m.words("validator[].message_byte_length", ref, 0, 1, V, BSX_CM_SLOT_WORDS);
        m.words("validator[].validator_byte_length", ref, 1, 1, V, BSX_CM_SLOT_WORDS);
        m.words("validator[].voting_power", ref, 2, 2, V, BSX_CM_SLOT_WORDS);
        m.words("total_voting_power", ref, bsx_cm_w_total(V), 2);
        m.words("signed_voting_power", ref, bsx_cm_w_total(V) + 2, 2);
        m.words("trusted_signed_voting_power (present_on_trusted_header)", ref, bsx_cm_w_total(V) + 4, 2);
        static const char* slot_b[BSX_CM_SLOT_BOOLS] = {"validator[].enabled", "validator[].signed", "validator[].present_on_trusted_header",
                                                       "validator[].signature_valid", "validator[].message_has_round",
                                                       "validator[].message_carries_header_hash", "validator[].counted"};
        for (uint32_t k = 0; k < BSX_CM_SLOT_BOOLS; k++) m.bools(slot_b[k], ref, k, 1, V, BSX_CM_SLOT_BOOLS);
        m.bools("leaf_enabled[]", ref, bsx_cm_b_leaf_enabled(V), 1, P, 1);
        if (P > 1) m.bools("node_enabled[]", ref, bsx_cm_b_node_enabled(V), 1, P - 1, 1);
        m.bools("two_thirds_ok (3 * signed > 2 * total)", ref, bsx_cm_b_tail(V), 1);
        m.bools("power_overflow (total > MaxTotalVotingPower)", ref, bsx_cm_b_tail(V) + 1, 1);
        m.bools("signatures_ok", ref, bsx_cm_b_tail(V) + 2, 1);
    } else if (section == BSX_SECTION_SKIP) {     // the rest of CombinedSkipCircuit::define (header_range.rs:32-59)
        const uint32_t V = param;
        if (V == 0 || V > 512) return BSX_ERR_BAD_ARG;
        const uint32_t P = bsx_pow2_ceil(V);
        const bsx_witness_layout L = bsx_skip_layout(V);
        const char* ref = "header_range.rs:42-48";
        m.words0 = 8ull * L.n_bytes;
        m.bools0 = m.words0 + L.n_words;
        m.bytes("trusted_header_hash (public input)", "header_range.rs:34", 0, 32);
        m.bytes("target_header_hash (public output)", "header_range.rs:57", 32, 32);
        m.bytes("data_commitment (public output)", "header_range.rs:58", 64, 32);
        tree_groups(m, "trusted.", V, bsx_sk_off_leaf(V), bsx_sk_off_leaf_hash(V), bsx_sk_off_inner(V), bsx_sk_off_node(V), bsx_sk_off_root(V));
        m.bytes("trusted.validator[].pubkey", ref, bsx_sk_off_pubkeys(V), 32, V, 32);
        static const char* pn[BSX_SK_N_PROOFS] = {"target.chain_id_proof", "target.height_proof", "target.validators_hash_proof",
                                                 "trusted.validators_hash_proof"};
        for (uint32_t k = 0; k < BSX_SK_N_PROOFS; k++) proof_groups(m, pn[k], ref, bsx_sk_off_proof(V, k), bsx_sk_proof_cap(k));
        m.words("trusted_block (public input)", "header_range.rs:33", BSX_SK_W_TRUSTED_BLOCK, 2);
        m.words("target_block (public input)", "header_range.rs:35", BSX_SK_W_TARGET_BLOCK, 2);
        m.words("proof_leaf_byte_length[] (chain_id, height, target vh, trusted vh)", ref, BSX_SK_W_LEAF_LEN, 1, BSX_SK_N_PROOFS, 1);
        m.words("trusted.validator[].validator_byte_length", ref, BSX_SK_W_SLOTS, 1, V, 3);
        m.words("trusted.validator[].voting_power", ref, BSX_SK_W_SLOTS + 1, 2, V, 3);
        m.words("trusted.total_voting_power", ref, bsx_sk_w_total(V), 2);
        m.words("trusted.overlap_voting_power (signed the target)", "fetcher.rs:76-80", bsx_sk_w_total(V) + 2, 2);
        m.bools("trusted.validator[].enabled", ref, 0, 1, V, 2);
        m.bools("trusted.validator[].signed_target", ref, 1, 1, V, 2);
        m.bools("trusted.leaf_enabled[]", ref, bsx_sk_b_leaf_enabled(V), 1, P, 1);
        if (P > 1) m.bools("trusted.node_enabled[]", ref, bsx_sk_b_node_enabled(V), 1, P - 1, 1);
        static const char* ck[BSX_SK_CHECK_BOOLS] = {"trusted_hash_ok", "height_ok", "chain_id_ok", "signatures_ok", "target_validators_hash_ok",
                                                    "trusted_validators_hash_ok", "two_thirds_ok", "one_third_ok", "power_overflow"};
        for (uint32_t k = 0; k < BSX_SK_CHECK_BOOLS; k++) m.bools(ck[k], ref, bsx_sk_b_checks(V) + k, 1);
    } else if (section == BSX_SECTION_STEP) {     // the rest of CombinedStepCircuit::define (next_header.rs:25-46)
        const bsx_witness_layout L = bsx_step_layout();
        const char* ref = "next_header.rs:32-36";
        m.words0 = 8ull * L.n_bytes;
        m.bools0 = m.words0 + L.n_words;
        m.bytes("prev_header_hash (public input)", "next_header.rs:27", 0, 32);
        m.bytes("next_header_hash (public output)", "next_header.rs:45", 32, 32);
        m.bytes("data_commitment (public output)", "next_header.rs:46", 64, 32);
        static const char* pn[BSX_ST_N_PROOFS] = {"next.chain_id_proof", "next.height_proof", "next.validators_hash_proof", "next.last_block_id_proof",
                                                 "prev.next_validators_hash_proof", "data_hash_proofs[0]"};
        for (uint32_t k = 0; k < BSX_ST_N_PROOFS; k++)
            proof_groups(m, pn[k], k == 5 ? "builder.rs:418-433" : ref, bsx_st_off_proof(k), bsx_st_proof_cap(k));
        m.bytes("data_root_tuple", "builder.rs:436-439", bsx_st_off_tuple(), 64);
        m.words("prev_block (public input)", "next_header.rs:26", BSX_ST_W_PREV_BLOCK, 2);
        m.words("next_block", "next_header.rs:29-30", BSX_ST_W_NEXT_BLOCK, 2);
        m.words("proof_leaf_byte_length[]", ref, BSX_ST_W_LEAF_LEN, 1, BSX_ST_N_PROOFS, 1);
        static const char* ck[BSX_ST_CHECK_BOOLS] = {"prev_hash_ok", "height_ok", "chain_id_ok", "signatures_ok", "validators_hash_ok",
                                                    "next_validators_hash_ok", "last_block_id_ok", "two_thirds_ok", "power_overflow",
                                                    "data_hash_root_ok (A10)"};
        for (uint32_t k = 0; k < BSX_ST_CHECK_BOOLS; k++) m.bools(ck[k], k == 9 ? "builder.rs:434" : ref, k, 1);
    } else if (section == BSX_SECTION_REDUCE) {   // one reduce node (circuits/builder.rs:337-395)
        const bsx_witness_layout L = bsx_reduce_layout();
        m.words0 = 8ull * L.n_bytes;
        m.bools0 = m.words0 + L.n_words;
        m.bytes("computed_data_merkle_root", "builder.rs:357-364", 0, 32);
        m.bytes("out.start_header", "builder.rs:390", 32, 32);
        m.bytes("out.end_header", "builder.rs:379-383", 64, 32);
        m.bytes("out.data_merkle_root", "builder.rs:367-371", 96, 32);
        m.words("out.start_block", "builder.rs:389", 0, 2);
        m.words("out.end_block", "builder.rs:374-378", 2, 2);
        m.bools("is_right_subchain_disabled", "builder.rs:344", 0, 1);
        m.bools("subchains_headers_linked", "builder.rs:348-349", 1, 1);
        m.bools("subchains_blocks_linked", "builder.rs:350", 2, 1);
        m.bools("subchains_linked", "builder.rs:351", 3, 1);
        m.bools("link_check", "builder.rs:352", 4, 1);
        m.bools("out.is_enabled", "builder.rs:388", 5, 1);
    } else if (section == BSX_SECTION_MAP) {
        const uint32_t B = param;
        if (!B || (B & (B - 1)) || B > BSX_MAX_BATCH) return BSX_ERR_BAD_ARG;
        const bsx_witness_layout L = bsx_map_layout(B);
        m.words0 = 8ull * L.n_bytes;
        m.bools0 = m.words0 + L.n_words;
        // ---- bytes section (ByteVariable = 8 BoolVariable, MSB first)
        m.bytes("ctx.start_header_hash", "builder.rs:14", bsx_off_ctx_start_header(), 32);
        m.bytes("ctx.end_header_hash", "builder.rs:16", bsx_off_ctx_end_header(), 32);
        m.bytes("data_comm_proof.start_header", "vars.rs:16", bsx_off_start_header(), 32);
        m.bytes("data_comm_proof.end_header", "vars.rs:17", bsx_off_end_header(), 32);
        m.bytes("data_comm_proof.data_hash_proofs[].proof", "vars.rs:18-21", bsx_off_dh_proofs(B), 128, B, BSX_DH_PROOF_SIZE);
        m.bytes("data_comm_proof.data_hash_proofs[].leaf", "vars.rs:18-21", bsx_off_dh_proofs(B) + 128, 34, B, BSX_DH_PROOF_SIZE);
        m.bytes("data_comm_proof.last_block_id_proofs[].proof", "vars.rs:22-25", bsx_off_lb_proofs(B), 128, B, BSX_LB_PROOF_SIZE);
        m.bytes("data_comm_proof.last_block_id_proofs[].leaf", "vars.rs:22-25", bsx_off_lb_proofs(B) + 128, 72, B, BSX_LB_PROOF_SIZE);
        m.bytes("slot[].data_hash_path (leaf hash, 4 nodes; last = data_hash_proof_root)", "builder.rs:189-193", bsx_off_slots(B), 160, B, BSX_SLOT_BYTES);
        m.bytes("slot[].last_block_id_path (last = last_block_id_proof_root)", "builder.rs:195-199", bsx_off_slots(B) + 160, 160, B, BSX_SLOT_BYTES);
        m.bytes("slot[].curr_header", "builder.rs:223", bsx_off_slots(B) + 320, 32, B, BSX_SLOT_BYTES);
        m.bytes("data_root_tuple[]", "builder.rs:137", bsx_off_tuples(B), 64, B, 64);
        m.bytes("leaf_hash[]", "builder.rs:144-147", bsx_off_leaf_hashes(B), 32, B, 32);
        if (B > 1) {
            m.bytes("tree.inner[] (levels bottom-up)", "builder.rs:144-147", bsx_off_inner(B), 32, B - 1, 32);
            m.bytes("tree.node[] (select(both enabled, inner, left); last = data_merkle_root)", "builder.rs:144-147", bsx_off_nodes(B), 32, B - 1, 32);
        }
        m.bytes("record.start_header", "builder.rs:263-270", bsx_off_record(B), 32);
        m.bytes("record.end_header", "builder.rs:263-270", bsx_off_record(B) + 32, 32);
        m.bytes("record.data_merkle_root", "builder.rs:263-270", bsx_off_record(B) + 64, 32);
        // ---- words section (U64Variable = limb 0 (low) then limb 1; builder.rs:124-128)
        m.words("ctx.start_block", "builder.rs:13", BSX_W_CTX_START, 2);
        m.words("ctx.end_block", "builder.rs:15", BSX_W_CTX_END, 2);
        m.words("batch_start_block", "builder.rs:315-316", BSX_W_BATCH_START, 2);
        m.words("batch_end_block", "builder.rs:317-322", BSX_W_BATCH_END, 2);
        m.words("last_block_to_process", "builder.rs:177", BSX_W_LAST_TO_PROCESS, 2);
        m.words("curr_idx[]", "builder.rs:182", BSX_W_CURR_IDX, 2, B, 2);
        m.words("temp_end_block_num", "builder.rs:236", bsx_w_temp_end(B), 2);
        m.words("end_block_num", "builder.rs:241", bsx_w_end_block_num(B), 2);
        m.words("nb_blocks_in_batch", "builder.rs:119", bsx_w_nb_blocks(B), 2);
        m.words("block_height[]", "builder.rs:134", bsx_w_block_height(B), 2, B, 2);
        m.words("record.start_block", "builder.rs:263-270", bsx_w_rec_start(B), 2);
        m.words("record.end_block", "builder.rs:263-270", bsx_w_rec_end(B), 2);
        // ---- bools section
        m.bools("is_batch_enabled", "builder.rs:174", BSX_B_BATCH_ENABLED, 1);
        static const char* slot_b[9][2] = {{"slot[].curr_block_disabled", "builder.rs:184"}, {"slot[].is_last_block", "builder.rs:185"},
                                           {"slot[].is_valid_prev_header", "builder.rs:205"}, {"slot[].prev_header_check", "builder.rs:206"},
                                           {"slot[].is_data_hash_proof_valid", "builder.rs:210"}, {"slot[].data_hash_check", "builder.rs:211"},
                                           {"slot[].root_matches_end_header", "builder.rs:216"}, {"slot[].end_header_check", "builder.rs:218"},
                                           {"slot[].curr_block_enabled", "builder.rs:225"}};
        for (uint32_t k = 0; k < 9; k++) m.bools(slot_b[k][0], slot_b[k][1], BSX_B_SLOTS + k, 1, B, BSX_SLOT_BOOLS);
        static const char* tail_b[6][2] = {{"is_last_block_disabled", "builder.rs:229"}, {"last_block_matches_end_header", "builder.rs:230"},
                                           {"end_header_check", "builder.rs:231"}, {"is_batch_end_lt_global_end", "builder.rs:235"},
                                           {"is_end_block_lt_start", "builder.rs:240"}, {"end_block_gte_start_block", "builder.rs:113"}};
        for (uint32_t k = 0; k < 6; k++) m.bools(tail_b[k][0], tail_b[k][1], bsx_b_tail(B) + k, 1);
        m.bools("leaf_enabled[]", "builder.rs:119-128", bsx_b_leaf_enabled(B), 1, B, 1);
        if (B > 1) m.bools("node_enabled[]", "builder.rs:144-147", bsx_b_node_enabled(B), 1, B - 1, 1);
        m.bools("record.is_enabled", "builder.rs:263-270", bsx_b_rec_enabled(B), 1);
    } else {
        return BSX_ERR_BAD_ARG;
    }
    *out_n = (uint32_t)m.v.size();
    if (entries) {
        if (capacity < m.v.size()) return BSX_ERR_BAD_ARG;
        memcpy(entries, m.v.data(), m.v.size() * sizeof(bsx_manifest_entry));
    }
    return BSX_OK;
}

extern "C" int bsx_witness_manifest(uint32_t batch_size, bsx_manifest_entry* entries, uint32_t capacity, uint32_t* out_n) {
    return bsx_witness_manifest_section(batch_size ? BSX_SECTION_MAP : BSX_SECTION_REDUCE, batch_size, entries, capacity, out_n);
}
