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

extern "C" int bsx_witness_manifest(uint32_t batch_size, bsx_manifest_entry* entries, uint32_t capacity, uint32_t* out_n) {
    if (!out_n) return BSX_ERR_BAD_ARG;
    Builder m;
    if (batch_size == 0) {   // one reduce node (circuits/builder.rs:337-395)
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
    } else {
        const uint32_t B = batch_size;
        if ((B & (B - 1)) || B > BSX_MAX_BATCH) return BSX_ERR_BAD_ARG;
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
    }
    *out_n = (uint32_t)m.v.size();
    if (entries) {
        if (capacity < m.v.size()) return BSX_ERR_BAD_ARG;
        memcpy(entries, m.v.data(), m.v.size() * sizeof(bsx_manifest_entry));
    }
    return BSX_OK;
}
