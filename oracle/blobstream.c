/* oracle/blobstream.c — CPU ORACLE (test infrastructure only; see orc.h).
 * Line-by-line restatement of the reference's header_range hot path:
 *   circuits/builder.rs:82-103   encode_data_root_tuple
 *   circuits/builder.rs:105-148  get_data_commitment  (+ plonky2x compute_root_from_leaves [UPSTREAM, SURVEY App. B])
 *   circuits/builder.rs:150-271  prove_subchain
 *   circuits/builder.rs:273-409  prove_data_commitment (range check, map, reduce, final asserts)
 *   circuits/builder.rs:411-443  prove_next_header_data_commitment
 *   circuits/input.rs:149-271    get_data_commitment_inputs (the hint body, data_commitment.rs:22-44)
 *   circuits/header_range.rs:32-59 CombinedSkipCircuit::define
 * U64Variable arithmetic is restated as wrapping u64 ([UPSTREAM] plonky2x add/sub gadgets). */
#include <stdlib.h>
#include <string.h>

#include "orc.h"

static const uint8_t DATA_HASH_PATH[4] = {0, 1, 1, 0};     /* circuits/builder.rs:166-167 */
static const uint8_t LAST_BLOCK_ID_PATH[4] = {0, 0, 1, 0}; /* circuits/builder.rs:168-169 */

static void put_u64_words(uint32_t* w, uint64_t v) { w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }

/* circuits/builder.rs:82-103 */
void orc_encode_data_root_tuple(const uint8_t data_hash[32], uint64_t height, uint8_t out[64]) {
    memset(out, 0, 24);                                              /* :93-96 */
    for (int i = 0; i < 8; i++) out[24 + i] = (uint8_t)(height >> (56 - 8 * i)); /* :90,97 U64 EVM encode = big endian */
    memcpy(out + 32, data_hash, 32);                                 /* :98 */
}

/* get_data_commitment<MAX_LEAVES>: data hash i is at data_hashes + i*stride. */
static void data_commitment_core(uint32_t B, const uint8_t* data_hashes, size_t stride, uint64_t start_block,
                                 uint64_t end_block, uint8_t out_root[32], uint32_t* assert_fail, uint8_t* cw) {
    bsx_witness_layout L = bsx_map_layout(B);
    uint8_t* Y = cw;
    uint32_t* W = cw ? (uint32_t*)(cw + L.off_words) : NULL;
    uint8_t* Bo = cw ? cw + L.off_bools : NULL;
    int gte = end_block >= start_block; /* :113 */
    if (!gte) *assert_fail |= BSX_A1_END_GTE_START;
    uint64_t nb = end_block - start_block; /* :119 */
    if ((nb >> 32) != 0) *assert_fail |= BSX_A2_NB_BLOCKS_U32; /* :128 */
    uint32_t nb_enabled = (uint32_t)nb;                         /* :124 limbs[0] */
    if (cw) {
        Bo[bsx_b_tail(B) + 5] = (uint8_t)gte;
        put_u64_words(W + bsx_w_nb_blocks(B), nb);
    }
    uint8_t(*nodes)[32] = malloc((size_t)B * 32);
    uint8_t* en = malloc(B);
    for (uint32_t i = 0; i < B; i++) {
        uint8_t tuple[64];
        uint64_t height = start_block + i; /* :134 */
        orc_encode_data_root_tuple(data_hashes + i * stride, height, tuple); /* :137 */
        orc_leaf_hash(tuple, 64, nodes[i]);
        en[i] = i < nb_enabled;
        if (cw) {
            put_u64_words(W + bsx_w_block_height(B) + 2 * i, height);
            memcpy(Y + bsx_off_tuples(B) + 64 * i, tuple, 64);
            memcpy(Y + bsx_off_leaf_hashes(B) + 32 * i, nodes[i], 32);
            Bo[bsx_b_leaf_enabled(B) + i] = en[i];
        }
    }
    /* compute_root_from_leaves [UPSTREAM]: per pair inner always computed; node = both ? inner : left */
    uint32_t k = 0;
    for (uint32_t n = B; n > 1; n /= 2) {
        for (uint32_t i = 0; i < n; i += 2, k++) {
            uint8_t inner[32];
            orc_inner_hash(nodes[i], nodes[i + 1], inner);
            int both = en[i] && en[i + 1];
            int any = en[i] || en[i + 1];
            uint8_t sel[32];
            memcpy(sel, both ? inner : nodes[i], 32);
            if (cw) {
                memcpy(Y + bsx_off_inner(B) + 32 * k, inner, 32);
                memcpy(Y + bsx_off_nodes(B) + 32 * k, sel, 32);
                Bo[bsx_b_node_enabled(B) + k] = (uint8_t)any;
            }
            memcpy(nodes[i / 2], sel, 32);
            en[i / 2] = (uint8_t)any;
        }
    }
    memcpy(out_root, nodes[0], 32);
    free(nodes);
    free(en);
}

int orc_get_data_commitment(const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start_block, uint64_t end_block,
                            uint8_t out_root[32], uint32_t* assert_fail) {
    if (!max_leaves || (max_leaves & (max_leaves - 1))) return BSX_ERR_BAD_ARG;
    uint32_t af = 0;
    data_commitment_core(max_leaves, data_hashes, 32, start_block, end_block, out_root, &af, NULL);
    if (assert_fail) *assert_fail = af;
    return af ? BSX_ERR_ASSERT : BSX_OK;
}

/* circuits/input.rs:149-271 */
int orc_data_commitment_inputs(const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                               uint64_t latest_block, uint64_t start_block, uint64_t end_block, uint32_t max_leaves,
                               uint8_t out_start_header[32], uint8_t out_end_header[32], bsx_data_hash_proof* out_dh,
                               bsx_last_block_id_proof* out_lb, uint8_t out_expected[32]) {
    if (end_block - start_block > (uint64_t)max_leaves) return BSX_ERR_RANGE_TOO_LONG; /* :154 */
    if (latest_block < 2) return BSX_ERR_BAD_ARG;
    uint64_t latest_safe = latest_block - 2;                                          /* :160-161 */
    uint64_t req_end = end_block < latest_safe ? end_block : latest_safe;             /* :162 */
    uint32_t n_dh = 0, n_lb = 0;
    if (start_block <= req_end) {
        if (start_block < first_height || req_end - first_height >= n_headers) return BSX_ERR_BAD_ARG;
        for (uint64_t i = start_block; i <= req_end; i++) {                           /* :167 */
            const bsx_header* h = &headers[i - first_height];
            int rc;
            if (i < req_end) {                                                        /* :172 */
                if ((rc = orc_header_hash(h, NULL, &out_dh[n_dh], NULL))) return rc;
                n_dh++;
            }
            if (i > start_block) {                                                    /* :187 */
                if ((rc = orc_header_hash(h, NULL, NULL, &out_lb[n_lb]))) return rc;
                n_lb++;
            }
        }
    }
    for (uint32_t i = n_dh; i < max_leaves; i++) memset(&out_dh[i], 0, sizeof out_dh[i]); /* :220-239 */
    for (uint32_t i = n_lb; i < max_leaves; i++) memset(&out_lb[i], 0, sizeof out_lb[i]);
    /* :241-244 expected_data_commitment — the node's answer; :70-72 zero when the range is empty.  Restated as the
     * RFC 6962 root over the tuples of [start, req_end) (verified against the 4 fixture answers). */
    if (out_expected) {
        if (req_end <= start_block) {
            memset(out_expected, 0, 32);
        } else {
            size_t n = (size_t)(req_end - start_block);
            uint8_t* tuples = malloc(n * 64);
            const uint8_t** items = malloc(n * sizeof *items);
            size_t* lens = malloc(n * sizeof *lens);
            for (size_t i = 0; i < n; i++) {
                orc_encode_data_root_tuple(out_dh[i].leaf + 2, start_block + i, tuples + 64 * i);
                items[i] = tuples + 64 * i;
                lens[i] = 64;
            }
            orc_merkle_root(items, lens, n, out_expected);
            free(tuples); free(items); free(lens);
        }
    }
    memset(out_start_header, 0, 32);                                                  /* :246-247 */
    memset(out_end_header, 0, 32);
    if (start_block < req_end) {                                                      /* :249 */
        int rc;
        if ((rc = orc_header_hash(&headers[start_block - first_height], out_start_header, NULL, NULL))) return rc;
        if ((rc = orc_header_hash(&headers[req_end - first_height], out_end_header, NULL, NULL))) return rc;
    }
    return BSX_OK;
}

/* circuits/builder.rs:150-271 */
int orc_prove_subchain(uint32_t B, const bsx_shared_ctx* range, const uint8_t start_header[32],
                       const uint8_t end_header[32], const bsx_data_hash_proof* dh, const bsx_last_block_id_proof* lb,
                       uint64_t batch_start, uint64_t batch_end, uint64_t E, const uint8_t H_E[32], bsx_subchain* out,
                       uint8_t* cw) {
    if (!B || (B & (B - 1)) || B > BSX_MAX_BATCH) return BSX_ERR_BAD_ARG;
    bsx_witness_layout L = bsx_map_layout(B);
    uint8_t* Y = cw;
    uint32_t* W = cw ? (uint32_t*)(cw + L.off_words) : NULL;
    uint8_t* Bo = cw ? cw + L.off_bools : NULL;
    uint32_t af = 0, first_bad = 0xffffffffu;
    if (cw) {
        memset(cw, 0, L.compact_stride);
        if (range) {
            memcpy(Y + bsx_off_ctx_start_header(), range->start_header_hash, 32);
            memcpy(Y + bsx_off_ctx_end_header(), range->end_header_hash, 32);
            put_u64_words(W + BSX_W_CTX_START, range->start_block);
            put_u64_words(W + BSX_W_CTX_END, range->end_block);
        }
        memcpy(Y + bsx_off_start_header(), start_header, 32);
        memcpy(Y + bsx_off_end_header(), end_header, 32);
        memcpy(Y + bsx_off_dh_proofs(B), dh, (size_t)B * sizeof *dh);
        memcpy(Y + bsx_off_lb_proofs(B), lb, (size_t)B * sizeof *lb);
        put_u64_words(W + BSX_W_BATCH_START, batch_start);
        put_u64_words(W + BSX_W_BATCH_END, batch_end);
    }
    int is_batch_enabled = batch_start < E; /* :174 */
    int curr_enabled = is_batch_enabled;    /* :175 */
    uint8_t curr_header[32];
    memcpy(curr_header, start_header, 32);  /* :176 */
    uint64_t last_to_process = E - 1;       /* :177 */
    if (cw) {
        Bo[BSX_B_BATCH_ENABLED] = (uint8_t)is_batch_enabled;
        put_u64_words(W + BSX_W_LAST_TO_PROCESS, last_to_process);
    }
    for (uint32_t i = 0; i < B; i++) { /* :180 */
        uint64_t curr_idx = batch_start + i;          /* :182 */
        int curr_disabled = !curr_enabled;            /* :184 */
        int is_last = last_to_process == curr_idx;    /* :185 */
        int is_not_last = !is_last;                   /* :186 */
        uint8_t dh_root[32], lb_root[32], dh_path[5][32], lb_path[5][32];
        orc_root_from_proof(dh[i].leaf, BSX_PROTOBUF_HASH_SIZE, dh[i].aunts, DATA_HASH_PATH, 4, dh_root, dh_path); /* :189-193 */
        orc_root_from_proof(lb[i].leaf, BSX_PROTOBUF_BLOCK_ID_SIZE, lb[i].aunts, LAST_BLOCK_ID_PATH, 4, lb_root, lb_path); /* :195-199 */
        const uint8_t* header_hash = lb[i].leaf + 2;  /* :204 */
        int valid_prev = memcmp(curr_header, header_hash, 32) == 0; /* :205 */
        int prev_check = curr_disabled || valid_prev;               /* :206 */
        int dh_valid = memcmp(dh_root, header_hash, 32) == 0;       /* :210 */
        int dh_check = curr_disabled || dh_valid;                   /* :211 */
        int root_matches_end = memcmp(lb_root, H_E, 32) == 0;       /* :216 */
        int end_check = is_not_last || root_matches_end;            /* :218 */
        if (!prev_check) af |= BSX_A3_PREV_HEADER;                  /* :207 */
        if (!dh_check) af |= BSX_A4_DATA_HASH_PROOF;                /* :212 */
        if (!end_check) af |= BSX_A5_END_HEADER;                    /* :219 */
        if ((!prev_check || !dh_check || !end_check) && first_bad == 0xffffffffu) first_bad = i;
        if (curr_enabled) memcpy(curr_header, lb_root, 32);         /* :223 */
        curr_enabled = curr_enabled && is_not_last;                 /* :225 */
        if (cw) {
            put_u64_words(W + BSX_W_CURR_IDX + 2 * i, curr_idx);
            uint8_t* s = Y + bsx_off_slots(B) + BSX_SLOT_BYTES * i;
            memcpy(s, dh_path, 160);
            memcpy(s + 160, lb_path, 160);
            memcpy(s + 320, curr_header, 32);
            uint8_t* b = Bo + BSX_B_SLOTS + BSX_SLOT_BOOLS * i;
            b[0] = (uint8_t)curr_disabled; b[1] = (uint8_t)is_last; b[2] = (uint8_t)valid_prev; b[3] = (uint8_t)prev_check;
            b[4] = (uint8_t)dh_valid; b[5] = (uint8_t)dh_check; b[6] = (uint8_t)root_matches_end; b[7] = (uint8_t)end_check;
            b[8] = (uint8_t)curr_enabled;
        }
    }
    int last_disabled = !curr_enabled;                                /* :229 */
    int last_matches = memcmp(curr_header, end_header, 32) == 0;      /* :230 */
    int end_header_check = last_disabled || last_matches;             /* :231 */
    if (!end_header_check) {                                          /* :232 */
        af |= BSX_A6_BATCH_END;
        if (first_bad == 0xffffffffu) first_bad = B;
    }
    int batch_end_lt = batch_end < E;                                 /* :235 */
    uint64_t temp_end = batch_end_lt ? batch_end : E;                 /* :236-240 */
    int end_lt_start = temp_end < batch_start;                        /* :241 */
    uint64_t end_block_num = end_lt_start ? batch_start : temp_end;   /* :242-243 */
    if (cw) {
        uint8_t* t = Bo + bsx_b_tail(B);
        t[0] = (uint8_t)last_disabled; t[1] = (uint8_t)last_matches; t[2] = (uint8_t)end_header_check;
        t[3] = (uint8_t)batch_end_lt; t[4] = (uint8_t)end_lt_start;
        put_u64_words(W + bsx_w_temp_end(B), temp_end);
        put_u64_words(W + bsx_w_end_block_num(B), end_block_num);
    }
    uint8_t root[32];
    uint32_t af_dc = 0;
    data_commitment_core(B, dh[0].leaf + 2, sizeof dh[0], batch_start, end_block_num, root, &af_dc, cw); /* :245-256 */
    af |= af_dc;
    if (af_dc && first_bad == 0xffffffffu) first_bad = B;
    memset(out, 0, sizeof *out);                                      /* :263-270 */
    out->is_enabled = (uint32_t)is_batch_enabled;
    out->start_block = batch_start;
    memcpy(out->start_header, start_header, 32);
    out->end_block = end_block_num;
    memcpy(out->end_header, curr_header, 32);
    memcpy(out->data_merkle_root, root, 32);
    out->assert_fail = af;
    out->first_bad_slot = first_bad;
    if (cw) {
        uint8_t* r = Y + bsx_off_record(B);
        memcpy(r, out->start_header, 32); memcpy(r + 32, out->end_header, 32); memcpy(r + 64, out->data_merkle_root, 32);
        put_u64_words(W + bsx_w_rec_start(B), out->start_block);
        put_u64_words(W + bsx_w_rec_end(B), out->end_block);
        Bo[bsx_b_rec_enabled(B)] = (uint8_t)is_batch_enabled;
    }
    return af ? BSX_ERR_ASSERT : BSX_OK;
}

/* circuits/builder.rs:337-395 (reduce closure) */
void orc_reduce_pair(const bsx_subchain* l, const bsx_subchain* r, bsx_subchain* out, uint8_t* cw) {
    bsx_witness_layout L = bsx_reduce_layout();
    int right_disabled = r->is_enabled == 0;                                        /* :344 */
    int headers_linked = memcmp(l->end_header, r->start_header, 32) == 0;           /* :348-349 */
    int blocks_linked = l->end_block == r->start_block;                             /* :350 */
    int linked = headers_linked && blocks_linked;                                   /* :351 */
    int link_check = right_disabled || linked;                                      /* :352 */
    uint8_t computed[32];
    orc_inner_hash(l->data_merkle_root, r->data_merkle_root, computed);             /* :357-364 */
    bsx_subchain o;
    memset(&o, 0, sizeof o);
    memcpy(o.data_merkle_root, right_disabled ? l->data_merkle_root : computed, 32); /* :367-371 */
    o.end_block = right_disabled ? l->end_block : r->end_block;                     /* :374-378 */
    memcpy(o.end_header, right_disabled ? l->end_header : r->end_header, 32);       /* :379-383 */
    o.is_enabled = l->is_enabled;                                                   /* :388 */
    o.start_block = l->start_block;                                                 /* :389 */
    memcpy(o.start_header, l->start_header, 32);                                    /* :390 */
    o.assert_fail = l->assert_fail | r->assert_fail | (link_check ? 0 : BSX_A8_REDUCE_LINK); /* :353 */
    o.first_bad_slot = 0xffffffffu;
    if (cw) {
        memset(cw, 0, L.compact_stride);
        memcpy(cw, computed, 32);
        memcpy(cw + 32, o.start_header, 32);
        memcpy(cw + 64, o.end_header, 32);
        memcpy(cw + 96, o.data_merkle_root, 32);
        uint32_t* W = (uint32_t*)(cw + L.off_words);
        put_u64_words(W, o.start_block);
        put_u64_words(W + 2, o.end_block);
        uint8_t* b = cw + L.off_bools;
        b[0] = (uint8_t)right_disabled; b[1] = (uint8_t)headers_linked; b[2] = (uint8_t)blocks_linked;
        b[3] = (uint8_t)linked; b[4] = (uint8_t)link_check; b[5] = (uint8_t)o.is_enabled;
    }
    *out = o;
}

/* plonky2x mapreduce [UPSTREAM]: adjacent pairs, level by level; n a power of two.
 * reduce_compact: n-1 node witnesses in level order (the n/2 parents of the leaves first). */
int orc_reduce(const bsx_subchain* records, uint32_t n, bsx_subchain* out, uint8_t* reduce_compact) {
    if (!n || (n & (n - 1))) return BSX_ERR_BAD_ARG;
    bsx_witness_layout L = bsx_reduce_layout();
    bsx_subchain* cur = malloc(n * sizeof *cur);
    memcpy(cur, records, n * sizeof *cur);
    uint32_t k = 0;
    for (uint32_t m = n; m > 1; m /= 2)
        for (uint32_t i = 0; i < m; i += 2, k++) {
            bsx_subchain t;
            orc_reduce_pair(&cur[i], &cur[i + 1], &t, reduce_compact ? reduce_compact + (size_t)k * L.compact_stride : NULL);
            if ((t.assert_fail & BSX_A8_REDUCE_LINK) && !((cur[i].assert_fail | cur[i + 1].assert_fail) & BSX_A8_REDUCE_LINK))
                t.first_bad_slot = k;
            else
                t.first_bad_slot = cur[i].first_bad_slot != 0xffffffffu ? cur[i].first_bad_slot : cur[i + 1].first_bad_slot;
            cur[i / 2] = t;
        }
    *out = cur[0];
    free(cur);
    return BSX_OK;
}

/* circuits/builder.rs:273-409 with the hint served from `headers` */
int orc_prove_data_commitment(uint32_t J, uint32_t B, const bsx_shared_ctx* range, const bsx_header* headers,
                              uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                              uint8_t out_commitment[32], bsx_subchain* out_result, bsx_subchain* records,
                              uint8_t* compact, uint32_t* status) {
    if (!J || (J & (J - 1)) || !B || (B & (B - 1)) || B > BSX_MAX_BATCH) return BSX_ERR_BAD_ARG;
    bsx_witness_layout L = bsx_map_layout(B);
    uint32_t st = 0;
    uint64_t S = range->start_block, E = range->end_block;
    uint64_t max_blocks = (uint64_t)J * B;                 /* :288 */
    if (!(E <= S + max_blocks)) st |= BSX_A7_RANGE;        /* :292-297 */
    bsx_subchain* recs = records ? records : malloc(J * sizeof *recs);
    bsx_data_hash_proof* dh = malloc(B * sizeof *dh);
    bsx_last_block_id_proof* lb = malloc(B * sizeof *lb);
    int rc = BSX_OK;
    for (uint32_t j = 0; j < J && rc == BSX_OK; j++) {     /* map closure :305-336 */
        uint64_t batch_start = S + (uint64_t)j * B;        /* :315-316 */
        uint64_t last_block = S + (uint64_t)j * B + (B - 1); /* :317-320 */
        uint64_t batch_end = last_block + 1;               /* :322 */
        uint8_t sh[32], eh[32];
        rc = orc_data_commitment_inputs(headers, first_height, n_headers, latest_block, batch_start, batch_end, B, sh, eh,
                                        dh, lb, NULL);     /* :325-332 -> data_commitment.rs:22-44 */
        if (rc) break;
        int r2 = orc_prove_subchain(B, range, sh, eh, dh, lb, batch_start, batch_end, E, range->end_header_hash, &recs[j],
                                    compact ? compact + (size_t)j * L.compact_stride : NULL); /* :335 */
        if (r2 != BSX_OK && r2 != BSX_ERR_ASSERT) rc = r2;
    }
    free(dh); free(lb);
    if (rc) { if (!records) free(recs); return rc; }
    bsx_subchain result;
    orc_reduce(recs, J, &result, compact ? compact + (size_t)J * L.compact_stride : NULL); /* :337-395 */
    st |= result.assert_fail;
    /* :400-406 */
    if (result.start_block != S || memcmp(result.start_header, range->start_header_hash, 32) != 0 ||
        result.end_block != E || memcmp(result.end_header, range->end_header_hash, 32) != 0)
        st |= BSX_A9_FINAL;
    memcpy(out_commitment, result.data_merkle_root, 32);   /* :408 */
    if (out_result) { *out_result = result; out_result->assert_fail = st; }
    if (status) *status = st;
    if (!records) free(recs);
    return st ? BSX_ERR_ASSERT : BSX_OK;
}

/* circuits/builder.rs:411-443 */
int orc_prove_next_header_data_commitment(uint64_t prev_block, const uint8_t prev_header_hash[32], uint64_t next_block,
                                          const bsx_header* header, uint64_t latest_block, uint8_t out[32]) {
    bsx_data_hash_proof dh;
    bsx_last_block_id_proof lb;
    uint8_t sh[32], eh[32];
    /* :415-423 hint with MAX_LEAVES = 1; only the header at prev_block is needed for data_hash_proofs[0] */
    if (next_block - prev_block > 1) return BSX_ERR_RANGE_TOO_LONG;
    if (latest_block < 2) return BSX_ERR_BAD_ARG;
    uint64_t req_end = next_block < latest_block - 2 ? next_block : latest_block - 2;
    memset(&dh, 0, sizeof dh);
    if (prev_block < req_end) {
        int rc = orc_header_hash(header, NULL, &dh, NULL);
        if (rc) return rc;
    }
    (void)lb; (void)sh; (void)eh;
    uint8_t root[32];
    orc_root_from_proof(dh.leaf, BSX_PROTOBUF_HASH_SIZE, dh.aunts, DATA_HASH_PATH, 4, root, NULL); /* :429-433 */
    uint8_t tuple[64];
    orc_encode_data_root_tuple(dh.leaf + 2, prev_block, tuple);  /* :436-439 */
    orc_leaf_hash(tuple, 64, out);                               /* :442 */
    if (memcmp(root, prev_header_hash, 32) != 0) return BSX_ERR_ASSERT; /* :434 (A10) */
    return BSX_OK;
}

/* P10 [UPSTREAM plonky2x vars]: bytes -> 8 bools MSB first; u32 limbs; bools */
void orc_expand_witness(const bsx_witness_layout* L, uint32_t n_jobs, const uint8_t* compact, uint64_t* out) {
    for (uint32_t j = 0; j < n_jobs; j++) {
        const uint8_t* c = compact + (size_t)j * L->compact_stride;
        uint64_t* o = out + (size_t)j * L->n_elements;
        for (uint32_t i = 0; i < L->n_bytes; i++)
            for (int b = 0; b < 8; b++) *o++ = (c[i] >> (7 - b)) & 1;
        const uint32_t* w = (const uint32_t*)(c + L->off_words);
        for (uint32_t i = 0; i < L->n_words; i++) *o++ = w[i];
        const uint8_t* bo = c + L->off_bools;
        for (uint32_t i = 0; i < L->n_bools; i++) *o++ = bo[i];
    }
}
