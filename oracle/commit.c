/* oracle/commit.c — CPU ORACLE (test infrastructure only; see orc.h).
 * Commit verification = the per-validator hot loop inside builder.skip / builder.step
 * (circuits/header_range.rs:42-48, circuits/next_header.rs:32-36).  The circuit body is
 * [UPSTREAM] tendermintx v1.0.0 (not in /root/reference); restated from the public Tendermint
 * light-client rules the reference's host twin applies (is_valid_skip, circuits/fetcher.rs:76-80)
 * and SURVEY Appendix A/B byte formats, all confirmed on the fixture commits:
 *   leaf_i   = 0a 22 0a 20 pk32 [10 varint(power)]            (SimpleValidator)
 *   valhash  = masked power-of-two Merkle tree over leaves, enabled = vals[i].enabled
 *              (== RFC 6962 root over the enabled prefix)
 *   h_i      = SHA512(R ‖ A ‖ M) mod L;  ok_i = [s]B == R + [h]A
 *   msg_ok_i = M carries the header hash at offset 16 (25 when a round field 0x19 is present)
 *   2/3 rule : 3 * signed_power > 2 * total_power
 *   1/3 rule : 3 * (trusted power of trusted validators that validly signed) > trusted total
 * PARITY UNPINNED beyond the V=2 fixture commits. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

void orc_sha512_challenge(const bsx_validator* v, uint8_t h[32], uint8_t digest[64]) {
    uint8_t buf[64 + BSX_VALIDATOR_MSG_MAX], dig[64];
    uint32_t len = v->message_len > BSX_VALIDATOR_MSG_MAX ? BSX_VALIDATOR_MSG_MAX : v->message_len;
    memcpy(buf, v->signature, 32);
    memcpy(buf + 32, v->pubkey, 32);
    memcpy(buf + 64, v->message, len);
    orc_sha512(buf, 64 + len, dig);
    orc_sc_reduce64(dig, h);
    if (digest) memcpy(digest, dig, 64);
}

int orc_validator_leaf(const uint8_t pk[32], uint64_t power, uint8_t out[BSX_VALIDATOR_LEAF_MAX]) {
    int n = 0;
    out[n++] = 0x0a; out[n++] = 0x22; out[n++] = 0x0a; out[n++] = 0x20;
    memcpy(out + n, pk, 32);
    n += 32;
    if (power) {
        out[n++] = 0x10;
        while (power >= 0x80) { out[n++] = (uint8_t)(power | 0x80); power >>= 7; }
        out[n++] = (uint8_t)power;
    }
    return n;
}

static uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p *= 2; return p; }

/* where a validator-set tree goes in a compact witness unit (include/bsx_layout.h): the commit unit's own set, or the skip
 * unit's trusted set */
typedef struct {
    uint8_t* cw;
    uint32_t off_leaf, off_leaf_hash, off_inner, off_node, off_root;
    uint32_t* leaf_len_word;      /* word of slot 0's validator_byte_length */
    uint32_t word_stride;
    uint8_t *leaf_enabled, *node_enabled;
} tree_dst;

static void validators_hash(const bsx_validator* vals, uint32_t v_max, uint8_t out[32], const tree_dst* w) {
    uint32_t P = next_pow2(v_max);
    uint8_t(*nodes)[32] = calloc(P, 32);
    uint8_t* en = calloc(P, 1);
    for (uint32_t i = 0; i < v_max; i++) {
        uint8_t leaf[BSX_VALIDATOR_LEAF_MAX] = {0};
        int n = orc_validator_leaf(vals[i].pubkey, vals[i].voting_power, leaf);
        orc_leaf_hash(leaf, (size_t)n, nodes[i]);
        en[i] = vals[i].enabled != 0;
        if (w) {
            memcpy(w->cw + w->off_leaf + 48 * i, leaf, 48);
            w->leaf_len_word[(size_t)i * w->word_stride] = (uint32_t)n;
        }
    }
    for (uint32_t i = v_max; i < P; i++) { /* padding slots: leaf of an all-zero validator, disabled */
        uint8_t leaf[BSX_VALIDATOR_LEAF_MAX], zero[32] = {0};
        int n = orc_validator_leaf(zero, 0, leaf);
        orc_leaf_hash(leaf, (size_t)n, nodes[i]);
    }
    if (w)
        for (uint32_t i = 0; i < P; i++) {
            memcpy(w->cw + w->off_leaf_hash + 32 * i, nodes[i], 32);
            w->leaf_enabled[i] = en[i];
        }
    uint32_t k = 0;
    for (uint32_t n = P; n > 1; n /= 2)
        for (uint32_t i = 0; i < n; i += 2, k++) {
            uint8_t inner[32], sel[32];
            orc_inner_hash(nodes[i], nodes[i + 1], inner);
            memcpy(sel, (en[i] && en[i + 1]) ? inner : nodes[i], 32);
            const uint8_t any = en[i] || en[i + 1];
            if (w) {
                memcpy(w->cw + w->off_inner + 32 * k, inner, 32);
                memcpy(w->cw + w->off_node + 32 * k, sel, 32);
                w->node_enabled[k] = any;
            }
            memcpy(nodes[i / 2], sel, 32);
            en[i / 2] = any;
        }
    memcpy(out, nodes[0], 32);
    if (w) memcpy(w->cw + w->off_root, nodes[0], 32);
    free(nodes);
    free(en);
}

static void put_u64(uint32_t* w, uint64_t v) { w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }

/* cw (optional): the COMMIT unit's compact witness, bsx_commit_layout(v_max).compact_stride bytes.  Every slot's challenge,
 * message predicates and leaf are evaluated whether or not the slot is enabled / signed (static circuit). */
void orc_verify_commit_w(const bsx_validator* vals, uint32_t v_max, const uint8_t header_hash[32], bsx_commit_result* out,
                         uint8_t* sig_ok, uint8_t* cw) {
    const uint32_t V = v_max;
    bsx_witness_layout L = bsx_commit_layout(V);
    uint32_t* W = cw ? (uint32_t*)(cw + L.off_words) : NULL;
    uint8_t* Bo = cw ? cw + L.off_bools : NULL;
    if (cw) {
        memset(cw, 0, L.compact_stride);
        memcpy(cw + bsx_cm_off_header_hash(), header_hash, 32);
    }
    memset(out, 0, sizeof *out);
    out->first_bad_signature = 0xffffffffu;
    unsigned __int128 exact_total = 0;
    for (uint32_t i = 0; i < v_max; i++) {
        const bsx_validator* v = &vals[i];
        uint8_t ok = 0;
        uint8_t h[32], dig[64];
        orc_sha512_challenge(v, h, dig);
        const int has_round = v->message_len > 12 && v->message[12] == 0x19;
        const uint32_t off = has_round ? 25 : 16;
        const int msg = v->message_len <= BSX_VALIDATOR_MSG_MAX && v->message_len >= off + 32 &&
                        memcmp(v->message + off, header_hash, 32) == 0;
        int sig = 0;
        if (v->enabled) {
            out->n_enabled++;
            out->total_power += v->voting_power;
            exact_total += v->voting_power;
            if (v->is_signed) {
                out->n_signed++;
                sig = orc_ed25519_verify_h(v->pubkey, v->signature, h);
                if (!sig) {
                    out->n_bad_signature++;
                    if (out->first_bad_signature == 0xffffffffu) out->first_bad_signature = i;
                }
                if (!msg) out->n_bad_message++;
                if (sig && msg) {
                    out->signed_power += v->voting_power;
                    if (v->present_on_trusted) out->trusted_signed_power += v->voting_power;
                }
                ok = (uint8_t)(sig != 0);
            }
        }
        if (sig_ok) sig_ok[i] = ok;
        if (cw) {
            memcpy(cw + bsx_cm_off_digest(V) + 64 * i, dig, 64);
            memcpy(cw + bsx_cm_off_challenge(V) + 32 * i, h, 32);
            uint8_t* rec = cw + bsx_cm_off_validators(V) + BSX_CM_VAL_BYTES * i;
            memcpy(rec, v->pubkey, 32);
            memcpy(rec + 32, v->signature, 64);
            memcpy(rec + 96, v->message, BSX_VALIDATOR_MSG_MAX);
            uint32_t* ws = W + BSX_CM_SLOT_WORDS * i;
            ws[0] = v->message_len;
            put_u64(ws + 2, v->voting_power);
            uint8_t* b = Bo + BSX_CM_SLOT_BOOLS * i;
            b[0] = v->enabled != 0; b[1] = v->is_signed != 0; b[2] = v->present_on_trusted != 0; b[3] = ok;
            b[4] = (uint8_t)has_round; b[5] = (uint8_t)msg; b[6] = (uint8_t)(v->enabled && v->is_signed && sig && msg);
        }
    }
    tree_dst td = {cw, bsx_cm_off_leaf(V), bsx_cm_off_leaf_hash(V), bsx_cm_off_inner(V), bsx_cm_off_node(V), bsx_cm_off_root(V),
                   W ? W + 1 : NULL, BSX_CM_SLOT_WORDS, Bo ? Bo + bsx_cm_b_leaf_enabled(V) : NULL, Bo ? Bo + bsx_cm_b_node_enabled(V) : NULL};
    validators_hash(vals, v_max, out->validators_hash, cw ? &td : NULL);
    /* Tendermint caps a set's total at MaxTotalVotingPower = MaxInt64 / 8; beyond it the u64 sums may have wrapped */
    out->power_overflow = exact_total > (unsigned __int128)BSX_MAX_TOTAL_VOTING_POWER;
    /* 3*signed > 2*total, in 128-bit to avoid overflow */
    out->two_thirds_ok = !out->power_overflow && (unsigned __int128)out->signed_power * 3 > (unsigned __int128)out->total_power * 2;
    if (cw) {
        put_u64(W + bsx_cm_w_total(V), out->total_power);
        put_u64(W + bsx_cm_w_total(V) + 2, out->signed_power);
        put_u64(W + bsx_cm_w_total(V) + 4, out->trusted_signed_power);
        uint8_t* t = Bo + bsx_cm_b_tail(V);
        t[0] = (uint8_t)out->two_thirds_ok; t[1] = (uint8_t)out->power_overflow;
        t[2] = (uint8_t)(!out->n_bad_signature && !out->n_bad_message);
    }
}

void orc_verify_commit(const bsx_validator* vals, uint32_t v_max, const uint8_t header_hash[32], bsx_commit_result* out,
                       uint8_t* sig_ok) {
    orc_verify_commit_w(vals, v_max, header_hash, out, sig_ok, NULL);
}

/* header-field inclusion proof record (include/bsx_layout.h): aunts[4][32], path[5][32], leaf[cap]; *len_word = leaf length.
 * Tree of the 14 encoded fields: tendermint Header::hash (RFC 6962 style); index <= 11 -> depth 4. */
static void field_proof(const bsx_header* h, uint32_t idx, uint8_t* rec, uint32_t cap, uint32_t* len_word) {
    const uint8_t* items[14];
    size_t lens[14];
    items[0] = h->version; items[1] = h->chain_id; items[2] = h->height; items[3] = h->time; items[4] = h->last_block_id;
    for (int i = 0; i < 8; i++) items[5 + i] = h->hash[i];
    items[13] = h->proposer;
    for (int i = 0; i < 14; i++) lens[i] = h->len[i];
    uint8_t aunts[4][32], path[5][32], root[32], bits[4];
    int depth = orc_merkle_proof(items, lens, 14, idx, aunts);
    (void)depth;
    for (int k = 0; k < 4; k++) bits[k] = (uint8_t)((idx >> k) & 1);
    orc_root_from_proof(items[idx], lens[idx], (const uint8_t(*)[32])aunts, bits, 4, root, path);
    memcpy(rec, aunts, 128);
    memcpy(rec + 128, path, 160);
    memset(rec + BSX_PROOF_FIXED, 0, cap);
    memcpy(rec + BSX_PROOF_FIXED, items[idx], lens[idx] < cap ? lens[idx] : cap);
    *len_word = (uint32_t)lens[idx];
}

static int varint_height_field(uint64_t h, uint8_t out[12]) {
    int n = 0;
    out[n++] = 0x08;
    while (h >= 0x80) { out[n++] = (uint8_t)(h | 0x80); h >>= 7; }
    out[n++] = (uint8_t)h;
    return n;
}

/* circuits/header_range.rs:32-59 */
int orc_header_range(uint32_t J, uint32_t B, const uint8_t input48[48], const bsx_header* headers, uint64_t first_height,
                     uint64_t n_headers, uint64_t latest_block, const bsx_validator* target_validators,
                     const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len,
                     uint8_t output64[64], bsx_commit_result* out_commit, uint8_t* compact) {
    uint64_t trusted_block = 0, target_block = 0;
    for (int i = 0; i < 8; i++) trusted_block = trusted_block << 8 | input48[i];      /* :33 evm_read U64 = big endian */
    const uint8_t* trusted_header_hash = input48 + 8;                                /* :34 */
    for (int i = 0; i < 8; i++) target_block = target_block << 8 | input48[40 + i];  /* :35 */
    if (!(target_block > trusted_block) || target_block - trusted_block > (uint64_t)J * B) return BSX_ERR_RANGE_TOO_LONG;
    if (trusted_block < first_height || target_block - first_height >= n_headers) return BSX_ERR_BAD_ARG;
    const bsx_header* th = &headers[target_block - first_height];
    const bsx_header* tr = &headers[trusted_block - first_height];
    /* builder.skip (:42-48) [UPSTREAM]: target header hash, commit, validator sets */
    uint8_t target_hash[32], trusted_hash[32];
    int rc;
    if ((rc = orc_header_hash(th, target_hash, NULL, NULL))) return rc;
    if ((rc = orc_header_hash(tr, trusted_hash, NULL, NULL))) return rc;
    /* The three header assertions below, the commit's verdicts and the power rules are all evaluated; the status reported
     * is the FIRST failing one in this fixed order (the product's k_skip_check uses the same order): voting-power overflow
     * (the tallies cannot be trusted) -> trusted hash -> height leaf -> chain-id leaf -> signatures -> validator-set hashes
     * -> 2/3 -> 1/3. */
    int header_assert = 0;
    if (memcmp(trusted_hash, trusted_header_hash, 32) != 0) header_assert = 1;
    uint8_t hf[12];
    int hn = varint_height_field(target_block, hf);
    if (th->len[BSX_BLOCK_HEIGHT_INDEX] != hn || memcmp(th->height, hf, (size_t)hn) != 0) header_assert = 1;
    /* builder.skip is called with C::CHAIN_ID_BYTES (:42-43): the target header's chain-id leaf is 0a len bytes */
    const int chain_ok = !(chain_id_len > 50 || th->len[1] != chain_id_len + 2 || th->chain_id[0] != 0x0a || th->chain_id[1] != chain_id_len ||
                           memcmp(th->chain_id + 2, chain_id, chain_id_len) != 0);
    if (!chain_ok) header_assert = 1;
    bsx_commit_result cr, trc;
    uint8_t* ok = malloc(v_max);
    /* compact (optional) = J map jobs, J - 1 reduce nodes, the COMMIT unit of the target commit, the SKIP unit */
    const bsx_witness_layout ML = bsx_map_layout(B), RL = bsx_reduce_layout(), CL = bsx_commit_layout(v_max), SL = bsx_skip_layout(v_max);
    uint8_t* ccw = compact ? compact + (size_t)J * ML.compact_stride + (size_t)(J - 1) * RL.compact_stride : NULL;
    uint8_t* scw = ccw ? ccw + CL.compact_stride : NULL;
    orc_verify_commit_w(target_validators, v_max, target_hash, &cr, ok, ccw);
    if (out_commit) *out_commit = cr;
    int status = BSX_OK;
    {
        unsigned __int128 tt = 0;
        for (uint32_t i = 0; i < v_max; i++) if (trusted_validators[i].enabled) tt += trusted_validators[i].voting_power;
        if (cr.power_overflow || tt > (unsigned __int128)BSX_MAX_TOTAL_VOTING_POWER) status = BSX_ERR_BAD_ARG;
    }
    if (!status && header_assert) status = BSX_ERR_ASSERT;
    if (!status && (cr.n_bad_signature || cr.n_bad_message)) status = BSX_ERR_BAD_SIGNATURE;
    /* validators_hash (field 7 = hash[2]) of both headers */
    if (!status && (th->len[7] != 34 || memcmp(th->hash[2] + 2, cr.validators_hash, 32) != 0)) status = BSX_ERR_ASSERT;
    uint32_t* SW = scw ? (uint32_t*)(scw + SL.off_words) : NULL;
    uint8_t* SB = scw ? scw + SL.off_bools : NULL;
    if (scw) memset(scw, 0, SL.compact_stride);
    tree_dst td = {scw, bsx_sk_off_leaf(v_max), bsx_sk_off_leaf_hash(v_max), bsx_sk_off_inner(v_max), bsx_sk_off_node(v_max),
                   bsx_sk_off_root(v_max), SW ? SW + BSX_SK_W_SLOTS : NULL, 3, SB ? SB + bsx_sk_b_leaf_enabled(v_max) : NULL,
                   SB ? SB + bsx_sk_b_node_enabled(v_max) : NULL};
    validators_hash(trusted_validators, v_max, trc.validators_hash, scw ? &td : NULL);
    if (!status && (tr->len[7] != 34 || memcmp(tr->hash[2] + 2, trc.validators_hash, 32) != 0)) status = BSX_ERR_ASSERT;
    if (!status && !cr.two_thirds_ok) status = BSX_ERR_VOTING_POWER;
    /* > 1/3 of the trusted power signed the target (fetcher.rs:76-80 is_valid_skip) */
    unsigned __int128 trusted_total = 0, overlap = 0;
    for (uint32_t i = 0; i < v_max; i++) {
        if (!trusted_validators[i].enabled) continue;
        trusted_total += trusted_validators[i].voting_power;
        for (uint32_t k = 0; k < v_max; k++)
            if (target_validators[k].enabled && target_validators[k].is_signed && ok[k] &&
                memcmp(target_validators[k].pubkey, trusted_validators[i].pubkey, 32) == 0) {
                overlap += trusted_validators[i].voting_power;
                if (SB) SB[2 * i + 1] = 1;
                break;
            }
    }
    free(ok);
    if (!status && !(overlap * 3 > trusted_total)) status = BSX_ERR_VOTING_POWER;
    if (out_commit) out_commit->trusted_signed_power = (uint64_t)overlap;
    /* prove_data_commitment (:50-55) */
    bsx_shared_ctx range;
    range.start_block = trusted_block;
    range.end_block = target_block;
    memcpy(range.start_header_hash, trusted_header_hash, 32);
    memcpy(range.end_header_hash, target_hash, 32);
    uint8_t commitment[32];
    uint32_t st = 0;
    rc = orc_prove_data_commitment(J, B, &range, headers, first_height, n_headers, latest_block, commitment, NULL, NULL,
                                   compact, &st);
    if (rc != BSX_OK && rc != BSX_ERR_ASSERT) return rc;
    memcpy(output64, target_hash, 32);      /* :57 */
    memcpy(output64 + 32, commitment, 32);  /* :58 */
    if (scw) {
        const uint32_t V = v_max;
        memcpy(scw, trusted_header_hash, 32);
        memcpy(scw + 32, target_hash, 32);
        memcpy(scw + 64, commitment, 32);
        for (uint32_t i = 0; i < V; i++) {
            memcpy(scw + bsx_sk_off_pubkeys(V) + 32 * i, trusted_validators[i].pubkey, 32);
            SW[BSX_SK_W_SLOTS + 3 * i + 1] = (uint32_t)trusted_validators[i].voting_power;
            SW[BSX_SK_W_SLOTS + 3 * i + 2] = (uint32_t)(trusted_validators[i].voting_power >> 32);
            SB[2 * i] = trusted_validators[i].enabled != 0;
        }
        field_proof(th, 1, scw + bsx_sk_off_proof(V, 0), bsx_sk_proof_cap(0), SW + BSX_SK_W_LEAF_LEN);
        field_proof(th, BSX_BLOCK_HEIGHT_INDEX, scw + bsx_sk_off_proof(V, 1), bsx_sk_proof_cap(1), SW + BSX_SK_W_LEAF_LEN + 1);
        field_proof(th, 7, scw + bsx_sk_off_proof(V, 2), bsx_sk_proof_cap(2), SW + BSX_SK_W_LEAF_LEN + 2);
        field_proof(tr, 7, scw + bsx_sk_off_proof(V, 3), bsx_sk_proof_cap(3), SW + BSX_SK_W_LEAF_LEN + 3);
        put_u64(SW + BSX_SK_W_TRUSTED_BLOCK, trusted_block);
        put_u64(SW + BSX_SK_W_TARGET_BLOCK, target_block);
        put_u64(SW + bsx_sk_w_total(V), (uint64_t)trusted_total);
        put_u64(SW + bsx_sk_w_total(V) + 2, (uint64_t)overlap);
        uint8_t* c = SB + bsx_sk_b_checks(V);
        c[0] = memcmp(trusted_hash, trusted_header_hash, 32) == 0;
        c[1] = th->len[BSX_BLOCK_HEIGHT_INDEX] == hn && memcmp(th->height, hf, (size_t)hn) == 0;
        c[2] = chain_ok;
        c[3] = !cr.n_bad_signature && !cr.n_bad_message;
        c[4] = th->len[7] == 34 && memcmp(th->hash[2] + 2, cr.validators_hash, 32) == 0;
        c[5] = tr->len[7] == 34 && memcmp(tr->hash[2] + 2, trc.validators_hash, 32) == 0;
        c[6] = (uint8_t)cr.two_thirds_ok;
        c[7] = (unsigned __int128)(uint64_t)overlap * 3 > (unsigned __int128)(uint64_t)trusted_total;   /* on the u64 sums, like the words above */
        c[8] = cr.power_overflow || trusted_total > (unsigned __int128)BSX_MAX_TOTAL_VOTING_POWER;
    }
    if (status) return status;
    return rc;
}

/* ------------------------------------------------------------------ cpu_baseline driver */
typedef struct {
    uint32_t n_ranges, J, B, v_max, reps;
    const bsx_shared_ctx* ranges;
    const bsx_header* headers;
    uint64_t headers_per_range;
    const uint64_t* latest;
    const bsx_validator *target, *trusted;
    int with_witness, n_threads, tid;
    const uint8_t* chain_id;
    uint32_t chain_id_len;
    uint8_t* out64;
    uint64_t checksum;
    int rc;
} job_t;

static void* worker(void* arg) {
    job_t* jb = arg;
    bsx_witness_layout L = bsx_map_layout(jb->B), R = bsx_reduce_layout(), CL = bsx_commit_layout(jb->v_max), SL = bsx_skip_layout(jb->v_max);
    size_t csz = (size_t)jb->J * L.compact_stride + (size_t)(jb->J - 1) * R.compact_stride + CL.compact_stride + SL.compact_stride;
    uint8_t* compact = jb->with_witness ? malloc(csz) : NULL;
    /* one map job's worth of expanded elements at a time (3.6 MB at B = 64): same work, bounded memory per thread */
    size_t wel = L.n_elements > (size_t)(jb->J - 1) * R.n_elements ? L.n_elements : (size_t)(jb->J - 1) * R.n_elements;
    if (CL.n_elements > wel) wel = CL.n_elements;
    if (SL.n_elements > wel) wel = SL.n_elements;
    uint64_t* wit = jb->with_witness ? malloc(wel * 8) : NULL;
    uint64_t cs = 0;
    uint8_t o64[64];
    const uint32_t n_tasks = jb->n_ranges * jb->reps;
    for (uint32_t t = (uint32_t)jb->tid; t < n_tasks; t += (uint32_t)jb->n_threads) {
        const uint32_t r = t % jb->n_ranges;
        uint8_t in48[48];
        const bsx_shared_ctx* rg = &jb->ranges[r];
        for (int i = 0; i < 8; i++) in48[i] = (uint8_t)(rg->start_block >> (56 - 8 * i));
        memcpy(in48 + 8, rg->start_header_hash, 32);
        for (int i = 0; i < 8; i++) in48[40 + i] = (uint8_t)(rg->end_block >> (56 - 8 * i));
        int rc = orc_header_range(jb->J, jb->B, in48, jb->headers + (size_t)r * jb->headers_per_range, rg->start_block,
                                  jb->headers_per_range, jb->latest[r], jb->target + (size_t)r * jb->v_max,
                                  jb->trusted + (size_t)r * jb->v_max, jb->v_max, jb->chain_id, jb->chain_id_len, o64, NULL, compact);
        if (rc) jb->rc = rc;
        if (t < jb->n_ranges) memcpy(jb->out64 + 64 * (size_t)r, o64, 64);
        if (compact) {
            for (uint32_t j = 0; j < jb->J; j++) {
                orc_expand_witness(&L, 1, compact + (size_t)j * L.compact_stride, wit);
                for (size_t i = 0; i < L.n_elements; i += 4099) cs += wit[i] * (i + 1);
            }
            if (jb->J > 1) {
                orc_expand_witness(&R, jb->J - 1, compact + (size_t)jb->J * L.compact_stride, wit);
                for (size_t i = 0; i < (size_t)(jb->J - 1) * R.n_elements; i += 61) cs += wit[i] * (i + 1);
            }
            const uint8_t* ccw = compact + (size_t)jb->J * L.compact_stride + (size_t)(jb->J - 1) * R.compact_stride;
            orc_expand_witness(&CL, 1, ccw, wit);                      /* the COMMIT unit, then the SKIP unit */
            for (size_t i = 0; i < CL.n_elements; i += 4099) cs += wit[i] * (i + 1);
            orc_expand_witness(&SL, 1, ccw + CL.compact_stride, wit);
            for (size_t i = 0; i < SL.n_elements; i += 61) cs += wit[i] * (i + 1);
        }
        for (int i = 0; i < 64; i++) cs += o64[i];
    }
    jb->checksum = cs;
    free(compact);
    free(wit);
    return NULL;
}

/* n_ranges * reps independent header_range tasks (task t works on range t % n_ranges) over n_threads threads */
int orc_bench_header_range(uint32_t n_ranges, uint32_t reps, uint32_t J, uint32_t B, const bsx_shared_ctx* ranges,
                           const bsx_header* headers, uint64_t headers_per_range, const uint64_t* latest_block,
                           const bsx_validator* target_validators, const bsx_validator* trusted_validators, uint32_t v_max,
                           const uint8_t* chain_id, uint32_t chain_id_len, int with_witness, int n_threads, uint8_t* out64,
                           uint64_t* checksum) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if (reps < 1) reps = 1;
    job_t* jobs = calloc((size_t)n_threads, sizeof *jobs);
    pthread_t* th = calloc((size_t)n_threads, sizeof *th);
    for (int t = 0; t < n_threads; t++) {
        job_t j = {n_ranges, J, B, v_max, reps, ranges, headers, headers_per_range, latest_block, target_validators,
                   trusted_validators, with_witness, n_threads, t, chain_id, chain_id_len, out64, 0, 0};
        jobs[t] = j;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    int rc = 0;
    uint64_t cs = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        cs += jobs[t].checksum;
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    if (checksum) *checksum = cs;
    free(jobs);
    free(th);
    return rc;
}

/* mode S (one commit per header; BASELINE configs #4/#5, next_header.rs:25-47 per header): n_commits * reps
 * orc_verify_commit tasks over n_threads threads; results / sig_ok of the first repetition are returned */
typedef struct {
    uint32_t n_commits, v_max, reps;
    const bsx_validator* vals;
    const uint8_t* hashes;
    bsx_commit_result* res;
    uint8_t* ok;
    int n_threads, tid;
} sjob_t;
static void* sworker(void* arg) {
    sjob_t* jb = arg;
    uint8_t* tmp_ok = malloc(jb->v_max);
    const uint32_t n_tasks = jb->n_commits * jb->reps;
    for (uint32_t t = (uint32_t)jb->tid; t < n_tasks; t += (uint32_t)jb->n_threads) {
        const uint32_t c = t % jb->n_commits;
        bsx_commit_result r;
        orc_verify_commit(jb->vals + (size_t)c * jb->v_max, jb->v_max, jb->hashes + 32 * (size_t)c, &r, tmp_ok);
        if (t < jb->n_commits) {
            jb->res[c] = r;
            if (jb->ok) memcpy(jb->ok + (size_t)c * jb->v_max, tmp_ok, jb->v_max);
        }
    }
    free(tmp_ok);
    return NULL;
}
int orc_bench_verify_commits(uint32_t n_commits, uint32_t reps, uint32_t v_max, const bsx_validator* validators,
                             const uint8_t* header_hashes, int n_threads, bsx_commit_result* out_results, uint8_t* out_sig_ok) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if (reps < 1) reps = 1;
    sjob_t* jobs = calloc((size_t)n_threads, sizeof *jobs);
    pthread_t* th = calloc((size_t)n_threads, sizeof *th);
    for (int t = 0; t < n_threads; t++) {
        sjob_t j = {n_commits, v_max, reps, validators, header_hashes, out_results, out_sig_ok, n_threads, t};
        jobs[t] = j;
        pthread_create(&th[t], NULL, sworker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return BSX_OK;
}


/* ---------------------------------------------------------------- skip-target search (fetcher.rs:60-87) */
/* is_valid_skip(start set, target set, target commit): [UPSTREAM] tendermintx v1.0.0 — PARITY UNPINNED.  Restated as:
 * the start-set validators that also signed the target commit hold more than 1/3 of the start set's voting power
 * (the rule builder.skip enforces in circuit).  Signatures are counted by presence (BlockIDFlag
 * commit), not verified: the operator's search does not verify them either (fetcher.rs:76-80 passes the bare commit). */
void orc_is_valid_skip(const bsx_validator* sv, const bsx_validator* tv, uint32_t v_max, orc_skip_eval* out) {
    unsigned __int128 start_total = 0, overlap = 0, signed_p = 0, target_total = 0;
    for (uint32_t k = 0; k < v_max; k++) {
        if (!tv[k].enabled) continue;
        target_total += tv[k].voting_power;
        if (tv[k].is_signed) signed_p += tv[k].voting_power;
    }
    for (uint32_t i = 0; i < v_max; i++) {
        if (!sv[i].enabled) continue;
        start_total += sv[i].voting_power;
        for (uint32_t k = 0; k < v_max; k++)
            if (tv[k].enabled && tv[k].is_signed && memcmp(tv[k].pubkey, sv[i].pubkey, 32) == 0) {
                overlap += sv[i].voting_power;
                break;
            }
    }
    out->overlap_power = (uint64_t)overlap;
    out->start_total_power = (uint64_t)start_total;
    out->signed_power = (uint64_t)signed_p;
    out->target_total_power = (uint64_t)target_total;
    const unsigned __int128 cap = BSX_MAX_TOTAL_VOTING_POWER;
    out->power_overflow = (start_total > cap || target_total > cap) ? 1u : 0u;
    out->valid = (!out->power_overflow && overlap * 3 > start_total) ? 1u : 0u;
}

/* fetcher.rs:60-87, line by line; candidates must contain every height the loop visits */
int orc_find_block_to_request(uint64_t start_block, uint64_t max_end_block, const bsx_validator* start_validators,
                              uint32_t n_candidates, const uint64_t* heights, const bsx_validator* cand, uint32_t v_max,
                              uint64_t* out_block, orc_skip_eval* out_evals) {
    if (max_end_block <= start_block) return BSX_ERR_BAD_ARG;
    for (uint32_t c = 0; c < n_candidates; c++) {
        orc_skip_eval e;
        orc_is_valid_skip(start_validators, cand + (size_t)c * v_max, v_max, &e);
        if (out_evals) out_evals[c] = e;
        if (e.power_overflow) return BSX_ERR_BAD_ARG;
    }
    uint64_t curr_end_block = max_end_block;                                   /* :61 */
    for (;;) {                                                                 /* :62 */
        if (curr_end_block - start_block == 1) { *out_block = curr_end_block; return BSX_OK; }   /* :63-65 */
        uint32_t c = 0;
        while (c < n_candidates && heights[c] != curr_end_block) c++;
        if (c == n_candidates) return BSX_ERR_BAD_ARG;                         /* the caller did not supply this height */
        orc_skip_eval e;
        orc_is_valid_skip(start_validators, cand + (size_t)c * v_max, v_max, &e);               /* :76-80 */
        if (e.valid) { *out_block = curr_end_block; return BSX_OK; }           /* :81 */
        curr_end_block = (curr_end_block + start_block) / 2;                   /* :84-85 */
    }
}


/* ---------------------------------------------------------------- next_header (circuits/next_header.rs:25-46) */
/* compact (optional): the COMMIT unit of the next header's commit, then the STEP unit (include/bsx_layout.h) */
int orc_next_header_w(const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                      uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                      uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit, uint8_t* compact) {
    uint64_t prev_block = 0;                                       /* :26 */
    for (int i = 0; i < 8; i++) prev_block = prev_block << 8 | input40[i];
    const uint8_t* prev_hash = input40 + 8;                        /* :27 */
    const uint64_t next_block = prev_block + 1;                    /* :29-30 */
    uint8_t hp[32], hn_[32];
    if (orc_header_check(prev_header) || orc_header_check(next_header)) return BSX_ERR_BAD_HEADER;
    orc_header_hash(prev_header, hp, NULL, NULL);
    orc_header_hash(next_header, hn_, NULL, NULL);
    /* builder.step :32-36 [UPSTREAM] */
    bsx_commit_result cr;
    const bsx_witness_layout CL = bsx_commit_layout(v_max ? v_max : 1), TL = bsx_step_layout();
    uint8_t* scw = compact ? compact + CL.compact_stride : NULL;
    uint8_t* ok = (uint8_t*)malloc(v_max ? v_max : 1);
    orc_verify_commit_w(next_validators, v_max, hn_, &cr, ok, compact);
    free(ok);
    if (out_commit) *out_commit = cr;
    int st = BSX_OK;
    if (cr.power_overflow) return BSX_ERR_BAD_ARG;
    const int c_prev = memcmp(hp, prev_hash, 32) == 0;
    if (!c_prev) st = BSX_ERR_ASSERT;
    uint8_t hf[12];
    int hl = 0;
    hf[hl++] = 0x08;
    for (uint64_t hv = next_block;; hv >>= 7) { if (hv >= 0x80) hf[hl++] = (uint8_t)(hv | 0x80); else { hf[hl++] = (uint8_t)hv; break; } }
    const int c_height = next_header->len[BSX_BLOCK_HEIGHT_INDEX] == hl && memcmp(next_header->height, hf, (size_t)hl) == 0;
    if (!st && !c_height) st = BSX_ERR_ASSERT;
    const int c_chain = !(chain_id_len > 50 || next_header->len[1] != chain_id_len + 2 || next_header->chain_id[0] != 0x0a ||
                          next_header->chain_id[1] != chain_id_len || memcmp(next_header->chain_id + 2, chain_id, chain_id_len) != 0);
    if (!st && !c_chain) st = BSX_ERR_ASSERT;   /* builder.step is called with C::CHAIN_ID_BYTES (next_header.rs:32-33) */
    const int c_sigs = !cr.n_bad_signature && !cr.n_bad_message;
    if (!st && !c_sigs) st = BSX_ERR_BAD_SIGNATURE;
    const int c_vh = next_header->len[7] == 34 && memcmp(next_header->hash[2] + 2, cr.validators_hash, 32) == 0;
    if (!st && !c_vh) st = BSX_ERR_ASSERT;
    const int c_nvh = prev_header->len[8] == 34 && memcmp(prev_header->hash[3] + 2, cr.validators_hash, 32) == 0;
    if (!st && !c_nvh) st = BSX_ERR_ASSERT;
    const int c_lb = next_header->len[BSX_LAST_BLOCK_ID_INDEX] >= 34 && memcmp(next_header->last_block_id + 2, hp, 32) == 0;
    if (!st && !c_lb) st = BSX_ERR_ASSERT;
    if (!st && !cr.two_thirds_ok) st = BSX_ERR_VOTING_POWER;
    uint8_t dc[32];
    const int rc = orc_prove_next_header_data_commitment(prev_block, prev_hash, next_block, prev_header, latest_block, dc);   /* :38-42 */
    if (rc != BSX_OK && rc != BSX_ERR_ASSERT) return rc;
    memcpy(output64, hn_, 32);                                     /* :44 */
    memcpy(output64 + 32, dc, 32);                                 /* :45 */
    if (scw) {
        memset(scw, 0, TL.compact_stride);
        uint32_t* W = (uint32_t*)(scw + TL.off_words);
        uint8_t* Bo = scw + TL.off_bools;
        memcpy(scw, prev_hash, 32);
        memcpy(scw + 32, hn_, 32);
        memcpy(scw + 64, dc, 32);
        field_proof(next_header, 1, scw + bsx_st_off_proof(0), bsx_st_proof_cap(0), W + BSX_ST_W_LEAF_LEN);
        field_proof(next_header, BSX_BLOCK_HEIGHT_INDEX, scw + bsx_st_off_proof(1), bsx_st_proof_cap(1), W + BSX_ST_W_LEAF_LEN + 1);
        field_proof(next_header, 7, scw + bsx_st_off_proof(2), bsx_st_proof_cap(2), W + BSX_ST_W_LEAF_LEN + 2);
        field_proof(next_header, BSX_LAST_BLOCK_ID_INDEX, scw + bsx_st_off_proof(3), bsx_st_proof_cap(3), W + BSX_ST_W_LEAF_LEN + 3);
        field_proof(prev_header, 8, scw + bsx_st_off_proof(4), bsx_st_proof_cap(4), W + BSX_ST_W_LEAF_LEN + 4);
        /* data_hash_proofs[0] of the MAX_LEAVES = 1 hint (builder.rs:418-423): real iff prev < min(next, latest - 2)
         * (input.rs:160-172), otherwise the all-zero proof (:220-239) — its path is then the zero proof's */
        const uint64_t req_end = next_block < latest_block - 2 ? next_block : latest_block - 2;
        uint8_t* pr = scw + bsx_st_off_proof(5);
        if (prev_block < req_end) {
            field_proof(prev_header, BSX_DATA_HASH_INDEX, pr, bsx_st_proof_cap(5), W + BSX_ST_W_LEAF_LEN + 5);
        } else {
            static const uint8_t DH_PATH[4] = {0, 1, 1, 0};
            uint8_t zl[BSX_PROTOBUF_HASH_SIZE] = {0}, za[4][32] = {{0}}, root[32], path[5][32];
            orc_root_from_proof(zl, BSX_PROTOBUF_HASH_SIZE, (const uint8_t(*)[32])za, DH_PATH, 4, root, path);
            memcpy(pr + 128, path, 160);
            W[BSX_ST_W_LEAF_LEN + 5] = BSX_PROTOBUF_HASH_SIZE;
        }
        orc_encode_data_root_tuple(pr + BSX_PROOF_FIXED + 2, prev_block, scw + bsx_st_off_tuple());   /* builder.rs:436-439 */
        put_u64(W + BSX_ST_W_PREV_BLOCK, prev_block);
        put_u64(W + BSX_ST_W_NEXT_BLOCK, next_block);
        Bo[0] = (uint8_t)c_prev; Bo[1] = (uint8_t)c_height; Bo[2] = (uint8_t)c_chain; Bo[3] = (uint8_t)c_sigs; Bo[4] = (uint8_t)c_vh;
        Bo[5] = (uint8_t)c_nvh; Bo[6] = (uint8_t)c_lb; Bo[7] = (uint8_t)cr.two_thirds_ok; Bo[8] = (uint8_t)cr.power_overflow;
        Bo[9] = memcmp(pr + 128 + 128, prev_hash, 32) == 0;        /* builder.rs:434 (A10): data_hash proof root == prev_header_hash */
    }
    return st ? st : rc;
}

int orc_next_header(const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                    uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                    uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit) {
    return orc_next_header_w(input40, prev_header, next_header, latest_block, next_validators, v_max, chain_id, chain_id_len,
                             output64, out_commit, NULL);
}

/* ---------------------------------------------------------------- mode-S fold (checker of bsx_dev_verify_commits' d_fold)
 * The unit of the mode-S all-gather: a rank's slice of per-header commit results (next_header.rs:25-47 per header) folded
 * into one 128-byte record.  digest(c) = inner(leaf(result bytes 0..64), leaf(result bytes 64..84 ‖ u32 LE index ‖ 40 zero
 * bytes)); root = binary tree of inner nodes over the digests padded with all-zero digests to a power of two. */
void orc_commit_fold(const bsx_commit_result* res, uint32_t n, uint32_t first_index, bsx_commit_fold* out) {
    uint32_t P = 1;
    while (P < n) P *= 2;
    uint8_t* d = calloc((size_t)P, 32);
    memset(out, 0, sizeof *out);
    out->first_failing = 0xffffffffu;
    for (uint32_t c = 0; c < n; c++) {
        const uint8_t* p = (const uint8_t*)&res[c];
        uint8_t a[32], b[32], t[64];
        orc_leaf_hash(p, 64, a);
        memset(t, 0, sizeof t);
        memcpy(t, p + 64, 20);
        const uint32_t idx = first_index + c;
        memcpy(t + 20, &idx, 4);
        orc_leaf_hash(t, 64, b);
        orc_inner_hash(a, b, d + 32 * (size_t)c);
        const int good = res[c].two_thirds_ok && !res[c].n_bad_signature && !res[c].n_bad_message && !res[c].power_overflow;
        out->n_ok += good ? 1 : 0;
        out->n_signatures_ok += res[c].n_signed - res[c].n_bad_signature;
        if (!good && out->first_failing == 0xffffffffu) out->first_failing = idx;
    }
    for (uint32_t w = P / 2; w >= 1; w /= 2)
        for (uint32_t i = 0; i < w; i++) {
            uint8_t nd[32];
            orc_inner_hash(d + 64 * (size_t)i, d + 64 * (size_t)i + 32, nd);
            memcpy(d + 32 * (size_t)i, nd, 32);
        }
    memcpy(out->root, d, 32);
    out->n_commits = n;
    out->first_index = first_index;
    free(d);
}
