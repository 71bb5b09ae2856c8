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

static void validators_hash(const bsx_validator* vals, uint32_t v_max, uint8_t out[32]) {
    uint32_t P = next_pow2(v_max);
    uint8_t(*nodes)[32] = calloc(P, 32);
    uint8_t* en = calloc(P, 1);
    for (uint32_t i = 0; i < v_max; i++) {
        uint8_t leaf[BSX_VALIDATOR_LEAF_MAX];
        int n = orc_validator_leaf(vals[i].pubkey, vals[i].voting_power, leaf);
        orc_leaf_hash(leaf, (size_t)n, nodes[i]);
        en[i] = vals[i].enabled != 0;
    }
    for (uint32_t i = v_max; i < P; i++) { /* padding slots: leaf of an all-zero validator, disabled */
        uint8_t leaf[BSX_VALIDATOR_LEAF_MAX], zero[32] = {0};
        int n = orc_validator_leaf(zero, 0, leaf);
        orc_leaf_hash(leaf, (size_t)n, nodes[i]);
    }
    for (uint32_t n = P; n > 1; n /= 2)
        for (uint32_t i = 0; i < n; i += 2) {
            uint8_t inner[32];
            orc_inner_hash(nodes[i], nodes[i + 1], inner);
            if (en[i] && en[i + 1]) memcpy(nodes[i / 2], inner, 32);
            else memcpy(nodes[i / 2], nodes[i], 32);
            en[i / 2] = en[i] || en[i + 1];
        }
    memcpy(out, nodes[0], 32);
    free(nodes);
    free(en);
}

void orc_verify_commit(const bsx_validator* vals, uint32_t v_max, const uint8_t header_hash[32], bsx_commit_result* out,
                       uint8_t* sig_ok) {
    memset(out, 0, sizeof *out);
    out->first_bad_signature = 0xffffffffu;
    unsigned __int128 exact_total = 0;
    for (uint32_t i = 0; i < v_max; i++) {
        const bsx_validator* v = &vals[i];
        uint8_t ok = 0;
        if (v->enabled) {
            out->n_enabled++;
            out->total_power += v->voting_power;
            exact_total += v->voting_power;
            if (v->is_signed) {
                out->n_signed++;
                uint8_t h[32];
                orc_sha512_challenge(v, h, NULL);
                int sig = orc_ed25519_verify_h(v->pubkey, v->signature, h);
                uint32_t off = (v->message_len > 12 && v->message[12] == 0x19) ? 25 : 16;
                int msg = v->message_len <= BSX_VALIDATOR_MSG_MAX && v->message_len >= off + 32 &&
                          memcmp(v->message + off, header_hash, 32) == 0;
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
    }
    validators_hash(vals, v_max, out->validators_hash);
    /* Tendermint caps a set's total at MaxTotalVotingPower = MaxInt64 / 8; beyond it the u64 sums may have wrapped */
    out->power_overflow = exact_total > (unsigned __int128)BSX_MAX_TOTAL_VOTING_POWER;
    /* 3*signed > 2*total, in 128-bit to avoid overflow */
    out->two_thirds_ok = !out->power_overflow && (unsigned __int128)out->signed_power * 3 > (unsigned __int128)out->total_power * 2;
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
    if (chain_id_len > 50 || th->len[1] != chain_id_len + 2 || th->chain_id[0] != 0x0a || th->chain_id[1] != chain_id_len ||
        memcmp(th->chain_id + 2, chain_id, chain_id_len) != 0) header_assert = 1;
    bsx_commit_result cr, trc;
    uint8_t* ok = malloc(v_max);
    orc_verify_commit(target_validators, v_max, target_hash, &cr, ok);
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
    validators_hash(trusted_validators, v_max, trc.validators_hash);
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
    bsx_witness_layout L = bsx_map_layout(jb->B), R = bsx_reduce_layout();
    size_t csz = (size_t)jb->J * L.compact_stride + (size_t)(jb->J - 1) * R.compact_stride;
    uint8_t* compact = jb->with_witness ? malloc(csz) : NULL;
    /* one map job's worth of expanded elements at a time (3.6 MB at B = 64): same work, bounded memory per thread */
    size_t wel = L.n_elements > (size_t)(jb->J - 1) * R.n_elements ? L.n_elements : (size_t)(jb->J - 1) * R.n_elements;
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
int orc_next_header(const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                    uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                    uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit) {
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
    uint8_t* ok = (uint8_t*)malloc(v_max ? v_max : 1);
    orc_verify_commit(next_validators, v_max, hn_, &cr, ok);
    free(ok);
    if (out_commit) *out_commit = cr;
    int st = BSX_OK;
    if (cr.power_overflow) return BSX_ERR_BAD_ARG;
    if (memcmp(hp, prev_hash, 32) != 0) st = BSX_ERR_ASSERT;
    uint8_t hf[12];
    int hl = 0;
    hf[hl++] = 0x08;
    for (uint64_t hv = next_block;; hv >>= 7) { if (hv >= 0x80) hf[hl++] = (uint8_t)(hv | 0x80); else { hf[hl++] = (uint8_t)hv; break; } }
    if (!st && (next_header->len[BSX_BLOCK_HEIGHT_INDEX] != hl || memcmp(next_header->height, hf, (size_t)hl) != 0)) st = BSX_ERR_ASSERT;
    if (!st && (chain_id_len > 50 || next_header->len[1] != chain_id_len + 2 || next_header->chain_id[0] != 0x0a ||
                next_header->chain_id[1] != chain_id_len || memcmp(next_header->chain_id + 2, chain_id, chain_id_len) != 0))
        st = BSX_ERR_ASSERT;   /* builder.step is called with C::CHAIN_ID_BYTES (next_header.rs:32-33) */
    if (!st && (cr.n_bad_signature || cr.n_bad_message)) st = BSX_ERR_BAD_SIGNATURE;
    if (!st && (next_header->len[7] != 34 || memcmp(next_header->hash[2] + 2, cr.validators_hash, 32) != 0)) st = BSX_ERR_ASSERT;
    if (!st && (prev_header->len[8] != 34 || memcmp(prev_header->hash[3] + 2, cr.validators_hash, 32) != 0)) st = BSX_ERR_ASSERT;
    if (!st && (next_header->len[BSX_LAST_BLOCK_ID_INDEX] < 34 || memcmp(next_header->last_block_id + 2, hp, 32) != 0)) st = BSX_ERR_ASSERT;
    if (!st && !cr.two_thirds_ok) st = BSX_ERR_VOTING_POWER;
    uint8_t dc[32];
    const int rc = orc_prove_next_header_data_commitment(prev_block, prev_hash, next_block, prev_header, latest_block, dc);   /* :38-42 */
    if (rc != BSX_OK && rc != BSX_ERR_ASSERT) return rc;
    memcpy(output64, hn_, 32);                                     /* :44 */
    memcpy(output64 + 32, dc, 32);                                 /* :45 */
    return st ? st : rc;
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
