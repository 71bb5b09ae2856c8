// pipeline.hip — the batched header_range pipeline behind the C ABI (include/bsx.h, bsx_pipeline_*).
//
// The object that produces the headline throughput: R independent header_range instances per step, inputs resident in
// HBM, cut into chunks that run on their own HIP streams in complementary phases (the integer-ALU-bound hashing of one chunk
// beside the HBM-bound witness expansion of the other), the commit check of every chunk on a side stream, double-buffered by
// step parity so that consecutive steps need no join.  Reference dataflow per range: CombinedSkipCircuit::define
// (circuits/header_range.rs:32-59) = builder.skip (:42-48) + prove_data_commitment (circuits/builder.rs:273-409, map closure
// :305-336 with the hint of circuits/data_commitment.rs:18-45 -> circuits/input.rs:149-271, reduce :337-395).
//
// Host code only: buffer ownership, stream/event choreography, argument validation.  All arithmetic is in kernels_*.hip.
// No environment variables, no Python, no torch: a Rust caller binds exactly this.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bsx_layout.h"
#include "api_internal.h"
#include "kernels.h"
#include "keycache.h"

using bsxapi::fail;
using bsxapi::pow2;
using bsxapi::use;

namespace {

constexpr size_t PIPE_POOL = 16;             // GPU_MAX_HW_QUEUES: more streams than that share queues anyway

struct TimingSlot {
    hipEvent_t ev[8];       // prove_subchain [0,1], map expansion [2,3], Poseidon commitment [4,5], the cross-GPU exchange [6,7] (its stream)
    bool sub, exp, caps, xchg;
};

constexpr uint32_t PIPE_HINT_LDS_PAD_COMPACT = 65536;   // compact pipeline: two hint workgroups per CU (bsxk_assemble_inputs) ...
constexpr uint64_t PIPE_HINT_THROTTLE_FROM_JOBS = 6144; // ... for chunks whose header hashing is long enough to hide the stretched hint
                                                        // (256 ranges: 1.22 -> 1.16 ms per step; 64 / 128 ranges lose 3 % with it)
constexpr uint64_t PIPE_LATENCY_FORM_BELOW = 16384;     // signatures per chunk: at or below, the commit check's latency form
constexpr uint32_t PIPE_KEY_ROWS_MAX = 8192;            // most rows of a chunk's fixed-key table (5.8 MB each: 47 GB of the 288)
struct Chunk {
    uint32_t R = 0, RT = 0;                  // owned ranges / ranges whose job slice this rank computes, in this chunk
    uint64_t nh_main = 0, nh_skip = 0, nh_all = 0;
    hipStream_t main = nullptr, side = nullptr, xchg = nullptr, copy = nullptr;
    // inputs
    uint8_t *headers_all = nullptr, *ranges = nullptr, *latest = nullptr;
    uint8_t *skip_ranges = nullptr, *skip_ranges_side = nullptr, *validators = nullptr, *trusted = nullptr;
    uint32_t* target_idx = nullptr;
    // per-header digests
    uint8_t *hashes_all = nullptr, *dh_aunts = nullptr, *lb_aunts = nullptr, *paths = nullptr;
    // map / reduce
    uint32_t* status = nullptr;              // [0] header, [1] assemble
    uint8_t *compact = nullptr, *records = nullptr, *partial = nullptr, *red_compact_local = nullptr, *gathered = nullptr;
    uint8_t *red_compact_top = nullptr, *results = nullptr, *output64 = nullptr;
    uint32_t* range_status = nullptr;
    // commit check
    uint8_t *h = nullptr, *ok = nullptr, *commit_res = nullptr, *trusted_res = nullptr, *keytable = nullptr, *rdec = nullptr, *ed_scratch = nullptr;
    uint32_t* skip_status = nullptr;
    uint8_t *target_hashes_pp[2] = {nullptr, nullptr}, *skip_hashes_pp[2] = {nullptr, nullptr}, *skip_headers_pp[2] = {nullptr, nullptr};
    // outputs
    uint64_t *witness_map = nullptr, *witness_red_local = nullptr, *witness_red_top = nullptr, *trees = nullptr;
    bool witness_map_vmm = false;
    uint64_t n_map_el = 0, n_red_local_el = 0, n_red_top_el = 0, trees_words = 0;
    // COMMIT / SKIP units of the owned ranges (builder.skip's variables, include/bsx_layout.h): compact images the commit chain's
    // kernels fill, their Goldilocks expansion and / or Poseidon trees
    uint8_t *commit_compact = nullptr, *skip_compact = nullptr;
    uint64_t *witness_commit = nullptr, *witness_skip = nullptr, *trees_commit = nullptr, *trees_skip = nullptr;
    bool fin_recorded = false, wit_pending = false, units_done_valid = false;
    uint64_t n_key_mismatch = 0;             // active slots the fixed-key table has no row for (bsx_pipeline_upload): the generic kernel's
    // the chunk's fixed-key table: rows keyed by public key (keycache.h), sized at upload for the distinct keys of the chunk's ranges
    bsx_keycache kc;
    uint32_t keytable_rows = 0;
    uint8_t* rowkeys = nullptr;              // device: the rows' key records as the build kernel reads them (bsx_validator each)
    uint32_t* rows = nullptr;                // device: [R][V] table row of every slot; unused while the map is the identity
    bool rows_identity = true;
    hipEvent_t ev_finalized = nullptr, ev_units_done = nullptr;
    size_t compact_bytes = 0, records_bytes = 0, headers_bytes = 0;
    // state
    int parity = 0;
    bool commit_done_valid[2] = {false, false}, inputs_consumed_valid = false, h2d_pending = false;
    hipEvent_t ev_sync = nullptr, ev_merkle = nullptr, ev_fill = nullptr, ev_fin = nullptr, ev_inputs_consumed = nullptr, ev_h2d = nullptr;
    hipEvent_t ev_commit_done[2] = {nullptr, nullptr}, ev_hash_tok = nullptr, ev_expand_tok = nullptr, ev_x_in = nullptr, ev_x_out = nullptr;
    uint8_t* host_image = nullptr;           // page-locked image of headers_all (input streaming)
    std::vector<TimingSlot> timing;
    size_t timing_used = 0;
};

}  // namespace

struct bsx_pipeline {
    bsx_ctx* ctx = nullptr;
    bsx_pipeline_config cfg{};
    uint32_t J = 0, B = 0, V = 0, R = 0, E = 0, K = 1, Rc = 0, rank = 0, world = 1, jf = 0, jc = 0;
    uint64_t step_index = 0;                 // steps enqueued so far: step i runs on buffer set i % K
    uint32_t last_set = 0;
    uint64_t hpr = 0, hfr = 0;
    bool with_witness = false, with_commit = false, with_caps = false, keyed = false, commit_beside_hash = false, fused_hint = true;
    uint32_t subchain_flags = 0, merkle_wgs = 0;
    uint32_t leaf_len = 0, cap_height = 0, n_leaves = 0;
    uint64_t tree_digests = 0;
    bsx_witness_layout ml{}, rl{}, cl{}, sl{};
    bool units = false;                      // COMMIT / SKIP units are produced (commit check on, and a witness or caps wanted)
    uint32_t n_leaves_cm = 0, cap_h_cm = 0, n_leaves_sk = 0, cap_h_sk = 0;
    uint64_t tree_digests_cm = 0, tree_digests_sk = 0;
    std::vector<Chunk> chunks;
    // Every stream a chunk's hashing / expansion (main) and commit check (side) may run on, created AND used once at
    // bsx_pipeline_create, in this order: HIP binds a stream to a hardware queue at its first command, and WHICH queues the hot
    // streams sit on decides how well the chunks' phases overlap (bsx_pipeline_autotune)
    std::vector<hipStream_t> pool;
    std::vector<uint32_t> assign;            // pool index of chunk i's main (2 i) and side (2 i + 1) stream
    uint8_t* touch = nullptr;                // 256 B the pool's first commands write + 8 B per rank (autotune's agreement on a step count)
    std::vector<void*> allocs;               // hipMalloc'ed blocks
    std::vector<void*> host_allocs;          // hipHostMalloc'ed blocks
    hipEvent_t hash_token = nullptr, expand_token = nullptr;     // aliases of a chunk's ev_hash_tok / ev_expand_tok
    hipEvent_t merkle_token = nullptr;       // compact mode: the previous k_header_merkle launch (any chunk, any set)
    Chunk* pending_verify = nullptr;
    bsx_allgather_fn allgather = nullptr;
    void* allgather_user = nullptr;
    void* rccl_comm = nullptr;               // bsx_pipeline_set_rccl: the all-gather is ncclAllGather on this communicator
    bool timing_on = false, streaming = false, uploaded = false, compact_tokens = false;
};

namespace {

int dalloc(bsx_pipeline* p, size_t bytes, void** out) {
    if (bytes < 256) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(BSX_ERR_HIP, "bsx_pipeline: hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(q); return fail(BSX_ERR_HIP, "bsx_pipeline: hipMemset: %s", hipGetErrorString(e)); }
    p->allocs.push_back(q);
    *out = q;
    return BSX_OK;
}
template <typename T> int dalloc_t(bsx_pipeline* p, size_t bytes, T** out) {
    void* q = nullptr;
    RET(dalloc(p, bytes, &q));
    *out = static_cast<T*>(q);
    return BSX_OK;
}
int new_event(hipEvent_t* e, bool timing = false) {
    HIPCHK(hipEventCreateWithFlags(e, timing ? hipEventDefault : hipEventDisableTiming));
    return BSX_OK;
}
int new_stream(hipStream_t* s) {
    HIPCHK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    return BSX_OK;
}
// `waiter` continues only after everything enqueued on `on` so far
int stream_wait_stream(hipStream_t waiter, hipStream_t on, hipEvent_t scratch) {
    HIPCHK(hipEventRecord(scratch, on));
    HIPCHK(hipStreamWaitEvent(waiter, scratch, 0));
    return BSX_OK;
}

int create_chunk(bsx_pipeline* p, Chunk& c) {
    const uint32_t jc = p->jc, V = p->V, world = p->world;
    c.R = p->Rc;
    c.RT = p->Rc * world;
    const uint32_t R = c.R, RT = c.RT;
    c.nh_main = (uint64_t)RT * p->hpr;
    c.nh_skip = p->with_commit ? (uint64_t)R * 2 : 0;
    c.nh_all = c.nh_main + c.nh_skip;
    {
        const size_t idx = (size_t)(&c - p->chunks.data());
        c.main = p->pool[p->assign[2 * idx]];
        c.side = p->pool[p->assign[2 * idx + 1]];
    }
    if (world > 1) RET(new_stream(&c.xchg));
    for (hipEvent_t* e : {&c.ev_sync, &c.ev_merkle, &c.ev_fill, &c.ev_fin, &c.ev_inputs_consumed, &c.ev_h2d, &c.ev_commit_done[0], &c.ev_commit_done[1],
                          &c.ev_hash_tok, &c.ev_expand_tok, &c.ev_x_in, &c.ev_x_out, &c.ev_finalized, &c.ev_units_done})
        RET(new_event(e));
    // one header block per step: this rank's slice of every range, then (owned ranges) the trusted and the target header of
    // the commit check as a 2-header block per range — hashed by ONE k_header_merkle launch
    c.headers_bytes = c.nh_all * sizeof(bsx_header);
    RET(dalloc_t(p, c.headers_bytes, &c.headers_all));
    RET(dalloc_t(p, c.nh_all * 32, &c.hashes_all));
    RET(dalloc_t(p, c.nh_all * 128, &c.dh_aunts));
    RET(dalloc_t(p, c.nh_all * 128, &c.lb_aunts));
    if (p->fused_hint) RET(dalloc_t(p, c.nh_all * BSX_HEADER_PATH_BYTES, &c.paths));
    RET(dalloc_t(p, (size_t)RT * sizeof(bsx_shared_ctx), &c.ranges));
    RET(dalloc_t(p, (size_t)RT * 8, &c.latest));
    RET(dalloc_t(p, 32, &c.status));
    c.compact_bytes = (size_t)RT * jc * p->ml.compact_stride;
    RET(dalloc_t(p, c.compact_bytes, &c.compact));
    c.records_bytes = (size_t)RT * jc * sizeof(bsx_subchain);
    RET(dalloc_t(p, c.records_bytes, &c.records));
    RET(dalloc_t(p, (size_t)RT * 128, &c.partial));
    const size_t n_local_nodes = (size_t)RT * (jc > 1 ? jc - 1 : 0);
    RET(dalloc_t(p, n_local_nodes * p->rl.compact_stride, &c.red_compact_local));
    RET(dalloc_t(p, (size_t)world * RT * 128, &c.gathered));
    RET(dalloc_t(p, (size_t)R * (world - 1) * p->rl.compact_stride, &c.red_compact_top));
    RET(dalloc_t(p, (size_t)R * 128, &c.results));
    RET(dalloc_t(p, (size_t)R * 64, &c.output64));
    RET(dalloc_t(p, (size_t)R * 4, &c.range_status));
    RET(dalloc_t(p, (size_t)R * sizeof(bsx_shared_ctx), &c.skip_ranges));
    RET(dalloc_t(p, (size_t)R * sizeof(bsx_shared_ctx), &c.skip_ranges_side));
    RET(dalloc_t(p, (size_t)R * 4, &c.target_idx));
    {
        std::vector<uint32_t> ones(R, 1u);                 // the target is header 1 of each (trusted, target) block
        HIPCHK(hipMemcpy(c.target_idx, ones.data(), (size_t)R * 4, hipMemcpyHostToDevice));
    }
    RET(dalloc_t(p, (size_t)R * V * sizeof(bsx_validator), &c.validators));
    RET(dalloc_t(p, (size_t)R * V * sizeof(bsx_validator), &c.trusted));
    RET(dalloc_t(p, (size_t)R * V * 32, &c.h));
    RET(dalloc_t(p, (size_t)R * V, &c.ok));
    RET(dalloc_t(p, (size_t)R * sizeof(bsx_commit_result), &c.commit_res));
    RET(dalloc_t(p, (size_t)R * sizeof(bsx_commit_result), &c.trusted_res));
    RET(dalloc_t(p, (size_t)R * 4, &c.skip_status));
    for (int q = 0; q < 2; q++) {
        RET(dalloc_t(p, (size_t)R * 32, &c.target_hashes_pp[q]));
        RET(dalloc_t(p, (size_t)R * 64, &c.skip_hashes_pp[q]));
        RET(dalloc_t(p, (size_t)R * 2 * sizeof(bsx_header), &c.skip_headers_pp[q]));
    }
    if (p->with_commit && p->keyed) {
        // the table itself is allocated by bsx_pipeline_upload, which knows how many distinct keys the chunk's ranges hold
        RET(dalloc_t(p, (size_t)R * V * 4, &c.rows));
        // beside an expansion (ALU to spare, one step to finish in) and for a FEW ranges (nothing to fill the GPU with anyway: the
        // commit chain's latency is the step) the latency form — R decoded beside the challenges, 8 / 16 lanes per signature, projective
        // compare: ~0.1 ms where the least-work form (one lane per signature, batch inversion) is a 0.7 ms chain whatever the batch
        if (p->with_witness || (uint64_t)R * V <= PIPE_LATENCY_FORM_BELOW) RET(dalloc_t(p, bsxk_ed25519_rdec_bytes((uint64_t)R * V), &c.rdec));
        else RET(dalloc_t(p, bsxk_ed25519_scratch_bytes((uint64_t)R * V), &c.ed_scratch));
    }
    c.n_map_el = (uint64_t)RT * jc * p->ml.n_elements;
    c.n_red_local_el = (uint64_t)n_local_nodes * p->rl.n_elements;
    c.n_red_top_el = (uint64_t)R * (world - 1) * p->rl.n_elements;
    if (p->with_witness) {
        // the map-job image (14.7 GB per chunk at the bench shape) comes from the HIP virtual-memory API: one physical handle,
        // deterministic placement (bsx_dev_alloc)
        const uint64_t bytes = (c.n_map_el + 2) * 8;
        static const uint64_t vmm_min = (uint64_t)bsx_knob("BSX_VMM_MIN_MB", 64) << 20;      // experiments build: where VMM backing starts
        if (bytes >= vmm_min) {
            void* q = nullptr;
            RET(bsx_dev_alloc(p->ctx, bytes, &q));
            c.witness_map = static_cast<uint64_t*>(q);
            c.witness_map_vmm = true;
        } else {
            RET(dalloc_t(p, bytes, &c.witness_map));
        }
        RET(dalloc_t(p, (c.n_red_local_el + 2) * 8, &c.witness_red_local));
        RET(dalloc_t(p, (c.n_red_top_el + 2) * 8, &c.witness_red_top));
    }
    if (p->with_caps) {
        c.trees_words = (uint64_t)RT * jc * p->tree_digests * 4;
        RET(dalloc_t(p, c.trees_words * 8, &c.trees));
    }
    if (p->units) {
        RET(dalloc_t(p, (size_t)R * p->cl.compact_stride, &c.commit_compact));     // zeroed: bytes no kernel writes stay zero
        RET(dalloc_t(p, (size_t)R * p->sl.compact_stride, &c.skip_compact));
        if (p->with_witness) {
            RET(dalloc_t(p, ((uint64_t)R * p->cl.n_elements + 2) * 8, &c.witness_commit));
            RET(dalloc_t(p, ((uint64_t)R * p->sl.n_elements + 2) * 8, &c.witness_skip));
        }
        if (p->with_caps) {
            RET(dalloc_t(p, (uint64_t)R * p->tree_digests_cm * 32, &c.trees_commit));
            RET(dalloc_t(p, (uint64_t)R * p->tree_digests_sk * 32, &c.trees_skip));
        }
    }
    return BSX_OK;
}

// at most PIPE_TIMING_SLOTS un-read chunk-steps per chunk (6 events each): a caller that leaves timing on and never calls
// bsx_pipeline_timing stops recording there instead of growing without bound
constexpr size_t PIPE_TIMING_SLOTS = 4096;
TimingSlot* timing_slot(Chunk& c) {
    if (c.timing_used >= PIPE_TIMING_SLOTS) return nullptr;
    if (c.timing_used == c.timing.size()) {
        TimingSlot t{};
        for (auto& e : t.ev)
            if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) return nullptr;
        c.timing.push_back(t);
    }
    TimingSlot* t = &c.timing[c.timing_used++];
    t->sub = t->exp = t->caps = t->xchg = false;
    return t;
}

// Stage 3 for the owned ranges on stream `st` (builder.skip, header_range.rs:42-48).
//   prep:   SHA-512 challenges + per-validator tables (small, memory-latency sensitive) — beside the hashing
//   verify: signature checks, tallies, skip conditions (integer ALU) — beside the expansion
// header-field inclusion proofs of every owned range's target header (chain id, height, validators_hash) and trusted header
// (validators_hash) into the SKIP units; `skip_headers` = the (trusted, target) header pairs
int field_proofs(bsx_pipeline* p, Chunk& c, hipStream_t st, const uint8_t* skip_headers) {
    bsxk_field_proofs_args fa{};
    fa.n_items = c.R; fa.headers = reinterpret_cast<const bsx_header*>(skip_headers); fa.headers_per_item = 2; fa.target_idx = c.target_idx;
    fa.unit = bsxk_unit(c.skip_compact, p->sl); fa.n_proofs = BSX_SK_N_PROOFS; fa.zero_paths = p->ctx->zero_paths;
    static const uint8_t hsel[BSX_SK_N_PROOFS] = {1, 1, 1, 0}, fld[BSX_SK_N_PROOFS] = {1, BSX_BLOCK_HEIGHT_INDEX, 7, 7};
    for (uint32_t k = 0; k < BSX_SK_N_PROOFS; k++)
        fa.proofs[k] = bsxk_proof_spec{hsel[k], fld[k], (uint16_t)bsx_sk_proof_cap(k), bsx_sk_off_proof(p->V, k), BSX_SK_W_LEAF_LEN + k, 0u};
    HIPCHK(bsxk_field_proofs(st, &fa));
    return BSX_OK;
}

int commit_part(bsx_pipeline* p, Chunk& c, hipStream_t st, bool prep, bool verify) {
    const uint32_t R = c.R, V = p->V;
    const uint64_t n = (uint64_t)R * V;
    auto* vals = reinterpret_cast<const bsx_validator*>(c.validators);
    auto* trs = reinterpret_cast<const bsx_validator*>(c.trusted);
    auto* cres = reinterpret_cast<bsx_commit_result*>(c.commit_res);
    auto* tres = reinterpret_cast<bsx_commit_result*>(c.trusted_res);
    // with units on, the kernels of the chain leave the variables they hold in the ranges' COMMIT / SKIP units as they go
    const bsxk_unit_dst cwd = bsxk_unit(c.commit_compact, p->cl), swd = bsxk_unit(c.skip_compact, p->sl), swd_t = bsxk_unit(c.skip_compact, p->sl, 1);
    const bsxk_unit_dst *cwp = p->units ? &cwd : nullptr, *swp = p->units ? &swd : nullptr, *swtp = p->units ? &swd_t : nullptr;
    if (prep) {
        // everything that needs nothing from this step's hashing: challenges, R decoded for the projective comparison, the
        // key-table check, the trusted set's hash and power sum — off the critical chain (verify -> tally -> skip conditions)
        HIPCHK(bsxk_sha512_challenge(st, vals, n, c.h, nullptr, V, cwp));
        // the fixed-key tables were validated against the uploaded validator sets (and rebuilt where a key changed) by
        // bsx_pipeline_upload: validator sets only change there, so a step launches neither the key compare nor the build
        if (p->keyed && c.rdec) HIPCHK(bsxk_ed25519_decode_r(st, vals, n, c.rdec));
        HIPCHK(bsxk_commit_tally(st, trs, R, V, nullptr, nullptr, tres, swtp));
        // resident inputs: the header pairs do not change under the prep phase; streamed inputs: see below
        if (p->units && !p->streaming) RET(field_proofs(p, c, st, c.headers_all + c.nh_main * sizeof(bsx_header)));
    }
    if (!verify) return BSX_OK;
    if (p->keyed)   // latency form beside an expansion (ALU to spare, one step to finish in); least-work form in the compact pipeline
        HIPCHK(bsxk_ed25519_verify_keyed(st, vals, c.h, n, V, c.keytable, c.keytable_rows, p->ctx->btab, c.ok, c.ed_scratch, c.rdec ? c.rdec : BSXK_ED_THROUGHPUT,
                                         (int64_t)c.n_key_mismatch, c.rows_identity ? nullptr : c.rows));
    else
        HIPCHK(bsxk_ed25519_verify(st, vals, c.h, n, c.ok));
    HIPCHK(bsxk_commit_tally(st, vals, R, V, c.target_hashes_pp[c.parity], c.ok, cres, cwp));
    // streamed inputs: headers_all is overwritten early in the next step while this check may still run -> private copy
    const uint8_t* skip_headers = p->streaming ? c.skip_headers_pp[c.parity] : c.headers_all + c.nh_main * sizeof(bsx_header);
    if (p->units && p->streaming) RET(field_proofs(p, c, st, skip_headers));
    HIPCHK(bsxk_skip_check(st, R, V, reinterpret_cast<const bsx_shared_ctx*>(c.skip_ranges_side), reinterpret_cast<const bsx_header*>(skip_headers), 2,
                           c.skip_hashes_pp[c.parity], vals, trs, c.ok, cres, tres, c.skip_status, nullptr, c.target_idx,
                           p->cfg.chain_id_len ? p->cfg.chain_id : nullptr, p->cfg.chain_id_len, swp));
    HIPCHK(hipEventRecord(c.ev_commit_done[c.parity], st));
    c.commit_done_valid[c.parity] = true;
    c.wit_pending = p->units;
    return BSX_OK;
}

// The COMMIT / SKIP units of the chunk's current step are complete once BOTH the commit chain (side stream) and finalize (main
// stream: the data commitment, header_range.rs:58) are through: expand them / hash them on the side stream, behind both.
int launch_commit_witness(bsx_pipeline* p, Chunk& c) {
    if (!p->units || !c.wit_pending || !c.fin_recorded) return BSX_OK;
    c.wit_pending = false;
    hipStream_t st = c.side;
    HIPCHK(hipStreamWaitEvent(st, c.ev_finalized, 0));
    if (p->with_witness) {
        HIPCHK(bsxk_expand_witness(st, &p->cl, c.R, c.commit_compact, c.witness_commit));
        HIPCHK(bsxk_expand_witness(st, &p->sl, c.R, c.skip_compact, c.witness_skip));
    }
    if (p->with_caps) {
        HIPCHK(bsxk_leaf_hashes(st, &p->cl, c.R, c.commit_compact, nullptr, p->leaf_len, p->n_leaves_cm, 1, 4 * p->tree_digests_cm, c.trees_commit));
        HIPCHK(bsxk_merkle_caps(st, c.trees_commit, c.R, 4 * p->tree_digests_cm, p->n_leaves_cm, p->cap_h_cm));
        HIPCHK(bsxk_leaf_hashes(st, &p->sl, c.R, c.skip_compact, nullptr, p->leaf_len, p->n_leaves_sk, 1, 4 * p->tree_digests_sk, c.trees_skip));
        HIPCHK(bsxk_merkle_caps(st, c.trees_skip, c.R, 4 * p->tree_digests_sk, p->n_leaves_sk, p->cap_h_sk));
    }
    HIPCHK(hipEventRecord(c.ev_units_done, st));
    c.units_done_valid = true;
    return BSX_OK;
}

bool commit_active(const bsx_pipeline* p, const Chunk& c) { return p->with_commit && c.R && c.nh_skip; }

// signature checks, tallies and skip conditions of the chunk's current step on its side stream; after_event (optional)
// delays them — the other chunk's header-hashing event, so that this ALU work runs beside that chunk's memory-leaning kernels
int launch_verify(bsx_pipeline* p, Chunk& c, hipEvent_t after_event) {
    if (!commit_active(p, c) || p->commit_beside_hash) return BSX_OK;
    HIPCHK(hipStreamWaitEvent(c.side, c.ev_fill, 0));
    if (after_event) HIPCHK(hipStreamWaitEvent(c.side, after_event, 0));
    RET(commit_part(p, c, c.side, false, true));
    return launch_commit_witness(p, c);
}

// stages 1-5 + local fold: everything before the cross-GPU exchange
int step_local(bsx_pipeline* p, Chunk& c, TimingSlot* ts) {
    const uint32_t B = p->B, jc = p->jc, RT = c.RT, R = c.R;
    hipStream_t st = c.main;
    if (c.h2d_pending) {                      // streamed inputs: this step's headers arrive on the copy stream
        HIPCHK(hipStreamWaitEvent(st, c.ev_h2d, 0));
        c.h2d_pending = false;
    }
    HIPCHK(hipMemsetAsync(c.status, 0, 32, st));
    c.fin_recorded = false;
    c.wit_pending = false;
    const bool commit = commit_active(p, c);
    if (commit && !p->commit_beside_hash) {
        // challenges + per-validator tables need nothing from this step: start them right away beside the hashing
        RET(stream_wait_stream(c.side, st, c.ev_sync));
        RET(commit_part(p, c, c.side, true, false));
    }
    // Without an expansion to hide behind, the step is one ALU-bound kernel (k_header_merkle: 41 of the 45 compressions per
    // header) and a chain of short, latency-bound ones.  Chunks / buffer sets left to themselves drift into the same phase
    // (two header hashings sharing the GPU, then two chains leaving it idle); the token lets exactly ONE header hashing run
    // at a time, back to back across chunks, sets and steps, with the other chunks' chains filling in beside it.
    if (p->compact_tokens && p->merkle_token) HIPCHK(hipStreamWaitEvent(st, p->merkle_token, 0));
    HIPCHK(bsxk_header_merkle(st, reinterpret_cast<const bsx_header*>(c.headers_all), commit ? c.nh_all : c.nh_main, c.hashes_all, c.dh_aunts,
                              c.lb_aunts, c.paths, c.status, p->merkle_wgs, p->compact_tokens ? 1u : 0u));
    HIPCHK(hipEventRecord(c.ev_merkle, st));
    if (p->compact_tokens) p->merkle_token = c.ev_merkle;
    if (commit) {
        c.parity ^= 1;
        if (c.commit_done_valid[c.parity]) HIPCHK(hipStreamWaitEvent(st, c.ev_commit_done[c.parity], 0));   // the check two steps ago read these
        uint8_t* skip_hashes = c.hashes_all + c.nh_main * 32;
        HIPCHK(bsxk_fill_end_hash(st, R, reinterpret_cast<bsx_shared_ctx*>(c.skip_ranges), skip_hashes, 2, c.target_idx, c.target_hashes_pp[c.parity],
                                  c.skip_hashes_pp[c.parity], 0));
        if (p->streaming)
            HIPCHK(hipMemcpyAsync(c.skip_headers_pp[c.parity], c.headers_all + c.nh_main * sizeof(bsx_header), (size_t)R * 2 * sizeof(bsx_header),
                                  hipMemcpyDeviceToDevice, st));
        HIPCHK(hipEventRecord(c.ev_fill, st));
        if (p->commit_beside_hash) {
            RET(stream_wait_stream(c.side, st, c.ev_sync));
            RET(commit_part(p, c, c.side, true, true));
        }
    }
    // ctx.end_header_hash of every range := the hash of its target header (what builder.skip hands to prove_data_commitment,
    // header_range.rs:42-55) wherever that header lies in this rank's slice: the caller's value is not trusted
    HIPCHK(bsxk_fill_end_hash(st, RT, reinterpret_cast<bsx_shared_ctx*>(c.ranges), c.hashes_all, p->hpr, nullptr, nullptr, nullptr, p->hfr));
    static const long abl = bsx_knob("BSX_ABLATE", 0);      // experiments build, timing only: 1 = no hint, 2 = no prove_subchain, 3 = neither
    if (!(abl & 1))
    HIPCHK(bsxk_assemble_inputs(st, RT, p->J, B, p->jf, jc, B, reinterpret_cast<const bsx_shared_ctx*>(c.ranges), reinterpret_cast<const uint64_t*>(c.latest),
                                reinterpret_cast<const bsx_header*>(c.headers_all), p->hpr, p->hfr, c.hashes_all, c.dh_aunts, c.lb_aunts, c.compact,
                                c.status + 1, c.paths, p->ctx->zero_paths, (p->compact_tokens && (uint64_t)RT * jc >= PIPE_HINT_THROTTLE_FROM_JOBS) ? PIPE_HINT_LDS_PAD_COMPACT : 0u));
    HIPCHK(hipEventRecord(c.ev_inputs_consumed, st));       // headers_all may be overwritten from here on (input streaming)
    c.inputs_consumed_valid = true;
    if (ts) { HIPCHK(hipEventRecord(ts->ev[0], st)); }
    if (!(abl & 2))
    HIPCHK(bsxk_prove_subchain(st, RT, B, jc, reinterpret_cast<const bsx_shared_ctx*>(c.ranges), c.compact, reinterpret_cast<bsx_subchain*>(c.records),
                               p->subchain_flags));
    if (ts) { HIPCHK(hipEventRecord(ts->ev[1], st)); ts->sub = true; }
    HIPCHK(bsxk_reduce(st, RT, jc, reinterpret_cast<const bsx_subchain*>(c.records), jc, 1, reinterpret_cast<bsx_subchain*>(c.partial),
                       jc > 1 ? c.red_compact_local : nullptr));
    return BSX_OK;
}

int stream_inputs(bsx_pipeline* p, Chunk& c) {
    if (!p->streaming || !c.host_image) return BSX_OK;
    if (c.inputs_consumed_valid) HIPCHK(hipStreamWaitEvent(c.copy, c.ev_inputs_consumed, 0));
    HIPCHK(hipMemcpyAsync(c.headers_all, c.host_image, c.headers_bytes, hipMemcpyHostToDevice, c.copy));
    HIPCHK(hipEventRecord(c.ev_h2d, c.copy));
    c.h2d_pending = true;
    return BSX_OK;
}

int exchange_begin(bsx_pipeline* p, Chunk& c, TimingSlot* ts) {
    if (!p->allgather) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline: world > 1 needs bsx_pipeline_set_allgather before the first step");
    RET(stream_wait_stream(c.xchg, c.main, c.ev_x_in));
    // timed on the exchange stream itself, from "this rank's partial records are ready" to "every rank's have arrived": what the
    // collective costs while the GPUs are busy — including the wait for the slowest rank
    if (ts) HIPCHK(hipEventRecord(ts->ev[6], c.xchg));
    if (p->allgather(p->allgather_user, c.partial, c.gathered, (uint64_t)c.RT * 128, c.xchg) != 0)
        return fail(BSX_ERR_HIP, "bsx_pipeline: the caller's all-gather failed");
    if (ts) { HIPCHK(hipEventRecord(ts->ev[7], c.xchg)); ts->xchg = true; }
    HIPCHK(hipEventRecord(c.ev_x_out, c.xchg));
    return BSX_OK;
}
// top fold straight from the all-gather layout [rank][range]: record k of owned range r = gathered[k][rank*R + r]
int exchange_end(bsx_pipeline* p, Chunk& c, const uint8_t** out_results) {
    HIPCHK(hipStreamWaitEvent(c.main, c.ev_x_out, 0));
    const uint8_t* own = c.gathered + (size_t)p->rank * c.R * 128;
    HIPCHK(bsxk_reduce(c.main, c.R, p->world, reinterpret_cast<const bsx_subchain*>(own), 1, c.RT, reinterpret_cast<bsx_subchain*>(c.results),
                       c.red_compact_top));
    *out_results = c.results;
    return BSX_OK;
}

int finalize(bsx_pipeline* p, Chunk& c, const uint8_t* records) {
    const uint8_t* own_ranges = p->with_commit ? c.skip_ranges : c.ranges + (size_t)p->rank * c.R * sizeof(bsx_shared_ctx);
    // the SKIP units of the previous step must have been consumed before this step's data commitments go into them
    if (p->units && c.units_done_valid) HIPCHK(hipStreamWaitEvent(c.main, c.ev_units_done, 0));
    HIPCHK(bsxk_finalize(c.main, c.R, p->J, p->B, reinterpret_cast<const bsx_shared_ctx*>(own_ranges), reinterpret_cast<const bsx_subchain*>(records),
                         p->with_commit ? c.target_hashes_pp[c.parity] : nullptr, c.output64, c.range_status,
                         p->units ? c.skip_compact : nullptr, p->sl.compact_stride));
    if (p->units) {
        HIPCHK(hipEventRecord(c.ev_finalized, c.main));
        c.fin_recorded = true;
        RET(launch_commit_witness(p, c));
    }
    return BSX_OK;
}

// finalize + (commit verification on the side stream) + witness expansion / Poseidon commitment.
// result_records == nullptr: the exchange was only begun; top fold, finalize and the top reduce nodes' expansion run behind
// the map-job expansion, so the collective's latency hides beside this chunk's own expansion.
int step_final(bsx_pipeline* p, Chunk& c, const uint8_t* result_records, bool do_launch_verify, hipEvent_t wait_before_expand, TimingSlot* ts) {
    hipStream_t st = c.main;
    const bool late = result_records == nullptr;
    if (!late) RET(finalize(p, c, result_records));
    if (do_launch_verify) {
        // integer-ALU work: start it beside the HBM-bound expansion (i.e. once finalize is done), not beside the hashing
        HIPCHK(hipEventRecord(c.ev_fin, st));
        RET(launch_verify(p, c, c.ev_fin));
    }
    if (wait_before_expand) HIPCHK(hipStreamWaitEvent(st, wait_before_expand, 0));
    if (p->with_witness) {
        if (ts) HIPCHK(hipEventRecord(ts->ev[2], st));
        HIPCHK(bsxk_expand_witness(st, &p->ml, c.RT * p->jc, c.compact, c.witness_map));
        if (ts) { HIPCHK(hipEventRecord(ts->ev[3], st)); ts->exp = true; }
        if (p->jc > 1) HIPCHK(bsxk_expand_witness(st, &p->rl, c.RT * (p->jc - 1), c.red_compact_local, c.witness_red_local));
    }
    if (p->with_caps) {
        // Poseidon Merkle cap of every map-job witness straight from the compact bytes (elements generated on the fly)
        if (ts) HIPCHK(hipEventRecord(ts->ev[4], st));
        HIPCHK(bsxk_leaf_hashes(st, &p->ml, c.RT * p->jc, c.compact, nullptr, p->leaf_len, p->n_leaves, 1, 4 * p->tree_digests, c.trees));
        HIPCHK(bsxk_merkle_caps(st, c.trees, c.RT * p->jc, 4 * p->tree_digests, p->n_leaves, p->cap_height));
        if (ts) { HIPCHK(hipEventRecord(ts->ev[5], st)); ts->caps = true; }
    }
    if (p->E > 1) {                           // release the other chunk's expansion
        HIPCHK(hipEventRecord(c.ev_expand_tok, st));
        p->expand_token = c.ev_expand_tok;
    }
    if (late) {
        const uint8_t* res = nullptr;
        RET(exchange_end(p, c, &res));
        RET(finalize(p, c, res));
    }
    if (p->with_witness && p->world > 1)
        HIPCHK(bsxk_expand_witness(st, &p->rl, c.R * (p->world - 1), c.red_compact_top, c.witness_red_top));
    return BSX_OK;
}

int join_impl(bsx_pipeline* p) {
    if (p->pending_verify) {                  // no later chunk to wait for: launch the deferred checks now
        Chunk* c = p->pending_verify;
        p->pending_verify = nullptr;
        RET(launch_verify(p, *c, nullptr));
    }
    for (Chunk& c : p->chunks) {
        HIPCHK(hipStreamSynchronize(c.main));
        HIPCHK(hipStreamSynchronize(c.side));
        if (c.xchg) HIPCHK(hipStreamSynchronize(c.xchg));
        if (c.copy) HIPCHK(hipStreamSynchronize(c.copy));
    }
    return BSX_OK;
}

// ---- RCCL, bound at run time (no link-time dependency: a single-GPU host never loads it).  If the process already holds an RCCL
// (a Rust host linking librccl, PyTorch) its symbols are used, so that a communicator the HOST created can be handed in; otherwise
// librccl.so is opened.  Only the five entry points the one collective of the path needs (rccl.h, ROCm 7.2).
struct RcclApi {
    struct UniqueId { char b[128]; };                                                 // ncclUniqueId: passed BY VALUE to ncclCommInitRank
    int (*get_unique_id)(void* id128) = nullptr;                                      // ncclGetUniqueId(ncclUniqueId*)
    int (*comm_init_rank)(void** comm, int n, UniqueId id, int rank) = nullptr;
    int (*comm_destroy)(void* comm) = nullptr;
    int (*all_gather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool ok = false;
    std::string why;
};
RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        if (!dlsym(RTLD_DEFAULT, "ncclAllGather")) {
            // a copy the process already mapped (PyTorch loads its own with local visibility), then the system's
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
                if (h) break;
            }
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                if (h) break;
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            }
            if (!h) { a.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return a; }
        }
        auto sym = [&](const char* n) -> void* { void* q = h ? dlsym(h, n) : nullptr; return q ? q : dlsym(RTLD_DEFAULT, n); };
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(sym("ncclAllGather"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
        a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_gather;
        if (!a.ok) a.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        return a;
    }();
    return api;
}
const char* rccl_err(int rc) { return rccl().error_string ? rccl().error_string(rc) : "RCCL error"; }
constexpr int NCCL_UINT8 = 1;                 // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

int rccl_allgather_cb(void* user, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream) {
    bsx_pipeline* p = static_cast<bsx_pipeline*>(user);
    const int rc = rccl().all_gather(d_send, d_recv, (size_t)bytes_per_rank, NCCL_UINT8, p->rccl_comm, static_cast<hipStream_t>(stream));
    if (rc != 0) { fail(BSX_ERR_HIP, "ncclAllGather: %s", rccl_err(rc)); return 1; }
    return 0;
}

}  // namespace

extern "C" {

int bsx_rccl_get_unique_id(uint8_t out_id[128]) {
    if (!out_id) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!rccl().ok) return fail(BSX_ERR_UNSUPPORTED, "RCCL unavailable: %s", rccl().why.c_str());
    const int rc = rccl().get_unique_id(out_id);
    if (rc != 0) return fail(BSX_ERR_HIP, "ncclGetUniqueId: %s", rccl_err(rc));
    return BSX_OK;
}
int bsx_rccl_comm_init_rank(bsx_ctx* ctx, uint32_t world, const uint8_t id[128], uint32_t rank, void** out_comm) {
    RET(use(ctx));                                                                    // the communicator binds to the context's device
    if (!id || !out_comm || !world || rank >= world) return fail(BSX_ERR_BAD_ARG, "bsx_rccl_comm_init_rank: bad arguments");
    if (!rccl().ok) return fail(BSX_ERR_UNSUPPORTED, "RCCL unavailable: %s", rccl().why.c_str());
    RcclApi::UniqueId u;
    memcpy(u.b, id, 128);
    void* comm = nullptr;
    const int rc = rccl().comm_init_rank(&comm, (int)world, u, (int)rank);
    if (rc != 0) return fail(BSX_ERR_HIP, "ncclCommInitRank: %s", rccl_err(rc));
    *out_comm = comm;
    return BSX_OK;
}
int bsx_rccl_comm_destroy(void* comm) {
    if (!comm) return BSX_OK;
    if (!rccl().ok) return fail(BSX_ERR_UNSUPPORTED, "RCCL unavailable: %s", rccl().why.c_str());
    const int rc = rccl().comm_destroy(comm);
    if (rc != 0) return fail(BSX_ERR_HIP, "ncclCommDestroy: %s", rccl_err(rc));
    return BSX_OK;
}

int bsx_pipeline_set_rccl(bsx_pipeline* p, void* nccl_comm) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    if (!nccl_comm) { p->rccl_comm = nullptr; p->allgather = nullptr; p->allgather_user = nullptr; return BSX_OK; }
    if (!rccl().ok) return fail(BSX_ERR_UNSUPPORTED, "RCCL unavailable: %s", rccl().why.c_str());
    p->rccl_comm = nccl_comm;
    p->allgather = rccl_allgather_cb;
    p->allgather_user = p;
    return BSX_OK;
}

// The configured all-gather, once, on a scratch payload: rank g sends 128 bytes of (g, i) and must find every rank's block in
// place.  Collective (every rank calls it); world 1 exercises the same call (a one-rank all-gather).  BSX_ERR_HIP on any mismatch.
int bsx_pipeline_check_allgather(bsx_pipeline* p) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    RET(use(p->ctx));
    if (!p->allgather) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_check_allgather: no all-gather set (bsx_pipeline_set_allgather / _set_rccl)");
    RET(join_impl(p));
    const uint32_t world = p->world;
    uint8_t* d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&d), 128 * ((size_t)world + 1)));
    struct Free { void* q; hipStream_t s; bool own; ~Free() { (void)hipFree(q); if (own) (void)hipStreamDestroy(s); } } guard{d, nullptr, false};
    hipStream_t st = p->chunks[0].xchg;
    if (!st) { HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); guard.s = st; guard.own = true; }
    uint8_t mine[128];
    for (int i = 0; i < 128; i++) mine[i] = (uint8_t)(p->rank * 37 + i);
    HIPCHK(hipMemcpyAsync(d, mine, 128, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d + 128, 0xee, 128 * (size_t)world, st));
    if (p->allgather(p->allgather_user, d, d + 128, 128, st) != 0) return fail(BSX_ERR_HIP, "bsx_pipeline_check_allgather: the all-gather failed: %s", bsxapi::g_err.c_str());
    std::vector<uint8_t> got(128 * (size_t)world);
    HIPCHK(hipMemcpyAsync(got.data(), d + 128, got.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t g = 0; g < world; g++)
        for (int i = 0; i < 128; i++)
            if (got[128 * (size_t)g + i] != (uint8_t)(g * 37 + i))
                return fail(BSX_ERR_HIP, "bsx_pipeline_check_allgather: rank %u's block is wrong at byte %d (world %u)", g, i, world);
    return BSX_OK;
}

int bsx_pipeline_create(bsx_ctx* ctx, const bsx_pipeline_config* cfg, bsx_pipeline** out) {
    RET(use(ctx));
    if (!cfg || !out) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_create: null pointer");
    *out = nullptr;
    const uint32_t J = cfg->nb_map_jobs, B = cfg->batch_size, V = cfg->v_max, world = cfg->world ? cfg->world : 1;
    if (!pow2(J) || J > 256) return fail(BSX_ERR_BAD_ARG, "NB_MAP_JOBS must be a power of two <= 256");
    if (!pow2(B) || B > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    if (cfg->flags & ~(BSX_PIPE_WITNESS | BSX_PIPE_COMMIT | BSX_PIPE_CAPS | BSX_PIPE_ED_GENERIC | BSX_PIPE_COMMIT_BESIDE_HASH | BSX_PIPE_RECOMPUTE_PATHS | BSX_PIPE_NO_UNITS))
        return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_create: unknown flags 0x%x", cfg->flags);
    if ((cfg->flags & BSX_PIPE_COMMIT) && (V == 0 || (int)V > bsxk_tally_vmax())) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", V, bsxk_tally_vmax());
    if (!cfg->n_ranges || !cfg->n_chunks || cfg->n_ranges % cfg->n_chunks) return fail(BSX_ERR_BAD_ARG, "n_chunks must divide n_ranges (both > 0)");
    if (cfg->rank >= world || J % world) return fail(BSX_ERR_BAD_ARG, "world %u must divide NB_MAP_JOBS %u and rank %u be below it", world, J, cfg->rank);
    if (!pow2(J / world)) return fail(BSX_ERR_BAD_ARG, "each rank needs a power-of-two slice of the map jobs (the local fold is a subtree of the reference's reduce tree)");
    if (cfg->chain_id_len > 50) return fail(BSX_ERR_BAD_ARG, "chain_id: at most 50 bytes");
    if (cfg->tune_subchain > 2) return fail(BSX_ERR_BAD_ARG, "tune_subchain must be 0, 1 or 2");
    if (cfg->n_sets > 8) return fail(BSX_ERR_BAD_ARG, "n_sets must be in 0..8");
    bsx_pipeline* p = new bsx_pipeline();
    p->ctx = ctx;
    p->cfg = *cfg;
    p->J = J; p->B = B; p->V = V ? V : 1; p->R = cfg->n_ranges; p->E = cfg->n_chunks; p->Rc = p->R / p->E;
    p->K = cfg->n_sets ? cfg->n_sets : 1;
    p->rank = cfg->rank; p->world = world;
    p->jc = J / world; p->jf = p->rank * p->jc;
    p->hpr = (uint64_t)p->jc * B + 1;          // headers this rank holds per range: its slice + the next one
    p->hfr = (uint64_t)p->jf * B;              // height offset of the first of them
    p->with_witness = cfg->flags & BSX_PIPE_WITNESS;
    p->with_commit = cfg->flags & BSX_PIPE_COMMIT;
    p->with_caps = cfg->flags & BSX_PIPE_CAPS;
    p->commit_beside_hash = cfg->flags & BSX_PIPE_COMMIT_BESIDE_HASH;
    p->fused_hint = !(cfg->flags & BSX_PIPE_RECOMPUTE_PATHS);
    // fixed-key tables at any chunk size: they are built where validator sets change (bsx_pipeline_upload), not per step, and the
    // generic kernel is a 1.4 ms chain of 256 doublings that a chunk of a few ranges has nothing to hide behind (rounds 2-3 kept it
    // below 8 ranges per chunk, when a step still paid for the tables: a 1-range step took 1.6 ms, a 16-range step 0.47)
    p->keyed = !(cfg->flags & BSX_PIPE_ED_GENERIC);
    p->ml = bsx_map_layout(B);
    p->rl = bsx_reduce_layout();
    p->cl = bsx_commit_layout(p->V);
    p->sl = bsx_skip_layout(p->V);
    p->units = p->with_commit && (p->with_witness || p->with_caps) && !(cfg->flags & BSX_PIPE_NO_UNITS);
    // Launch forms (measured): beside an expansion the header hashing is held to 2 workgroups per CU and
    // prove_subchain keeps its stages in separate launches, so that the expansion's waves keep half of the register file;
    // alone, both take the whole GPU.
    p->subchain_flags = p->fused_hint ? BSX_SUBCHAIN_PATHS_FROM_HINT : 0;
    const bool beside_expansion = p->with_witness && p->E > 1;
    if (beside_expansion && p->fused_hint) p->subchain_flags |= BSX_SUBCHAIN_SEPARATE_LAUNCHES;
    if (cfg->tune_subchain == 1) p->subchain_flags &= ~BSX_SUBCHAIN_SEPARATE_LAUNCHES;
    if (cfg->tune_subchain == 2 && p->fused_hint) p->subchain_flags |= BSX_SUBCHAIN_SEPARATE_LAUNCHES;
    if (cfg->tune_merkle_workgroups) {
        p->merkle_wgs = cfg->tune_merkle_workgroups == 0xffffffffu ? 0 : cfg->tune_merkle_workgroups;
    } else if (beside_expansion) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) p->merkle_wgs = 2u * (uint32_t)prop.multiProcessorCount;
    }
    if (p->with_caps) {
        p->leaf_len = cfg->leaf_len ? cfg->leaf_len : 135;
        p->n_leaves = bsx_witness_leaf_count(p->ml.n_elements, p->leaf_len);
        uint32_t lg = 0;
        while ((1u << (lg + 1)) <= p->n_leaves) lg++;
        p->cap_height = cfg->cap_height || cfg->leaf_len ? cfg->cap_height : 4;
        if (p->cap_height > lg) p->cap_height = lg;
        p->tree_digests = bsx_poseidon_tree_digests(p->n_leaves, p->cap_height);
        if (!p->n_leaves || !p->tree_digests) { delete p; return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_create: bad leaf_len / cap_height"); }
        // the COMMIT / SKIP units' trees: same row length, cap height clipped to each tree's own depth
        auto tree_of = [&](uint64_t n_elements, uint32_t& n_leaves, uint32_t& cap_h, uint64_t& digests) {
            n_leaves = bsx_witness_leaf_count(n_elements, p->leaf_len);
            uint32_t l2 = 0;
            while ((1u << (l2 + 1)) <= n_leaves) l2++;
            cap_h = p->cap_height > l2 ? l2 : p->cap_height;
            digests = bsx_poseidon_tree_digests(n_leaves, cap_h);
        };
        tree_of(p->cl.n_elements, p->n_leaves_cm, p->cap_h_cm, p->tree_digests_cm);
        tree_of(p->sl.n_elements, p->n_leaves_sk, p->cap_h_sk, p->tree_digests_sk);
    }
    p->compact_tokens = !p->with_witness && (p->E > 1 || p->K > 1);
    p->chunks.resize((size_t)p->E * p->K);
    {
        const size_t hot = 2 * p->chunks.size();
        const size_t n_pool = hot > PIPE_POOL ? hot : PIPE_POOL;
        void* touch = nullptr;
        if (hipMalloc(&touch, 256 + 8 * (size_t)world) != hipSuccess) { bsx_pipeline_destroy(p); return fail(BSX_ERR_HIP, "bsx_pipeline_create: out of device memory"); }
        p->allocs.push_back(touch);
        p->touch = static_cast<uint8_t*>(touch);
        for (size_t i = 0; i < n_pool; i++) {
            hipStream_t st = nullptr;
            const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e != hipSuccess) { bsx_pipeline_destroy(p); return fail(BSX_ERR_HIP, "bsx_pipeline_create: hipStreamCreate: %s", hipGetErrorString(e)); }
            p->pool.push_back(st);
            (void)hipMemsetAsync(touch, 0, 256, st);                   // first command: binds the stream's hardware queue, in creation order
            (void)hipStreamSynchronize(st);
        }
        p->assign.resize(hot);
        for (size_t i = 0; i < hot; i++) p->assign[i] = (uint32_t)i;   // default: consecutive queues
    }
    for (Chunk& c : p->chunks) {
        const int rc = create_chunk(p, c);
        if (rc != BSX_OK) {
            const std::string msg = bsxapi::g_err;
            bsx_pipeline_destroy(p);
            bsxapi::g_err = msg;
            return rc;
        }
    }
    // the allocations were zeroed with null-stream memsets, which the pipeline's non-blocking streams do not wait for: drain
    // them here, or an upload enqueued right away could be overtaken and wiped by a memset still pending
    {
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            bsx_pipeline_destroy(p);
            return fail(BSX_ERR_HIP, "bsx_pipeline_create: %s", hipGetErrorString(e));
        }
    }
    *out = p;
    return BSX_OK;
}

void bsx_pipeline_destroy(bsx_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    for (Chunk& c : p->chunks) {
        for (hipStream_t s : {c.main, c.side})
            if (s) (void)hipStreamSynchronize(s);
        for (hipStream_t s : {c.xchg, c.copy})
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        for (hipEvent_t e : {c.ev_sync, c.ev_merkle, c.ev_fill, c.ev_fin, c.ev_inputs_consumed, c.ev_h2d, c.ev_commit_done[0], c.ev_commit_done[1],
                             c.ev_hash_tok, c.ev_expand_tok, c.ev_x_in, c.ev_x_out, c.ev_finalized, c.ev_units_done})
            if (e) (void)hipEventDestroy(e);
        for (TimingSlot& t : c.timing)
            for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
        if (c.witness_map && c.witness_map_vmm) (void)bsx_dev_free(p->ctx, c.witness_map);
    }
    for (hipStream_t st : p->pool) (void)hipStreamDestroy(st);
    for (void* q : p->allocs) (void)hipFree(q);
    for (void* q : p->host_allocs) (void)hipHostFree(q);
    delete p;
}

int bsx_pipeline_set_allgather(bsx_pipeline* p, bsx_allgather_fn fn, void* user) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    p->allgather = fn;
    p->allgather_user = user;
    return BSX_OK;
}

int bsx_pipeline_upload(bsx_pipeline* p, const bsx_pipeline_inputs* in) {
    if (!p || !in) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_upload: null pointer");
    RET(use(p->ctx));
    if (!in->headers || !in->ranges || !in->latest) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_upload: headers / ranges / latest are required");
    if (p->with_commit && (!in->target_validators || !in->trusted_validators)) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_upload: validator sets are required with BSX_PIPE_COMMIT");
    const uint64_t HPR = in->headers_per_range;
    if (HPR < p->hfr + p->hpr) return fail(BSX_ERR_BAD_ARG, "headers_per_range %llu does not cover this rank's slice (needs %llu)", (unsigned long long)HPR,
                                           (unsigned long long)(p->hfr + p->hpr));
    RET(join_impl(p));                         // steps in flight read the buffers about to be overwritten
    const uint32_t R = p->R, Rc = p->Rc, V = p->V;
    // every owned range is validated BEFORE the first copy is enqueued: an error return leaves the device buffers and the caller's
    // memory untouched (nothing in flight), and the previous upload — if any — stays valid
    if (p->with_commit)
        for (uint32_t k = 0; k < R; k++) {
            const bsx_shared_ctx& rg = in->ranges[(size_t)p->rank * R + k];
            const uint64_t ti = rg.end_block - rg.start_block;
            if (rg.end_block <= rg.start_block || ti >= HPR)
                return fail(BSX_ERR_BAD_ARG, "range %zu: target header (end - start = %llu) is not among the %llu supplied headers", (size_t)p->rank * R + k,
                            (unsigned long long)ti, (unsigned long long)HPR);
        }
    p->uploaded = false;                       // until every copy below has landed
    struct DrainOnError {                      // a failing HIP call mid-way: nothing may still read the caller's buffers on return
        bsx_pipeline* p; bool armed = true;
        ~DrainOnError() { if (armed) for (Chunk& c : p->chunks) (void)hipStreamSynchronize(c.main); }
    } drain{p};
    std::vector<bsx_header> sk((size_t)Rc * 2);
    for (uint32_t ce = 0; ce < p->E * p->K; ce++) {
        const uint32_t e = ce % p->E;                                  // every buffer set holds the same inputs
        Chunk& c = p->chunks[ce];
        hipStream_t st = c.main;
        for (uint32_t g = 0; g < p->world; g++) {
            const size_t r0 = (size_t)g * R + (size_t)e * Rc;          // first global range of this (chunk, owner) block
            const size_t i0 = (size_t)g * Rc;                          // its position inside the chunk
            HIPCHK(hipMemcpy2DAsync(c.headers_all + i0 * p->hpr * sizeof(bsx_header), p->hpr * sizeof(bsx_header),
                                    in->headers + r0 * HPR + p->hfr, HPR * sizeof(bsx_header), p->hpr * sizeof(bsx_header), Rc,
                                    hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.ranges + i0 * sizeof(bsx_shared_ctx), in->ranges + r0, (size_t)Rc * sizeof(bsx_shared_ctx), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.latest + i0 * 8, in->latest + r0, (size_t)Rc * 8, hipMemcpyHostToDevice, st));
        }
        if (p->with_commit) {
            const size_t r0 = (size_t)p->rank * R + (size_t)e * Rc;    // owned block
            for (uint32_t k = 0; k < Rc; k++) {
                const bsx_shared_ctx& rg = in->ranges[r0 + k];
                const uint64_t ti = rg.end_block - rg.start_block;     // validated above
                sk[2 * k] = in->headers[(r0 + k) * HPR];               // trusted header: height S
                sk[2 * k + 1] = in->headers[(r0 + k) * HPR + ti];      // target header: height E
            }
            HIPCHK(hipMemcpyAsync(c.headers_all + c.nh_main * sizeof(bsx_header), sk.data(), (size_t)Rc * 2 * sizeof(bsx_header), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.skip_ranges, in->ranges + r0, (size_t)Rc * sizeof(bsx_shared_ctx), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.skip_ranges_side, in->ranges + r0, (size_t)Rc * sizeof(bsx_shared_ctx), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.validators, in->target_validators + r0 * V, (size_t)Rc * V * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c.trusted, in->trusted_validators + r0 * V, (size_t)Rc * V * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
            if (p->keyed) {
                // Fixed-key Ed25519 tables: validator sets change HERE and nowhere else, so this is where the rows are compared with
                // the new keys and rebuilt where they differ (k_keytable_check + k_table_entries; rows persist) — not once per step.
                // Rows are keyed by PUBLIC KEY (keycache.h): the ranges of a chunk may be signed by different validator sets (sets
                // change along the chain: header_range.rs:42-48, fetcher.rs:60-87) and still share the table; it is sized for the
                // chunk's distinct keys.  Only keys beyond PIPE_KEY_ROWS_MAX go to the generic kernel (counted: the step sizes that pass).
                const bsx_validator* sets = in->target_validators + r0 * V;
                uint32_t want = V;
                {
                    bsx_keycache probe;                       // distinct active keys of the chunk (a dry run of the map)
                    probe.init(V, PIPE_KEY_ROWS_MAX);
                    std::vector<uint32_t> tmp((size_t)Rc * V), d;
                    (void)probe.assign(sets, Rc, tmp.data(), d, nullptr);
                    uint32_t hi = V;
                    for (uint32_t q : d) if (q + 1 > hi) hi = q + 1;
                    want = hi;
                }
                if (want > c.keytable_rows) {                // a larger table: the old rows are gone with it
                    if (c.keytable) {
                        for (auto it = p->allocs.begin(); it != p->allocs.end(); ++it)
                            if (*it == c.keytable) { p->allocs.erase(it); break; }
                        (void)hipFree(c.keytable);
                        c.keytable = nullptr;
                    }
                    if (c.rowkeys) {
                        for (auto it = p->allocs.begin(); it != p->allocs.end(); ++it)
                            if (*it == c.rowkeys) { p->allocs.erase(it); break; }
                        (void)hipFree(c.rowkeys);
                        c.rowkeys = nullptr;
                    }
                    const uint32_t rows = want + want / 8;    // head room: the next upload's few new keys need no new table
                    RET(dalloc_t(p, bsxk_keytable_bytes(rows), &c.keytable));
                    RET(dalloc_t(p, (size_t)rows * sizeof(bsx_validator), &c.rowkeys));
                    c.keytable_rows = rows;
                    c.kc.init(V, rows);
                    // dalloc zeroes with a null-stream memset, which this chunk's non-blocking stream does not wait for: drained here, or
                    // the build below could be overtaken and wiped by it (the same trap as in bsx_pipeline_create)
                    HIPCHK(hipDeviceSynchronize());
                }
                std::vector<uint32_t> rows_h((size_t)Rc * V), dirty;
                uint64_t deferred = 0;
                c.rows_identity = c.kc.assign(sets, Rc, rows_h.data(), dirty, &deferred);
                c.n_key_mismatch = deferred;
                if (!c.rows_identity) HIPCHK(hipMemcpyAsync(c.rows, rows_h.data(), rows_h.size() * 4, hipMemcpyHostToDevice, st));
                if (!dirty.empty()) {
                    std::vector<bsx_validator> rk(c.keytable_rows);
                    memset(rk.data(), 0, rk.size() * sizeof(bsx_validator));
                    for (uint32_t q = 0; q < c.keytable_rows; q++) memcpy(rk[q].pubkey, &c.kc.keys[(size_t)q * 32], 32);
                    HIPCHK(hipMemcpyAsync(c.rowkeys, rk.data(), rk.size() * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
                    HIPCHK(bsxk_ed25519_keytable(st, reinterpret_cast<const bsx_validator*>(c.rowkeys), c.keytable_rows, c.keytable));
                    HIPCHK(hipStreamSynchronize(st));        // rows_h / rk are this scope's
                } else {
                    HIPCHK(hipStreamSynchronize(st));
                }
            }
        }
        HIPCHK(hipStreamSynchronize(st));      // `sk` and the caller's buffers are free again
        if (c.host_image) {                    // keep the streamed image in step with the resident block
            HIPCHK(hipMemcpyAsync(c.host_image, c.headers_all, c.headers_bytes, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    drain.armed = false;
    p->uploaded = true;
    return BSX_OK;
}

int bsx_pipeline_enable_input_streaming(bsx_pipeline* p, int on) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    RET(use(p->ctx));
    RET(join_impl(p));
    if (!on) {
        p->streaming = false;
        for (Chunk& c : p->chunks) c.h2d_pending = false;
        return BSX_OK;
    }
    for (Chunk& c : p->chunks) {
        if (!c.copy) RET(new_stream(&c.copy));
        if (!c.host_image) {
            void* q = nullptr;
            HIPCHK(hipHostMalloc(&q, c.headers_bytes, hipHostMallocDefault));
            p->host_allocs.push_back(q);
            c.host_image = static_cast<uint8_t*>(q);
        }
        HIPCHK(hipMemcpy(c.host_image, c.headers_all, c.headers_bytes, hipMemcpyDeviceToHost));
    }
    p->streaming = true;
    return BSX_OK;
}

// One step over all chunks.  Two tokens keep the chunks in complementary phases: only one chunk hashes at a time and only
// one expands at a time, so chunk e+1's (ALU-bound) hashing always runs beside chunk e's (HBM-bound) expansion — without the
// tokens the streams drift into the same phase and the overlap is lost.
int bsx_pipeline_step(bsx_pipeline* p) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    RET(use(p->ctx));
    if (!p->uploaded) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_step before bsx_pipeline_upload");
    const bool multi = p->E > 1;
    const uint32_t set = (uint32_t)(p->step_index % p->K);
    p->step_index++;
    p->last_set = set;
    for (uint32_t e = 0; e < p->E; e++) {
        Chunk& c = p->chunks[(size_t)set * p->E + e];
        TimingSlot* ts = p->timing_on ? timing_slot(c) : nullptr;
        if (multi && !p->compact_tokens && p->hash_token) HIPCHK(hipStreamWaitEvent(c.main, p->hash_token, 0));
        RET(step_local(p, c, ts));
        RET(stream_inputs(p, c));
        if (p->pending_verify) {
            // the previous chunk's signature checks: enqueued now so that they can wait for THIS chunk's k_header_merkle
            // (both are integer-ALU bound; the rest of this chunk's hashing phase leans on memory)
            Chunk* pv = p->pending_verify;
            p->pending_verify = nullptr;
            RET(launch_verify(p, *pv, c.ev_merkle));
        }
        const uint8_t* res = c.partial;       // single GPU: the local fold already is the range result
        if (p->world > 1) {
            RET(exchange_begin(p, c, ts));    // collective in flight; finished behind this chunk's expansion
            res = nullptr;
        }
        if (multi) {
            HIPCHK(hipEventRecord(c.ev_hash_tok, c.main));
            p->hash_token = c.ev_hash_tok;
        }
        const bool defer = multi;             // the commit check is launched with the NEXT chunk's header hashing as its gate
        RET(step_final(p, c, res, !defer, multi ? p->expand_token : nullptr, ts));
        if (defer && commit_active(p, c) && !p->commit_beside_hash) p->pending_verify = &c;
        // the commit check is NOT joined here: its inputs are double-buffered by step parity, so it may run on into the
        // chunk's next step; bsx_pipeline_join / _get_results wait for it
    }
    return BSX_OK;
}

// Which hardware queues the chunks' streams sit on decides how well their phases overlap: the same pipeline measured 77 .. 98 M
// headers/s (header_range_2048, two chunks, one box) by nothing but the position of its four hot streams among the process's
// queues — the command processor serves queues that share a dispatch pipe one kernel at a time, and a kernel with more
// workgroups than the GPU holds (every large launch here) keeps its pipe busy for its whole duration.  The mapping of HSA
// queues to pipes is the driver's, so it is measured: every candidate assignment of pool streams (consecutive quartets at each
// offset, mains-then-sides at each offset) runs `steps_per_trial` real steps; the fastest stays.
int bsx_pipeline_autotune(bsx_pipeline* p, uint32_t steps_per_trial, bsx_pipeline_autotune_result* out) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    RET(use(p->ctx));
    if (!p->uploaded) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_autotune before bsx_pipeline_upload");
    if (p->world > 1 && !p->allgather) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline: world > 1 needs bsx_pipeline_set_allgather before bsx_pipeline_autotune");
    const bool auto_steps = steps_per_trial == 0;
    const size_t nc = p->chunks.size(), hot = 2 * nc, P = p->pool.size();
    std::vector<std::vector<uint32_t>> cand;
    cand.push_back(p->assign);                                        // trial 0: what the pipeline was created with
    for (size_t base = 0; base + hot <= P; base++)
        for (int pattern = 0; pattern < 2; pattern++) {
            std::vector<uint32_t> a(hot);
            for (size_t i = 0; i < nc; i++) {
                a[2 * i] = (uint32_t)(base + (pattern ? i : 2 * i));
                a[2 * i + 1] = (uint32_t)(base + (pattern ? nc + i : 2 * i + 1));
            }
            if (std::find(cand.begin(), cand.end(), a) == cand.end()) cand.push_back(a);
        }
    auto apply = [&](const std::vector<uint32_t>& a) {
        p->assign = a;
        for (size_t i = 0; i < nc; i++) { p->chunks[i].main = p->pool[a[2 * i]]; p->chunks[i].side = p->pool[a[2 * i + 1]]; }
    };
    const bool timing_was = p->timing_on;
    p->timing_on = false;
    int rc = BSX_OK;
    // `n` steps on assignment `a`, joined on both sides; ms per step (< 0: a step failed, rc holds the status)
    auto run = [&](const std::vector<uint32_t>& a, uint32_t n) -> double {
        if ((rc = join_impl(p)) != BSX_OK) return -1;                 // streams are swapped only while nothing is in flight
        apply(a);
        if ((rc = bsx_pipeline_step(p)) != BSX_OK || (rc = join_impl(p)) != BSX_OK) return -1;   // untimed: first launches on these queues
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t k = 0; k < n; k++)
            if ((rc = bsx_pipeline_step(p)) != BSX_OK) return -1;
        if ((rc = join_impl(p)) != BSX_OK) return -1;
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
    };
    std::vector<double> ms(cand.size(), 0.0);
    double worst_ms = 0;
    if (auto_steps) {                                                 // ~20 ms of steps per trial, 3 .. 32
        const double one = run(cand[0], 3);                           // also the warm-up of the whole exercise
        if (one < 0) { p->timing_on = timing_was; return rc; }
        const double want = 20.0 / (one > 0.05 ? one : 0.05);
        steps_per_trial = want < 3 ? 3 : want > 32 ? 32 : (uint32_t)want;
        if (p->world > 1) {
            // every step holds a collective: all ranks must run the SAME number of steps.  Agree on the smallest count through
            // the caller's own all-gather (8 bytes per rank)
            uint64_t mine = steps_per_trial;
            std::vector<uint64_t> all(p->world);
            hipStream_t xs = p->chunks[0].xchg;
            hipError_t e = hipMemcpyAsync(p->touch, &mine, 8, hipMemcpyHostToDevice, xs);
            if (e == hipSuccess) e = hipStreamSynchronize(xs);
            if (e != hipSuccess) { p->timing_on = timing_was; return fail(BSX_ERR_HIP, "bsx_pipeline_autotune: %s", hipGetErrorString(e)); }
            if (p->allgather(p->allgather_user, p->touch, p->touch + 256, 8, xs) != 0) {
                p->timing_on = timing_was;
                return fail(BSX_ERR_HIP, "bsx_pipeline_autotune: the all-gather callback failed");
            }
            e = hipStreamSynchronize(xs);
            if (e == hipSuccess) e = hipMemcpy(all.data(), p->touch + 256, 8 * (size_t)p->world, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { p->timing_on = timing_was; return fail(BSX_ERR_HIP, "bsx_pipeline_autotune: %s", hipGetErrorString(e)); }
            for (uint64_t v : all) if (v >= 1 && v < steps_per_trial) steps_per_trial = (uint32_t)v;
        }
    }
    for (size_t t = 0; t < cand.size(); t++) {
        ms[t] = run(cand[t], steps_per_trial);
        if (ms[t] < 0) { p->timing_on = timing_was; return rc; }
        if (ms[t] > worst_ms) worst_ms = ms[t];
    }
    // the three fastest once more, twice as long: a trial is short, and the winner's margin is often within its noise
    std::vector<size_t> order(cand.size());
    for (size_t t = 0; t < order.size(); t++) order[t] = t;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ms[x] < ms[y]; });
    const double first_ms = ms[0];
    size_t best = order[0];
    double best_ms = 1e30;
    for (size_t k = 0; k < 3 && k < order.size(); k++) {
        const double m = run(cand[order[k]], 2 * steps_per_trial);
        if (m < 0) { p->timing_on = timing_was; return rc; }
        if (m < best_ms) { best_ms = m; best = order[k]; }
    }
    apply(cand[best]);
    p->timing_on = timing_was;
    for (Chunk& c : p->chunks) c.timing_used = 0;
    if (out) {
        memset(out, 0, sizeof *out);
        const int q = bsxk_queue_groups(p->pool.data(), (uint32_t)p->pool.size(), nullptr);
        out->hw_queues = q > 0 ? (uint32_t)q : 0;
        out->n_trials = (uint32_t)cand.size();
        out->best_trial = (uint32_t)best;
        out->steps_per_trial = steps_per_trial;
        out->initial_ms = first_ms; out->best_ms = best_ms; out->worst_ms = worst_ms;
        for (size_t i = 0; i < hot && i < 16; i++) out->assignment[i] = cand[best][i];
    }
    return BSX_OK;
}

int bsx_pipeline_join(bsx_pipeline* p) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    RET(use(p->ctx));
    return join_impl(p);
}

int bsx_pipeline_get_results(bsx_pipeline* p, bsx_pipeline_results* out) {
    if (!p || !out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    RET(use(p->ctx));
    RET(join_impl(p));
    out->header_status = out->assemble_status = 0;
    const uint32_t R = p->R, Rc = p->Rc;
    for (uint32_t e = 0; e < p->E; e++) {
        Chunk& c = p->chunks[(size_t)p->last_set * p->E + e];
        uint32_t st[2] = {0, 0};
        HIPCHK(hipMemcpy(st, c.status, 8, hipMemcpyDeviceToHost));
        out->header_status |= st[0];
        out->assemble_status |= st[1];
        const size_t k0 = (size_t)e * Rc;
        if (out->output64) HIPCHK(hipMemcpy(out->output64 + k0 * 64, c.output64, (size_t)Rc * 64, hipMemcpyDeviceToHost));
        if (out->range_status) HIPCHK(hipMemcpy(out->range_status + k0, c.range_status, (size_t)Rc * 4, hipMemcpyDeviceToHost));
        if (out->skip_status) {
            if (p->with_commit) HIPCHK(hipMemcpy(out->skip_status + k0, c.skip_status, (size_t)Rc * 4, hipMemcpyDeviceToHost));
            else memset(out->skip_status + k0, 0, (size_t)Rc * 4);
        }
        if (out->commit) {
            if (p->with_commit) HIPCHK(hipMemcpy(out->commit + k0, c.commit_res, (size_t)Rc * sizeof(bsx_commit_result), hipMemcpyDeviceToHost));
            else memset(out->commit + k0, 0, (size_t)Rc * sizeof(bsx_commit_result));
        }
        if (out->records)
            for (uint32_t g = 0; g < p->world; g++)
                HIPCHK(hipMemcpy(out->records + ((size_t)g * R + k0) * p->jc, c.records + (size_t)g * Rc * p->jc * sizeof(bsx_subchain),
                                 (size_t)Rc * p->jc * sizeof(bsx_subchain), hipMemcpyDeviceToHost));
    }
    return BSX_OK;
}

int bsx_pipeline_buffer(bsx_pipeline* p, uint32_t chunk, uint32_t which, void** out_d_ptr, uint64_t* out_bytes) {
    if (!p || !out_d_ptr || !out_bytes) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (chunk >= p->E * p->K) return fail(BSX_ERR_BAD_ARG, "chunk %u out of range (n_sets x n_chunks = %u)", chunk, p->E * p->K);
    Chunk& c = p->chunks[chunk];
    void* q = nullptr;
    uint64_t n = 0;
    switch (which) {
    case BSX_PIPE_BUF_WITNESS_MAP: q = c.witness_map; n = c.witness_map ? c.n_map_el * 8 : 0; break;
    case BSX_PIPE_BUF_WITNESS_REDUCE_LOCAL: q = c.witness_red_local; n = c.witness_red_local ? c.n_red_local_el * 8 : 0; break;
    case BSX_PIPE_BUF_WITNESS_REDUCE_TOP: q = c.witness_red_top; n = c.witness_red_top ? c.n_red_top_el * 8 : 0; break;
    case BSX_PIPE_BUF_COMPACT: q = c.compact; n = c.compact_bytes; break;
    case BSX_PIPE_BUF_TREES: q = c.trees; n = c.trees_words * 8; break;
    case BSX_PIPE_BUF_PARTIAL: q = c.partial; n = (uint64_t)c.RT * 128; break;
    case BSX_PIPE_BUF_HEADERS: q = c.headers_all; n = c.headers_bytes; break;
    case BSX_PIPE_BUF_RECORDS: q = c.records; n = c.records_bytes; break;
    case BSX_PIPE_BUF_GATHERED: q = c.gathered; n = (uint64_t)p->world * c.RT * 128; break;
    case BSX_PIPE_BUF_REDUCE_COMPACT_LOCAL: q = c.red_compact_local; n = (uint64_t)c.RT * (p->jc > 1 ? p->jc - 1 : 0) * p->rl.compact_stride; break;
    case BSX_PIPE_BUF_HASHES: q = c.hashes_all; n = c.nh_all * 32; break;
    case BSX_PIPE_BUF_DH_AUNTS: q = c.dh_aunts; n = c.nh_all * 128; break;
    case BSX_PIPE_BUF_LB_AUNTS: q = c.lb_aunts; n = c.nh_all * 128; break;
    case BSX_PIPE_BUF_RANGES: q = p->with_commit && p->world == 1 ? c.skip_ranges : c.ranges; n = (uint64_t)(p->with_commit && p->world == 1 ? c.R : c.RT) * sizeof(bsx_shared_ctx); break;
    case BSX_PIPE_BUF_PATHS: q = c.paths; n = c.paths ? c.nh_all * BSX_HEADER_PATH_BYTES : 0; break;
    case BSX_PIPE_BUF_WITNESS_COMMIT: q = c.witness_commit; n = c.witness_commit ? (uint64_t)c.R * p->cl.n_elements * 8 : 0; break;
    case BSX_PIPE_BUF_WITNESS_SKIP: q = c.witness_skip; n = c.witness_skip ? (uint64_t)c.R * p->sl.n_elements * 8 : 0; break;
    case BSX_PIPE_BUF_COMPACT_COMMIT: q = c.commit_compact; n = c.commit_compact ? (uint64_t)c.R * p->cl.compact_stride : 0; break;
    case BSX_PIPE_BUF_COMPACT_SKIP: q = c.skip_compact; n = c.skip_compact ? (uint64_t)c.R * p->sl.compact_stride : 0; break;
    case BSX_PIPE_BUF_TREES_COMMIT: q = c.trees_commit; n = c.trees_commit ? (uint64_t)c.R * p->tree_digests_cm * 32 : 0; break;
    case BSX_PIPE_BUF_TREES_SKIP: q = c.trees_skip; n = c.trees_skip ? (uint64_t)c.R * p->tree_digests_sk * 32 : 0; break;
    default: return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_buffer: unknown buffer %u", which);
    }
    *out_d_ptr = n ? q : nullptr;
    *out_bytes = n;
    return BSX_OK;
}

int bsx_pipeline_set_timing(bsx_pipeline* p, int on) {
    if (!p) return fail(BSX_ERR_BAD_ARG, "null pipeline");
    p->timing_on = on != 0;
    return BSX_OK;
}

int bsx_pipeline_timing2(bsx_pipeline* p, bsx_pipeline_timing_result2* out, uint32_t out_bytes) {
    if (!p || !out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (out_bytes < sizeof(bsx_pipeline_timing_result)) return fail(BSX_ERR_BAD_ARG, "bsx_pipeline_timing2: out_bytes %u < 32", out_bytes);
    RET(use(p->ctx));
    RET(join_impl(p));
    double sub = 0, ex = 0, caps = 0;
    uint32_t n_sub = 0, n_ex = 0, n_caps = 0;
    std::vector<float> xs;
    for (Chunk& c : p->chunks) {
        for (size_t i = 0; i < c.timing_used; i++) {
            TimingSlot& t = c.timing[i];
            float ms = 0;
            if (t.sub) { HIPCHK(hipEventElapsedTime(&ms, t.ev[0], t.ev[1])); sub += ms; n_sub++; }
            if (t.exp) { HIPCHK(hipEventElapsedTime(&ms, t.ev[2], t.ev[3])); ex += ms; n_ex++; }
            if (t.caps) { HIPCHK(hipEventElapsedTime(&ms, t.ev[4], t.ev[5])); caps += ms; n_caps++; }
            if (t.xchg) { HIPCHK(hipEventElapsedTime(&ms, t.ev[6], t.ev[7])); xs.push_back(ms); }
        }
        c.timing_used = 0;
    }
    bsx_pipeline_timing_result2 r{};
    r.prove_subchain_ms = n_sub ? sub / n_sub : 0;
    r.expand_map_ms = n_ex ? ex / n_ex : 0;
    r.caps_ms = n_caps ? caps / n_caps : 0;
    r.launches = n_sub;
    r.exchanges = (uint32_t)xs.size();
    if (!xs.empty()) {
        std::sort(xs.begin(), xs.end());
        double t = 0;
        for (float v : xs) t += v;
        r.allgather_ms_avg = t / xs.size();
        r.allgather_ms_min = xs.front();
        r.allgather_ms_median = xs[xs.size() / 2];
        r.allgather_ms_max = xs.back();
    }
    memcpy(out, &r, out_bytes < sizeof r ? out_bytes : sizeof r);      // never more than the caller's struct holds
    return BSX_OK;
}

int bsx_pipeline_timing(bsx_pipeline* p, bsx_pipeline_timing_result* out) {
    bsx_pipeline_timing_result2 r{};
    RET(bsx_pipeline_timing2(p, out ? &r : nullptr, sizeof r));
    out->prove_subchain_ms = r.prove_subchain_ms;
    out->expand_map_ms = r.expand_map_ms;
    out->caps_ms = r.caps_ms;
    out->launches = r.launches;
    out->_pad = 0;
    return BSX_OK;
}

}  // extern "C"
