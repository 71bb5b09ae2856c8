// kernels.h — the kernel launchers of libbsx.so (kernels_*.hip), as seen by the host-side translation units
// (api.hip, api_poseidon.hip, pipeline.hip).  Every launcher only enqueues on the given stream and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/bsx.h"

// Experiment knobs of the launchers (tools/*.sh sweeps).  A release build reads NO environment variable: the value is the default,
// resolved at compile time; `make EXPERIMENTS=1` (-DBSX_EXPERIMENTS) builds the variant the sweep scripts drive.
#ifdef BSX_EXPERIMENTS
static inline long bsx_knob(const char* name, long dflt) {
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
}
#else
#define bsx_knob(name, dflt) (static_cast<long>(dflt))
#endif

// Experiments build only: the launchers note WHICH kernel form they launched, so that the tests that force a form through a knob
// (tests/test_gpu_ed_variants.py, tests/test_gpu_engine.py::test_expansion_kernel_variants_agree) can assert that the forced form is
// the one that ran — bsx_debug_last_launch_form(which) is exported by libbsx_exp.so and absent from the product library.
//   which 0 (fixed-key Ed25519 signature kernel): 0x100 | lanes      k_ed25519_verify_keyed_proj<8 / 16>
//                                                 0x200              k_ed25519_verify_keyed_mixed
//                                                 0x300              k_ed25519_verify_keyed_small
//                                                 0x400 | SPLIT | BYKEY << 4 | DEFER (scratch) << 5   k_ed25519_verify_keyed<DEFER, BYKEY, SPLIT>
//   which 1 (witness expansion):                  staging chunk | non-temporal << 16 | capped grid << 17
#define BSX_FORM_ED 0u
#define BSX_FORM_EXPAND 1u
#ifdef BSX_EXPERIMENTS
extern "C" void bsxk_debug_note_form(uint32_t which, uint32_t form);
extern "C" uint32_t bsx_debug_last_launch_form(uint32_t which);
#define BSX_NOTE_FORM(which, form) bsxk_debug_note_form((which), (form))
#else
#define BSX_NOTE_FORM(which, form) ((void)0)
#endif

// Where a kernel of the commit chain leaves the witness variables it holds (include/bsx_layout.h): unit c at base + c * stride
// (COMMIT units: c = commit; SKIP / STEP units: c = range); base == nullptr: no witness.  mode (k_commit_tally): 0 = the COMMIT
// unit's own validator set, 1 = the trusted set inside a SKIP unit.
struct bsxk_unit_dst {
    uint8_t* base;
    uint32_t stride, off_words, off_bools, mode;
};
inline bsxk_unit_dst bsxk_unit(uint8_t* base, const bsx_witness_layout& L, uint32_t mode = 0) {
    return bsxk_unit_dst{base, L.compact_stride, L.off_words, L.off_bools, mode};
}

extern "C" {
// tap (optional): the root of header `idx` of the launch is ALSO stored at dst_a / dst_b (32 bytes each, either may be null) — the
// host tier's ctx.end_header_hash and dense target hash, which k_fill_end_hash would copy one kernel boundary later
// Per-range form (the coalescing front end: R requests of `hpr` headers each in one launch): ranges != nullptr -> the root of header
// r * hpr + (ranges[r].end_block - ranges[r].start_block) is stored into ranges[r].end_header_hash and dense + 32 r (= k_fill_end_hash)
struct bsxk_merkle_tap { uint64_t idx; uint8_t* dst_a; uint8_t* dst_b; bsx_shared_ctx* ranges; uint64_t hpr; uint8_t* dense; };
// status_group (coalescing front end): != 0 -> one status word per `status_group` consecutive headers (status[me / status_group])
hipError_t bsxk_header_merkle(hipStream_t, const bsx_header*, uint64_t, uint8_t*, uint8_t*, uint8_t*, uint8_t*, uint32_t*, uint32_t, uint32_t,
                              const bsxk_merkle_tap* tap = nullptr, uint64_t status_group = 0);
hipError_t bsxk_zero_paths(hipStream_t, uint8_t*);
hipError_t bsxk_assemble_inputs(hipStream_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*,
                                const uint64_t*, const bsx_header*, uint64_t, uint64_t, const uint8_t*, const uint8_t*, const uint8_t*,
                                uint8_t*, uint32_t*, const uint8_t*, const uint8_t*, uint32_t lds_pad = 0, const uint32_t* spans = nullptr,
                                uint32_t status_per_range = 0, const uint32_t* jobs = nullptr);
// the hint's expected_data_commitment (input.rs:241-244 with :70-72) of n coalesced requests, straight from their assembled compact
// images: root over the data hashes of [start, min(end, latest - 2)), all-zero when that range is empty.  out: n x 32
hipError_t bsxk_expected_commitments(hipStream_t, uint32_t n, uint32_t B, const bsx_shared_ctx* ranges, const uint32_t* spans, const uint64_t* latest,
                                     const uint32_t* jobs /* optional: request r starts at ranges[r].start_block + jobs[r] * B */, const uint8_t* compact,
                                     uint8_t* out);
hipError_t bsxk_prove_subchain(hipStream_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*, uint8_t*, bsx_subchain*, uint32_t);
hipError_t bsxk_reduce(hipStream_t, uint32_t, uint32_t, const bsx_subchain*, uint64_t, uint64_t, bsx_subchain*, uint8_t*);
hipError_t bsxk_reduce_finalize(hipStream_t, uint32_t, uint32_t, const bsx_subchain*, bsx_subchain*, uint8_t*, uint32_t, uint32_t, const bsx_shared_ctx*,
                                const uint8_t*, uint8_t*, uint32_t*);
hipError_t bsxk_finalize(hipStream_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*, const bsx_subchain*, const uint8_t*,
                         uint8_t*, uint32_t*, uint8_t*, uint32_t);
hipError_t bsxk_expand_witness(hipStream_t, const bsx_witness_layout*, uint32_t, const uint8_t*, uint64_t*);
hipError_t bsxk_sha512_challenge(hipStream_t, const bsx_validator*, uint64_t, uint8_t*, uint8_t*, uint32_t, const bsxk_unit_dst*);
hipError_t bsxk_ed25519_verify(hipStream_t, const bsx_validator*, const uint8_t*, uint64_t, uint8_t*);
// digit width of a key table (round 5): BSXK_KT_BITS = the default (request-driven paths), BSXK_KT_BITS_WIDE for a resident validator set
// verified millions of times (64 MB per key, 16 + 16 instead of 22 + 16 additions per signature); part of the rows' layout tag: a
// kernel told the wrong width finds no usable row and defers every slot
#define BSXK_KT_BITS 12
#define BSXK_KT_BITS_WIDE 16
int bsxk_keytable_default_bits(void);
int bsxk_keytable_bits_ok(int);
uint64_t bsxk_keytable_bytes(uint32_t, int w = BSXK_KT_BITS);
hipError_t bsxk_ed25519_keytable(hipStream_t, const bsx_validator*, uint32_t, uint8_t*, int w = BSXK_KT_BITS);
// rows (optional, n u32): signature i is checked against table row rows[i] instead of row i % v_max (0xffffffff = no row: deferred to
// the generic kernel) — validator sets that differ between the commits of a batch share ONE table whose rows are keyed by public key
// (bsx_keycache, api_internal.h).  Whatever the map says, a lane verifies against a row only when the row's key IS its public key.
hipError_t bsxk_ed25519_verify_keyed(hipStream_t, const bsx_validator*, const uint8_t*, uint64_t, uint32_t, const uint8_t*, uint32_t, const uint8_t*, uint8_t*, void*, const void*, int64_t,
                                     const uint32_t* rows = nullptr, int w = BSXK_KT_BITS);
// active (enabled and signed) slots of commits 1.. whose public key differs from the first commit's slot of the same index: what a
// fixed-key table built from the first commit's keys cannot serve (host-side compare; api.hip)
uint64_t bsxh_key_mismatches(const bsx_validator* validators, uint64_t n_commits, uint32_t v_max);
// sentinel for bsxk_ed25519_verify_keyed's last argument: no decoded R, and (with a scratch) prefer the form with the least total work
#ifndef BSXK_ED_THROUGHPUT
#define BSXK_ED_THROUGHPUT (reinterpret_cast<const void*>(static_cast<uintptr_t>(1)))
#endif
uint64_t bsxk_ed25519_rdec_bytes(uint64_t);
hipError_t bsxk_ed25519_decode_r(hipStream_t, const bsx_validator*, uint64_t, void*);
hipError_t bsxk_ed25519_btable(hipStream_t, uint8_t*);
uint64_t bsxk_ed25519_btable_bytes();
uint64_t bsxk_ed25519_scratch_bytes(uint64_t);
hipError_t bsxk_commit_tally(hipStream_t, const bsx_validator*, uint32_t, uint32_t, const uint8_t*, const uint8_t*, bsx_commit_result*, const bsxk_unit_dst*);
// the signature-dependent half of the tally behind an EARLY bsxk_commit_tally(ok = nullptr): verdict bools, signed sums, 2/3 rule
hipError_t bsxk_commit_sums(hipStream_t, const bsx_validator*, uint32_t, uint32_t, const uint8_t*, const uint8_t*, bsx_commit_result*, const bsxk_unit_dst*);
hipError_t bsxk_skip_check(hipStream_t, uint32_t, uint32_t, const bsx_shared_ctx*, const bsx_header*, uint64_t, const uint8_t*,
                           const bsx_validator*, const bsx_validator*, const uint8_t*, bsx_commit_result*, const bsx_commit_result*,
                           uint32_t*, uint8_t*, const uint32_t*, const uint8_t*, uint32_t, const bsxk_unit_dst*);
// header-field inclusion proofs (tendermintx *ProofVariable: aunts, path digests, leaf, leaf length) of the SKIP / STEP units
struct bsxk_proof_spec {
    uint8_t header;       // which of the item's headers (0 / 1)
    uint8_t field;        // header field index (<= 11)
    uint16_t cap;         // leaf capacity in the record
    uint32_t off;         // byte offset of the proof record inside the unit
    uint32_t len_word;    // index of the leaf-length word
    uint32_t zero_if;     // bit set in flags[item] -> the proof is the hint's all-zero padding (leaf length = cap_zero_len)
};
struct bsxk_field_proofs_args {
    uint32_t n_items;
    const bsx_header* headers;       // item r: headers[r * headers_per_item + index(r, h)]
    uint64_t headers_per_item;
    const bsx_shared_ctx* ranges;    // optional: header 1 of item r is at index end_block - start_block (header 0 at index 0)
    const uint32_t* target_idx;      // optional: overrides that index
    const uint32_t* flags;           // optional, per item
    bsxk_unit_dst unit;
    uint32_t n_proofs;
    bsxk_proof_spec proofs[6];
    const uint8_t* zero_paths;       // the all-zero data_hash proof's path digests (ctx->zero_paths)
};
hipError_t bsxk_field_proofs(hipStream_t, const bsxk_field_proofs_args*);
// step conditions of CombinedStepCircuit (next_header.rs:25-46) + prove_next_header_data_commitment (builder.rs:411-443), one item
struct bsxk_step_args {
    const bsx_header* headers;       // [prev, next]
    const uint8_t* hashes;           // their hashes, 64 bytes
    const uint8_t* input40;          // device copy of the public input
    const bsx_commit_result* commit; // the next header's commit
    uint32_t* step_status;           // out: bsx_status of the step verification
    uint32_t* dc_status;             // out: 0 or BSX_A10_NEXT_HEADER
    uint8_t* output64;               // out
    bsxk_unit_dst unit;              // STEP unit (required: the data-hash proof is read from it)
    uint32_t chain_id_len;
    uint8_t chain_id[52];
};
hipError_t bsxk_step_check(hipStream_t, const bsxk_step_args*);
hipError_t bsxk_encode_tuple(hipStream_t, const uint8_t*, uint64_t, uint8_t*);
// packed wire headers -> bsx_header records (kernels_misc.hip k_unpack_headers): desc = (block offset | 0xffffffff, n headers) per slot
hipError_t bsxk_unpack_headers(hipStream_t, const uint8_t* packed, const uint32_t* desc, uint32_t n_slots, uint32_t hpr, const uint32_t* wipe_to, bsx_header* out);
hipError_t bsxk_data_commitment(hipStream_t, const uint8_t*, uint32_t, uint64_t, uint64_t, uint8_t*, uint32_t*);
hipError_t bsxk_fill_end_hash(hipStream_t, uint32_t, bsx_shared_ctx*, const uint8_t*, uint64_t, const uint32_t*, uint8_t*, uint8_t*, uint64_t);
int bsxk_tally_vmax(void);
hipError_t bsxk_skip_eval(hipStream_t, const bsx_validator*, const bsx_validator*, uint32_t, uint32_t, bsx_skip_eval*);
uint64_t bsxk_commit_fold_scratch_bytes(uint32_t);
hipError_t bsxk_commit_fold(hipStream_t, const bsx_commit_result*, uint32_t, uint32_t, void*, bsx_commit_fold*);
}

// number of distinct hardware queues the given streams sit on (measured; calibrate.hip); < 0: a bsx_status error code negated
extern "C" int bsxk_queue_groups(hipStream_t* streams, uint32_t n, uint32_t* groups);
// compute units of the CURRENT device (hipDeviceAttributeMultiprocessorCount, cached per device; 256 if the query fails)
extern "C" uint32_t bsxk_compute_units();

extern "C" {
hipError_t bsxk_poseidon_permute(hipStream_t, const uint64_t*, uint64_t, uint64_t*);
hipError_t bsxk_leaf_hashes(hipStream_t, const bsx_witness_layout*, uint32_t, const uint8_t*, const uint64_t*, uint32_t, uint32_t, int,
                            uint64_t, uint64_t*);
hipError_t bsxk_merkle_caps(hipStream_t, uint64_t*, uint32_t, uint64_t, uint32_t, uint32_t);
hipError_t bsxk_merkle_one_level(hipStream_t, uint64_t*, uint64_t);
}
