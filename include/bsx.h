/*
 * bsx.h — C ABI of the MI355X-native witness engine for Blobstream X `header_range`.
 *
 * This is the drop-in boundary for the ONE hot path of succinctlabs/blobstreamx
 * (circuits/header_range.rs + circuits/data_commitment.rs + circuits/builder.rs +
 * the host side of the hint, circuits/input.rs).  Every entry point names the reference
 * interface it replaces (file:line under /root/reference).  Plain pointers and sizes only;
 * no C++/torch types cross this line.  All multi-byte integers are host-endian (little)
 * except where the reference is big-endian (EVM-packed u64, the data-root tuple height).
 *
 * Two tiers:
 *   bsx_*      host-pointer tier: synchronous, copies in/out — what a Rust `AsyncHint` /
 *              builder shim binds.  Small inputs and results travel through the
 *              context's own page-locked staging; LARGE buffers (the headers of a range, the
 *              witness) are copied straight from / to the caller's memory — page-lock them
 *              (hipHostMalloc / hipHostRegister) and a header_range_2048 witness arrives in
 *              2.5 ms instead of 12.
 *   bsx_dev_*  device-pointer tier: every pointer marked `d_` is HIP device memory, the call
 *              only enqueues kernels on `stream` (a hipStream_t passed as void*) and returns.
 *              The host tier is implemented on top of it.
 *
 * Ownership: the caller allocates every buffer; the library never frees caller memory and
 * never keeps a caller pointer after return.  Errors: every function returns a bsx_status;
 * nothing aborts (the reference panics: circuits/input.rs:93,100,104,154,173).  A
 * thread-local message is available from bsx_last_error().  The library has NO CPU
 * compute fallback: without a HIP device bsx_init fails with BSX_ERR_NO_DEVICE.
 */
#ifndef BSX_H
#define BSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0x00030100 (round 5): additions only — the coalescing front end (bsx_batcher_*, bsx_submit_*, bsx_wait / bsx_poll, bsx_enable_coalescing,
 * bsx_map_job), the *_cap witness entry points, key tables with a digit width (bsx_*_keytable*_w, BSX_COMMITS_KEYTABLE_WIDE),
 * BSX_COMMITS_TALLY_BESIDE.  Every round-4 entry point keeps its signature and meaning.
 * 0x00030200 (round 6): additions only — bsx_pipeline_timing2 (the all-gather statistics; bsx_pipeline_timing_result is the 32-byte
 * round-4 struct again: round 5 had grown it in place, an overrun for hosts built against the round-4 header), page-locked / packed
 * header uploads of the coalescing front end. */
#define BSX_VERSION 0x00030200

/* ------------------------------------------------------------------ constants (circuits/consts.rs) */
#define BSX_HASH_SIZE 32                 /* consts.rs:1  HASH_SIZE */
#define BSX_PROTOBUF_HASH_SIZE 34        /* consts.rs:4  PROTOBUF_HASH_SIZE_BYTES */
#define BSX_PROTOBUF_BLOCK_ID_SIZE 72    /* consts.rs:7  PROTOBUF_BLOCK_ID_SIZE_BYTES */
#define BSX_HEADER_PROOF_DEPTH 4         /* consts.rs:10 HEADER_PROOF_DEPTH */
#define BSX_ENC_DATA_ROOT_TUPLE_SIZE 64  /* consts.rs:18 ENC_DATA_ROOT_TUPLE_SIZE_BYTES */
#define BSX_BLOCK_HEIGHT_INDEX 2         /* consts.rs:21 */
#define BSX_LAST_BLOCK_ID_INDEX 4        /* consts.rs:22 */
#define BSX_DATA_HASH_INDEX 6            /* consts.rs:23 */
#define BSX_HEADER_FIELDS 14             /* tendermint 0.33.2 Header::hash: 14 Merkle leaves */
#define BSX_MAX_BATCH 256                /* largest BATCH_SIZE one workgroup folds (bins use 32/64) */
#define BSX_VALIDATOR_MSG_MAX 124        /* tendermintx VALIDATOR_MESSAGE_BYTES_LENGTH_MAX [UPSTREAM] */
#define BSX_VALIDATOR_LEAF_MAX 48        /* SimpleValidator protobuf: 0a 22 0a 20 pk32 10 varint(<=10) */

/* ------------------------------------------------------------------ status codes */
typedef enum bsx_status {
    BSX_OK = 0,
    BSX_ERR_NO_DEVICE = 1,       /* no HIP device / HIP runtime error at init: there is no CPU fallback */
    BSX_ERR_HIP = 2,             /* a HIP call failed (message has the hipError string) */
    BSX_ERR_BAD_ARG = 3,         /* null pointer, zero/invalid size, batch not a power of two, ... */
    BSX_ERR_RANGE_TOO_LONG = 4,  /* input.rs:154 / builder.rs:292-297 (A7) */
    BSX_ERR_BAD_HEADER = 5,      /* a packed header violates the field-size rules below */
    BSX_ERR_ASSERT = 6,          /* a circuit assertion A1..A10 is false; see the returned masks */
    BSX_ERR_BAD_SIGNATURE = 7,   /* a `signed` validator's Ed25519 signature does not verify */
    BSX_ERR_VOTING_POWER = 8,    /* 2/3 or 1/3 threshold not met */
    BSX_ERR_UNSUPPORTED = 9
} bsx_status;

/* Assertion bits (the circuit `assert_is_equal` sites in circuits/builder.rs). */
#define BSX_A1_END_GTE_START   (1u << 0)  /* builder.rs:113-114 */
#define BSX_A2_NB_BLOCKS_U32   (1u << 1)  /* builder.rs:128     */
#define BSX_A3_PREV_HEADER     (1u << 2)  /* builder.rs:205-207 */
#define BSX_A4_DATA_HASH_PROOF (1u << 3)  /* builder.rs:210-212 */
#define BSX_A5_END_HEADER      (1u << 4)  /* builder.rs:216-219 */
#define BSX_A6_BATCH_END       (1u << 5)  /* builder.rs:229-232 */
#define BSX_A7_RANGE           (1u << 6)  /* builder.rs:292-297 */
#define BSX_A8_REDUCE_LINK     (1u << 7)  /* builder.rs:350-355 */
#define BSX_A9_FINAL           (1u << 8)  /* builder.rs:401-406 */
#define BSX_A10_NEXT_HEADER    (1u << 9)  /* builder.rs:434     */

/* ------------------------------------------------------------------ data layouts */

/* One Tendermint header as its 14 protobuf-encoded Merkle leaves (tendermint 0.33.2
 * Header::hash, used at circuits/input.rs:250-261 and through get_inclusion_proof at
 * input.rs:175-179,188-195).  Fixed 512-byte record, every field 4-byte aligned so a wave
 * can stage records with 16-byte coalesced loads.  len[i] is the encoded length of field i
 * (0 = empty field, hashed as the empty leaf 0x00).  Capacity rules (violations ->
 * BSX_ERR_BAD_HEADER): every field except 4 is <= 54 bytes (0x00 prefix + field fit one
 * SHA-256 block; the capacities below already guarantee it); field 4 (last_block_id) is <= 76. */
typedef struct bsx_header {
    uint8_t len[BSX_HEADER_FIELDS];
    uint8_t _pad[2];
    uint8_t version[24];        /* 0: Consensus{block,app}            08 .. 10 ..          */
    uint8_t chain_id[52];       /* 1: StringValue                     0a len ..            */
    uint8_t height[12];         /* 2: Int64Value                      08 varint            */
    uint8_t time[20];           /* 3: Timestamp                       08 secs 10 nanos     */
    uint8_t last_block_id[76];  /* 4: BlockID (72 B when parts.total < 128; consts.rs:7)   */
    uint8_t hash[8][36];        /* 5..12: last_commit, data, validators, next_validators,
                                          consensus, app, last_results, evidence: 0a 20 h32 */
    uint8_t proposer[24];       /* 13: BytesValue                     0a 14 addr20         */
} bsx_header;                   /* sizeof == 512 */

/* InclusionProof<HEADER_PROOF_DEPTH, PROTOBUF_HASH_SIZE_BYTES> as the hint emits it
 * (circuits/input.rs:203-217; MerkleInclusionProofVariable{proof, leaf}, vars.rs:17-20).
 * Packed, no padding: the byte image is exactly the variable's serialization. */
typedef struct bsx_data_hash_proof {
    uint8_t aunts[BSX_HEADER_PROOF_DEPTH][BSX_HASH_SIZE];
    uint8_t leaf[BSX_PROTOBUF_HASH_SIZE];
} bsx_data_hash_proof;          /* sizeof == 162 */

/* InclusionProof<HEADER_PROOF_DEPTH, PROTOBUF_BLOCK_ID_SIZE_BYTES> (input.rs:208-217; vars.rs:21-25). */
typedef struct bsx_last_block_id_proof {
    uint8_t aunts[BSX_HEADER_PROOF_DEPTH][BSX_HASH_SIZE];
    uint8_t leaf[BSX_PROTOBUF_BLOCK_ID_SIZE];
} bsx_last_block_id_proof;      /* sizeof == 200 */

/* DataCommitmentSharedCtx (circuits/builder.rs:12-18). */
typedef struct bsx_shared_ctx {
    uint64_t start_block;
    uint64_t end_block;
    uint8_t start_header_hash[BSX_HASH_SIZE];
    uint8_t end_header_hash[BSX_HASH_SIZE];
} bsx_shared_ctx;               /* sizeof == 80 */

/* MapReduceSubchainVariable (circuits/vars.rs:28-36) padded to 128 bytes: the unit that
 * the map stage hands to the reduce stage and that the multi-GPU all-gather moves. */
typedef struct bsx_subchain {
    uint64_t start_block;
    uint64_t end_block;
    uint8_t start_header[BSX_HASH_SIZE];
    uint8_t end_header[BSX_HASH_SIZE];
    uint8_t data_merkle_root[BSX_HASH_SIZE];
    uint32_t is_enabled;        /* 0/1 */
    uint32_t assert_fail;       /* OR of BSX_A* bits that FAILED while producing this record */
    uint32_t first_bad_slot;    /* lowest slot (map) / node (reduce) index that failed, or 0xffffffff */
    uint32_t _pad;
} bsx_subchain;                 /* sizeof == 128 */

/* One validator of one commit, as tendermintx's SkipOffchainInputs / StepOffchainInputs hand
 * it to the circuit [UPSTREAM tendermintx v1.0.0; reference call sites
 * circuits/header_range.rs:42-48,67 and circuits/next_header.rs:32-36,55]. 256 bytes. */
typedef struct bsx_validator {
    uint8_t pubkey[32];
    uint8_t signature[64];
    uint8_t message[BSX_VALIDATOR_MSG_MAX];  /* CanonicalVote sign-bytes, zero padded */
    uint32_t message_len;
    uint64_t voting_power;
    uint8_t enabled;              /* slot holds a validator (index < validator-set size) */
    uint8_t is_signed;            /* block_id_flag == Commit and a signature is present */
    uint8_t present_on_trusted;   /* skip only: this validator is also in the trusted set */
    uint8_t _pad[21];
} bsx_validator;                  /* sizeof == 256 */

/* Result of verifying one commit (skip/step inner loop). */
typedef struct bsx_commit_result {
    uint8_t validators_hash[BSX_HASH_SIZE]; /* Merkle root over the enabled validators' leaves */
    uint64_t total_power;                   /* sum over enabled */
    uint64_t signed_power;                  /* sum over enabled & signed & signature valid */
    uint64_t trusted_signed_power;          /* sum over enabled & signed & present_on_trusted */
    uint32_t n_enabled;
    uint32_t n_signed;
    uint32_t n_bad_signature;               /* signed validators whose signature failed */
    uint32_t first_bad_signature;           /* index or 0xffffffff */
    uint32_t n_bad_message;                 /* signed validators whose message lacks the header hash */
    uint32_t two_thirds_ok;                 /* 3*signed_power > 2*total_power */
    uint32_t power_overflow;                /* the enabled voting powers add up to more than BSX_MAX_TOTAL_VOTING_POWER
                                               (Tendermint's MaxTotalVotingPower = MaxInt64 / 8): the u64 sums above are
                                               meaningless; every entry point that consumes them fails with BSX_ERR_BAD_ARG */
    uint32_t _pad[3];
} bsx_commit_result;                        /* sizeof == 96 */
#define BSX_MAX_TOTAL_VOTING_POWER 1152921504606846975ull   /* (2^63 - 1) / 8 */

/* One context per HIP device (several per device are fine).  Threading: calls on DISTINCT contexts run concurrently.  The
 * host tier keeps per-context state (scratch arena, page-locked staging, the persistent Ed25519 key table, a second stream),
 * so host-tier calls on ONE context serialise on a lock inside the context: sharing a context between threads (e.g. the
 * async hints of a tokio runtime, header_range.rs:180-181) is safe, they simply take turns — use one context per worker for
 * concurrency.  Device-tier calls (bsx_dev_*) only enqueue on the caller's stream and keep no per-call state. */
typedef struct bsx_ctx bsx_ctx;

/* ------------------------------------------------------------------ lifecycle */
uint32_t bsx_version(void);
/* OPTIONAL, once per process, BEFORE its first HIP call (the HIP runtime reads the variable when it initialises): ask for 16
 * hardware queues (GPU_MAX_HW_QUEUES=16) unless the caller's environment already sets the variable.  The pipeline tier drives up
 * to 16 streams whose kernels must overlap; on HIP's default of 4 queues they share queues (measured 3.2 -> 4.8 ms per step).
 * The library never changes the environment on its own — results do not depend on this, only the pipeline's speed.  Returns 1
 * (set), 0 (left alone: already set) or -1. */
int bsx_prepare_process(void);
/* device = HIP ordinal.  Fails with BSX_ERR_NO_DEVICE when no GPU is visible (no CPU fallback). */
int bsx_init(int device, bsx_ctx** out);
void bsx_shutdown(bsx_ctx* ctx);
const char* bsx_last_error(void);
const char* bsx_status_str(int status);
/* Number of HIP devices visible, or 0.  Never fails. */
int bsx_device_count(void);

/* ------------------------------------------------------------------ witness layout queries
 * The witness of one map job is three dense sections: `bytes` (every byte
 * becomes 8 Goldilocks elements, MSB first — plonky2x ByteVariable), `words` (u32 limbs / U32,
 * one element each; U64Variable = lo limb then hi limb) and `bools` (one element each).
 * Expanded size in u64 elements = 8*bytes + words + bools. */
typedef struct bsx_witness_layout {
    uint32_t batch_size;
    uint32_t n_bytes;   /* compact byte section length  */
    uint32_t n_words;   /* compact u32 section length (count of u32) */
    uint32_t n_bools;   /* compact bool section length (count of u8) */
    uint32_t compact_stride;   /* bytes between consecutive jobs' compact witnesses (16-aligned) */
    uint32_t off_words;        /* byte offset of the u32 section inside one compact witness */
    uint32_t off_bools;        /* byte offset of the bool section */
    uint32_t _pad;
    uint64_t n_elements;       /* expanded u64 count per job */
} bsx_witness_layout;
int bsx_map_witness_layout(uint32_t batch_size, bsx_witness_layout* out);
int bsx_reduce_witness_layout(bsx_witness_layout* out);   /* one reduce node */

/* Witness manifest: which elements of an expanded witness belong to which circuit variable — what a plonky2x shim needs
 * to hand every value to its `Variable` (DataCommitmentProofVariable / MapReduceSubchainVariable, circuits/vars.rs:13-36;
 * intermediate variables of circuits/builder.rs:105-271,337-395) without hard-coding offsets.  One entry per variable
 * group; record r of a group occupies elements [element_offset + r*record_stride, .. + elements_per_record).  The entries
 * tile [0, n_elements) exactly once.  batch_size = 0: one reduce node.  entries may be NULL to query the count.
 * Host-side only (works without a GPU). */
#define BSX_KIND_BYTES 0u   /* BytesVariable: 8 BoolVariable elements per byte, MSB first; elements_per_record = 8 * bytes */
#define BSX_KIND_U32 1u     /* u32 limbs: a U64Variable is 2 elements, limb 0 (low) first (builder.rs:124-128) */
#define BSX_KIND_BOOL 2u    /* BoolVariable: one element, 0/1 */
typedef struct bsx_manifest_entry {
    char name[80];              /* variable (group) name as in the reference, `[]` = one record per slot / node */
    char reference[24];         /* file:line under circuits/ that creates it */
    uint32_t kind;
    uint32_t repeat;            /* records in the group: BATCH_SIZE, BATCH_SIZE - 1 or 1 */
    uint64_t element_offset;    /* first element of record 0 */
    uint64_t elements_per_record;
    uint64_t record_stride;     /* elements between consecutive records (== elements_per_record when repeat == 1) */
} bsx_manifest_entry;           /* sizeof == 136 */
int bsx_witness_manifest(uint32_t batch_size, bsx_manifest_entry* entries, uint32_t capacity, uint32_t* out_n);

/* Round 4: the witness of the WHOLE circuit.  CombinedSkipCircuit::define is builder.skip ∘ prove_data_commitment
 * (header_range.rs:42-55): besides the map jobs and reduce nodes above, a proof request's witness holds one COMMIT unit (the
 * per-validator loop of builder.skip against the target header: pubkeys, signatures, messages, SHA-512 digests and reduced
 * challenges, signature / message verdicts, validator leaves and the masked validator-set tree, power sums, 2/3 rule) and one
 * SKIP unit (public inputs / outputs, header-field inclusion proofs for chain id / height / validators_hash with their path
 * digests, the trusted validator set's leaves and tree, the 1/3 overlap rule, every assertion bool) — include/bsx_layout.h
 * documents every variable.  CombinedStepCircuit::define (next_header.rs:25-46) = one COMMIT unit + one STEP unit.
 * Same three dense sections per unit (bytes -> 8 bools MSB first, u32 words, bools), so bsx_dev_expand_witness and the fused
 * Poseidon leaf hashing take them through the layouts below.  [UPSTREAM] tendermintx v1.0.0: ordering = this layout. */
#define BSX_SECTION_MAP 0u      /* param = BATCH_SIZE */
#define BSX_SECTION_REDUCE 1u   /* param ignored */
#define BSX_SECTION_COMMIT 2u   /* param = validator slots V (1..512) */
#define BSX_SECTION_SKIP 3u     /* param = V */
#define BSX_SECTION_STEP 4u     /* param ignored */
int bsx_witness_manifest_section(uint32_t section, uint32_t param, bsx_manifest_entry* entries, uint32_t capacity, uint32_t* out_n);
int bsx_commit_witness_layout(uint32_t v_max, bsx_witness_layout* out);
int bsx_skip_witness_layout(uint32_t v_max, bsx_witness_layout* out);
int bsx_step_witness_layout(bsx_witness_layout* out);
/* u64 elements of the witness bsx_header_range returns: nb_map_jobs map jobs, nb_map_jobs - 1 reduce nodes (level order), then
 * the COMMIT unit of the target commit, then the SKIP unit.  0 on invalid arguments. */
uint64_t bsx_header_range_witness_elements(uint32_t nb_map_jobs, uint32_t batch_size, uint32_t v_max);
/* u64 elements of the witness bsx_next_header returns: the COMMIT unit of the next header's commit, then the STEP unit. */
uint64_t bsx_next_header_witness_elements(uint32_t v_max);

/* ------------------------------------------------------------------ host tier */

/* encode_data_root_tuple — circuits/builder.rs:23-27,82-103.  out = 0x00*24 ‖ height BE ‖ data_hash. */
int bsx_encode_data_root_tuple(bsx_ctx* ctx, const uint8_t data_hash[32], uint64_t height, uint8_t out[64]);

/* get_data_commitment<MAX_LEAVES> — circuits/builder.rs:33-38,105-148.  data_hashes: max_leaves×32.
 * Fails with BSX_ERR_ASSERT (A1/A2) like the circuit. max_leaves must be a power of two <= BSX_MAX_BATCH. */
int bsx_get_data_commitment(bsx_ctx* ctx, const uint8_t* data_hashes, uint32_t max_leaves,
                            uint64_t start_block, uint64_t end_block, uint8_t out_root[32]);

/* Header hashes + the two inclusion proofs per header — tendermint Header::hash and tendermintx
 * InputDataFetcher::get_inclusion_proof as used at circuits/input.rs:175-179,188-195,250-261.
 * Any out pointer may be NULL. */
int bsx_header_hashes(bsx_ctx* ctx, const bsx_header* headers, uint64_t n,
                      uint8_t* out_hashes /* n×32 */,
                      bsx_data_hash_proof* out_dh /* n */, bsx_last_block_id_proof* out_lb /* n */);

/* DataCommitmentInputFetcher::get_data_commitment_inputs<MAX_LEAVES> — circuits/input.rs:57-60,149-271,
 * i.e. the body of `DataCommitmentOffchainInputs<MAX_LEAVES>::hint` (circuits/data_commitment.rs:18-45)
 * with the RPC replaced by the caller's header array: headers[i] is the header at height
 * first_height+i and must cover [start_block, min(end_block, latest_block-2)].
 * Outputs mirror DataCommitmentInputs (input.rs:29-37); out_dh/out_lb hold max_leaves entries. */
int bsx_data_commitment_inputs(bsx_ctx* ctx, const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                               uint64_t latest_block, uint64_t start_block, uint64_t end_block, uint32_t max_leaves,
                               uint8_t out_start_header[32], uint8_t out_end_header[32],
                               bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb,
                               uint8_t out_expected_data_commitment[32]);

/* prove_subchain<BATCH_SIZE> — circuits/builder.rs:45-52,150-271.  One map job.
 * Returns BSX_ERR_ASSERT when a circuit assertion fails (record->assert_fail says which).
 * witness (optional): expanded Goldilocks witness, bsx_map_witness_layout(batch).n_elements u64. */
int bsx_prove_subchain(bsx_ctx* ctx, uint32_t batch_size,
                       const uint8_t start_header[32], const uint8_t end_header[32],
                       const bsx_data_hash_proof* dh, const bsx_last_block_id_proof* lb,
                       uint64_t batch_start_block, uint64_t batch_end_block,
                       uint64_t global_end_block, const uint8_t global_end_header_hash[32],
                       bsx_subchain* out_record, uint64_t* witness);

/* The reduce closure of prove_data_commitment — circuits/builder.rs:337-395 — applied as the
 * binary tree plonky2x mapreduce builds: n (power of two) records -> 1. */
int bsx_reduce(bsx_ctx* ctx, const bsx_subchain* records, uint32_t n, bsx_subchain* out);

/* prove_data_commitment<C, NB_MAP_JOBS, BATCH_SIZE> — circuits/builder.rs:58-67,273-409 — with the
 * hint (data_commitment.rs:22-44 -> input.rs:149-271) served from `headers` (height first_height+i).
 * records (optional): nb_map_jobs map outputs.  witness (optional): nb_map_jobs map-job witnesses,
 * then nb_map_jobs-1 reduce-node witnesses (level order, leaves' parents first). */
int bsx_prove_data_commitment(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size,
                              const bsx_shared_ctx* range,
                              const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                              uint8_t out_data_commitment[32], bsx_subchain* out_result,
                              bsx_subchain* records, uint64_t* witness);

/* prove_next_header_data_commitment — circuits/builder.rs:73-78,411-443. `header` is the header at
 * prev_block_number. */
int bsx_prove_next_header_data_commitment(bsx_ctx* ctx, uint64_t prev_block_number, const uint8_t prev_header_hash[32],
                                          uint64_t next_block_number, const bsx_header* header, uint64_t latest_block,
                                          uint8_t out_data_commitment[32]);

/* Commit verification: the per-validator hot loop inside builder.skip / builder.step
 * (circuits/header_range.rs:42-48, circuits/next_header.rs:32-36; body [UPSTREAM] tendermintx
 * v1.0.0; host twin is_valid_skip at circuits/fetcher.rs:76-80).  n_commits commits of
 * v_max validator slots each.  header_hashes: n_commits×32, the hash every signed message must
 * carry (offset 16, or 25 when a round field is present).  out_sig_ok: n_commits×v_max bytes
 * (1 = signed and valid).  Does not fail on bad signatures: the result says so.  witness: the Goldilocks witness of the loop —
 * per validator the hint's record, the SHA-512 digest and reduced challenge, signature / message verdicts, the SimpleValidator
 * leaf, then the masked validator-set tree, power sums and the 2/3 bool (COMMIT unit, include/bsx_layout.h). */
int bsx_verify_commits(bsx_ctx* ctx, const bsx_validator* validators, uint32_t n_commits, uint32_t v_max,
                       const uint8_t* header_hashes, bsx_commit_result* out_results, uint8_t* out_sig_ok,
                       uint64_t* witness /* optional: n_commits COMMIT units, bsx_commit_witness_layout(v_max).n_elements u64 each */);

/* CombinedSkipCircuit::define — circuits/header_range.rs:32-59: input48 = u64 BE trusted_block ‖
 * bytes32 trusted_header_hash ‖ u64 BE target_block (header_range.rs:33-35); output64 =
 * target_header_hash ‖ data_commitment (header_range.rs:57-58).  The target header is
 * headers[target-first_height]; its commit is target_validators (v_max slots); the trusted set is
 * trusted_validators (pubkey, voting_power, enabled used).  Checks skip conditions
 * [UPSTREAM tendermintx skip]: trusted < target <= trusted + nb_map_jobs*batch_size, signatures,
 * validators_hash of both sets against the headers' field 7, 2/3 of target power signed, more
 * than 1/3 of trusted power signed; the target header's chain-id leaf (field 1: 0a len bytes) equals chain_id — the
 * circuit constant C::CHAIN_ID_BYTES that builder.skip is called with (header_range.rs:42-43; config.rs:6-28),
 * chain_id_len <= 50.  A header of another chain signed by the same keys fails with BSX_ERR_ASSERT.
 * witness (optional): bsx_header_range_witness_elements(nb_map_jobs, batch_size, v_max) u64 — the WHOLE circuit's variables:
 * nb_map_jobs map jobs, nb_map_jobs - 1 reduce nodes (level order), the COMMIT unit of the target commit, the SKIP unit.  It is
 * filled whenever the return code is BSX_OK, BSX_ERR_ASSERT, BSX_ERR_BAD_SIGNATURE or BSX_ERR_VOTING_POWER (a failing witness
 * shows which assertion bool is 0). */
int bsx_header_range(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48],
                     const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                     const bsx_validator* target_validators, const bsx_validator* trusted_validators, uint32_t v_max,
                     const uint8_t* chain_id, uint32_t chain_id_len,
                     uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness);

/* Capacity-checked forms.  Round 4 APPENDED the COMMIT and SKIP / STEP units to the witnesses of bsx_header_range,
 * bsx_next_header and bsx_verify_commits: a host that still sizes its buffer the round-3 way (map jobs + reduce nodes) would be
 * overrun by the trailing device-to-host copies.  These take the buffer's capacity in u64 elements and fail with BSX_ERR_BAD_ARG —
 * before anything is enqueued — when it is smaller than bsx_header_range_witness_elements / bsx_next_header_witness_elements /
 * n_commits * bsx_commit_witness_layout(v_max).n_elements.  New hosts should bind these. */
int bsx_header_range_cap(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48],
                         const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                         const bsx_validator* target_validators, const bsx_validator* trusted_validators, uint32_t v_max,
                         const uint8_t* chain_id, uint32_t chain_id_len,
                         uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness, uint64_t witness_capacity_elements);
int bsx_next_header_cap(bsx_ctx* ctx, const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                        uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max,
                        const uint8_t* chain_id, uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit,
                        uint64_t* witness, uint64_t witness_capacity_elements);
int bsx_verify_commits_cap(bsx_ctx* ctx, const bsx_validator* validators, uint32_t n_commits, uint32_t v_max,
                           const uint8_t* header_hashes, bsx_commit_result* out_results, uint8_t* out_sig_ok,
                           uint64_t* witness, uint64_t witness_capacity_elements);

/* ------------------------------------------------------------------ wire-format ingest
 * Tendermint RPC JSON -> the packed layouts above; host-side byte formatting only (works without a GPU).
 * Replaces serde + tendermint-rs decoding at circuits/input.rs:19-27,67-110,120-145 and circuits/fetcher.rs:44-58,89-132;
 * accepts the reference's fixture files (circuits/fixtures/mocha-4) verbatim. */
const char* bsx_ingest_last_error(void);
/* header.json, or any response carrying result.header / result.signed_header.header */
int bsx_ingest_header_json(const char* json, size_t len, bsx_header* out_header, uint64_t* out_height);
/* signed_block.json (result.{header, commit, validator_set}) or a /commit response plus, in validators_json, the
 * /validators response.  out_validators (optional): v_max slots in validator-set order, each with the sign-bytes it
 * signed; out_block_hash = commit.block_id.hash (the node's own header hash, for cross-checking). */
int bsx_ingest_signed_block_json(const char* json, size_t len, const char* validators_json, size_t validators_len,
                                 bsx_header* out_header, uint8_t out_block_hash[32], bsx_validator* out_validators,
                                 uint32_t v_max, uint32_t* out_n_validators, uint64_t* out_height);
/* data_commitment.json -> the 32-byte commitment the node reports (circuits/input.rs:104-109) */
int bsx_ingest_data_commitment_json(const char* json, size_t len, uint8_t out[32]);

/* ------------------------------------------------------------------ device tier (async; d_* = device memory)
 * Every call only enqueues kernels on `stream` (hipStream_t as void*; NULL = the HIP default stream, i.e. PyTorch's
 * default stream) and returns, so calls are ordered with the caller's own work on that stream.
 * Device status words are ORed into, never cleared: the caller zeroes them (hipMemsetAsync) before a pass. */

/* Device memory for the LARGE streaming buffers of the device tier (the expanded witness: tens of GB written once per
 * pass).  Backed by the HIP virtual-memory API (hipMemCreate + hipMemMap of one physical handle): on MI355X the store
 * bandwidth of a multi-GB buffer depends on where its physical pages lie — hipMalloc'ed buffers of one process ran the
 * same store sweep at 5.5 .. 6.6 TB/s, slices of one big hipMalloc arena at 5.4 .. 6.2 TB/s by offset, VMM-backed buffers
 * at 6.0 .. 6.25 TB/s every time (tools/exp_vmm.hip) — so this replaces the round-1 "allocate candidates and
 * time them" probe with one deterministic allocation.  Zero-filled.  Free with bsx_dev_free (or bsx_shutdown).  A freed block
 * stays mapped inside the context and is recycled by the next bsx_dev_alloc it fits (mapping and unmapping multi-GB ranges over and
 * over ended in GPU memory faults on ROCm 7.2); bsx_trim returns the recycled blocks' memory to the device. */
int bsx_dev_alloc(bsx_ctx* ctx, uint64_t bytes, void** out_ptr);
int bsx_dev_free(bsx_ctx* ctx, void* ptr);

/* Launch tuning of a context (optional; results never depend on it).
 * BSX_TUNE_MERKLE_WORKGROUPS: resident workgroups of the header-hashing kernel; its workgroups then stride over the
 *   header groups.  0 (default) = one workgroup per 64 headers, the fastest form when the kernel has the GPU to itself
 *   (4 waves per SIMD: the register file is full).  A pipeline that runs the HBM-bound witness expansion of one chunk
 *   beside the hashing of the next (blobstreamx_amd/engine.py) sets 512 = 2 workgroups per CU, which leaves half of the
 *   register file to the expansion's waves: +2 % whole-step throughput on MI355X. */
#define BSX_TUNE_MERKLE_WORKGROUPS 1u
/* BSX_TUNE_HOST_GRAPHS (default 0): 1 = bsx_header_range captures its launch sequence into a hipGraph the second time a shape
 *   (circuit sizes, range length, chain id) is requested and replays it from then on: one graph launch per proof request
 *   instead of ~25 kernel launches and 8 event operations.  Off by default: measured on ROCm 7.2 / MI355X the replay takes
 *   0.56 ms against 0.33 ms for the direct launches — the runtime executes the graph's three parallel branches (hashing chain,
 *   commit check, R decoding + trusted tally) one after the other, so the request pays the SUM of the chains. */
#define BSX_TUNE_HOST_GRAPHS 2u
int bsx_set_tuning(bsx_ctx* ctx, uint32_t key, uint64_t value);

/* Give back what the host tier keeps between calls: the scratch arena (grown to the largest call seen, e.g. 115 MB after a
 * witness download), the persistent fixed-key Ed25519 table (5.8 MB per validator slot) and a captured launch graph.  The next
 * host-tier call re-creates what it needs (one cold key-table build, ~2 ms at V = 100).  Waits for the device.  Returns the
 * bytes of device memory released in *freed_bytes (may be NULL). */
int bsx_trim(bsx_ctx* ctx, uint64_t* freed_bytes);

/* P5: four lanes (of four waves) per header. d_hashes n*32; d_dh_aunts / d_lb_aunts n*128 (4 aunts, leaf-adjacent first); any may be
 * NULL.  d_status (1 u32, optional): bit0 = a header violates the field-size rules.
 * d_paths (optional, n * BSX_HEADER_PATH_BYTES): the 7 distinct digests of the two inclusion-proof paths of each header
 * [L6, n67, L4, n45, n4567, left, root] — what prove_subchain's get_root_from_merkle_proof calls (builder.rs:189-199)
 * materialise for a proof taken from this header; give them to bsx_dev_assemble_inputs to save re-deriving them. */
#define BSX_HEADER_PATH_BYTES 224
int bsx_dev_header_merkle(bsx_ctx* ctx, void* stream, const bsx_header* d_headers, uint64_t n,
                          uint8_t* d_hashes, uint8_t* d_dh_aunts, uint8_t* d_lb_aunts, uint8_t* d_paths, uint32_t* d_status);

/* The hint for many map jobs at once (input.rs:149-271): job j of range r covers
 * [S_r + j*B, S_r + j*B + span) (span == B for map jobs, input.rs:154 requires span <= B).  Headers of range r start
 * at d_headers[r*headers_per_range] = height S_r + header_first_rel (0 unless a device holds only the headers of
 * its own job slice); d_latest[r] is the chain head the hint clamps against
 * (input.rs:160-162).  Only jobs [job_first, job_first+job_count) are written (multi-GPU sharding); compact witnesses
 * are indexed [range][job - job_first] with stride bsx_map_witness_layout(B).compact_stride.  Writes the
 * DataCommitmentProofVariable part, the ctx and the batch bounds of each job's compact witness.
 * d_status bits: 1 = inclusion-proof leaf is not 34/72 bytes (input.rs:173,190), 2 = headers not supplied / latest < 2.
 * d_paths (optional, from bsx_dev_header_merkle over the same headers): also writes every slot's dh_path / lb_path
 * digests (padding slots: the digests of the all-zero proof), after which bsx_dev_prove_subchain may be called with
 * BSX_SUBCHAIN_PATHS_FROM_HINT. */
int bsx_dev_assemble_inputs(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t nb_map_jobs, uint32_t batch_size,
                            uint32_t job_first, uint32_t job_count, uint32_t span,
                            const bsx_shared_ctx* d_ranges, const uint64_t* d_latest,
                            const bsx_header* d_headers, uint64_t headers_per_range, uint64_t header_first_rel,
                            const uint8_t* d_hashes, const uint8_t* d_dh_aunts, const uint8_t* d_lb_aunts,
                            uint8_t* d_compact, uint32_t* d_status, const uint8_t* d_paths);

/* prove_subchain for n_ranges*job_count map jobs (builder.rs:150-271 incl. get_data_commitment :105-148).  Reads the
 * proofs, start/end header and batch bounds from each compact witness, global end block/hash from d_ranges[r];
 * completes the compact witness and writes d_records[range][job].  flags: see below. */
int bsx_dev_prove_subchain(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t batch_size, uint32_t job_count,
                           const bsx_shared_ctx* d_ranges, uint8_t* d_compact, bsx_subchain* d_records, uint32_t flags);
/* flags: the proofs in d_compact were produced by bsx_dev_assemble_inputs WITH d_paths, i.e. their path digests are nodes
 * of header trees this library hashed itself and already sit in the slot section: get_root_from_merkle_proof's 19
 * compressions per slot are not repeated (the witness is bit-identical; every link assertion A3-A6 is still evaluated on
 * those digests).  Never set it for proofs that came from a caller. */
#define BSX_SUBCHAIN_PATHS_FROM_HINT 1u
/* with BSX_SUBCHAIN_PATHS_FROM_HINT: keep the stages in separate launches (tuple leaf hashes, one launch per wide tree
 * level, predicates) instead of the single launch that is the default.  The single launch is 40 % faster on its own
 * (0.235 -> 0.137 ms per 262,144 slots) but holds 4 waves x 128 registers per SIMD; a pipeline that runs an HBM-bound
 * witness expansion beside this call (engine.py) loses 0.5 % per step to it and sets this flag.  Same results. */
#define BSX_SUBCHAIN_SEPARATE_LAUNCHES 2u

/* Binary reduce (builder.rs:337-395) of `n` consecutive records per range -> 1, n a power of two <= 256.
 * d_reduce_compact (optional) receives n-1 reduce-node compact witnesses per range. */
int bsx_dev_reduce(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t n,
                   const bsx_subchain* d_records, bsx_subchain* d_out, uint8_t* d_reduce_compact);
/* Same fold with record k of range r at d_records[r*stride_range + k*stride_record] (strides in records).  The top
 * fold of the multi-GPU path reads the all-gather result [rank][range] in place: d_records = gathered + first owned
 * range, stride_range = 1, stride_record = ranges per rank-block — no transposition pass on the data path. */
int bsx_dev_reduce_strided(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t n, const bsx_subchain* d_records,
                           uint64_t stride_range, uint64_t stride_record, bsx_subchain* d_out,
                           uint8_t* d_reduce_compact);

/* Final assertions of prove_data_commitment (builder.rs:292-297,400-406) + the 64-byte public output
 * (header_range.rs:57-58).  d_target_hashes (optional, n_ranges*32): first half of the output; NULL = ctx end hash.
 * d_status: n_ranges u32 (OR of failed BSX_A* bits; overwritten). */
int bsx_dev_finalize(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t nb_map_jobs, uint32_t batch_size,
                     const bsx_shared_ctx* d_ranges, const bsx_subchain* d_results, const uint8_t* d_target_hashes,
                     uint8_t* d_output64, uint32_t* d_status);

/* P10: compact -> Goldilocks.  n_jobs compact witnesses (layout) -> n_jobs*layout.n_elements u64, 16-byte aligned. */
int bsx_dev_expand_witness(bsx_ctx* ctx, void* stream, const bsx_witness_layout* layout, uint32_t n_jobs,
                           const uint8_t* d_compact, uint64_t* d_witness);

/* d_ranges[r].end_header_hash := d_hashes[r*headers_per_range + (end_block - start_block)] (the target header hash
 * builder.skip returns and prove_data_commitment consumes, header_range.rs:42-55).  d_target_index (optional,
 * n_ranges u32) overrides the position of the target header inside the range's header block.  d_target_hashes
 * (optional, n_ranges*32) receives the same hashes densely, the layout bsx_dev_commit_tally / bsx_dev_finalize take.
 * d_hashes_copy (optional, n_ranges*headers_per_range*32) receives a copy of the ranges' hashes: a consumer on another
 * stream (the commit check) can then keep reading them while d_hashes is rewritten by the next pass. */
int bsx_dev_fill_end_hash(bsx_ctx* ctx, void* stream, uint32_t n_ranges, bsx_shared_ctx* d_ranges,
                          const uint8_t* d_hashes, uint64_t headers_per_range, const uint32_t* d_target_index,
                          uint8_t* d_target_hashes, uint8_t* d_hashes_copy);

/* P6: h = SHA512(R ‖ A ‖ M) mod L per validator slot. d_h: n*32 (LE scalar), d_digest (optional) n*64. */
int bsx_dev_sha512_challenge(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint64_t n,
                             uint8_t* d_h, uint8_t* d_digest);
/* P7: [s]B == R + [h]A per validator slot. d_ok: n bytes (1 valid, 0 invalid or not signed/enabled). */
int bsx_dev_ed25519_verify(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h,
                           uint64_t n, uint8_t* d_ok);
/* P7, fixed-key form.  A range batch is signed by one validator set, so the per-key work of P7 (decoding A and its
 * multiples table) is done once per key instead of once per signature.  bsx_dev_ed25519_keytable builds the table
 * for the public keys of d_validators[0..n_keys) into d_table (bsx_ed25519_keytable_bytes(n_keys) bytes — 5.8 MB per key:
 * j * 2^(12k) * (-A), k < 22, j <= 2048, affine, one cache line per entry — 128-byte aligned; 2.2 ms for 100 keys); the
 * table of the base point B (16-bit digits, 64 MB) belongs to the context (built by bsx_init).
 * bsx_dev_ed25519_verify_keyed then checks n = n_commits*v_max slots, slot i of every commit against table row i:
 * 22 + 16 mixed point additions per signature, no doubling, no decompression.  A slot whose public key differs from its table row (validator-set change inside the batch, or
 * i >= n_keys) is verified by the generic per-signature path inside the same kernel, so the accept set is
 * exactly bsx_dev_ed25519_verify's for any input.
 * d_table persists between calls: its first n_keys * 64 bytes (the key records) must be ZERO before the first call;
 * afterwards bsx_dev_ed25519_keytable rebuilds only the rows whose public key (or n_keys) changed since the previous call
 * on the same buffer — a validator set is stable for hours, so steady-state calls cost one key compare per row.  Zero the
 * key records again to force a rebuild. */
uint64_t bsx_ed25519_keytable_bytes(uint32_t n_keys);
/* The digit width of a key table is a property of the table (round 5).  BSX_KEYTABLE_BITS (the plain entry points above and below;
 * every request-driven path: a new validator set's table is built on the spot, 22 us per key) or BSX_KEYTABLE_BITS_WIDE (64 MB and
 * 0.24 ms per key; mode S, BSX_COMMITS_KEYTABLE_WIDE).  Other widths: BSX_ERR_BAD_ARG / 0 bytes. */
#define BSX_KEYTABLE_BITS 12u
#define BSX_KEYTABLE_BITS_WIDE 16u
uint64_t bsx_ed25519_keytable_bytes_w(uint32_t n_keys, uint32_t digit_bits);
int bsx_dev_ed25519_keytable_w(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_keys, void* d_table, uint32_t digit_bits);
int bsx_dev_ed25519_verify_keyed_w(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n, uint32_t v_max,
                                   const void* d_table, uint32_t n_keys, uint8_t* d_ok, void* d_scratch, uint32_t digit_bits);
int bsx_dev_ed25519_keytable(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_keys,
                             void* d_table);
int bsx_dev_ed25519_verify_keyed(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h,
                                 uint64_t n, uint32_t v_max, const void* d_table, uint32_t n_keys, uint8_t* d_ok,
                                 void* d_scratch);
/* P7, latency form for small batches (one proof: <= 100 signatures; a pipelined chunk: a few thousand), where a launch takes
 * as long as ONE signature's dependent chain.  bsx_dev_ed25519_decode_r decodes every R strictly (RFC 8032 §5.1.3) into
 * d_decoded_r (bsx_ed25519_decoded_r_bytes(n) bytes, 16-byte aligned) — independent of the challenges, so a caller runs it
 * early, off the critical path; bsx_dev_ed25519_verify_keyed_r then sums the 38 table entries of a signature on 8-16 lanes
 * and compares projectively with the decoded R: no field inversion in the chain (0.35 -> ~0.05 ms per 12,800 signatures).
 * Same verdicts as bsx_dev_ed25519_verify for any input (slots whose key differs from their table row fall back). */
uint64_t bsx_ed25519_decoded_r_bytes(uint64_t n);
int bsx_dev_ed25519_decode_r(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint64_t n, void* d_decoded_r);
int bsx_dev_ed25519_verify_keyed_r(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n,
                                   uint32_t v_max, const void* d_table, uint32_t n_keys, const void* d_decoded_r, uint8_t* d_ok);
/* d_scratch (optional, bsx_ed25519_verify_scratch_bytes(n) bytes, 16-byte aligned): with it the signature lanes stop
 * before the point encoding (a field inversion, a third of a verification) and a second kernel encodes 8 results per lane
 * with ONE inversion (Montgomery's trick); the verdicts are identical.  NULL: every lane inverts for itself. */
uint64_t bsx_ed25519_verify_scratch_bytes(uint64_t n);
/* CombinedStepCircuit::define — circuits/next_header.rs:25-46 (bin/next_header{,_mocha}.rs).  input40 =
 * prev_block_number (u64 big endian) ‖ prev_header_hash; output64 = next_header_hash ‖ data_commitment.
 * builder.step (:32-36) is [UPSTREAM] tendermintx v1.0.0; checked here: prev_header hashes to
 * prev_header_hash; next_header's height is prev + 1 and its last_block_id points at prev_header; the supplied
 * validator set hashes to next_header.validators_hash and to prev_header.next_validators_hash; every signed
 * validator's Ed25519 signature verifies over a message carrying the next header hash; signed power > 2/3; the next
 * header's chain-id leaf equals chain_id (C::CHAIN_ID_BYTES, next_header.rs:32-33).
 * The data commitment is prove_next_header_data_commitment (builder.rs:411-443).  Failure codes as bsx_header_range. */
int bsx_next_header(bsx_ctx* ctx, const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                    uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max,
                    const uint8_t* chain_id, uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit,
                    uint64_t* witness /* optional: bsx_next_header_witness_elements(v_max) u64 = COMMIT unit, STEP unit */);

/* ------------------------------------------------------------------ mode S: a commit on EVERY header, sharded with the headers
 * BASELINE configs #4/#5 ("N headers x V validators"): N back-to-back next_header / skip verifications (circuits/next_header.rs:
 * 25-47 per header; batches are independent: builder.rs:305-336).  One call = the whole commit check of n_commits commits of
 * v_max validator slots on `stream`: SHA-512 challenges, fixed-key Ed25519 (tables of the first commit's keys in d_keytable —
 * persistent, bsx_ed25519_keytable_bytes(v_max) bytes, key records zero before the first call — slots whose key differs fall
 * back to the generic path inside the kernel), batch-inverted encodings, tallies + validator-set hashes, and the FOLD of the
 * commit results into one 128-byte record.  Across GPUs rank g verifies commits [g*N/world, (g+1)*N/world) of the range and ONE
 * all-gather of the 128-byte folds tells every rank whether the whole range verified (bench.py --mode S --gpus N).
 * d_scratch: bsx_dev_verify_commits_scratch_bytes(n_commits, v_max) bytes, 256-byte aligned.  n_commits <= BSX_COMMIT_FOLD_MAX. */
#define BSX_COMMIT_FOLD_MAX 2048u
typedef struct bsx_commit_fold {
    uint8_t root[32];            /* SHA-256 tree over the per-commit digests (kernels_misc.hip k_commit_fold) */
    uint64_t n_commits;
    uint64_t n_ok;               /* commits with 2/3 signed and no bad signature / message / power overflow */
    uint64_t n_signatures_ok;    /* signed validators whose signature verified */
    uint32_t first_index;        /* global index of this slice's first commit */
    uint32_t first_failing;      /* lowest global index of a commit that is not ok, or 0xffffffff */
    uint32_t _pad[16];
} bsx_commit_fold;               /* sizeof == 128 */
uint64_t bsx_dev_verify_commits_scratch_bytes(uint32_t n_commits, uint32_t v_max);
int bsx_dev_verify_commits(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_commits, uint32_t v_max,
                           const uint8_t* d_header_hashes, uint32_t first_index, void* d_keytable, void* d_scratch, uint8_t* d_ok,
                           bsx_commit_result* d_results, bsx_commit_fold* d_fold,
                           uint8_t* d_commit_compact /* optional, 16-byte aligned, n_commits * bsx_commit_witness_layout(v_max).compact_stride
                                                        bytes, ZERO before the first call on the buffer: every commit's COMMIT unit in compact
                                                        form (config #5's witness; bsx_dev_expand_witness turns it into Goldilocks elements) */,
                           uint32_t flags /* BSX_COMMITS_*: promises of a caller whose validators are resident across calls */);
/* The caller has run bsx_dev_ed25519_keytable(d_validators, v_max, d_keytable) since the validators' keys last changed: the call
 * skips the per-call key compare of the table rows (one launch + a no-op build launch per call otherwise). */
#define BSX_COMMITS_KEYTABLE_READY 1u
/* Every enabled and signed slot i of every commit carries the public key of the FIRST commit's slot i (the caller compared them
 * when it uploaded the validators): no slot can be deferred to the generic per-signature kernel, whose scan launch is skipped.  A
 * false promise never turns into an accepted signature: a slot the fixed-key kernel had to defer then counts as a BAD signature. */
#define BSX_COMMITS_KEYS_UNIFORM 2u
/* The validator-set trees (SimpleValidator leaves, masked tree, total power: nothing there depends on the signatures) run BESIDE the
 * signature check on a stream of the context instead of behind it, and only the signature-dependent sums follow the check (round 5:
 * one step in flight loses the tally's 0.1 ms).  The call then uses the context's side stream and two of its events: calls that carry
 * this flag on ONE context must be issued by one thread at a time (any number may be in flight), and not while a host-tier call of
 * another thread runs on that context. */
#define BSX_COMMITS_TALLY_BESIDE 4u
/* d_keytable holds BSX_KEYTABLE_BITS_WIDE-bit digits (bsx_ed25519_keytable_bytes_w / bsx_dev_ed25519_keytable_w below): 16 + 16
 * instead of 22 + 16 table additions per signature for 64 MB instead of 5.8 MB of table per key — the form for a validator set that
 * stays resident over many millions of signatures (2048 x 100: verification 0.54 -> 0.49 ms).  The width is part of every row's
 * layout tag: a table of the other width is never read as this one (its slots are deferred, and counted as bad under
 * BSX_COMMITS_KEYS_UNIFORM). */
#define BSX_COMMITS_KEYTABLE_WIDE 8u
/* flags 0 = the call is self-contained (as in round 3). */

/* ------------------------------------------------------------------ operator skip-target search
 * circuits/fetcher.rs:60-87 find_block_to_request: starting at max_end_block, return the first candidate c with
 * is_valid_skip(start set, c's set, c's commit); otherwise halve the distance to start_block; c - start_block == 1 is
 * returned unconditionally.  The reference fetches one candidate per iteration over HTTP; here the caller hands over the
 * commits of the whole halving sequence and all predicates are evaluated in ONE launch (one workgroup per candidate),
 * the walk itself is the reference's loop.  is_valid_skip is [UPSTREAM] tendermintx v1.0.0 (PARITY UNPINNED): evaluated
 * as "start-set validators that signed the candidate's commit (flag is_signed, signatures not verified — the operator
 * does not verify them either) hold > 1/3 of the start set's power", the rule builder.skip enforces in circuit. */
typedef struct bsx_skip_eval {
    uint64_t overlap_power;       /* start-set power of validators that signed the candidate */
    uint64_t start_total_power;
    uint64_t signed_power;        /* candidate-set power that signed */
    uint64_t target_total_power;
    uint32_t valid;               /* overlap_power * 3 > start_total_power (0 when power_overflow) */
    uint32_t power_overflow;      /* a set's total exceeds BSX_MAX_TOTAL_VOTING_POWER: bsx_find_block_to_request -> BSX_ERR_BAD_ARG */
} bsx_skip_eval;                  /* sizeof == 40 */
/* d_start_validators: v_max records (pubkey, voting_power, enabled); d_candidate_validators: n_candidates * v_max
 * records (pubkey, voting_power, enabled, is_signed). */
int bsx_dev_skip_eval(bsx_ctx* ctx, void* stream, const bsx_validator* d_start_validators,
                      const bsx_validator* d_candidate_validators, uint32_t n_candidates, uint32_t v_max,
                      bsx_skip_eval* d_out);
/* Host tier.  candidate_heights must contain every height the loop visits (BSX_ERR_BAD_ARG otherwise);
 * out_evals (optional) receives the n_candidates evaluations. */
int bsx_find_block_to_request(bsx_ctx* ctx, uint64_t start_block, uint64_t max_end_block,
                              const bsx_validator* start_validators, uint32_t n_candidates,
                              const uint64_t* candidate_heights, const bsx_validator* candidate_validators,
                              uint32_t v_max, uint64_t* out_block, bsx_skip_eval* out_evals);

/* P8+P9: validator-set hash, voting-power tallies and message checks; one workgroup per commit; v_max <= 512.
 * d_header_hashes / d_ok may be NULL (then only validators_hash, total_power, n_enabled are meaningful). */
int bsx_dev_commit_tally(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_commits,
                         uint32_t v_max, const uint8_t* d_header_hashes, const uint8_t* d_ok,
                         bsx_commit_result* d_results);
/* Skip conditions of CombinedSkipCircuit per range ([UPSTREAM] tendermintx skip; fetcher.rs:76-80): trusted header
 * hash == public input, target height leaf, target chain-id leaf == chain_id, signatures, both validator-set hashes against the headers' field 7,
 * 2/3 of the target power, > 1/3 of the trusted power.  d_skip_status[r] = bsx_status; d_target_hashes (optional)
 * receives the target header hashes; d_target_res[r].trusted_signed_power := trusted-set overlap. */
int bsx_dev_skip_check(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t v_max, const bsx_shared_ctx* d_ranges,
                       const bsx_header* d_headers, uint64_t headers_per_range, const uint8_t* d_hashes,
                       const bsx_validator* d_target, const bsx_validator* d_trusted, const uint8_t* d_target_ok,
                       bsx_commit_result* d_target_res, const bsx_commit_result* d_trusted_res,
                       uint32_t* d_skip_status, uint8_t* d_target_hashes, const uint32_t* d_target_index,
                       const uint8_t* chain_id /* HOST pointer, copied into the launch */, uint32_t chain_id_len);

/* ------------------------------------------------------------------ Poseidon over Goldilocks
 * plonky2's `PoseidonGoldilocksConfig` — the hash config of every reference binary (plonky2x DefaultParameters,
 * bin/header_range_2048.rs:1-17); reached from builder.build()/prove and the recursion inside mapreduce
 * (circuits/builder.rs:301-302).  Implementation [UPSTREAM] plonky2 53c5bc3e (Cargo.lock:3110-3112): width 12, rate 8,
 * 4 + 22 + 4 rounds, x^7, circulant + diagonal MDS; constants regenerated (tools/gen_poseidon_constants.py).
 * Field elements are u64; inputs are reduced mod p = 2^64 - 2^32 + 1 on the way in, outputs are canonical.
 * PARITY: the reference tree holds no Poseidon value -> pinned to plonky2's public test vectors only. */
#define BSX_POSEIDON_WIDTH 12
#define BSX_POSEIDON_DIGEST 4              /* HashOut<F>: 4 field elements */
#define BSX_GOLDILOCKS_ORDER 0xFFFFFFFF00000001ull

/* Poseidon::poseidon — n independent 12-element states (in and out may alias). */
int bsx_poseidon_permute(bsx_ctx* ctx, const uint64_t* states, uint64_t n, uint64_t* out);
/* hash_n_to_hash_no_pad::<F, PoseidonPermutation> of n_inputs vectors of `len` elements each (row major) -> n_inputs x 4. */
int bsx_poseidon_hash_no_pad(bsx_ctx* ctx, const uint64_t* elements, uint64_t n_inputs, uint32_t len, uint64_t* out_digests);
/* PoseidonHash::two_to_one — n (left, right) digest pairs -> n digests. */
int bsx_poseidon_two_to_one(bsx_ctx* ctx, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out_digests);
/* Digests a tree stores: all levels from the n_leaves leaf digests up to the 2^cap_height cap nodes, bottom-up,
 * = 2*n_leaves - 2^cap_height (the cap is the LAST 2^cap_height digests).  0 on invalid arguments. */
uint64_t bsx_poseidon_tree_digests(uint32_t n_leaves, uint32_t cap_height);
/* Rows (Merkle leaves) of a witness of n_elements in rows of leaf_len, rounded up to a power of two (MerkleTree::new
 * needs one; plonky2 pads circuits to a power-of-two degree): the padding rows are all-zero elements. */
uint32_t bsx_witness_leaf_count(uint64_t n_elements, uint32_t leaf_len);
/* MerkleTree::<F, PoseidonHash>::new(leaves, cap_height): leaves = rows of leaf_len consecutive elements of `elements`
 * (zero padded to n_leaves rows; leaf digest = hash_or_noop).  out_tree: bsx_poseidon_tree_digests() x 4 u64. */
int bsx_poseidon_merkle_tree(bsx_ctx* ctx, const uint64_t* elements, uint64_t n_elements, uint32_t leaf_len,
                             uint32_t n_leaves, uint32_t cap_height, uint64_t* out_tree);
/* The witness-column commitment of n_jobs witnesses (layout->n_elements u64 each, as bsx_header_range /
 * bsx_prove_data_commitment return them): one Merkle cap per job, out_caps n_jobs x 2^cap_height x 4. */
int bsx_witness_merkle_caps(bsx_ctx* ctx, const bsx_witness_layout* layout, const uint64_t* witness, uint32_t n_jobs,
                            uint32_t leaf_len, uint32_t cap_height, uint64_t* out_caps);

/* device tier */
int bsx_dev_poseidon_permute(bsx_ctx* ctx, void* stream, const uint64_t* d_states, uint64_t n, uint64_t* d_out);
/* Leaf digests of n_trees MATERIALISED witnesses (d_elements: n_trees x n_elements u64) into
 * d_trees[t*tree_stride + 4*j], j < n_leaves (tree_stride in u64 words, >= 4*bsx_poseidon_tree_digests()). */
int bsx_dev_poseidon_leaf_hashes(bsx_ctx* ctx, void* stream, const uint64_t* d_elements, uint32_t n_trees, uint64_t n_elements,
                                 uint32_t leaf_len, uint32_t n_leaves, uint64_t tree_stride, uint64_t* d_trees);
/* FUSED form: the same digests straight from the COMPACT witnesses (what bsx_dev_prove_subchain / bsx_dev_reduce leave in
 * d_compact): elements are produced on the fly, the 64x expanded image is never materialised. */
int bsx_dev_witness_leaf_hashes(bsx_ctx* ctx, void* stream, const bsx_witness_layout* layout, uint32_t n_jobs,
                                const uint8_t* d_compact, uint32_t leaf_len, uint32_t n_leaves, uint64_t tree_stride,
                                uint64_t* d_trees);
/* Upper levels of n_trees trees whose leaf digests are in place, down to the cap (2^cap_height nodes). */
int bsx_dev_poseidon_merkle_caps(bsx_ctx* ctx, void* stream, uint64_t* d_trees, uint32_t n_trees, uint64_t tree_stride,
                                 uint32_t n_leaves, uint32_t cap_height);

/* ------------------------------------------------------------------ device ceilings for roofline reporting
 * Measures, in about 50 ms, what THIS device sustains on the bodies the hot path's kernels are made of (the library's own
 * device functions, alone, 8 waves per SIMD, no memory traffic) and on streaming stores.  A roofline fraction quoted against
 * these is reproducible on any box; bench.py runs it at start-up.  Rates are per second over the whole device. */
typedef struct bsx_calibration {
    double valu_add_u32_lane_ops_per_s;       /* full-rate 2-source 32-bit VALU issue */
    double valu_mad_u64_u32_lane_ops_per_s;   /* the multiply-add every field multiplication is made of */
    double valu_alignbit_lane_ops_per_s;      /* every SHA rotate */
    double sha256_compress_per_s;             /* 64-byte blocks */
    double sha512_compress_per_s;             /* 128-byte blocks */
    double fe25519_mul_per_s, fe25519_sq_per_s;   /* GF(2^255 - 19), 10 x 25.5-bit limbs */
    double goldilocks_mul_per_s;              /* 64 x 64 -> mod 2^64 - 2^32 + 1 */
    double hbm_store_bytes_per_s;             /* non-temporal 16-byte stores over 1 GiB */
    uint32_t compute_units, clock_mhz;
} bsx_calibration;                            /* sizeof == 80 */
int bsx_calibrate(bsx_ctx* ctx, bsx_calibration* out);

/* ------------------------------------------------------------------ batched pipeline (the throughput path)
 * R independent header_range instances per step with every input resident in HBM — what a prover farm that keeps a GPU
 * busy calls instead of one bsx_header_range per proof.  The pipeline object owns its device buffers, HIP streams and
 * events; nothing here depends on environment variables or on a host language runtime.  One step over the ranges of a chunk:
 *
 *   header_merkle (input.rs:175-195,250-261) -> fill_end_hash (header_range.rs:42-55) -> hint of every map job
 *   (data_commitment.rs:22-44) -> prove_subchain (builder.rs:150-271,305-336) -> reduce (builder.rs:337-395) ->
 *   [all-gather across GPUs, top fold] -> final asserts + public output (builder.rs:292-297,400-406; header_range.rs:57-58)
 *   -> witness expansion and/or Poseidon caps; on a side stream the commit check of the target header (builder.skip,
 *   header_range.rs:42-48): SHA-512 challenges, fixed-key Ed25519, tallies + validator-set hashes, skip conditions.
 *
 * n_chunks > 1 cuts the step into chunks on their own streams kept in complementary phases by event tokens: the integer-
 * ALU-bound hashing of chunk e+1 runs beside the HBM-bound witness expansion of chunk e; consecutive steps pipeline the same
 * way (steps are NOT joined: commit-check inputs are double-buffered by step parity).  bsx_pipeline_step only enqueues.
 *
 * Multi-GPU: rank g of `world` computes map jobs [g*J/world, (g+1)*J/world) of ALL world*n_ranges ranges, folds
 * them locally, ONE all-gather of a 128-byte record per (range, rank) per chunk, and the owner of a range (range index /
 * n_ranges) runs the last log2(world) reduce levels, the final assertions and that range's commit check.  The collective
 * itself is the caller's (RCCL ncclAllGather, torch.distributed, ... — INTEGRATION.md): see bsx_pipeline_set_allgather. */
/* Threading: one pipeline = one caller at a time (its calls are not re-entrant); different pipelines — on one context or on
 * several — may be driven by different threads concurrently. */
typedef struct bsx_pipeline bsx_pipeline;

#define BSX_PIPE_WITNESS 1u             /* materialise the Goldilocks witness in HBM every step: map jobs + reduce nodes, and with
                                           BSX_PIPE_COMMIT the COMMIT + SKIP units of every owned range (the whole circuit) */
#define BSX_PIPE_COMMIT 2u              /* verify the target commit of every owned range (builder.skip) */
#define BSX_PIPE_CAPS 4u                /* Poseidon Merkle cap of every map-job witness — and with BSX_PIPE_COMMIT of every COMMIT / SKIP
                                           unit — hashed straight from the compact bytes (plonky2 PoseidonGoldilocksConfig; rows of
                                           leaf_len elements, cap_height) */
#define BSX_PIPE_ED_GENERIC 8u          /* per-signature Ed25519 kernel instead of the fixed-key tables (same verdicts) */
#define BSX_PIPE_COMMIT_BESIDE_HASH 16u /* run the whole commit check beside the hashing phase instead of beside the expansion */
#define BSX_PIPE_RECOMPUTE_PATHS 32u    /* prove_subchain re-derives both proof paths per slot (builder.rs:189-199 literally)
                                           instead of taking the digests the header hashing already produced (same witness) */
#define BSX_PIPE_NO_UNITS 64u           /* with BSX_PIPE_COMMIT + BSX_PIPE_WITNESS / _CAPS: do NOT materialise the COMMIT / SKIP units (the round-3
                                           witness: map jobs + reduce nodes only) — the A/B that prices the units' side-stream work (bench.py units_ab) */
typedef struct bsx_pipeline_config {
    uint32_t nb_map_jobs, batch_size, v_max;
    uint32_t n_ranges;                  /* header_range instances this rank OWNS per step */
    uint32_t n_chunks;                  /* pipelined chunks; must divide n_ranges (1 = no chunking) */
    uint32_t rank, world;               /* job-slice sharding; world must divide nb_map_jobs into power-of-two slices */
    uint32_t flags;                     /* BSX_PIPE_* */
    uint32_t leaf_len, cap_height;      /* BSX_PIPE_CAPS: 0, 0 = plonky2 standard_recursion_config (135 wires, cap height 4) */
    uint32_t chain_id_len;              /* C::CHAIN_ID_BYTES (header_range.rs:42-43), at most 50 bytes */
    uint8_t chain_id[52];
    /* launch forms; results never depend on them.  0 = automatic (the measured choice for the configuration) */
    uint32_t tune_merkle_workgroups;    /* resident workgroups of the header-hashing kernel; 0xffffffff = one per 64 headers */
    uint32_t tune_subchain;             /* 1 = prove_subchain as one launch, 2 = its stages in separate launches */
    /* Buffer sets (0 / 1 = one): with K sets step i runs on set i mod K, so step i + 1 starts on its own buffers while step i's
     * chain of short kernels drains — software pipelining ACROSS steps, the form for the compact path (no BSX_PIPE_WITNESS),
     * whose step has nothing HBM-bound to hide behind.  Every set holds the same uploaded inputs; results / buffers are those
     * of the set of the most recent step (bsx_pipeline_buffer: chunk index = set * n_chunks + chunk). */
    uint32_t n_sets;
    uint32_t _reserved;
} bsx_pipeline_config;                  /* sizeof == 112 */

int bsx_pipeline_create(bsx_ctx* ctx, const bsx_pipeline_config* cfg, bsx_pipeline** out);
void bsx_pipeline_destroy(bsx_pipeline* p);

/* Inputs of one step in HOST memory, for all world*n_ranges ranges this rank touches, in global order r = g*n_ranges + k
 * (g = owning rank, k < n_ranges; chunk e takes k in [e*n_ranges/n_chunks, (e+1)*n_ranges/n_chunks) of every g).  The
 * library takes this rank's header slice of every range and, for the ranges it owns, the (trusted, target) headers and both
 * validator sets.  Synchronous; the buffers may be reused on return. */
typedef struct bsx_pipeline_inputs {
    const bsx_header* headers;          /* [world*n_ranges][headers_per_range]: header k of range r = height S_r + k */
    uint64_t headers_per_range;         /* >= nb_map_jobs*batch_size + 1 */
    const bsx_shared_ctx* ranges;       /* [world*n_ranges] start = trusted block / hash, end = target block.  end_header_hash: at world 1
                                           ignored — every step sets it to the hash of the target header it has just hashed, as
                                           builder.skip hands it to prove_data_commitment (header_range.rs:42-55); at world > 1 a
                                           rank whose job slice does not contain the target header works with the caller's value
                                           (the owner's final assertions reject a wrong one) */
    const uint64_t* latest;             /* [world*n_ranges] chain head the hint clamps against (input.rs:160-162) */
    const bsx_validator* target_validators;    /* [world*n_ranges][v_max] — only owned ranges are read (BSX_PIPE_COMMIT) */
    const bsx_validator* trusted_validators;   /* same */
} bsx_pipeline_inputs;
int bsx_pipeline_upload(bsx_pipeline* p, const bsx_pipeline_inputs* in);

/* PCIe-inclusive operation: keep a page-locked host image of the header block and re-upload it on a copy stream EVERY step,
 * overlapped with the previous step's compute (a caller whose inputs are not resident).  Call after bsx_pipeline_upload. */
int bsx_pipeline_enable_input_streaming(bsx_pipeline* p, int on);

/* One-time tuning of WHERE the pipeline's streams run.  HIP binds a stream to a hardware queue; queues that share a dispatch pipe
 * of the command processor are served one kernel at a time, and which queues share is the driver's business — measured on
 * MI355X the same pipeline runs at 77 .. 98 M headers/s by nothing but the position of its four hot streams among the process's
 * queues.  The pipeline therefore owns a pool of 16 streams (bound to their queues in creation order at bsx_pipeline_create) and
 * this call times every candidate assignment with `steps_per_trial` (0 = about 20 ms worth, 3 .. 32) real steps each — results stay valid, steps are
 * steps — and keeps the fastest (~0.7 s at the bench shape).  Optional; without it the first streams of the pool are used.
 * Call after bsx_pipeline_upload.  With world > 1 it is COLLECTIVE: every rank calls it (each step holds the all-gather); the
 * ranks run the same number of steps (with steps_per_trial 0 they agree on one through the all-gather callback) and each keeps
 * its own best assignment. */
typedef struct bsx_pipeline_autotune_result {
    uint32_t n_trials, best_trial, steps_per_trial;
    uint32_t hw_queues;                        /* how many of the pool's 16 streams can run kernels CONCURRENTLY, MEASURED (two spinning waves
                                                  on two streams take twice as long when they share a hardware queue or a dispatch pipe;
                                                  streams are grouped greedily).  <= 4 means HIP's default GPU_MAX_HW_QUEUES: the chunks'
                                                  streams share queues and their phases cannot overlap — call bsx_prepare_process()
                                                  before the process's first HIP call.  11-16 on MI355X with 16 queues */
    double initial_ms, best_ms, worst_ms;      /* per step: the assignment the pipeline had, the one it keeps (re-timed twice as long
                                                  with the two runners-up), the slowest tried */
    uint32_t assignment[16];                   /* pool index of chunk i's main (2 i) and side (2 i + 1) stream */
} bsx_pipeline_autotune_result;                /* sizeof == 104 */
int bsx_pipeline_autotune(bsx_pipeline* p, uint32_t steps_per_trial, bsx_pipeline_autotune_result* out);

/* Enqueue one step over all chunks; returns without waiting.  Steps may be issued back to back. */
int bsx_pipeline_step(bsx_pipeline* p);
/* Block until everything enqueued has finished (also launches commit checks still deferred). */
int bsx_pipeline_join(bsx_pipeline* p);

/* The one collective of the multi-GPU path.  fn must perform an all-gather ORDERED ON `stream`: when work enqueued on
 * `stream` after fn returns runs, d_recv holds [world][bytes_per_rank] (rank-major), block g = rank g's d_send.  With RCCL:
 * `ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, comm, (hipStream_t)stream)`.  Called once per chunk per step, in
 * chunk order, on every rank.  Return 0 on success. */
typedef int (*bsx_allgather_fn)(void* user, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream);
int bsx_pipeline_set_allgather(bsx_pipeline* p, bsx_allgather_fn fn, void* user);
/* The same collective as RCCL's ncclAllGather, called by the library itself on its exchange stream (round 4: no host-language
 * callback on the data path).  nccl_comm = an ncclComm_t of `world` ranks whose rank order is the pipeline's: one the host created
 * with its own RCCL binding, or bsx_rccl_comm_init_rank below.  RCCL is bound at run time (the process's own copy if it has one,
 * else librccl.so): BSX_ERR_UNSUPPORTED when there is none.  NULL clears it.  Replaces the map->reduce hand-off of
 * circuits/builder.rs:337-395 across GPUs. */
int bsx_pipeline_set_rccl(bsx_pipeline* p, void* nccl_comm);
/* Communicator plumbing for hosts without an RCCL binding of their own: rank 0 calls bsx_rccl_get_unique_id and hands the 128
 * bytes to the other ranks by any means (the reference's hosts already talk HTTP); every rank then calls
 * bsx_rccl_comm_init_rank (collective; binds to ctx's device). */
int bsx_rccl_get_unique_id(uint8_t out_id[128]);
int bsx_rccl_comm_init_rank(bsx_ctx* ctx, uint32_t world, const uint8_t id[128], uint32_t rank, void** out_comm);
int bsx_rccl_comm_destroy(void* comm);
/* Runs the configured all-gather once on a 128-byte test pattern and checks every rank's block (collective; also valid at world 1):
 * a wrong communicator / rank order fails here instead of in the first step's assertions. */
int bsx_pipeline_check_allgather(bsx_pipeline* p);

/* Results of the most recent step (joins first).  Every pointer is optional.  Owned ranges are indexed k < n_ranges. */
typedef struct bsx_pipeline_results {
    uint8_t* output64;                  /* [n_ranges][64] target_header_hash ‖ data_commitment (header_range.rs:57-58) */
    uint32_t* range_status;             /* [n_ranges] OR of failed BSX_A* bits */
    uint32_t* skip_status;              /* [n_ranges] bsx_status of the skip verification (BSX_PIPE_COMMIT) */
    bsx_commit_result* commit;          /* [n_ranges] */
    bsx_subchain* records;              /* [world*n_ranges][nb_map_jobs/world] map-job records of this rank's slice, global range order */
    uint32_t header_status, assemble_status;   /* out: OR over chunks of the device status words (0 = clean) */
} bsx_pipeline_results;
int bsx_pipeline_get_results(bsx_pipeline* p, bsx_pipeline_results* out);

/* Device buffers of a chunk, for in-place consumers (a prover reading the witness from HBM) and tests.  Ranges inside a chunk
 * are ordered [g][k - e*Rc].  *out_bytes is 0 when the configuration has no such buffer. */
#define BSX_PIPE_BUF_WITNESS_MAP 0u          /* u64 [ranges][jobs of the slice][layout.n_elements] */
#define BSX_PIPE_BUF_WITNESS_REDUCE_LOCAL 1u /* u64 [ranges][jobs-1 of the slice][reduce layout n_elements] */
#define BSX_PIPE_BUF_WITNESS_REDUCE_TOP 2u   /* u64 [owned ranges][world-1][...] */
#define BSX_PIPE_BUF_COMPACT 3u              /* compact witnesses [ranges][jobs] (bsx_map_witness_layout().compact_stride) */
#define BSX_PIPE_BUF_TREES 4u                /* BSX_PIPE_CAPS: u64 [map jobs][bsx_poseidon_tree_digests()][4]; the cap is each tree's tail */
#define BSX_PIPE_BUF_PARTIAL 5u              /* locally folded record of every range, 128 B each (the all-gather's send buffer) */
#define BSX_PIPE_BUF_HEADERS 6u              /* the chunk's header block */
#define BSX_PIPE_BUF_RECORDS 7u              /* map-job records [ranges][jobs of the slice] */
#define BSX_PIPE_BUF_GATHERED 8u             /* the all-gather's receive buffer [world][ranges][128] */
#define BSX_PIPE_BUF_REDUCE_COMPACT_LOCAL 9u /* compact witnesses of the local reduce nodes [ranges][jobs-1] */
#define BSX_PIPE_BUF_HASHES 10u              /* header hashes of the chunk's header block, 32 B each */
#define BSX_PIPE_BUF_DH_AUNTS 11u            /* data_hash proof aunts, 128 B per header */
#define BSX_PIPE_BUF_LB_AUNTS 12u            /* last_block_id proof aunts, 128 B per header */
#define BSX_PIPE_BUF_PATHS 13u               /* BSX_HEADER_PATH_BYTES per header (absent with BSX_PIPE_RECOMPUTE_PATHS) */
#define BSX_PIPE_BUF_RANGES 14u              /* bsx_shared_ctx of the chunk's ranges (end_header_hash filled by the step) */
/* BSX_PIPE_COMMIT with BSX_PIPE_WITNESS and / or BSX_PIPE_CAPS: builder.skip's variables of every owned range (round 4) */
#define BSX_PIPE_BUF_WITNESS_COMMIT 15u      /* u64 [owned ranges][bsx_commit_witness_layout(v_max).n_elements]: the target commit's COMMIT unit */
#define BSX_PIPE_BUF_WITNESS_SKIP 16u        /* u64 [owned ranges][bsx_skip_witness_layout(v_max).n_elements] */
#define BSX_PIPE_BUF_COMPACT_COMMIT 17u      /* compact COMMIT units [owned ranges][compact_stride] */
#define BSX_PIPE_BUF_COMPACT_SKIP 18u        /* compact SKIP units */
#define BSX_PIPE_BUF_TREES_COMMIT 19u        /* BSX_PIPE_CAPS: Poseidon trees of the COMMIT units, u64 [owned ranges][digests][4] (rows of leaf_len
                                                elements; cap height = min(cap_height, log2 leaves)); the cap is each tree's tail */
#define BSX_PIPE_BUF_TREES_SKIP 20u          /* same for the SKIP units */
int bsx_pipeline_buffer(bsx_pipeline* p, uint32_t chunk, uint32_t which, void** out_d_ptr, uint64_t* out_bytes);

/* Kernel timing with HIP events on the launch streams: when on, every step brackets prove_subchain, the map-job witness
 * expansion, the Poseidon commitment and (world > 1) the all-gather of every chunk.  bsx_pipeline_timing joins, returns the average launch durations
 * (ms; 0 when not applicable) over the steps since the last call and resets. */
int bsx_pipeline_set_timing(bsx_pipeline* p, int on);
typedef struct bsx_pipeline_timing_result {
    double prove_subchain_ms, expand_map_ms, caps_ms;
    uint32_t launches;                  /* chunk-steps averaged */
    uint32_t _pad;
} bsx_pipeline_timing_result;           /* sizeof == 32: the round-4 struct, unchanged (round 5 had grown it in place) */
int bsx_pipeline_timing(bsx_pipeline* p, bsx_pipeline_timing_result* out);
/* The same averages plus the all-gather statistics (world > 1).  `out_bytes` = sizeof of the caller's struct: the library writes at most that
 * many bytes, so the struct can grow again without overrunning a host built against an older header (BSX_ERR_BAD_ARG below 32). */
typedef struct bsx_pipeline_timing_result2 {
    double prove_subchain_ms, expand_map_ms, caps_ms;
    uint32_t launches;                  /* chunk-steps averaged */
    uint32_t exchanges;                 /* world > 1: all-gathers timed (one per chunk-step) */
    /* world > 1: the one collective of the path (the map -> reduce hand-off across GPUs, builder.rs:337-395), HIP events on the
     * exchange stream from "this rank's folded records are ready" to "every rank's have arrived" — includes waiting for the slowest rank */
    double allgather_ms_avg, allgather_ms_min, allgather_ms_median, allgather_ms_max;
} bsx_pipeline_timing_result2;          /* sizeof == 64 */
int bsx_pipeline_timing2(bsx_pipeline* p, bsx_pipeline_timing_result2* out, uint32_t out_bytes);

/* ------------------------------------------------------------------ coalescing front end (round 5; completion and uploads: round 6)
 * The reference's own call shape: ONE range per `prove` call under a multi-thread runtime (circuits/header_range.rs:180-181) and
 * ONE hint call per map job — 32 `async fn hint` calls per proof (circuits/builder.rs:325-332 -> circuits/data_commitment.rs:22-44),
 * each followed by prove_subchain (builder.rs:335).  One such call alone is a string of dependent single-wave kernels (0.26 ms for
 * 2-3 % of the GPU), and K callers on K contexts time-slice the same queues.  A batcher COALESCES concurrent requests of one
 * circuit shape: bsx_submit_* copies the request's inputs into the open batch's page-locked staging on the CALLER's thread and
 * returns a ticket at once (the input buffers may be reused; the OUTPUT pointers must stay valid until bsx_wait returns); a worker
 * closes the batch after a short window — an idle GPU waits for a 12 us lull only, never for the window — and runs ONE launch set
 * over its R requests (the host tier's own kernels with n_ranges = R); every ticket completes with ITS OWN status: header / hint
 * status words are per request on the device, so a malformed or tampered request never fails its batch-mates.  A completed batch
 * wakes only its own waiters (a futex word per lane); each waiter copies its own results out, the worker takes out the rest.  n_lanes
 * batches are in flight (the H2D copy of one beside the kernels of another).  bsx_wait returns exactly what the synchronous call would have
 * returned (code and bsx_last_error text).  Thread-safe: any number of threads may submit and wait on one batcher. */
typedef struct bsx_batcher bsx_batcher;
typedef uint64_t bsx_ticket;                /* never 0 */
typedef struct bsx_batcher_config {
    uint32_t nb_map_jobs, batch_size, v_max;   /* the circuit the requests belong to (bin/header_range_2048.rs:6-17: 32, 64, 100) */
    uint32_t max_requests;                  /* most header_range requests one launch set takes; 0 = 32 (hint-level kinds: at least 64) */
    uint32_t window_us;                     /* how long an open batch keeps collecting WHILE earlier batches occupy the GPU; 0 = 50 */
    uint32_t n_lanes;                       /* batches in flight per request kind; 0 = 3 */
    uint32_t chain_id_len;                  /* C::CHAIN_ID_BYTES (header_range.rs:42-43), at most 50 bytes */
    uint8_t chain_id[52];
    uint32_t flags;                         /* must be 0 */
    uint32_t key_rows;                      /* rows of each lane's fixed-key Ed25519 table (5.8 MB per row), keyed by PUBLIC KEY: requests signed by
                                               different validator sets share it (validator sets change along the chain: fetcher.rs:60-87);
                                               0 = 2 * v_max + 32; keys beyond its capacity in one launch set go to the generic kernel */
    uint32_t _reserved[2];
} bsx_batcher_config;                       /* sizeof == 96 */
int bsx_batcher_create(bsx_ctx* ctx, const bsx_batcher_config* cfg, bsx_batcher** out);
/* Finishes the batches in flight, fails requests that were only collected (BSX_ERR_BAD_ARG "destroyed before the request ran"), wakes
 * every waiter and lets threads inside bsx_wait / bsx_poll / a submit leave before anything is freed.  Tickets cannot be waited for
 * afterwards. */
void bsx_batcher_destroy(bsx_batcher* b);

/* bsx_header_range (CombinedSkipCircuit::define, header_range.rs:32-59) without a witness.  Argument meaning as bsx_header_range;
 * the circuit shape, v_max and chain id are the batcher's.  Errors detectable from the arguments alone are returned HERE (no
 * ticket is issued); everything else through bsx_wait. */
int bsx_submit_header_range(bsx_batcher* b, const uint8_t input48[48], const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                            uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                            uint8_t output64[64], bsx_commit_result* out_commit /* optional */, bsx_ticket* out_ticket);
/* Round 6 — the upload of a one-range-per-call host: 1.05 MB of 512-byte header records per call, 23 % of it padding,
 * copied once into the batcher's staging.
 *   BSX_SUBMIT_INPUTS_STAY     the input buffers stay valid and unchanged until bsx_wait has returned: headers that lie in page-locked
 *                              memory (hipHostMalloc / bsx_host_register) are uploaded from where they lie — no staging copy
 *   BSX_SUBMIT_PACKED_HEADERS  `headers` points at a PACKED block (below) and `n_headers` is its size in BYTES; the block crosses PCIe
 *                              packed (~408 B per header) and is laid out as bsx_header records in HBM.  first_height = height of the
 *                              block's first header
 * flags == 0 is bsx_submit_header_range. */
#define BSX_SUBMIT_INPUTS_STAY 1u
#define BSX_SUBMIT_PACKED_HEADERS 2u
int bsx_submit_header_range_ex(bsx_batcher* b, const uint8_t input48[48], const void* headers, uint64_t first_height, uint64_t n_headers,
                               uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                               uint8_t output64[64], bsx_commit_result* out_commit /* optional */, bsx_ticket* out_ticket, uint32_t flags);
/* Synchronous form with packed headers: submit + wait on a context whose batcher has this shape; any other context unpacks on the host
 * and calls bsx_header_range. */
int bsx_header_range_packed(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const void* packed, uint64_t packed_bytes,
                            uint64_t first_height, uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                            uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit /* optional */);
/* Packed wire headers: the 14 protobuf-encoded fields of each header (what get_signed_header_range fetches per block, input.rs:120-145;
 * ~394 B) back to back instead of padded to bsx_header's fixed capacities:
 *   u32 n_headers, u32 n_bytes (the whole block) | u32 off[n_headers + 1] (byte offsets into the data section; off[n] = its size) |
 *   padding to a multiple of 16 | data: per header u8 len[14], then field 0 .. 13, len[f] bytes each
 * bsx_pack_headers writes a block from records (out_cap >= bsx_packed_headers_bound(n) always suffices), bsx_unpack_headers is its inverse
 * (host code, no GPU; out == NULL with out_cap_headers == 0 only validates).  A field length over its capacity travels as it is and is
 * reported as BSX_ERR_BAD_HEADER by the request, exactly as for a 512-byte record; a block whose own offsets are inconsistent is
 * BSX_ERR_BAD_HEADER at submit. */
uint64_t bsx_packed_headers_bound(uint64_t n_headers);
int bsx_pack_headers(const bsx_header* headers, uint64_t n_headers, void* out, uint64_t out_cap, uint64_t* out_bytes);
int bsx_unpack_headers(const void* packed, uint64_t packed_bytes, bsx_header* out, uint64_t out_cap_headers, uint64_t* out_n);
/* hipHostRegister / hipHostUnregister for hosts that do not link HIP: page-lock a header buffer ONCE, reuse it for every call. */
int bsx_host_register(bsx_ctx* ctx, void* p, uint64_t bytes);
int bsx_host_unregister(bsx_ctx* ctx, void* p);
/* bsx_data_commitment_inputs — the body of `DataCommitmentOffchainInputs<MAX_LEAVES>::hint` (data_commitment.rs:18-45 ->
 * input.rs:149-271) with MAX_LEAVES = the batcher's batch_size. */
int bsx_submit_data_commitment_inputs(bsx_batcher* b, const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                                      uint64_t start_block, uint64_t end_block, uint8_t out_start_header[32], uint8_t out_end_header[32],
                                      bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb,
                                      uint8_t out_expected_data_commitment[32] /* optional */, bsx_ticket* out_ticket);
/* The whole map closure of prove_data_commitment (builder.rs:305-336) for ONE map job as one request: the hint for
 * [S + job B, S + (job + 1) B) (builder.rs:315-332 -> data_commitment.rs:22-44) and prove_subchain on what it returned (builder.rs:335),
 * with `range` = the map's shared ctx (DataCommitmentSharedCtx, builder.rs:12-18: global start / end block and header hashes).
 * headers[i] is the header at height first_height + i and must cover [batch_start, min(batch_end, latest_block - 2)].  The proofs are
 * the library's own, so their path digests come straight from the header trees (19 of prove_subchain's 21 compressions per slot are not
 * repeated; the record is bit-identical).  out_record as bsx_prove_subchain; the four hint outputs are optional. */
int bsx_submit_map_job(bsx_batcher* b, const bsx_shared_ctx* range, uint32_t job_index, const bsx_header* headers, uint64_t first_height,
                       uint64_t n_headers, uint64_t latest_block, uint8_t out_start_header[32], uint8_t out_end_header[32], bsx_data_hash_proof* out_dh,
                       bsx_last_block_id_proof* out_lb, bsx_subchain* out_record, bsx_ticket* out_ticket);
/* bsx_prove_subchain (builder.rs:45-52,150-271) with BATCH_SIZE = the batcher's batch_size, without a witness. */
int bsx_submit_prove_subchain(bsx_batcher* b, const uint8_t start_header[32], const uint8_t end_header[32], const bsx_data_hash_proof* dh,
                              const bsx_last_block_id_proof* lb, uint64_t batch_start_block, uint64_t batch_end_block, uint64_t global_end_block,
                              const uint8_t global_end_header_hash[32], bsx_subchain* out_record, bsx_ticket* out_ticket);
/* Blocks until the request has completed; returns ITS status.  A ticket may be waited for once or several times, from any thread,
 * until 16,384 later requests have been issued on the batcher (then: BSX_ERR_BAD_ARG, "expired"). */
int bsx_wait(bsx_batcher* b, bsx_ticket ticket);
/* Non-blocking form for hosts with their own executor (a Rust future polls it): *out_done = 1 once bsx_wait would not block. */
int bsx_poll(bsx_batcher* b, bsx_ticket ticket, int* out_done);
/* Synchronous form of bsx_submit_map_job on a context: submit + wait when the context has a batcher of this batch_size attached
 * (bsx_enable_coalescing), otherwise bsx_data_commitment_inputs followed by bsx_prove_subchain on the serial path.  Returns
 * BSX_ERR_ASSERT when an assertion of the map job fails (out_record->assert_fail says which). */
int bsx_map_job(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const bsx_shared_ctx* range, uint32_t job_index, const bsx_header* headers,
                uint64_t first_height, uint64_t n_headers, uint64_t latest_block, uint8_t out_start_header[32], uint8_t out_end_header[32],
                bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb, bsx_subchain* out_record);
/* Attach a batcher to the context: from then on the SYNCHRONOUS host-tier calls bsx_header_range (witness == NULL, same
 * nb_map_jobs / batch_size / v_max / chain id), bsx_data_commitment_inputs (max_leaves == batch_size) and bsx_prove_subchain
 * (batch_size equal, witness == NULL) made on this context by ANY number of threads are submit + wait on it — the reference's
 * callers need no change beyond this one call.  Calls of another shape, or with a witness, take the serial path as before.
 * cfg == NULL detaches (and destroys) it. */
int bsx_enable_coalescing(bsx_ctx* ctx, const bsx_batcher_config* cfg);
/* A caller that is about to submit a burst (the 32 hints of one proof from one thread) corks the batcher first: while corked, an
 * open batch closes only when it is full (or after 20 ms); uncorking (on = 0) lets the open batches go at once.  Like TCP_CORK: purely
 * a hint. */
int bsx_batcher_cork(bsx_batcher* b, int on);
typedef struct bsx_batcher_stats {
    /* per request kind (0 header_range, 1 data_commitment_inputs / map jobs, 2 prove_subchain): launch sets run, requests served, the
     * largest set, and where the worker's time went, summed over the sets: first claim -> closed, closed -> every slot staged (for
     * header_range this includes enqueuing the header uploads as slots arrive), staged -> last enqueue returned, -> the GPU is done,
     * -> every ticket completed */
    struct { uint64_t batches, requests, max_batch, close_wait_ns, stage_wait_ns, enqueue_ns, gpu_wait_ns, complete_ns; } kind[3];
} bsx_batcher_stats;                        /* sizeof == 192 */
int bsx_batcher_get_stats(bsx_batcher* b, bsx_batcher_stats* out);
/* the batcher bsx_enable_coalescing attached (NULL: none) — for bsx_batcher_get_stats / explicit submits beside the synchronous calls */
bsx_batcher* bsx_context_batcher(bsx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* BSX_H */
