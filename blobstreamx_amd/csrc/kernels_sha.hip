// kernels_sha.hip — SHA-256 side of the header_range hot path for gfx950 (CDNA4, wave64).
//
//   k_header_merkle    P5   tendermint Header::hash + inclusion proofs (circuits/input.rs:175-179,188-195,250-261)
//   k_assemble_inputs  hint DataCommitmentOffchainInputs::hint -> get_data_commitment_inputs (circuits/data_commitment.rs:22-44,
//                           circuits/input.rs:149-271): gathers proofs into each map job's compact witness, zero padding rules
//   prove_subchain (circuits/builder.rs:150-271 incl. get_data_commitment :105-148) as three barrier-free stages:
//     k_slot_hashes    P1-P3 one lane per header slot: both inclusion-proof paths, the data-root tuple and its leaf hash
//     k_tree_level     P2   one lane per tree node of one level of compute_root_from_leaves, across ALL map jobs
//     k_batch_finish   P4   chain-link predicates (wave ballots), batch tail, MapReduceSubchainVariable record
//   k_reduce           reduce closure of prove_data_commitment (circuits/builder.rs:337-395)
//   k_finalize         range check + final asserts + public output (builder.rs:292-297,400-406; header_range.rs:57-58)
//   k_expand_witness   P10  bytes -> Goldilocks elements ([UPSTREAM] plonky2x ByteVariable = 8 bools MSB first)
//
// Mapping: one lane = one independent hash chain (header / slot / tree node); digests stay in VGPRs between the
// hashes of a chain and go to HBM exactly once (they ARE the witness).  The SHA kernels are integer-ALU bound
// (measured ceiling on MI355X: ~28 G compressions/s, tools/microbench.hip), so they are built for occupancy: no LDS,
// no block barriers, every lane busy at every tree level.  Round-1 measurements (profiles/) showed the LDS-staged,
// barrier-synchronised variant parked 51 % of its wave cycles.  No MFMA: bitwise integer work.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"
#include "kernels.h"
#include "sha256.h"

namespace bsx {

// Kernels of the hashing chain (the critical path of a pipelined chunk) raise their wave priority above the commit
// check's kernels, which share the integer ALUs but have a whole pass of slack (engine.py); the HBM-bound expansion
// runs at priority 3.
#define BSX_CHAIN_PRIO() __builtin_amdgcn_s_setprio(1)

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ void store_u32_a2(uint8_t* dst, uint32_t v) {  // dst 2-byte aligned (odd batch sizes only)
    if (reinterpret_cast<uintptr_t>(dst) & 2) {
        reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;
        reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);
    } else {
        *reinterpret_cast<uint32_t*>(dst) = v;
    }
}
__device__ __forceinline__ uint32_t load_u32_a2(const uint8_t* src) {
    if (reinterpret_cast<uintptr_t>(src) & 2)
        return (uint32_t)reinterpret_cast<const uint16_t*>(src)[0] | ((uint32_t)reinterpret_cast<const uint16_t*>(src)[1] << 16);
    return *reinterpret_cast<const uint32_t*>(src);
}
// digest <-> 32 bytes in global memory; the byte sections of the compact witness are 4-byte aligned for even
// batch sizes and only 2-byte aligned for B == 1, hence the _a2 accessors (the branch is wave-uniform per job)
__device__ __forceinline__ void store_digest_global(uint8_t* dst, const Digest& d) {
    if (reinterpret_cast<uintptr_t>(dst) & 2) {
#pragma unroll
        for (int k = 0; k < 8; k++) store_u32_a2(dst + 4 * k, bswap32(d.w[k]));
    } else {
        uint32_t* p = reinterpret_cast<uint32_t*>(dst);
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = bswap32(d.w[k]);
    }
}
__device__ __forceinline__ void store_digest_global16(uint8_t* dst, const Digest& d) {  // dst 16-byte aligned
    uint4* p = reinterpret_cast<uint4*>(dst);
    p[0] = make_uint4(bswap32(d.w[0]), bswap32(d.w[1]), bswap32(d.w[2]), bswap32(d.w[3]));
    p[1] = make_uint4(bswap32(d.w[4]), bswap32(d.w[5]), bswap32(d.w[6]), bswap32(d.w[7]));
}
__device__ __forceinline__ Digest load_digest_global(const uint8_t* src) {
    Digest d;
    if (reinterpret_cast<uintptr_t>(src) & 2) {
#pragma unroll
        for (int k = 0; k < 8; k++) d.w[k] = bswap32(load_u32_a2(src + 4 * k));
    } else {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src);
#pragma unroll
        for (int k = 0; k < 8; k++) d.w[k] = bswap32(p[k]);
    }
    return d;
}

// 16-byte global accesses at ANY byte alignment: gfx950 under ROCm serves misaligned dword / dwordx4 global accesses
// (checked on the device by tools/unaligned_test.hip); the packed witness sections are only 2-byte aligned.
__device__ __forceinline__ uint4 ldu4(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stu4(uint8_t* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void store_digest_u(uint8_t* dst, const Digest& d) {
    stu4(dst, make_uint4(bswap32(d.w[0]), bswap32(d.w[1]), bswap32(d.w[2]), bswap32(d.w[3])));
    stu4(dst + 16, make_uint4(bswap32(d.w[4]), bswap32(d.w[5]), bswap32(d.w[6]), bswap32(d.w[7])));
}

// ND dwords of a field that starts at a 4-byte aligned global address
template <int ND>
__device__ __forceinline__ Digest leaf_from_global(const uint32_t* f, int len) {
    uint32_t d[14];
#pragma unroll
    for (int j = 0; j < 14; j++) d[j] = (j < ND) ? f[j] : 0u;
    return leaf_hash_1block(d, len);
}

// ------------------------------------------------------------------------------------------------ k_header_merkle
// The 14-leaf tree of a header (15 leaf blocks + 13 x 2 inner blocks = 41 compressions) is cut at its natural joints —
// tendermint splits 14 = 8 | 6, 8 = 4 | 4, 6 = 4 | 2 — into four SUB-TREES hashed by four different waves of a
// workgroup, one lane per header each:
//     role 0  leaves 0..3   -> n0123   10 compressions        role 2  leaves 8..11  -> n8_11  10 compressions
//     role 1  leaves 4..7   -> n4567   11 (last_block_id = 2)  role 3  leaves 12,13  -> n12_13  4, then after the barrier the
//                                                                      three joining nodes left, right, root: + 6 = 10
// Why not one lane per whole header (round 1): 41 compressions per lane is a coarse unit.  It needed 199 VGPRs (80 record
// dwords resident) = 2 waves per SIMD, and 2049 x 128 headers = 4104 waves on 2048 slots ran as two full rounds plus a
// round of 8 straggler waves: 0.57 ms where the compressions take 0.39 ms at the ALU ceiling.  Quarter units need a quarter
// of the record in registers (4 waves per SIMD) and balance 4x finer; the roles rotate with the group index so that every
// SIMD of a CU sees the same mix.  Digests cross waves through 9 KB of LDS (stride 9 dwords: conflict free).
// A lane reads each field of its own 512-byte record right before hashing it and stores a digest as soon as it exists:
// values that live across a call of the (not inlined) compression function sit in callee-saved registers, which the
// AMDGPU calling convention interleaves with caller-saved ones — a record quarter kept resident costs twice its size in
// register index.  The price is a memory round trip per leaf (every call drains the wave's outstanding operations);
// with four waves per SIMD the other three issue meanwhile.
constexpr int HM_THREADS = 256;           // 4 waves = the 4 sub-trees of 64 headers
constexpr int HM_GROUP = 64;              // headers per workgroup pass
constexpr int HM_LDS_STRIDE = 9;

// leaf of a field held in registers: W = dwords [BASE, BASE + N) of the record, field at dword OFF, ND dwords
template <int ND, int OFF, int BASE, int N>
__device__ __forceinline__ Digest leaf_from_regs(const uint32_t (&W)[N], int len) {
    uint32_t d[14];
#pragma unroll
    for (int j = 0; j < 14; j++) d[j] = (j < ND) ? W[OFF - BASE + j] : 0u;
    return leaf_hash_1block(d, len);
}
// dwords [BASE, BASE + N) of a record, N a multiple of 4
template <int BASE, int N>
__device__ __forceinline__ void hm_load(const uint8_t* rec, uint32_t (&W)[N]) {
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
        const uint4 v = ldu4(rec + 4 * BASE + 16 * k);
        W[4 * k] = v.x; W[4 * k + 1] = v.y; W[4 * k + 2] = v.z; W[4 * k + 3] = v.w;
    }
}
// field length i (byte i of the record), clamped to the field's capacity (bsx.h); a violation sets `bad`
__device__ __forceinline__ int hm_len(uint32_t lens, int i, int cap, bool& bad) {
    int l = (int)((lens >> (8 * (i & 3))) & 0xff);
    if (l > cap) { bad = true; l = cap; }
    return l;
}

// paths (optional): per header the 7 distinct digests of the two inclusion-proof PATHS prove_subchain materialises for it
// (builder.rs:189-199) — [L6, n67, L4, n45, n4567, left, root]: data_hash path = L6, n67, n4567, left, root; last_block_id path
// = L4, n45, n4567, left, root.  They are nodes of the tree hashed here anyway; handing them to the hint (k_assemble_inputs)
// lets prove_subchain skip re-deriving them from the proofs (19 of its 21 compressions per slot).
constexpr uint32_t HM_PATH_BYTES = 7 * 32;
__global__ __launch_bounds__(HM_THREADS, 4) void k_header_merkle(const bsx_header* __restrict__ hdr, uint64_t n,
                                                              uint8_t* __restrict__ hashes, uint8_t* __restrict__ dh_aunts,
                                                              uint8_t* __restrict__ lb_aunts, uint8_t* __restrict__ paths,
                                                              uint32_t* __restrict__ status, uint32_t low_prio, bsxk_merkle_tap tap,
                                                              uint64_t status_group) {
    // beside an expansion: above the commit check's waves (BSX_CHAIN_PRIO).  In the compact pipeline this kernel is the bulk
    // ALU work that the OTHER buffer set's short chain kernels (hint, prove_subchain, reduce, ...) must get through: it yields
    if (!low_prio) BSX_CHAIN_PRIO();
    // double-buffered by pass parity: the joining wave of pass p reads buffer p & 1 while the other three waves already
    // write pass p + 1's sub-tree roots into the other one; pass p + 2 reuses buffer p & 1 only behind the barrier of pass
    // p + 1, which the joining wave of pass p reaches after its reads
    __shared__ uint32_t sub2[2][3][HM_GROUP * HM_LDS_STRIDE];
    uint32_t pass = 0;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_groups = (n + HM_GROUP - 1) / HM_GROUP;
    // grid-strided: the launcher may cap the grid (BSX_MERKLE_WGS)
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x, pass ^= 1) {
        uint32_t (*sub)[HM_GROUP * HM_LDS_STRIDE] = sub2[pass];
        const uint32_t role = (wave + (uint32_t)grp) & 3;            // wave-uniform
        const uint64_t me = grp * HM_GROUP + lane;
        const bool live = me < n;
        const uint8_t* my = reinterpret_cast<const uint8_t*>(hdr + (live ? me : 0));
        // byte offsets: lengths 0..13, version 16, chain_id 40, height 92, time 104, last_block_id 124, hash[j] 200+36j, proposer 488
        bool bad = false;
        Digest top;                                                  // this role's sub-tree root
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(my);
        const uint32_t lens = rec[role];                             // lengths 4 role .. 4 role + 3
        if (role == 0) {
            const Digest L0 = leaf_from_global<6>(rec + 4, hm_len(lens, 0, 24, bad));
            const Digest L1 = leaf_from_global<13>(rec + 10, hm_len(lens, 1, 52, bad));
            const Digest n01 = inner_hash(L0, L1);
            const Digest L2 = leaf_from_global<3>(rec + 23, hm_len(lens, 2, 12, bad));
            const Digest L3 = leaf_from_global<5>(rec + 26, hm_len(lens, 3, 20, bad));
            top = inner_hash(n01, inner_hash(L2, L3));               // n0123
            if (live) {
                if (lb_aunts) store_digest_u(lb_aunts + me * 128 + 64, top);
                if (dh_aunts) store_digest_u(dh_aunts + me * 128 + 64, top);
            }
        } else if (role == 1) {
            uint8_t* p = paths ? paths + me * HM_PATH_BYTES : nullptr;
            Digest L4;
            {
                uint32_t d[19];
#pragma unroll
                for (int j = 0; j < 19; j++) d[j] = rec[31 + j];
                const int l4 = hm_len(lens, 4, 76, bad);
                L4 = (l4 <= 54) ? leaf_hash_1block(d, l4) : leaf_hash_2block(d, l4);
            }
            if (live && p) store_digest_u(p + 64, L4);
            const Digest L5 = leaf_from_global<9>(rec + 50, hm_len(lens, 5, 36, bad));
            if (live && lb_aunts) store_digest_u(lb_aunts + me * 128, L5);                 // index 4: [L5, n67, n0123, right]
            const Digest n45 = inner_hash(L4, L5);
            if (live && p) store_digest_u(p + 96, n45);
            if (live && dh_aunts) store_digest_u(dh_aunts + me * 128 + 32, n45);           // index 6: [L7, n45, n0123, right]
            const Digest L6 = leaf_from_global<9>(rec + 59, hm_len(lens, 6, 36, bad));    // data_hash
            if (live && p) store_digest_u(p, L6);
            const Digest L7 = leaf_from_global<9>(rec + 68, hm_len(lens, 7, 36, bad));
            if (live && dh_aunts) store_digest_u(dh_aunts + me * 128, L7);
            const Digest n67 = inner_hash(L6, L7);
            if (live && p) store_digest_u(p + 32, n67);
            if (live && lb_aunts) store_digest_u(lb_aunts + me * 128 + 32, n67);
            top = inner_hash(n45, n67);                              // n4567
            if (live && p) store_digest_u(p + 128, top);
        } else if (role == 2) {
            const Digest L8 = leaf_from_global<9>(rec + 77, hm_len(lens, 8, 36, bad));
            const Digest L9 = leaf_from_global<9>(rec + 86, hm_len(lens, 9, 36, bad));
            const Digest n89 = inner_hash(L8, L9);
            const Digest L10 = leaf_from_global<9>(rec + 95, hm_len(lens, 10, 36, bad));
            const Digest L11 = leaf_from_global<9>(rec + 104, hm_len(lens, 11, 36, bad));
            top = inner_hash(n89, inner_hash(L10, L11));             // n8_11
        } else {
            const Digest L12 = leaf_from_global<9>(rec + 113, hm_len(lens, 12, 36, bad));
            const Digest L13 = leaf_from_global<6>(rec + 122, hm_len(lens, 13, 24, bad));
            top = inner_hash(L12, L13);                              // n12_13
        }
        if (role != 3) {
#pragma unroll
            for (int k = 0; k < 8; k++) sub[role][lane * HM_LDS_STRIDE + k] = top.w[k];
        }
        __syncthreads();
        if (role == 3) {
            Digest a, b;
#pragma unroll
            for (int k = 0; k < 8; k++) { a.w[k] = sub[0][lane * HM_LDS_STRIDE + k]; b.w[k] = sub[1][lane * HM_LDS_STRIDE + k]; }
            const Digest left = inner_hash(a, b);
#pragma unroll
            for (int k = 0; k < 8; k++) a.w[k] = sub[2][lane * HM_LDS_STRIDE + k];
            const Digest right = inner_hash(a, top);
            const Digest root = inner_hash(left, right);
            if (live) {
                if (hashes) store_digest_u(hashes + me * 32, root);
                if (paths) {
                    store_digest_u(paths + me * HM_PATH_BYTES + 160, left);
                    store_digest_u(paths + me * HM_PATH_BYTES + 192, root);
                }
                if (lb_aunts) store_digest_u(lb_aunts + me * 128 + 96, right);
                if (dh_aunts) store_digest_u(dh_aunts + me * 128 + 96, right);
                if (me == tap.idx) {
                    if (tap.dst_a) store_digest_u(tap.dst_a, root);
                    if (tap.dst_b) store_digest_u(tap.dst_b, root);
                }
                if (tap.ranges) {                                    // the target header of its request (coalescing front end)
                    const uint64_t r = me / tap.hpr, k = me - r * tap.hpr;
                    if (tap.ranges[r].end_block - tap.ranges[r].start_block == k) {
                        store_digest_u(tap.ranges[r].end_header_hash, root);
                        if (tap.dense) store_digest_u(tap.dense + 32 * r, root);
                    }
                }
            }
        }
        // wave-ballot reduction of the "bad header" predicate: one atomic per wave.  status_group != 0 (the coalescing front end:
        // one status word per request, `status_group` headers each): a malformed header marks its own request only
        const unsigned long long m = __ballot(live && bad);
        if (m && status) {
            if (status_group) { if (live && bad) atomicOr(status + me / status_group, 1u); }
            else if (lane == 0) atomicOr(status, 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_assemble_inputs
// One workgroup per (range, owned job).  Byte-exact gather of the hint output into the compact witness.
struct AssembleArgs {
    uint32_t n_ranges, nb_map_jobs, batch, job_first, job_count, span;   // span = batch_end - batch_start (== batch for map jobs)
    const bsx_shared_ctx* ranges;
    const uint64_t* latest;
    const bsx_header* headers;
    uint64_t headers_per_range, header_first_rel;   // headers[r*hpr + k] is the header at height S_r + header_first_rel + k
    const uint8_t *hashes, *dh_aunts, *lb_aunts;
    uint8_t* compact;
    uint32_t compact_stride, off_words;
    uint32_t* status;
    const uint8_t* paths;        // optional (k_header_merkle): the slots' path digests are gathered too
    const uint8_t* zero_paths;   // path digests of the all-zero proofs (padding slots): dh[5] then lb[5]
    const uint32_t* spans;       // optional, per range: overrides `span` (coalesced hint requests of different lengths in one launch)
    const uint32_t* jobs;        // optional, per range (job_count must be 1): the map job this request is — overrides job_first, and the
                                 // range's header block then starts at the JOB's first height (header_first_rel = job * batch)
    uint32_t status_per_range;   // 1: status word r belongs to range r (coalescing front end); 0: one shared word
};

// AS_IT = 16-byte pieces per lane = ceil(24 * B / 256): a template parameter so that the in-flight buffer (and with it
// the register footprint: 96 VGPRs for the general case, 24 at B = 64) matches the batch size — more resident waves
// = more loads in flight when the kernel runs beside the expansion.
template <uint32_t AS_IT>
__global__ __launch_bounds__(256) void k_assemble_inputs(AssembleArgs a) {
    BSX_CHAIN_PRIO();
    const uint32_t r = blockIdx.x / a.job_count, jl = blockIdx.x % a.job_count, j = a.jobs ? a.jobs[r] : a.job_first + jl;
    const uint32_t B = a.batch;
    const uint64_t header_first_rel = a.jobs ? (uint64_t)j * B : a.header_first_rel;
    const bsx_shared_ctx rg = a.ranges[r];
    const uint64_t S = rg.start_block;
    const uint64_t batch_start = S + (uint64_t)j * B;          // builder.rs:315-316
    const uint64_t batch_end = batch_start + (a.spans ? a.spans[r] : a.span);   // builder.rs:317-322 (span == B for map jobs)
    const uint64_t latest = a.latest[r];
    const uint64_t latest_safe = latest - 2;                   // input.rs:160-161
    const uint64_t req_end = batch_end < latest_safe ? batch_end : latest_safe;  // input.rs:162
    // number of real proofs: dh for [start, req_end), lb for (start, req_end]  (input.rs:167-198)
    const uint64_t n_real = (batch_start <= req_end) ? (req_end - batch_start) : 0;
    const bool have_hdrs = batch_start < req_end;              // input.rs:249
    const uint64_t hbase = (uint64_t)r * a.headers_per_range - header_first_rel;    // virtual index of height S
    bool oob = false;
    if (batch_start <= req_end && ((batch_start - S) < header_first_rel || (req_end - S - header_first_rel) >= a.headers_per_range)) oob = true;
    uint8_t* cw = a.compact + ((uint64_t)r * a.job_count + jl) * a.compact_stride;
    uint32_t* cw32 = reinterpret_cast<uint32_t*>(cw);

    // All loads of the workgroup are issued before its first store (one memory round trip, see below).
    // bytes [0,128): ctx hashes, start/end header (value loaded here, stored with the rest)
    uint32_t v_ctx = 0;
    if (threadIdx.x < 32) {
        const uint32_t t = threadIdx.x, k = t & 7;
        if (t < 8) v_ctx = reinterpret_cast<const uint32_t*>(a.ranges[r].start_header_hash)[k];
        else if (t < 16) v_ctx = reinterpret_cast<const uint32_t*>(a.ranges[r].end_header_hash)[k];
        else if (t < 24) v_ctx = (have_hdrs && !oob) ? reinterpret_cast<const uint32_t*>(a.hashes + (hbase + (batch_start - S)) * 32)[k] : 0u;
        else v_ctx = (have_hdrs && !oob) ? reinterpret_cast<const uint32_t*>(a.hashes + (hbase + (req_end - S)) * 32)[k] : 0u;
    }
    // proofs: the packed stream [128, 128 + 362*B) as 24 pieces per slot, 16 bytes each except two tails:
    //   0..7   data_hash aunts      8..9  data_hash leaf[0..32)     10  leaf[32..34)  (2 bytes)
    //   11..18 last_block_id aunts  19..22 last_block_id leaf[0..64) 23  leaf[64..72)  (8 bytes)
    // Proof records are only 2-byte aligned in the packed image; gfx950 serves misaligned global dword / dwordx4
    // accesses (tools/unaligned_test.hip), so every piece moves as ONE 16-byte load and one store.  All of a lane's
    // loads are issued before its first store: beside the HBM-bound expansion of the other chunk a load round trip is
    // several times longer, and this gather is pure latency (the 16-bit-unit version took 1.3-1.8 ms there, 0.12 alone).
    const uint64_t h0 = hbase + (batch_start - S);
    bool bad_leaf = false;
    constexpr uint32_t AS_PIECES = 24, AS_MAX_IT = AS_IT;
    const uint32_t n_items = AS_PIECES * B;
    uint8_t* dh_dst = cw + 128;
    uint8_t* lb_dst = cw + 128 + (uint32_t)BSX_DH_PROOF_SIZE * B;
    const uint8_t* dummy = reinterpret_cast<const uint8_t*>(a.headers);
    uint4 val[AS_MAX_IT];
#pragma unroll
    for (uint32_t it = 0; it < AS_MAX_IT; it++) {
        if (it * 256u >= n_items) break;                                // block-uniform
        static_assert(offsetof(bsx_header, hash) + 36 == 236 && offsetof(bsx_header, last_block_id) == 124, "bsx_header offsets");
        const uint32_t t = threadIdx.x + it * 256u;
        const uint32_t slot = t / AS_PIECES, pc = t % AS_PIECES;
        const bool real = (t < n_items) && !oob && (slot < n_real);
        const bool is_dh = pc < 11;
        const uint32_t q = is_dh ? pc : pc - 11;                        // 0..7 aunts, 8.. leaf pieces
        const uint64_t hidx = h0 + slot + (is_dh ? 0 : 1);
        const uint8_t* hdr = reinterpret_cast<const uint8_t*>(a.headers + hidx);
        const uint8_t* aunt = (is_dh ? a.dh_aunts : a.lb_aunts) + hidx * 128 + 16 * q;
        const uint8_t* leaf = hdr + (is_dh ? 236 : 124) + 16 * (q - 8);   // header.hash[1] / header.last_block_id (bsx.h)
        const uint8_t* p = real ? (q < 8 ? aunt : leaf) : dummy;
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        val[it] = real ? v : make_uint4(0, 0, 0, 0);
    }
    // leaf-length rules of the hint (input.rs:173,190): one lane per real proof
    for (uint32_t s = threadIdx.x; s < 2 * B; s += blockDim.x) {
        const uint32_t slot = s % B;
        const bool is_dh = s < B;
        if (!oob && slot < n_real) {
            const bsx_header* h = a.headers + h0 + slot + (is_dh ? 0 : 1);
            if (is_dh ? (h->len[BSX_DATA_HASH_INDEX] != BSX_PROTOBUF_HASH_SIZE) : (h->len[BSX_LAST_BLOCK_ID_INDEX] != BSX_PROTOBUF_BLOCK_ID_SIZE))
                bad_leaf = true;
        }
    }
    if (threadIdx.x < 32) cw32[threadIdx.x] = v_ctx;
#pragma unroll
    for (uint32_t it = 0; it < AS_MAX_IT; it++) {
        const uint32_t t = threadIdx.x + it * 256u;
        if (t >= n_items) break;
        const uint32_t slot = t / AS_PIECES, pc = t % AS_PIECES;
        const bool is_dh = pc < 11;
        const uint32_t q = is_dh ? pc : pc - 11;
        uint8_t* d = (is_dh ? dh_dst + (uint32_t)BSX_DH_PROOF_SIZE * slot : lb_dst + (uint32_t)BSX_LB_PROOF_SIZE * slot) + 16 * q;
        if (pc == 10) *reinterpret_cast<uint16_t*>(d) = (uint16_t)val[it].x;
        else if (pc == 23) *reinterpret_cast<uint2*>(d) = make_uint2(val[it].x, val[it].y);
        else *reinterpret_cast<uint4*>(d) = val[it];
    }
    // The slots' path digests (builder.rs:189-199): dh_path[5] = nodes of header h's tree, lb_path[5] = nodes of header h+1's —
    // 20 pieces of 16 bytes per slot, gathered like the proofs (second burst); padding slots take the zero-proof constants.
    if (a.paths) {
        constexpr uint32_t P_PIECES = 20, P_IT = (P_PIECES * BSX_MAX_BATCH + 255) / 256;
        const uint32_t n_p = P_PIECES * B;
        uint8_t* slots = cw + bsx_off_slots(B);
        for (uint32_t it0 = 0; it0 < P_IT; it0 += 4) {
            uint4 pv[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t t = threadIdx.x + (it0 + u) * 256u;
                const uint32_t slot = t / P_PIECES, pc = t % P_PIECES;
                const bool is_dh = pc < 10;
                const uint32_t j = (is_dh ? pc : pc - 10) >> 1, half = pc & 1;
                const uint32_t node = j >= 2 ? j + 2 : (is_dh ? j : j + 2);        // [L6, n67, L4, n45, n4567, left, root]
                const bool real = (t < n_p) && !oob && (slot < n_real);
                const uint8_t* src = real ? a.paths + (h0 + slot + (is_dh ? 0 : 1)) * HM_PATH_BYTES + 32 * node + 16 * half
                                          : a.zero_paths + (is_dh ? 0 : 160) + 32 * j + 16 * half;
                pv[u] = (t < n_p) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t t = threadIdx.x + (it0 + u) * 256u;
                if (t >= n_p) break;
                const uint32_t slot = t / P_PIECES, pc = t % P_PIECES;
                stu4(slots + BSX_SLOT_BYTES * slot + 16 * pc, pv[u]);              // dh_path at +0, lb_path at +160
            }
            if ((it0 + 4) * 256u >= n_p) break;
        }
    }
    if (threadIdx.x == 0) {
        uint32_t* W = reinterpret_cast<uint32_t*>(cw + a.off_words);
        W[BSX_W_CTX_START] = (uint32_t)rg.start_block; W[BSX_W_CTX_START + 1] = (uint32_t)(rg.start_block >> 32);
        W[BSX_W_CTX_END] = (uint32_t)rg.end_block; W[BSX_W_CTX_END + 1] = (uint32_t)(rg.end_block >> 32);
        W[BSX_W_BATCH_START] = (uint32_t)batch_start; W[BSX_W_BATCH_START + 1] = (uint32_t)(batch_start >> 32);
        W[BSX_W_BATCH_END] = (uint32_t)batch_end; W[BSX_W_BATCH_END + 1] = (uint32_t)(batch_end >> 32);
        if (a.status) {
            if (oob || latest < 2) atomicOr(a.status + (a.status_per_range ? r : 0u), 4u);
        }
    }
    if (__ballot(bad_leaf) && (threadIdx.x & 63) == 0 && a.status) atomicOr(a.status + (a.status_per_range ? r : 0u), 2u);
}

// ------------------------------------------------------------------------------------------------ prove_subchain
struct SubchainArgs {
    uint32_t n_jobs, batch;                  // n_jobs = n_ranges * job_count (compact witnesses are consecutive)
    uint32_t job_count;                      // map jobs per range in this call (range of job q = q / job_count)
    const bsx_shared_ctx* ranges;
    uint8_t* compact;
    uint32_t compact_stride, off_words, off_bools;
    bsx_subchain* records;
    uint32_t level, width, level_off;        // k_tree_level: nodes per job at this level, offset of the level in inner[]/node[]
    uint32_t top_inputs;                     // k_batch_finish: 0 = every remaining level + the batch tail; else it stops below the level
                                             // of width top_inputs / 2 and k_batch_top finishes (level / width / level_off = its first)
    uint32_t exp_skip;                       // experiments build only (BSX_BF_SKIP): bit 0 = k_batch_finish drops its per-slot scattered stores,
                                             // bit 1 = and takes its strided predicate inputs from one broadcast line — WRONG results, timing only:
                                             // the upper bound of what ANY re-layout of the compact witness could win
};
#ifdef BSX_EXPERIMENTS
#define BF_SKIP(a, bit) (((a).exp_skip >> (bit)) & 1u)
#else
#define BF_SKIP(a, bit) 0u
#endif

// dword k of a byte region starting at global byte address p (2-byte aligned): funnel of two aligned dwords
__device__ __forceinline__ uint32_t gdword_at(const uint8_t* p, int k) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3) + k;
    const uint32_t sh = (uint32_t)(a & 3) * 8;
    return sh ? funnel_r(q[1], q[0], sh) : q[0];
}

// ---- stage 1: one lane per slot, no communication (builder.rs:180-199, 134-137 + leaf hashes of :144-147)
constexpr int SH_THREADS = 256;
// PATHS = false: the path digests are already in the slot section (written by the hint from the header trees,
// k_assemble_inputs with `paths`): only the data-root tuple and its leaf hash remain (2 compressions instead of 21)
template <bool PATHS>
__global__ __launch_bounds__(SH_THREADS) void k_slot_hashes(SubchainArgs a) {
    BSX_CHAIN_PRIO();
    const uint32_t B = a.batch;
    const uint64_t gs = (uint64_t)blockIdx.x * SH_THREADS + threadIdx.x;     // global slot index
    if (gs >= (uint64_t)a.n_jobs * B) return;
    const uint32_t q = (uint32_t)(gs / B), i = (uint32_t)(gs % B);
    uint8_t* cw = a.compact + (uint64_t)q * a.compact_stride;
    const uint32_t* W = reinterpret_cast<const uint32_t*>(cw + a.off_words);
    const uint64_t batch_start = (uint64_t)W[BSX_W_BATCH_START] | ((uint64_t)W[BSX_W_BATCH_START + 1] << 32);
    uint8_t* sl = cw + bsx_off_slots(B) + BSX_SLOT_BYTES * i;

    // Every call of the (not inlined) compression function starts by draining the wave's outstanding memory operations
    // (the callee's prologue s_waitcnt), so a load or a store placed between two compressions costs a full memory round
    // trip — several times longer while the other chunk's expansion saturates HBM.  Hence two phases, each ONE burst
    // of 16-byte loads, then all its compressions on registers, then ONE burst of stores (was ~22 round trips per lane).
    uint32_t data_hash_le[8];   // data_hash_proofs[i].leaf[2..34] as LE dwords (builder.rs:250)
    {
        const uint8_t* pr = cw + bsx_off_dh_proofs(B) + BSX_DH_PROOF_SIZE * i;
        uint4 A[8];
        if (PATHS) {
#pragma unroll
            for (int k = 0; k < 8; k++) A[k] = ldu4(pr + 16 * k);
        }
        const uint4 l0 = ldu4(pr + 128), l1 = ldu4(pr + 144);
        const uint32_t l2 = (uint32_t)reinterpret_cast<const uint16_t*>(pr + 160)[0];
        const uint32_t lf[9] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2};
#pragma unroll
        for (int k = 0; k < 8; k++) data_hash_le[k] = funnel_r(lf[k + 1], lf[k], 16);
        if (PATHS) {
        Digest d[5];
        d[0] = leaf_hash_34(lf);
        // path [0,1,1,0] (builder.rs:166-167): h = bit ? inner(aunt, h) : inner(h, aunt)
#pragma unroll
        for (int lvl = 0; lvl < 4; lvl++) {
            const uint32_t al[8] = {A[2 * lvl].x, A[2 * lvl].y, A[2 * lvl].z, A[2 * lvl].w,
                                    A[2 * lvl + 1].x, A[2 * lvl + 1].y, A[2 * lvl + 1].z, A[2 * lvl + 1].w};
            const Digest aunt = digest_from_le(al);
            d[lvl + 1] = (lvl == 1 || lvl == 2) ? inner_hash(aunt, d[lvl]) : inner_hash(d[lvl], aunt);
        }
#pragma unroll
        for (int j = 0; j < 5; j++) store_digest_u(sl + 32 * j, d[j]);
        }
    }
    if (PATHS) {
        const uint8_t* pr = cw + bsx_off_lb_proofs(B) + BSX_LB_PROOF_SIZE * i;
        uint4 A[8], l[4];
#pragma unroll
        for (int k = 0; k < 8; k++) A[k] = ldu4(pr + 16 * k);
#pragma unroll
        for (int k = 0; k < 4; k++) l[k] = ldu4(pr + 128 + 16 * k);
        const uint2 l4 = *reinterpret_cast<const uint2*>(pr + 192);
        const uint32_t lf[18] = {l[0].x, l[0].y, l[0].z, l[0].w, l[1].x, l[1].y, l[1].z, l[1].w, l[2].x, l[2].y, l[2].z, l[2].w,
                                 l[3].x, l[3].y, l[3].z, l[3].w, l4.x, l4.y};
        Digest d[5];
        d[0] = leaf_hash_72(lf);
        // path [0,0,1,0] (builder.rs:168-169)
#pragma unroll
        for (int lvl = 0; lvl < 4; lvl++) {
            const uint32_t al[8] = {A[2 * lvl].x, A[2 * lvl].y, A[2 * lvl].z, A[2 * lvl].w,
                                    A[2 * lvl + 1].x, A[2 * lvl + 1].y, A[2 * lvl + 1].z, A[2 * lvl + 1].w};
            const Digest aunt = digest_from_le(al);
            d[lvl + 1] = (lvl == 2) ? inner_hash(aunt, d[lvl]) : inner_hash(d[lvl], aunt);
        }
#pragma unroll
        for (int j = 0; j < 5; j++) store_digest_u(sl + 160 + 32 * j, d[j]);
    }
    {
        // data-root tuple (builder.rs:82-103,134-137) and its leaf hash
        const uint64_t curr_idx = batch_start + i;
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = 0;
        t[6] = (uint32_t)(curr_idx >> 32);
        t[7] = (uint32_t)curr_idx;
#pragma unroll
        for (int k = 0; k < 8; k++) t[8 + k] = bswap32(data_hash_le[k]);
        const Digest tleaf = leaf_hash_tuple(t);
        uint8_t* tp = cw + bsx_off_tuples(B) + 64 * i;
        if (!BF_SKIP(a, 0)) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                stu4(tp + 16 * k, make_uint4(bswap32(t[4 * k]), bswap32(t[4 * k + 1]), bswap32(t[4 * k + 2]), bswap32(t[4 * k + 3])));
            store_digest_u(cw + bsx_off_leaf_hashes(B) + 32 * i, tleaf);
        }
    }
}

// path digests of the ALL-ZERO proofs (the hint's padding slots, input.rs:220-239): dh[5] then lb[5], 320 bytes; computed
// once per context (bsx_init) with the same operations k_slot_hashes<true> applies to a zero proof
__global__ void k_zero_paths(uint8_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t z[18] = {0};
    const Digest zero = digest_from_le(z);
    Digest d[5];
    d[0] = leaf_hash_34(z);
    for (int lvl = 0; lvl < 4; lvl++) d[lvl + 1] = (lvl == 1 || lvl == 2) ? inner_hash(zero, d[lvl]) : inner_hash(d[lvl], zero);
    for (int j = 0; j < 5; j++) store_digest_u(out + 32 * j, d[j]);
    d[0] = leaf_hash_72(z);
    for (int lvl = 0; lvl < 4; lvl++) d[lvl + 1] = (lvl == 2) ? inner_hash(zero, d[lvl]) : inner_hash(d[lvl], zero);
    for (int j = 0; j < 5; j++) store_digest_u(out + 160 + 32 * j, d[j]);
}

// batch bounds of a job (builder.rs:235-243) from its compact witness and the global end block
__device__ __forceinline__ void batch_bounds(const uint32_t* W, uint64_t E, uint64_t& bs, uint64_t& be, uint64_t& temp_end,
                                             uint64_t& end_block_num) {
    bs = (uint64_t)W[BSX_W_BATCH_START] | ((uint64_t)W[BSX_W_BATCH_START + 1] << 32);
    be = (uint64_t)W[BSX_W_BATCH_END] | ((uint64_t)W[BSX_W_BATCH_END + 1] << 32);
    temp_end = (be < E) ? be : E;                       // :235-240
    end_block_num = (temp_end < bs) ? bs : temp_end;    // :241-243
}

// ---- stage 2: one level of compute_root_from_leaves [UPSTREAM plonky2x; SURVEY App. B] for all jobs at once:
// inner = inner_hash(l, r) always; node = both children enabled ? inner : l; enabled = l || r (prefix mask).
constexpr int TR_THREADS = 256;
// node t of tree level `level` (width nodes, stored at level_off) of the job whose compact witness is cw, from its two
// children l, r; nb = nb_enabled_leaves (builder.rs:119,124, low limb).  Returns the selected node.
__device__ __forceinline__ Digest tree_node(uint8_t* cw, uint32_t off_bools, uint32_t B, uint32_t level, uint32_t level_off, uint32_t t,
                                            uint32_t nb, const Digest& l, const Digest& r) {
    const uint32_t span = 1u << level;                   // leaves under a node of this level
    const Digest in = inner_hash(l, r);
    const bool en_l = t * span < nb, en_r = t * span + span / 2 < nb;
    const Digest node = (en_l && en_r) ? in : l;
    store_digest_global(cw + bsx_off_inner(B) + 32 * (level_off + t), in);
    store_digest_global(cw + bsx_off_nodes(B) + 32 * (level_off + t), node);
    cw[off_bools + bsx_b_node_enabled(B) + level_off + t] = en_l || en_r;
    return node;
}
__global__ __launch_bounds__(TR_THREADS) void k_tree_level(SubchainArgs a) {
    BSX_CHAIN_PRIO();
    const uint32_t B = a.batch, width = a.width;
    const uint64_t g = (uint64_t)blockIdx.x * TR_THREADS + threadIdx.x;
    if (g >= (uint64_t)a.n_jobs * width) return;
    const uint32_t q = (uint32_t)(g / width), t = (uint32_t)(g % width);
    uint8_t* cw = a.compact + (uint64_t)q * a.compact_stride;
    const uint32_t* W = reinterpret_cast<const uint32_t*>(cw + a.off_words);
    const uint64_t E = a.ranges[q / a.job_count].end_block;
    uint64_t bs, be, te, ebn;
    batch_bounds(W, E, bs, be, te, ebn);
    const uint32_t nb = (uint32_t)(ebn - bs);
    // children: level 1 reads the leaf hashes, upper levels the previous level's selected nodes
    const uint8_t* ch = (a.level == 1) ? cw + bsx_off_leaf_hashes(B) + 64 * t
                                       : cw + bsx_off_nodes(B) + 32 * (a.level_off - 2 * width + 2 * t);
    const Digest l = load_digest_global(ch), r = load_digest_global(ch + 32);
    tree_node(cw, a.off_bools, B, a.level, a.level_off, t, nb, l, r);
}

// batch tail + MapReduceSubchainVariable record of job qj (builder.rs:229-270) given the root of its commitment tree and the
// assertion bits of its slots
__device__ __forceinline__ void batch_tail(const SubchainArgs& a, uint32_t qj, const Digest& root, uint32_t fail, uint32_t first_bad) {
    const uint32_t B = a.batch;
    uint8_t* cwj = a.compact + (uint64_t)qj * a.compact_stride;
    uint32_t* Wj = reinterpret_cast<uint32_t*>(cwj + a.off_words);
    uint8_t* Bj = cwj + a.off_bools;
    const uint64_t Ej = a.ranges[qj / a.job_count].end_block;
    uint64_t bs, be, te, ebn;
    batch_bounds(Wj, Ej, bs, be, te, ebn);
    const bool enabled = bs < Ej;                                      // :174
    const uint64_t jst = Ej - 1 - bs;                                  // slot index of the last block (wrapping)
    const uint32_t mj = enabled ? (uint32_t)((jst < (uint64_t)B) ? jst + 1 : B) : 0u;
    const uint8_t* sl = cwj + bsx_off_slots(B);
    const Digest first = load_digest_global(cwj + bsx_off_start_header());
    const bool curr_enabled_end = enabled && !(jst < (uint64_t)B);     // enabled after the last slot
    const Digest curr_final = (mj > 0) ? load_digest_global(sl + BSX_SLOT_BYTES * (mj - 1) + 160 + 128) : first;
    const Digest end_header = load_digest_global(cwj + bsx_off_end_header());
    const bool last_disabled = !curr_enabled_end;                     // :229
    const bool last_matches = digest_eq(curr_final, end_header);      // :230
    const bool end_header_check = last_disabled || last_matches;      // :231
    const bool gte = ebn >= bs;                                       // :113 (A1)
    const uint64_t nb_blocks = ebn - bs;                              // :119
    const uint64_t last = Ej - 1;                                     // :177
    if (!end_header_check) { fail |= BSX_A6_BATCH_END; if (first_bad == 0xffffffffu) first_bad = B; }
    if (!gte) { fail |= BSX_A1_END_GTE_START; if (first_bad == 0xffffffffu) first_bad = B; }
    if ((nb_blocks >> 32) != 0) { fail |= BSX_A2_NB_BLOCKS_U32; if (first_bad == 0xffffffffu) first_bad = B; }
    uint8_t* t = Bj + bsx_b_tail(B);
    t[0] = last_disabled; t[1] = last_matches; t[2] = end_header_check; t[3] = be < Ej; t[4] = te < bs; t[5] = gte;
    Bj[BSX_B_BATCH_ENABLED] = enabled;
    Bj[bsx_b_rec_enabled(B)] = enabled;
    Wj[BSX_W_LAST_TO_PROCESS] = (uint32_t)last; Wj[BSX_W_LAST_TO_PROCESS + 1] = (uint32_t)(last >> 32);
    Wj[bsx_w_temp_end(B)] = (uint32_t)te; Wj[bsx_w_temp_end(B) + 1] = (uint32_t)(te >> 32);
    Wj[bsx_w_end_block_num(B)] = (uint32_t)ebn; Wj[bsx_w_end_block_num(B) + 1] = (uint32_t)(ebn >> 32);
    Wj[bsx_w_nb_blocks(B)] = (uint32_t)nb_blocks; Wj[bsx_w_nb_blocks(B) + 1] = (uint32_t)(nb_blocks >> 32);
    Wj[bsx_w_rec_start(B)] = (uint32_t)bs; Wj[bsx_w_rec_start(B) + 1] = (uint32_t)(bs >> 32);
    Wj[bsx_w_rec_end(B)] = (uint32_t)ebn; Wj[bsx_w_rec_end(B) + 1] = (uint32_t)(ebn >> 32);
    uint8_t* rec_b = cwj + bsx_off_record(B);
    store_digest_global(rec_b, first);
    store_digest_global(rec_b + 32, curr_final);
    store_digest_global(rec_b + 64, root);
    bsx_subchain* out = a.records + qj;
    out->start_block = bs;
    out->end_block = ebn;
    store_digest_global(out->start_header, first);
    store_digest_global(out->end_header, curr_final);
    store_digest_global(out->data_merkle_root, root);
    out->is_enabled = enabled ? 1u : 0u;
    out->assert_fail = fail;
    out->first_bad_slot = first_bad;
    out->_pad = 0;
}

// ---- stage 3: predicates + batch tail (builder.rs:174-270), no hashing.  256 consecutive slots per workgroup;
// per-slot assertion bits are reduced with a wave ballot and at most one LDS atomic per failing lane.
constexpr int BF_THREADS = 256;
constexpr uint32_t BF_TOP_WIDTH = 8;
// FUSED (the hint supplied the path digests, BSX_SUBCHAIN_PATHS_FROM_HINT): the whole of prove_subchain in this one launch —
// the slot's data-root tuple and its leaf hash (what k_slot_hashes<false> does), then EVERY level of the commitment tree
// through LDS, then the predicates.  The separate launches (k_slot_hashes, one k_tree_level per wide level) each held a few
// compressions per lane and cost 25-45 us of launch gap and memory round trips: 0.23 -> 0.1 ms per 262,144 slots.
template <bool FUSED>
__global__ __launch_bounds__(BF_THREADS, 4) void k_batch_finish(SubchainArgs a) {
    BSX_CHAIN_PRIO();
    __shared__ uint32_t job_fail[BF_THREADS];
    __shared__ uint32_t job_first_bad[BF_THREADS];
    // the top of every job's commitment tree (levels of width <= BF_TOP_WIDTH; FUSED: all levels) is folded here instead of
    // by one k_tree_level launch per level: those launches held 4 .. 32 nodes per job and cost ~28 us each, mostly launch gap
    __shared__ uint32_t top_nodes[2][BF_THREADS / 2 * 8];
    __shared__ uint32_t leaf_lds[FUSED ? BF_THREADS * 8 : 8];
    const uint32_t B = a.batch, tid = threadIdx.x;
    const uint64_t total = (uint64_t)a.n_jobs * B;
    const uint64_t gs0 = (uint64_t)blockIdx.x * BF_THREADS;
    const uint64_t gs = gs0 + tid;
    const bool live = gs < total;
    const uint32_t q = live ? (uint32_t)(gs / B) : 0, i = live ? (uint32_t)(gs % B) : 0;
    const uint32_t jl = live ? q - (uint32_t)(gs0 / B) : 0;          // job index inside the block (B <= BF_THREADS, aligned)
    job_fail[tid] = 0;
    job_first_bad[tid] = 0xffffffffu;
    __syncthreads();
    uint8_t* cw = a.compact + (uint64_t)q * a.compact_stride;
    uint32_t* W = reinterpret_cast<uint32_t*>(cw + a.off_words);
    uint8_t* Bo = cw + a.off_bools;
    const bsx_shared_ctx* rg = a.ranges + q / a.job_count;
    const uint64_t E = rg->end_block;
    uint64_t batch_start, batch_end, temp_end, end_block_num;
    batch_bounds(W, E, batch_start, batch_end, temp_end, end_block_num);
    const uint64_t curr_idx = batch_start + i;                       // builder.rs:182 / :134
    Digest tleaf = Digest{};
    if (FUSED && live) {
        // data-root tuple (builder.rs:82-103,134-137) and its leaf hash (:144-147): data_hash = data_hash_proofs[i].leaf[2..34]
        const uint8_t* pr = cw + bsx_off_dh_proofs(B) + BSX_DH_PROOF_SIZE * (BF_SKIP(a, 1) ? 0u : i);
        const uint4 l0 = ldu4(pr + 128), l1 = ldu4(pr + 144);
        const uint32_t l2 = (uint32_t)reinterpret_cast<const uint16_t*>(pr + 160)[0];
        const uint32_t lf[9] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2};
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = 0;
        t[6] = (uint32_t)(curr_idx >> 32);
        t[7] = (uint32_t)curr_idx;
#pragma unroll
        for (int k = 0; k < 8; k++) t[8 + k] = bswap32(funnel_r(lf[k + 1], lf[k], 16));
        tleaf = leaf_hash_tuple(t);
        uint8_t* tp = cw + bsx_off_tuples(B) + 64 * i;
        if (!BF_SKIP(a, 0)) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                stu4(tp + 16 * k, make_uint4(bswap32(t[4 * k]), bswap32(t[4 * k + 1]), bswap32(t[4 * k + 2]), bswap32(t[4 * k + 3])));
            store_digest_u(cw + bsx_off_leaf_hashes(B) + 32 * i, tleaf);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) leaf_lds[tid * 8 + k] = tleaf.w[k];
    }

    // Enabled slots form a prefix [0, m) of the batch (closed form of the :174-175,:225 recurrence).
    const bool batch_enabled = batch_start < E;                      // :174
    const uint64_t last_to_process = E - 1;                          // :177
    const uint64_t jstar = last_to_process - batch_start;            // slot index of the last block (wrapping)
    const uint32_t m = batch_enabled ? (uint32_t)((jstar < (uint64_t)B) ? jstar + 1 : B) : 0u;
    const bool en_before = i < m;                                    // curr_block_enabled entering slot i
    const bool is_last = (last_to_process == curr_idx);              // :185
    const uint8_t* slots = cw + bsx_off_slots(B);
    // curr_header entering slot i = start_header (i == 0 or m == 0) else lb_root of slot min(i, m) - 1
    const Digest start_header = load_digest_global(cw + bsx_off_start_header());
    const uint32_t i_ld = BF_SKIP(a, 1) ? 0u : i;                   // (experiments: every lane reads slot 0's lines)
    const Digest dh_root = load_digest_global(slots + BSX_SLOT_BYTES * i_ld + 128);
    const Digest lb_root = load_digest_global(slots + BSX_SLOT_BYTES * i_ld + 160 + 128);
    Digest curr_before = start_header, curr_after = start_header;
    if (B <= 64) {
        // a job's slots are consecutive lanes of ONE wave (B divides the workgroup): the neighbouring slots' lb_root comes through the
        // wave's crossbar instead of two more 32-byte loads at slot stride (round 5; 3 instead of 5 strided digest loads per slot)
        const uint32_t base = (tid & 63u) & ~(B - 1u);
        const uint32_t ib = min(i, m), ia = min(i + 1, m);
        const int sb = (int)(base + (ib ? ib - 1 : 0)), sa = (int)(base + (ia ? ia - 1 : 0));
        Digest nb, na;
#pragma unroll
        for (int k = 0; k < 8; k++) { nb.w[k] = __shfl(lb_root.w[k], sb, 64); na.w[k] = __shfl(lb_root.w[k], sa, 64); }
        if (i > 0 && m > 0) curr_before = nb;
        if (m > 0) curr_after = na;
    } else {
        if (i > 0 && m > 0) curr_before = load_digest_global(slots + BSX_SLOT_BYTES * (min(i, m) - 1) + 160 + 128);
        if (m > 0) curr_after = load_digest_global(slots + BSX_SLOT_BYTES * (min(i + 1, m) - 1) + 160 + 128);
    }
    Digest claimed;                                                  // last_block_id_proofs[i].leaf[2..34] (:204)
    {
        // 2-byte aligned in the packed image: two misaligned 16-byte loads (gfx950 serves them) instead of nine dwords + funnels
        const uint8_t* lf = cw + bsx_off_lb_proofs(B) + BSX_LB_PROOF_SIZE * i_ld + 128 + 2;
        const uint4 c0 = ldu4(lf), c1 = ldu4(lf + 16);
        const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        claimed = digest_from_le(c);
    }
    const Digest H_E = load_digest_global(rg->end_header_hash);
    const bool valid_prev = digest_eq(curr_before, claimed);         // :205
    const bool prev_check = !en_before || valid_prev;                // :206
    const bool dh_valid = digest_eq(dh_root, claimed);               // :210
    const bool dh_check = !en_before || dh_valid;                    // :211
    const bool root_matches_end = digest_eq(lb_root, H_E);           // :216
    const bool end_check = !is_last || root_matches_end;             // :218
    const bool en_after = en_before && !is_last;                     // :225
    const uint32_t nb_enabled = (uint32_t)(end_block_num - batch_start);   // :119,124
    if (live && !BF_SKIP(a, 0)) {
        store_digest_global(cw + bsx_off_slots(B) + BSX_SLOT_BYTES * i + 320, curr_after);   // :223
        const uint2 idx2 = make_uint2((uint32_t)curr_idx, (uint32_t)(curr_idx >> 32));
        *reinterpret_cast<uint2*>(W + BSX_W_CURR_IDX + 2 * i) = idx2;
        *reinterpret_cast<uint2*>(W + bsx_w_block_height(B) + 2 * i) = idx2;
        // the slot's nine bools (9-byte stride): one misaligned 8-byte store + one byte instead of nine byte stores
        uint8_t* b = Bo + BSX_B_SLOTS + BSX_SLOT_BOOLS * i;
        const uint32_t blo = (uint32_t)!en_before | ((uint32_t)is_last << 8) | ((uint32_t)valid_prev << 16) | ((uint32_t)prev_check << 24);
        const uint32_t bhi = (uint32_t)dh_valid | ((uint32_t)dh_check << 8) | ((uint32_t)root_matches_end << 16) | ((uint32_t)end_check << 24);
        *reinterpret_cast<uint2*>(b) = make_uint2(blo, bhi);
        b[8] = en_after;
        Bo[bsx_b_leaf_enabled(B) + i] = i < nb_enabled;
    }
    {
        const uint32_t f = live ? ((prev_check ? 0u : BSX_A3_PREV_HEADER) | (dh_check ? 0u : BSX_A4_DATA_HASH_PROOF) |
                                   (end_check ? 0u : BSX_A5_END_HEADER)) : 0u;
        if (__ballot(f != 0)) {          // rare path: some lane of this wave failed an assertion
            if (f) { atomicOr(&job_fail[jl], f); atomicMin(&job_first_bad[jl], i); }
        }
    }
    __syncthreads();
    // tree levels: level a.level (width a.width, at a.level_off) reads what the last k_tree_level launch (or, for
    // B <= 2 * BF_TOP_WIDTH, k_slot_hashes) left in global memory — FUSED: this workgroup's leaf hashes in LDS; the levels
    // above hand over through LDS.  The nodes of a level are dealt to the workgroup's lanes DENSELY (node n of the block =
    // lane n, whatever job it belongs to), so that a level of 128 / 64 / 32 nodes occupies 2 / 1 / 1 waves and the other waves
    // skip it, instead of every wave issuing every level for a shrinking prefix of its lanes.
    __shared__ uint32_t job_nb[BF_THREADS];
    if (live && i == 0) job_nb[jl] = nb_enabled;
    __syncthreads();
    // From here on a wave stays only while some lane of it still has a node to hash or a job's tail to write (lanes < jobs_here *
    // max(width, 1)); the others END.  A barrier counts the surviving waves of the workgroup only, and a wave that has ended gives
    // its registers and its slot to the next workgroup — held to the last barrier instead, three of four waves sat idle through
    // 10 of the workgroup's 15 dependent compressions (B = 64: 0.43 of the wave slots doing work; now 0.77).
    const uint32_t q0 = (uint32_t)(gs0 / B), jobs_here = BF_THREADS / B;
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(tid & ~63u);
    uint32_t cur = 0;
    const bool top = a.top_inputs != 0;                                 // k_batch_top takes over below width stop_w: hand it the slots' bits
    const uint32_t stop_w = top ? a.top_inputs / 2 : 0;
    if (top && tid < jobs_here && q0 + tid < a.n_jobs) {
        a.records[q0 + tid].assert_fail = job_fail[tid];
        a.records[q0 + tid].first_bad_slot = job_first_bad[tid];
    }
    if (B > 1) {
        if (a.width <= stop_w || wave_base >= jobs_here * a.width) return;
        uint32_t level = a.level, level_off = a.level_off;
        for (uint32_t width = a.width; width >= 1; width /= 2, level++) {
            const uint32_t jn = tid / width, t = tid % width;          // job (inside the block) and node of this lane
            if (jn < jobs_here && q0 + jn < a.n_jobs) {
                uint8_t* cwn = a.compact + (uint64_t)(q0 + jn) * a.compact_stride;
                Digest l, r;
                if (FUSED && width == a.width) {                         // level 1: slot jn * B + 2 t of this workgroup
                    const uint32_t* s = &leaf_lds[(jn * B + 2 * t) * 8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { l.w[k] = s[k]; r.w[k] = s[8 + k]; }
                } else if (width == a.width) {
                    const uint8_t* ch = (level == 1) ? cwn + bsx_off_leaf_hashes(B) + 64 * t
                                                     : cwn + bsx_off_nodes(B) + 32 * (level_off - 2 * width + 2 * t);
                    l = load_digest_global(ch); r = load_digest_global(ch + 32);
                } else {
                    const uint32_t* s = &top_nodes[cur][(jn * a.width + 2 * t) * 8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { l.w[k] = s[k]; r.w[k] = s[8 + k]; }
                }
                const Digest node = tree_node(cwn, a.off_bools, B, level, level_off, t, job_nb[jn], l, r);
                uint32_t* d = &top_nodes[cur ^ 1][(jn * a.width + t) * 8];
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = node.w[k];
            }
            __syncthreads();
            cur ^= 1;
            level_off += width;
            // lanes of the next level / of the tail; none once k_batch_top has the rest
            const uint32_t later = (top && width / 2 <= stop_w) ? 0u : jobs_here * (width / 2 > 1 ? width / 2 : 1);
            if (wave_base >= later) return;
        }
    }
    if (top) return;
    // batch tail + record (builder.rs:229-270): lane jn of the workgroup for its job jn
    if (tid < jobs_here && q0 + tid < a.n_jobs) {
        Digest root;
        if (B == 1) root = FUSED ? tleaf : load_digest_global(a.compact + (uint64_t)(q0 + tid) * a.compact_stride + bsx_off_leaf_hashes(B));   // own slot
        else {
#pragma unroll
            for (int k = 0; k < 8; k++) root.w[k] = top_nodes[cur][(tid * a.width) * 8 + k];   // node 0 of the last level written
        }
        batch_tail(a, q0 + tid, root, job_fail[tid], job_first_bad[tid]);
    }
}

// The top of every job's commitment tree, one lane per job: N sub-tree roots (the level k_batch_finish stopped at, or the leaf hashes)
// -> N - 1 nodes depth first (log2 N digests live), then the batch tail.  A workgroup of k_batch_finish holds too few nodes of
// these levels to fill a wave (B = 64: 16, 8, 4 of 64 lanes, each level 2 dependent compressions behind a barrier) — a third of the
// kernel's wave time at a quarter of the lanes; here the same nodes are 1 / 8 of the lanes' time at full width.
template <int N>
__device__ __forceinline__ Digest top_subtree(const SubchainArgs& a, uint8_t* cw, uint32_t nb, uint32_t first) {
    if constexpr (N == 1) {
        const uint8_t* in = (a.level == 1) ? cw + bsx_off_leaf_hashes(a.batch) : cw + bsx_off_nodes(a.batch) + 32 * (a.level_off - 2 * a.width);
        return load_digest_global(in + 32 * first);
    } else {
        const Digest l = top_subtree<N / 2>(a, cw, nb, first);
        const Digest r = top_subtree<N / 2>(a, cw, nb, first + N / 2);
        uint32_t level = a.level, off = a.level_off, w = a.width;
#pragma unroll
        for (int n = 2; n < N; n *= 2) { level++; off += w; w /= 2; }
        return tree_node(cw, a.off_bools, a.batch, level, off, first / N, nb, l, r);
    }
}
template <int N>
__global__ __launch_bounds__(64) void k_batch_top(SubchainArgs a) {
    BSX_CHAIN_PRIO();
    const uint32_t q = blockIdx.x * 64 + threadIdx.x;
    if (q >= a.n_jobs) return;
    uint8_t* cw = a.compact + (uint64_t)q * a.compact_stride;
    uint64_t bs, be, te, ebn;
    batch_bounds(reinterpret_cast<const uint32_t*>(cw + a.off_words), a.ranges[q / a.job_count].end_block, bs, be, te, ebn);
    const Digest root = top_subtree<N>(a, cw, (uint32_t)(ebn - bs), 0);
    batch_tail(a, q, root, a.records[q].assert_fail, a.records[q].first_bad_slot);
}

// ------------------------------------------------------------------------------------------------ k_reduce
// One workgroup per range: n (power of two <= 256) records -> 1, level by level in LDS.
struct ReduceArgs {
    uint32_t n_ranges, n;
    uint64_t stride_range, stride_record;   // record k of range r = records[r * stride_range + k * stride_record]
    const bsx_subchain* records;
    bsx_subchain* out;
    uint8_t* reduce_compact;       // optional: (n-1) node witnesses per range
    uint32_t compact_stride, off_words, off_bools;
    // optional: the final assertions + public output of k_finalize for the range, by the lane that holds the root record (the host
    // tier's single proof request: one launch and one kernel boundary less on its critical chain)
    uint32_t fin_jobs, fin_batch;
    const bsx_shared_ctx* fin_ranges;
    const uint8_t* fin_target_hashes;
    uint8_t* fin_output64;
    uint32_t* fin_status;
};
__global__ __launch_bounds__(128) void k_reduce(ReduceArgs a) {
    __shared__ bsx_subchain rec[256];
    const uint32_t r = blockIdx.x, tid = threadIdx.x, n = a.n;
    {   // copy in: n records of 128 bytes, 8 lanes per record
        uint4* dst = reinterpret_cast<uint4*>(rec);
        for (uint32_t c = tid; c < n * 8; c += blockDim.x) {
            const uint4* src = reinterpret_cast<const uint4*>(a.records + (uint64_t)r * a.stride_range + (uint64_t)(c >> 3) * a.stride_record);
            dst[c] = src[c & 7];
        }
    }
    __syncthreads();
    uint32_t k0 = 0;
    for (uint32_t mcount = n; mcount > 1; mcount /= 2) {
        bsx_subchain o;
        const bool act = tid < mcount / 2;
        if (act) {
            const bsx_subchain& l = rec[2 * tid];
            const bsx_subchain& rt = rec[2 * tid + 1];
            const Digest l_end = load_digest_global(l.end_header), r_start = load_digest_global(rt.start_header);
            const Digest l_root = load_digest_global(l.data_merkle_root), r_root = load_digest_global(rt.data_merkle_root);
            const bool right_disabled = rt.is_enabled == 0;                       // builder.rs:344
            const bool headers_linked = digest_eq(l_end, r_start);                // :348-349
            const bool blocks_linked = l.end_block == rt.start_block;             // :350
            const bool linked = headers_linked && blocks_linked;                  // :351
            const bool link_check = right_disabled || linked;                     // :352
            const Digest computed = inner_hash(l_root, r_root);                   // :357-364
            const Digest root = right_disabled ? l_root : computed;               // :367-371
            o.start_block = l.start_block;                                        // :389
            o.end_block = right_disabled ? l.end_block : rt.end_block;            // :374-378
            {   // 32-byte fields as dwords (the records are 8-byte aligned in LDS)
                const uint32_t* ls = reinterpret_cast<const uint32_t*>(l.start_header);
                const uint32_t* le = reinterpret_cast<const uint32_t*>(l.end_header);
                const uint32_t* re = reinterpret_cast<const uint32_t*>(rt.end_header);
                uint32_t* os = reinterpret_cast<uint32_t*>(o.start_header);
                uint32_t* oe = reinterpret_cast<uint32_t*>(o.end_header);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    os[q] = ls[q];                                                // :390
                    oe[q] = right_disabled ? le[q] : re[q];                       // :379-383
                }
            }
            store_digest_global(o.data_merkle_root, root);
            o.is_enabled = l.is_enabled;                                          // :388
            const uint32_t child = l.assert_fail | rt.assert_fail;
            o.assert_fail = child | (link_check ? 0u : BSX_A8_REDUCE_LINK);
            const uint32_t k = k0 + tid;
            if (!link_check && !(child & BSX_A8_REDUCE_LINK)) o.first_bad_slot = k;
            else o.first_bad_slot = (l.first_bad_slot != 0xffffffffu) ? l.first_bad_slot : rt.first_bad_slot;
            o._pad = 0;
            if (a.reduce_compact) {
                uint8_t* cw = a.reduce_compact + ((uint64_t)r * (n - 1) + k) * a.compact_stride;
                store_digest_global(cw, computed);
                {
                    uint32_t* c32 = reinterpret_cast<uint32_t*>(cw);
                    const uint32_t* os = reinterpret_cast<const uint32_t*>(o.start_header);
                    const uint32_t* oe = reinterpret_cast<const uint32_t*>(o.end_header);
                    const uint32_t* om = reinterpret_cast<const uint32_t*>(o.data_merkle_root);
#pragma unroll
                    for (int q = 0; q < 8; q++) { c32[8 + q] = os[q]; c32[16 + q] = oe[q]; c32[24 + q] = om[q]; }
                }
                uint32_t* W = reinterpret_cast<uint32_t*>(cw + a.off_words);
                W[0] = (uint32_t)o.start_block; W[1] = (uint32_t)(o.start_block >> 32);
                W[2] = (uint32_t)o.end_block; W[3] = (uint32_t)(o.end_block >> 32);
                uint8_t* b = cw + a.off_bools;
                b[0] = right_disabled; b[1] = headers_linked; b[2] = blocks_linked; b[3] = linked; b[4] = link_check; b[5] = (uint8_t)o.is_enabled;
            }
        }
        __syncthreads();
        if (act) rec[tid] = o;
        __syncthreads();
        k0 += mcount / 2;
    }
    if (tid < 8) reinterpret_cast<uint4*>(a.out + r)[tid] = reinterpret_cast<uint4*>(rec)[tid];
    if (a.fin_output64 && tid == 0) {                                     // builder.rs:292-297,400-406; header_range.rs:57-58 (= k_finalize)
        const bsx_shared_ctx& rg = a.fin_ranges[r];
        const bsx_subchain& res = rec[0];
        uint32_t st = res.assert_fail;
        if (!(rg.end_block <= rg.start_block + (uint64_t)a.fin_jobs * a.fin_batch)) st |= BSX_A7_RANGE;
        bool ok = res.start_block == rg.start_block && res.end_block == rg.end_block;
        for (int q = 0; q < 32; q++) ok = ok && res.start_header[q] == rg.start_header_hash[q] && res.end_header[q] == rg.end_header_hash[q];
        if (!ok) st |= BSX_A9_FINAL;
        const uint8_t* th = a.fin_target_hashes ? a.fin_target_hashes + 32 * (uint64_t)r : rg.end_header_hash;
        for (int q = 0; q < 32; q++) { a.fin_output64[64 * (uint64_t)r + q] = th[q]; a.fin_output64[64 * (uint64_t)r + 32 + q] = res.data_merkle_root[q]; }
        if (a.fin_status) a.fin_status[r] = st;
    }
}

// ------------------------------------------------------------------------------------------------ k_finalize
// dc_units (optional): the SKIP unit of range r (include/bsx_layout.h) starts at dc_units + r * dc_stride; its public output
// data_commitment (byte 64, header_range.rs:58) is this kernel's to write
__global__ void k_finalize(uint32_t n_ranges, uint32_t nb_map_jobs, uint32_t batch, const bsx_shared_ctx* ranges,
                           const bsx_subchain* results, const uint8_t* target_hashes, uint8_t* output64, uint32_t* status,
                           uint8_t* dc_units, uint32_t dc_stride) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const bsx_shared_ctx& rg = ranges[r];
    const bsx_subchain& res = results[r];
    uint32_t st = res.assert_fail;
    const uint64_t max_blocks = (uint64_t)nb_map_jobs * batch;
    if (!(rg.end_block <= rg.start_block + max_blocks)) st |= BSX_A7_RANGE;            // builder.rs:292-297
    bool ok = res.start_block == rg.start_block && res.end_block == rg.end_block;        // :400-406
    for (int q = 0; q < 32; q++) ok = ok && res.start_header[q] == rg.start_header_hash[q] && res.end_header[q] == rg.end_header_hash[q];
    if (!ok) st |= BSX_A9_FINAL;
    if (output64) {
        const uint8_t* th = target_hashes ? target_hashes + 32 * (uint64_t)r : rg.end_header_hash;
        for (int q = 0; q < 32; q++) { output64[64 * (uint64_t)r + q] = th[q]; output64[64 * (uint64_t)r + 32 + q] = res.data_merkle_root[q]; }  // header_range.rs:57-58
    }
    if (dc_units)
        for (int q = 0; q < 32; q++) dc_units[(uint64_t)r * dc_stride + 64 + q] = res.data_merkle_root[q];
    if (status) status[r] = st;
}

// ------------------------------------------------------------------------------------------------ k_expand_witness
// HBM-write bound: every lane emits 16 bytes (two Goldilocks elements) per store, lane-contiguous (1 KiB per wave
// store).  A job's expanded image starts at job * n_elements * 8 bytes (8-byte aligned only), so lanes are aligned to
// GLOBAL 128-byte lines and the elements of a pair that belong to a neighbouring job are left to that job's lanes.
struct ExpandArgs {
    bsx_witness_layout lay;
    uint32_t n_jobs, blocks_per_job;
    const uint8_t* compact;
    uint64_t* out;
};
__device__ __forceinline__ uint64_t expand_elem(const ExpandArgs& a, const uint8_t* c, int64_t e) {
    const int64_t nbits = 8ll * a.lay.n_bytes;
    if (e < nbits) return (c[e >> 3] >> (7 - (e & 7))) & 1u;
    e -= nbits;
    if (e < (int64_t)a.lay.n_words) return reinterpret_cast<const uint32_t*>(c + a.lay.off_words)[e];
    e -= a.lay.n_words;
    return c[a.lay.off_bools + e];
}
// Each workgroup stages EX_CHUNK source bytes in LDS with ONE coalesced dword load per lane, then every lane reads
// the byte(s) of its pair from LDS: the vector-memory pipe carries (almost) nothing but the 16-byte stores.
constexpr int EX_THREADS = 256;
template <int EX_CHUNK, bool NT>
__global__ __launch_bounds__(EX_THREADS) void k_expand_witness(ExpandArgs a) {
    constexpr int EX_PAIRS_PER_BLOCK = EX_CHUNK * 4;
    __shared__ uint32_t lds[EX_CHUNK / 4 + 2];
    const uint32_t tid = threadIdx.x;
    // memory-bound waves issue first; the ALU-bound commit-verification waves co-resident on the CU fill the gaps
    // (measured +1.5 % on the step with the Ed25519 side stream running)
    __builtin_amdgcn_s_setprio(3);
    // work item = (job, 1 KiB source chunk); a one-shot grid has one item per workgroup, a capped grid strides
    for (uint32_t item = blockIdx.x; item < a.n_jobs * a.blocks_per_job; item += gridDim.x) {
    const uint32_t job = item / a.blocks_per_job, bx = item % a.blocks_per_job;
    const uint8_t* c = a.compact + (uint64_t)job * a.lay.compact_stride;
    const uint64_t nel = a.lay.n_elements;
    const uint64_t g0 = (uint64_t)job * nel;                 // global element index of this job's first element
    // Lanes are aligned to GLOBAL 128-byte lines: local pair p covers local elements 2p - sh, 2p - sh + 1 with
    // sh = g0 mod 16, so that every wave-store (64 lanes x 16 B) is eight whole 128-byte lines wherever the job's image
    // starts (a job is 449,755 elements: its base is only 8-byte aligned).  With 16-byte alignment only, every wave-store
    // straddled two lines that a neighbouring wave completed later: non-temporal stores then reached HBM as partial
    // lines (WRITE_SIZE 1.03x the algorithmic bytes) at a rate that depended on the buffer's placement.
    const uint32_t sh = (uint32_t)(g0 & 15);
    const uint32_t nbits = 8u * a.lay.n_bytes;
    const uint32_t npairs = (uint32_t)((nel + sh + 1) / 2);
    const uint32_t pbase = bx * EX_PAIRS_PER_BLOCK;
    const int32_t byte0 = (int32_t)(pbase / 4) - 4;          // staged window starts one dword early (pairs look back up to 15 bits)
    for (uint32_t t = tid; t < EX_CHUNK / 4 + 2; t += EX_THREADS) {
        const int32_t bi = byte0 + 4 * (int32_t)t;
        lds[t] = (bi >= 0 && (uint32_t)bi < a.lay.compact_stride) ? reinterpret_cast<const uint32_t*>(c)[bi >> 2] : 0u;
    }
    __syncthreads();
    const uint8_t* lb = reinterpret_cast<const uint8_t*>(lds);
    uint64_t* base = a.out + g0 - sh;                        // 128-byte aligned
#pragma unroll 4
    for (int u = 0; u < EX_PAIRS_PER_BLOCK / EX_THREADS; u++) {
        const uint32_t p = pbase + (uint32_t)u * EX_THREADS + tid;
        if (p >= npairs) break;
        const int64_t e0 = 2ll * p - sh, e1 = e0 + 1;        // local element indices of the pair (negative: previous job's)
        const bool in0 = e0 >= 0 && e0 < (int64_t)nel, in1 = e1 >= 0 && e1 < (int64_t)nel;
        uint64_t v0, v1;
        if (in0 && e1 < (int64_t)nbits) {                    // both elements are bits (the overwhelmingly common case)
            const uint32_t b0 = (uint32_t)e0, b1 = (uint32_t)e1;
            const uint32_t by0 = lb[(int32_t)(b0 >> 3) - byte0], by1 = lb[(int32_t)(b1 >> 3) - byte0];
            v0 = (by0 >> (7 - (b0 & 7))) & 1u;
            v1 = (by1 >> (7 - (b1 & 7))) & 1u;
        } else {
            v0 = in0 ? expand_elem(a, c, e0) : 0;
            v1 = in1 ? expand_elem(a, c, e1) : 0;
        }
        uint64_t* dst = base + 2 * (uint64_t)p;
        if (in0 && in1) {
            typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
            const v2u64 vv = {v0, v1};
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v2u64*>(dst));
            else *reinterpret_cast<v2u64*>(dst) = vv;
        } else {
            if (in0) dst[0] = v0;
            if (in1) dst[1] = v1;
        }
    }
    __syncthreads();   // the staging buffer is reused by the next item
    }
}

}  // namespace bsx

// ------------------------------------------------------------------------------------------------ launchers (called by api.hip)
extern "C" {
using namespace bsx;

hipError_t bsxk_header_merkle(hipStream_t s, const bsx_header* hdr, uint64_t n, uint8_t* hashes, uint8_t* dh, uint8_t* lb, uint8_t* paths,
                              uint32_t* status, uint32_t max_wgs, uint32_t low_prio, const bsxk_merkle_tap* tap, uint64_t status_group) {
    if (!n) return hipSuccess;
    uint32_t grid = (uint32_t)((n + HM_GROUP - 1) / HM_GROUP);
    // cap on the grid (the workgroups then stride over the header groups): the context's BSX_TUNE_MERKLE_WORKGROUPS (an experiments
    // build also reads BSX_MERKLE_WGS); 0 = one workgroup per 64 headers
    static const long env_cap = bsx_knob("BSX_MERKLE_WGS", -1);
    const long cap = env_cap >= 0 ? env_cap : (long)max_wgs;
    if (cap > 0 && grid > (uint32_t)cap) grid = (uint32_t)cap;
    static const long env_lp = bsx_knob("BSX_MERKLE_LOW_PRIO", -1);
    hipLaunchKernelGGL(k_header_merkle, dim3(grid), dim3(HM_THREADS), 0, s, hdr, n, hashes, dh, lb, paths, status, env_lp >= 0 ? (uint32_t)env_lp : low_prio,
                       tap ? *tap : bsxk_merkle_tap{~0ull, nullptr, nullptr, nullptr, 0, nullptr}, status_group);
    return hipGetLastError();
}
hipError_t bsxk_zero_paths(hipStream_t s, uint8_t* out) {
    hipLaunchKernelGGL(k_zero_paths, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError();
}
hipError_t bsxk_assemble_inputs(hipStream_t s, uint32_t n_ranges, uint32_t J, uint32_t B, uint32_t job_first, uint32_t job_count, uint32_t span,
                                const bsx_shared_ctx* ranges, const uint64_t* latest, const bsx_header* headers, uint64_t hpr, uint64_t hfr,
                                const uint8_t* hashes, const uint8_t* dh, const uint8_t* lb, uint8_t* compact, uint32_t* status,
                                const uint8_t* paths, const uint8_t* zero_paths, uint32_t lds_pad, const uint32_t* spans, uint32_t status_per_range,
                                const uint32_t* jobs) {
    if (!n_ranges || !job_count) return hipSuccess;
    const bsx_witness_layout L = bsx_map_layout(B);
    AssembleArgs a{n_ranges, J, B, job_first, job_count, span, ranges, latest, headers, hpr, hfr, hashes, dh, lb, compact, L.compact_stride, L.off_words, status,
                   paths, zero_paths, spans, jobs, status_per_range};
    const uint32_t it = (24u * B + 255u) / 256u;
    // lds_pad: bytes of (unused) dynamic LDS per workgroup = a cap on the hint's workgroups resident per CU.  The hint is 0.7 GB of
    // 16-byte copies: at full occupancy it saturates HBM for 0.17 ms, and the header hashing it runs beside in the compact pipeline
    // WAITS for its own loads (a round trip per leaf) — with two hint workgroups per CU (64 KB each) the copies spread over 0.3 ms
    // and the step is 0.07 ms shorter.  Beside an expansion the opposite holds (loads in flight are what the hint needs): 0 there.
    // BSX_HINT_LDS (experiments) overrides.
    static const long env_lds = bsx_knob("BSX_HINT_LDS", -1);
    const size_t hint_lds = env_lds >= 0 ? (size_t)env_lds : (size_t)lds_pad;
#define BSX_AS_LAUNCH(N) hipLaunchKernelGGL(k_assemble_inputs<N>, dim3(n_ranges * job_count), dim3(256), hint_lds, s, a)
    if (it <= 1) BSX_AS_LAUNCH(1);
    else if (it <= 2) BSX_AS_LAUNCH(2);
    else if (it <= 3) BSX_AS_LAUNCH(3);
    else if (it <= 6) BSX_AS_LAUNCH(6);
    else if (it <= 12) BSX_AS_LAUNCH(12);
    else BSX_AS_LAUNCH(24);
#undef BSX_AS_LAUNCH
    return hipGetLastError();
}
hipError_t bsxk_prove_subchain(hipStream_t s, uint32_t n_ranges, uint32_t B, uint32_t job_count, const bsx_shared_ctx* ranges,
                               uint8_t* compact, bsx_subchain* records, uint32_t flags) {
    if (!n_ranges || !job_count) return hipSuccess;
    const bsx_witness_layout L = bsx_map_layout(B);
    const uint32_t n_jobs = n_ranges * job_count;
    SubchainArgs a{n_jobs, B, job_count, ranges, compact, L.compact_stride, L.off_words, L.off_bools, records, 0, 0, 0, 0, 0};
    a.exp_skip = (uint32_t)bsx_knob("BSX_BF_SKIP", 0);
    const uint64_t slots = (uint64_t)n_jobs * B;
    // BSX_SUBCHAIN_FUSED=0 / 1 (experiments) overrides BSX_SUBCHAIN_SEPARATE_LAUNCHES
    static const long env_fuse = bsx_knob("BSX_SUBCHAIN_FUSED", -1);
    const bool fuse = env_fuse >= 0 ? env_fuse != 0 : !(flags & BSX_SUBCHAIN_SEPARATE_LAUNCHES);
    if ((flags & BSX_SUBCHAIN_PATHS_FROM_HINT) && fuse) {
        a.level = 1; a.width = B / 2; a.level_off = 0;        // every level inside the one launch ...
        // ... of a small batch (a proof request: one kernel boundary less on its chain); a big one hands the levels that no longer
        // fill half a wave of a workgroup to k_batch_top
        static const long env_top = bsx_knob("BSX_SUBCHAIN_TOP", 1);
        SubchainArgs t = a;
        if (env_top && n_jobs >= 1024 && B > 1) {
            const uint32_t jobs_here = BF_THREADS / B;
            for (uint32_t w = a.width; w >= 1; t.level_off += w, w /= 2, t.level++)
                if (jobs_here * w < 32) { a.top_inputs = 2 * w; t.width = w; break; }
            if (a.top_inputs > 32) a.top_inputs = 0;           // (B = 256 asks for 64: not instantiated)
        }
        hipLaunchKernelGGL(k_batch_finish<true>, dim3((uint32_t)((slots + BF_THREADS - 1) / BF_THREADS)), dim3(BF_THREADS), 0, s, a);
        if (a.top_inputs) {
            t.top_inputs = a.top_inputs;
            const dim3 g((n_jobs + 63) / 64), b(64);
            switch (a.top_inputs) {
                case 2: hipLaunchKernelGGL(k_batch_top<2>, g, b, 0, s, t); break;
                case 4: hipLaunchKernelGGL(k_batch_top<4>, g, b, 0, s, t); break;
                case 8: hipLaunchKernelGGL(k_batch_top<8>, g, b, 0, s, t); break;
                case 16: hipLaunchKernelGGL(k_batch_top<16>, g, b, 0, s, t); break;
                default: hipLaunchKernelGGL(k_batch_top<32>, g, b, 0, s, t); break;
            }
        }
        return hipGetLastError();
    }
    if (flags & BSX_SUBCHAIN_PATHS_FROM_HINT)
        hipLaunchKernelGGL(k_slot_hashes<false>, dim3((uint32_t)((slots + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL(k_slot_hashes<true>, dim3((uint32_t)((slots + SH_THREADS - 1) / SH_THREADS)), dim3(SH_THREADS), 0, s, a);
    uint32_t level_off = 0, level = 1;
    for (uint32_t width = B / 2; width > BF_TOP_WIDTH; width /= 2, level++) {
        a.level = level; a.width = width; a.level_off = level_off;
        const uint64_t nodes = (uint64_t)n_jobs * width;
        hipLaunchKernelGGL(k_tree_level, dim3((uint32_t)((nodes + TR_THREADS - 1) / TR_THREADS)), dim3(TR_THREADS), 0, s, a);
        level_off += width;
    }
    // the remaining levels (width <= BF_TOP_WIDTH) run inside k_batch_finish
    a.level = level; a.width = B / 2 < BF_TOP_WIDTH ? B / 2 : BF_TOP_WIDTH; a.level_off = level_off;
    hipLaunchKernelGGL(k_batch_finish<false>, dim3((uint32_t)((slots + BF_THREADS - 1) / BF_THREADS)), dim3(BF_THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t bsxk_reduce(hipStream_t s, uint32_t n_ranges, uint32_t n, const bsx_subchain* records, uint64_t stride_range,
                       uint64_t stride_record, bsx_subchain* out, uint8_t* reduce_compact) {
    if (!n_ranges) return hipSuccess;
    const bsx_witness_layout L = bsx_reduce_layout();
    ReduceArgs a{n_ranges, n, stride_range, stride_record, records, out, reduce_compact, L.compact_stride, L.off_words, L.off_bools, 0, 0, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_reduce, dim3(n_ranges), dim3(128), 0, s, a);
    return hipGetLastError();
}
// reduce + the final assertions / public output in ONE launch (n records per range, consecutive)
hipError_t bsxk_reduce_finalize(hipStream_t s, uint32_t n_ranges, uint32_t n, const bsx_subchain* records, bsx_subchain* out, uint8_t* reduce_compact,
                                uint32_t J, uint32_t B, const bsx_shared_ctx* ranges, const uint8_t* target_hashes, uint8_t* output64, uint32_t* status) {
    if (!n_ranges) return hipSuccess;
    const bsx_witness_layout L = bsx_reduce_layout();
    ReduceArgs a{n_ranges, n, n, 1, records, out, reduce_compact, L.compact_stride, L.off_words, L.off_bools, J, B, ranges, target_hashes, output64, status};
    hipLaunchKernelGGL(k_reduce, dim3(n_ranges), dim3(128), 0, s, a);
    return hipGetLastError();
}
hipError_t bsxk_finalize(hipStream_t s, uint32_t n_ranges, uint32_t J, uint32_t B, const bsx_shared_ctx* ranges, const bsx_subchain* results,
                         const uint8_t* target_hashes, uint8_t* output64, uint32_t* status, uint8_t* dc_units, uint32_t dc_stride) {
    if (!n_ranges) return hipSuccess;
    hipLaunchKernelGGL(k_finalize, dim3((n_ranges + 63) / 64), dim3(64), 0, s, n_ranges, J, B, ranges, results, target_hashes, output64, status,
                       dc_units, dc_stride);
    return hipGetLastError();
}
hipError_t bsxk_expand_witness(hipStream_t s, const bsx_witness_layout* lay, uint32_t n_jobs, const uint8_t* compact, uint64_t* out) {
    if (!n_jobs) return hipSuccess;
    const uint64_t npairs = (lay->n_elements + 16) / 2;      // a job's pair grid may start up to 15 elements before its first element
    // Staging chunk and store flavour (tools/exp_expand_sweep.sh, tools/exp_alloc_variants.sh; round 2, after the stores became
    // 128-byte-line aligned): non-temporal stores with 256-byte chunks, one workgroup per item.  Alone the launch runs at
    // 2.1-2.6 ms (14.97 GB: 5.8-7.1 TB/s) depending on the box; plain stores 2.6-2.8 ms.  In the pipelined step, beside the
    // other chunk's hashing: 256 B uncapped 89.0-89.4 / 93.2-93.8 M headers/s on two boxes, 512 B 85.0-85.3 / 92.9-94.0, 256 B
    // capped at 262,144 workgroups (the round-1 choice) 79.9-85.1, plain stores 74-81.
    static const long chunk = bsx_knob("BSX_EXPAND_CHUNK", 256);
    static const long nt = bsx_knob("BSX_EXPAND_NT", 1);
    const uint32_t ppb = (uint32_t)chunk * 4;
    const uint32_t gx = (uint32_t)((npairs + ppb - 1) / ppb);
    ExpandArgs a{*lay, n_jobs, gx, compact, out};
    // BSX_EXPAND_BLOCKS > 0 caps the grid (workgroups then stride over the items); default: one workgroup per item
    static const long cap = bsx_knob("BSX_EXPAND_BLOCKS", 0);
    uint64_t grid = (uint64_t)gx * n_jobs;
    if (cap > 0 && grid > (uint64_t)cap) grid = (uint64_t)cap;
#define BSX_EX_LAUNCH(C, N)                                                                                              \
    do {                                                                                                                 \
        BSX_NOTE_FORM(BSX_FORM_EXPAND, (uint32_t)(C) | ((N) ? 1u << 16 : 0u) | (grid < (uint64_t)gx * n_jobs ? 1u << 17 : 0u)); \
        hipLaunchKernelGGL((k_expand_witness<C, N>), dim3((uint32_t)grid), dim3(EX_THREADS), 0, s, a);                      \
    } while (0)
#ifdef BSX_EXPERIMENTS
    // the sweep's other nine instantiations exist in the experiments build only: the product library launches (256, non-temporal)
    if (chunk == 256) { if (nt) BSX_EX_LAUNCH(256, true); else BSX_EX_LAUNCH(256, false); }
    else if (chunk == 512) { if (nt) BSX_EX_LAUNCH(512, true); else BSX_EX_LAUNCH(512, false); }
    else if (chunk == 2048) { if (nt) BSX_EX_LAUNCH(2048, true); else BSX_EX_LAUNCH(2048, false); }
    else if (chunk == 4096) { if (nt) BSX_EX_LAUNCH(4096, true); else BSX_EX_LAUNCH(4096, false); }
    else { if (nt) BSX_EX_LAUNCH(1024, true); else BSX_EX_LAUNCH(1024, false); }
#else
    (void)nt;
    BSX_EX_LAUNCH(256, true);
#endif
#undef BSX_EX_LAUNCH
    return hipGetLastError();
}
}
