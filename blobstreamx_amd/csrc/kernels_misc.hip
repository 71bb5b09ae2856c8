// kernels_misc.hip — small single-purpose kernels behind the host tier of the C ABI:
//   k_encode_tuple      encode_data_root_tuple               (circuits/builder.rs:82-103)
//   k_data_commitment   get_data_commitment<MAX_LEAVES>      (circuits/builder.rs:105-148) on bare data hashes
//   k_fill_end_hash     ctx.end_header_hash := hash of the target header (header_range.rs:42-55: the output of
//                       builder.skip feeds prove_data_commitment)
#include <hip/hip_runtime.h>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"
#include "kernels.h"
#include "sha256.h"

namespace bsx {

__global__ void k_encode_tuple(const uint8_t* data_hash, uint64_t height, uint8_t* out) {
    const int t = threadIdx.x;   // 64 lanes, one output byte each
    uint8_t b = 0;
    if (t >= 24 && t < 32) b = (uint8_t)(height >> (8 * (31 - t)));   // :90,97 U64 EVM encoding = big endian
    if (t >= 32) b = data_hash[t - 32];                               // :98
    out[t] = b;                                                       // :93-96 24 zero bytes first
}

// get_data_commitment (builder.rs:105-148) by one workgroup: lane tid < B holds leaf tid's data hash at `dh` (32 bytes, any alignment);
// returns through nodes[cur][0..8) — the caller reads the root after the call.  B = power of two <= 256.
__device__ __forceinline__ int data_commitment_tree(uint32_t (*nodes)[256 * 8], const uint8_t* dh, uint32_t B, uint64_t start_block, uint32_t nb_enabled) {
    const uint32_t tid = threadIdx.x;
    if (tid < B) {
        const uint64_t height = start_block + tid;      // :134
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = 0;
        t[6] = (uint32_t)(height >> 32);
        t[7] = (uint32_t)height;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint8_t* p = dh + 4 * k;
            t[8 + k] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        }
        const Digest lh = leaf_hash_tuple(t);
#pragma unroll
        for (int k = 0; k < 8; k++) nodes[0][tid * 8 + k] = lh.w[k];
    }
    __syncthreads();
    int cur = 0;
    for (uint32_t width = B / 2, span = 2; width >= 1; width /= 2, span *= 2) {
        if (tid < width) {
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = nodes[cur][(2 * tid) * 8 + k]; r.w[k] = nodes[cur][(2 * tid + 1) * 8 + k]; }
            const Digest in = inner_hash(l, r);
            const bool en_l = tid * span < nb_enabled, en_r = tid * span + span / 2 < nb_enabled;
            const Digest node = (en_l && en_r) ? in : l;
#pragma unroll
            for (int k = 0; k < 8; k++) nodes[cur ^ 1][tid * 8 + k] = node.w[k];
        }
        __syncthreads();
        cur ^= 1;
    }
    return cur;
}

// One workgroup, max_leaves (power of two <= 256) lanes.  out_flags: BSX_A1 / BSX_A2 bits.
__global__ __launch_bounds__(256) void k_data_commitment(const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start_block,
                                                         uint64_t end_block, uint8_t* out_root, uint32_t* out_flags) {
    __shared__ uint32_t nodes[2][256 * 8];
    const uint32_t tid = threadIdx.x, B = max_leaves;
    const bool gte = end_block >= start_block;          // :113
    const uint64_t nb = end_block - start_block;        // :119
    const uint32_t nb_enabled = (uint32_t)nb;           // :124
    const int cur = data_commitment_tree(nodes, data_hashes + 32 * (uint64_t)(tid < B ? tid : 0), B, start_block, nb_enabled);
    if (tid < 8) {
        const uint32_t w = nodes[cur][tid];
        out_root[4 * tid] = (uint8_t)(w >> 24); out_root[4 * tid + 1] = (uint8_t)(w >> 16);
        out_root[4 * tid + 2] = (uint8_t)(w >> 8); out_root[4 * tid + 3] = (uint8_t)w;
    }
    if (tid == 0) *out_flags = (gte ? 0u : BSX_A1_END_GTE_START) | ((nb >> 32) ? BSX_A2_NB_BLOCKS_U32 : 0u);
}

// The hint's expected_data_commitment for n coalesced requests (input.rs:241-244; zero for an empty range, :70-72): one workgroup
// per request, data hashes read from the request's assembled compact image (data_hash_proofs[i].leaf[2..34)).
__global__ __launch_bounds__(256) void k_expected_commitments(uint32_t B, const bsx_shared_ctx* ranges, const uint32_t* spans, const uint64_t* latest,
                                                              const uint32_t* jobs, const uint8_t* compact, uint32_t compact_stride, uint8_t* out) {
    __shared__ uint32_t nodes[2][256 * 8];
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    const uint64_t start = ranges[r].start_block + (jobs ? (uint64_t)jobs[r] * B : 0), end = start + spans[r];
    const uint64_t lim = latest[r] - 2;                                      // input.rs:160-162
    const uint64_t req_end = end < lim ? end : lim;
    const uint8_t* cw = compact + (uint64_t)r * compact_stride;
    const bool empty = req_end <= start;                                     // block-uniform
    const int cur = data_commitment_tree(nodes, cw + bsx_off_dh_proofs(B) + BSX_DH_PROOF_SIZE * (tid < B ? tid : 0) + 128 + 2, B, start,
                                         empty ? 0u : (uint32_t)(req_end - start));
    if (tid < 8) {
        const uint32_t w = empty ? 0u : nodes[cur][tid];
        uint8_t* o = out + 32 * (uint64_t)r;
        o[4 * tid] = (uint8_t)(w >> 24); o[4 * tid + 1] = (uint8_t)(w >> 16); o[4 * tid + 2] = (uint8_t)(w >> 8); o[4 * tid + 3] = (uint8_t)w;
    }
}

__global__ void k_fill_end_hash(uint32_t n_ranges, bsx_shared_ctx* ranges, const uint8_t* hashes, uint64_t hpr,
                                const uint32_t* target_idx, uint8_t* target_out, uint8_t* hashes_copy, uint64_t first_rel) {
    const uint32_t r = blockIdx.x, t = threadIdx.x;   // 32 lanes
    if (r >= n_ranges) return;
    // first_rel: the range's header block starts at height S + first_rel (a rank that holds only its job slice): ranges whose
    // target header lies outside the block keep their end hash (no slot of this rank's slice can be the range's last block)
    const uint64_t rel = target_idx ? (uint64_t)target_idx[r] : ranges[r].end_block - ranges[r].start_block - first_rel;
    if (!target_idx && (ranges[r].end_block - ranges[r].start_block < first_rel || rel >= hpr)) return;
    const uint64_t idx = (uint64_t)r * hpr + rel;
    const uint8_t b = hashes[idx * 32 + t];
    ranges[r].end_header_hash[t] = b;
    if (target_out) target_out[(uint64_t)r * 32 + t] = b;   // dense copy for the commit tally / finalize
    if (hashes_copy)                                        // private copy of the range's hashes for a consumer on another stream
        for (uint64_t k = 0; k < hpr; k++) hashes_copy[((uint64_t)r * hpr + k) * 32 + t] = hashes[((uint64_t)r * hpr + k) * 32 + t];
}

// ------------------------------------------------------------------------------------------------ k_commit_fold
// Mode S across GPUs ("in S mode commits shard with their headers"; next_header.rs:25-47 per header): a rank's
// slice of per-header commit results folded into ONE 128-byte record — the unit of the mode-S all-gather.
//   digest(c) = inner( leaf(result[c] bytes 0..64), leaf(result[c] bytes 64..84 ‖ u32 LE commit index ‖ 40 zero bytes) )
//   root      = binary SHA-256 tree (RFC 6962 inner nodes) over digest(0..n) padded with all-zero digests to a power of two
// leaf / inner are the Tendermint Merkle primitives (0x00 / 0x01 prefixes).
// Round 5: the per-commit digests (6 compressions each) and the tree are spread over one lane per commit — workgroups of 256 commits
// fold their own sub-tree in LDS (k_commit_fold_parts: 6 + 16 dependent compressions), one wave joins the <= 8 sub-roots
// (k_commit_fold_top: 6 more).  Round 3's single workgroup hashed 8 commits per lane and the wide levels in turns: 78 dependent
// compressions = 0.26 ms of one-wave latency behind every mode-S step.
struct FoldPart { uint32_t root[8]; unsigned long long ok, sigs; uint32_t first, pad[3]; };      // 64 bytes
static_assert(sizeof(FoldPart) == 64, "FoldPart");

__device__ __forceinline__ void fold_write(bsx_commit_fold* out, const uint32_t root[8], uint32_t n, uint32_t first_index, unsigned long long ok,
                                           unsigned long long sigs, uint32_t first) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t w = root[k];
        out->root[4 * k] = (uint8_t)(w >> 24); out->root[4 * k + 1] = (uint8_t)(w >> 16);
        out->root[4 * k + 2] = (uint8_t)(w >> 8); out->root[4 * k + 3] = (uint8_t)w;
    }
    out->n_commits = n;
    out->n_ok = ok;
    out->n_signatures_ok = sigs;
    out->first_index = first_index;
    out->first_failing = first;
    for (int k = 0; k < 16; k++) out->_pad[k] = 0;
}

// Pb = digests per workgroup (a power of two <= 256; P / gridDim.x).  gridDim.x == 1: writes the fold record itself.
__global__ __launch_bounds__(256) void k_commit_fold_parts(const bsx_commit_result* __restrict__ res, uint32_t n, uint32_t first_index, uint32_t Pb,
                                                           FoldPart* __restrict__ parts, bsx_commit_fold* __restrict__ out) {
    __shared__ uint32_t nodes[2][256 * 8];
    __shared__ unsigned long long s_ok, s_sigs;
    __shared__ uint32_t s_first;
    const uint32_t tid = threadIdx.x, c = blockIdx.x * Pb + tid;
    if (tid == 0) { s_ok = 0; s_sigs = 0; s_first = 0xffffffffu; }
    __syncthreads();
    if (tid < Pb) {
        Digest d;
#pragma unroll
        for (int k = 0; k < 8; k++) d.w[k] = 0;
        if (c < n) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(res + c);
            uint32_t t[16];
#pragma unroll
            for (int k = 0; k < 16; k++) t[k] = bswap32(p[k]);          // bytes 0..64 as big-endian message words
            const Digest a = leaf_hash_tuple(t);
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = bswap32(p[16 + k]);      // n_bad_signature .. power_overflow
            t[5] = bswap32(first_index + c);
#pragma unroll
            for (int k = 6; k < 16; k++) t[k] = 0;
            const Digest b = leaf_hash_tuple(t);
            d = inner_hash(a, b);
            const bsx_commit_result& r = res[c];
            const bool good = r.two_thirds_ok && !r.n_bad_signature && !r.n_bad_message && !r.power_overflow;
            atomicAdd(&s_ok, (unsigned long long)(good ? 1 : 0));
            atomicAdd(&s_sigs, (unsigned long long)(r.n_signed - r.n_bad_signature));
            if (!good) atomicMin(&s_first, first_index + c);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) nodes[0][tid * 8 + k] = d.w[k];
    }
    __syncthreads();
    int cur = 0;
    for (uint32_t width = Pb / 2; width >= 1; width /= 2) {
        if (tid < width) {
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = nodes[cur][(2 * tid) * 8 + k]; r.w[k] = nodes[cur][(2 * tid + 1) * 8 + k]; }
            const Digest nd = inner_hash(l, r);
#pragma unroll
            for (int k = 0; k < 8; k++) nodes[cur ^ 1][tid * 8 + k] = nd.w[k];
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid == 0) {
        uint32_t root[8];
#pragma unroll
        for (int k = 0; k < 8; k++) root[k] = nodes[cur][k];
        if (gridDim.x == 1) {
            fold_write(out, root, n, first_index, s_ok, s_sigs, s_first);
        } else {
            FoldPart* o = parts + blockIdx.x;
#pragma unroll
            for (int k = 0; k < 8; k++) o->root[k] = root[k];
            o->ok = s_ok; o->sigs = s_sigs; o->first = s_first; o->pad[0] = o->pad[1] = o->pad[2] = 0;
        }
    }
}

// nb = 2, 4 or 8 sub-roots -> the record.  One wave.
__global__ __launch_bounds__(64) void k_commit_fold_top(const FoldPart* __restrict__ parts, uint32_t nb, uint32_t n, uint32_t first_index,
                                                        bsx_commit_fold* __restrict__ out) {
    __shared__ uint32_t nodes[2][8 * 8];
    const uint32_t tid = threadIdx.x;
    if (tid < nb * 8) nodes[0][tid] = parts[tid >> 3].root[tid & 7];
    __syncthreads();
    int cur = 0;
    for (uint32_t width = nb / 2; width >= 1; width /= 2) {
        if (tid < width) {
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = nodes[cur][(2 * tid) * 8 + k]; r.w[k] = nodes[cur][(2 * tid + 1) * 8 + k]; }
            const Digest nd = inner_hash(l, r);
#pragma unroll
            for (int k = 0; k < 8; k++) nodes[cur ^ 1][tid * 8 + k] = nd.w[k];
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid == 0) {
        unsigned long long ok = 0, sigs = 0;
        uint32_t first = 0xffffffffu;
        for (uint32_t b = 0; b < nb; b++) { ok += parts[b].ok; sigs += parts[b].sigs; first = parts[b].first < first ? parts[b].first : first; }
        uint32_t root[8];
#pragma unroll
        for (int k = 0; k < 8; k++) root[k] = nodes[cur][k];
        fold_write(out, root, n, first_index, ok, sigs, first);
    }
}


// ------------------------------------------------------------------------------------------------ k_field_proofs
// Header-field inclusion proofs of the SKIP / STEP units (include/bsx_layout.h; tendermintx ChainIdProofVariable / HeightProofVariable
// / HashInclusionProofVariable [UPSTREAM], the hint outputs builder.skip / builder.step verify against the header hash,
// header_range.rs:42-48, next_header.rs:32-36): one wave per (item, header) hashes the header's 14-leaf tree (tendermint
// Header::hash: 14 = 8 | 6, 6 = 4 | 2) level by level through LDS and copies out, for every requested field index (<= 11: depth 4),
// the 4 aunts, the 5 path digests (leaf hash .. header hash), the zero-padded leaf and its length.  Independent of the signatures:
// runs in the commit check's prep phase.  Latency only (5 dependent hashing steps).
__device__ __forceinline__ void fp_field(uint32_t i, uint32_t& off, uint32_t& cap) {   // byte offset / capacity of field i in bsx_header
    if (i == 0) { off = 16; cap = 24; }
    else if (i == 1) { off = 40; cap = 52; }
    else if (i == 2) { off = 92; cap = 12; }
    else if (i == 3) { off = 104; cap = 20; }
    else if (i == 4) { off = 124; cap = 76; }
    else if (i <= 12) { off = 200 + 36 * (i - 5); cap = 36; }
    else { off = 488; cap = 24; }
}
__global__ __launch_bounds__(64) void k_field_proofs(bsxk_field_proofs_args a) {
    __shared__ uint32_t lv[4][16][8];      // level k (0 = leaf hashes): node j
    __shared__ uint32_t root_w[8];
    const uint32_t item = blockIdx.x >> 1, hsel = blockIdx.x & 1, lane = threadIdx.x;
    bool any = false;
    for (uint32_t q = 0; q < a.n_proofs; q++) any = any || a.proofs[q].header == hsel;
    if (!any) return;                      // block-uniform
    uint64_t hidx = 0;
    if (hsel) hidx = a.target_idx ? (uint64_t)a.target_idx[item] : a.ranges ? a.ranges[item].end_block - a.ranges[item].start_block : 1;
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(a.headers + (uint64_t)item * a.headers_per_item + hidx);
    if (lane < 14) {
        uint32_t off, cap;
        fp_field(lane, off, cap);
        uint32_t len = (rec[lane >> 2] >> (8 * (lane & 3))) & 0xffu;
        if (len > cap) len = cap;
        uint32_t d[19];
#pragma unroll
        for (uint32_t j = 0; j < 19; j++) d[j] = 4 * j < cap ? rec[off / 4 + j] : 0u;
        const Digest h = len <= 54 ? leaf_hash_1block(d, (int)len) : leaf_hash_2block(d, (int)len);
#pragma unroll
        for (int k = 0; k < 8; k++) lv[0][lane][k] = h.w[k];
    }
    __syncthreads();
    // level 1: 7 pairs; level 2: 3 pairs + the odd node (leaves 12,13) passed up; level 3: 2 pairs; root
    for (uint32_t level = 1; level <= 3; level++) {
        const uint32_t pairs = level == 1 ? 7 : level == 2 ? 3 : 2;
        if (lane < pairs) {
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = lv[level - 1][2 * lane][k]; r.w[k] = lv[level - 1][2 * lane + 1][k]; }
            const Digest h = inner_hash(l, r);
#pragma unroll
            for (int k = 0; k < 8; k++) lv[level][lane][k] = h.w[k];
        } else if (level == 2 && lane == 3) {
#pragma unroll
            for (int k = 0; k < 8; k++) lv[2][3][k] = lv[1][6][k];
        }
        __syncthreads();
    }
    if (lane == 0) {
        Digest l, r;
#pragma unroll
        for (int k = 0; k < 8; k++) { l.w[k] = lv[3][0][k]; r.w[k] = lv[3][1][k]; }
        const Digest h = inner_hash(l, r);
#pragma unroll
        for (int k = 0; k < 8; k++) root_w[k] = h.w[k];
    }
    __syncthreads();
    uint8_t* cw = a.unit.base + (uint64_t)item * a.unit.stride;
    uint32_t* W = reinterpret_cast<uint32_t*>(cw + a.unit.off_words);
    const uint32_t fl = a.flags ? a.flags[item] : 0u;
    for (uint32_t q = 0; q < a.n_proofs; q++) {
        const bsxk_proof_spec sp = a.proofs[q];
        if (sp.header != hsel) continue;
        uint32_t* dst = reinterpret_cast<uint32_t*>(cw + sp.off);
        const bool zero = (fl & sp.zero_if) != 0;
        const uint32_t idx = sp.field;
        uint32_t off, cap;
        fp_field(idx, off, cap);
        uint32_t len = (rec[idx >> 2] >> (8 * (idx & 3))) & 0xffu;
        if (len > cap) len = cap;
        if (lane < 32) {                                   // aunts: level k's sibling of the path node
            const uint32_t k = lane >> 3, w = lane & 7;
            dst[lane] = zero ? 0u : bswap32(lv[k][(idx >> k) ^ 1][w]);
        }
        if (lane < 40) {                                   // path: leaf hash, the 4 inner nodes (last = header hash)
            const uint32_t k = lane >> 3, w = lane & 7;
            const uint32_t v = k < 4 ? lv[k][idx >> k][w] : root_w[w];
            dst[32 + lane] = zero ? reinterpret_cast<const uint32_t*>(a.zero_paths)[lane] : bswap32(v);
        }
        if (4 * lane < sp.cap) {                           // leaf, zero padded to the record's capacity
            uint32_t v = 0;
            if (!zero && 4 * lane < cap && 4 * lane < len) {
                v = rec[off / 4 + lane];
                const uint32_t r = len - 4 * lane;
                if (r < 4) v &= 0xffffffffu >> (32 - 8 * r);
            }
            dst[BSX_PROOF_FIXED / 4 + lane] = v;
        }
        if (lane == 0) W[sp.len_word] = zero ? BSX_PROTOBUF_HASH_SIZE : len;
    }
}

// ------------------------------------------------------------------------------------------------ k_step_check
// Step conditions of CombinedStepCircuit::define (next_header.rs:25-46; builder.step [UPSTREAM] tendermintx v1.0.0)
// and prove_next_header_data_commitment (builder.rs:411-443) for one request: one wave.  Status = the FIRST failing condition in the
// order the oracle uses: power overflow, prev hash, height, chain id, signatures, validators_hash, next_validators_hash,
// last_block_id, 2/3.  Reads data_hash_proofs[0] (leaf + path) from the STEP unit, where k_field_proofs left it.
__global__ __launch_bounds__(64) void k_step_check(bsxk_step_args a) {
    const uint32_t tid = threadIdx.x, q = tid & 31;
    const bsx_header* ph = a.headers;
    const bsx_header* nh = a.headers + 1;
    const bsx_commit_result* cr = a.commit;
    uint8_t* cw = a.unit.base;
    uint32_t* W = reinterpret_cast<uint32_t*>(cw + a.unit.off_words);
    uint8_t* Bo = cw + a.unit.off_bools;
    uint64_t prev_block = 0;                                               // next_header.rs:26 (big endian)
    for (int i = 0; i < 8; i++) prev_block = prev_block << 8 | a.input40[i];
    const uint64_t next_block = prev_block + 1;                            // :29-30
    const uint8_t b_prev_in = a.input40[8 + q];                            // :27
    const uint8_t b_hp = a.hashes[q], b_hn = a.hashes[32 + q];
    const uint8_t b_vh = nh->hash[2][2 + q], b_nvh = ph->hash[3][2 + q], b_cvh = cr->validators_hash[q];
    const uint8_t b_lb = nh->last_block_id[2 + q];
    const uint8_t b_height = nh->height[q < 12 ? q : 0];
    const uint8_t b_chain = nh->chain_id[tid < 52 ? tid : 0];
    const uint8_t l_chain = nh->len[1], l_height = nh->len[BSX_BLOCK_HEIGHT_INDEX], l_vh = nh->len[7], l_nvh = ph->len[8], l_lb = nh->len[BSX_LAST_BLOCK_ID_INDEX];
    const uint8_t* dhp = cw + bsx_st_off_proof(5);
    const uint8_t b_dh_root = dhp[128 + 128 + q];                          // path[4] of data_hash_proofs[0]
    int hn = 1;
    uint8_t want = 0x08;
    {
        uint64_t hv = next_block;
        int pos = 1;
        for (;;) {
            const bool more = hv >= 0x80;
            const uint8_t byte = (uint8_t)(more ? (hv | 0x80) : hv);
            if ((int)q == pos) want = byte;
            pos++;
            if (!more) break;
            hv >>= 7;
        }
        hn = pos;
    }
    const bool c_prev = __ballot(b_hp != b_prev_in) == 0;
    const bool c_height = (l_height == hn) && __ballot((int)q < hn && b_height != want) == 0;
    const uint32_t cl = a.chain_id_len;
    const uint8_t want_c = tid == 0 ? 0x0a : tid == 1 ? (uint8_t)cl : a.chain_id[tid >= 2 && tid < 52 ? tid - 2 : 0];
    const bool c_chain = (l_chain == cl + 2) && __ballot(tid < cl + 2 && b_chain != want_c) == 0;
    const bool c_vh = (l_vh == 34) && __ballot(b_vh != b_cvh) == 0;
    const bool c_nvh = (l_nvh == 34) && __ballot(b_nvh != b_cvh) == 0;
    const bool c_lb = (l_lb >= 34) && __ballot(b_lb != b_hp) == 0;
    const bool c_a10 = __ballot(b_dh_root != b_prev_in) == 0;               // builder.rs:434
    if (tid < 32) { cw[q] = b_prev_in; cw[32 + q] = b_hn; a.output64[q] = b_hn; }
    if (tid == 0) {
        const bool c_sigs = !(cr->n_bad_signature || cr->n_bad_message);
        const bool two_thirds = cr->two_thirds_ok != 0, overflow = cr->power_overflow != 0;
        uint32_t st = BSX_OK;
        if (overflow) st = BSX_ERR_BAD_ARG;
        if (!st && !c_prev) st = BSX_ERR_ASSERT;
        if (!st && !c_height) st = BSX_ERR_ASSERT;
        if (!st && !c_chain) st = BSX_ERR_ASSERT;
        if (!st && !c_sigs) st = BSX_ERR_BAD_SIGNATURE;
        if (!st && !c_vh) st = BSX_ERR_ASSERT;
        if (!st && !c_nvh) st = BSX_ERR_ASSERT;
        if (!st && !c_lb) st = BSX_ERR_ASSERT;
        if (!st && !two_thirds) st = BSX_ERR_VOTING_POWER;
        // which condition failed first, for the host's message: bits 8.. = 1 + index into the list above
        uint32_t why = 0;
        if (overflow) why = 1; else if (!c_prev) why = 2; else if (!c_height) why = 3; else if (!c_chain) why = 4; else if (!c_sigs) why = 5;
        else if (!c_vh) why = 6; else if (!c_nvh) why = 7; else if (!c_lb) why = 8; else if (!two_thirds) why = 9;
        *a.step_status = st | (why << 8);
        *a.dc_status = c_a10 ? 0u : BSX_A10_NEXT_HEADER;
        // data-root tuple (builder.rs:436-439) and its leaf hash (:442)
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = 0;
        t[6] = (uint32_t)(prev_block >> 32);
        t[7] = (uint32_t)prev_block;
        const uint8_t* lf = dhp + BSX_PROOF_FIXED + 2;
#pragma unroll
        for (int k = 0; k < 8; k++) t[8 + k] = ((uint32_t)lf[4 * k] << 24) | ((uint32_t)lf[4 * k + 1] << 16) | ((uint32_t)lf[4 * k + 2] << 8) | lf[4 * k + 3];
        const Digest dc = leaf_hash_tuple(t);
        uint32_t* tp = reinterpret_cast<uint32_t*>(cw + bsx_st_off_tuple());
#pragma unroll
        for (int k = 0; k < 16; k++) tp[k] = bswap32(t[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            reinterpret_cast<uint32_t*>(cw + 64)[k] = bswap32(dc.w[k]);
            reinterpret_cast<uint32_t*>(a.output64 + 32)[k] = bswap32(dc.w[k]);
        }
        W[BSX_ST_W_PREV_BLOCK] = (uint32_t)prev_block; W[BSX_ST_W_PREV_BLOCK + 1] = (uint32_t)(prev_block >> 32);
        W[BSX_ST_W_NEXT_BLOCK] = (uint32_t)next_block; W[BSX_ST_W_NEXT_BLOCK + 1] = (uint32_t)(next_block >> 32);
        Bo[0] = c_prev; Bo[1] = c_height; Bo[2] = c_chain; Bo[3] = c_sigs; Bo[4] = c_vh; Bo[5] = c_nvh; Bo[6] = c_lb; Bo[7] = two_thirds;
        Bo[8] = overflow; Bo[9] = c_a10;
    }
}

// ---------------------------------------------------------------------------------------------------------------- packed wire headers
// A header on the wire is ~394 bytes of protobuf fields; the 512-byte bsx_header record pads every field to a fixed,
// 4-byte aligned capacity so that the hashing kernels stage records with 16-byte loads — 23 % of an upload is padding.  Hosts may hand
// the coalescing front end PACKED headers instead (include/bsx.h: bsx_pack_headers: len[14] then the fields back to back); the bytes
// cross PCIe packed and this kernel lays them out as records in HBM, where the padding is free.  One wave per header, 8 output bytes
// per lane; slots that did not come packed are left alone.  desc[2 r] = byte offset of slot r's block in `packed` (0xffffffff: the slot
// was uploaded as records), desc[2 r + 1] = headers in it; a block = u32 off[n + 1] (relative to its 16-aligned data), then the data.
static constexpr uint32_t kFieldAt[BSX_HEADER_FIELDS + 1] = {16, 40, 92, 104, 124, 200, 236, 272, 308, 344, 380, 416, 452, 488, 512};   // offsetof each field of bsx_header
__global__ __launch_bounds__(256) void k_unpack_headers(const uint8_t* __restrict__ packed, const uint32_t* __restrict__ desc, uint32_t n_slots, uint32_t hpr,
                                                        const uint32_t* __restrict__ wipe_to, bsx_header* __restrict__ out) {
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = wave / hpr, h = wave % hpr;
    if (slot >= n_slots) return;
    const uint32_t boff = desc[2 * slot], n = desc[2 * slot + 1];
    if (boff == 0xffffffffu) return;
    uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out + (size_t)slot * hpr + h)) + lane;
    if (h >= n) {                                            // behind the request's headers: zero records, as far as an earlier request reached
        if (h < wipe_to[slot]) *dst = make_uint2(0u, 0u);
        return;
    }
    const uint32_t* off = reinterpret_cast<const uint32_t*>(packed + boff);
    const uint8_t* data = packed + boff + ((4u * (n + 1u) + 15u) & ~15u);
    const uint32_t b0 = off[h], b1 = off[h + 1];             // validated on the host at submit: monotone, inside the block, >= 14 apart
    const uint8_t* src = data + b0;
    uint32_t len[BSX_HEADER_FIELDS], pre[BSX_HEADER_FIELDS];
    uint32_t acc = BSX_HEADER_FIELDS;
#pragma unroll
    for (int f = 0; f < BSX_HEADER_FIELDS; f++) { len[f] = src[f]; pre[f] = acc; acc += len[f]; }
    const uint32_t avail = b1 - b0;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t p = lane * 8u + k;
        uint32_t byte = 0;
        if (p < BSX_HEADER_FIELDS) byte = src[p];            // the length bytes as they came: a length over its field's capacity is the
        else if (p >= 16) {                                  // hashing kernel's to flag (BSX_ERR_BAD_HEADER), not ours to hide
            uint32_t lf = len[0], pf = pre[0], at0 = kFieldAt[0];
#pragma unroll
            for (int q = 1; q < BSX_HEADER_FIELDS; q++)
                if (p >= kFieldAt[q]) { lf = len[q]; pf = pre[q]; at0 = kFieldAt[q]; }
            const uint32_t i = p - at0, at = pf + i;         // i < the field's capacity by construction: an over-long field cannot spill
            if (i < lf && at < avail) byte = src[at];
        }
        w[k >> 2] |= byte << (8 * (k & 3));
    }
    *dst = make_uint2(w[0], w[1]);
}

}  // namespace bsx

extern "C" {
using namespace bsx;
hipError_t bsxk_field_proofs(hipStream_t s, const bsxk_field_proofs_args* a) {
    if (!a->n_items || !a->n_proofs) return hipSuccess;
    hipLaunchKernelGGL(k_field_proofs, dim3(2 * a->n_items), dim3(64), 0, s, *a);
    return hipGetLastError();
}
hipError_t bsxk_step_check(hipStream_t s, const bsxk_step_args* a) {
    hipLaunchKernelGGL(k_step_check, dim3(1), dim3(64), 0, s, *a);
    return hipGetLastError();
}
// n <= BSX_COMMIT_FOLD_MAX commits; scratch: bsxk_commit_fold_scratch_bytes(n)
uint64_t bsxk_commit_fold_scratch_bytes(uint32_t n) {
    uint32_t P = 1;
    while (P < n) P *= 2;
    return (uint64_t)P * 32;
}
hipError_t bsxk_unpack_headers(hipStream_t s, const uint8_t* packed, const uint32_t* desc, uint32_t n_slots, uint32_t hpr, const uint32_t* wipe_to,
                               bsx_header* out) {
    if (!n_slots) return hipSuccess;
    const uint64_t waves = (uint64_t)n_slots * hpr;
    hipLaunchKernelGGL(k_unpack_headers, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, packed, desc, n_slots, hpr, wipe_to, out);
    return hipGetLastError();
}
hipError_t bsxk_commit_fold(hipStream_t s, const bsx_commit_result* res, uint32_t n, uint32_t first_index, void* scratch, bsx_commit_fold* out) {
    uint32_t P = 1;
    while (P < n) P *= 2;
    const uint32_t nb = P > 256 ? P / 256 : 1, Pb = P / nb;
    FoldPart* parts = static_cast<FoldPart*>(scratch);
    hipLaunchKernelGGL(k_commit_fold_parts, dim3(nb), dim3(256), 0, s, res, n, first_index, Pb, parts, out);
    if (nb > 1) hipLaunchKernelGGL(k_commit_fold_top, dim3(1), dim3(64), 0, s, parts, nb, n, first_index, out);
    return hipGetLastError();
}
hipError_t bsxk_encode_tuple(hipStream_t s, const uint8_t* data_hash, uint64_t height, uint8_t* out) {
    hipLaunchKernelGGL(k_encode_tuple, dim3(1), dim3(64), 0, s, data_hash, height, out);
    return hipGetLastError();
}
hipError_t bsxk_expected_commitments(hipStream_t s, uint32_t n, uint32_t B, const bsx_shared_ctx* ranges, const uint32_t* spans, const uint64_t* latest,
                                     const uint32_t* jobs, const uint8_t* compact, uint8_t* out) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_expected_commitments, dim3(n), dim3(256), 0, s, B, ranges, spans, latest, jobs, compact, bsx_map_layout(B).compact_stride, out);
    return hipGetLastError();
}
hipError_t bsxk_data_commitment(hipStream_t s, const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start, uint64_t end,
                                uint8_t* out_root, uint32_t* out_flags) {
    hipLaunchKernelGGL(k_data_commitment, dim3(1), dim3(256), 0, s, data_hashes, max_leaves, start, end, out_root, out_flags);
    return hipGetLastError();
}
hipError_t bsxk_fill_end_hash(hipStream_t s, uint32_t n_ranges, bsx_shared_ctx* ranges, const uint8_t* hashes, uint64_t hpr,
                              const uint32_t* target_idx, uint8_t* target_out, uint8_t* hashes_copy, uint64_t first_rel) {
    if (!n_ranges) return hipSuccess;
    hipLaunchKernelGGL(k_fill_end_hash, dim3(n_ranges), dim3(32), 0, s, n_ranges, ranges, hashes, hpr, target_idx, target_out, hashes_copy, first_rel);
    return hipGetLastError();
}
}

#ifdef BSX_EXPERIMENTS
// last launch form per kernel family (kernels.h BSX_NOTE_FORM): experiments build only, read by the variant tests
#include <atomic>
static std::atomic<uint32_t> g_last_form[4];
extern "C" void bsxk_debug_note_form(uint32_t which, uint32_t form) { if (which < 4) g_last_form[which].store(form, std::memory_order_relaxed); }
extern "C" uint32_t bsx_debug_last_launch_form(uint32_t which) { return which < 4 ? g_last_form[which].load(std::memory_order_relaxed) : 0u; }
#endif
