// kernels_misc.hip — small single-purpose kernels behind the host tier of the C ABI:
//   k_encode_tuple      encode_data_root_tuple               (circuits/builder.rs:82-103)
//   k_data_commitment   get_data_commitment<MAX_LEAVES>      (circuits/builder.rs:105-148) on bare data hashes
//   k_fill_end_hash     ctx.end_header_hash := hash of the target header (header_range.rs:42-55: the output of
//                       builder.skip feeds prove_data_commitment)
#include <hip/hip_runtime.h>

#include "../../include/bsx.h"
#include "sha256.h"

namespace bsx {

__global__ void k_encode_tuple(const uint8_t* data_hash, uint64_t height, uint8_t* out) {
    const int t = threadIdx.x;   // 64 lanes, one output byte each
    uint8_t b = 0;
    if (t >= 24 && t < 32) b = (uint8_t)(height >> (8 * (31 - t)));   // :90,97 U64 EVM encoding = big endian
    if (t >= 32) b = data_hash[t - 32];                               // :98
    out[t] = b;                                                       // :93-96 24 zero bytes first
}

// One workgroup, max_leaves (power of two <= 256) lanes.  out_flags: BSX_A1 / BSX_A2 bits.
__global__ __launch_bounds__(256) void k_data_commitment(const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start_block,
                                                         uint64_t end_block, uint8_t* out_root, uint32_t* out_flags) {
    __shared__ uint32_t nodes[2][256 * 8];
    const uint32_t tid = threadIdx.x, B = max_leaves;
    const bool gte = end_block >= start_block;          // :113
    const uint64_t nb = end_block - start_block;        // :119
    const uint32_t nb_enabled = (uint32_t)nb;           // :124
    if (tid < B) {
        const uint64_t height = start_block + tid;      // :134
        uint32_t t[16];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = 0;
        t[6] = (uint32_t)(height >> 32);
        t[7] = (uint32_t)height;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint8_t* p = data_hashes + 32 * (uint64_t)tid + 4 * k;
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
    if (tid < 8) {
        const uint32_t w = nodes[cur][tid];
        out_root[4 * tid] = (uint8_t)(w >> 24); out_root[4 * tid + 1] = (uint8_t)(w >> 16);
        out_root[4 * tid + 2] = (uint8_t)(w >> 8); out_root[4 * tid + 3] = (uint8_t)w;
    }
    if (tid == 0) *out_flags = (gte ? 0u : BSX_A1_END_GTE_START) | ((nb >> 32) ? BSX_A2_NB_BLOCKS_U32 : 0u);
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
// leaf / inner are the Tendermint Merkle primitives (0x00 / 0x01 prefixes).  One workgroup; digests live in `scratch` (P x 32 B).
__global__ __launch_bounds__(256) void k_commit_fold(const bsx_commit_result* __restrict__ res, uint32_t n, uint32_t first_index, uint32_t P,
                                                     uint32_t* __restrict__ scratch, bsx_commit_fold* __restrict__ out) {
    __shared__ unsigned long long s_ok, s_sigs;
    __shared__ uint32_t s_first;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_ok = 0; s_sigs = 0; s_first = 0xffffffffu; }
    __syncthreads();
    uint64_t ok = 0, sigs = 0;
    uint32_t first = 0xffffffffu;
    for (uint32_t c = tid; c < P; c += 256) {
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
            ok += good ? 1 : 0;
            sigs += r.n_signed - r.n_bad_signature;
            if (!good && first == 0xffffffffu) first = first_index + c;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) scratch[(uint64_t)c * 8 + k] = d.w[k];
    }
    atomicAdd(&s_ok, (unsigned long long)ok);
    atomicAdd(&s_sigs, (unsigned long long)sigs);
    atomicMin(&s_first, first);
    __syncthreads();
    for (uint32_t width = P / 2; width >= 1; width /= 2) {
        // parents are written over the left half of the level (node i <- children 2i, 2i+1): a lane reads both children before
        // the barrier and writes after it, so no lane overwrites a digest another lane still has to read
        Digest nd[8];
        uint32_t cnt = 0;
        for (uint32_t i = tid; i < width; i += 256, cnt++) {
            Digest l, r;
#pragma unroll
            for (int k = 0; k < 8; k++) { l.w[k] = scratch[(uint64_t)(2 * i) * 8 + k]; r.w[k] = scratch[(uint64_t)(2 * i + 1) * 8 + k]; }
            if (cnt < 8) nd[cnt] = inner_hash(l, r);
        }
        __syncthreads();
        cnt = 0;
        for (uint32_t i = tid; i < width; i += 256, cnt++)
            if (cnt < 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) scratch[(uint64_t)i * 8 + k] = nd[cnt].w[k];
            }
        __syncthreads();
    }
    if (tid < 8) {
        const uint32_t w = scratch[tid];
        out->root[4 * tid] = (uint8_t)(w >> 24); out->root[4 * tid + 1] = (uint8_t)(w >> 16);
        out->root[4 * tid + 2] = (uint8_t)(w >> 8); out->root[4 * tid + 3] = (uint8_t)w;
    }
    if (tid == 0) {
        out->n_commits = n;
        out->n_ok = s_ok;
        out->n_signatures_ok = s_sigs;
        out->first_index = first_index;
        out->first_failing = s_first;
        for (int k = 0; k < 16; k++) out->_pad[k] = 0;
    }
}

}  // namespace bsx

extern "C" {
using namespace bsx;
// n <= BSX_COMMIT_FOLD_MAX commits; scratch: bsxk_commit_fold_scratch_bytes(n)
uint64_t bsxk_commit_fold_scratch_bytes(uint32_t n) {
    uint32_t P = 1;
    while (P < n) P *= 2;
    return (uint64_t)P * 32;
}
hipError_t bsxk_commit_fold(hipStream_t s, const bsx_commit_result* res, uint32_t n, uint32_t first_index, void* scratch, bsx_commit_fold* out) {
    uint32_t P = 1;
    while (P < n) P *= 2;
    hipLaunchKernelGGL(k_commit_fold, dim3(1), dim3(256), 0, s, res, n, first_index, P, static_cast<uint32_t*>(scratch), out);
    return hipGetLastError();
}
hipError_t bsxk_encode_tuple(hipStream_t s, const uint8_t* data_hash, uint64_t height, uint8_t* out) {
    hipLaunchKernelGGL(k_encode_tuple, dim3(1), dim3(64), 0, s, data_hash, height, out);
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
