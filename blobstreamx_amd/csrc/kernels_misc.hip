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
                                const uint32_t* target_idx, uint8_t* target_out, uint8_t* hashes_copy) {
    const uint32_t r = blockIdx.x, t = threadIdx.x;   // 32 lanes
    if (r >= n_ranges) return;
    const uint64_t idx = (uint64_t)r * hpr + (target_idx ? (uint64_t)target_idx[r] : ranges[r].end_block - ranges[r].start_block);
    const uint8_t b = hashes[idx * 32 + t];
    ranges[r].end_header_hash[t] = b;
    if (target_out) target_out[(uint64_t)r * 32 + t] = b;   // dense copy for the commit tally / finalize
    if (hashes_copy)                                        // private copy of the range's hashes for a consumer on another stream
        for (uint64_t k = 0; k < hpr; k++) hashes_copy[((uint64_t)r * hpr + k) * 32 + t] = hashes[((uint64_t)r * hpr + k) * 32 + t];
}

}  // namespace bsx

extern "C" {
using namespace bsx;
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
                              const uint32_t* target_idx, uint8_t* target_out, uint8_t* hashes_copy) {
    if (!n_ranges) return hipSuccess;
    hipLaunchKernelGGL(k_fill_end_hash, dim3(n_ranges), dim3(32), 0, s, n_ranges, ranges, hashes, hpr, target_idx, target_out, hashes_copy);
    return hipGetLastError();
}
}
