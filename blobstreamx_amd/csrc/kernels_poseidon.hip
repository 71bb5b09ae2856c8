// kernels_poseidon.hip — Poseidon over Goldilocks on gfx950 (P11).
//
//   k_poseidon_permute        one lane per 12-word state (the permutation itself: KATs, host tier)
//   k_leaf_hashes<FUSED>      Merkle leaf digests (PoseidonHash::hash_or_noop) of a witness matrix, one lane per leaf:
//                             FUSED = false  rows of a MATERIALISED Goldilocks witness (u64 elements in HBM)
//                             FUSED = true   rows of the witness generated ON THE FLY from the COMPACT bytes: the 64x
//                                            expanded image (8 B per bit) is never written or read — the expansion
//                                            (P10, k_expand_witness' job) is fused into its consumer
//   k_copy_zero_leaves        the all-zero padding rows of every tree share one digest: computed once (an extra lane of
//                             k_leaf_hashes), copied
//   k_merkle_level            PoseidonHash::two_to_one over one tree level of all trees, one lane per parent
//
// What it replaces: plonky2 `Poseidon::poseidon`, `hash_n_to_hash_no_pad`, `PoseidonHash::{hash_or_noop,two_to_one}`,
// `MerkleTree::new(leaves, cap_height)` [UPSTREAM plonky2 53c5bc3e, Cargo.lock:3110-3112] as reached from
// builder.build()/prove and the mapreduce recursion (circuits/builder.rs:301-302) under PoseidonGoldilocksConfig
// (bin/header_range_2048.rs:1-17).  Integer-ALU bound (one permutation = ~470 field multiplications + 30 MDS layers per
// 64 B absorbed): no LDS, no barriers, no MFMA; bytes are irrelevant (the fused kernel reads 1 compact byte per
// permutation), the roofline is field multiplications/s (tools/microbench.hip).
#include <hip/hip_runtime.h>

#include "../../include/bsx.h"
#include "poseidon.h"
#include "poseidon_consts.h"

namespace bsx {

__constant__ uint64_t POSEIDON_RC[BSX_POSEIDON_TABLE_N] = {BSX_POSEIDON_TABLE};

constexpr int PS_THREADS = 256;

__global__ __launch_bounds__(PS_THREADS) void k_poseidon_permute(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t me = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    if (me >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canonical(in[me * 12 + i]);
    poseidon_permute(s, POSEIDON_RC);
#pragma unroll
    for (int i = 0; i < 12; i++) out[me * 12 + i] = gl_canonical(s[i]);
}

struct LeafArgs {
    bsx_witness_layout lay;        // FUSED: compact layout of one job; else only n_elements is used
    uint32_t n_trees, leaf_len, n_leaves, noop_small;
    uint64_t n_rows;               // rows that hold at least one element: ceil(n_elements / leaf_len)
    uint64_t tree_stride;          // u64 words between consecutive trees in `tree`
    const uint8_t* compact;        // FUSED
    const uint64_t* elements;      // !FUSED: n_trees * n_elements
    uint64_t* tree;
};

// element e of job `job` straight from the compact witness (same mapping as k_expand_witness / bsx.h):
// bytes -> 8 bits MSB first, then the u32 words, then the bools
__device__ __forceinline__ uint64_t compact_elem(const bsx_witness_layout& lay, const uint8_t* c, uint64_t e) {
    const uint64_t nbits = 8ull * lay.n_bytes;
    if (e < nbits) return (c[e >> 3] >> (7 - (e & 7))) & 1u;
    e -= nbits;
    if (e < lay.n_words) return reinterpret_cast<const uint32_t*>(c + lay.off_words)[e];
    e -= lay.n_words;
    return c[lay.off_bools + e];
}

#ifndef BSX_LEAF_WAVES
#define BSX_LEAF_WAVES 1
#endif
template <bool FUSED>
__global__ __launch_bounds__(PS_THREADS, BSX_LEAF_WAVES) void k_leaf_hashes(LeafArgs a) {
    const uint64_t g = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    const uint64_t n_real = (uint64_t)a.n_trees * a.n_rows;
    // one extra lane hashes the all-zero row (the padding rows [n_rows, n_leaves) of every tree share that digest) into slot
    // (tree 0, row n_rows): as one more lane of this launch it costs nothing, as a launch of its own it was 1.3 ms of pure
    // single-lane latency (17 dependent permutations)
    const bool zero_lane = a.n_rows < a.n_leaves && g == n_real;
    if (g >= n_real && !zero_lane) return;
    const uint64_t job = zero_lane ? 0 : g / a.n_rows, j = zero_lane ? a.n_rows : g % a.n_rows;
    const uint64_t nel = zero_lane ? 0 : a.lay.n_elements, e0 = j * a.leaf_len;
    uint64_t d[4];
    if (FUSED) {
        const uint8_t* c = a.compact + job * a.lay.compact_stride;
        const bsx_witness_layout lay = a.lay;
        auto get = [&](uint64_t k) -> uint64_t { const uint64_t e = e0 + k; return e < nel ? compact_elem(lay, c, e) : 0ull; };
        if (a.noop_small) poseidon_hash_or_noop(get, a.leaf_len, POSEIDON_RC, d);
        else poseidon_hash_no_pad(get, a.leaf_len, POSEIDON_RC, d);
    } else {
        const uint64_t* el = a.elements + job * nel;
        auto get = [&](uint64_t k) -> uint64_t { const uint64_t e = e0 + k; return e < nel ? el[e] : 0ull; };
        if (a.noop_small) poseidon_hash_or_noop(get, a.leaf_len, POSEIDON_RC, d);
        else poseidon_hash_no_pad(get, a.leaf_len, POSEIDON_RC, d);
    }
    uint64_t* o = a.tree + job * a.tree_stride + 4 * j;
    reinterpret_cast<ulonglong2*>(o)[0] = make_ulonglong2(d[0], d[1]);
    reinterpret_cast<ulonglong2*>(o)[1] = make_ulonglong2(d[2], d[3]);
}

// rows [n_rows, n_leaves) of every tree are all zero and share one digest: the extra lane of k_leaf_hashes wrote it to slot
// (tree 0, row n_rows); spread it (19 % of the leaf slots at B = 64, leaf_len 135)
__global__ __launch_bounds__(PS_THREADS) void k_copy_zero_leaves(LeafArgs a) {
    const uint64_t n_pad = a.n_leaves - a.n_rows;
    const uint64_t g = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    if (g >= (uint64_t)a.n_trees * n_pad || g == 0) return;
    const uint64_t job = g / n_pad, j = a.n_rows + g % n_pad;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a.tree + 4 * a.n_rows);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(a.tree + job * a.tree_stride + 4 * j);
    o[0] = src[0];
    o[1] = src[1];
}

// one level of every tree: parents [off + width, off + width + width / 2) from children [off, off + width)
__global__ __launch_bounds__(PS_THREADS) void k_merkle_level(uint64_t* __restrict__ tree, uint32_t n_trees, uint64_t tree_stride,
                                                             uint64_t level_off, uint32_t width) {
    const uint64_t g = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    const uint32_t half = width / 2;
    if (g >= (uint64_t)n_trees * half) return;
    const uint64_t job = g / half, t = g % half;
    uint64_t* base = tree + job * tree_stride + 4 * level_off;
    const ulonglong2* ch = reinterpret_cast<const ulonglong2*>(base + 8 * t);
    const ulonglong2 a0 = ch[0], a1 = ch[1], b0 = ch[2], b1 = ch[3];
    const uint64_t l[4] = {a0.x, a0.y, a1.x, a1.y}, r[4] = {b0.x, b0.y, b1.x, b1.y};
    uint64_t d[4];
    poseidon_two_to_one(l, r, POSEIDON_RC, d);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(base + 4 * (uint64_t)width + 4 * t);
    o[0] = make_ulonglong2(d[0], d[1]);
    o[1] = make_ulonglong2(d[2], d[3]);
}

}  // namespace bsx

extern "C" {
using namespace bsx;

hipError_t bsxk_poseidon_permute(hipStream_t s, const uint64_t* in, uint64_t n, uint64_t* out) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_poseidon_permute, dim3((uint32_t)((n + PS_THREADS - 1) / PS_THREADS)), dim3(PS_THREADS), 0, s, in, n, out);
    return hipGetLastError();
}

// leaf digests of n_trees witness matrices into tree[t][0..n_leaves); compact != NULL selects the fused form
hipError_t bsxk_leaf_hashes(hipStream_t s, const bsx_witness_layout* lay, uint32_t n_trees, const uint8_t* compact, const uint64_t* elements,
                            uint32_t leaf_len, uint32_t n_leaves, int noop_small, uint64_t tree_stride, uint64_t* tree) {
    if (!n_trees || !n_leaves) return hipSuccess;
    uint64_t n_rows = (lay->n_elements + leaf_len - 1) / leaf_len;
    if (n_rows > n_leaves) n_rows = n_leaves;
    LeafArgs a{*lay, n_trees, leaf_len, n_leaves, (uint32_t)noop_small, n_rows, tree_stride, compact, elements, tree};
    const uint64_t lanes = (uint64_t)n_trees * n_rows + (n_rows < n_leaves ? 1 : 0);
    if (lanes) {
        const dim3 grid((uint32_t)((lanes + PS_THREADS - 1) / PS_THREADS));
        if (compact) hipLaunchKernelGGL(k_leaf_hashes<true>, grid, dim3(PS_THREADS), 0, s, a);
        else hipLaunchKernelGGL(k_leaf_hashes<false>, grid, dim3(PS_THREADS), 0, s, a);
    }
    if (n_rows < n_leaves) {
        const uint64_t pads = (uint64_t)n_trees * (n_leaves - n_rows);
        hipLaunchKernelGGL(k_copy_zero_leaves, dim3((uint32_t)((pads + PS_THREADS - 1) / PS_THREADS)), dim3(PS_THREADS), 0, s, a);
    }
    return hipGetLastError();
}

// upper levels of n_trees trees whose n_leaves leaf digests are in place: down to 2^cap_height nodes
hipError_t bsxk_merkle_caps(hipStream_t s, uint64_t* tree, uint32_t n_trees, uint64_t tree_stride, uint32_t n_leaves, uint32_t cap_height) {
    uint64_t off = 0;
    for (uint32_t w = n_leaves; w > (1u << cap_height); w /= 2) {
        const uint64_t lanes = (uint64_t)n_trees * (w / 2);
        hipLaunchKernelGGL(k_merkle_level, dim3((uint32_t)((lanes + PS_THREADS - 1) / PS_THREADS)), dim3(PS_THREADS), 0, s, tree, n_trees,
                           tree_stride, off, w);
        off += w;
    }
    return hipGetLastError();
}
}

// one level over `width` (even, any size) digests at tree[0 .. width): parents to tree[width ..) — two_to_one batches
extern "C" hipError_t bsxk_merkle_one_level(hipStream_t s, uint64_t* tree, uint64_t width) {
    if (width < 2) return hipSuccess;
    const uint64_t lanes = width / 2;
    hipLaunchKernelGGL(bsx::k_merkle_level, dim3((uint32_t)((lanes + bsx::PS_THREADS - 1) / bsx::PS_THREADS)), dim3(bsx::PS_THREADS), 0, s,
                       tree, 1u, (uint64_t)0, (uint64_t)0, (uint32_t)width);
    return hipGetLastError();
}
