// api_poseidon.hip — host side of the Poseidon/Goldilocks entry points of include/bsx.h.
// Argument validation, H2D/D2H and layout bookkeeping only; every field operation runs in kernels_poseidon.hip.
#include <vector>

#include "api_internal.h"

#include "kernels.h"

using bsxapi::DBuf;
using bsxapi::fail;
using bsxapi::pow2;
using bsxapi::use;

#define H2D(dst, src, n) HIPCHK(hipMemcpyAsync((dst), (src), (n), hipMemcpyHostToDevice, st))
#define D2H(dst, src, n) HIPCHK(hipMemcpyAsync((dst), (src), (n), hipMemcpyDeviceToHost, st))
#define SYNC() HIPCHK(hipStreamSynchronize(st))

static int tree_args_ok(uint32_t leaf_len, uint32_t n_leaves, uint32_t cap_height) {
    if (!leaf_len) return fail(BSX_ERR_BAD_ARG, "leaf_len is 0");
    if (!pow2(n_leaves)) return fail(BSX_ERR_BAD_ARG, "n_leaves must be a power of two (MerkleTree::new, log2_strict)");
    if (cap_height > 31 || (1u << cap_height) > n_leaves) return fail(BSX_ERR_BAD_ARG, "cap_height %u exceeds log2(n_leaves)", cap_height);
    return BSX_OK;
}

extern "C" {

uint64_t bsx_poseidon_tree_digests(uint32_t n_leaves, uint32_t cap_height) {
    if (!pow2(n_leaves) || cap_height > 31 || (1u << cap_height) > n_leaves) return 0;
    return 2ull * n_leaves - (1ull << cap_height);
}

uint32_t bsx_witness_leaf_count(uint64_t n_elements, uint32_t leaf_len) {
    if (!leaf_len || !n_elements) return 0;
    const uint64_t rows = (n_elements + leaf_len - 1) / leaf_len;
    uint64_t p = 1;
    while (p < rows) p *= 2;
    return p > 0x80000000ull ? 0 : (uint32_t)p;
}

int bsx_dev_poseidon_permute(bsx_ctx* ctx, void* stream, const uint64_t* d_states, uint64_t n, uint64_t* d_out) {
    RET(use(ctx));
    if (n && (!d_states || !d_out)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_poseidon_permute(static_cast<hipStream_t>(stream), d_states, n, d_out));
    return BSX_OK;
}

int bsx_dev_poseidon_leaf_hashes(bsx_ctx* ctx, void* stream, const uint64_t* d_elements, uint32_t n_trees, uint64_t n_elements,
                                 uint32_t leaf_len, uint32_t n_leaves, uint64_t tree_stride, uint64_t* d_trees) {
    RET(use(ctx));
    if (!d_elements || !d_trees) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!leaf_len || !n_leaves) return fail(BSX_ERR_BAD_ARG, "leaf_len / n_leaves is 0");
    if (tree_stride < 4ull * n_leaves || (tree_stride & 1)) return fail(BSX_ERR_BAD_ARG, "tree_stride must be even and >= 4 * n_leaves");
    if ((n_elements + leaf_len - 1) / leaf_len > n_leaves) return fail(BSX_ERR_BAD_ARG, "n_leaves rows of leaf_len do not cover n_elements");
    bsx_witness_layout lay{};
    lay.n_elements = n_elements;
    HIPCHK(bsxk_leaf_hashes(static_cast<hipStream_t>(stream), &lay, n_trees, nullptr, d_elements, leaf_len, n_leaves, 1, tree_stride, d_trees));
    return BSX_OK;
}

int bsx_dev_witness_leaf_hashes(bsx_ctx* ctx, void* stream, const bsx_witness_layout* layout, uint32_t n_jobs, const uint8_t* d_compact,
                                uint32_t leaf_len, uint32_t n_leaves, uint64_t tree_stride, uint64_t* d_trees) {
    RET(use(ctx));
    if (!layout || !d_compact || !d_trees) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!leaf_len || !n_leaves) return fail(BSX_ERR_BAD_ARG, "leaf_len / n_leaves is 0");
    if (tree_stride < 4ull * n_leaves || (tree_stride & 1)) return fail(BSX_ERR_BAD_ARG, "tree_stride must be even and >= 4 * n_leaves");
    if ((layout->n_elements + leaf_len - 1) / leaf_len > n_leaves) return fail(BSX_ERR_BAD_ARG, "n_leaves rows of leaf_len do not cover the witness");
    HIPCHK(bsxk_leaf_hashes(static_cast<hipStream_t>(stream), layout, n_jobs, d_compact, nullptr, leaf_len, n_leaves, 1, tree_stride, d_trees));
    return BSX_OK;
}

int bsx_dev_poseidon_merkle_caps(bsx_ctx* ctx, void* stream, uint64_t* d_trees, uint32_t n_trees, uint64_t tree_stride, uint32_t n_leaves,
                                 uint32_t cap_height) {
    RET(use(ctx));
    if (!d_trees) return fail(BSX_ERR_BAD_ARG, "null pointer");
    RET(tree_args_ok(1, n_leaves, cap_height));
    if (tree_stride < 4 * bsx_poseidon_tree_digests(n_leaves, cap_height) || (tree_stride & 1))
        return fail(BSX_ERR_BAD_ARG, "tree_stride must be even and >= 4 * bsx_poseidon_tree_digests()");
    HIPCHK(bsxk_merkle_caps(static_cast<hipStream_t>(stream), d_trees, n_trees, tree_stride, n_leaves, cap_height));
    return BSX_OK;
}

// ------------------------------------------------------------------------------------------------ host tier
int bsx_poseidon_permute(bsx_ctx* ctx, const uint64_t* states, uint64_t n, uint64_t* out) {
    RET(use(ctx));
    bsxapi::ArenaScope arena_scope_(ctx);
    if (n && (!states || !out)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!n) return BSX_OK;
    hipStream_t st = ctx->stream;
    DBuf d;
    RET(d.alloc(n * 96));
    H2D(d.p, states, n * 96);
    HIPCHK(bsxk_poseidon_permute(st, d.as<uint64_t>(), n, d.as<uint64_t>()));
    D2H(out, d.p, n * 96);
    SYNC();
    return BSX_OK;
}

int bsx_poseidon_hash_no_pad(bsx_ctx* ctx, const uint64_t* elements, uint64_t n_inputs, uint32_t len, uint64_t* out_digests) {
    RET(use(ctx));
    bsxapi::ArenaScope arena_scope_(ctx);
    if (n_inputs && (!out_digests || (len && !elements))) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!n_inputs) return BSX_OK;
    if (n_inputs > 0x7fffffffull) return fail(BSX_ERR_BAD_ARG, "too many inputs");
    if (len == 0) {   // the sponge absorbs nothing: digest of the zero state without a permutation
        for (uint64_t i = 0; i < n_inputs * 4; i++) out_digests[i] = 0;
        return BSX_OK;
    }
    hipStream_t st = ctx->stream;
    DBuf de, dd;
    RET(de.alloc(n_inputs * len * 8));
    RET(dd.alloc(n_inputs * 32));
    H2D(de.p, elements, n_inputs * len * 8);
    bsx_witness_layout lay{};
    lay.n_elements = n_inputs * len;
    HIPCHK(bsxk_leaf_hashes(st, &lay, 1, nullptr, de.as<uint64_t>(), len, (uint32_t)n_inputs, 0, 4 * n_inputs, dd.as<uint64_t>()));
    D2H(out_digests, dd.p, n_inputs * 32);
    SYNC();
    return BSX_OK;
}

int bsx_poseidon_two_to_one(bsx_ctx* ctx, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out_digests) {
    RET(use(ctx));
    bsxapi::ArenaScope arena_scope_(ctx);
    if (n && (!left || !right || !out_digests)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!n) return BSX_OK;
    if (n > 0x3fffffffull) return fail(BSX_ERR_BAD_ARG, "too many pairs");
    hipStream_t st = ctx->stream;
    // one tree level of width 2n: children interleaved (l0, r0, l1, r1, ...), parents behind them
    std::vector<uint64_t> lvl(8 * n);
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) { lvl[8 * i + k] = left[4 * i + k]; lvl[8 * i + 4 + k] = right[4 * i + k]; }
    DBuf d;
    RET(d.alloc(12 * n * 8));
    H2D(d.p, lvl.data(), 8 * n * 8);
    HIPCHK(bsxk_merkle_one_level(st, d.as<uint64_t>(), 2 * n));
    D2H(out_digests, d.as<uint64_t>() + 8 * n, 4 * n * 8);
    SYNC();
    return BSX_OK;
}

int bsx_poseidon_merkle_tree(bsx_ctx* ctx, const uint64_t* elements, uint64_t n_elements, uint32_t leaf_len, uint32_t n_leaves,
                             uint32_t cap_height, uint64_t* out_tree) {
    RET(use(ctx));
    bsxapi::ArenaScope arena_scope_(ctx);
    if (!elements || !out_tree) return fail(BSX_ERR_BAD_ARG, "null pointer");
    RET(tree_args_ok(leaf_len, n_leaves, cap_height));
    if ((n_elements + leaf_len - 1) / leaf_len > n_leaves) return fail(BSX_ERR_BAD_ARG, "n_leaves rows of leaf_len do not cover n_elements");
    hipStream_t st = ctx->stream;
    const uint64_t nd = bsx_poseidon_tree_digests(n_leaves, cap_height);
    DBuf de, dt;
    RET(de.alloc(n_elements * 8));
    RET(dt.alloc(nd * 32));
    H2D(de.p, elements, n_elements * 8);
    bsx_witness_layout lay{};
    lay.n_elements = n_elements;
    HIPCHK(bsxk_leaf_hashes(st, &lay, 1, nullptr, de.as<uint64_t>(), leaf_len, n_leaves, 1, 4 * nd, dt.as<uint64_t>()));
    HIPCHK(bsxk_merkle_caps(st, dt.as<uint64_t>(), 1, 4 * nd, n_leaves, cap_height));
    D2H(out_tree, dt.p, nd * 32);
    SYNC();
    return BSX_OK;
}

int bsx_witness_merkle_caps(bsx_ctx* ctx, const bsx_witness_layout* layout, const uint64_t* witness, uint32_t n_jobs, uint32_t leaf_len,
                            uint32_t cap_height, uint64_t* out_caps) {
    RET(use(ctx));
    bsxapi::ArenaScope arena_scope_(ctx);
    if (!layout || !witness || !out_caps) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!n_jobs) return BSX_OK;
    const uint32_t n_leaves = bsx_witness_leaf_count(layout->n_elements, leaf_len);
    if (!n_leaves) return fail(BSX_ERR_BAD_ARG, "leaf_len is 0 or the witness is empty");
    RET(tree_args_ok(leaf_len, n_leaves, cap_height));
    hipStream_t st = ctx->stream;
    const uint64_t nd = bsx_poseidon_tree_digests(n_leaves, cap_height), nel = layout->n_elements, ncap = 1ull << cap_height;
    DBuf de, dt;
    RET(de.alloc((size_t)n_jobs * nel * 8));
    RET(dt.alloc((size_t)n_jobs * nd * 32));
    H2D(de.p, witness, (size_t)n_jobs * nel * 8);
    HIPCHK(bsxk_leaf_hashes(st, layout, n_jobs, nullptr, de.as<uint64_t>(), leaf_len, n_leaves, 1, 4 * nd, dt.as<uint64_t>()));
    HIPCHK(bsxk_merkle_caps(st, dt.as<uint64_t>(), n_jobs, 4 * nd, n_leaves, cap_height));
    for (uint32_t j = 0; j < n_jobs; j++)
        D2H(out_caps + (size_t)j * ncap * 4, dt.as<uint64_t>() + ((size_t)j * nd + nd - ncap) * 4, ncap * 32);
    SYNC();
    return BSX_OK;
}

}  // extern "C"
