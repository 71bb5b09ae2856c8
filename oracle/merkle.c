/* oracle/merkle.c — CPU ORACLE (test infrastructure only; see orc.h).
 * Tendermint "simple" Merkle tree (RFC 6962 style: leaf = SHA256(0x00 ‖ x), inner =
 * SHA256(0x01 ‖ l ‖ r), split at the largest power of two strictly below n; empty = SHA256("")),
 * the 14-leaf header tree (tendermint 0.33.2 Header::hash, called at circuits/input.rs:250-261)
 * and inclusion proofs (tendermintx get_inclusion_proof, called at circuits/input.rs:175-179,
 * 188-195).  Byte formats confirmed against the five mocha-4 fixture blocks. */
#include <string.h>

#include "orc.h"

void orc_leaf_hash(const uint8_t* leaf, size_t len, uint8_t out[32]) {
    uint8_t buf[1 + 256];
    buf[0] = 0x00;
    memcpy(buf + 1, leaf, len);
    orc_sha256(buf, len + 1, out);
}

void orc_inner_hash(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) {
    uint8_t buf[65];
    buf[0] = 0x01;
    memcpy(buf + 1, l, 32);
    memcpy(buf + 33, r, 32);
    orc_sha256(buf, 65, out);
}

static size_t split_point(size_t n) {
    size_t k = 1;
    while (k * 2 < n) k *= 2;
    return k;
}

void orc_merkle_root(const uint8_t* const* items, const size_t* lens, size_t n, uint8_t out[32]) {
    if (n == 0) {
        orc_sha256((const uint8_t*)"", 0, out);
        return;
    }
    if (n == 1) {
        orc_leaf_hash(items[0], lens[0], out);
        return;
    }
    size_t k = split_point(n);
    uint8_t l[32], r[32];
    orc_merkle_root(items, lens, k, l);
    orc_merkle_root(items + k, lens + k, n - k, r);
    orc_inner_hash(l, r, out);
}

int orc_merkle_proof(const uint8_t* const* items, const size_t* lens, size_t n, size_t idx, uint8_t aunts[][32]) {
    if (n <= 1) return 0;
    size_t k = split_point(n);
    int d;
    if (idx < k) {
        d = orc_merkle_proof(items, lens, k, idx, aunts);
        orc_merkle_root(items + k, lens + k, n - k, aunts[d]);
    } else {
        d = orc_merkle_proof(items + k, lens + k, n - k, idx - k, aunts);
        orc_merkle_root(items, lens, k, aunts[d]);
    }
    return d + 1;
}

/* plonky2x get_root_from_merkle_proof [UPSTREAM; SURVEY Appendix B]: h = leaf_hash(leaf); per level
 * h = path[i] ? inner(aunt_i, h) : inner(h, aunt_i).  Called at circuits/builder.rs:189-199,429-433. */
void orc_root_from_proof(const uint8_t* leaf, size_t leaf_len, const uint8_t aunts[][32], const uint8_t* path_bits,
                         int depth, uint8_t out[32], uint8_t path_digests[][32]) {
    uint8_t h[32], t[32];
    orc_leaf_hash(leaf, leaf_len, h);
    if (path_digests) memcpy(path_digests[0], h, 32);
    for (int i = 0; i < depth; i++) {
        if (path_bits[i])
            orc_inner_hash(aunts[i], h, t);
        else
            orc_inner_hash(h, aunts[i], t);
        memcpy(h, t, 32);
        if (path_digests) memcpy(path_digests[i + 1], h, 32);
    }
    memcpy(out, h, 32);
}

/* ---- packed header -> 14 leaves */
static void header_items(const bsx_header* h, const uint8_t* items[14], size_t lens[14]) {
    items[0] = h->version; items[1] = h->chain_id; items[2] = h->height; items[3] = h->time;
    items[4] = h->last_block_id;
    for (int i = 0; i < 8; i++) items[5 + i] = h->hash[i];
    items[13] = h->proposer;
    for (int i = 0; i < 14; i++) lens[i] = h->len[i];
}

int orc_header_check(const bsx_header* h) {
    static const uint8_t cap[14] = {24, 52, 12, 20, 76, 36, 36, 36, 36, 36, 36, 36, 36, 24};
    for (int i = 0; i < 14; i++) {
        if (h->len[i] > cap[i]) return BSX_ERR_BAD_HEADER;
        if (i != 4 && h->len[i] > 54) return BSX_ERR_BAD_HEADER;
    }
    return BSX_OK;
}

int orc_header_hash(const bsx_header* h, uint8_t out_hash[32], bsx_data_hash_proof* dh, bsx_last_block_id_proof* lb) {
    const uint8_t* items[14];
    size_t lens[14];
    int rc = orc_header_check(h);
    if (rc) return rc;
    header_items(h, items, lens);
    if (out_hash) orc_merkle_root(items, lens, 14, out_hash);
    if (dh) {
        /* circuits/input.rs:172-181: leaf = data_hash.encode_vec() (34 B), index DATA_HASH_INDEX */
        if (lens[BSX_DATA_HASH_INDEX] != BSX_PROTOBUF_HASH_SIZE) return BSX_ERR_BAD_HEADER;
        if (orc_merkle_proof(items, lens, 14, BSX_DATA_HASH_INDEX, dh->aunts) != BSX_HEADER_PROOF_DEPTH)
            return BSX_ERR_BAD_HEADER;
        memcpy(dh->leaf, items[BSX_DATA_HASH_INDEX], BSX_PROTOBUF_HASH_SIZE);
    }
    if (lb) {
        /* circuits/input.rs:187-197: leaf = Protobuf::<RawBlockId>::encode_vec(last_block_id) (72 B) */
        if (lens[BSX_LAST_BLOCK_ID_INDEX] != BSX_PROTOBUF_BLOCK_ID_SIZE) return BSX_ERR_BAD_HEADER;
        if (orc_merkle_proof(items, lens, 14, BSX_LAST_BLOCK_ID_INDEX, lb->aunts) != BSX_HEADER_PROOF_DEPTH)
            return BSX_ERR_BAD_HEADER;
        memcpy(lb->leaf, items[BSX_LAST_BLOCK_ID_INDEX], BSX_PROTOBUF_BLOCK_ID_SIZE);
    }
    return BSX_OK;
}
