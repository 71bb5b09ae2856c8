// keycache.h — which row of a fixed-key Ed25519 table holds which public key.  Host-side bookkeeping only (32-byte compares, a hash
// map, an LRU stamp); the arithmetic is in kernels_ed.hip.
//
// builder.skip verifies the target header's commit against the TARGET validator set (circuits/header_range.rs:42-48), and validator
// sets change along the chain — that is what `skip` exists for (circuits/fetcher.rs:60-87 searches a target whose set still
// overlaps the trusted one).  Rounds 2-4 built ONE table per chunk from the first range's set, row i = slot i, and sent every slot of
// another range whose key differed to the generic kernel (256 doublings: a 1.4 ms chain).  Here rows are keyed by PUBLIC KEY: a slot
// first tries row = its own index (the common case: every range of a batch signed by one set — then the map is the identity and no
// row array is uploaded at all), then the map, then takes the least recently used row that no slot of the current batch needs.  Rows
// persist across batches (a validator keeps its key for months); a key the table has no room for is deferred to the generic kernel
// as before.  The device side never trusts the map: a lane uses a row only if the row's key record IS its public key.
#pragma once
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/bsx.h"

struct bsx_keycache {
    uint32_t V = 0, N = 0;                    // validator slots per set; table rows (N >= V)
    uint64_t gen = 0;                         // batch counter: rows stamped with the current one are pinned
    std::vector<uint8_t> keys;                // N x 32
    std::vector<uint8_t> used;                // N
    std::vector<uint64_t> stamp;              // N
    std::unordered_multimap<uint64_t, uint32_t> map;   // first 8 key bytes -> row
    uint64_t n_assigned = 0, n_deferred = 0;  // statistics: rows (re)built, slots left to the generic kernel

    void init(uint32_t v, uint32_t n) {
        V = v; N = n < v ? v : n;
        keys.assign((size_t)N * 32, 0); used.assign(N, 0); stamp.assign(N, 0); map.clear(); gen = 0;
    }
    static uint64_t h8(const uint8_t* pk) { uint64_t x; memcpy(&x, pk, 8); return x; }
    void unmap(uint32_t row) {
        auto rg = map.equal_range(h8(&keys[(size_t)row * 32]));
        for (auto it = rg.first; it != rg.second; ++it)
            if (it->second == row) { map.erase(it); return; }
    }
    int64_t find(const uint8_t* pk) const {
        auto rg = map.equal_range(h8(pk));
        for (auto it = rg.first; it != rg.second; ++it)
            if (memcmp(&keys[(size_t)it->second * 32], pk, 32) == 0) return it->second;
        return -1;
    }
    // One batch of R validator sets (V slots each, host memory).  rows_out (R * V u32): the table row of every slot (its own index for
    // inactive slots; 0xffffffff when the table has no room: deferred).  dirty: the rows whose key changed — the caller rewrites
    // their key records and rebuilds them BEFORE the signature check.  Returns true when the map is the identity (rows_out need not
    // be uploaded: pass nullptr to the kernels).
    bool assign(const bsx_validator* sets, uint32_t R, uint32_t* rows_out, std::vector<uint32_t>& dirty, uint64_t* out_deferred) {
        gen++;
        dirty.clear();
        bool identity = true;
        uint64_t deferred = 0;
        for (uint32_t r = 0; r < R; r++)
            for (uint32_t i = 0; i < V; i++) {
                const bsx_validator& x = sets[(size_t)r * V + i];
                uint32_t& out = rows_out[(size_t)r * V + i];
                out = i;
                if (!(x.enabled && x.is_signed)) continue;
                if (used[i] && memcmp(&keys[(size_t)i * 32], x.pubkey, 32) == 0) { stamp[i] = gen; continue; }       // row = slot: the common case
                int64_t q = find(x.pubkey);
                if (q < 0) {
                    // a free row — the slot's own first (a new validator set then settles into rows 0 .. V-1) — else the least recently used
                    // row nobody in this batch needs
                    if (!used[i]) q = i;
                    else {
                        uint64_t best = gen;
                        for (uint32_t k = 0; k < N; k++) {
                            if (!used[k]) { q = k; break; }
                            if (stamp[k] < best) { best = stamp[k]; q = k; }
                        }
                    }
                    if (q < 0) { out = 0xffffffffu; identity = false; deferred++; continue; }
                    if (used[q]) unmap((uint32_t)q);
                    memcpy(&keys[(size_t)q * 32], x.pubkey, 32);
                    used[q] = 1;
                    map.emplace(h8(x.pubkey), (uint32_t)q);
                    dirty.push_back((uint32_t)q);
                    n_assigned++;
                }
                stamp[q] = gen;
                out = (uint32_t)q;
                if ((uint32_t)q != i) identity = false;
            }
        n_deferred += deferred;
        if (out_deferred) *out_deferred = deferred;
        return identity;
    }
};
