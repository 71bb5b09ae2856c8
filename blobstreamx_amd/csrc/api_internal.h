// api_internal.h — helpers shared by the host-side translation units of libbsx.so (api.hip, api_poseidon.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/bsx.h"

#include <vector>

// Scratch arena of the host tier: one device allocation per context, bump-allocated by the DBufs of a host-tier call
// and rewound when the outermost call returns — a hipMalloc/hipFree pair per buffer per call cost more than the
// kernels of a single header_range (bench.py `latency`).  A call that needs more than the arena holds takes overflow
// chunks for its duration; the arena is then regrown once to that call's total.
struct bsx_arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, need = 0;
    int depth = 0;
    std::vector<void*> overflow;
};
struct bsx_vmm_block {
    void* va;
    size_t size;
    hipMemGenericAllocationHandle_t handle;
};
// shape of a bsx_header_range request: everything its launch sequence depends on (a captured graph is replayed only for an
// identical key, including the addresses the arena / staging / key table handed out)
struct HrGraphKey {
    uint32_t J, B, V, chain_id_len;
    uint64_t hpr, span;
    uint8_t chain_id[56];
    void *arena_base, *hstage, *keytab;
    size_t arena_cap;
};
struct bsx_ctx {
    int device;
    hipStream_t stream;
    hipStream_t stream3 = nullptr;       // host tier: work the commit check needs but need not wait for in line
    hipEvent_t ev_d = nullptr, ev_e = nullptr;
    hipStream_t stream4 = nullptr;       // host tier: the target set's leaves + tree + total, beside the challenges (round 4)
    hipEvent_t ev_f = nullptr, ev_g = nullptr;
    bool graphs_enabled = false;         // BSX_TUNE_HOST_GRAPHS (off: ROCm 7.2 runs a graph's parallel branches one after the other)
    bool hr_seen = false;                // the previous bsx_header_range request was graphable, with key hr_seen_key
    HrGraphKey hr_seen_key{}, hr_key{};
    hipGraphExec_t hr_exec = nullptr;    // captured launch sequence of bsx_header_range for hr_key
    void* hr_d_headers = nullptr;        // where that sequence reads the range's headers
    hipStream_t stream2 = nullptr;       // host tier: the commit check of a header_range runs beside its hashing chain
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bsx_arena arena;
    std::vector<bsx_vmm_block> vmm;      // bsx_dev_alloc blocks still alive
    std::vector<bsx_vmm_block> vmm_free; // released by bsx_dev_free, still mapped: recycled by the next bsx_dev_alloc that fits
    uint8_t* zero_paths = nullptr;       // 320 B: path digests of the hint's zero-padded proofs (k_zero_paths)
    uint8_t* keytab = nullptr;           // host tier's persistent fixed-key Ed25519 table (rows survive between calls)
    uint32_t keytab_rows = 0;
    std::vector<uint8_t> keytab_mirror;  // host copy of the public keys the table's rows were last built for (bsx_header_range)
    bool keytab_mirror_valid = false;
    uint8_t* btab = nullptr;             // fixed-key Ed25519 table of the base point B (built by bsx_init)
    uint8_t* hstage = nullptr;           // page-locked host staging of the host tier's small inputs / results (SmallIO)
    size_t hstage_cap = 0;
    hipEvent_t ev_c = nullptr;
    uint32_t merkle_wgs = 0;             // BSX_TUNE_MERKLE_WORKGROUPS
    // The host tier keeps per-context mutable state (arena, page-locked staging, key table, stream2 + events): host-tier
    // entry points serialise on this lock, so ONE context may be shared by any number of threads (they take turns);
    // recursive because host-tier entry points nest (bsx_next_header -> bsx_header_hashes).  The device tier holds no
    // per-call context state and does not take it.
    std::recursive_mutex host_mu;
    // bsx_enable_coalescing: synchronous host-tier calls of the batcher's shape are submit + wait on it (batcher.hip)
    std::atomic<bsx_batcher*> batcher{nullptr};
    std::atomic<uint32_t> sync_range_callers{0};   // threads inside a synchronous bsx_header_range on a coalescing context (lone-caller test)
    std::atomic<uint64_t> sync_range_crowd_ns{0};  // steady-clock time a thread last found another one inside that wrapper
};

namespace bsxapi {
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);    // sets the thread-local message of bsx_last_error(), returns code
int use(bsx_ctx* ctx);                       // hipSetDevice(ctx->device)

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return bsxapi::fail(BSX_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define RET(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != BSX_OK) return rc_; \
    } while (0)

inline bool pow2(uint32_t x) { return x && !(x & (x - 1)); }

extern thread_local bsx_arena* tl_arena;      // arena of the host-tier call running on this thread (null: plain hipMalloc)

// RAII of one host-tier entry point (nested entry points share the outermost scope)
struct ArenaScope {
    std::lock_guard<std::recursive_mutex> lock;   // first member: taken before, released after the arena bookkeeping
    bsx_arena* a;
    bsx_arena* prev;
    explicit ArenaScope(bsx_ctx* ctx) : lock(ctx->host_mu), a(&ctx->arena), prev(tl_arena) {
        tl_arena = a;
        a->depth++;
    }
    ~ArenaScope() {
        if (--a->depth == 0) {
            if (!a->overflow.empty()) {                       // grow once to what this call needed
                for (void* q : a->overflow) (void)hipFree(q);
                a->overflow.clear();
                if (a->base) (void)hipFree(a->base);
                a->base = nullptr;
                a->cap = a->need + a->need / 4 + (1u << 20);
                if (hipMalloc(reinterpret_cast<void**>(&a->base), a->cap) != hipSuccess) { a->base = nullptr; a->cap = 0; }
            }
            a->off = 0;
            a->need = 0;
        }
        tl_arena = prev;
    }
};

// device buffer with the lifetime of one host-tier call
struct DBuf {
    void* p = nullptr;
    bool owned = false;
    ~DBuf() { if (p && owned) (void)hipFree(p); }
    int alloc(size_t n) {
        if (n == 0) n = 16;
        n = (n + 255) & ~(size_t)255;
        bsx_arena* a = tl_arena;
        if (a) {
            a->need += n;
            if (a->off + n <= a->cap) {
                p = a->base + a->off;
                a->off += n;
                return BSX_OK;
            }
        }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return fail(BSX_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        if (a) a->overflow.push_back(p); else owned = true;
        return BSX_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace bsxapi
