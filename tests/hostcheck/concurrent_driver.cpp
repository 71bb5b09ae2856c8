// tests/hostcheck/concurrent_driver.cpp — K native threads calling the host tier of libbsx.so concurrently, the reference's shape of
// use: one `prove` call per range under a multi-thread runtime (/root/reference/circuits/header_range.rs:180-181) and one hint call
// per map job, 32 per proof (/root/reference/circuits/builder.rs:325-332).  Measurement harness for bench.py's `latency.concurrent`
// and `hint_concurrent` legs and for tests: Python threads spend ~20 us under the GIL per ctypes call, which at 50,000 calls/s is
// the whole budget — the callers have to be native for the LIBRARY to be what is measured.  Includes include/bsx.h only.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/bsx.h"

namespace {
struct SpinBarrier {
    std::atomic<int> count{0}, gen{0};
    int n;
    explicit SpinBarrier(int n_) : n(n_) {}
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            while (gen.load(std::memory_order_acquire) == g) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
    }
};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" {

// K threads; thread k calls bsx_header_range on ctxs[shared ? 0 : k] with ITS inputs back to back for `seconds`.
// out_lat: K x cap per-call latencies in ms; out_counts: K; out64: K x 64 (last output); out_rc: K (first non-zero status, or 0).
// Returns the wall time of the measured interval in seconds (< 0: bad arguments).
// form 0: headers[k] = bsx_header records, n_headers[k] of them (bsx_header_range); form 1: headers[k] = a packed block of n_headers[k]
// BYTES (bsx_header_range_packed)
double cd_header_range_loop2(void** ctxs, int shared, int K, double seconds, uint32_t J, uint32_t B, uint32_t V, const uint8_t* input48 /* K x 48 */,
                             const void* const* headers, const uint64_t* first_height, const uint64_t* n_headers, const uint64_t* latest,
                             const bsx_validator* const* tv, const bsx_validator* const* rv, const uint8_t* chain_id, uint32_t chain_id_len, float* out_lat,
                             int cap, int* out_counts, uint8_t* out64, int* out_rc, int form) {
    if (K <= 0 || !ctxs || cap <= 0) return -1;
    SpinBarrier go(K + 1);
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int k = 0; k < K; k++) {
        out_counts[k] = 0;
        out_rc[k] = 0;
        th.emplace_back([&, k] {
            bsx_ctx* ctx = static_cast<bsx_ctx*>(ctxs[shared ? 0 : k]);
            uint8_t o[64];
            bsx_commit_result cr;
            int n = 0;
            go.wait();
            while (!stop.load(std::memory_order_relaxed)) {
                const double t0 = now_ms();
                const int rc = form == 1
                    ? bsx_header_range_packed(ctx, J, B, input48 + 48 * (size_t)k, headers[k], n_headers[k], first_height[k], latest[k], tv[k], rv[k], V, chain_id,
                                              chain_id_len, o, &cr)
                    : bsx_header_range(ctx, J, B, input48 + 48 * (size_t)k, static_cast<const bsx_header*>(headers[k]), first_height[k], n_headers[k], latest[k], tv[k],
                                       rv[k], V, chain_id, chain_id_len, o, &cr, nullptr);
                const double t1 = now_ms();
                if (rc != BSX_OK && !out_rc[k]) out_rc[k] = rc;
                if (n < cap) out_lat[(size_t)k * cap + n] = (float)(t1 - t0);
                n++;
            }
            out_counts[k] = n;
            memcpy(out64 + 64 * (size_t)k, o, 64);
        });
    }
    go.wait();
    const double t0 = now_ms();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto& t : th) t.join();
    return (now_ms() - t0) / 1e3;
}

double cd_header_range_loop(void** ctxs, int shared, int K, double seconds, uint32_t J, uint32_t B, uint32_t V, const uint8_t* input48,
                            const bsx_header* const* headers, const uint64_t* first_height, const uint64_t* n_headers, const uint64_t* latest,
                            const bsx_validator* const* tv, const bsx_validator* const* rv, const uint8_t* chain_id, uint32_t chain_id_len, float* out_lat,
                            int cap, int* out_counts, uint8_t* out64, int* out_rc) {
    return cd_header_range_loop2(ctxs, shared, K, seconds, J, B, V, input48, reinterpret_cast<const void* const*>(headers), first_height, n_headers, latest, tv, rv,
                                 chain_id, chain_id_len, out_lat, cap, out_counts, out64, out_rc, 0);
}

// The map-job hints of ONE proof, one thread each (builder.rs:325-332: `async_hint` per map job under the runtime): thread j calls
// bsx_data_commitment_inputs for [S + j B, S + (j + 1) B) and — with_subchain = 1 — bsx_prove_subchain on what it returned
// (builder.rs:335); with_subchain = 2: both as ONE bsx_map_job call.  `reps` bursts, all threads released together by a spin barrier; out_wall_ms[rep] = release -> last return.
// out_records: J records of the last burst (with_subchain), out_start_end: J x 64 (start ‖ end header of the last burst).
int cd_hint_burst(void* ctx_, int J, uint32_t B, int reps, int with_subchain, const bsx_header* headers /* J B + 1, height S + i */, uint64_t S,
                  uint64_t latest, uint64_t global_end, const uint8_t* global_end_hash, float* out_wall_ms, bsx_subchain* out_records,
                  uint8_t* out_start_end) {
    if (J <= 0 || reps <= 0 || !ctx_) return -1;
    bsx_ctx* ctx = static_cast<bsx_ctx*>(ctx_);
    SpinBarrier bar(J + 1);
    std::atomic<int> first_rc{0};
    std::vector<std::thread> th;
    for (int j = 0; j < J; j++) {
        th.emplace_back([&, j] {
            std::vector<bsx_data_hash_proof> dh(B);
            std::vector<bsx_last_block_id_proof> lb(B);
            uint8_t sh[32], eh[32];
            bsx_subchain rec;
            memset(&rec, 0, sizeof rec);
            const uint64_t bs = S + (uint64_t)j * B, be = bs + B;
            for (int r = 0; r < reps; r++) {
                bar.wait();
                int rc;
                if (with_subchain == 2) {                                  // the whole map closure as ONE call (builder.rs:305-336)
                    bsx_shared_ctx rg;
                    memset(&rg, 0, sizeof rg);
                    rg.start_block = S; rg.end_block = global_end;
                    memcpy(rg.end_header_hash, global_end_hash, 32);
                    rc = bsx_map_job(ctx, (uint32_t)J, B, &rg, (uint32_t)j, headers + (size_t)j * B, bs, (uint64_t)B + 1, latest, sh, eh, dh.data(), lb.data(), &rec);
                    if (rc == BSX_ERR_ASSERT) rc = BSX_OK;
                } else {
                    rc = bsx_data_commitment_inputs(ctx, headers + (size_t)j * B, bs, (uint64_t)B + 1, latest, bs, be, B, sh, eh, dh.data(), lb.data(), nullptr);
                }
                if (rc == BSX_OK && with_subchain == 1) {
                    rc = bsx_prove_subchain(ctx, B, sh, eh, dh.data(), lb.data(), bs, be, global_end, global_end_hash, &rec, nullptr);
                    if (rc == BSX_ERR_ASSERT) rc = BSX_OK;                 // a failing assertion is a result, not a harness error
                }
                if (rc != BSX_OK) { int z = 0; first_rc.compare_exchange_strong(z, rc); }
                bar.wait();
            }
            if (out_records) out_records[j] = rec;
            if (out_start_end) { memcpy(out_start_end + 64 * (size_t)j, sh, 32); memcpy(out_start_end + 64 * (size_t)j + 32, eh, 32); }
        });
    }
    for (int r = 0; r < reps; r++) {
        const double t0 = now_ms();
        bar.wait();                  // release
        bar.wait();                  // all returned
        out_wall_ms[r] = (float)(now_ms() - t0);
    }
    for (auto& t : th) t.join();
    return first_rc.load();
}

}  // extern "C"
