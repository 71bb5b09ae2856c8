// batcher.hip — the coalescing front end of the host tier (include/bsx.h, bsx_batcher_* / bsx_submit_* / bsx_wait).
//
// The reference's own call shape is ONE range per `prove` call under a multi-thread runtime (circuits/header_range.rs:180-181)
// and ONE hint call per map job, 32 async hints per proof (circuits/builder.rs:325-332 -> circuits/data_commitment.rs:22-44,
// `async fn hint`).  Served one by one, such a call is a string of ~10 dependent kernels of a single wave each: 0.26 ms for 2-3 % of
// the GPU, and K concurrent callers on K contexts time-slice the same queues (round 4: 16 callers = 1.4x one caller).  The same
// kernels take R ranges per launch at almost the same latency (bsx_pipeline_step: 1 range 0.15 ms, 16 ranges 0.22 ms).
//
// So concurrent requests are COALESCED: a submit claims a slot of the batch that is currently open, copies its inputs into the
// batch's page-locked staging (on the caller's thread, callers in parallel) and returns a ticket; the worker thread of the lane that
// owns the batch closes it after a short window (an idle GPU never waits longer than a debounce gap), uploads the staged inputs
// with a handful of copies, runs ONE launch set over the R requests — the host tier's own kernels with n_ranges = R — downloads the
// small results in one block and publishes the batch: its waiters (and nobody else) are woken through the lane's futex word, each copies
// ITS OWN results and decodes ITS OWN status out of that block, and the worker takes out what nobody has (round 6).  Statuses are per
// request on the device (header / hint status words are indexed by request, assertion masks and skip statuses always were), so a
// malformed or tampered request never fails its batch-mates.  n_lanes batches are in flight: the H2D copy of one runs beside the
// kernels of another.
//
// Three request kinds, each with its own lanes: header_range (CombinedSkipCircuit::define, header_range.rs:32-59),
// data_commitment_inputs (the hint, data_commitment.rs:18-45 -> input.rs:149-271) and prove_subchain (builder.rs:150-271).
//
// Host code only: slots, staging, stream/event choreography, status decoding.  All arithmetic is in kernels_*.hip.
#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"
#include "api_internal.h"
#include "kernels.h"
#include "keycache.h"

using bsxapi::fail;
using bsxapi::pow2;

namespace {

inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

constexpr uint32_t RING = 1u << 14;            // completion records kept: a ticket older than this many requests has expired
constexpr uint32_t GAP_NS = 12000;             // debounce: a batch closes once no claim arrived for this long ... (BSX_BATCH_GAP_NS in the experiments build;
                                               // 3 / 6 / 12 / 25 / 50 us measured, tools/exp_batch_gap.py: 12 us gathers 18 instead of 15 of a proof's 32 hints per set)
constexpr uint64_t CORK_MAX_NS = 20000000;     // a corked batch is released after 20 ms whatever happens
// Completion (round 6, VERDICT r5 #2a): no global condition variable.  A ticket's record names the lane and the batch of that lane it rides
// in; a waiter parks on THAT lane's done counter (a futex word), so a batch's completion wakes its own waiters and nobody else, with one
// system call and no mutex to queue on afterwards.  The woken caller then copies ITS OWN results out of the lane's page-locked block
// (consume); the worker sweeps what nobody has taken yet before it reopens the lane, so results never depend on a waiter showing up.
struct Lane;
struct DoneRec {
    // ticket * 4 + state: 0 = not consumed yet, 1 = somebody is copying the results out, 2 = final (rc / msg valid, outputs at the
    // caller's pointers).  0 (the whole word) = the record is being re-assigned: ADVICE r5 — the writer invalidates FIRST, so a late
    // bsx_wait for the ticket of RING requests ago can never read the new request's rc / msg under the old number
    std::atomic<uint64_t> st{0};
    Lane* lane = nullptr;
    uint32_t slot = 0;
    uint32_t batch_no = 0;                     // the lane's done counter when the ticket was issued: done once the counter has moved on
    int rc = 0;
    char msg[236] = {0};
};
inline void futex_wait(std::atomic<uint32_t>* w, uint32_t seen) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* w) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
static_assert(sizeof(std::atomic<uint32_t>) == 4, "futex word");

// what a request leaves behind at submit: where its results go (the caller's pointers: they must stay valid until bsx_wait returns)
struct Req {
    uint64_t seq = 0;
    uint8_t* output64 = nullptr;               // header_range
    bsx_commit_result* out_commit = nullptr;
    uint8_t *out_start = nullptr, *out_end = nullptr, *out_expected = nullptr;   // hint
    bsx_data_hash_proof* out_dh = nullptr;
    bsx_last_block_id_proof* out_lb = nullptr;
    bsx_subchain* out_record = nullptr;        // prove_subchain
    uint32_t n_headers = 0;                    // headers staged in the slot
    const bsx_header* direct = nullptr;        // page-locked caller memory that stays valid until bsx_wait: uploaded from there, not staged
    uint32_t packed_bytes = 0;                 // > 0: the headers came PACKED (wire.cpp): the request's block in the lane's packed staging ...
    uint32_t packed_off = 0xffffffffu;         // ... at this offset (assigned at claim)
    bool want_record = false;                  // hint kind: a map-job request (prove_subchain behind the hint)
};

struct Kind;
struct Lane {
    Kind* kind = nullptr;
    uint32_t index = 0;
    std::thread th;
    std::atomic<uint32_t> done_count{0};       // batches of this lane that have completed: the futex word its waiters park on
    std::atomic<uint32_t> waiters{0};          // threads parked (or about to park) on done_count: no wake-up call when nobody is
    int batch_rc = BSX_OK;                     // of the batch that completed last (a failed launch fails every ticket of the batch)
    std::string batch_err;
    enum State : int { FREE = 0, OPEN = 1, CLOSED = 2 };
    std::atomic<int> state{FREE};
    std::atomic<uint32_t> n_claimed{0};
    std::unique_ptr<std::atomic<uint8_t>[]> slot_ready;   // set by the submitter once its inputs are in the slot
    std::atomic<uint64_t> t_first{0}, t_last{0};
    std::vector<Req> reqs;
    std::vector<uint32_t> slot_hwm;            // headers the slot's STAGING holds (stage_headers keeps everything behind them zero)
    std::vector<uint32_t> dev_hwm;             // headers the slot's DEVICE block holds (the worker keeps everything behind them zero)
    // resources (kind-specific use)
    hipStream_t st = nullptr, sb = nullptr, s3 = nullptr, s4 = nullptr;
    hipEvent_t ev[8] = {nullptr};
    std::vector<void*> dallocs, hallocs;
    uint8_t *h_headers = nullptr, *h_small = nullptr, *h_out = nullptr;          // page-locked
    uint8_t *h_packed = nullptr, *d_packed = nullptr, *h_desc = nullptr, *d_desc = nullptr;   // packed wire headers (header_range kind)
    uint32_t packed_used = 0;                  // bytes of h_packed the open batch has handed out (under the kind's mutex)
    // device
    uint8_t *d_headers = nullptr, *d_hashes = nullptr, *d_dh = nullptr, *d_lb = nullptr, *d_paths = nullptr;
    uint8_t *d_ranges = nullptr, *d_latest = nullptr, *d_spans = nullptr, *d_compact = nullptr, *d_records = nullptr;
    uint8_t *d_tv = nullptr, *d_rv = nullptr, *d_h = nullptr, *d_ok = nullptr, *d_tres = nullptr, *d_th = nullptr, *d_th2 = nullptr, *d_rdec = nullptr;
    uint8_t *d_keytab = nullptr, *d_out = nullptr, *d_expected = nullptr;
    // the lane's fixed-key table: rows keyed by public key (keycache.h); key records of the rows as the build kernel reads them
    bsx_keycache kc;
    uint8_t *h_rowkeys = nullptr, *d_rowkeys = nullptr, *h_rows = nullptr, *d_rows = nullptr;
    std::vector<uint32_t> kc_dirty;
    std::string err;                           // lane_init failure
    uint64_t t_staged = 0, t_enqueued = 0;     // phase marks of the batch in flight (statistics)
};

}  // namespace

struct bsx_batcher {
    bsx_ctx* ctx = nullptr;
    bsx_batcher_config cfg{};
    uint32_t J = 0, B = 0, V = 0, M = 0, n_lanes = 0;
    uint64_t window_ns = 0, gap_ns = GAP_NS;
    uint64_t hpr = 0;                          // headers per header_range slot: J * B + 1
    uint32_t key_rows = 0;                     // rows of a lane's fixed-key table
    std::atomic<uint64_t> next_seq{1};
    std::vector<DoneRec> ring;
    std::atomic<uint32_t> inside{0};           // threads inside bsx_wait / bsx_poll / a submit: bsx_batcher_destroy lets them leave first
    std::unique_ptr<Kind> range_kind, hint_kind, subchain_kind;
    std::mutex mu_kinds;                       // lazy start of a kind's lanes
    std::atomic<int> corked{0};                // bsx_batcher_cork: open batches close only when full
    bsx_batcher() : ring(RING) {}
};

namespace {

struct Kind {
    bsx_batcher* b;
    const char* name;
    uint32_t M;                                // slots per batch
    std::mutex mu;
    std::condition_variable cv_open, cv_work;
    std::vector<std::unique_ptr<Lane>> lanes;
    std::atomic<Lane*> open{nullptr};          // written under `mu`; bsxb_range_idle peeks at it without
    std::deque<Lane*> free_;
    std::atomic<uint32_t> in_flight{0};
    std::atomic<bool> stop{false};             // written under `mu`; the closing spin peeks at it
    std::atomic<bool> started{false};
    int init_rc = BSX_OK;
    std::string init_err;
    // statistics
    std::atomic<uint64_t> n_batches{0}, n_requests{0}, max_batch{0}, sum_close_wait_ns{0}, sum_stage_wait_ns{0}, sum_enqueue_ns{0}, sum_gpu_wait_ns{0},
        sum_complete_ns{0};


    Kind(bsx_batcher* b_, const char* n, uint32_t m) : b(b_), name(n), M(m) {}
    virtual ~Kind() {}
    virtual int lane_init(Lane& l) = 0;
    virtual int launch(Lane& l, uint32_t R) = 0;                      // enqueue everything and wait for it
    // request r of the batch that just completed on `l`: decode ITS status words, copy ITS results from the lane's page-locked block to
    // the caller's pointers; returns the code the synchronous call would have returned (text into msg[236]).  Runs on the waiter's
    // thread — or on the worker's for tickets nobody is waiting for yet
    virtual int consume_one(Lane& l, uint32_t r, char* msg) = 0;
    virtual void on_launch_failed(Lane&) {}

    // whoever gets there first takes the results out (the waiter of the ticket or the worker's sweep); the other waits the few
    // microseconds the copy takes.  Returns false if the record no longer belongs to `ticket`.
    bool consume(DoneRec& d, uint64_t ticket) {
        uint64_t want = ticket * 4;
        if (d.st.compare_exchange_strong(want, ticket * 4 + 1, std::memory_order_acq_rel)) {
            Lane& l = *d.lane;
            char msg[sizeof d.msg];
            msg[0] = 0;
            int rc;
            if (l.batch_rc != BSX_OK) {
                rc = l.batch_rc;
                strncpy(msg, l.batch_err.c_str(), sizeof msg - 1);
                msg[sizeof msg - 1] = 0;
            } else {
                rc = consume_one(l, d.slot, msg);
            }
            d.rc = rc;
            memcpy(d.msg, msg, sizeof d.msg);
            d.st.store(ticket * 4 + 2, std::memory_order_release);
            return true;
        }
        while (want == ticket * 4 + 1) { cpu_relax(); want = d.st.load(std::memory_order_acquire); }
        return want == ticket * 4 + 2;
    }
    // the batch on `l` is over (its results are in l.h_out, or batch_rc says why not): wake its waiters, then take out whatever they
    // have not.  After this every ticket of the batch is final and the lane's blocks are free for the next batch
    void publish(Lane& l, uint32_t R, int rc, const std::string& err) {
        l.batch_rc = rc;
        l.batch_err = err;
        l.done_count.fetch_add(1);                                         // seq_cst: pairs with the waiter's waiters++ / re-check
        if (l.waiters.load() != 0) futex_wake_all(&l.done_count);
        for (uint32_t r = R; r-- > 0;) (void)consume(b->ring[l.reqs[r].seq % RING], l.reqs[r].seq);
    }

    int dalloc(Lane& l, size_t bytes, uint8_t** out) {
        if (bytes < 256) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e == hipSuccess) e = hipMemset(q, 0, bytes);
        if (e != hipSuccess) { if (q) (void)hipFree(q); return fail(BSX_ERR_HIP, "bsx_batcher(%s): hipMalloc(%zu): %s", name, bytes, hipGetErrorString(e)); }
        l.dallocs.push_back(q);
        *out = static_cast<uint8_t*>(q);
        return BSX_OK;
    }
    int halloc(Lane& l, size_t bytes, uint8_t** out) {
        if (bytes < 4096) bytes = 4096;
        void* q = nullptr;
        hipError_t e = hipHostMalloc(&q, bytes, hipHostMallocDefault);
        if (e != hipSuccess) return fail(BSX_ERR_HIP, "bsx_batcher(%s): hipHostMalloc(%zu): %s", name, bytes, hipGetErrorString(e));
        memset(q, 0, bytes);
        l.hallocs.push_back(q);
        *out = static_cast<uint8_t*>(q);
        return BSX_OK;
    }
    int lane_common(Lane& l) {
        for (hipStream_t* s : {&l.st, &l.sb, &l.s3, &l.s4}) HIPCHK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        for (auto& e : l.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        l.reqs.resize(M);
        l.slot_hwm.assign(M, 0);
        l.dev_hwm.assign(M, 0);
        l.slot_ready.reset(new std::atomic<uint8_t>[M]);
        for (uint32_t i = 0; i < M; i++) l.slot_ready[i].store(0);
        return BSX_OK;
    }
    void lane_free(Lane& l) {
        for (hipStream_t s : {l.st, l.sb, l.s3, l.s4}) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        for (auto e : l.ev) if (e) (void)hipEventDestroy(e);
        for (void* q : l.dallocs) (void)hipFree(q);
        for (void* q : l.hallocs) (void)hipHostFree(q);
    }

    // the worker of one lane: waits until its batch is the open one and holds a request, closes it (window policy), runs it
    void run(Lane* lp) {
        Lane& l = *lp;
        (void)hipSetDevice(b->ctx->device);
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (l.state.load() == Lane::OPEN && l.n_claimed.load() > 0); });
                if (stop) {
                    // requests claimed into the open batch must not be dropped — their waiters would hang (or touch a
                    // deleted batcher).  No new claim can arrive (claim() refuses under this mutex once `stop` is set)
                    const uint32_t n = l.state.load() == Lane::OPEN ? l.n_claimed.load() : 0;
                    if (n) {
                        l.state.store(Lane::CLOSED);
                        if (open.load() == &l) open.store(nullptr);
                        lk.unlock();
                        await_slots(l, n);                                         // a submitter may still be copying into its slot
                        publish(l, n, BSX_ERR_BAD_ARG, "bsx_batcher: destroyed before the request ran");
                    }
                    return;
                }
            }
            // window: full, or no claim for GAP_NS while the GPU is idle, or the window has run out (earlier batches keep the GPU busy:
            // waiting is free until then).  Spinning: the wait is tens of microseconds, below a timed condition wait's resolution
            const uint64_t t_open = l.t_first.load();
            for (;;) {
                const uint32_t n = l.n_claimed.load(std::memory_order_acquire);
                if (n >= M || stop.load(std::memory_order_relaxed)) break;     // destroy: what has been collected runs now
                const uint64_t t = now_ns();
                if (b->corked.load(std::memory_order_relaxed)) {                // the caller announced a burst: full batches only ...
                    if (t - l.t_first.load() >= CORK_MAX_NS) break;             // ... but a forgotten cork must not hang a waiter
                    cpu_relax();
                    continue;
                }
                const bool quiet = t - l.t_last.load(std::memory_order_acquire) >= b->gap_ns;
                if (quiet && (in_flight.load() == 0 || t - t_open >= b->window_ns)) break;
                if (t - l.t_first.load() >= 4 * b->window_ns + 1000000) break;   // never hold a request hostage to a stream of late claims
                cpu_relax();
            }
            uint32_t R;
            {
                std::lock_guard<std::mutex> lk(mu);
                R = l.n_claimed.load();
                l.state.store(Lane::CLOSED);
                in_flight.fetch_add(1);
                Lane* nx = nullptr;
                if (!free_.empty()) { nx = free_.front(); free_.pop_front(); nx->state.store(Lane::OPEN); }
                open.store(nx);
            }
            cv_open.notify_all();
            sum_close_wait_ns.fetch_add(now_ns() - t_open);
            bsxapi::g_err.clear();
            const uint64_t t_closed = now_ns();
            l.t_staged = l.t_enqueued = 0;
            const int rc = launch(l, R);
            const uint64_t t_done = now_ns();
            const std::string err = bsxapi::g_err;
            if (rc != BSX_OK) on_launch_failed(l);
            publish(l, R, rc, err);
            if (l.t_staged && l.t_enqueued) {
                sum_stage_wait_ns.fetch_add(l.t_staged - t_closed);
                sum_enqueue_ns.fetch_add(l.t_enqueued - l.t_staged);
                sum_gpu_wait_ns.fetch_add(t_done - l.t_enqueued);
            }
            sum_complete_ns.fetch_add(now_ns() - t_done);
            n_batches.fetch_add(1);
            n_requests.fetch_add(R);
            uint64_t mb = max_batch.load();
            while (R > mb && !max_batch.compare_exchange_weak(mb, R)) {}
            {
                std::lock_guard<std::mutex> lk(mu);
                l.n_claimed.store(0);
                l.packed_used = 0;
                for (uint32_t i = 0; i < R; i++) l.slot_ready[i].store(0, std::memory_order_relaxed);
                in_flight.fetch_sub(1);
                if (!open.load()) { open.store(&l); l.state.store(Lane::OPEN); } else { l.state.store(Lane::FREE); free_.push_back(&l); }
            }
            cv_open.notify_all();
        }
    }

    int start() {
        if (started.load()) return init_rc;
        started.store(true);
        (void)hipSetDevice(b->ctx->device);
        for (uint32_t i = 0; i < b->n_lanes; i++) {
            lanes.emplace_back(new Lane());
            Lane& l = *lanes.back();
            l.kind = this;
            l.index = i;
            int rc = lane_common(l);
            if (rc == BSX_OK) rc = lane_init(l);
            if (rc != BSX_OK) { init_rc = rc; init_err = bsxapi::g_err; return rc; }
        }
        if (hipDeviceSynchronize() != hipSuccess) { init_rc = BSX_ERR_HIP; init_err = "bsx_batcher: device error while creating lanes"; return init_rc; }
        lanes[0]->state.store(Lane::OPEN);
        open.store(lanes[0].get());
        for (size_t i = 1; i < lanes.size(); i++) free_.push_back(lanes[i].get());
        for (auto& l : lanes) l->th = std::thread(&Kind::run, this, l.get());
        return BSX_OK;
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        cv_open.notify_all();
        for (auto& l : lanes) if (l->th.joinable()) l->th.join();
    }
    void free_lanes() {
        (void)hipSetDevice(b->ctx->device);
        for (auto& l : lanes) lane_free(*l);
        lanes.clear();
    }

    // claim a slot of the open batch; the caller then stages its inputs into (lane, idx) and calls ready()
    int claim(Lane** out_lane, uint32_t* out_idx, const Req& proto, bsx_ticket* ticket) {
        std::unique_lock<std::mutex> lk(mu);
        cv_open.wait(lk, [&] { Lane* o = open.load(); return stop || (o && o->n_claimed.load() < M); });
        if (stop) return fail(BSX_ERR_BAD_ARG, "bsx_batcher: destroyed while a submit was waiting");
        Lane* l = open.load();
        const uint32_t idx = l->n_claimed.load();
        const uint64_t seq = b->next_seq.fetch_add(1);                   // shared by the three kinds (each under its own mutex)
        DoneRec& d = b->ring[seq % RING];
        {   // the record of the ticket RING requests ago: final long since — unless that request is somehow still on its way
            const uint64_t old = d.st.load(std::memory_order_acquire);
            if (old != 0 && (old & 3) != 2) return fail(BSX_ERR_UNSUPPORTED, "bsx_batcher: %u requests are outstanding (completion ring full)", RING);
        }
        const uint64_t t = now_ns();
        if (idx == 0) l->t_first.store(t);
        l->t_last.store(t, std::memory_order_release);
        Req& rq = l->reqs[idx];
        rq = proto;
        rq.seq = seq;
        if (proto.packed_bytes) {
            rq.packed_off = (l->packed_used + 15u) & ~15u;
            l->packed_used = rq.packed_off + proto.packed_bytes;
        }
        d.st.store(0, std::memory_order_release);                        // invalidate first (a late waiter of the old ticket reads "expired")
        d.lane = l;
        d.slot = idx;
        d.batch_no = l->done_count.load(std::memory_order_relaxed);
        d.st.store(seq * 4, std::memory_order_release);
        l->n_claimed.store(idx + 1, std::memory_order_release);
        *ticket = rq.seq;
        *out_lane = l;
        *out_idx = idx;
        lk.unlock();
        if (idx == 0) cv_work.notify_all();
        return BSX_OK;
    }
    static void ready(Lane* l, uint32_t idx) { l->slot_ready[idx].store(1, std::memory_order_release); }
    // the worker's side: slot idx holds its request's inputs (submitters may still be copying when the batch closes)
    static void await_slot(Lane& l, uint32_t idx) { while (!l.slot_ready[idx].load(std::memory_order_acquire)) cpu_relax(); }
    static void await_slots(Lane& l, uint32_t R) { for (uint32_t i = 0; i < R; i++) await_slot(l, i); }
};

#define LHIP(expr)                                                                                           \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(BSX_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ================================================================================================ header_range
// out block (device and page-locked mirror), arrays of M entries each
struct RangeOut {
    size_t o64, res, hst, ast, fst, skip, commit, total;
    explicit RangeOut(uint32_t M) {
        o64 = 0; res = o64 + (size_t)M * 64; hst = res + (size_t)M * 128; ast = hst + (size_t)M * 4; fst = ast + (size_t)M * 4;
        skip = fst + (size_t)M * 4; commit = skip + (size_t)M * 4; total = commit + 2 * (size_t)M * sizeof(bsx_commit_result);
    }
};                                             // commit: R target-set results, then R trusted-set results (one tally launch over both)
struct RangeSmall {                            // page-locked staging of the small inputs, arrays of M entries each
    size_t ranges, latest, tv, rv, total;
    RangeSmall(uint32_t M, uint32_t V) {
        ranges = 0; latest = ranges + (size_t)M * sizeof(bsx_shared_ctx);          // ranges + latest: ONE copy
        tv = latest + (size_t)M * 8; tv = (tv + 255) & ~(size_t)255;
        rv = tv + (size_t)M * V * sizeof(bsx_validator);
        total = rv + (size_t)M * V * sizeof(bsx_validator);
    }
};

struct RangeKind : Kind {
    RangeKind(bsx_batcher* b_) : Kind(b_, "header_range", b_->M) {}
    int lane_init(Lane& l) override {
        const uint32_t J = b->J, B = b->B, V = b->V;
        const uint64_t nh = (uint64_t)M * b->hpr;
        const bsx_witness_layout L = bsx_map_layout(B);
        const RangeOut O(M);
        const RangeSmall S(M, V);
        RET(halloc(l, nh * sizeof(bsx_header), &l.h_headers));
        RET(halloc(l, S.total, &l.h_small));
        RET(halloc(l, O.total, &l.h_out));
        RET(dalloc(l, nh * sizeof(bsx_header), &l.d_headers));
        RET(dalloc(l, nh * 32, &l.d_hashes));
        RET(dalloc(l, nh * 128, &l.d_dh));
        RET(dalloc(l, nh * 128, &l.d_lb));
        RET(dalloc(l, nh * BSX_HEADER_PATH_BYTES, &l.d_paths));
        RET(dalloc(l, (size_t)M * (sizeof(bsx_shared_ctx) + 8), &l.d_ranges));        // ranges, then latest (as staged)
        l.d_latest = l.d_ranges + (size_t)M * sizeof(bsx_shared_ctx);
        RET(dalloc(l, (size_t)M * J * L.compact_stride, &l.d_compact));
        RET(dalloc(l, (size_t)M * J * sizeof(bsx_subchain), &l.d_records));
        RET(dalloc(l, 2 * (size_t)M * V * sizeof(bsx_validator), &l.d_tv));          // R target sets, then R trusted sets
        RET(dalloc(l, (size_t)M * V * 32, &l.d_h));
        RET(dalloc(l, (size_t)M * V, &l.d_ok));
        RET(dalloc(l, (size_t)M * 32, &l.d_th));
        RET(dalloc(l, (size_t)M * 32, &l.d_th2));
        RET(dalloc(l, bsxk_ed25519_rdec_bytes((uint64_t)M * V), &l.d_rdec));
        RET(dalloc(l, O.total, &l.d_out));
        // packed wire headers: every request's block (offsets + at most 510 bytes per header), 16-aligned; descriptors (8 B) + wipe marks (4 B)
        const size_t pk = (size_t)M * (((4 * (b->hpr + 1) + 15) & ~(size_t)15) + b->hpr * 510 + 16);
        RET(halloc(l, pk, &l.h_packed));
        RET(dalloc(l, pk, &l.d_packed));
        RET(halloc(l, (size_t)M * 12, &l.h_desc));
        RET(dalloc(l, (size_t)M * 12, &l.d_desc));
        // every lane owns its fixed-key table (5.8 MB per row): a lane that rebuilds rows for new keys must not do so under another
        // lane's signature check.  Rows are keyed by public key (keycache.h): key_rows of them (default 2 V + 32), so that requests
        // signed by DIFFERENT validator sets share the table; V rows if the larger table does not fit.  No table at all: the generic
        // per-signature kernel
        for (uint32_t rows : {b->key_rows, V}) {
            void* q = nullptr;
            if (hipMalloc(&q, bsxk_keytable_bytes(rows)) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (hipMemset(q, 0, (size_t)rows * 64) != hipSuccess) { (void)hipFree(q); continue; }
            l.dallocs.push_back(q);
            l.d_keytab = static_cast<uint8_t*>(q);
            l.kc.init(V, rows);
            RET(halloc(l, (size_t)rows * sizeof(bsx_validator), &l.h_rowkeys));
            RET(dalloc(l, (size_t)rows * sizeof(bsx_validator), &l.d_rowkeys));
            RET(halloc(l, (size_t)M * V * 4, &l.h_rows));
            RET(dalloc(l, (size_t)M * V * 4, &l.d_rows));
            break;
        }
        return BSX_OK;
    }

    int launch(Lane& l, uint32_t R) override {
        const uint32_t J = b->J, B = b->B, V = b->V;
        const uint64_t hpr = b->hpr, nh = (uint64_t)R * hpr, n = (uint64_t)R * V;
        const RangeOut O(M);
        const RangeSmall S(M, V);
        bsx_ctx* ctx = b->ctx;
        hipStream_t st = l.st, sb = l.sb, s3 = l.s3, s4 = l.s4;
        hipEvent_t ev_c = l.ev[0], ev_d = l.ev[1], ev_a = l.ev[2], ev_f = l.ev[3], ev_b = l.ev[5];
        struct Drain { Lane& l; ~Drain() { for (hipStream_t s : {l.s4, l.s3, l.sb, l.st}) (void)hipStreamSynchronize(s); } } drain{l};
        auto* tv = reinterpret_cast<const bsx_validator*>(l.d_tv);
        auto* rv = tv + n;                                                  // right behind the R target sets: one tally launch takes both
        auto* ranges = reinterpret_cast<bsx_shared_ctx*>(l.d_ranges);
        auto* cres = reinterpret_cast<bsx_commit_result*>(l.d_out + O.commit);
        auto* tres = cres + R;
        await_slots(l, R);
        l.t_staged = now_ns();
        // the fixed-key table against this batch's keys, on the host: rows follow the FIRST request's validator set; slots of the
        // other requests whose key differs are counted (the signature check sizes — or skips — its generic-kernel pass from the count)
        const bsx_validator* h_tv = reinterpret_cast<const bsx_validator*>(l.h_small + S.tv);
        uint64_t n_mismatch = 0;
        bool rows_identity = true;
        if (l.d_keytab) {
            rows_identity = l.kc.assign(h_tv, R, reinterpret_cast<uint32_t*>(l.h_rows), l.kc_dirty, &n_mismatch);
            for (uint32_t q : l.kc_dirty) memcpy(l.h_rowkeys + (size_t)q * sizeof(bsx_validator), &l.kc.keys[(size_t)q * 32], 32);
        }
        // A launch set of FEW requests is a latency problem: its commit check spreads over three streams (R decoding ‖ challenges ‖
        // tallies), as in bsx_header_range.  From 4 requests on the headers' upload (>= 4 MB) is the longest piece of the hashing chain,
        // the commit chain fits behind it on ONE stream, and every stream and event less is front-end time the other lanes' sets get
        const bool wide = R < 4;
        hipStream_t q3 = wide ? s3 : sb, q4 = wide ? s4 : sb;
        // small inputs first: the commit chain starts from them
        LHIP(hipMemcpyAsync(l.d_ranges, l.h_small + S.ranges, (size_t)M * (sizeof(bsx_shared_ctx) + 8), hipMemcpyHostToDevice, st));
        LHIP(hipMemcpyAsync(l.d_tv, l.h_small + S.tv, (size_t)n * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
        LHIP(hipMemcpyAsync(l.d_tv + (size_t)n * sizeof(bsx_validator), l.h_small + S.rv, (size_t)n * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
        LHIP(hipMemsetAsync(l.d_out + O.hst, 0, (size_t)M * 8, st));              // header + hint status words of every slot
        if (l.d_keytab && !l.kc_dirty.empty())                                    // new keys: their rows' key records (the build compares against them)
            LHIP(hipMemcpyAsync(l.d_rowkeys, l.h_rowkeys, (size_t)l.kc.N * sizeof(bsx_validator), hipMemcpyHostToDevice, st));
        if (l.d_keytab && !rows_identity) LHIP(hipMemcpyAsync(l.d_rows, l.h_rows, (size_t)n * 4, hipMemcpyHostToDevice, st));
        LHIP(hipEventRecord(ev_c, st));
        // The headers: ONE copy per run of consecutive staged slots (normally one run = the whole batch).  Measured (tools/h2d_bench.hip):
        // page-locked copies of 1 MB reach 37 GB/s, of 4 MB 50, of 16 MB 55 of the 57 GB/s this PCIe link gives.  Round 6 tried the other
        // order — every SUBMITTER enqueues the upload of its own slot on a lane-owned upload stream the moment it is staged, so that the
        // link works through the collecting window: K = 2 .. 16 callers lost 9-23 % (K = 16: 40.8 against 44.9 M headers/s; sets shrank
        // from 11.5 to 7.8 requests and the GPU phase did not get shorter), K = 64 gained 3 % — eleven 1 MB copies cost more link time
        // than one 12 MB copy wins by starting 0.1 ms early (profiles/r6_batcher_upload_ab.txt).  A request whose headers are
        // page-locked caller memory (BSX_SUBMIT_INPUTS_STAY) is uploaded from there; packed requests below.
        uint32_t n_packed = 0, packed_end = 0;
        for (uint32_t r = 0; r < R;) {
            const Req& rq = l.reqs[r];
            uint8_t* dst = l.d_headers + (size_t)r * hpr * sizeof(bsx_header);
            if (rq.packed_bytes) {                                          // laid out as records by k_unpack_headers below
                n_packed++;
                if (rq.packed_off + rq.packed_bytes > packed_end) packed_end = rq.packed_off + rq.packed_bytes;
                r++;
                continue;
            }
            if (rq.direct) {
                const uint32_t had = l.dev_hwm[r];
                LHIP(hipMemcpyAsync(dst, rq.direct, (size_t)rq.n_headers * sizeof(bsx_header), hipMemcpyHostToDevice, st));
                if (had > rq.n_headers) LHIP(hipMemsetAsync(dst + (size_t)rq.n_headers * sizeof(bsx_header), 0, (size_t)(had - rq.n_headers) * sizeof(bsx_header), st));
                l.dev_hwm[r] = rq.n_headers;
                r++;
                continue;
            }
            // the staging is zero behind each request's headers: copying whole slots also clears what longer requests left on the device
            // (the last slot of the run only as far as it — or its predecessor in that slot — reaches)
            uint32_t e = r;
            while (e + 1 < R && !l.reqs[e + 1].direct && !l.reqs[e + 1].packed_bytes) e++;
            const uint32_t last_n = l.reqs[e].n_headers > l.dev_hwm[e] ? l.reqs[e].n_headers : l.dev_hwm[e];
            LHIP(hipMemcpyAsync(dst, l.h_headers + (size_t)r * hpr * sizeof(bsx_header), ((size_t)(e - r) * hpr + last_n) * sizeof(bsx_header), hipMemcpyHostToDevice, st));
            for (uint32_t q = r; q <= e; q++) l.dev_hwm[q] = l.reqs[q].n_headers;
            r = e + 1;
        }
        if (n_packed) {
            // the packed blocks cross PCIe as they are (~408 instead of 512 bytes per header) in ONE copy; the records are laid out in HBM
            uint32_t* desc = reinterpret_cast<uint32_t*>(l.h_desc);
            uint32_t* wipe = desc + 2 * (size_t)M;
            for (uint32_t r = 0; r < R; r++) {
                const Req& rq = l.reqs[r];
                desc[2 * r] = rq.packed_bytes ? rq.packed_off : 0xffffffffu;
                desc[2 * r + 1] = rq.n_headers;
                wipe[r] = l.dev_hwm[r];
                if (rq.packed_bytes) l.dev_hwm[r] = rq.n_headers;
            }
            LHIP(hipMemcpyAsync(l.d_packed, l.h_packed, packed_end, hipMemcpyHostToDevice, st));
            LHIP(hipMemcpyAsync(l.d_desc, l.h_desc, (size_t)M * 12, hipMemcpyHostToDevice, st));
            LHIP(bsxk_unpack_headers(st, l.d_packed, reinterpret_cast<const uint32_t*>(l.d_desc), R, (uint32_t)hpr,
                                     reinterpret_cast<const uint32_t*>(l.d_desc) + 2 * (size_t)M, reinterpret_cast<bsx_header*>(l.d_headers)));
        }
        // st: header hashes + both inclusion-proof paths of every header of every request; a malformed header marks ITS request.  The
        // hash of every request's target header becomes its ctx.end_header_hash, the first output half and what the signed messages must
        // carry (builder.skip, header_range.rs:42-48): stored by the lane that joins that header's tree (per-range tap)
        const bsxk_merkle_tap tap{~0ull, nullptr, nullptr, ranges, hpr, l.d_th};
        LHIP(bsxk_header_merkle(st, reinterpret_cast<const bsx_header*>(l.d_headers), nh, l.d_hashes, l.d_dh, l.d_lb, l.d_paths,
                                reinterpret_cast<uint32_t*>(l.d_out + O.hst), 0, 0, &tap, hpr));
        LHIP(hipEventRecord(ev_a, st));
        // commit chain: R decoded strictly ahead of time (needs only the validator records), the challenges, validator-set hash + total
        // power of the target AND the trusted sets (one launch over 2 R sets)
        LHIP(hipStreamWaitEvent(sb, ev_c, 0));
        if (wide) { LHIP(hipStreamWaitEvent(s3, ev_c, 0)); LHIP(hipStreamWaitEvent(s4, ev_c, 0)); }
        if (l.d_keytab) LHIP(bsxk_ed25519_decode_r(q3, tv, n, l.d_rdec));
        if (wide) LHIP(hipEventRecord(ev_d, s3));
        LHIP(bsxk_sha512_challenge(sb, tv, n, l.d_h, nullptr, V, nullptr));
        LHIP(bsxk_commit_tally(q4, tv, 2 * R, V, nullptr, nullptr, cres, nullptr));
        if (wide) LHIP(hipEventRecord(ev_f, s4));
        if (l.d_keytab) {
            // rows whose key changed are rebuilt (k_keytable_check compares every row's record with the key it should hold and rebuilds
            // the ones that differ; an unchanged table launches nothing)
            if (!l.kc_dirty.empty()) LHIP(bsxk_ed25519_keytable(sb, reinterpret_cast<const bsx_validator*>(l.d_rowkeys), l.kc.N, l.d_keytab));
            if (wide) LHIP(hipStreamWaitEvent(sb, ev_d, 0));
            LHIP(bsxk_ed25519_verify_keyed(sb, tv, l.d_h, n, V, l.d_keytab, l.kc.N, ctx->btab, l.d_ok, nullptr, l.d_rdec, (int64_t)n_mismatch,
                                           rows_identity ? nullptr : reinterpret_cast<const uint32_t*>(l.d_rows)));
        } else {
            LHIP(bsxk_ed25519_verify(sb, tv, l.d_h, n, l.d_ok));
        }
        LHIP(hipStreamWaitEvent(sb, ev_a, 0));                              // target hashes (dense) and the ranges' header hashes
        if (wide) LHIP(hipStreamWaitEvent(sb, ev_f, 0));
        LHIP(bsxk_commit_sums(sb, tv, R, V, l.d_th, l.d_ok, cres, nullptr));
        LHIP(bsxk_skip_check(sb, R, V, ranges, reinterpret_cast<const bsx_header*>(l.d_headers), hpr, l.d_hashes, tv, rv, l.d_ok, cres, tres,
                             reinterpret_cast<uint32_t*>(l.d_out + O.skip), l.d_th2, nullptr, b->cfg.chain_id_len ? b->cfg.chain_id : nullptr,
                             b->cfg.chain_id_len, nullptr));
        LHIP(hipEventRecord(ev_b, sb));
        // st: prove_data_commitment (header_range.rs:50-55): the hint of every map job, prove_subchain, reduce + final assertions
        LHIP(bsxk_assemble_inputs(st, R, J, B, 0, J, B, ranges, reinterpret_cast<const uint64_t*>(l.d_latest), reinterpret_cast<const bsx_header*>(l.d_headers),
                                  hpr, 0, l.d_hashes, l.d_dh, l.d_lb, l.d_compact, reinterpret_cast<uint32_t*>(l.d_out + O.ast), l.d_paths, ctx->zero_paths,
                                  0, nullptr, 1));
        LHIP(bsxk_prove_subchain(st, R, B, J, ranges, l.d_compact, reinterpret_cast<bsx_subchain*>(l.d_records), BSX_SUBCHAIN_PATHS_FROM_HINT));
        LHIP(bsxk_reduce_finalize(st, R, J, reinterpret_cast<const bsx_subchain*>(l.d_records), reinterpret_cast<bsx_subchain*>(l.d_out + O.res), nullptr, J, B,
                                  ranges, l.d_th, l.d_out + O.o64, reinterpret_cast<uint32_t*>(l.d_out + O.fst)));
        LHIP(hipStreamWaitEvent(st, ev_b, 0));
        LHIP(hipMemcpyAsync(l.h_out, l.d_out, O.total, hipMemcpyDeviceToHost, st));
        l.t_enqueued = now_ns();
        LHIP(hipStreamSynchronize(st));
        return BSX_OK;
    }

    static int say(char* msg, int rc, const char* text) {
        strncpy(msg, text, 235);
        msg[235] = 0;
        return rc;
    }
    int consume_one(Lane& l, uint32_t r, char* msg) override {
        const RangeOut O(M);
        Req& rq = l.reqs[r];
        uint32_t hs, as, fst, skip;
        memcpy(&hs, l.h_out + O.hst + 4 * (size_t)r, 4);
        memcpy(&as, l.h_out + O.ast + 4 * (size_t)r, 4);
        memcpy(&fst, l.h_out + O.fst + 4 * (size_t)r, 4);
        memcpy(&skip, l.h_out + O.skip + 4 * (size_t)r, 4);
        bsx_commit_result cr;
        memcpy(&cr, l.h_out + O.commit + sizeof cr * (size_t)r, sizeof cr);
        // the order of bsx_header_range: malformed inputs first, then the skip verification, then the data commitment's assertions
        if (hs & 1u) return say(msg, BSX_ERR_BAD_HEADER, "a packed header violates the field-size rules of bsx_header");
        if (as & 2u) return say(msg, BSX_ERR_BAD_HEADER, "an inclusion-proof leaf is not 34 / 72 bytes (circuits/input.rs:173,190)");
        if (as & 4u) return say(msg, BSX_ERR_BAD_ARG, "the supplied headers do not cover [start, min(end, latest-2)] or latest < 2");
        memcpy(rq.output64, l.h_out + O.o64 + 64 * (size_t)r, 64);
        if (rq.out_commit) *rq.out_commit = cr;
        if (skip) {
            snprintf(msg, 236, "skip verification failed: %s (bad signatures %u, first %u; bad messages %u; signed %llu of %llu; trusted overlap %llu)",
                     bsx_status_str((int)skip), cr.n_bad_signature, cr.first_bad_signature, cr.n_bad_message, (unsigned long long)cr.signed_power,
                     (unsigned long long)cr.total_power, (unsigned long long)cr.trusted_signed_power);
            return (int)skip;
        }
        if (fst) {
            snprintf(msg, 236, "prove_data_commitment: assertion mask 0x%x (A7 builder.rs:292-297, A8 :350-355, A9 :401-406; A1-A6 from the map jobs)", fst);
            return BSX_ERR_ASSERT;
        }
        return BSX_OK;
    }
    // launch() commits host-side state (key-cache rows, the dirty list, dev_hwm) while it enqueues; a launch that failed half
    // way leaves the device behind that state.  Start over: no row holds a key, every device slot may hold a full-length tail
    void on_launch_failed(Lane& l) override {
        if (l.d_keytab) {
            l.kc.init(b->V, l.kc.N);
            l.kc_dirty.clear();
            (void)hipMemset(l.d_keytab, 0, (size_t)l.kc.N * 64);
            (void)hipMemset(l.d_rowkeys, 0, (size_t)l.kc.N * sizeof(bsx_validator));
            memset(l.h_rowkeys, 0, (size_t)l.kc.N * sizeof(bsx_validator));
        }
        for (auto& h : l.dev_hwm) h = (uint32_t)b->hpr;
        (void)hipGetLastError();
    }
};

// ================================================================================================ the hint (data_commitment_inputs) / a map job
// One request = the hint of one map job (data_commitment.rs:22-44 -> input.rs:149-271) and, for a map-job request, prove_subchain on
// what the hint returned (the whole map closure, builder.rs:305-336).  Per request: ctx (bsx_shared_ctx), the job index inside it and
// the span; a plain hint is "job 0 of a range that starts at start_block".
struct HintKind : Kind {
    uint32_t hpr;                              // B + 1 headers per slot
    size_t img_bytes;                          // what a request gets back of its compact image: ctx/start/end hashes + both proof arrays
    HintKind(bsx_batcher* b_) : Kind(b_, "data_commitment_inputs", b_->M < 64 ? 64 : b_->M), hpr(b_->B + 1), img_bytes(bsx_off_slots(b_->B)) {}
    struct Small {
        size_t ranges, latest, spans, jobs, total;
        explicit Small(uint32_t M) { ranges = 0; latest = (size_t)M * 80; spans = latest + (size_t)M * 8; jobs = spans + (size_t)M * 4; total = jobs + (size_t)M * 4; }
    };
    struct Out {                               // page-locked results; the device block holds everything but `img` at the same offsets - expected
        size_t img, expected, hst, ast, records, total;
        Out(uint32_t M, size_t ib) {
            img = 0; expected = ((size_t)M * ib + 255) & ~(size_t)255; hst = expected + (size_t)M * 32; ast = hst + (size_t)M * 4;
            records = (ast + (size_t)M * 4 + 255) & ~(size_t)255; total = records + (size_t)M * sizeof(bsx_subchain);
        }
    };
    int lane_init(Lane& l) override {
        const uint64_t nh = (uint64_t)M * hpr;
        const bsx_witness_layout L = bsx_map_layout(b->B);
        const Small S(M);
        const Out O(M, img_bytes);
        RET(halloc(l, nh * sizeof(bsx_header), &l.h_headers));
        RET(halloc(l, S.total, &l.h_small));
        RET(halloc(l, O.total, &l.h_out));
        RET(dalloc(l, nh * sizeof(bsx_header), &l.d_headers));
        RET(dalloc(l, nh * 32, &l.d_hashes));
        RET(dalloc(l, nh * 128, &l.d_dh));
        RET(dalloc(l, nh * 128, &l.d_lb));
        RET(dalloc(l, nh * BSX_HEADER_PATH_BYTES, &l.d_paths));
        RET(dalloc(l, S.total, &l.d_ranges));                 // ranges, latest, spans, jobs in one block (same offsets as the staging)
        RET(dalloc(l, (size_t)M * L.compact_stride, &l.d_compact));
        RET(dalloc(l, O.total - O.expected, &l.d_out));       // expected, hst, ast, records
        return BSX_OK;
    }
    int launch(Lane& l, uint32_t R) override {
        const uint32_t B = b->B;
        const bsx_witness_layout L = bsx_map_layout(B);
        const Small S(M);
        const Out O(M, img_bytes);
        hipStream_t st = l.st;
        struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
        const uint64_t nh = (uint64_t)R * hpr;
        uint8_t* d_exp = l.d_out;
        uint32_t* d_hst = reinterpret_cast<uint32_t*>(l.d_out + (O.hst - O.expected));
        uint32_t* d_ast = reinterpret_cast<uint32_t*>(l.d_out + (O.ast - O.expected));
        bsx_subchain* d_rec = reinterpret_cast<bsx_subchain*>(l.d_out + (O.records - O.expected));
        auto* ranges = reinterpret_cast<const bsx_shared_ctx*>(l.d_ranges + S.ranges);
        auto* spans = reinterpret_cast<const uint32_t*>(l.d_ranges + S.spans);
        auto* latest = reinterpret_cast<const uint64_t*>(l.d_ranges + S.latest);
        await_slots(l, R);
        l.t_staged = now_ns();
        bool want_expected = false, want_records = false;
        for (uint32_t r = 0; r < R; r++) { want_expected |= l.reqs[r].out_expected != nullptr; want_records |= l.reqs[r].want_record; }
        LHIP(hipMemcpyAsync(l.d_ranges, l.h_small, S.total, hipMemcpyHostToDevice, st));            // 96 bytes per slot: all M at once
        LHIP(hipMemsetAsync(d_hst, 0, (size_t)M * 8, st));
        LHIP(hipMemcpyAsync(l.d_headers, l.h_headers, nh * sizeof(bsx_header), hipMemcpyHostToDevice, st));
        // map-job requests take the slots' path digests from the header trees hashed here (the fused hint): prove_subchain then
        // computes the tuple leaf hashes and the commitment tree only — the proofs ARE this library's own
        LHIP(bsxk_header_merkle(st, reinterpret_cast<const bsx_header*>(l.d_headers), nh, l.d_hashes, l.d_dh, l.d_lb, want_records ? l.d_paths : nullptr, d_hst, 0, 0,
                                nullptr, hpr));
        // the hint's image is read back as it is: every byte of it is written (zero padding included), the bytes behind it are not ours
        LHIP(bsxk_assemble_inputs(st, R, 1, B, 0, 1, B, ranges, latest, reinterpret_cast<const bsx_header*>(l.d_headers), hpr, 0, l.d_hashes, l.d_dh, l.d_lb,
                                  l.d_compact, d_ast, want_records ? l.d_paths : nullptr, b->ctx->zero_paths, 0, spans, 1,
                                  reinterpret_cast<const uint32_t*>(l.d_ranges + S.jobs)));
        if (want_expected) LHIP(bsxk_expected_commitments(st, R, B, ranges, spans, latest, reinterpret_cast<const uint32_t*>(l.d_ranges + S.jobs), l.d_compact, d_exp));
        if (want_records) LHIP(bsxk_prove_subchain(st, R, B, 1, ranges, l.d_compact, d_rec, BSX_SUBCHAIN_PATHS_FROM_HINT));
        LHIP(hipMemcpy2DAsync(l.h_out + O.img, img_bytes, l.d_compact, L.compact_stride, img_bytes, R, hipMemcpyDeviceToHost, st));
        LHIP(hipMemcpyAsync(l.h_out + O.expected, l.d_out, O.total - O.expected, hipMemcpyDeviceToHost, st));
        l.t_enqueued = now_ns();
        LHIP(hipStreamSynchronize(st));
        return BSX_OK;
    }
    int consume_one(Lane& l, uint32_t r, char* msg) override {
        const uint32_t B = b->B;
        const Out O(M, img_bytes);
        Req& rq = l.reqs[r];
        uint32_t hs, as;
        memcpy(&hs, l.h_out + O.hst + 4 * (size_t)r, 4);
        memcpy(&as, l.h_out + O.ast + 4 * (size_t)r, 4);
        if (hs & 1u) return RangeKind::say(msg, BSX_ERR_BAD_HEADER, "a packed header violates the field-size rules of bsx_header");
        if (as & 2u) return RangeKind::say(msg, BSX_ERR_BAD_HEADER, "an inclusion-proof leaf is not 34 / 72 bytes (circuits/input.rs:173,190)");
        if (as & 4u) return RangeKind::say(msg, BSX_ERR_BAD_ARG, "the supplied headers do not cover [start, min(end, latest-2)] or latest < 2");
        const uint8_t* img = l.h_out + O.img + img_bytes * (size_t)r;
        if (rq.out_start) memcpy(rq.out_start, img + bsx_off_start_header(), 32);
        if (rq.out_end) memcpy(rq.out_end, img + bsx_off_end_header(), 32);
        if (rq.out_dh) memcpy(rq.out_dh, img + bsx_off_dh_proofs(B), (size_t)B * sizeof(bsx_data_hash_proof));
        if (rq.out_lb) memcpy(rq.out_lb, img + bsx_off_lb_proofs(B), (size_t)B * sizeof(bsx_last_block_id_proof));
        if (rq.out_expected) memcpy(rq.out_expected, l.h_out + O.expected + 32 * (size_t)r, 32);
        if (rq.want_record) {
            memcpy(rq.out_record, l.h_out + O.records + sizeof(bsx_subchain) * (size_t)r, sizeof(bsx_subchain));
            if (rq.out_record->assert_fail) {
                snprintf(msg, 236, "prove_subchain: assertion mask 0x%x, first failing slot %u (A3 builder.rs:205-207, A4 :210-212, A5 :216-219, A6 :229-232)",
                         rq.out_record->assert_fail, rq.out_record->first_bad_slot);
                return BSX_ERR_ASSERT;
            }
        }
        return BSX_OK;
    }
};

// ================================================================================================ prove_subchain
struct SubchainKind : Kind {
    SubchainKind(bsx_batcher* b_) : Kind(b_, "prove_subchain", b_->M < 64 ? 64 : b_->M) {}
    int lane_init(Lane& l) override {
        const bsx_witness_layout L = bsx_map_layout(b->B);
        RET(halloc(l, (size_t)M * L.compact_stride, &l.h_headers));       // the requests' compact images
        RET(halloc(l, (size_t)M * sizeof(bsx_shared_ctx), &l.h_small));
        RET(halloc(l, (size_t)M * sizeof(bsx_subchain), &l.h_out));
        RET(dalloc(l, (size_t)M * L.compact_stride, &l.d_compact));
        RET(dalloc(l, (size_t)M * sizeof(bsx_shared_ctx), &l.d_ranges));
        RET(dalloc(l, (size_t)M * sizeof(bsx_subchain), &l.d_records));
        return BSX_OK;
    }
    int launch(Lane& l, uint32_t R) override {
        const bsx_witness_layout L = bsx_map_layout(b->B);
        hipStream_t st = l.st;
        struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{st};
        await_slots(l, R);
        l.t_staged = now_ns();
        LHIP(hipMemcpyAsync(l.d_compact, l.h_headers, (size_t)R * L.compact_stride, hipMemcpyHostToDevice, st));
        LHIP(hipMemcpyAsync(l.d_ranges, l.h_small, (size_t)R * sizeof(bsx_shared_ctx), hipMemcpyHostToDevice, st));
        // the proofs are the CALLER's: both paths are re-derived per slot (builder.rs:189-199 literally)
        LHIP(bsxk_prove_subchain(st, R, b->B, 1, reinterpret_cast<const bsx_shared_ctx*>(l.d_ranges), l.d_compact, reinterpret_cast<bsx_subchain*>(l.d_records), 0));
        LHIP(hipMemcpyAsync(l.h_out, l.d_records, (size_t)R * sizeof(bsx_subchain), hipMemcpyDeviceToHost, st));
        l.t_enqueued = now_ns();
        LHIP(hipStreamSynchronize(st));
        return BSX_OK;
    }
    int consume_one(Lane& l, uint32_t r, char* msg) override {
        Req& rq = l.reqs[r];
        memcpy(rq.out_record, l.h_out + sizeof(bsx_subchain) * (size_t)r, sizeof(bsx_subchain));
        if (rq.out_record->assert_fail) {
            snprintf(msg, 236, "prove_subchain: assertion mask 0x%x, first failing slot %u (A3 builder.rs:205-207, A4 :210-212, A5 :216-219, A6 :229-232)",
                     rq.out_record->assert_fail, rq.out_record->first_bad_slot);
            return BSX_ERR_ASSERT;
        }
        return BSX_OK;
    }
};

template <typename K> int kind_of(bsx_batcher* b, std::unique_ptr<Kind>& slot, Kind** out) {
    std::lock_guard<std::mutex> lk(b->mu_kinds);
    if (!slot) slot.reset(new K(b));
    const int rc = slot->start();
    if (rc != BSX_OK) return fail(rc, "%s", slot->init_err.c_str());
    *out = slot.get();
    return BSX_OK;
}

struct Inside {                                // a thread inside the batcher (see bsx_batcher_destroy)
    bsx_batcher* b;
    explicit Inside(bsx_batcher* b_) : b(b_) { b->inside.fetch_add(1, std::memory_order_acq_rel); }
    ~Inside() { b->inside.fetch_sub(1, std::memory_order_acq_rel); }
};

// stage `n` headers into slot `idx` of the lane (stride `hpr` headers); a slot that held more headers before gets its tail cleared
void stage_headers(Lane* l, uint32_t idx, uint64_t hpr, const bsx_header* src, uint64_t n) {
    uint8_t* dst = l->h_headers + (size_t)idx * hpr * sizeof(bsx_header);
    memcpy(dst, src, n * sizeof(bsx_header));
    if (l->slot_hwm[idx] > n) memset(dst + n * sizeof(bsx_header), 0, (l->slot_hwm[idx] - n) * sizeof(bsx_header));
    l->slot_hwm[idx] = (uint32_t)n;
}

}  // namespace

extern "C" {

const bsx_batcher_config* bsxb_config(const bsx_batcher* b) { return &b->cfg; }
// nothing of the header_range kind is collecting or in flight (a hint to the synchronous wrapper; racy by nature, harmless either way)
bool bsxb_range_idle(bsx_batcher* b) {
    Kind* k = b->range_kind.get();
    if (!k || !k->started.load(std::memory_order_acquire)) return true;
    if (k->in_flight.load(std::memory_order_relaxed) != 0) return false;
    Lane* o = k->open.load(std::memory_order_acquire);   // without the kind's lock: only its claim counter is looked at
    return !o || o->n_claimed.load(std::memory_order_relaxed) == 0;
}

int bsx_batcher_create(bsx_ctx* ctx, const bsx_batcher_config* cfg, bsx_batcher** out) {
    RET(bsxapi::use(ctx));
    if (!cfg || !out) return fail(BSX_ERR_BAD_ARG, "bsx_batcher_create: null pointer");
    *out = nullptr;
    if (!pow2(cfg->nb_map_jobs) || cfg->nb_map_jobs > 256) return fail(BSX_ERR_BAD_ARG, "NB_MAP_JOBS must be a power of two <= 256");
    if (!pow2(cfg->batch_size) || cfg->batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    if (cfg->v_max == 0 || (int)cfg->v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", cfg->v_max, bsxk_tally_vmax());
    if (cfg->chain_id_len > 50) return fail(BSX_ERR_BAD_ARG, "chain_id: at most 50 bytes");
    if (cfg->max_requests > 256 || cfg->n_lanes > 8 || cfg->window_us > 10000 || cfg->key_rows > 65536)
        return fail(BSX_ERR_BAD_ARG, "bsx_batcher_create: max_requests <= 256, n_lanes <= 8, window_us <= 10000, key_rows <= 65536");
    if (cfg->flags) return fail(BSX_ERR_BAD_ARG, "bsx_batcher_create: unknown flags 0x%x", cfg->flags);
    bsx_batcher* b = new bsx_batcher();
    b->ctx = ctx;
    b->cfg = *cfg;
    b->J = cfg->nb_map_jobs; b->B = cfg->batch_size; b->V = cfg->v_max;
    b->M = cfg->max_requests ? cfg->max_requests : 32;
    b->n_lanes = cfg->n_lanes ? cfg->n_lanes : 3;
    b->window_ns = (uint64_t)(cfg->window_us ? cfg->window_us : 50) * 1000;
    b->gap_ns = (uint64_t)bsx_knob("BSX_BATCH_GAP_NS", GAP_NS);
    b->hpr = (uint64_t)b->J * b->B + 1;
    b->key_rows = cfg->key_rows ? (cfg->key_rows < b->V ? b->V : cfg->key_rows) : 2 * b->V + 32;
    *out = b;
    return BSX_OK;
}

void bsx_batcher_destroy(bsx_batcher* b) {
    if (!b) return;
    // the workers finish what is in flight and fail what was only claimed (every ticket becomes final, every waiter is woken) ...
    for (auto* k : {&b->range_kind, &b->hint_kind, &b->subchain_kind})
        if (*k) (*k)->stop_workers();
    // ... the threads that were inside bsx_wait / bsx_poll / a submit leave (they used to be left on a deleted object) ...
    while (b->inside.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    // ... and only then do the lanes' blocks go
    for (auto* k : {&b->range_kind, &b->hint_kind, &b->subchain_kind})
        if (*k) (*k)->free_lanes();
    delete b;
}

// BSX_SUBMIT_INPUTS_STAY: the caller's input buffers stay valid until the ticket has been waited for (always so for the synchronous calls
// of a context with coalescing enabled): headers that lie in page-locked memory are then uploaded straight from there instead of through
// the staging.  BSX_SUBMIT_PACKED_HEADERS: `headers` is a packed block (wire.cpp), `n_headers` its size in bytes.
int bsx_packed_headers_check(const void* packed, uint64_t packed_bytes, uint64_t* out_n);
int bsx_submit_header_range(bsx_batcher* b, const uint8_t input48[48], const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                            uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                            uint8_t output64[64], bsx_commit_result* out_commit, bsx_ticket* out_ticket) {
    return bsx_submit_header_range_ex(b, input48, headers, first_height, n_headers, latest_block, target_validators, trusted_validators, output64, out_commit,
                                      out_ticket, 0);
}
int bsx_submit_header_range_ex(bsx_batcher* b, const uint8_t input48[48], const void* headers_, uint64_t first_height, uint64_t n_headers,
                               uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                               uint8_t output64[64], bsx_commit_result* out_commit, bsx_ticket* out_ticket, uint32_t flags) {
    if (!b || !out_ticket) return fail(BSX_ERR_BAD_ARG, "bsx_submit_header_range: null batcher / ticket");
    Inside in(b);
    if (!input48 || !headers_ || !target_validators || !trusted_validators || !output64) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (flags & ~(BSX_SUBMIT_INPUTS_STAY | BSX_SUBMIT_PACKED_HEADERS)) return fail(BSX_ERR_BAD_ARG, "bsx_submit_header_range_ex: unknown flags 0x%x", flags);
    const bool packed = (flags & BSX_SUBMIT_PACKED_HEADERS) != 0;
    const uint64_t packed_bytes = packed ? n_headers : 0;
    if (packed) {
        if (packed_bytes > 0xffffffffull) return fail(BSX_ERR_BAD_ARG, "packed header block larger than 4 GiB");
        const int rc = bsx_packed_headers_check(headers_, packed_bytes, &n_headers);
        if (rc != BSX_OK) return fail(rc, "bsx_submit_header_range_ex: the packed header block is inconsistent (sizes / offsets; bsx_pack_headers writes the format)");
    }
    // the checks bsx_header_range makes before it touches the device (header_range.rs:33-35: evm_read u64 big endian, bytes32, u64)
    uint64_t trusted_block = 0, target_block = 0;
    for (int i = 0; i < 8; i++) trusted_block = trusted_block << 8 | input48[i];
    for (int i = 0; i < 8; i++) target_block = target_block << 8 | input48[40 + i];
    const uint64_t span_max = (uint64_t)b->J * b->B;
    if (!(target_block > trusted_block) || target_block - trusted_block > span_max)
        return fail(BSX_ERR_RANGE_TOO_LONG, "skip: need trusted < target <= trusted + %llu", (unsigned long long)span_max);
    if (trusted_block < first_height || target_block - first_height >= n_headers) return fail(BSX_ERR_BAD_ARG, "trusted/target header not supplied");
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    const uint64_t h_first = trusted_block - first_height;
    uint64_t avail = n_headers - h_first;
    if (avail > b->hpr) avail = b->hpr;
    Kind* k = nullptr;
    RET(kind_of<RangeKind>(b, b->range_kind, &k));
    Req proto;
    proto.output64 = output64;
    proto.out_commit = out_commit;
    proto.n_headers = (uint32_t)avail;
    const bsx_header* h0 = packed ? nullptr : static_cast<const bsx_header*>(headers_) + h_first;
    // the request's part of a packed block: headers [h_first, h_first + avail) — their offsets rebased, then their bytes
    const uint8_t* pk = static_cast<const uint8_t*>(headers_);
    uint32_t pk_b0 = 0, pk_b1 = 0;
    size_t pk_data_at = 0;
    if (packed) {
        memcpy(&pk_b0, pk + 8 + 4 * h_first, 4);
        memcpy(&pk_b1, pk + 8 + 4 * (h_first + avail), 4);
        pk_data_at = (4 * (avail + 1) + 15) & ~(size_t)15;
        proto.packed_bytes = (uint32_t)(pk_data_at + (pk_b1 - pk_b0));
    } else if (flags & BSX_SUBMIT_INPUTS_STAY) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h0) == hipSuccess && at.type == hipMemoryTypeHost) proto.direct = h0;
        else (void)hipGetLastError();           // pageable memory is not an error: it is staged
    }
    Lane* l = nullptr;
    uint32_t idx = 0;
    RET(k->claim(&l, &idx, proto, out_ticket));
    const RangeSmall S(k->M, b->V);
    bsx_shared_ctx range{};
    range.start_block = trusted_block;
    range.end_block = target_block;
    memcpy(range.start_header_hash, input48 + 8, 32);
    memcpy(l->h_small + S.ranges + sizeof range * (size_t)idx, &range, sizeof range);
    memcpy(l->h_small + S.latest + 8 * (size_t)idx, &latest_block, 8);
    const size_t vbytes = (size_t)b->V * sizeof(bsx_validator);
    memcpy(l->h_small + S.tv + vbytes * idx, target_validators, vbytes);
    memcpy(l->h_small + S.rv + vbytes * idx, trusted_validators, vbytes);
    if (packed) {
        uint8_t* dst = l->h_packed + l->reqs[idx].packed_off;
        uint32_t* off = reinterpret_cast<uint32_t*>(dst);
        const uint64_t data0 = (8 + 4 * (n_headers + 1) + 15) & ~(uint64_t)15;
        for (uint64_t i = 0; i <= avail; i++) {
            uint32_t x;
            memcpy(&x, pk + 8 + 4 * (h_first + i), 4);
            off[i] = x - pk_b0;
        }
        memcpy(dst + pk_data_at, pk + data0 + pk_b0, pk_b1 - pk_b0);
    } else if (!proto.direct) {
        stage_headers(l, idx, b->hpr, h0, avail);
    }
    Kind::ready(l, idx);
    return BSX_OK;
}

int bsx_submit_data_commitment_inputs(bsx_batcher* b, const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                                      uint64_t start_block, uint64_t end_block, uint8_t out_start_header[32], uint8_t out_end_header[32],
                                      bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb, uint8_t out_expected_data_commitment[32],
                                      bsx_ticket* out_ticket) {
    if (!b || !out_ticket) return fail(BSX_ERR_BAD_ARG, "bsx_submit_data_commitment_inputs: null batcher / ticket");
    Inside in(b);
    if (!out_start_header || !out_end_header || !out_dh || !out_lb) return fail(BSX_ERR_BAD_ARG, "null output");
    if (end_block - start_block > (uint64_t)b->B) return fail(BSX_ERR_RANGE_TOO_LONG, "end - start > MAX_LEAVES (circuits/input.rs:154)");
    if (!headers || !n_headers) return fail(BSX_ERR_BAD_ARG, "no headers supplied");
    if (start_block < first_height || start_block - first_height >= n_headers)
        return fail(BSX_ERR_BAD_ARG, "header for start block %llu not supplied (first_height %llu, n %llu)", (unsigned long long)start_block, (unsigned long long)first_height, (unsigned long long)n_headers);
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    uint64_t avail = n_headers - (start_block - first_height);
    const uint64_t hpr = (uint64_t)b->B + 1;
    if (avail > hpr) avail = hpr;
    {   // the headers the hint reads: [start, req_end] (input.rs:160-198) — checked here, the slot's tail is zero headers
        const uint64_t lim = latest_block - 2, req_end = end_block < lim ? end_block : lim;
        if (start_block <= req_end && req_end - start_block >= avail)
            return fail(BSX_ERR_BAD_ARG, "the supplied headers do not cover [start, min(end, latest-2)] or latest < 2");
    }
    Kind* k = nullptr;
    RET(kind_of<HintKind>(b, b->hint_kind, &k));
    Req proto;
    proto.out_start = out_start_header; proto.out_end = out_end_header; proto.out_dh = out_dh; proto.out_lb = out_lb;
    proto.out_expected = out_expected_data_commitment;
    proto.n_headers = (uint32_t)avail;
    Lane* l = nullptr;
    uint32_t idx = 0;
    RET(k->claim(&l, &idx, proto, out_ticket));
    const HintKind::Small S(k->M);
    bsx_shared_ctx range{};
    range.start_block = start_block;
    range.end_block = end_block;
    const uint32_t span = (uint32_t)(end_block - start_block);
    memcpy(l->h_small + S.ranges + sizeof range * (size_t)idx, &range, sizeof range);
    memcpy(l->h_small + S.latest + 8 * (size_t)idx, &latest_block, 8);
    memcpy(l->h_small + S.spans + 4 * (size_t)idx, &span, 4);
    memset(l->h_small + S.jobs + 4 * (size_t)idx, 0, 4);
    stage_headers(l, idx, hpr, headers + (start_block - first_height), avail);
    Kind::ready(l, idx);
    return BSX_OK;
}

int bsx_submit_map_job(bsx_batcher* b, const bsx_shared_ctx* range, uint32_t job_index, const bsx_header* headers, uint64_t first_height,
                       uint64_t n_headers, uint64_t latest_block, uint8_t out_start_header[32], uint8_t out_end_header[32], bsx_data_hash_proof* out_dh,
                       bsx_last_block_id_proof* out_lb, bsx_subchain* out_record, bsx_ticket* out_ticket) {
    if (!b || !out_ticket) return fail(BSX_ERR_BAD_ARG, "bsx_submit_map_job: null batcher / ticket");
    Inside in(b);
    if (!range || !out_record) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (job_index >= b->J) return fail(BSX_ERR_BAD_ARG, "job_index %u is not below NB_MAP_JOBS = %u", job_index, b->J);
    if (!headers || !n_headers) return fail(BSX_ERR_BAD_ARG, "no headers supplied");
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    const uint64_t bs = range->start_block + (uint64_t)job_index * b->B, be = bs + b->B;      // builder.rs:315-322
    const uint64_t hpr = (uint64_t)b->B + 1;
    uint64_t avail = 0;
    {   // the headers the hint reads: [batch_start, req_end] (input.rs:160-198); a batch behind the chain head's reach reads none
        const uint64_t lim = latest_block - 2, req_end = be < lim ? be : lim;
        if (bs <= req_end) {
            if (bs < first_height || bs - first_height >= n_headers) return fail(BSX_ERR_BAD_ARG, "header for batch start %llu not supplied", (unsigned long long)bs);
            avail = n_headers - (bs - first_height);
            if (avail > hpr) avail = hpr;
            if (req_end - bs >= avail) return fail(BSX_ERR_BAD_ARG, "the supplied headers do not cover [start, min(end, latest-2)] or latest < 2");
        }
    }
    Kind* k = nullptr;
    RET(kind_of<HintKind>(b, b->hint_kind, &k));
    Req proto;
    proto.out_start = out_start_header; proto.out_end = out_end_header; proto.out_dh = out_dh; proto.out_lb = out_lb;
    proto.out_record = out_record;
    proto.want_record = true;
    proto.n_headers = (uint32_t)avail;
    Lane* l = nullptr;
    uint32_t idx = 0;
    RET(k->claim(&l, &idx, proto, out_ticket));
    const HintKind::Small S(k->M);
    const uint32_t span = b->B;
    memcpy(l->h_small + S.ranges + sizeof *range * (size_t)idx, range, sizeof *range);
    memcpy(l->h_small + S.latest + 8 * (size_t)idx, &latest_block, 8);
    memcpy(l->h_small + S.spans + 4 * (size_t)idx, &span, 4);
    memcpy(l->h_small + S.jobs + 4 * (size_t)idx, &job_index, 4);
    stage_headers(l, idx, hpr, avail ? headers + (bs - first_height) : headers, avail);
    Kind::ready(l, idx);
    return BSX_OK;
}

int bsx_submit_prove_subchain(bsx_batcher* b, const uint8_t start_header[32], const uint8_t end_header[32], const bsx_data_hash_proof* dh,
                              const bsx_last_block_id_proof* lb, uint64_t batch_start_block, uint64_t batch_end_block, uint64_t global_end_block,
                              const uint8_t global_end_header_hash[32], bsx_subchain* out_record, bsx_ticket* out_ticket) {
    if (!b || !out_ticket) return fail(BSX_ERR_BAD_ARG, "bsx_submit_prove_subchain: null batcher / ticket");
    Inside in(b);
    if (!start_header || !end_header || !dh || !lb || !global_end_header_hash || !out_record) return fail(BSX_ERR_BAD_ARG, "null pointer");
    Kind* k = nullptr;
    RET(kind_of<SubchainKind>(b, b->subchain_kind, &k));
    Req proto;
    proto.out_record = out_record;
    Lane* l = nullptr;
    uint32_t idx = 0;
    RET(k->claim(&l, &idx, proto, out_ticket));
    const uint32_t B = b->B;
    const bsx_witness_layout L = bsx_map_layout(B);
    // caller bytes -> compact witness image (no arithmetic: the kernel completes it), exactly as bsx_prove_subchain builds it
    uint8_t* img = l->h_headers + (size_t)idx * L.compact_stride;
    memset(img, 0, 128);
    memcpy(img + bsx_off_ctx_end_header(), global_end_header_hash, 32);
    memcpy(img + bsx_off_start_header(), start_header, 32);
    memcpy(img + bsx_off_end_header(), end_header, 32);
    memcpy(img + bsx_off_dh_proofs(B), dh, (size_t)B * sizeof *dh);
    memcpy(img + bsx_off_lb_proofs(B), lb, (size_t)B * sizeof *lb);
    memset(img + bsx_off_slots(B), 0, L.compact_stride - bsx_off_slots(B));
    uint32_t* W = reinterpret_cast<uint32_t*>(img + L.off_words);
    W[BSX_W_CTX_END] = (uint32_t)global_end_block; W[BSX_W_CTX_END + 1] = (uint32_t)(global_end_block >> 32);
    W[BSX_W_BATCH_START] = (uint32_t)batch_start_block; W[BSX_W_BATCH_START + 1] = (uint32_t)(batch_start_block >> 32);
    W[BSX_W_BATCH_END] = (uint32_t)batch_end_block; W[BSX_W_BATCH_END + 1] = (uint32_t)(batch_end_block >> 32);
    bsx_shared_ctx range{};
    range.end_block = global_end_block;
    memcpy(range.end_header_hash, global_end_header_hash, 32);
    memcpy(l->h_small + sizeof range * (size_t)idx, &range, sizeof range);
    Kind::ready(l, idx);
    return BSX_OK;
}

namespace {
int expired(uint64_t ticket) { return fail(BSX_ERR_BAD_ARG, "ticket %llu has expired (more than %u requests ago)", (unsigned long long)ticket, RING); }

// the ticket's record, if it still is the ticket's: which lane, which of the lane's batches
bool locate(bsx_batcher* b, uint64_t ticket, DoneRec** out, Lane** lane, uint32_t* batch_no) {
    DoneRec& d = b->ring[ticket % RING];
    if ((d.st.load(std::memory_order_acquire) >> 2) != ticket) return false;
    *lane = d.lane;
    *batch_no = d.batch_no;
    std::atomic_thread_fence(std::memory_order_acquire);
    if ((d.st.load(std::memory_order_acquire) >> 2) != ticket) return false;    // re-assigned while the fields were read
    *out = &d;
    return true;
}
}  // namespace

int bsx_poll(bsx_batcher* b, bsx_ticket ticket, int* out_done) {
    if (!b || !out_done || !ticket) return fail(BSX_ERR_BAD_ARG, "bsx_poll: null pointer / ticket 0");
    if (ticket >= b->next_seq.load()) return fail(BSX_ERR_BAD_ARG, "bsx_poll: ticket %llu was never issued", (unsigned long long)ticket);
    Inside in(b);
    DoneRec* d = nullptr;
    Lane* l = nullptr;
    uint32_t bn = 0;
    if (!locate(b, ticket, &d, &l, &bn)) return expired(ticket);
    *out_done = 0;
    if ((d->st.load(std::memory_order_acquire) & 3) != 2) {
        if (l->done_count.load(std::memory_order_acquire) == bn) return BSX_OK;      // its batch is still collecting or on the GPU
        if (!l->kind->consume(*d, ticket)) return expired(ticket);                   // "done" means the outputs are where the caller wants them
    }
    *out_done = 1;
    return BSX_OK;
}

int bsx_wait(bsx_batcher* b, bsx_ticket ticket) {
    if (!b || !ticket) return fail(BSX_ERR_BAD_ARG, "bsx_wait: null batcher / ticket 0");
    if (ticket >= b->next_seq.load()) return fail(BSX_ERR_BAD_ARG, "bsx_wait: ticket %llu was never issued", (unsigned long long)ticket);
    Inside in(b);
    DoneRec* d = nullptr;
    Lane* l = nullptr;
    uint32_t bn = 0;
    if (!locate(b, ticket, &d, &l, &bn)) return expired(ticket);
    if ((d->st.load(std::memory_order_acquire) & 3) != 2) {
        // park on the lane's counter until the batch this ticket rides in has completed: a short spin first (a launch set takes 0.2-0.5 ms,
        // a futex sleep + wake ~50 us of it), then the futex.  Only this lane's completions wake this thread
        for (int i = 0; i < 256 && l->done_count.load(std::memory_order_acquire) == bn; i++) cpu_relax();
        while (l->done_count.load(std::memory_order_acquire) == bn) {
            l->waiters.fetch_add(1);                                                   // seq_cst: pairs with publish()
            if (l->done_count.load() == bn) futex_wait(&l->done_count, bn);
            l->waiters.fetch_sub(1);
        }
        if (!l->kind->consume(*d, ticket)) return expired(ticket);                     // this thread copies its own results out
    }
    const int rc = d->rc;
    char msg[sizeof d->msg];
    memcpy(msg, d->msg, sizeof msg);
    msg[sizeof msg - 1] = 0;
    // the record is re-assigned RING requests later: what was read above is this ticket's only if the record still carries it
    std::atomic_thread_fence(std::memory_order_acquire);
    if (d->st.load(std::memory_order_acquire) != ticket * 4 + 2) return expired(ticket);
    if (rc != BSX_OK) bsxapi::g_err = msg;
    return rc;
}

int bsx_batcher_cork(bsx_batcher* b, int on) {
    if (!b) return fail(BSX_ERR_BAD_ARG, "null batcher");
    b->corked.store(on ? 1 : 0);
    return BSX_OK;
}

int bsx_batcher_get_stats(bsx_batcher* b, bsx_batcher_stats* out) {
    if (!b || !out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    memset(out, 0, sizeof *out);
    const std::unique_ptr<Kind>* ks[3] = {&b->range_kind, &b->hint_kind, &b->subchain_kind};
    for (int i = 0; i < 3; i++) {
        if (!*ks[i]) continue;
        out->kind[i].batches = (*ks[i])->n_batches.load();
        out->kind[i].requests = (*ks[i])->n_requests.load();
        out->kind[i].max_batch = (*ks[i])->max_batch.load();
        out->kind[i].close_wait_ns = (*ks[i])->sum_close_wait_ns.load();
        out->kind[i].stage_wait_ns = (*ks[i])->sum_stage_wait_ns.load();
        out->kind[i].enqueue_ns = (*ks[i])->sum_enqueue_ns.load();
        out->kind[i].gpu_wait_ns = (*ks[i])->sum_gpu_wait_ns.load();
        out->kind[i].complete_ns = (*ks[i])->sum_complete_ns.load();
    }
    return BSX_OK;
}

}  // extern "C"
