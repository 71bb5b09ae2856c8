// api.hip — host side of libbsx.so: the C ABI declared in include/bsx.h.
//
// Mirrors the reference's builder/hint interface for the header_range path (same names, argument meaning and
// failure conditions — circuits/builder.rs:20-79, circuits/data_commitment.rs:18-45, circuits/input.rs:39-61,
// circuits/header_range.rs:32-59) with return codes instead of panics.  All arithmetic runs in the HIP kernels; the
// host code only validates arguments, moves caller bytes into the documented device layouts and turns device
// status words into bsx_status codes.  There is NO CPU compute path: without a GPU bsx_init fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bsx.h"
#include "../../include/bsx_layout.h"
#include "api_internal.h"

#include "kernels.h"

static_assert(sizeof(bsx_header) == 512, "bsx_header");
static_assert(sizeof(bsx_data_hash_proof) == 162 && sizeof(bsx_last_block_id_proof) == 200, "proofs");
static_assert(sizeof(bsx_shared_ctx) == 80 && sizeof(bsx_subchain) == 128, "records");
static_assert(sizeof(bsx_validator) == 256 && sizeof(bsx_commit_result) == 96, "commit");
static_assert(sizeof(bsx_witness_layout) == 40, "layout");
static_assert(sizeof(bsx_skip_eval) == 40, "skip eval");
static_assert(sizeof(bsx_batcher_config) == 96 && sizeof(bsx_batcher_stats) == 192 && sizeof(bsx_pipeline_timing_result) == 32 && sizeof(bsx_pipeline_timing_result2) == 64, "batcher / timing");
static_assert(sizeof(bsx_commit_fold) == 128 && sizeof(bsx_pipeline_config) == 112 && sizeof(bsx_calibration) == 80, "pipeline / fold / calibration");

namespace bsxapi {
thread_local std::string g_err;
thread_local bsx_arena* tl_arena = nullptr;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int use(bsx_ctx* ctx) {
    if (!ctx) return fail(BSX_ERR_BAD_ARG, "null context");
    HIPCHK(hipSetDevice(ctx->device));
    return BSX_OK;
}
}  // namespace bsxapi

namespace {
constexpr size_t BSX_VMM_RECYCLE_BLOCKS = 8;     // bsx_dev_free keeps at most this many blocks mapped for recycling ...
constexpr size_t BSX_VMM_RECYCLE_BYTES = 96ull << 30;   // ... and at most this many bytes (a third of the 288 GB)
using bsxapi::DBuf;
using bsxapi::fail;
using bsxapi::g_err;
using bsxapi::pow2;
using bsxapi::use;

// device tier: the caller's stream exactly as given (NULL = the HIP default stream, which is what PyTorch's default
// stream is), so bsx_dev_* calls are ordered with the caller's own work on that stream
hipStream_t S(bsx_ctx*, void* s) { return static_cast<hipStream_t>(s); }

}  // namespace

// HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.  A pipeline
// drives 2 main + 2 commit-check streams beside the context's own two, copy and exchange streams: with 4 queues the second
// chunk's stream lands on the first chunk's side-stream queue (measured: 3.2 -> 4.8 ms per step).  The variable is read when
// the HIP runtime initialises, i.e. at the first HIP call of the process.  The library does NOT touch the process environment
// (round 3's load-time constructor did): a host that wants the 16 queues calls bsx_prepare_process() before its first HIP call
// (or exports the variable itself), and bsx_pipeline_autotune reports how many queues the pool really got.
int bsx_prepare_process(void) {
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;            // the caller's choice stands
    return setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0 ? 1 : -1;
}

extern "C" {

uint32_t bsx_version(void) { return BSX_VERSION; }

int bsx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bsx_init(int device, bsx_ctx** out) {
    if (!out) return fail(BSX_ERR_BAD_ARG, "bsx_init: null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(BSX_ERR_NO_DEVICE, "no HIP device visible (%s); libbsx has no CPU fallback", e == hipSuccess ? "count == 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(BSX_ERR_BAD_ARG, "device %d out of range (0..%d)", device, n - 1);
    if (hipSetDevice(device) != hipSuccess) return fail(BSX_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    bsx_ctx* c = new bsx_ctx{};
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(BSX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_d, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_e, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_f, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_g, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_c, hipEventDisableTiming);
    // constants of the hint's zero-padded proofs (kernels_sha.hip k_zero_paths)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->zero_paths), 320);
    if (e == hipSuccess) e = bsxk_zero_paths(c->stream, c->zero_paths);
    // the fixed-key Ed25519 table of the base point B (16 x 32768 affine entries of 128 B = 64 MB; kernels_ed.hip), shared by
    // every keyed verification
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->btab), bsxk_ed25519_btable_bytes());
    if (e == hipSuccess) e = bsxk_ed25519_btable(c->stream, c->btab);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bsx_shutdown(c);                  // releases whatever was created
        return fail(BSX_ERR_HIP, "bsx_init: %s", hipGetErrorString(e));
    }
    *out = c;
    return BSX_OK;
}

void bsx_shutdown(bsx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (bsx_batcher* bt = ctx->batcher.exchange(nullptr)) bsx_batcher_destroy(bt);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->hr_exec) (void)hipGraphExecDestroy(ctx->hr_exec);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->ev_d) (void)hipEventDestroy(ctx->ev_d);
    if (ctx->ev_e) (void)hipEventDestroy(ctx->ev_e);
    if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
    if (ctx->ev_f) (void)hipEventDestroy(ctx->ev_f);
    if (ctx->ev_g) (void)hipEventDestroy(ctx->ev_g);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->zero_paths) (void)hipFree(ctx->zero_paths);
    if (ctx->keytab) (void)hipFree(ctx->keytab);
    if (ctx->btab) (void)hipFree(ctx->btab);
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    if (ctx->ev_c) (void)hipEventDestroy(ctx->ev_c);
    for (auto& b : ctx->vmm) { (void)hipMemUnmap(b.va, b.size); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.va, b.size); }
    for (auto& b : ctx->vmm_free) { (void)hipMemUnmap(b.va, b.size); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.va, b.size); }
    delete ctx;
}

const char* bsx_last_error(void) { return g_err.c_str(); }

const char* bsx_status_str(int s) {
    static const char* N[] = {"BSX_OK", "BSX_ERR_NO_DEVICE", "BSX_ERR_HIP", "BSX_ERR_BAD_ARG", "BSX_ERR_RANGE_TOO_LONG",
                              "BSX_ERR_BAD_HEADER", "BSX_ERR_ASSERT", "BSX_ERR_BAD_SIGNATURE", "BSX_ERR_VOTING_POWER", "BSX_ERR_UNSUPPORTED"};
    return (s >= 0 && s < 10) ? N[s] : "BSX_ERR_?";
}

int bsx_map_witness_layout(uint32_t batch_size, bsx_witness_layout* out) {
    if (!out || !pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "batch_size must be a power of two <= %d", BSX_MAX_BATCH);
    *out = bsx_map_layout(batch_size);
    return BSX_OK;
}
int bsx_reduce_witness_layout(bsx_witness_layout* out) {
    if (!out) return fail(BSX_ERR_BAD_ARG, "null out");
    *out = bsx_reduce_layout();
    return BSX_OK;
}
int bsx_commit_witness_layout(uint32_t v_max, bsx_witness_layout* out) {
    if (!out || v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_BAD_ARG, "v_max must be in 1..%d", bsxk_tally_vmax());
    *out = bsx_commit_layout(v_max);
    return BSX_OK;
}
int bsx_skip_witness_layout(uint32_t v_max, bsx_witness_layout* out) {
    if (!out || v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_BAD_ARG, "v_max must be in 1..%d", bsxk_tally_vmax());
    *out = bsx_skip_layout(v_max);
    return BSX_OK;
}
int bsx_step_witness_layout(bsx_witness_layout* out) {
    if (!out) return fail(BSX_ERR_BAD_ARG, "null out");
    *out = bsx_step_layout();
    return BSX_OK;
}
uint64_t bsx_header_range_witness_elements(uint32_t J, uint32_t B, uint32_t v_max) {
    if (!pow2(J) || J > 256 || !pow2(B) || B > BSX_MAX_BATCH || v_max == 0 || (int)v_max > bsxk_tally_vmax()) return 0;
    return (uint64_t)J * bsx_map_layout(B).n_elements + (uint64_t)(J - 1) * bsx_reduce_layout().n_elements + bsx_commit_layout(v_max).n_elements +
           bsx_skip_layout(v_max).n_elements;
}
uint64_t bsx_next_header_witness_elements(uint32_t v_max) {
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return 0;
    return bsx_commit_layout(v_max).n_elements + bsx_step_layout().n_elements;
}

// ------------------------------------------------------------------------------------------------ device tier
#define DEV_ENTER() RET(use(ctx))
#define HOST_ENTER()  \
    RET(use(ctx));    \
    bsxapi::ArenaScope arena_scope_(ctx)

int bsx_dev_alloc(bsx_ctx* ctx, uint64_t bytes, void** out_ptr) {
    DEV_ENTER();
    if (!out_ptr || !bytes) return fail(BSX_ERR_BAD_ARG, "bsx_dev_alloc: null out / zero size");
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mu);      // ctx->vmm (pipelines of several threads may share a context)
    *out_ptr = nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    size_t gran = 0;
    HIPCHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran < (2u << 20)) gran = 2u << 20;
    const size_t size = (bytes + gran - 1) / gran * gran;
    // a block this context released earlier (bsx_dev_free keeps them mapped): the smallest one that fits.  Repeated
    // reserve / map / unmap / free cycles of multi-GB ranges ended in GPU memory-access faults on ROCm 7.2 (round 4: the
    // fourth witness pipeline of one process, bench.py's range sweep; never with hipMalloc-backed buffers), so a virtual range
    // is mapped ONCE and recycled; bsx_trim / bsx_shutdown give the memory back.
    {
        size_t best = ctx->vmm_free.size();
        for (size_t i = 0; i < ctx->vmm_free.size(); i++)
            if (ctx->vmm_free[i].size >= size && (best == ctx->vmm_free.size() || ctx->vmm_free[i].size < ctx->vmm_free[best].size)) best = i;
        if (best != ctx->vmm_free.size()) {
            const bsx_vmm_block b = ctx->vmm_free[best];
            HIPCHK(hipMemsetAsync(b.va, 0, size, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            ctx->vmm_free.erase(ctx->vmm_free.begin() + (long)best);
            ctx->vmm.push_back(b);
            *out_ptr = b.va;
            return BSX_OK;
        }
    }
    bsx_vmm_block b{nullptr, size, {}};
    hipError_t ce = hipMemCreate(&b.handle, size, &prop, 0);
    if (ce != hipSuccess && !ctx->vmm_free.empty()) {
        // recycled blocks nobody holds (none of them fits this request) are what stands in the way: give them back and retry once
        (void)hipGetLastError();
        for (auto& f : ctx->vmm_free) { (void)hipMemUnmap(f.va, f.size); (void)hipMemRelease(f.handle); (void)hipMemAddressFree(f.va, f.size); }
        ctx->vmm_free.clear();
        ce = hipMemCreate(&b.handle, size, &prop, 0);
    }
    if (ce != hipSuccess) return fail(BSX_ERR_HIP, "bsx_dev_alloc(%llu): hipMemCreate: %s", (unsigned long long)bytes, hipGetErrorString(ce));
    hipError_t e = hipMemAddressReserve(&b.va, size, 1ull << 30, nullptr, 0);
    bool mapped = false;
    if (e == hipSuccess) { e = hipMemMap(b.va, size, 0, b.handle, 0); mapped = e == hipSuccess; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess(b.va, size, &acc, 1);
    if (e == hipSuccess) e = hipMemsetAsync(b.va, 0, size, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (mapped) (void)hipMemUnmap(b.va, size);
        (void)hipMemRelease(b.handle);
        if (b.va) (void)hipMemAddressFree(b.va, size);
        return fail(BSX_ERR_HIP, "bsx_dev_alloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    ctx->vmm.push_back(b);
    *out_ptr = b.va;
    return BSX_OK;
}

int bsx_dev_free(bsx_ctx* ctx, void* ptr) {
    DEV_ENTER();
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mu);
    for (size_t i = 0; i < ctx->vmm.size(); i++)
        if (ctx->vmm[i].va == ptr) {
            // the block stays registered until its teardown has succeeded: a failure here leaves it to bsx_shutdown
            const bsx_vmm_block b = ctx->vmm[i];
            HIPCHK(hipDeviceSynchronize());                 // nothing in flight may still use it when the next owner clears it
            ctx->vmm.erase(ctx->vmm.begin() + (long)i);
            ctx->vmm_free.push_back(b);                     // stays mapped: recycled by bsx_dev_alloc, released by bsx_trim / bsx_shutdown
            // the recycle list is bounded: at most BSX_VMM_RECYCLE_BLOCKS blocks / BSX_VMM_RECYCLE_BYTES bytes, the SMALLEST released first (the large images are the
            // ones whose re-mapping faulted on ROCm 7.2; a caller cycling through many shapes must not pin one image per shape)
            auto cached = [&] { size_t t = 0; for (auto& f : ctx->vmm_free) t += f.size; return t; };
            while (ctx->vmm_free.size() > BSX_VMM_RECYCLE_BLOCKS || (ctx->vmm_free.size() > 1 && cached() > BSX_VMM_RECYCLE_BYTES)) {
                size_t k = 0;
                for (size_t q = 1; q < ctx->vmm_free.size(); q++) if (ctx->vmm_free[q].size < ctx->vmm_free[k].size) k = q;
                const bsx_vmm_block f = ctx->vmm_free[k];
                (void)hipMemUnmap(f.va, f.size); (void)hipMemRelease(f.handle); (void)hipMemAddressFree(f.va, f.size);
                ctx->vmm_free.erase(ctx->vmm_free.begin() + (long)k);
            }
            return BSX_OK;
        }
    return fail(BSX_ERR_BAD_ARG, "bsx_dev_free: pointer was not returned by bsx_dev_alloc on this context");
}

int bsx_trim(bsx_ctx* ctx, uint64_t* freed_bytes) {
    if (!ctx) return fail(BSX_ERR_BAD_ARG, "null context");
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mu);
    RET(use(ctx));
    if (ctx->arena.depth) return fail(BSX_ERR_BAD_ARG, "bsx_trim inside a host-tier call");
    HIPCHK(hipDeviceSynchronize());
    uint64_t freed = 0;
    if (ctx->hr_exec) { (void)hipGraphExecDestroy(ctx->hr_exec); ctx->hr_exec = nullptr; }   // it holds arena / key-table addresses
    ctx->hr_seen = false;
    if (ctx->arena.base) { freed += ctx->arena.cap; (void)hipFree(ctx->arena.base); ctx->arena.base = nullptr; ctx->arena.cap = 0; }
    if (ctx->keytab) { freed += bsxk_keytable_bytes(ctx->keytab_rows); (void)hipFree(ctx->keytab); ctx->keytab = nullptr; ctx->keytab_rows = 0; }
    for (auto& b : ctx->vmm_free) {                         // recycled bsx_dev_alloc blocks nobody holds
        freed += b.size;
        (void)hipMemUnmap(b.va, b.size); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.va, b.size);
    }
    ctx->vmm_free.clear();
    if (freed_bytes) *freed_bytes = freed;
    return BSX_OK;
}

int bsx_set_tuning(bsx_ctx* ctx, uint32_t key, uint64_t value) {
    if (!ctx) return fail(BSX_ERR_BAD_ARG, "null context");
    switch (key) {
    case BSX_TUNE_MERKLE_WORKGROUPS:
        if (value > 0xffffffffull) return fail(BSX_ERR_BAD_ARG, "BSX_TUNE_MERKLE_WORKGROUPS: %llu out of range", (unsigned long long)value);
        ctx->merkle_wgs = (uint32_t)value;
        return BSX_OK;
    case BSX_TUNE_HOST_GRAPHS:
        ctx->graphs_enabled = value != 0;
        if (!ctx->graphs_enabled && ctx->hr_exec) { (void)hipGraphExecDestroy(ctx->hr_exec); ctx->hr_exec = nullptr; }
        ctx->hr_seen = false;
        return BSX_OK;
    default:
        return fail(BSX_ERR_BAD_ARG, "bsx_set_tuning: unknown key %u", key);
    }
}

int bsx_dev_header_merkle(bsx_ctx* ctx, void* stream, const bsx_header* d_headers, uint64_t n, uint8_t* d_hashes,
                          uint8_t* d_dh_aunts, uint8_t* d_lb_aunts, uint8_t* d_paths, uint32_t* d_status) {
    DEV_ENTER();
    if (n && !d_headers) return fail(BSX_ERR_BAD_ARG, "null headers");
    HIPCHK(bsxk_header_merkle(S(ctx, stream), d_headers, n, d_hashes, d_dh_aunts, d_lb_aunts, d_paths, d_status, ctx->merkle_wgs, 0));
    return BSX_OK;
}

int bsx_dev_assemble_inputs(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t nb_map_jobs, uint32_t batch_size,
                            uint32_t job_first, uint32_t job_count, uint32_t span, const bsx_shared_ctx* d_ranges,
                            const uint64_t* d_latest, const bsx_header* d_headers, uint64_t headers_per_range,
                            uint64_t header_first_rel, const uint8_t* d_hashes, const uint8_t* d_dh_aunts,
                            const uint8_t* d_lb_aunts, uint8_t* d_compact, uint32_t* d_status, const uint8_t* d_paths) {
    DEV_ENTER();
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "batch_size must be a power of two <= %d", BSX_MAX_BATCH);
    if (span > batch_size) return fail(BSX_ERR_RANGE_TOO_LONG, "end - start = %u > MAX_LEAVES = %u (input.rs:154)", span, batch_size);
    if (job_first + job_count > nb_map_jobs) return fail(BSX_ERR_BAD_ARG, "job slice [%u,%u) exceeds nb_map_jobs %u", job_first, job_first + job_count, nb_map_jobs);
    if (!d_ranges || !d_latest || !d_headers || !d_hashes || !d_dh_aunts || !d_lb_aunts || !d_compact) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_assemble_inputs(S(ctx, stream), n_ranges, nb_map_jobs, batch_size, job_first, job_count, span, d_ranges, d_latest,
                                d_headers, headers_per_range, header_first_rel, d_hashes, d_dh_aunts, d_lb_aunts, d_compact, d_status,
                                d_paths, ctx->zero_paths));
    return BSX_OK;
}

int bsx_dev_prove_subchain(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t batch_size, uint32_t job_count,
                           const bsx_shared_ctx* d_ranges, uint8_t* d_compact, bsx_subchain* d_records, uint32_t flags) {
    DEV_ENTER();
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "batch_size must be a power of two <= %d", BSX_MAX_BATCH);
    if (!d_ranges || !d_compact || !d_records) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (flags & ~(BSX_SUBCHAIN_PATHS_FROM_HINT | BSX_SUBCHAIN_SEPARATE_LAUNCHES)) return fail(BSX_ERR_BAD_ARG, "unknown flags 0x%x", flags);
    HIPCHK(bsxk_prove_subchain(S(ctx, stream), n_ranges, batch_size, job_count, d_ranges, d_compact, d_records, flags));
    return BSX_OK;
}

int bsx_dev_reduce(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t n, const bsx_subchain* d_records, bsx_subchain* d_out,
                   uint8_t* d_reduce_compact) {
    DEV_ENTER();
    if (!pow2(n) || n > 256) return fail(BSX_ERR_BAD_ARG, "reduce fan-in must be a power of two <= 256 (got %u)", n);
    if (!d_records || !d_out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_reduce(S(ctx, stream), n_ranges, n, d_records, n, 1, d_out, d_reduce_compact));
    return BSX_OK;
}

int bsx_dev_reduce_strided(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t n, const bsx_subchain* d_records,
                           uint64_t stride_range, uint64_t stride_record, bsx_subchain* d_out,
                   uint8_t* d_reduce_compact) {
    DEV_ENTER();
    if (!pow2(n) || n > 256) return fail(BSX_ERR_BAD_ARG, "reduce fan-in must be a power of two <= 256 (got %u)", n);
    if (!d_records || !d_out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_reduce(S(ctx, stream), n_ranges, n, d_records, stride_range, stride_record, d_out, d_reduce_compact));
    return BSX_OK;
}

int bsx_dev_finalize(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t nb_map_jobs, uint32_t batch_size,
                     const bsx_shared_ctx* d_ranges, const bsx_subchain* d_results, const uint8_t* d_target_hashes,
                     uint8_t* d_output64, uint32_t* d_status) {
    DEV_ENTER();
    if (!d_ranges || !d_results) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_finalize(S(ctx, stream), n_ranges, nb_map_jobs, batch_size, d_ranges, d_results, d_target_hashes, d_output64, d_status, nullptr, 0));
    return BSX_OK;
}

int bsx_dev_expand_witness(bsx_ctx* ctx, void* stream, const bsx_witness_layout* layout, uint32_t n_jobs, const uint8_t* d_compact,
                           uint64_t* d_witness) {
    DEV_ENTER();
    if (!layout || !d_compact || !d_witness) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (reinterpret_cast<uintptr_t>(d_witness) & 15) return fail(BSX_ERR_BAD_ARG, "witness buffer must be 16-byte aligned");
    HIPCHK(bsxk_expand_witness(S(ctx, stream), layout, n_jobs, d_compact, d_witness));
    return BSX_OK;
}

int bsx_dev_fill_end_hash(bsx_ctx* ctx, void* stream, uint32_t n_ranges, bsx_shared_ctx* d_ranges, const uint8_t* d_hashes,
                          uint64_t headers_per_range, const uint32_t* d_target_index, uint8_t* d_target_hashes, uint8_t* d_hashes_copy) {
    DEV_ENTER();
    if (!d_ranges || !d_hashes) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_fill_end_hash(S(ctx, stream), n_ranges, d_ranges, d_hashes, headers_per_range, d_target_index, d_target_hashes, d_hashes_copy, 0));
    return BSX_OK;
}

int bsx_dev_sha512_challenge(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint64_t n, uint8_t* d_h, uint8_t* d_digest) {
    DEV_ENTER();
    if (n && (!d_validators || !d_h)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_sha512_challenge(S(ctx, stream), d_validators, n, d_h, d_digest, 1, nullptr));
    return BSX_OK;
}

int bsx_dev_ed25519_verify(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n, uint8_t* d_ok) {
    DEV_ENTER();
    if (n && (!d_validators || !d_h || !d_ok)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_ed25519_verify(S(ctx, stream), d_validators, d_h, n, d_ok));
    return BSX_OK;
}

uint64_t bsx_ed25519_keytable_bytes(uint32_t n_keys) { return bsxk_keytable_bytes(n_keys); }
uint64_t bsx_ed25519_keytable_bytes_w(uint32_t n_keys, uint32_t digit_bits) { return bsxk_keytable_bits_ok((int)digit_bits) ? bsxk_keytable_bytes(n_keys, (int)digit_bits) : 0; }
uint64_t bsx_ed25519_verify_scratch_bytes(uint64_t n) { return bsxk_ed25519_scratch_bytes(n); }

int bsx_dev_ed25519_keytable_w(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_keys, void* d_table, uint32_t digit_bits) {
    DEV_ENTER();
    if (n_keys && (!d_validators || !d_table)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (((uintptr_t)d_table & 127) != 0) return fail(BSX_ERR_BAD_ARG, "key table must be 128-byte aligned (one cache line per entry)");
    if (!bsxk_keytable_bits_ok((int)digit_bits)) return fail(BSX_ERR_BAD_ARG, "key-table digit width %u: %d or %d", digit_bits, BSXK_KT_BITS, BSXK_KT_BITS_WIDE);
    HIPCHK(bsxk_ed25519_keytable(S(ctx, stream), d_validators, n_keys, static_cast<uint8_t*>(d_table), (int)digit_bits));
    return BSX_OK;
}
int bsx_dev_ed25519_keytable(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_keys, void* d_table) {
    return bsx_dev_ed25519_keytable_w(ctx, stream, d_validators, n_keys, d_table, BSXK_KT_BITS);
}

int bsx_dev_ed25519_verify_keyed(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n,
                                 uint32_t v_max, const void* d_table, uint32_t n_keys, uint8_t* d_ok, void* d_scratch) {
    return bsx_dev_ed25519_verify_keyed_w(ctx, stream, d_validators, d_h, n, v_max, d_table, n_keys, d_ok, d_scratch, BSXK_KT_BITS);
}
int bsx_dev_ed25519_verify_keyed_w(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n,
                                   uint32_t v_max, const void* d_table, uint32_t n_keys, uint8_t* d_ok, void* d_scratch, uint32_t digit_bits) {
    DEV_ENTER();
    if (!bsxk_keytable_bits_ok((int)digit_bits)) return fail(BSX_ERR_BAD_ARG, "key-table digit width %u: %d or %d", digit_bits, BSXK_KT_BITS, BSXK_KT_BITS_WIDE);
    if (n && (!d_validators || !d_h || !d_ok)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (v_max == 0) return fail(BSX_ERR_BAD_ARG, "v_max is 0");
    if (n_keys && !d_table) return fail(BSX_ERR_BAD_ARG, "null key table");
    if ((uintptr_t)d_table & 127) return fail(BSX_ERR_BAD_ARG, "key table must be 128-byte aligned (one cache line per entry)");
    if ((uintptr_t)d_scratch & 15) return fail(BSX_ERR_BAD_ARG, "scratch must be 16-byte aligned");
    HIPCHK(bsxk_ed25519_verify_keyed(S(ctx, stream), d_validators, d_h, n, v_max, static_cast<const uint8_t*>(d_table), n_keys, ctx->btab, d_ok, d_scratch, nullptr, -1,
                                     nullptr, (int)digit_bits));
    return BSX_OK;
}

uint64_t bsx_ed25519_decoded_r_bytes(uint64_t n) { return bsxk_ed25519_rdec_bytes(n); }

int bsx_dev_ed25519_decode_r(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint64_t n, void* d_decoded_r) {
    DEV_ENTER();
    if (n && (!d_validators || !d_decoded_r)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if ((uintptr_t)d_decoded_r & 15) return fail(BSX_ERR_BAD_ARG, "decoded-R buffer must be 16-byte aligned");
    HIPCHK(bsxk_ed25519_decode_r(S(ctx, stream), d_validators, n, d_decoded_r));
    return BSX_OK;
}

int bsx_dev_ed25519_verify_keyed_r(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, const uint8_t* d_h, uint64_t n,
                                   uint32_t v_max, const void* d_table, uint32_t n_keys, const void* d_decoded_r, uint8_t* d_ok) {
    DEV_ENTER();
    if (n && (!d_validators || !d_h || !d_ok || !d_decoded_r)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (v_max == 0) return fail(BSX_ERR_BAD_ARG, "v_max is 0");
    if (n_keys && !d_table) return fail(BSX_ERR_BAD_ARG, "null key table");
    if ((uintptr_t)d_table & 127) return fail(BSX_ERR_BAD_ARG, "key table must be 128-byte aligned (one cache line per entry)");
    HIPCHK(bsxk_ed25519_verify_keyed(S(ctx, stream), d_validators, d_h, n, v_max, static_cast<const uint8_t*>(d_table), n_keys, ctx->btab, d_ok, nullptr, d_decoded_r, -1));
    return BSX_OK;
}

int bsx_dev_skip_eval(bsx_ctx* ctx, void* stream, const bsx_validator* d_start_validators, const bsx_validator* d_candidate_validators,
                      uint32_t n_candidates, uint32_t v_max, bsx_skip_eval* d_out) {
    DEV_ENTER();
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    if (n_candidates && (!d_start_validators || !d_candidate_validators || !d_out)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_skip_eval(S(ctx, stream), d_start_validators, d_candidate_validators, n_candidates, v_max, d_out));
    return BSX_OK;
}

int bsx_dev_commit_tally(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_commits, uint32_t v_max,
                         const uint8_t* d_header_hashes, const uint8_t* d_ok, bsx_commit_result* d_results) {
    DEV_ENTER();
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    if (n_commits && (!d_validators || !d_results)) return fail(BSX_ERR_BAD_ARG, "null pointer");
    HIPCHK(bsxk_commit_tally(S(ctx, stream), d_validators, n_commits, v_max, d_header_hashes, d_ok, d_results, nullptr));
    return BSX_OK;
}

int bsx_dev_skip_check(bsx_ctx* ctx, void* stream, uint32_t n_ranges, uint32_t v_max, const bsx_shared_ctx* d_ranges,
                       const bsx_header* d_headers, uint64_t headers_per_range, const uint8_t* d_hashes, const bsx_validator* d_target,
                       const bsx_validator* d_trusted, const uint8_t* d_target_ok, bsx_commit_result* d_target_res,
                       const bsx_commit_result* d_trusted_res, uint32_t* d_skip_status, uint8_t* d_target_hashes,
                       const uint32_t* d_target_index, const uint8_t* chain_id, uint32_t chain_id_len) {
    DEV_ENTER();
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    if (!d_ranges || !d_headers || !d_hashes || !d_target || !d_trusted || !d_target_ok || !d_target_res || !d_trusted_res || !d_skip_status)
        return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (chain_id_len > 50 || (chain_id_len && !chain_id)) return fail(BSX_ERR_BAD_ARG, "chain_id: at most 50 bytes");
    HIPCHK(bsxk_skip_check(S(ctx, stream), n_ranges, v_max, d_ranges, d_headers, headers_per_range, d_hashes, d_target, d_trusted,
                           d_target_ok, d_target_res, d_trusted_res, d_skip_status, d_target_hashes, d_target_index, chain_id, chain_id_len, nullptr));
    return BSX_OK;
}

uint64_t bsx_dev_verify_commits_scratch_bytes(uint32_t n_commits, uint32_t v_max) {
    const uint64_t n = (uint64_t)n_commits * v_max;
    return ((n * 32 + 255) & ~255ull) + ((bsxk_ed25519_scratch_bytes(n) + 255) & ~255ull) + bsxk_commit_fold_scratch_bytes(n_commits);
}

int bsx_dev_verify_commits(bsx_ctx* ctx, void* stream, const bsx_validator* d_validators, uint32_t n_commits, uint32_t v_max,
                           const uint8_t* d_header_hashes, uint32_t first_index, void* d_keytable, void* d_scratch, uint8_t* d_ok,
                           bsx_commit_result* d_results, bsx_commit_fold* d_fold, uint8_t* d_commit_compact, uint32_t flags) {
    DEV_ENTER();
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    if (!n_commits || n_commits > BSX_COMMIT_FOLD_MAX) return fail(BSX_ERR_UNSUPPORTED, "n_commits %u not in 1..%u", n_commits, BSX_COMMIT_FOLD_MAX);
    if (!d_validators || !d_header_hashes || !d_keytable || !d_scratch || !d_ok || !d_results || !d_fold) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (((uintptr_t)d_keytable & 127) || ((uintptr_t)d_scratch & 255)) return fail(BSX_ERR_BAD_ARG, "key table must be 128-byte, scratch 256-byte aligned");
    if (flags & ~(BSX_COMMITS_KEYTABLE_READY | BSX_COMMITS_KEYS_UNIFORM | BSX_COMMITS_TALLY_BESIDE | BSX_COMMITS_KEYTABLE_WIDE)) return fail(BSX_ERR_BAD_ARG, "bsx_dev_verify_commits: unknown flags 0x%x", flags);
    hipStream_t st = S(ctx, stream);
    const uint64_t n = (uint64_t)n_commits * v_max;
    uint8_t* d_h = static_cast<uint8_t*>(d_scratch);
    uint8_t* d_ed = d_h + ((n * 32 + 255) & ~255ull);
    uint8_t* d_fs = d_ed + ((bsxk_ed25519_scratch_bytes(n) + 255) & ~255ull);
    if (d_commit_compact && ((uintptr_t)d_commit_compact & 15)) return fail(BSX_ERR_BAD_ARG, "d_commit_compact must be 16-byte aligned");
    const bsx_witness_layout CL = bsx_commit_layout(v_max);
    const bsxk_unit_dst cw = bsxk_unit(d_commit_compact, CL);
    const bool beside = (flags & BSX_COMMITS_TALLY_BESIDE) != 0;
    const int kt_bits = (flags & BSX_COMMITS_KEYTABLE_WIDE) ? BSXK_KT_BITS_WIDE : BSXK_KT_BITS;
    // an error return behind the side-stream launch must not leave the trees running on buffers the caller is about to reclaim
    struct SideGuard { hipStream_t s; bool armed; ~SideGuard() { if (armed) (void)hipStreamSynchronize(s); } } side{ctx->stream4, false};
    if (beside) {
        side.armed = true;
        // the trees on the context's side stream, ordered behind whatever the caller's stream holds (the previous step's readers of
        // d_results / the units) and joined in front of the sums
        hipStream_t s4 = ctx->stream4;
        HIPCHK(hipEventRecord(ctx->ev_f, st));
        HIPCHK(hipStreamWaitEvent(s4, ctx->ev_f, 0));
        HIPCHK(bsxk_commit_tally(s4, d_validators, n_commits, v_max, d_header_hashes, nullptr, d_results, d_commit_compact ? &cw : nullptr));
        HIPCHK(hipEventRecord(ctx->ev_g, s4));
    }
    HIPCHK(bsxk_sha512_challenge(st, d_validators, n, d_h, nullptr, v_max, d_commit_compact ? &cw : nullptr));
    if (!(flags & BSX_COMMITS_KEYTABLE_READY)) HIPCHK(bsxk_ed25519_keytable(st, d_validators, v_max, static_cast<uint8_t*>(d_keytable), kt_bits));
    HIPCHK(bsxk_ed25519_verify_keyed(st, d_validators, d_h, n, v_max, static_cast<const uint8_t*>(d_keytable), v_max, ctx->btab, d_ok, d_ed, nullptr,
                                     (flags & BSX_COMMITS_KEYS_UNIFORM) ? 0 : -1, nullptr, kt_bits));
    if (beside) {
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_g, 0));
        HIPCHK(bsxk_commit_sums(st, d_validators, n_commits, v_max, d_header_hashes, d_ok, d_results, d_commit_compact ? &cw : nullptr));
    } else {
        HIPCHK(bsxk_commit_tally(st, d_validators, n_commits, v_max, d_header_hashes, d_ok, d_results, d_commit_compact ? &cw : nullptr));
    }
    HIPCHK(bsxk_commit_fold(st, d_results, n_commits, first_index, d_fs, d_fold));
    side.armed = false;                      // everything is enqueued and joined on the caller's stream
    return BSX_OK;
}

// host-side bookkeeping for the fixed-key signature check (no arithmetic: 32-byte compares)
uint64_t bsxh_key_mismatches(const bsx_validator* v, uint64_t n_commits, uint32_t v_max) {
    uint64_t n = 0;
    for (uint64_t c = 1; c < n_commits; c++)
        for (uint32_t i = 0; i < v_max; i++) {
            const bsx_validator& x = v[c * v_max + i];
            if (x.enabled && x.is_signed && memcmp(x.pubkey, v[i].pubkey, 32) != 0) n++;
        }
    return n;
}

// ------------------------------------------------------------------------------------------------ host tier
// The host tier's fixed-key Ed25519 table lives in the context: a proof request verifies ONE commit of <= 100 signatures, for
// which the generic kernel is pure latency (256 dependent doublings per lane: 1.3 ms), while consecutive requests are signed
// by the same validator set — with the table kept, a request pays one key compare per row (0.011 ms) and the 32-doubling
// keyed kernel.  Rows are rebuilt only when a key changes (bsx_dev_ed25519_keytable).
static int ctx_keytable(bsx_ctx* ctx, uint32_t v_max, uint8_t** out, hipStream_t stream) {
    if (ctx->keytab_rows != v_max) {          // the table layout depends on the row count
        ctx->keytab_mirror_valid = false;
        if (ctx->keytab) (void)hipFree(ctx->keytab);
        ctx->keytab = nullptr;
        ctx->keytab_rows = 0;
        if (hipMalloc(reinterpret_cast<void**>(&ctx->keytab), bsxk_keytable_bytes(v_max)) != hipSuccess) {
            // 5.8 MB per validator slot (2.9 GB at v_max = 512) did not fit: the caller verifies with the generic
            // per-signature kernel instead (same accept set, no table)
            (void)hipGetLastError();
            ctx->keytab = nullptr;
            *out = nullptr;
            return BSX_OK;
        }
        HIPCHK(hipMemsetAsync(ctx->keytab, 0, (size_t)v_max * 64, stream));          // key records: nothing to reuse yet
        ctx->keytab_rows = v_max;
    }
    *out = ctx->keytab;
    return BSX_OK;
}
// signature check of n = n_commits * v_max slots: keyed (table of the first commit's keys, kept in the context) or, when the
// table could not be allocated, generic
static int ctx_verify(bsx_ctx* ctx, hipStream_t st, const bsx_validator* dv, const uint8_t* dh, uint64_t n, uint32_t v_max, uint8_t* dok,
                      void* dscratch, void* drdec, int64_t n_deferred) {
    uint8_t* tab = nullptr;
    RET(ctx_keytable(ctx, v_max, &tab, st));
    if (!tab) {
        HIPCHK(bsxk_ed25519_verify(st, dv, dh, n, dok));
        return BSX_OK;
    }
    ctx->keytab_mirror_valid = false;                       // rows now follow THIS call's validators (device memory: not mirrored)
    HIPCHK(bsxk_ed25519_keytable(st, dv, v_max, tab));
    // small batches (a proof request): R decoded ahead, 16 lanes per signature, projective comparison — no inversion on the chain
    if (drdec && !dscratch) HIPCHK(bsxk_ed25519_decode_r(st, dv, n, drdec));
    HIPCHK(bsxk_ed25519_verify_keyed(st, dv, dh, n, v_max, tab, v_max, ctx->btab, dok, dscratch, dscratch ? nullptr : drdec, n_deferred));
    return BSX_OK;
}

static int ctx_hstage(bsx_ctx* ctx, size_t bytes, uint8_t** out);
// Small results of a host-tier call: a D2H copy into the caller's (pageable) memory blocks the host until the data has
// arrived — one stream round trip per copy.  Staged: each copy lands in the context's page-locked staging buffer (truly
// asynchronous), ONE synchronisation, then plain memcpys to the caller's pointers.  Large results go direct.
struct StagedD2H {
    bsx_ctx* ctx;
    hipStream_t st;
    uint8_t* base = nullptr;
    size_t off = 0;
    struct Item { void* dst; size_t off, n; };
    std::vector<Item> items;
    static constexpr size_t CAP = 1u << 20, DIRECT_FROM = 256u << 10;
    StagedD2H(bsx_ctx* c, hipStream_t s) : ctx(c), st(s) {}
    int copy(void* dst, const void* dsrc, size_t n) {
        if (!n) return BSX_OK;
        const size_t a = (n + 63) & ~(size_t)63;
        if (n >= DIRECT_FROM || off + a > CAP) { HIPCHK(hipMemcpyAsync(dst, dsrc, n, hipMemcpyDeviceToHost, st)); return BSX_OK; }
        if (!base) RET(ctx_hstage(ctx, CAP, &base));
        HIPCHK(hipMemcpyAsync(base + off, dsrc, n, hipMemcpyDeviceToHost, st));
        items.push_back(Item{dst, off, n});
        off += a;
        return BSX_OK;
    }
    int sync() {
        HIPCHK(hipStreamSynchronize(st));
        for (const Item& it : items) memcpy(it.dst, base + it.off, it.n);
        items.clear();
        off = 0;
        return BSX_OK;
    }
};
#define H2D(dst, src, n) HIPCHK(hipMemcpyAsync((dst), (src), (n), hipMemcpyHostToDevice, st))
#define D2H(dst, src, n) HIPCHK(hipMemcpyAsync((dst), (src), (n), hipMemcpyDeviceToHost, st))
#define SYNC() HIPCHK(hipStreamSynchronize(st))
// Every stream a host-tier call launched on is drained before its ArenaScope rewinds the arena — also on the error paths: a
// side-stream kernel still reading or writing arena buffers after the rewind would race with the next call's buffers and events.
struct HostDrain {
    hipStream_t s[4];
    ~HostDrain() { for (int i = 3; i >= 0; i--) if (s[i]) (void)hipStreamSynchronize(s[i]); }
};

static int header_status_to_rc(uint32_t hs, uint32_t as) {
    if (hs & 1u) return fail(BSX_ERR_BAD_HEADER, "a packed header violates the field-size rules of bsx_header");
    if (as & 2u) return fail(BSX_ERR_BAD_HEADER, "an inclusion-proof leaf is not 34 / 72 bytes (circuits/input.rs:173,190)");
    if (as & 4u) return fail(BSX_ERR_BAD_ARG, "the supplied headers do not cover [start, min(end, latest-2)] or latest < 2");
    return BSX_OK;
}

// ---- coalescing (batcher.hip): a context with a batcher attached turns its synchronous calls of the batcher's shape into submit + wait
int bsx_enable_coalescing(bsx_ctx* ctx, const bsx_batcher_config* cfg) {
    if (!ctx) return fail(BSX_ERR_BAD_ARG, "null context");
    bsx_batcher* nb = nullptr;
    if (cfg) RET(bsx_batcher_create(ctx, cfg, &nb));
    if (bsx_batcher* old = ctx->batcher.exchange(nb)) bsx_batcher_destroy(old);     // callers must not be inside a coalesced call while it is replaced
    return BSX_OK;
}
bsx_batcher* bsx_context_batcher(bsx_ctx* ctx) { return ctx ? ctx->batcher.load() : nullptr; }
const bsx_batcher_config* bsxb_config(const bsx_batcher* b);                          // batcher.hip
bool bsxb_range_idle(bsx_batcher* b);

// page-lock caller memory for hosts that do not link HIP themselves (a Rust shim registers its header buffer ONCE and reuses it:
// every bsx_header_range / BSX_SUBMIT_INPUTS_STAY submit from it is then uploaded from where it lies, no staging copy)
int bsx_host_register(bsx_ctx* ctx, void* p, uint64_t bytes) {
    RET(use(ctx));
    if (!p || !bytes) return fail(BSX_ERR_BAD_ARG, "bsx_host_register: null pointer / zero size");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return BSX_OK;
}
int bsx_host_unregister(bsx_ctx* ctx, void* p) {
    RET(use(ctx));
    if (!p) return fail(BSX_ERR_BAD_ARG, "bsx_host_unregister: null pointer");
    HIPCHK(hipHostUnregister(p));
    return BSX_OK;
}

int bsx_encode_data_root_tuple(bsx_ctx* ctx, const uint8_t data_hash[32], uint64_t height, uint8_t out[64]) {
    HOST_ENTER();
    if (!data_hash || !out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    hipStream_t st = ctx->stream;
    DBuf d;
    RET(d.alloc(96));
    H2D(d.p, data_hash, 32);
    HIPCHK(bsxk_encode_tuple(st, d.as<uint8_t>(), height, d.as<uint8_t>() + 32));
    D2H(out, d.as<uint8_t>() + 32, 64);
    SYNC();
    return BSX_OK;
}

int bsx_get_data_commitment(bsx_ctx* ctx, const uint8_t* data_hashes, uint32_t max_leaves, uint64_t start_block, uint64_t end_block,
                            uint8_t out_root[32]) {
    HOST_ENTER();
    if (!data_hashes || !out_root) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!pow2(max_leaves) || max_leaves > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "MAX_LEAVES must be a power of two <= %d", BSX_MAX_BATCH);
    hipStream_t st = ctx->stream;
    DBuf d;
    RET(d.alloc((size_t)max_leaves * 32 + 64));
    uint8_t* dr = d.as<uint8_t>() + (size_t)max_leaves * 32;
    H2D(d.p, data_hashes, (size_t)max_leaves * 32);
    HIPCHK(bsxk_data_commitment(st, d.as<uint8_t>(), max_leaves, start_block, end_block, dr, reinterpret_cast<uint32_t*>(dr + 32)));
    uint8_t tmp[36];
    D2H(tmp, dr, 36);
    SYNC();
    memcpy(out_root, tmp, 32);
    uint32_t flags;
    memcpy(&flags, tmp + 32, 4);
    if (flags) return fail(BSX_ERR_ASSERT, "get_data_commitment: assertion mask 0x%x (A1 end>=start builder.rs:113-114, A2 u32 range :128)", flags);
    return BSX_OK;
}

int bsx_header_hashes(bsx_ctx* ctx, const bsx_header* headers, uint64_t n, uint8_t* out_hashes, bsx_data_hash_proof* out_dh,
                      bsx_last_block_id_proof* out_lb) {
    HOST_ENTER();
    if (n && !headers) return fail(BSX_ERR_BAD_ARG, "null headers");
    if (!n) return BSX_OK;
    hipStream_t st = ctx->stream;
    DBuf dh, dout, dst;
    RET(dh.alloc(n * sizeof(bsx_header)));
    RET(dout.alloc(n * (32 + 128 + 128)));
    RET(dst.alloc(4));
    uint8_t* d_hash = dout.as<uint8_t>();
    uint8_t* d_dh = d_hash + n * 32;
    uint8_t* d_lb = d_dh + n * 128;
    H2D(dh.p, headers, n * sizeof(bsx_header));
    HIPCHK(hipMemsetAsync(dst.p, 0, 4, st));
    HIPCHK(bsxk_header_merkle(st, dh.as<bsx_header>(), n, d_hash, d_dh, d_lb, nullptr, dst.as<uint32_t>(), 0, 0));
    std::vector<uint8_t> tmp(n * 288);
    uint32_t hs = 0;
    StagedD2H back(ctx, st);
    RET(back.copy(tmp.data(), d_hash, n * 288));
    RET(back.copy(&hs, dst.p, 4));
    RET(back.sync());
    RET(header_status_to_rc(hs, 0));
    if (out_hashes) memcpy(out_hashes, tmp.data(), n * 32);
    for (uint64_t i = 0; i < n; i++) {
        if (out_dh) {   // circuits/input.rs:172-181: leaf = the encoded data_hash field
            if (headers[i].len[BSX_DATA_HASH_INDEX] != BSX_PROTOBUF_HASH_SIZE) return fail(BSX_ERR_BAD_HEADER, "header %llu: data_hash leaf is %u bytes, not 34", (unsigned long long)i, headers[i].len[BSX_DATA_HASH_INDEX]);
            memcpy(out_dh[i].aunts, tmp.data() + n * 32 + i * 128, 128);
            memcpy(out_dh[i].leaf, headers[i].hash[1], BSX_PROTOBUF_HASH_SIZE);
        }
        if (out_lb) {   // circuits/input.rs:187-197
            if (headers[i].len[BSX_LAST_BLOCK_ID_INDEX] != BSX_PROTOBUF_BLOCK_ID_SIZE) return fail(BSX_ERR_BAD_HEADER, "header %llu: last_block_id leaf is %u bytes, not 72", (unsigned long long)i, headers[i].len[BSX_LAST_BLOCK_ID_INDEX]);
            memcpy(out_lb[i].aunts, tmp.data() + n * 160 + i * 128, 128);
            memcpy(out_lb[i].leaf, headers[i].last_block_id, BSX_PROTOBUF_BLOCK_ID_SIZE);
        }
    }
    return BSX_OK;
}

// The small inputs and results of bsx_header_range as ONE device block mirrored in page-locked host memory: one H2D and one
// D2H per call instead of four + eight copies between pageable stack variables and separate device buffers (each of those
// is a staged, host-blocking copy of 10-20 us: a third of the 0.4 ms a proof request took).
//   in : range (80 B) @0, latest @128, target validators @256, trusted validators @256 + v_max * 256
//   out: output64 @+0, result record @+64, header status @+192, hint status @+196, final status @+200, skip status @+204,
//        commit result @+256 (96 B)
struct SmallIO {
    uint8_t* d = nullptr;
    uint8_t* h = nullptr;
    size_t in_bytes = 0, out_off = 0;
    static constexpr size_t OUT_BYTES = 384;
    uint8_t* dout(size_t off) const { return d + out_off + off; }
    const uint8_t* hout(size_t off) const { return h + out_off + off; }
};
static int ctx_hstage(bsx_ctx* ctx, size_t bytes, uint8_t** out) {
    if (bytes < StagedD2H::CAP) bytes = StagedD2H::CAP;        // never shrinks below what StagedD2H takes
    if (ctx->hstage_cap < bytes) {
        if (ctx->hstage) (void)hipHostFree(ctx->hstage);
        ctx->hstage = nullptr;
        ctx->hstage_cap = 0;
        const size_t cap = (bytes + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->hstage), cap, hipHostMallocDefault));
        ctx->hstage_cap = cap;
    }
    *out = ctx->hstage;
    return BSX_OK;
}
static void dbuf_alias(DBuf& b, void* p) { b.p = p; b.owned = false; }

// Shared by the hint-level entry points: uploads headers [S .. ) of one range and runs P5.
struct RangeDev {
    DBuf headers, hashes, dh, lb, paths, ranges, latest, hstatus, astatus;
    uint64_t hpr = 0;
};
static int upload_range(bsx_ctx* ctx, hipStream_t st, const bsx_header* headers, uint64_t first_height, uint64_t n_headers,
                        uint64_t S_, const bsx_shared_ctx& range, uint64_t latest_block, RangeDev& rd, const SmallIO* io = nullptr,
                        bool skip_header_copy = false, const bsxk_merkle_tap* tap = nullptr, bool hash_later = false) {
    if (!headers || !n_headers) return fail(BSX_ERR_BAD_ARG, "no headers supplied");
    if (S_ < first_height || S_ - first_height >= n_headers) return fail(BSX_ERR_BAD_ARG, "header for start block %llu not supplied (first_height %llu, n %llu)", (unsigned long long)S_, (unsigned long long)first_height, (unsigned long long)n_headers);
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    const bsx_header* h0 = headers + (S_ - first_height);
    rd.hpr = n_headers - (S_ - first_height);
    RET(rd.headers.alloc(rd.hpr * sizeof(bsx_header)));
    RET(rd.hashes.alloc(rd.hpr * 32));
    RET(rd.dh.alloc(rd.hpr * 128));
    RET(rd.lb.alloc(rd.hpr * 128));
    RET(rd.paths.alloc(rd.hpr * BSX_HEADER_PATH_BYTES));
    if (io) {                              // range / latest already uploaded, status words already zeroed (SmallIO)
        dbuf_alias(rd.ranges, io->d);
        dbuf_alias(rd.latest, io->d + 128);
        dbuf_alias(rd.hstatus, io->dout(192));
        dbuf_alias(rd.astatus, io->dout(196));
        if (!skip_header_copy) H2D(rd.headers.p, h0, rd.hpr * sizeof(bsx_header));
    } else {
        RET(rd.ranges.alloc(sizeof(bsx_shared_ctx)));
        RET(rd.latest.alloc(8));
        RET(rd.hstatus.alloc(4));
        RET(rd.astatus.alloc(4));
        H2D(rd.headers.p, h0, rd.hpr * sizeof(bsx_header));
        H2D(rd.ranges.p, &range, sizeof range);
        H2D(rd.latest.p, &latest_block, 8);
        HIPCHK(hipMemsetAsync(rd.hstatus.p, 0, 4, st));
        HIPCHK(hipMemsetAsync(rd.astatus.p, 0, 4, st));
    }
    if (hash_later) return BSX_OK;          // the caller launches hash_range_headers() itself (more enqueues while the copy is in flight)
    HIPCHK(bsxk_header_merkle(st, rd.headers.as<bsx_header>(), rd.hpr, rd.hashes.as<uint8_t>(), rd.dh.as<uint8_t>(), rd.lb.as<uint8_t>(),
                              rd.paths.as<uint8_t>(), rd.hstatus.as<uint32_t>(), 0, 0, tap));
    (void)ctx;
    return BSX_OK;
}
static int hash_range_headers(hipStream_t st, RangeDev& rd, const bsxk_merkle_tap* tap) {
    HIPCHK(bsxk_header_merkle(st, rd.headers.as<bsx_header>(), rd.hpr, rd.hashes.as<uint8_t>(), rd.dh.as<uint8_t>(), rd.lb.as<uint8_t>(),
                              rd.paths.as<uint8_t>(), rd.hstatus.as<uint32_t>(), 0, 0, tap));
    return BSX_OK;
}

int bsx_data_commitment_inputs(bsx_ctx* ctx, const bsx_header* headers, uint64_t first_height, uint64_t n_headers, uint64_t latest_block,
                               uint64_t start_block, uint64_t end_block, uint32_t max_leaves, uint8_t out_start_header[32],
                               uint8_t out_end_header[32], bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb,
                               uint8_t out_expected_data_commitment[32]) {
    if (bsx_batcher* bt = ctx ? ctx->batcher.load() : nullptr)
        if (bsxb_config(bt)->batch_size == max_leaves) {
            bsx_ticket t = 0;
            RET(bsx_submit_data_commitment_inputs(bt, headers, first_height, n_headers, latest_block, start_block, end_block, out_start_header, out_end_header,
                                                  out_dh, out_lb, out_expected_data_commitment, &t));
            return bsx_wait(bt, t);
        }
    HOST_ENTER();
    if (!out_start_header || !out_end_header || !out_dh || !out_lb) return fail(BSX_ERR_BAD_ARG, "null output");
    if (!max_leaves || max_leaves > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "MAX_LEAVES must be in 1..%d", BSX_MAX_BATCH);
    if (end_block - start_block > (uint64_t)max_leaves) return fail(BSX_ERR_RANGE_TOO_LONG, "end - start > MAX_LEAVES (circuits/input.rs:154)");
    uint32_t P = 1;
    while (P < max_leaves) P *= 2;                 // the kernels fold power-of-two batches; extra slots stay zero padded
    hipStream_t st = ctx->stream;
    bsx_shared_ctx range{};
    range.start_block = start_block;
    range.end_block = end_block;
    RangeDev rd;
    RET(upload_range(ctx, st, headers, first_height, n_headers, start_block, range, latest_block, rd));
    const bsx_witness_layout L = bsx_map_layout(P);
    DBuf cw, aux;
    RET(cw.alloc(L.compact_stride));
    RET(aux.alloc((size_t)P * 32 + 64));
    HIPCHK(hipMemsetAsync(cw.p, 0, L.compact_stride, st));
    HIPCHK(bsxk_assemble_inputs(st, 1, 1, P, 0, 1, (uint32_t)(end_block - start_block), rd.ranges.as<bsx_shared_ctx>(), rd.latest.as<uint64_t>(),
                                rd.headers.as<bsx_header>(), rd.hpr, 0, rd.hashes.as<uint8_t>(), rd.dh.as<uint8_t>(), rd.lb.as<uint8_t>(),
                                cw.as<uint8_t>(), rd.astatus.as<uint32_t>(), nullptr, nullptr));
    std::vector<uint8_t> img(L.compact_stride);
    uint32_t hs = 0, as = 0;
    D2H(img.data(), cw.p, L.compact_stride);
    D2H(&hs, rd.hstatus.p, 4);
    D2H(&as, rd.astatus.p, 4);
    SYNC();
    RET(header_status_to_rc(hs, as));
    memcpy(out_start_header, img.data() + bsx_off_start_header(), 32);
    memcpy(out_end_header, img.data() + bsx_off_end_header(), 32);
    memcpy(out_dh, img.data() + bsx_off_dh_proofs(P), (size_t)max_leaves * sizeof(bsx_data_hash_proof));
    memcpy(out_lb, img.data() + bsx_off_lb_proofs(P), (size_t)max_leaves * sizeof(bsx_last_block_id_proof));
    if (out_expected_data_commitment) {
        // input.rs:241-244 with :70-72: the node's commitment over [start, req_end) — zero when the range is empty
        const uint64_t req_end = end_block < latest_block - 2 ? end_block : latest_block - 2;
        if (req_end <= start_block) {
            memset(out_expected_data_commitment, 0, 32);
        } else {
            std::vector<uint8_t> dhs((size_t)P * 32, 0);
            for (uint32_t i = 0; i < max_leaves; i++) memcpy(dhs.data() + 32 * i, out_dh[i].leaf + 2, 32);
            uint8_t* dr = aux.as<uint8_t>() + (size_t)P * 32;
            H2D(aux.p, dhs.data(), dhs.size());
            HIPCHK(bsxk_data_commitment(st, aux.as<uint8_t>(), P, start_block, req_end, dr, reinterpret_cast<uint32_t*>(dr + 32)));
            D2H(out_expected_data_commitment, dr, 32);
            SYNC();
        }
    }
    return BSX_OK;
}

static int subchain_rc(const bsx_subchain& rec) {
    if (rec.assert_fail)
        return fail(BSX_ERR_ASSERT, "prove_subchain: assertion mask 0x%x, first failing slot %u (A3 builder.rs:205-207, A4 :210-212, A5 :216-219, A6 :229-232)",
                    rec.assert_fail, rec.first_bad_slot);
    return BSX_OK;
}

int bsx_prove_subchain(bsx_ctx* ctx, uint32_t batch_size, const uint8_t start_header[32], const uint8_t end_header[32],
                       const bsx_data_hash_proof* dh, const bsx_last_block_id_proof* lb, uint64_t batch_start_block,
                       uint64_t batch_end_block, uint64_t global_end_block, const uint8_t global_end_header_hash[32],
                       bsx_subchain* out_record, uint64_t* witness) {
    if (bsx_batcher* bt = (ctx && !witness) ? ctx->batcher.load() : nullptr)
        if (bsxb_config(bt)->batch_size == batch_size) {
            bsx_ticket t = 0;
            RET(bsx_submit_prove_subchain(bt, start_header, end_header, dh, lb, batch_start_block, batch_end_block, global_end_block, global_end_header_hash,
                                          out_record, &t));
            return bsx_wait(bt, t);
        }
    HOST_ENTER();
    if (!start_header || !end_header || !dh || !lb || !global_end_header_hash || !out_record) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    const uint32_t B = batch_size;
    const bsx_witness_layout L = bsx_map_layout(B);
    hipStream_t st = ctx->stream;
    // caller bytes -> compact witness image (no arithmetic: the kernel completes it)
    std::vector<uint8_t> img(L.compact_stride, 0);
    memcpy(img.data() + bsx_off_ctx_end_header(), global_end_header_hash, 32);
    memcpy(img.data() + bsx_off_start_header(), start_header, 32);
    memcpy(img.data() + bsx_off_end_header(), end_header, 32);
    memcpy(img.data() + bsx_off_dh_proofs(B), dh, (size_t)B * sizeof *dh);
    memcpy(img.data() + bsx_off_lb_proofs(B), lb, (size_t)B * sizeof *lb);
    uint32_t* W = reinterpret_cast<uint32_t*>(img.data() + L.off_words);
    W[BSX_W_CTX_END] = (uint32_t)global_end_block; W[BSX_W_CTX_END + 1] = (uint32_t)(global_end_block >> 32);
    W[BSX_W_BATCH_START] = (uint32_t)batch_start_block; W[BSX_W_BATCH_START + 1] = (uint32_t)(batch_start_block >> 32);
    W[BSX_W_BATCH_END] = (uint32_t)batch_end_block; W[BSX_W_BATCH_END + 1] = (uint32_t)(batch_end_block >> 32);
    bsx_shared_ctx range{};
    range.end_block = global_end_block;
    memcpy(range.end_header_hash, global_end_header_hash, 32);
    DBuf cw, rg, rec, wit;
    RET(cw.alloc(L.compact_stride));
    RET(rg.alloc(sizeof range));
    RET(rec.alloc(sizeof(bsx_subchain)));
    H2D(cw.p, img.data(), L.compact_stride);
    H2D(rg.p, &range, sizeof range);
    HIPCHK(bsxk_prove_subchain(st, 1, B, 1, rg.as<bsx_shared_ctx>(), cw.as<uint8_t>(), rec.as<bsx_subchain>(), 0));   // caller's proofs
    if (witness) {
        RET(wit.alloc(L.n_elements * 8));
        HIPCHK(bsxk_expand_witness(st, &L, 1, cw.as<uint8_t>(), wit.as<uint64_t>()));
        D2H(witness, wit.p, L.n_elements * 8);
    }
    D2H(out_record, rec.p, sizeof(bsx_subchain));
    SYNC();
    return subchain_rc(*out_record);
}

// the map closure of prove_data_commitment for one map job (builder.rs:305-336): coalesced on a context with a batcher, else the two
// host-tier calls in turn
int bsx_map_job(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const bsx_shared_ctx* range, uint32_t job_index, const bsx_header* headers,
                uint64_t first_height, uint64_t n_headers, uint64_t latest_block, uint8_t out_start_header[32], uint8_t out_end_header[32],
                bsx_data_hash_proof* out_dh, bsx_last_block_id_proof* out_lb, bsx_subchain* out_record) {
    if (!ctx) return fail(BSX_ERR_BAD_ARG, "null context");
    if (!range || !out_record || !headers) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    if (job_index >= nb_map_jobs) return fail(BSX_ERR_BAD_ARG, "job_index %u is not below NB_MAP_JOBS = %u", job_index, nb_map_jobs);
    if (bsx_batcher* bt = ctx->batcher.load())
        if (bsxb_config(bt)->batch_size == batch_size && bsxb_config(bt)->nb_map_jobs >= nb_map_jobs) {
            bsx_ticket t = 0;
            RET(bsx_submit_map_job(bt, range, job_index, headers, first_height, n_headers, latest_block, out_start_header, out_end_header, out_dh, out_lb,
                                   out_record, &t));
            return bsx_wait(bt, t);
        }
    const uint64_t bs = range->start_block + (uint64_t)job_index * batch_size, be = bs + batch_size;
    std::vector<bsx_data_hash_proof> dh(out_dh ? 0 : batch_size);
    std::vector<bsx_last_block_id_proof> lb(out_lb ? 0 : batch_size);
    uint8_t sh[32], eh[32];
    bsx_data_hash_proof* pdh = out_dh ? out_dh : dh.data();
    bsx_last_block_id_proof* plb = out_lb ? out_lb : lb.data();
    // a batch the chain head cannot reach reads no header: the hint still wants a non-empty array
    const uint64_t lim = latest_block >= 2 ? latest_block - 2 : 0, req_end = be < lim ? be : lim;
    if (bs <= req_end) {
        RET(bsx_data_commitment_inputs(ctx, headers, first_height, n_headers, latest_block, bs, be, batch_size, sh, eh, pdh, plb, nullptr));
    } else {
        // input.rs:220-262 with nothing to fetch: zero proofs, dummy (zero) start / end headers
        if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
        memset(sh, 0, 32); memset(eh, 0, 32);
        memset(pdh, 0, (size_t)batch_size * sizeof *pdh); memset(plb, 0, (size_t)batch_size * sizeof *plb);
    }
    if (out_start_header) memcpy(out_start_header, sh, 32);
    if (out_end_header) memcpy(out_end_header, eh, 32);
    return bsx_prove_subchain(ctx, batch_size, sh, eh, pdh, plb, bs, be, range->end_block, range->end_header_hash, out_record, nullptr);
}

int bsx_reduce(bsx_ctx* ctx, const bsx_subchain* records, uint32_t n, bsx_subchain* out) {
    HOST_ENTER();
    if (!records || !out) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!pow2(n) || n > 256) return fail(BSX_ERR_BAD_ARG, "reduce fan-in must be a power of two <= 256 (got %u)", n);
    hipStream_t st = ctx->stream;
    DBuf d;
    RET(d.alloc((size_t)(n + 1) * sizeof(bsx_subchain)));
    H2D(d.p, records, (size_t)n * sizeof(bsx_subchain));
    HIPCHK(bsxk_reduce(st, 1, n, d.as<bsx_subchain>(), n, 1, d.as<bsx_subchain>() + n, nullptr));
    D2H(out, d.as<bsx_subchain>() + n, sizeof(bsx_subchain));
    SYNC();
    return BSX_OK;
}

// prove_data_commitment on one range whose device state (headers hashed) is in rd.  d_target_hashes optional.
// the host-side end of run_data_commitment: statuses -> return code, outputs
static int finish_data_commitment(const uint8_t o[64], const bsx_subchain& result, uint32_t hs, uint32_t as, uint32_t stv,
                                  uint8_t out_commitment[32], uint8_t output64[64], bsx_subchain* out_result, uint32_t* out_status) {
    RET(header_status_to_rc(hs, as));
    if (out_commitment) memcpy(out_commitment, o + 32, 32);
    if (output64) memcpy(output64, o, 64);
    if (out_result) { *out_result = result; out_result->assert_fail = stv; }
    if (out_status) *out_status = stv;
    if (stv) return fail(BSX_ERR_ASSERT, "prove_data_commitment: assertion mask 0x%x (A7 builder.rs:292-297, A8 :350-355, A9 :401-406; A1-A6 from the map jobs)", stv);
    return BSX_OK;
}
// io (bsx_header_range): results go to the SmallIO block and are NOT fetched here — the caller fetches the block once and
// calls finish_data_commitment itself
static int run_data_commitment(bsx_ctx* ctx, hipStream_t st, uint32_t J, uint32_t B, RangeDev& rd, const uint8_t* d_target_hashes,
                               uint8_t out_commitment[32], uint8_t output64[64], bsx_subchain* out_result, bsx_subchain* records,
                               uint64_t* witness, uint32_t* out_status, const SmallIO* io = nullptr) {
    const bsx_witness_layout L = bsx_map_layout(B), R = bsx_reduce_layout();
    DBuf cw, rcw, recs, res, o64, stw, wit;
    RET(cw.alloc((size_t)J * L.compact_stride));
    RET(rcw.alloc((size_t)(J > 1 ? J - 1 : 1) * R.compact_stride));
    RET(recs.alloc((size_t)J * sizeof(bsx_subchain)));
    if (io) {
        dbuf_alias(o64, io->dout(0));
        dbuf_alias(res, io->dout(64));
        dbuf_alias(stw, io->dout(200));
    } else {
        RET(res.alloc(sizeof(bsx_subchain)));
        RET(o64.alloc(64));
        RET(stw.alloc(4));
    }
    // every byte of the three sections is written by the hint and prove_subchain below (padding slots included); only the alignment
    // gaps between the sections are not, and nothing reads those — cleared when the image is expanded for a caller, skipped on the
    // latency path (one launch less in front of the hint)
    if (witness) HIPCHK(hipMemsetAsync(cw.p, 0, (size_t)J * L.compact_stride, st));
    HIPCHK(bsxk_assemble_inputs(st, 1, J, B, 0, J, B, rd.ranges.as<bsx_shared_ctx>(), rd.latest.as<uint64_t>(), rd.headers.as<bsx_header>(), rd.hpr, 0,
                                rd.hashes.as<uint8_t>(), rd.dh.as<uint8_t>(), rd.lb.as<uint8_t>(), cw.as<uint8_t>(), rd.astatus.as<uint32_t>(),
                                rd.paths.as<uint8_t>(), ctx->zero_paths));
    // the proofs are the hint's own (nodes of the header trees hashed above): their path digests are already in place
    HIPCHK(bsxk_prove_subchain(st, 1, B, J, rd.ranges.as<bsx_shared_ctx>(), cw.as<uint8_t>(), recs.as<bsx_subchain>(),
                               BSX_SUBCHAIN_PATHS_FROM_HINT));
    // reduce (builder.rs:337-395) + final assertions and public output (:292-297,400-406; header_range.rs:57-58) in one launch
    HIPCHK(bsxk_reduce_finalize(st, 1, J, recs.as<bsx_subchain>(), res.as<bsx_subchain>(), rcw.as<uint8_t>(), J, B, rd.ranges.as<bsx_shared_ctx>(),
                                d_target_hashes, o64.as<uint8_t>(), stw.as<uint32_t>()));
    if (witness) {
        const size_t nmap = (size_t)J * L.n_elements, nred = (size_t)(J - 1) * R.n_elements;
        RET(wit.alloc((nmap + nred) * 8 + 16));
        HIPCHK(bsxk_expand_witness(st, &L, J, cw.as<uint8_t>(), wit.as<uint64_t>()));
        if (J > 1) {
            // the reduce section starts at element nmap; keep the kernel's 16-byte pair alignment by expanding into an
            // aligned scratch and copying out
            DBuf wr;
            RET(wr.alloc(nred * 8 + 16));
            HIPCHK(bsxk_expand_witness(st, &R, J - 1, rcw.as<uint8_t>(), wr.as<uint64_t>()));
            HIPCHK(hipMemcpyAsync(wit.as<uint64_t>() + nmap, wr.p, nred * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        D2H(witness, wit.p, (nmap + nred) * 8);
    }
    if (io) {
        if (records) D2H(records, recs.p, (size_t)J * sizeof(bsx_subchain));
        return BSX_OK;
    }
    uint8_t o[64];
    bsx_subchain result;
    uint32_t hs = 0, as = 0, stv = 0;
    StagedD2H back(ctx, st);
    RET(back.copy(o, o64.p, 64));
    RET(back.copy(&result, res.p, sizeof result));
    if (records) RET(back.copy(records, recs.p, (size_t)J * sizeof(bsx_subchain)));
    RET(back.copy(&hs, rd.hstatus.p, 4));
    RET(back.copy(&as, rd.astatus.p, 4));
    RET(back.copy(&stv, stw.p, 4));
    RET(back.sync());
    return finish_data_commitment(o, result, hs, as, stv, out_commitment, output64, out_result, out_status);
}

int bsx_prove_data_commitment(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const bsx_shared_ctx* range, const bsx_header* headers,
                              uint64_t first_height, uint64_t n_headers, uint64_t latest_block, uint8_t out_data_commitment[32],
                              bsx_subchain* out_result, bsx_subchain* records, uint64_t* witness) {
    HOST_ENTER();
    if (!range || !out_data_commitment) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (!pow2(nb_map_jobs) || nb_map_jobs > 256) return fail(BSX_ERR_BAD_ARG, "NB_MAP_JOBS must be a power of two <= 256");
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    hipStream_t st = ctx->stream;
    RangeDev rd;
    RET(upload_range(ctx, st, headers, first_height, n_headers, range->start_block, *range, latest_block, rd));
    return run_data_commitment(ctx, st, nb_map_jobs, batch_size, rd, nullptr, out_data_commitment, nullptr, out_result, records, witness, nullptr);
}

int bsx_prove_next_header_data_commitment(bsx_ctx* ctx, uint64_t prev_block_number, const uint8_t prev_header_hash[32],
                                          uint64_t next_block_number, const bsx_header* header, uint64_t latest_block,
                                          uint8_t out_data_commitment[32]) {
    HOST_ENTER();
    if (!prev_header_hash || !header || !out_data_commitment) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (next_block_number - prev_block_number > 1) return fail(BSX_ERR_RANGE_TOO_LONG, "end - start > MAX_LEAVES = 1 (circuits/input.rs:154)");
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    // builder.rs:415-423: hint with MAX_LEAVES = 1.  data_hash_proofs[0] is real iff prev < min(next, latest-2)
    // (input.rs:162,172); the last_block_id proof the hint also returns is not used by this circuit.
    const uint64_t req_end = next_block_number < latest_block - 2 ? next_block_number : latest_block - 2;
    bsx_data_hash_proof dh;
    bsx_last_block_id_proof lb;
    memset(&dh, 0, sizeof dh);
    memset(&lb, 0, sizeof lb);
    if (prev_block_number < req_end) RET(bsx_header_hashes(ctx, header, 1, nullptr, &dh, nullptr));
    // one-slot map job: slot.dh_path[4] = data_hash_proof_root (:429-433), leaf_hash[0] = leaf_hash(tuple) (:436-442)
    const bsx_witness_layout L = bsx_map_layout(1);
    hipStream_t st = ctx->stream;
    std::vector<uint8_t> img(L.compact_stride, 0);
    memcpy(img.data() + bsx_off_dh_proofs(1), &dh, sizeof dh);
    memcpy(img.data() + bsx_off_lb_proofs(1), &lb, sizeof lb);
    uint32_t* W = reinterpret_cast<uint32_t*>(img.data() + L.off_words);
    W[BSX_W_BATCH_START] = (uint32_t)prev_block_number; W[BSX_W_BATCH_START + 1] = (uint32_t)(prev_block_number >> 32);
    W[BSX_W_BATCH_END] = (uint32_t)next_block_number; W[BSX_W_BATCH_END + 1] = (uint32_t)(next_block_number >> 32);
    bsx_shared_ctx range{};
    range.end_block = next_block_number;
    DBuf cw, rg, rec;
    RET(cw.alloc(L.compact_stride));
    RET(rg.alloc(sizeof range));
    RET(rec.alloc(sizeof(bsx_subchain)));
    H2D(cw.p, img.data(), L.compact_stride);
    H2D(rg.p, &range, sizeof range);
    HIPCHK(bsxk_prove_subchain(st, 1, 1, 1, rg.as<bsx_shared_ctx>(), cw.as<uint8_t>(), rec.as<bsx_subchain>(), 0));
    D2H(img.data(), cw.p, L.compact_stride);
    SYNC();
    memcpy(out_data_commitment, img.data() + bsx_off_leaf_hashes(1), 32);
    if (memcmp(img.data() + bsx_off_slots(1) + 128, prev_header_hash, 32) != 0)   // :434 (A10)
        return fail(BSX_ERR_ASSERT, "prove_next_header_data_commitment: data_hash proof root != prev_header_hash (A10, builder.rs:434)");
    return BSX_OK;
}

// fetcher.rs:60-87 find_block_to_request over pre-fetched candidates: every is_valid_skip of the halving sequence in
// one launch, then the reference's loop on the host
int bsx_find_block_to_request(bsx_ctx* ctx, uint64_t start_block, uint64_t max_end_block, const bsx_validator* start_validators,
                              uint32_t n_candidates, const uint64_t* candidate_heights, const bsx_validator* candidate_validators,
                              uint32_t v_max, uint64_t* out_block, bsx_skip_eval* out_evals) {
    HOST_ENTER();
    if (!out_block || !start_validators || (n_candidates && (!candidate_heights || !candidate_validators)))
        return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (max_end_block <= start_block) return fail(BSX_ERR_BAD_ARG, "max_end_block must be above start_block");
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    hipStream_t st = ctx->stream;
    std::vector<bsx_skip_eval> ev(n_candidates);
    if (n_candidates) {
        DBuf ds, dc, de;
        RET(ds.alloc((size_t)v_max * sizeof(bsx_validator)));
        RET(dc.alloc((size_t)n_candidates * v_max * sizeof(bsx_validator)));
        RET(de.alloc((size_t)n_candidates * sizeof(bsx_skip_eval)));
        H2D(ds.p, start_validators, (size_t)v_max * sizeof(bsx_validator));
        H2D(dc.p, candidate_validators, (size_t)n_candidates * v_max * sizeof(bsx_validator));
        HIPCHK(bsxk_skip_eval(st, ds.as<bsx_validator>(), dc.as<bsx_validator>(), n_candidates, v_max, de.as<bsx_skip_eval>()));
        D2H(ev.data(), de.p, (size_t)n_candidates * sizeof(bsx_skip_eval));
        SYNC();
    }
    if (out_evals) memcpy(out_evals, ev.data(), (size_t)n_candidates * sizeof(bsx_skip_eval));
    for (uint32_t c = 0; c < n_candidates; c++)
        if (ev[c].power_overflow)
            return fail(BSX_ERR_BAD_ARG, "candidate %u: a validator set's voting power exceeds MaxTotalVotingPower (MaxInt64 / 8)", c);
    uint64_t curr_end_block = max_end_block;                                  // :61
    for (;;) {                                                                // :62
        if (curr_end_block - start_block == 1) break;                         // :63-65
        uint32_t c = 0;
        while (c < n_candidates && candidate_heights[c] != curr_end_block) c++;
        if (c == n_candidates)
            return fail(BSX_ERR_BAD_ARG, "find_block_to_request visits height %llu, which is not among the candidates",
                        (unsigned long long)curr_end_block);
        if (ev[c].valid) break;                                               // :76-82
        curr_end_block = (curr_end_block + start_block) / 2;                  // :84-85
    }
    *out_block = curr_end_block;
    return BSX_OK;
}

int bsx_verify_commits(bsx_ctx* ctx, const bsx_validator* validators, uint32_t n_commits, uint32_t v_max, const uint8_t* header_hashes,
                       bsx_commit_result* out_results, uint8_t* out_sig_ok, uint64_t* witness) {
    HOST_ENTER();
    if (!validators || !header_hashes || !out_results) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    if (!n_commits) return BSX_OK;
    hipStream_t st = ctx->stream;
    const uint64_t n = (uint64_t)n_commits * v_max;
    DBuf dv, dh, dok, dhh, dres;
    RET(dv.alloc(n * sizeof(bsx_validator)));
    RET(dh.alloc(n * 32));
    RET(dok.alloc(n));
    RET(dhh.alloc((size_t)n_commits * 32));
    RET(dres.alloc((size_t)n_commits * sizeof(bsx_commit_result)));
    H2D(dv.p, validators, n * sizeof(bsx_validator));
    H2D(dhh.p, header_hashes, (size_t)n_commits * 32);
    // witness (optional): one COMMIT unit per commit (include/bsx_layout.h), written by the kernels of the chain as they go
    const bsx_witness_layout CL = bsx_commit_layout(v_max);
    DBuf dcw, dwit;
    bsxk_unit_dst cwd = bsxk_unit(nullptr, CL);
    if (witness) {
        RET(dcw.alloc((size_t)n_commits * CL.compact_stride));
        RET(dwit.alloc((size_t)n_commits * CL.n_elements * 8 + 16));
        HIPCHK(hipMemsetAsync(dcw.p, 0, (size_t)n_commits * CL.compact_stride, st));
        cwd.base = dcw.as<uint8_t>();
    }
    const bsxk_unit_dst* cwp = witness ? &cwd : nullptr;
    HIPCHK(bsxk_sha512_challenge(st, dv.as<bsx_validator>(), n, dh.as<uint8_t>(), nullptr, v_max, cwp));
    {
        // per-key tables from the first commit's slots (kept in the context between calls); slots whose key differs fall
        // back to the generic path inside bsxk_ed25519_verify_keyed: same accept set for any input
        DBuf dscr, drd;                             // batch inversion pays from tens of thousands of signatures on; below: the latency form
        if (n >= 65536) RET(dscr.alloc(bsxk_ed25519_scratch_bytes(n)));
        else RET(drd.alloc(bsxk_ed25519_rdec_bytes(n)));
        RET(ctx_verify(ctx, st, dv.as<bsx_validator>(), dh.as<uint8_t>(), n, v_max, dok.as<uint8_t>(), dscr.p, drd.p,
                       (int64_t)bsxh_key_mismatches(validators, n_commits, v_max)));
    }
    HIPCHK(bsxk_commit_tally(st, dv.as<bsx_validator>(), n_commits, v_max, dhh.as<uint8_t>(), dok.as<uint8_t>(), dres.as<bsx_commit_result>(), cwp));
    if (witness) {
        HIPCHK(bsxk_expand_witness(st, &CL, n_commits, dcw.as<uint8_t>(), dwit.as<uint64_t>()));
        D2H(witness, dwit.p, (size_t)n_commits * CL.n_elements * 8);
    }
    StagedD2H back(ctx, st);
    RET(back.copy(out_results, dres.p, (size_t)n_commits * sizeof(bsx_commit_result)));
    if (out_sig_ok) RET(back.copy(out_sig_ok, dok.p, n));
    RET(back.sync());
    for (uint32_t c = 0; c < n_commits; c++)
        if (out_results[c].power_overflow)
            return fail(BSX_ERR_BAD_ARG, "commit %u: the voting powers add up to more than MaxTotalVotingPower (MaxInt64 / 8); tallies are meaningless", c);
    return BSX_OK;
}

static int bsx_header_range_serial(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const bsx_header* headers,
                                   uint64_t first_height, uint64_t n_headers, uint64_t latest_block, const bsx_validator* target_validators,
                                   const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len,
                                   uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness);
int bsx_header_range(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const bsx_header* headers,
                     uint64_t first_height, uint64_t n_headers, uint64_t latest_block, const bsx_validator* target_validators,
                     const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len,
                     uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness) {
    if (bsx_batcher* bt = (ctx && !witness) ? ctx->batcher.load() : nullptr) {
        const bsx_batcher_config* bc = bsxb_config(bt);
        if (bc->nb_map_jobs == nb_map_jobs && bc->batch_size == batch_size && bc->v_max == v_max && bc->chain_id_len == chain_id_len &&
            (chain_id_len == 0 || (chain_id && chain_id_len <= 50 && memcmp(bc->chain_id, chain_id, chain_id_len) == 0))) {
            // A LONE caller gains nothing from the batcher's hops (staging copy, debounce, the worker's wake-up and the completion's: 0.32
            // against 0.26 ms): a call that finds NO other thread inside this wrapper, nothing of the kind collecting or in flight and the
            // context's serial path free runs on that path directly.  Two or more steady callers are (almost) always inside at the same
            // time — whoever finds another one inside stamps the context, and nobody takes the serial path within 2 ms of such a stamp:
            // they coalesce as before (taking every momentary gap of the batcher instead cost them 7-12 %: one request less per set).
            struct Inside { std::atomic<uint32_t>& n; uint32_t others; explicit Inside(std::atomic<uint32_t>& c) : n(c), others(c.fetch_add(1)) {} ~Inside() { n.fetch_sub(1); } }
                inside(ctx->sync_range_callers);
            std::unique_lock<std::recursive_mutex> alone(ctx->host_mu, std::defer_lock);
#ifdef BSX_NO_SYNC_FAST_PATH                          // A/B builds: always through the batcher
            const bool fast_alone = false;
#else
            const bool fast_alone = true;
#endif
            const uint64_t t_now = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
            if (inside.others) ctx->sync_range_crowd_ns.store(t_now, std::memory_order_relaxed);
            const bool crowd = t_now - ctx->sync_range_crowd_ns.load(std::memory_order_relaxed) < 2000000ull;
            if (!(fast_alone && inside.others == 0 && !crowd && bsxb_range_idle(bt) && alone.try_lock())) {
                bsx_ticket t = 0;
                RET(bsx_submit_header_range_ex(bt, input48, headers, first_height, n_headers, latest_block, target_validators, trusted_validators, output64,
                                               out_commit, &t, BSX_SUBMIT_INPUTS_STAY));
                return bsx_wait(bt, t);
            }
            return bsx_header_range_serial(ctx, nb_map_jobs, batch_size, input48, headers, first_height, n_headers, latest_block, target_validators,
                                           trusted_validators, v_max, chain_id, chain_id_len, output64, out_commit, witness);
        }
    }
    return bsx_header_range_serial(ctx, nb_map_jobs, batch_size, input48, headers, first_height, n_headers, latest_block, target_validators, trusted_validators,
                                   v_max, chain_id, chain_id_len, output64, out_commit, witness);
}

// bsx_header_range with the headers PACKED (wire.cpp: ~408 instead of 512 bytes per header over PCIe).  On a context whose batcher has this
// shape the block is staged and uploaded packed and laid out as records in HBM (k_unpack_headers); any other context unpacks on the
// host and takes the ordinary path, so the call is always valid.
int bsx_header_range_packed(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const void* packed, uint64_t packed_bytes,
                            uint64_t first_height, uint64_t latest_block, const bsx_validator* target_validators, const bsx_validator* trusted_validators,
                            uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit) {
    if (!ctx || !packed) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (bsx_batcher* bt = ctx->batcher.load()) {
        const bsx_batcher_config* bc = bsxb_config(bt);
        if (bc->nb_map_jobs == nb_map_jobs && bc->batch_size == batch_size && bc->v_max == v_max && bc->chain_id_len == chain_id_len &&
            (chain_id_len == 0 || (chain_id && chain_id_len <= 50 && memcmp(bc->chain_id, chain_id, chain_id_len) == 0))) {
            bsx_ticket t = 0;
            RET(bsx_submit_header_range_ex(bt, input48, packed, first_height, packed_bytes, latest_block, target_validators, trusted_validators, output64,
                                           out_commit, &t, BSX_SUBMIT_PACKED_HEADERS));
            return bsx_wait(bt, t);
        }
    }
    uint64_t n = 0;
    if (bsx_unpack_headers(packed, packed_bytes, nullptr, 0, &n) != BSX_OK) return fail(BSX_ERR_BAD_HEADER, "bsx_header_range_packed: the packed header block is inconsistent");
    std::vector<bsx_header> recs(n);
    if (bsx_unpack_headers(packed, packed_bytes, recs.data(), n, &n) != BSX_OK) return fail(BSX_ERR_BAD_HEADER, "bsx_header_range_packed: the packed header block is inconsistent");
    return bsx_header_range(ctx, nb_map_jobs, batch_size, input48, recs.data(), first_height, n, latest_block, target_validators, trusted_validators, v_max, chain_id,
                            chain_id_len, output64, out_commit, nullptr);
}

// the serial host-tier path of bsx_header_range (one request, the context's own streams and arena)
static int bsx_header_range_serial(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const bsx_header* headers,
                                   uint64_t first_height, uint64_t n_headers, uint64_t latest_block, const bsx_validator* target_validators,
                                   const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len,
                                   uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness) {
    HOST_ENTER();
    const auto t_entry = std::chrono::steady_clock::now();
    if (!input48 || !headers || !target_validators || !trusted_validators || !output64) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (chain_id_len > 50 || (chain_id_len && !chain_id)) return fail(BSX_ERR_BAD_ARG, "chain_id: at most 50 bytes");
    if (!pow2(nb_map_jobs) || nb_map_jobs > 256) return fail(BSX_ERR_BAD_ARG, "NB_MAP_JOBS must be a power of two <= 256");
    if (!pow2(batch_size) || batch_size > BSX_MAX_BATCH) return fail(BSX_ERR_BAD_ARG, "BATCH_SIZE must be a power of two <= %d", BSX_MAX_BATCH);
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    // header_range.rs:33-35: evm_read u64 (big endian), bytes32, u64
    uint64_t trusted_block = 0, target_block = 0;
    for (int i = 0; i < 8; i++) trusted_block = trusted_block << 8 | input48[i];
    for (int i = 0; i < 8; i++) target_block = target_block << 8 | input48[40 + i];
    if (!(target_block > trusted_block) || target_block - trusted_block > (uint64_t)nb_map_jobs * batch_size)
        return fail(BSX_ERR_RANGE_TOO_LONG, "skip: need trusted < target <= trusted + %llu", (unsigned long long)nb_map_jobs * batch_size);
    if (trusted_block < first_height || target_block - first_height >= n_headers) return fail(BSX_ERR_BAD_ARG, "trusted/target header not supplied");
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");   // also on the graph-replay path, which skips upload_range's checks
    hipStream_t st = ctx->stream;
    // Three streams: the hashing chain (header hashes, hint, prove_subchain, reduce, finalize) on `st`; the commit check
    // (challenges, key table, signatures, tallies, skip conditions — latency bound at <= 100 signatures) beside it on `sb`; and
    // what the commit check needs but does not have to wait for in line (R decoded for the projective comparison, the trusted
    // set's hash and power sum) on `s3`; both tallies on `s4`.  All four are drained before the arena is rewound, also on the error paths (HostDrain).
    hipStream_t sb = ctx->stream2, s3 = ctx->stream3, s4 = ctx->stream4;
    HostDrain drain{{st, sb, s3, s4}};
    bsx_shared_ctx range{};
    range.start_block = trusted_block;
    range.end_block = target_block;
    memcpy(range.start_header_hash, input48 + 8, 32);
    RangeDev rd;
    DBuf dio, dv, dtv, dh, dok, dres, dtres, dskip, dth, dth2, drd;
    // small inputs and results: one block, one copy each way (SmallIO)
    SmallIO io;
    const size_t vbytes = (size_t)v_max * sizeof(bsx_validator);
    io.in_bytes = 256 + 2 * vbytes;
    io.out_off = io.in_bytes;
    RET(dio.alloc(io.out_off + SmallIO::OUT_BYTES));
    io.d = dio.as<uint8_t>();
    RET(ctx_hstage(ctx, io.out_off + SmallIO::OUT_BYTES, &io.h));
    memset(io.h, 0, 256);
    memcpy(io.h, &range, sizeof range);
    memcpy(io.h + 128, &latest_block, 8);
    memcpy(io.h + 256, target_validators, vbytes);
    memcpy(io.h + 256 + vbytes, trusted_validators, vbytes);
    memset(io.h + io.out_off, 0, SmallIO::OUT_BYTES);                   // the result block starts zeroed: same copy
    // The launch sequence below depends only on the circuit's shape, not on the request's bytes (those travel through the
    // staging block and the header buffer, whose addresses are stable while the arena is): the SECOND request of a shape is
    // captured into a hipGraph and every later one replays it — one hipGraphLaunch instead of ~25 launches, 8 event operations
    // and 2 copies.  Opt-in (BSX_TUNE_HOST_GRAPHS): on ROCm 7.2 the replay runs the three branches serially (0.56 vs 0.33 ms).
    HrGraphKey key{};
    key.J = nb_map_jobs; key.B = batch_size; key.V = v_max; key.hpr = n_headers - (trusted_block - first_height);
    key.span = target_block - trusted_block; key.chain_id_len = chain_id_len;
    if (chain_id_len) memcpy(key.chain_id, chain_id, chain_id_len);
    key.arena_base = ctx->arena.base; key.arena_cap = ctx->arena.cap; key.hstage = ctx->hstage; key.keytab = ctx->keytab;
    // the fixed-key table's rows against this request's keys, on the host (mirror of the keys the rows were built for)
    bool keys_same = ctx->keytab && ctx->keytab_rows == v_max && ctx->keytab_mirror_valid && ctx->keytab_mirror.size() == (size_t)v_max * 32;
    for (uint32_t i = 0; keys_same && i < v_max; i++) keys_same = memcmp(ctx->keytab_mirror.data() + 32 * (size_t)i, target_validators[i].pubkey, 32) == 0;
    // a captured launch sequence holds no key compare: it is recorded and replayed only for requests whose keys the table already has
    const bool graphable = !witness && ctx->graphs_enabled && ctx->arena.base && ctx->arena.overflow.empty() && keys_same;
    const bool replay = graphable && ctx->hr_exec && memcmp(&key, &ctx->hr_key, sizeof key) == 0;
    const bool capture = graphable && !replay && ctx->hr_seen && memcmp(&key, &ctx->hr_seen_key, sizeof key) == 0;
    if (replay) {
        const bsx_header* h0 = headers + (trusted_block - first_height);
        HIPCHK(hipMemcpyAsync(ctx->hr_d_headers, h0, key.hpr * sizeof(bsx_header), hipMemcpyHostToDevice, st));
        HIPCHK(hipGraphLaunch(ctx->hr_exec, st));
        SYNC();
    } else {
    ctx->hr_seen = graphable;
    ctx->hr_seen_key = key;
    if (capture) {
        // the headers come from the caller's (possibly pageable) memory: copied outside the graph, straight to the address the
        // captured kernels read
        if (ctx->hr_exec) { (void)hipGraphExecDestroy(ctx->hr_exec); ctx->hr_exec = nullptr; }
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    }
    struct EndCaptureOnError {
        hipStream_t s; bool active;
        ~EndCaptureOnError() { if (active) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); } }
    } cap_guard{st, capture};
    // witness: the COMMIT unit of the target commit and the SKIP unit (include/bsx_layout.h), written by the kernels of the commit
    // chain into these compact images; expanded behind both chains
    const bsx_witness_layout CL = bsx_commit_layout(v_max), SL = bsx_skip_layout(v_max);
    DBuf dccw, dscw;
    bsxk_unit_dst cwd = bsxk_unit(nullptr, CL), swd = bsxk_unit(nullptr, SL), swd_t = bsxk_unit(nullptr, SL, 1);
    if (witness) {
        RET(dccw.alloc(CL.compact_stride));
        RET(dscw.alloc(SL.compact_stride));
        cwd.base = dccw.as<uint8_t>();
        swd.base = swd_t.base = dscw.as<uint8_t>();
    }
    const bsxk_unit_dst *cwp = witness ? &cwd : nullptr, *swp = witness ? &swd : nullptr, *swtp = witness ? &swd_t : nullptr;
    if (witness) {                                                      // unit bytes no kernel writes (alignment gaps) must be zero
        HIPCHK(hipMemsetAsync(dccw.p, 0, CL.compact_stride, st));
        HIPCHK(hipMemsetAsync(dscw.p, 0, SL.compact_stride, st));
    }
    HIPCHK(hipMemcpyAsync(io.d, io.h, io.out_off + SmallIO::OUT_BYTES, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(ctx->ev_c, st));
    dbuf_alias(dv, io.d + 256);
    dbuf_alias(dtv, io.d + 256 + vbytes);
    dbuf_alias(dskip, io.dout(204));
    dbuf_alias(dres, io.dout(256));
    RET(dth.alloc(32));
    // the headers: a pageable megabyte is staged by the runtime (25 us on this thread) and then copied (25 us of DMA) — the head of
    // the hashing chain, the longer one.  Enqueued FIRST; everything else that has to be enqueued before the header hashing can start
    // (the side streams' waits, the R decoding) is enqueued while the DMA runs
    const bsxk_merkle_tap tap{target_block - trusted_block, io.d + offsetof(bsx_shared_ctx, end_header_hash), dth.as<uint8_t>(), nullptr, 0, nullptr};
    RET(upload_range(ctx, st, headers, first_height, n_headers, trusted_block, range, latest_block, rd, &io, capture, &tap, /*hash_later=*/true));
    HIPCHK(hipStreamWaitEvent(s3, ctx->ev_c, 0));                       // the commit check's inputs
    RET(drd.alloc(bsxk_ed25519_rdec_bytes(v_max)));
    // s3: R decoded (a square-root chain as long as a field inversion — the longest kernel of the commit check, and it needs only the
    // validator records)
    uint8_t* tab = nullptr;
    RET(ctx_keytable(ctx, v_max, &tab, sb));
    if (tab) HIPCHK(bsxk_ed25519_decode_r(s3, dv.as<bsx_validator>(), v_max, drd.p));
    HIPCHK(hipEventRecord(ctx->ev_d, s3));                              // the signature check waits for this
    // builder.skip (header_range.rs:42-48): the target header hash becomes ctx.end_header_hash (and the first output half) — stored by
    // the header hashing itself (the lane that joins the target header's tree), not by a k_fill_end_hash launch behind it
    RET(hash_range_headers(st, rd, &tap));
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_c, 0));
    HIPCHK(hipStreamWaitEvent(s4, ctx->ev_c, 0));
    RET(dh.alloc((size_t)v_max * 32));
    RET(dok.alloc(v_max));
    RET(dtres.alloc(sizeof(bsx_commit_result)));
    RET(dth2.alloc(32));
    HIPCHK(hipEventRecord(ctx->ev_a, st));                              // header hashes
    if (witness) {
        // header-field inclusion proofs of the target (chain id, height, validators_hash) and the trusted header (validators_hash)
        bsxk_field_proofs_args fa{};
        fa.n_items = 1; fa.headers = rd.headers.as<bsx_header>(); fa.headers_per_item = rd.hpr; fa.ranges = rd.ranges.as<bsx_shared_ctx>();
        fa.unit = swd; fa.n_proofs = BSX_SK_N_PROOFS; fa.zero_paths = ctx->zero_paths;
        static const uint8_t hsel[BSX_SK_N_PROOFS] = {1, 1, 1, 0}, fld[BSX_SK_N_PROOFS] = {1, BSX_BLOCK_HEIGHT_INDEX, 7, 7};
        for (uint32_t k = 0; k < BSX_SK_N_PROOFS; k++)
            fa.proofs[k] = bsxk_proof_spec{hsel[k], fld[k], (uint16_t)bsx_sk_proof_cap(k), bsx_sk_off_proof(v_max, k), BSX_SK_W_LEAF_LEN + k, 0u};
        HIPCHK(hipStreamWaitEvent(s3, ctx->ev_a, 0));                   // the range's headers are uploaded on `st`
        HIPCHK(bsxk_field_proofs(s3, &fa));
    }
    HIPCHK(hipEventRecord(ctx->ev_e, s3));                              // the skip conditions wait for this (field proofs)
    const uint8_t* d_target_hash = rd.hashes.as<uint8_t>() + (target_block - trusted_block) * 32;
    HIPCHK(bsxk_sha512_challenge(sb, dv.as<bsx_validator>(), v_max, dh.as<uint8_t>(), nullptr, v_max, cwp));
    // validator-set hash + total power of the target set: nothing here depends on the signatures — on its own stream beside the
    // challenges and the R decoding (on `sb` its 16 dependent compressions sat between the challenges and the verification: 50 us of
    // the commit chain, which had become the longer one)
    HIPCHK(bsxk_commit_tally(s4, dv.as<bsx_validator>(), 1, v_max, nullptr, nullptr, dres.as<bsx_commit_result>(), cwp));
    HIPCHK(hipEventRecord(ctx->ev_f, s4));
    // ... and the trusted set's behind it (only the skip conditions need it)
    HIPCHK(bsxk_commit_tally(s4, dtv.as<bsx_validator>(), 1, v_max, nullptr, nullptr, dtres.as<bsx_commit_result>(), swtp));
    HIPCHK(hipEventRecord(ctx->ev_g, s4));
    if (tab) {
        // the table rows are compared with the request's keys on the HOST (a mirror of the keys the rows were built for): an unchanged
        // validator set — every request of a prover's working day — launches neither the key compare nor the (no-op) build
        if (!keys_same) {
            HIPCHK(bsxk_ed25519_keytable(sb, dv.as<bsx_validator>(), v_max, tab));
            ctx->keytab_mirror.resize((size_t)v_max * 32);
            for (uint32_t i = 0; i < v_max; i++) memcpy(ctx->keytab_mirror.data() + 32 * (size_t)i, target_validators[i].pubkey, 32);
            ctx->keytab_mirror_valid = true;
        }
        HIPCHK(hipStreamWaitEvent(sb, ctx->ev_d, 0));
        HIPCHK(bsxk_ed25519_verify_keyed(sb, dv.as<bsx_validator>(), dh.as<uint8_t>(), v_max, v_max, tab, v_max, ctx->btab, dok.as<uint8_t>(), nullptr, drd.p, 0));   // one commit: the table rows ARE its keys, nothing is deferred
    } else {
        HIPCHK(bsxk_ed25519_verify(sb, dv.as<bsx_validator>(), dh.as<uint8_t>(), v_max, dok.as<uint8_t>()));
        HIPCHK(hipStreamWaitEvent(sb, ctx->ev_d, 0));
    }
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_a, 0));                       // header hashes (target hash, field-7 checks) from `st`
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_f, 0));                       // the target set's tally
    // the signature-dependent half of the tally (the validator leaves, the tree and the total ran EARLY, beside the R decoding)
    HIPCHK(bsxk_commit_sums(sb, dv.as<bsx_validator>(), 1, v_max, d_target_hash, dok.as<uint8_t>(), dres.as<bsx_commit_result>(), cwp));
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_e, 0));
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_g, 0));
    HIPCHK(bsxk_skip_check(sb, 1, v_max, rd.ranges.as<bsx_shared_ctx>(), rd.headers.as<bsx_header>(), rd.hpr, rd.hashes.as<uint8_t>(),
                           dv.as<bsx_validator>(), dtv.as<bsx_validator>(), dok.as<uint8_t>(), dres.as<bsx_commit_result>(),
                           dtres.as<bsx_commit_result>(), dskip.as<uint32_t>(), dth2.as<uint8_t>(), nullptr, chain_id, chain_id_len, swp));
    HIPCHK(hipEventRecord(ctx->ev_b, sb));
    // prove_data_commitment (header_range.rs:50-55) and the public output (:57-58); results stay in the block
    RET(run_data_commitment(ctx, st, nb_map_jobs, batch_size, rd, dth.as<uint8_t>(), nullptr, nullptr, nullptr, nullptr, witness, nullptr, &io));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_b, 0));
    DBuf dwc, dws;
    if (witness) {
        // both chains are done: the data commitment joins the SKIP unit's public outputs (header_range.rs:58), then the two units are
        // expanded behind the map jobs and reduce nodes
        const size_t nmr = (size_t)nb_map_jobs * bsx_map_layout(batch_size).n_elements + (size_t)(nb_map_jobs - 1) * bsx_reduce_layout().n_elements;
        HIPCHK(hipMemcpyAsync(dscw.as<uint8_t>() + 64, io.dout(0) + 32, 32, hipMemcpyDeviceToDevice, st));
        RET(dwc.alloc(CL.n_elements * 8 + 16));
        RET(dws.alloc(SL.n_elements * 8 + 16));
        HIPCHK(bsxk_expand_witness(st, &CL, 1, dccw.as<uint8_t>(), dwc.as<uint64_t>()));
        HIPCHK(bsxk_expand_witness(st, &SL, 1, dscw.as<uint8_t>(), dws.as<uint64_t>()));
        D2H(witness + nmr, dwc.p, CL.n_elements * 8);
        D2H(witness + nmr + CL.n_elements, dws.p, SL.n_elements * 8);
    }
    HIPCHK(hipMemcpyAsync(io.h + io.out_off, io.dout(0), SmallIO::OUT_BYTES, hipMemcpyDeviceToHost, st));
    static const bool trace_host = bsx_knob("BSX_TRACE_HOST", 0) != 0;       // experiments build: host enqueue time vs wait for the GPU
    const auto t_enq = std::chrono::steady_clock::now();
    if (capture) {
        hipGraph_t g = nullptr;
        cap_guard.active = false;
        HIPCHK(hipStreamEndCapture(st, &g));
        const hipError_t ie = hipGraphInstantiate(&ctx->hr_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) { ctx->hr_exec = nullptr; return fail(BSX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
        ctx->hr_key = key;
        ctx->hr_d_headers = rd.headers.p;
        const bsx_header* h0 = headers + (trusted_block - first_height);          // nothing has run yet: the capture only recorded
        HIPCHK(hipMemcpyAsync(ctx->hr_d_headers, h0, key.hpr * sizeof(bsx_header), hipMemcpyHostToDevice, st));
        HIPCHK(hipGraphLaunch(ctx->hr_exec, st));
    }
    SYNC();
    if (trace_host)
        fprintf(stderr, "bsx_header_range: enqueue %.1f us, wait %.1f us\n", std::chrono::duration<double, std::micro>(t_enq - t_entry).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count());
    }
    bsx_subchain result;
    bsx_commit_result cr;
    uint32_t hs, as, stv, skip;
    memcpy(&result, io.hout(64), sizeof result);
    memcpy(&hs, io.hout(192), 4);
    memcpy(&as, io.hout(196), 4);
    memcpy(&stv, io.hout(200), 4);
    memcpy(&skip, io.hout(204), 4);
    memcpy(&cr, io.hout(256), sizeof cr);
    const int rc = finish_data_commitment(io.hout(0), result, hs, as, stv, nullptr, output64, nullptr, nullptr);
    if (rc != BSX_OK && rc != BSX_ERR_ASSERT) return rc;
    const std::string dc_err = g_err;
    if (out_commit) *out_commit = cr;
    if (skip) return fail((int)skip, "skip verification failed: %s (bad signatures %u, first %u; bad messages %u; signed %llu of %llu; trusted overlap %llu)",
                          bsx_status_str((int)skip), cr.n_bad_signature, cr.first_bad_signature, cr.n_bad_message,
                          (unsigned long long)cr.signed_power, (unsigned long long)cr.total_power, (unsigned long long)cr.trusted_signed_power);
    if (rc) g_err = dc_err;
    return rc;
}

// capacity-checked forms: the witness buffer's size is part of the call (the round-4 witnesses are larger than round 3's)
int bsx_header_range_cap(bsx_ctx* ctx, uint32_t nb_map_jobs, uint32_t batch_size, const uint8_t input48[48], const bsx_header* headers,
                         uint64_t first_height, uint64_t n_headers, uint64_t latest_block, const bsx_validator* target_validators,
                         const bsx_validator* trusted_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len,
                         uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness, uint64_t witness_capacity_elements) {
    if (witness) {
        const uint64_t need = bsx_header_range_witness_elements(nb_map_jobs, batch_size, v_max);
        if (!need) return fail(BSX_ERR_BAD_ARG, "bsx_header_range_cap: invalid circuit shape");
        if (witness_capacity_elements < need)
            return fail(BSX_ERR_BAD_ARG, "witness buffer holds %llu elements, the whole-circuit witness needs %llu (map jobs + reduce nodes + COMMIT + SKIP units)",
                        (unsigned long long)witness_capacity_elements, (unsigned long long)need);
    }
    return bsx_header_range(ctx, nb_map_jobs, batch_size, input48, headers, first_height, n_headers, latest_block, target_validators, trusted_validators, v_max,
                            chain_id, chain_id_len, output64, out_commit, witness);
}
int bsx_verify_commits_cap(bsx_ctx* ctx, const bsx_validator* validators, uint32_t n_commits, uint32_t v_max, const uint8_t* header_hashes,
                           bsx_commit_result* out_results, uint8_t* out_sig_ok, uint64_t* witness, uint64_t witness_capacity_elements) {
    if (witness) {
        if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
        const uint64_t need = (uint64_t)n_commits * bsx_commit_layout(v_max).n_elements;
        if (witness_capacity_elements < need)
            return fail(BSX_ERR_BAD_ARG, "witness buffer holds %llu elements, %u COMMIT units need %llu", (unsigned long long)witness_capacity_elements, n_commits,
                        (unsigned long long)need);
    }
    return bsx_verify_commits(ctx, validators, n_commits, v_max, header_hashes, out_results, out_sig_ok, witness);
}
int bsx_next_header_cap(bsx_ctx* ctx, const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header, uint64_t latest_block,
                        const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id, uint32_t chain_id_len, uint8_t output64[64],
                        bsx_commit_result* out_commit, uint64_t* witness, uint64_t witness_capacity_elements) {
    if (witness) {
        const uint64_t need = bsx_next_header_witness_elements(v_max);
        if (!need) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
        if (witness_capacity_elements < need)
            return fail(BSX_ERR_BAD_ARG, "witness buffer holds %llu elements, COMMIT unit + STEP unit need %llu", (unsigned long long)witness_capacity_elements,
                        (unsigned long long)need);
    }
    return bsx_next_header(ctx, input40, prev_header, next_header, latest_block, next_validators, v_max, chain_id, chain_id_len, output64, out_commit, witness);
}

// CombinedStepCircuit::define (circuits/next_header.rs:25-46) as ONE enqueue: header hashes, the commit check of the next
// header's validator set (challenges, fixed-key Ed25519, tally + validator-set hash), the header-field inclusion proofs, the step
// conditions and prove_next_header_data_commitment (k_step_check) all run on the device; the host decodes two status words.
// witness (optional): the COMMIT unit then the STEP unit (include/bsx_layout.h), bsx_next_header_witness_elements(v_max) u64.
int bsx_next_header(bsx_ctx* ctx, const uint8_t input40[40], const bsx_header* prev_header, const bsx_header* next_header,
                    uint64_t latest_block, const bsx_validator* next_validators, uint32_t v_max, const uint8_t* chain_id,
                    uint32_t chain_id_len, uint8_t output64[64], bsx_commit_result* out_commit, uint64_t* witness) {
    HOST_ENTER();
    if (!input40 || !prev_header || !next_header || !next_validators || !output64) return fail(BSX_ERR_BAD_ARG, "null pointer");
    if (chain_id_len > 50 || (chain_id_len && !chain_id)) return fail(BSX_ERR_BAD_ARG, "chain_id: at most 50 bytes");
    if (v_max == 0 || (int)v_max > bsxk_tally_vmax()) return fail(BSX_ERR_UNSUPPORTED, "v_max %u not in 1..%d", v_max, bsxk_tally_vmax());
    uint64_t prev_block = 0;                                                    // next_header.rs:26 evm_read u64 (big endian)
    for (int i = 0; i < 8; i++) prev_block = prev_block << 8 | input40[i];
    const uint64_t next_block = prev_block + 1;                                 // :29-30
    hipStream_t st = ctx->stream;
    const bsx_witness_layout CL = bsx_commit_layout(v_max), TL = bsx_step_layout();
    const size_t vbytes = (size_t)v_max * sizeof(bsx_validator);
    // one staged block each way: [2 headers][input40 (64)][flags (64)][validators]
    const size_t in_bytes = 2 * sizeof(bsx_header) + 128 + vbytes;
    DBuf din, dhash, dst_, dh, dok, dres, dout, dccw, dscw, drd, dscr, dwit;
    uint8_t* hs = nullptr;
    RET(ctx_hstage(ctx, in_bytes, &hs));
    memcpy(hs, prev_header, sizeof(bsx_header));
    memcpy(hs + sizeof(bsx_header), next_header, sizeof(bsx_header));
    memset(hs + 2 * sizeof(bsx_header), 0, 128);
    memcpy(hs + 2 * sizeof(bsx_header), input40, 40);
    // builder.rs:415-423: the MAX_LEAVES = 1 hint returns a real data_hash proof iff prev < min(next, latest - 2) (input.rs:160-172)
    const uint64_t req_end = next_block < latest_block - 2 ? next_block : latest_block - 2;
    const uint32_t flags = prev_block < req_end ? 0u : 1u;
    memcpy(hs + 2 * sizeof(bsx_header) + 64, &flags, 4);
    memcpy(hs + 2 * sizeof(bsx_header) + 128, next_validators, vbytes);
    RET(din.alloc(in_bytes));
    RET(dhash.alloc(64));
    RET(dst_.alloc(16));
    RET(dh.alloc((size_t)v_max * 32));
    RET(dok.alloc(v_max));
    RET(dres.alloc(sizeof(bsx_commit_result)));
    RET(dout.alloc(64));
    RET(dccw.alloc(CL.compact_stride));
    RET(dscw.alloc(TL.compact_stride));
    RET(drd.alloc(bsxk_ed25519_rdec_bytes(v_max)));
    H2D(din.p, hs, in_bytes);
    HIPCHK(hipMemsetAsync(dst_.p, 0, 16, st));
    HIPCHK(hipMemsetAsync(dccw.p, 0, CL.compact_stride, st));
    HIPCHK(hipMemsetAsync(dscw.p, 0, TL.compact_stride, st));
    const bsx_header* d_hdr = din.as<bsx_header>();
    const uint8_t* d_in40 = din.as<uint8_t>() + 2 * sizeof(bsx_header);
    const uint32_t* d_flags = reinterpret_cast<const uint32_t*>(d_in40 + 64);
    const bsx_validator* d_val = reinterpret_cast<const bsx_validator*>(d_in40 + 128);
    uint32_t* d_st = dst_.as<uint32_t>();                                       // [0] header status, [1] step status, [2] A10
    const bsxk_unit_dst cwd = bsxk_unit(dccw.as<uint8_t>(), CL), swd = bsxk_unit(dscw.as<uint8_t>(), TL);
    // Four streams, as in bsx_header_range: R decoding (the longest kernel, needs only the validator records), the challenges, the
    // validator-set tally + the header-field proofs, and the two header hashes run side by side; the verification waits for the first
    // two, the sums for the tally and the header hash, the step conditions for everything (one stream: 0.36 ms per call; now 0.2x)
    hipStream_t sb = ctx->stream2, s3 = ctx->stream3, s4 = ctx->stream4;
    HostDrain drain{{st, sb, s3, s4}};                                  // before the first side-stream launch (see bsx_header_range)
    HIPCHK(hipEventRecord(ctx->ev_c, st));                              // inputs uploaded, units cleared
    HIPCHK(hipStreamWaitEvent(s3, ctx->ev_c, 0));
    uint8_t* tab = nullptr;
    RET(ctx_keytable(ctx, v_max, &tab, sb));
    if (tab) HIPCHK(bsxk_ed25519_decode_r(s3, d_val, v_max, drd.p));
    HIPCHK(hipEventRecord(ctx->ev_d, s3));
    HIPCHK(bsxk_header_merkle(st, d_hdr, 2, dhash.as<uint8_t>(), nullptr, nullptr, nullptr, d_st, 0, 0));
    HIPCHK(hipEventRecord(ctx->ev_a, st));                              // header hashes
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_c, 0));
    HIPCHK(hipStreamWaitEvent(s4, ctx->ev_c, 0));
    // builder.step (:32-36) [UPSTREAM tendermintx v1.0.0]: the commit of the next header
    HIPCHK(bsxk_sha512_challenge(sb, d_val, v_max, dh.as<uint8_t>(), nullptr, v_max, &cwd));
    // validator leaves, the masked tree and the total power: nothing there depends on the signatures
    HIPCHK(bsxk_commit_tally(s4, d_val, 1, v_max, nullptr, nullptr, dres.as<bsx_commit_result>(), &cwd));
    HIPCHK(hipEventRecord(ctx->ev_f, s4));
    {
        bsxk_field_proofs_args fa{};
        fa.n_items = 1; fa.headers = d_hdr; fa.headers_per_item = 2; fa.flags = d_flags;
        fa.unit = swd; fa.n_proofs = BSX_ST_N_PROOFS; fa.zero_paths = ctx->zero_paths;
        static const uint8_t hsel[BSX_ST_N_PROOFS] = {1, 1, 1, 1, 0, 0};
        static const uint8_t fld[BSX_ST_N_PROOFS] = {1, BSX_BLOCK_HEIGHT_INDEX, 7, BSX_LAST_BLOCK_ID_INDEX, 8, BSX_DATA_HASH_INDEX};
        for (uint32_t k = 0; k < BSX_ST_N_PROOFS; k++)
            fa.proofs[k] = bsxk_proof_spec{hsel[k], fld[k], (uint16_t)bsx_st_proof_cap(k), bsx_st_off_proof(k), BSX_ST_W_LEAF_LEN + k, k == 5 ? 1u : 0u};
        HIPCHK(bsxk_field_proofs(s4, &fa));
    }
    HIPCHK(hipEventRecord(ctx->ev_g, s4));
    if (tab) {
        // the table rows against this request's keys on the HOST (mirror of the keys the rows were built for), as in bsx_header_range
        bool keys_same = ctx->keytab_rows == v_max && ctx->keytab_mirror_valid && ctx->keytab_mirror.size() == (size_t)v_max * 32;
        for (uint32_t i = 0; keys_same && i < v_max; i++) keys_same = memcmp(ctx->keytab_mirror.data() + 32 * (size_t)i, next_validators[i].pubkey, 32) == 0;
        if (!keys_same) {
            HIPCHK(bsxk_ed25519_keytable(sb, d_val, v_max, tab));
            ctx->keytab_mirror.resize((size_t)v_max * 32);
            for (uint32_t i = 0; i < v_max; i++) memcpy(ctx->keytab_mirror.data() + 32 * (size_t)i, next_validators[i].pubkey, 32);
            ctx->keytab_mirror_valid = true;
        }
        HIPCHK(hipStreamWaitEvent(sb, ctx->ev_d, 0));
        HIPCHK(bsxk_ed25519_verify_keyed(sb, d_val, dh.as<uint8_t>(), v_max, v_max, tab, v_max, ctx->btab, dok.as<uint8_t>(), nullptr, drd.p, 0));   // one commit: the table IS its keys
    } else {
        HIPCHK(bsxk_ed25519_verify(sb, d_val, dh.as<uint8_t>(), v_max, dok.as<uint8_t>()));
        HIPCHK(hipStreamWaitEvent(sb, ctx->ev_d, 0));
    }
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_a, 0));                       // the next header's hash (the signed messages carry it)
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_f, 0));
    HIPCHK(bsxk_commit_sums(sb, d_val, 1, v_max, dhash.as<uint8_t>() + 32, dok.as<uint8_t>(), dres.as<bsx_commit_result>(), &cwd));
    HIPCHK(hipStreamWaitEvent(sb, ctx->ev_g, 0));
    {
        bsxk_step_args sa{};
        sa.headers = d_hdr; sa.hashes = dhash.as<uint8_t>(); sa.input40 = d_in40; sa.commit = dres.as<bsx_commit_result>();
        sa.step_status = d_st + 1; sa.dc_status = d_st + 2; sa.output64 = dout.as<uint8_t>(); sa.unit = swd;
        sa.chain_id_len = chain_id_len;
        if (chain_id_len) memcpy(sa.chain_id, chain_id, chain_id_len);
        HIPCHK(bsxk_step_check(sb, &sa));
    }
    HIPCHK(hipEventRecord(ctx->ev_b, sb));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_b, 0));
    if (witness) {
        RET(dwit.alloc((CL.n_elements + TL.n_elements) * 8 + 32));
        RET(dscr.alloc(TL.n_elements * 8 + 16));
        HIPCHK(bsxk_expand_witness(st, &CL, 1, dccw.as<uint8_t>(), dwit.as<uint64_t>()));
        HIPCHK(bsxk_expand_witness(st, &TL, 1, dscw.as<uint8_t>(), dscr.as<uint64_t>()));
        D2H(witness, dwit.p, CL.n_elements * 8);
        D2H(witness + CL.n_elements, dscr.p, TL.n_elements * 8);
    }
    uint8_t o64[64];
    bsx_commit_result cr;
    uint32_t stw[4] = {0, 0, 0, 0};
    StagedD2H back(ctx, st);
    RET(back.copy(o64, dout.p, 64));
    RET(back.copy(&cr, dres.p, sizeof cr));
    RET(back.copy(stw, dst_.p, 16));
    RET(back.sync());
    if (stw[0] & 1u) return fail(BSX_ERR_BAD_HEADER, "a packed header violates the field-size rules of bsx_header");
    if (out_commit) *out_commit = cr;
    if (cr.power_overflow)
        return fail(BSX_ERR_BAD_ARG, "commit 0: the voting powers add up to more than MaxTotalVotingPower (MaxInt64 / 8); tallies are meaningless");
    if (latest_block < 2) return fail(BSX_ERR_BAD_ARG, "latest_block < 2");
    memcpy(output64, o64, 64);                                                  // :44-45
    const int stc = (int)(stw[1] & 0xffu);
    if (stc) {
        static const char* why[] = {"", "voting power overflow", "prev header does not hash to prev_header_hash", "next header's height is not prev_block_number + 1",
                                    "next header's chain id is not the circuit's CHAIN_ID_BYTES", "a signed validator's signature or message is bad",
                                    "validator set does not hash to the next header's validators_hash",
                                    "validator set does not hash to the prev header's next_validators_hash",
                                    "next header's last_block_id does not point at the prev header", "less than 2/3 of the voting power signed"};
        const uint32_t w = (stw[1] >> 8) & 0xffu;
        return fail(stc, "step verification failed: %s", w < 10 ? why[w] : "");
    }
    if (stw[2])
        return fail(BSX_ERR_ASSERT, "prove_next_header_data_commitment: data_hash proof root != prev_header_hash (A10, builder.rs:434)");
    return BSX_OK;
}

}  // extern "C"
