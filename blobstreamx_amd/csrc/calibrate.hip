// calibrate.hip — bsx_calibrate: the integer-ALU and store ceilings of THIS device, measured in ~50 ms, for roofline reporting.
//
// The hot path's kernels are integer-ALU bound (SHA-256/512, GF(2^255-19), Goldilocks) or HBM-store bound (witness expansion);
// their ceilings differ by up to 15 % between boxes of the same model, so a roofline fraction has to be priced against
// ceilings measured in the same process on the same device.  Each body below is the arithmetic a kernel is made
// of — the library's own device functions — alone, at 8 waves per SIMD, with no memory traffic.
#include <atomic>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>

#include "api_internal.h"
#include "fe25519.h"
#include "goldilocks.h"
#include "sha256.h"
#include "sha512.h"

using bsxapi::fail;
using bsxapi::use;

namespace bsx {

// MODE 0: v_add_u32 (full-rate 2-source VALU)   1: v_mad_u64_u32   2: v_alignbit_b32 (every SHA rotate)
template <int MODE>
__global__ void k_cal_issue(uint32_t* out, int iters) {
    uint32_t a[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; q[i] = a[i] * 0x100000001ull; }
    const uint32_t m = threadIdx.x | 3, k = blockIdx.x | 5;
    for (int r = 0; r < iters; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(m), "v"(k) : "vcc");
                if (MODE == 2) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[i]));
            }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// BODY 0: sha256_compress_fn   1: sha512_compress   2: fe_mul   3: fe_sq   4: gl_mul
template <int BODY>
__global__ void k_cal_body(uint32_t* out, int iters) {
    uint32_t x = 0;
    if (BODY == 0) {
        Digest st;
        Block16 b;
        for (int i = 0; i < 8; i++) st.w[i] = threadIdx.x * 31 + i + blockIdx.x;
        for (int i = 0; i < 16; i++) b.w[i] = threadIdx.x * 17 + i;
        for (int r = 0; r < iters; r++) { st = sha256_compress_fn(st, b); b.w[r & 15] ^= st.w[3]; }
        for (int i = 0; i < 8; i++) x ^= st.w[i];
    } else if (BODY == 1) {
        uint64_t st[8], w[16];
        for (int i = 0; i < 8; i++) st[i] = threadIdx.x * 31 + i + blockIdx.x;
        for (int i = 0; i < 16; i++) w[i] = threadIdx.x * 17 + i;
        for (int r = 0; r < iters; r++) { uint64_t ww[16]; for (int i = 0; i < 16; i++) ww[i] = w[i] ^ st[i & 7]; sha512_compress(st, ww); }
        for (int i = 0; i < 8; i++) x ^= (uint32_t)st[i] ^ (uint32_t)(st[i] >> 32);
    } else if (BODY == 2 || BODY == 3) {
        fe f, g;
        for (int i = 0; i < 10; i++) { f.v[i] = (int32_t)((threadIdx.x * 2654435761u + i * 40503u) & 0x1ffffff); g.v[i] = (int32_t)((blockIdx.x * 97u + i * 7919u + threadIdx.x) & 0x1ffffff); }
        for (int r = 0; r < iters; r++) {
            if (BODY == 2) { f = fe_mul(f, g); g = fe_mul(g, f); }
            else { f = fe_sq(f); g = fe_sq(g); }
        }
        for (int i = 0; i < 10; i++) x ^= (uint32_t)f.v[i] ^ (uint32_t)g.v[i];
    } else {
        uint64_t v[4] = {threadIdx.x + 3ull, blockIdx.x * 0x9e3779b97f4a7c15ull + 1, threadIdx.x * 0x100000001ull + 7, 0xdeadbeefcafef00dull};
        for (int r = 0; r < iters; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = gl_mul(v[i], v[(i + 1) & 3] | 1);
        for (int i = 0; i < 4; i++) x ^= (uint32_t)v[i] ^ (uint32_t)(v[i] >> 32);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

__global__ void k_cal_store(ulonglong2* out, size_t n16, unsigned long long v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    for (; i < n16; i += stride) {
        v2u64 y = {v + i, v ^ i};
        __builtin_nontemporal_store(y, reinterpret_cast<v2u64*>(&out[i]));
    }
}

// one wave that keeps its queue busy for `ticks` of the 100 MHz wall clock (bsxk_queue_groups)
__global__ void k_spin(unsigned long long ticks, uint32_t* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0) out[0] = 1;
}

}  // namespace bsx

// How many distinct HARDWARE queues a set of HIP streams sits on: HIP binds streams to GPU_MAX_HW_QUEUES queues (default 4), and
// kernels of streams that share one run one after the other whatever the stream flags say.  Measured, not read from the
// environment: stream s is put in the group of the first representative it serialises with — two single-wave kernels that each
// spin 0.3 ms take 0.3 ms together on different queues and 0.6 ms on one.  groups[i] (optional) receives stream i's group.
extern "C" uint32_t bsxk_compute_units() {
    static std::atomic<uint32_t> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    uint32_t v = cached[dev].load(std::memory_order_relaxed);
    if (!v) {
        int n = 0;
        v = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? (uint32_t)n : 256u;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int bsxk_queue_groups(hipStream_t* streams, uint32_t n, uint32_t* groups) {
    constexpr unsigned long long TICKS = 30000;          // 0.3 ms at 100 MHz
    uint32_t reps[64], n_groups = 0;
    auto pair_ms = [&](hipStream_t a, hipStream_t b, double* ms) -> int {
        HIPCHK(hipStreamSynchronize(a));
        HIPCHK(hipStreamSynchronize(b));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(bsx::k_spin, dim3(1), dim3(64), 0, a, TICKS, (uint32_t*)nullptr);
        hipLaunchKernelGGL(bsx::k_spin, dim3(1), dim3(64), 0, b, TICKS, (uint32_t*)nullptr);
        HIPCHK(hipStreamSynchronize(a));
        HIPCHK(hipStreamSynchronize(b));
        *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return BSX_OK;
    };
    if (n > 64) n = 64;
    for (uint32_t s = 0; s < n; s++) {
        uint32_t g = n_groups;
        for (uint32_t k = 0; k < n_groups && g == n_groups; k++) {
            double ms = 0;
            RET(pair_ms(streams[reps[k]], streams[s], &ms));
            if (ms > 0.48) g = k;                        // serialised: same hardware queue
        }
        if (g == n_groups) reps[n_groups++] = s;
        if (groups) groups[s] = g;
    }
    return (int)n_groups;
}

namespace {
template <typename F>
int time_launch(hipStream_t st, hipEvent_t e0, hipEvent_t e1, F launch, int reps, double* ms_out) {
    launch();                                   // warm-up (code fetch, clocks)
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) launch();
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(hipGetLastError());
    *ms_out = (double)ms / reps;
    return BSX_OK;
}
}  // namespace

extern "C" int bsx_calibrate(bsx_ctx* ctx, bsx_calibration* out) {
    RET(use(ctx));
    if (!out) return fail(BSX_ERR_BAD_ARG, "bsx_calibrate: null out");
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mu);
    using namespace bsx;
    memset(out, 0, sizeof *out);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int blocks = cus * 8, threads = 256;          // 8 waves per SIMD
    hipStream_t st = ctx->stream;
    uint32_t* buf = nullptr;
    const size_t store_bytes = 1ull << 30;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&buf), store_bytes));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t ee = hipEventCreate(&e0);
    if (ee == hipSuccess) ee = hipEventCreate(&e1);
    int rc = ee == hipSuccess ? BSX_OK : fail(BSX_ERR_HIP, "bsx_calibrate: hipEventCreate: %s", hipGetErrorString(ee));
    double ms = 0;
    const double lanes = (double)blocks * threads;
#define CAL(field, launch_expr, units_per_launch)                                                    \
    if (rc == BSX_OK) {                                                                              \
        rc = time_launch(st, e0, e1, [&] { launch_expr; }, 3, &ms);                                  \
        if (rc == BSX_OK) out->field = (units_per_launch) / (ms * 1e-3);                             \
    }
    CAL(valu_add_u32_lane_ops_per_s, hipLaunchKernelGGL(k_cal_issue<0>, dim3(blocks), dim3(threads), 0, st, buf, 100), lanes * 100 * 64)
    CAL(valu_mad_u64_u32_lane_ops_per_s, hipLaunchKernelGGL(k_cal_issue<1>, dim3(blocks), dim3(threads), 0, st, buf, 100), lanes * 100 * 64)
    CAL(valu_alignbit_lane_ops_per_s, hipLaunchKernelGGL(k_cal_issue<2>, dim3(blocks), dim3(threads), 0, st, buf, 100), lanes * 100 * 64)
    CAL(sha256_compress_per_s, hipLaunchKernelGGL(k_cal_body<0>, dim3(blocks), dim3(threads), 0, st, buf, 60), lanes * 60)
    CAL(sha512_compress_per_s, hipLaunchKernelGGL(k_cal_body<1>, dim3(blocks), dim3(threads), 0, st, buf, 20), lanes * 20)
    CAL(fe25519_mul_per_s, hipLaunchKernelGGL(k_cal_body<2>, dim3(blocks), dim3(threads), 0, st, buf, 200), lanes * 200 * 2)
    CAL(fe25519_sq_per_s, hipLaunchKernelGGL(k_cal_body<3>, dim3(blocks), dim3(threads), 0, st, buf, 200), lanes * 200 * 2)
    CAL(goldilocks_mul_per_s, hipLaunchKernelGGL(k_cal_body<4>, dim3(blocks), dim3(threads), 0, st, buf, 1000), lanes * 1000 * 4)
    CAL(hbm_store_bytes_per_s, hipLaunchKernelGGL(k_cal_store, dim3(65536), dim3(256), 0, st, reinterpret_cast<ulonglong2*>(buf), store_bytes / 16, 7ull),
        (double)store_bytes)
#undef CAL
    out->compute_units = (uint32_t)cus;
    out->clock_mhz = (uint32_t)(prop.clockRate / 1000);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return rc;
}
