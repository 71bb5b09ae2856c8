// bsx_common.h — shared macros for the gfx950 device code.
// Everything arithmetic is written as BSX_HDI (host+device inline) so that tests/hostcheck can run the
// very same source on the CPU (a test harness, never a product path: libbsx.so has no host compute).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define BSX_HDI __host__ __device__ __forceinline__
#define BSX_HD_NOINLINE static __host__ __device__ __attribute__((noinline))
#else
#define BSX_HDI static inline __attribute__((always_inline))
#define BSX_HD_NOINLINE static __attribute__((noinline))
#endif

// The device code is written for gfx950 (CDNA4) ONLY and relies on three of its properties:
//   * s_barrier counts the waves of a workgroup that are still alive: k_batch_finish / k_commit_tally let whole waves RETURN before
//     later __syncthreads() of their workgroup (divergent barrier participation is undefined in the portable HIP model);
//   * global memory serves misaligned dword / dwordx4 accesses (unaligned-access mode; tools/unaligned_test.hip): the 2-byte aligned
//     proof records of the packed witness image move as single 16-byte loads and stores;
//   * wave64, v_bitop3 / v_alignbit / v_mad_u64_u32 rates as measured.
// Every launch form that depends on them is pinned by tests over all shapes (tests/test_gpu_parity.py B = 1 .. 256,
// tests/test_gpu_units.py V = 1 .. 512).  There is no portable fallback by design: another target must not compile this silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libbsx device code is gfx950-only (wave-exit before s_barrier, unaligned global access): build with --offload-arch=gfx950"
#endif

namespace bsx {

// ({hi,lo} >> (8*bytes)) & 0xffffffff — compiles to one v_alignbit_b32 / v_alignbyte_b32
BSX_HDI uint32_t funnel_r(uint32_t hi, uint32_t lo, int bits) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> bits);
}
BSX_HDI uint32_t rotr32(uint32_t x, int n) { return funnel_r(x, x, n); }
BSX_HDI uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
// three-input xor: one v_bitop3_b32 (truth table 0x96) on gfx950
BSX_HDI uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

}  // namespace bsx
