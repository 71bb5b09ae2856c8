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
