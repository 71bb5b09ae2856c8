// api_internal.h — helpers shared by the host-side translation units of libbsx.so (api.hip, api_poseidon.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/bsx.h"

struct bsx_ctx {
    int device;
    hipStream_t stream;
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

// device buffer with the lifetime of one host-tier call
struct DBuf {
    void* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return fail(BSX_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        return BSX_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
}  // namespace bsxapi
