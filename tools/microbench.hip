// tools/microbench.hip — measured ceilings on the box: SHA-256 compressions/s (inline vs called body, by occupancy)
// and pure-store HBM bandwidth (plain / nontemporal dwordx4).  Used to price the kernels in DESIGN.md; not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../blobstreamx_amd/csrc/sha256.h"
using namespace bsx;

template <bool CALL>
__global__ void k_sha(uint32_t* out, int iters) {
    Digest st; Block16 b;
    for (int i = 0; i < 8; i++) st.w[i] = threadIdx.x * 31 + i + blockIdx.x;
    for (int i = 0; i < 16; i++) b.w[i] = threadIdx.x * 17 + i;
    for (int r = 0; r < iters; r++) {
        if (CALL) st = sha256_compress_fn(st, b);
        else { uint32_t s[8], w[16]; for (int i=0;i<8;i++) s[i]=st.w[i]; for (int i=0;i<16;i++) w[i]=b.w[i];
               // inline copy of the round function
               uint32_t a=s[0],bb=s[1],c=s[2],d=s[3],e=s[4],f=s[5],g=s[6],h=s[7];
               #pragma unroll
               for (int i=0;i<64;i++){ if(i>=16){uint32_t w15=w[(i+1)&15],w2=w[(i+14)&15]; w[i&15]=w[i&15]+xor3(rotr32(w15,7),rotr32(w15,18),w15>>3)+w[(i+9)&15]+xor3(rotr32(w2,17),rotr32(w2,19),w2>>10);}
                 uint32_t t1=h+xor3(rotr32(e,6),rotr32(e,11),rotr32(e,25))+(g^(e&(f^g)))+sha256_k(i)+w[i&15];
                 uint32_t t2=xor3(rotr32(a,2),rotr32(a,13),rotr32(a,22))+(bb^((a^bb)&(c^bb)));
                 h=g;g=f;f=e;e=d+t1;d=c;c=bb;bb=a;a=t1+t2;}
               st.w[0]+=a;st.w[1]+=bb;st.w[2]+=c;st.w[3]+=d;st.w[4]+=e;st.w[5]+=f;st.w[6]+=g;st.w[7]+=h; }
        b.w[r & 15] ^= st.w[3];
    }
    uint32_t x = 0; for (int i = 0; i < 8; i++) x ^= st.w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int MODE>
__global__ void k_store(ulonglong2* out, size_t n16, unsigned long long v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        ulonglong2 x = make_ulonglong2(v + i, v ^ i);
        if (MODE == 0) out[i] = x;
        else { typedef unsigned long long v2u64 __attribute__((ext_vector_type(2))); v2u64 y = {x.x, x.y}; __builtin_nontemporal_store(y, reinterpret_cast<v2u64*>(&out[i])); }
    }
}
__global__ void k_copy(const ulonglong2* in, ulonglong2* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = in[i];
}
static float timeit(void (*f)(void*), void* a, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(a); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; i++) f(a); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
struct ShaArgs { uint32_t* out; int blocks, threads, iters; bool call; };
static void run_sha(void* p) { ShaArgs* a = (ShaArgs*)p;
    if (a->call) hipLaunchKernelGGL(k_sha<true>, dim3(a->blocks), dim3(a->threads), 0, 0, a->out, a->iters);
    else hipLaunchKernelGGL(k_sha<false>, dim3(a->blocks), dim3(a->threads), 0, 0, a->out, a->iters); }
struct StArgs { ulonglong2* out; const ulonglong2* in; size_t n16; int mode, blocks; };
static void run_st(void* p) { StArgs* a = (StArgs*)p;
    if (a->mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->n16, 7ull);
    else if (a->mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->n16, 7ull);
    else hipLaunchKernelGGL(k_copy, dim3(a->blocks), dim3(256), 0, 0, a->in, a->out, a->n16); }
int main() {
    uint32_t* out; hipMalloc(&out, 64u << 20);
    const int iters = 200;
    for (int call = 0; call < 2; call++)
        for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
            ShaArgs a{out, 256 * wps, 256, iters, call != 0};
            float ms = timeit(run_sha, &a, 3);
            double comp = (double)a.blocks * a.threads * iters;
            printf("sha256 %s waves/SIMD=%d : %.3f ms  %.2f G compressions/s\n", call ? "called " : "inlined", wps, ms, comp / ms / 1e6);
        }
    size_t bytes = 4ull << 30; ulonglong2 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    for (int mode = 0; mode < 3; mode++) for (int blocks : {2048, 8192, 65536}) {
        StArgs s{a, b, bytes / 16, mode, blocks};
        float ms = timeit(run_st, &s, 5);
        printf("%s blocks=%d : %.3f ms  %.0f GB/s%s\n", mode == 0 ? "store plain" : mode == 1 ? "store nontemporal" : "copy (read+write)", blocks, ms,
               (mode == 2 ? 2.0 : 1.0) * bytes / ms / 1e6, mode == 2 ? " (sum of both directions)" : "");
    }
    return 0;
}
