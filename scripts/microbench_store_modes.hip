// microbench_store_modes.hip -- phase 1's byte mix (10 B read : 8 B written per entry, 1 Gi entries) with every cache-policy
// flavour of the product store: does any of them lift the write side of HBM?  (phase 1 time ~ R/6.2 + W/4.5 TB/s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ void store16(dbl2* p, dbl2 v) {
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ val, const uint16_t* __restrict__ col, double* __restrict__ out, size_t chunk) {
    const size_t b = (size_t)blockIdx.x * chunk, e = b + chunk;
    for (size_t p = b + 2 * threadIdx.x; p < e; p += 8 * 2048) {
        dbl2 a[8]; ush2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t q = p + (size_t)u * 2048; a[u] = __builtin_nontemporal_load((const dbl2*)(val + q)); c[u] = __builtin_nontemporal_load((const ush2*)(col + q)); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t q = p + (size_t)u * 2048; dbl2 o; o.x = a[u].x * (double)c[u].x; o.y = a[u].y * (double)c[u].y; store16<MODE>((dbl2*)(out + q), o); }
    }
}
template <int MODE> int run(const char* tag, const double* val, const uint16_t* col, double* out, size_t n) {
    const size_t chunk = 1 << 17;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<(unsigned)(n / chunk), 1024>>>(val, col, out, chunk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) k<MODE><<<(unsigned)(n / chunk), 1024>>>(val, col, out, chunk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%s,ms=%.3f,TBps=%.2f\n", tag, ms, (double)n * 18 / ms * 1e-9);
    return 0;
}
int main() {
    const size_t n = 1ull << 30;
    double *val, *out; uint16_t* col;
    CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&out, n * 8 + (64 << 20))); CK(hipMalloc(&col, n * 2));
    CK(hipMemset(val, 0, n * 8)); CK(hipMemset(col, 0, n * 2));
    printf("store_flavour,ms,TBps (10 B read + 8 B written per entry, 1 Gi entries)\n");
    run<0>("plain", val, col, out, n); run<1>("nt", val, col, out, n); run<2>("sc1", val, col, out, n); run<3>("sc0 sc1", val, col, out, n);
    run<4>("sc0", val, col, out, n); run<5>("sc1 nt", val, col, out, n); run<6>("sc0 sc1 nt", val, col, out, n); run<7>("sc0 nt", val, col, out, n);
    // does the relative placement of the written stream matter (channel / bank interleave)?  nt stores, out shifted
    for (size_t off : {(size_t)0, (size_t)32, (size_t)512, (size_t)4096, (size_t)65536, (size_t)(1 << 20) + 4096, (size_t)(16 << 20) + 65536 + 512}) {
        char tag[64]; snprintf(tag, sizeof tag, "nt out+%zu B", off * 8);
        run<1>(tag, val, col, out + off, n - (32 << 20));
    }
    return 0;
}
