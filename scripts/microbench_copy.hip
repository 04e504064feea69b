// microbench_copy.hip -- streaming read+write rates vs workgroup shape (aid for tuning pb_expand)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush4 __attribute__((ext_vector_type(4)));

// each WG streams a contiguous chunk: reads val (8B) + col (2B), writes 8B; D vector steps in flight
template <int THREADS, int D, bool WRITE>
__global__ __launch_bounds__(THREADS) void k(const double* __restrict__ val, const uint16_t* __restrict__ col,
                                             double* __restrict__ out, size_t chunk) {
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 1.0;
    __syncthreads();
    const size_t b = (size_t)blockIdx.x * chunk, e = b + chunk;
    for (size_t p = b + 4 * threadIdx.x; p < e; p += (size_t)D * 4 * THREADS) {
        dbl2 a[D], c[D]; ush4 j[D];
#pragma unroll
        for (int u = 0; u < D; ++u) { size_t q = p + (size_t)u * 4 * THREADS; if (q < e) { a[u] = __builtin_nontemporal_load((const dbl2*)(val + q)); c[u] = __builtin_nontemporal_load((const dbl2*)(val + q + 2)); j[u] = __builtin_nontemporal_load((const ush4*)(col + q)); } else { a[u] = 0; c[u] = 0; j[u] = 0; } }
#pragma unroll
        for (int u = 0; u < D; ++u) { size_t q = p + (size_t)u * 4 * THREADS; if (q < e) {
            dbl2 r0 = a[u] * (double)j[u].x, r1 = c[u] * (double)j[u].z;
            if (WRITE) { __builtin_nontemporal_store(r0, (dbl2*)(out + q)); __builtin_nontemporal_store(r1, (dbl2*)(out + q + 2)); }
            else if (r0.x == 1.234e300) out[0] = r1.y; } }
    }
}
template <int THREADS, int D, bool WRITE>
int run(const char* tag, const double* val, const uint16_t* col, double* out, size_t n, size_t chunk, size_t ldsBytes) {
    CK(hipFuncSetAttribute((const void*)k<THREADS, D, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    unsigned grid = (unsigned)(n / chunk);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<THREADS, D, WRITE>), dim3(grid), dim3(THREADS), ldsBytes, 0, val, col, out, chunk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<THREADS, D, WRITE>), dim3(grid), dim3(THREADS), ldsBytes, 0, val, col, out, chunk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    double bytes = (double)n * (WRITE ? 18 : 10);
    printf("%s,threads=%d,depth=%d,lds=%zu,chunk=%zu,ms=%.3f,TBps=%.2f\n", tag, THREADS, D, ldsBytes, chunk, ms, bytes / ms * 1e-9);
    return 0;
}
int main() {
    const size_t n = 1ull << 30;   // 1 Gi entries
    double *val, *out; uint16_t* col;
    CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&col, n * 2));
    CK(hipMemset(val, 0, n * 8)); CK(hipMemset(col, 0, n * 2));
    run<1024, 4, true>("rw", val, col, out, n, 1 << 17, 128 << 10);
    run<1024, 4, true>("rw", val, col, out, n, 1 << 17, 64 << 10);
    run<1024, 4, true>("rw", val, col, out, n, 1 << 17, 1024);
    run<1024, 2, true>("rw", val, col, out, n, 1 << 17, 1024);
    run<512, 4, true>("rw", val, col, out, n, 1 << 16, 64 << 10);
    run<512, 4, true>("rw", val, col, out, n, 1 << 16, 1024);
    run<256, 4, true>("rw", val, col, out, n, 1 << 15, 1024);
    run<256, 2, true>("rw", val, col, out, n, 1 << 14, 1024);
    run<256, 1, true>("rw", val, col, out, n, 1 << 14, 1024);
    run<1024, 4, false>("ro", val, col, out, n, 1 << 17, 128 << 10);
    run<1024, 4, false>("ro", val, col, out, n, 1 << 17, 1024);
    run<256, 4, false>("ro", val, col, out, n, 1 << 15, 1024);
    run<256, 2, false>("ro", val, col, out, n, 1 << 14, 1024);
    return 0;
}
