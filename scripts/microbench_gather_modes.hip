// microbench_gather_modes.hip -- does any cache-policy bit change the 128-B-line cost of an 8-byte gather
// that misses L2?  (plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt, via inline asm on global_load_dwordx2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int MODE>
__device__ __forceinline__ double gload(const double* p) {
    double v;
    if (MODE == 0) return *p;
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// 8 independent gathers per lane issued back to back, ONE wait (asm with explicit waits would serialise; use 8 asm loads + 1 wait)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ idx, const double* __restrict__ x, double* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = (i + u * 256 < n) ? __builtin_nontemporal_load(idx + i + u * 256) : 0;
    double v[8];
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[c[u]];
    } else {
        const double* p0 = x + c[0]; const double* p1 = x + c[1]; const double* p2 = x + c[2]; const double* p3 = x + c[3];
        const double* p4 = x + c[4]; const double* p5 = x + c[5]; const double* p6 = x + c[6]; const double* p7 = x + c[7];
#define LD8(BITS) asm volatile( \
        "global_load_dwordx2 %0, %8, off " BITS "\n\tglobal_load_dwordx2 %1, %9, off " BITS "\n\tglobal_load_dwordx2 %2, %10, off " BITS "\n\tglobal_load_dwordx2 %3, %11, off " BITS "\n\t" \
        "global_load_dwordx2 %4, %12, off " BITS "\n\tglobal_load_dwordx2 %5, %13, off " BITS "\n\tglobal_load_dwordx2 %6, %14, off " BITS "\n\tglobal_load_dwordx2 %7, %15, off " BITS "\n\ts_waitcnt vmcnt(0)" \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory")
        if (MODE == 1) LD8("nt");
        if (MODE == 2) LD8("sc1");
        if (MODE == 3) LD8("sc0 sc1");
        if (MODE == 4) LD8("sc0 sc1 nt");
        if (MODE == 5) LD8("sc0");
    }
    double acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void fill_idx(uint32_t* idx, size_t n, uint32_t xlen, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)(splitmix64(seed ^ i) % xlen);
}
template <int MODE> int run(const char* tag, const uint32_t* idx, const double* x, double* out, size_t n) {
    unsigned grid = (unsigned)((n + 2047) / 2048);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, idx, x, out, n); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%s,%.3f ms,%.1f Ggathers/s\n", tag, ms, n / ms * 1e-6);
    return 0;
}
int main() {
    const size_t n = 200u * 1000 * 1000;
    for (uint32_t xlen : {10u * 1000 * 1000, 80u * 1000 * 1000}) {
        uint32_t* idx; double *x, *out;
        CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, (size_t)xlen * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(x, 0, (size_t)xlen * 8));
        hipLaunchKernelGGL(fill_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, xlen, 99); CK(hipDeviceSynchronize());
        printf("x = %u doubles (%.0f MB)\n", xlen, xlen * 8e-6);
        run<0>("plain", idx, x, out, n); run<1>("nt", idx, x, out, n); run<2>("sc1", idx, x, out, n);
        run<3>("sc0 sc1", idx, x, out, n); run<4>("sc0 sc1 nt", idx, x, out, n); run<5>("sc0", idx, x, out, n);
        CK(hipFree(idx)); CK(hipFree(x)); CK(hipFree(out));
    }
    return 0;
}
