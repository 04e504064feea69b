// microbench_scatter.hip -- what does it cost to TRANSPOSE the tile grid on the write side of phase 1?
// Source order: slice-major tiles (s, b) of L entries each; destination order: bin-major (b, s).
// Reads val (8 B) + col (2 B) as a stream, writes 8 B either in place ("stream") or to the transposed
// tile position ("wscatter"): runs of L x 8 B at stride S*L*8 B.  Lanes take consecutive entries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr uint64_t CAP = 1200ull << 20;            // entries per buffer
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int D, int SCATTER>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ val, const uint16_t* __restrict__ col,
                                          double* __restrict__ out, uint32_t n, uint32_t chunk, uint32_t L,
                                          uint32_t sShift, uint32_t bShift) {
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 1.0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * chunk, e = min(n, b0 + chunk);
    for (uint32_t p = b0 + threadIdx.x; p < e; p += D * 1024) {
        double a[D]; uint16_t j[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const uint32_t q = p + u * 1024;
            if (q < e) {
                uint32_t src = q;
                if (SCATTER == 2) {
                    const uint32_t tile = q / L, within = q - tile * L;
                    const uint32_t s = tile >> bShift, b = tile & ((1u << bShift) - 1);
                    src = ((b << sShift) + s) * L + within;
                }
                a[u] = __builtin_nontemporal_load(val + src); j[u] = __builtin_nontemporal_load(col + q);
            } else { a[u] = 0; j[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const uint32_t q = p + u * 1024;
            if (q < e) {
                uint32_t d = q;
                if (SCATTER == 1) {
                    const uint32_t tile = q / L, within = q - tile * L;
                    const uint32_t s = tile >> bShift, b = tile & ((1u << bShift) - 1);
                    d = ((b << sShift) + s) * L + within;
                }
                __builtin_nontemporal_store(a[u] * (double)j[u], out + d);
            }
        }
    }
}

// the same stream with TWO consecutive entries per lane (16-B value load/store, 4-B column load)
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush2 __attribute__((ext_vector_type(2)));
template <int D>
__global__ __launch_bounds__(1024) void k2(const double* __restrict__ val, const uint16_t* __restrict__ col,
                                           double* __restrict__ out, uint32_t n, uint32_t chunk) {
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 1.0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * chunk, e = min(n, b0 + chunk);
    for (uint32_t p = b0 + 2 * threadIdx.x; p < e; p += D * 2048) {
        dbl2 a[D]; ush2 j[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const uint32_t q = p + u * 2048;
            if (q < e) { a[u] = __builtin_nontemporal_load((const dbl2*)(val + q)); j[u] = __builtin_nontemporal_load((const ush2*)(col + q)); } else { a[u] = 0; j[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const uint32_t q = p + u * 2048;
            if (q < e) { dbl2 r; r.x = a[u].x * (double)j[u].x; r.y = a[u].y * (double)j[u].y; __builtin_nontemporal_store(r, (dbl2*)(out + q)); }
        }
    }
}
template <int D>
int run2(const double* val, const uint16_t* col, double* out, uint32_t n) {
    const uint32_t chunk = 1u << 17;
    if (n > CAP) return 1;
    CK(hipFuncSetAttribute((const void*)k2<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    const unsigned grid = (n + chunk - 1) / chunk;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k2<D>), dim3(grid), dim3(1024), 128 << 10, 0, val, col, out, n, chunk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k2<D>), dim3(grid), dim3(1024), 128 << 10, 0, val, col, out, n, chunk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("stream2,L=0,S=0,B=0,entries=%u,depth=%d,ms=%.3f,ns_per_kentry=%.3f,TBps=%.2f\n", n, D, ms, ms * 1e6 / (n / 1000.0), 18.0 * n / ms * 1e-9);
    return 0;
}

template <int D, int SCATTER>
int run(const char* tag, const double* val, const uint16_t* col, double* out, uint32_t L, uint32_t sShift, uint32_t bShift) {
    const uint32_t n = L << (sShift + bShift), chunk = 1u << 17;
    if ((uint64_t)L << (sShift + bShift) > CAP) { printf("%s,L=%u: %llu entries exceed the buffers, skipped\n", tag, L, (unsigned long long)L << (sShift + bShift)); return 1; }
    CK(hipFuncSetAttribute((const void*)k<D, SCATTER>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    const unsigned grid = (n + chunk - 1) / chunk;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<D, SCATTER>), dim3(grid), dim3(1024), 128 << 10, 0, val, col, out, n, chunk, L, sShift, bShift);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<D, SCATTER>), dim3(grid), dim3(1024), 128 << 10, 0, val, col, out, n, chunk, L, sShift, bShift);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%s,L=%u,S=%u,B=%u,entries=%u,depth=%d,ms=%.3f,ns_per_kentry=%.3f,TBps=%.2f\n", tag, L, 1u << sShift, 1u << bShift, n, D, ms,
           ms * 1e6 / (n / 1000.0), 18.0 * n / ms * 1e-9);
    return 0;
}

int main() {
    const size_t cap = CAP;
    double *val, *out; uint16_t* col;
    CK(hipMalloc(&val, cap * 8)); CK(hipMalloc(&out, cap * 8)); CK(hipMalloc(&col, cap * 2));
    CK(hipMemset(val, 0, cap * 8)); CK(hipMemset(col, 0, cap * 2));
    run<4, 0>("stream", val, col, out, 67, 12, 12);
    run<8, 0>("stream", val, col, out, 67, 12, 12);
    run2<2>(val, col, out, 67u << 24);
    run2<4>(val, col, out, 67u << 24);
    run<4, 1>("wscatter", val, col, out, 67, 12, 12);
    run<4, 1>("wscatter", val, col, out, 68, 12, 12);
    run<4, 1>("wscatter", val, col, out, 72, 12, 12);
    run<4, 1>("wscatter", val, col, out, 80, 12, 11);
    run<8, 1>("wscatter", val, col, out, 67, 12, 12);
    run<4, 1>("wscatter", val, col, out, 64, 12, 12);
    run<4, 1>("wscatter", val, col, out, 16, 13, 13);
    run<4, 1>("wscatter", val, col, out, 33, 12, 13);
    run<4, 1>("wscatter", val, col, out, 105, 12, 11);
    run<4, 1>("wscatter", val, col, out, 268, 11, 11);
    run<4, 1>("wscatter", val, col, out, 1072, 10, 10);
    for (uint32_t L : {67u, 64u, 72u}) run<4, 2>("rscatter", val, col, out, L, 12, 12);
    run<4, 2>("rscatter", val, col, out, 16, 13, 13);
    run<4, 2>("rscatter", val, col, out, 33, 12, 13);
    run<4, 2>("rscatter", val, col, out, 105, 12, 11);
    run<4, 2>("rscatter", val, col, out, 268, 11, 11);
    return 0;
}
