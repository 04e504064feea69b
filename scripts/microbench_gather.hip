// microbench_gather.hip -- how fast can one MI355X gather 8-byte elements?
// Measurement aid for DESIGN.md ("why the x gather, not the AS/JA stream, bounds SpMV").
// Each lane does ITER gathers x[idx] with idx drawn inside a window; patterns:
//   window size (L1 / L2 / MALL / HBM resident), lanes per shared 128-B line,
//   element width 8 or 4 bytes, and the same gather served from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// idx[i] precomputed (streamed, coalesced) -> gather
template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ idx, const T* __restrict__ x,
                                                     T* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 8 + threadIdx.x;
    T acc = 0;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = (i + u * 256 < n) ? __builtin_nontemporal_load(idx + i + u * 256) : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += x[c[u]];
    if (acc == (T)12345.678) out[0] = acc;   // keep
}

// same but x window staged in LDS first (window <= 16K doubles)
__global__ __launch_bounds__(256) void gather_lds_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ x,
                                                         double* __restrict__ out, size_t n, uint32_t win, int reps) {
    extern __shared__ double xs[];
    for (uint32_t k = threadIdx.x; k < win; k += 256) xs[k] = x[k];
    __syncthreads();
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        size_t i = ((size_t)blockIdx.x * reps + r) * 256 * 8 + threadIdx.x;
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = (i + u * 256 < n) ? __builtin_nontemporal_load(idx + i + u * 256) : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += xs[c[u]];
    }
    if (acc == 12345.678) out[0] = acc;
}

// pure stream of 12 B/elem (AS+JA) for reference
__global__ __launch_bounds__(256) void stream_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ v,
                                                     double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x * 8 + threadIdx.x;
    double acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i + u * 256 < n) acc += __builtin_nontemporal_load(v + i + u * 256) * (double)__builtin_nontemporal_load(idx + i + u * 256);
    if (acc == 12345.678) out[0] = acc;
}

__global__ void fill_idx(uint32_t* idx, size_t n, uint32_t window, uint32_t share, uint64_t seed, uint32_t xlen) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // groups of `share` consecutive elements fall into the same 128-B line (16 doubles)
    size_t g = i / share;
    // window slides with position so that the whole x is eventually touched (like a band)
    uint64_t centre = (uint64_t)((double)i / (double)n * (double)(xlen - window));
    uint32_t line = (uint32_t)(splitmix64(seed ^ g) % (window / 16));
    uint32_t within = (uint32_t)(splitmix64(seed + 77 + i) % 16);
    idx[i] = (uint32_t)(centre / 16 * 16 + (uint64_t)line * 16 + within);
}

template <typename F>
float timeit(F f, int iters = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const size_t n = 200u * 1000 * 1000;          // gathers per launch
    const uint32_t xlen = 10u * 1000 * 1000;
    uint32_t* idx; double* x; float* xf; double* out; double* v;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, (size_t)xlen * 8 + 4096)); CK(hipMalloc(&xf, (size_t)xlen * 4 + 4096));
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&v, n * 8));
    CK(hipMemset(x, 0, (size_t)xlen * 8)); CK(hipMemset(xf, 0, (size_t)xlen * 4)); CK(hipMemset(v, 0, n * 8));
    const unsigned grid = (unsigned)((n + 2047) / 2048);
    printf("pattern,window_elems,share,ms,Ggathers_per_s,lanes_per_clk_per_CU@2.4GHz\n");
    {
        float ms = timeit([&] { hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, idx, v, out, n); });
        printf("stream12B,0,0,%.3f,%.1f,(%.2f TB/s)\n", ms, n / ms * 1e-6, n * 12.0 / ms * 1e-9);
    }
    const uint32_t windows[] = {2048, 16384, 32768, 262144, 1u << 20, xlen / 16 * 16};
    const uint32_t shares[] = {1, 2, 4, 16};
    for (uint32_t w : windows)
        for (uint32_t s : shares) {
            hipLaunchKernelGGL(fill_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, w, s, 1234, xlen);
            CK(hipDeviceSynchronize());
            float ms = timeit([&] { hipLaunchKernelGGL((gather_kernel<double>), dim3(grid), dim3(256), 0, 0, idx, x, out, n); });
            printf("tcp_f64,%u,%u,%.3f,%.1f,%.3f\n", w, s, ms, n / ms * 1e-6, n / (ms * 1e-3) / 256 / 2.4e9);
            if (s == 1) {
                ms = timeit([&] { hipLaunchKernelGGL((gather_kernel<float>), dim3(grid), dim3(256), 0, 0, idx, xf, (float*)out, n); });
                printf("tcp_f32,%u,%u,%.3f,%.1f,%.3f\n", w, s, ms, n / ms * 1e-6, n / (ms * 1e-3) / 256 / 2.4e9);
            }
        }
    // LDS-served gather: window 16384 doubles (128 KiB), indices local to the window
    {
        const uint32_t w = 16384;
        hipLaunchKernelGGL(fill_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, w, 1, 99, w);
        CK(hipDeviceSynchronize());
        for (int reps : {1, 8, 64}) {
            unsigned g2 = (unsigned)((n + 2048ull * reps - 1) / (2048ull * reps));
            CK(hipFuncSetAttribute((const void*)gather_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, w * 8));
            float ms = timeit([&] { hipLaunchKernelGGL(gather_lds_kernel, dim3(g2), dim3(256), w * 8, 0, idx, x, out, n, w, reps); });
            printf("lds_f64_reps%d,%u,1,%.3f,%.1f,%.3f\n", reps, w, ms, n / ms * 1e-6, n / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
