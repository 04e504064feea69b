// microbench_gather_smem.hip -- 8-byte gathers issued as SCALAR loads (s_load_dwordx2 through the scalar
// cache) instead of vector loads: does the L2 fetch a smaller unit for them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void fill_idx(uint32_t* idx, size_t n, uint32_t xlen, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)(splitmix64(seed ^ i) % xlen);
}
// each wave: 64 indices (one per lane, coalesced load), then 64 scalar loads, 8 in flight at a time
__global__ __launch_bounds__(256) void k_smem(const uint32_t* __restrict__ idx, const double* __restrict__ x, double* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = i < n ? idx[i] : 0;
    double acc = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t cu = __builtin_amdgcn_readlane(c, l + u);
            v[u] = x[cu];                                   // wave-uniform address -> s_load_dwordx2
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_vmem(const uint32_t* __restrict__ idx, const double* __restrict__ x, double* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = i < n ? idx[i] : 0;
    double acc = x[c];
    if (acc == 12345.678) out[0] = acc;
}
template <typename K> int run(const char* tag, K kern, const uint32_t* idx, const double* x, double* out, size_t n) {
    unsigned grid = (unsigned)((n + 255) / 256);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, idx, x, out, n); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%s,%.3f ms,%.1f Ggathers/s\n", tag, ms, n / ms * 1e-6);
    return 0;
}
int main() {
    const size_t n = 200u * 1000 * 1000;
    const uint32_t xlen = 10u * 1000 * 1000;
    uint32_t* idx; double *x, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, (size_t)xlen * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(x, 0, (size_t)xlen * 8));
    hipLaunchKernelGGL(fill_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, xlen, 99); CK(hipDeviceSynchronize());
    run("vector loads (1 per lane)", k_vmem, idx, x, out, n);
    run("scalar loads (64 per wave)", k_smem, idx, x, out, n);
    return 0;
}
