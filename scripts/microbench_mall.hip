// microbench_mall.hip -- does the 256 MiB Infinity Cache help a produce -> consume stream?
// For buffer sizes S: (a) read S repeatedly, (b) write S repeatedly, (c) write S then read S (the two-phase SpMV's product
// stream in a chunked schedule), each with and without a 10 B/nnz-like HBM stream of never-reused data alongside.
// Output: size_MB, pattern, TB/s of the S-buffer traffic (and of the side stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(1024) void rd(const double* __restrict__ p, size_t n, double* sink) {
    dbl2 acc = 0;
    const size_t stride = (size_t)gridDim.x * 1024 * 2;
    for (size_t i = ((size_t)blockIdx.x * 1024 + threadIdx.x) * 2; i < n; i += stride * 4) {
        dbl2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { size_t q = i + u * stride; q = q < n ? q : i; v[u] = NT ? __builtin_nontemporal_load((const dbl2*)(p + q)) : *(const dbl2*)(p + q); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc.x + acc.y == 1.234e300) *sink = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(1024) void wr(double* __restrict__ p, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * 1024 * 2;
    for (size_t i = ((size_t)blockIdx.x * 1024 + threadIdx.x) * 2; i < n; i += stride) {
        dbl2 o; o.x = v; o.y = v + 1;
        if (NT) __builtin_nontemporal_store(o, (dbl2*)(p + i)); else *(dbl2*)(p + i) = o;
    }
}
// produce: read `big` (10 B/entry-like: 8 B here) streaming, write `small`;  consume: read `small` (+ 2 B/entry of big: skipped)
template <bool NT>
__global__ __launch_bounds__(1024) void produce(const double* __restrict__ big, double* __restrict__ small, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024 * 2;
    for (size_t i = ((size_t)blockIdx.x * 1024 + threadIdx.x) * 2; i < n; i += stride) {
        dbl2 v = __builtin_nontemporal_load((const dbl2*)(big + i));
        v *= 1.5;
        if (NT) __builtin_nontemporal_store(v, (dbl2*)(small + i)); else *(dbl2*)(small + i) = v;
    }
}
static float timeit(hipEvent_t a, hipEvent_t b) { float ms; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    const size_t bigN = 3ull << 28;                    // 6 GiB of doubles: never cache resident
    double *big, *small, *sink;
    CK(hipMalloc(&big, bigN * 8)); CK(hipMalloc(&small, (size_t)2 << 30)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(big, 0, bigN * 8)); CK(hipMemset(small, 0, (size_t)2 << 30));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 2048, reps = 12;
    printf("size_MB,pattern,policy,ms_per_pass,TBps\n");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n = mb * (1 << 20) / 8;
        for (int nt = 0; nt < 2; ++nt) {
            // (a) repeated read
            if (nt) rd<true><<<grid, 1024>>>(small, n, sink); else rd<false><<<grid, 1024>>>(small, n, sink);
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) { if (nt) rd<true><<<grid, 1024>>>(small, n, sink); else rd<false><<<grid, 1024>>>(small, n, sink); }
            CK(hipEventRecord(b));
            float ms = timeit(a, b) / reps;
            printf("%zu,read,%s,%.4f,%.2f\n", mb, nt ? "nt" : "plain", ms, n * 8.0 / ms * 1e-9);
            // (b) repeated write
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) { if (nt) wr<true><<<grid, 1024>>>(small, n, 1.0); else wr<false><<<grid, 1024>>>(small, n, 1.0); }
            CK(hipEventRecord(b));
            ms = timeit(a, b) / reps;
            printf("%zu,write,%s,%.4f,%.2f\n", mb, nt ? "nt" : "plain", ms, n * 8.0 / ms * 1e-9);
            // (c) write then read, alternating
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) {
                if (nt) wr<true><<<grid, 1024>>>(small, n, 1.0); else wr<false><<<grid, 1024>>>(small, n, 1.0);
                if (nt) rd<true><<<grid, 1024>>>(small, n, sink); else rd<false><<<grid, 1024>>>(small, n, sink);
            }
            CK(hipEventRecord(b));
            ms = timeit(a, b) / reps;
            printf("%zu,write+read,%s,%.4f,%.2f\n", mb, nt ? "nt" : "plain", ms, 2 * n * 8.0 / ms * 1e-9);
            // (d) produce (HBM stream in, small out) then consume (small in): the chunked two-phase schedule
            size_t off = 0;
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) {
                if (off + n > bigN) off = 0;
                if (nt) produce<true><<<grid, 1024>>>(big + off, small, n); else produce<false><<<grid, 1024>>>(big + off, small, n);
                if (nt) rd<true><<<grid, 1024>>>(small, n, sink); else rd<false><<<grid, 1024>>>(small, n, sink);
                off += n;
            }
            CK(hipEventRecord(b));
            ms = timeit(a, b) / reps;
            printf("%zu,produce+consume(24B/entry),%s,%.4f,%.2f\n", mb, nt ? "nt" : "plain", ms, 3 * n * 8.0 / ms * 1e-9);
        }
    }
    return 0;
}
