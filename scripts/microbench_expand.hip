// microbench_expand.hip -- variants of phase 1 of the two-phase SpMV (pb_expand) on c5-like data:
// work items of ~109 K entries, each with its own 16 Ki-column x slice staged in LDS, random 16-bit local
// columns.  Variants differ in the lane -> entry map (4 consecutive entries per lane with 16-B accesses, or
// consecutive lanes on consecutive entries), in how the two register batches are rotated (copy / ping-pong)
// and in whether loads sit under a bounds branch or use clamped addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double   dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush4 __attribute__((ext_vector_type(4)));
typedef uint16_t ush2 __attribute__((ext_vector_type(2)));
constexpr uint32_t C = 16384, T = 1024;

__device__ __forceinline__ void stage(double* xs, const double* __restrict__ x, uint64_t col0) {
    double xv[C / T];
#pragma unroll
    for (uint32_t i = 0; i < C / T; ++i) xv[i] = x[col0 + threadIdx.x + i * T];
#pragma unroll
    for (uint32_t i = 0; i < C / T; ++i) xs[threadIdx.x + i * T] = xv[i];
}

// ---- A: 4 consecutive entries per lane, copy rotation, branches (the shipped kernel) -------------------------
template <int D> struct R4 { dbl2 a[D], b[D]; ush4 c[D]; };
template <int D, bool CLAMP>
__device__ __forceinline__ void load4(R4<D>& r, uint32_t p, uint32_t e, const double* __restrict__ val, const uint16_t* __restrict__ col) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        uint32_t q = p + u * 4 * T;
        if (CLAMP) {
            q = min(q, e - 4);
            r.a[u] = __builtin_nontemporal_load((const dbl2*)(val + q)); r.b[u] = __builtin_nontemporal_load((const dbl2*)(val + q + 2));
            r.c[u] = __builtin_nontemporal_load((const ush4*)(col + q));
        } else if (q < e) {
            r.a[u] = __builtin_nontemporal_load((const dbl2*)(val + q)); r.b[u] = __builtin_nontemporal_load((const dbl2*)(val + q + 2));
            r.c[u] = __builtin_nontemporal_load((const ush4*)(col + q));
        } else { r.a[u] = 0; r.b[u] = 0; r.c[u] = 0; }
    }
}
template <int D>
__device__ __forceinline__ void store4(const R4<D>& r, uint32_t p, uint32_t e, const double* xs, double* __restrict__ out) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const uint32_t q = p + u * 4 * T;
        if (q < e) {
            dbl2 r0, r1;
            r0.x = r.a[u].x * xs[r.c[u].x]; r0.y = r.a[u].y * xs[r.c[u].y];
            r1.x = r.b[u].x * xs[r.c[u].z]; r1.y = r.b[u].y * xs[r.c[u].w];
            __builtin_nontemporal_store(r0, (dbl2*)(out + q)); __builtin_nontemporal_store(r1, (dbl2*)(out + q + 2));
        }
    }
}
template <int D, int ROT, bool CLAMP>     // ROT 0: cur = nxt copy, 1: ping-pong
__global__ __launch_bounds__(T) void k4(const uint3* __restrict__ work, const double* __restrict__ val, const uint16_t* __restrict__ col,
                                        const double* __restrict__ x, double* __restrict__ out) {
    extern __shared__ double xs[];
    const uint3 w = work[blockIdx.x];
    const uint32_t e = w.z;
    uint32_t p = w.y + 4 * threadIdx.x;
    R4<D> cur, nxt;
    load4<D, CLAMP>(cur, p, e, val, col);
    stage(xs, x, (uint64_t)w.x * C);
    __syncthreads();
    constexpr uint32_t B = D * 4 * T;
    if (ROT == 0) {
        for (; p < e; p += B) { load4<D, CLAMP>(nxt, p + B, e, val, col); store4<D>(cur, p, e, xs, out); cur = nxt; }
    } else {
        for (; p < e; p += 2 * B) {
            load4<D, CLAMP>(nxt, p + B, e, val, col); store4<D>(cur, p, e, xs, out);
            load4<D, CLAMP>(cur, p + 2 * B, e, val, col); store4<D>(nxt, p + B, e, xs, out);
        }
    }
}

// ---- B: V consecutive entries per lane with dense wavefront accesses (V = 1: 8-B, V = 2: 16-B) ------------
template <int D, int V> struct RD { double a[D][V]; uint16_t c[D][V]; };
template <int D, int V, bool CLAMP>
__device__ __forceinline__ void loadd(RD<D, V>& r, uint32_t p, uint32_t e, const double* __restrict__ val, const uint16_t* __restrict__ col) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        uint32_t q = p + u * V * T;
        const bool in = q < e;
        if (CLAMP) q = min(q, e - V);
        if (CLAMP || in) {
            if (V == 1) { r.a[u][0] = __builtin_nontemporal_load(val + q); r.c[u][0] = __builtin_nontemporal_load(col + q); }
            else { const dbl2 a = __builtin_nontemporal_load((const dbl2*)(val + q)); const ush2 c = __builtin_nontemporal_load((const ush2*)(col + q));
                   r.a[u][0] = a.x; r.a[u][V - 1] = a.y; r.c[u][0] = c.x; r.c[u][V - 1] = c.y; }
        } else { for (int i = 0; i < V; ++i) { r.a[u][i] = 0; r.c[u][i] = 0; } }
    }
}
template <int D, int V>
__device__ __forceinline__ void stored(const RD<D, V>& r, uint32_t p, uint32_t e, const double* xs, double* __restrict__ out) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const uint32_t q = p + u * V * T;
        if (q < e) {
            if (V == 1) __builtin_nontemporal_store(r.a[u][0] * xs[r.c[u][0]], out + q);
            else { dbl2 o; o.x = r.a[u][0] * xs[r.c[u][0]]; o.y = r.a[u][V - 1] * xs[r.c[u][V - 1]]; __builtin_nontemporal_store(o, (dbl2*)(out + q)); }
        }
    }
}
template <int D, int V, int ROT, bool CLAMP>
__global__ __launch_bounds__(T) void kd(const uint3* __restrict__ work, const double* __restrict__ val, const uint16_t* __restrict__ col,
                                        const double* __restrict__ x, double* __restrict__ out) {
    extern __shared__ double xs[];
    const uint3 w = work[blockIdx.x];
    const uint32_t e = w.z;
    uint32_t p = w.y + V * threadIdx.x;
    RD<D, V> cur, nxt;
    loadd<D, V, CLAMP>(cur, p, e, val, col);
    stage(xs, x, (uint64_t)w.x * C);
    __syncthreads();
    constexpr uint32_t B = D * V * T;
    if (ROT == 0) {
        for (; p < e; p += B) { loadd<D, V, CLAMP>(nxt, p + B, e, val, col); stored<D, V>(cur, p, e, xs, out); cur = nxt; }
    } else {
        for (; p < e; p += 2 * B) {
            loadd<D, V, CLAMP>(nxt, p + B, e, val, col); stored<D, V>(cur, p, e, xs, out);
            loadd<D, V, CLAMP>(cur, p + 2 * B, e, val, col); stored<D, V>(nxt, p + B, e, xs, out);
        }
    }
}

__global__ void fill(uint16_t* col, double* val, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 12345; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    col[i] = (uint16_t)(z & (C - 1)); val[i] = 1.0 + (double)(z >> 40) * 1e-9;
}

template <typename K>
int run(const char* tag, K kern, const uint3* work, uint32_t nWork, const double* val, const uint16_t* col, const double* x, double* out, uint64_t n) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(nWork), dim3(T), C * 8, 0, work, val, col, x, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nWork), dim3(T), C * 8, 0, work, val, col, x, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    double chk = 0; CK(hipMemcpy(&chk, out + n / 2, 8, hipMemcpyDeviceToHost));
    printf("%s,ms=%.3f,ns_per_kentry=%.3f,check=%.6f\n", tag, ms, ms * 1e6 / (n / 1000.0), chk);
    return 0;
}

int main() {
    const uint32_t S = 4096, per = 3, piece = 109216;           // 3 work items per slice, multiples of 16
    const uint64_t n = (uint64_t)S * per * piece;                 // 1.34 G entries
    double *val, *out, *x; uint16_t* col; uint3* work;
    CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&col, n * 2)); CK(hipMalloc(&x, (uint64_t)S * C * 8));
    CK(hipMemset(x, 0, (uint64_t)S * C * 8));
    {   std::vector<double> hx(C); for (uint32_t i = 0; i < C; ++i) hx[i] = 1.0 + i;     // slice 0 non-trivial; others zero
        CK(hipMemcpy(x, hx.data(), C * 8, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, col, val, n);
    std::vector<uint3> hw;
    for (uint32_t s = 0; s < S; ++s) for (uint32_t k = 0; k < per; ++k) { const uint32_t b = (s * per + k) * piece; hw.push_back(make_uint3(s, b, b + piece)); }
    CK(hipMalloc(&work, hw.size() * sizeof(uint3))); CK(hipMemcpy(work, hw.data(), hw.size() * sizeof(uint3), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    // the same work list shifted by 4 entries (32 B): wavefront accesses no longer start on a 128-B line
    uint3* workS; { std::vector<uint3> h2 = hw; for (auto& w : h2) { w.y += 4; w.z += 4; } h2.back().z -= 4;
        CK(hipMalloc(&workS, h2.size() * sizeof(uint3))); CK(hipMemcpy(workS, h2.data(), h2.size() * sizeof(uint3), hipMemcpyHostToDevice)); }
    const uint32_t nW = (uint32_t)hw.size();
#define RUN(tag, ...) if (run(tag, __VA_ARGS__, work, nW, val, col, x, out, n)) return 1
    RUN("vec4_D4_copy_branch", (k4<4, 0, false>));
    RUN("vec4_D4_pingpong_branch", (k4<4, 1, false>));
    RUN("vec4_D4_pingpong_clamp", (k4<4, 1, true>));
    RUN("vec4_D2_pingpong_clamp", (k4<2, 1, true>));
    RUN("dense1_D8_copy_branch", (kd<8, 1, 0, false>));
    RUN("dense1_D8_pingpong_clamp", (kd<8, 1, 1, true>));
    RUN("dense1_D16_pingpong_clamp", (kd<16, 1, 1, true>));
    RUN("dense2_D4_pingpong_clamp", (kd<4, 2, 1, true>));
    RUN("dense2_D8_pingpong_clamp", (kd<8, 2, 1, true>));
    RUN("vec4_D4_copy_branch", (k4<4, 0, false>));
#define RUNS(tag, ...) if (run(tag, __VA_ARGS__, workS, nW, val, col, x, out, n)) return 1
    RUNS("shift4:vec4_D4_copy_branch", (k4<4, 0, false>));
    RUNS("shift4:dense1_D8_copy_branch", (kd<8, 1, 0, false>));
    RUNS("shift4:dense2_D8_pingpong_clamp", (kd<8, 2, 1, true>));
    return 0;
}
