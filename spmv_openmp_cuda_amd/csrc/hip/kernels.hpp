// kernels.hpp -- hand-written gfx950 SpMV kernels (fp64 values, 32-bit column ids).
//
// What each kernel restates (reference paths relative to its root):
//   csr_scalar_kernel      cudaSpMVRowsCSR                   src/SpMV_CUDA.cu:33-49
//   csr_vector_kernel      cudaSpMVWarpPerRowCSR             src/SpMV_CUDA.cu:52-73   (as intended: every row)
//   csr_stream2_kernel     both of the above, MI355X-first: coalesced span load -> LDS -> per-row reduce
//   ell_colmajor_thread    cudaSpMVRowsELL                   src/SpMV_CUDA.cu:79-96
//   ell_rowmajor_thread    cudaSpMVRowsELLNNTransposed       src/SpMV_CUDA.cu:99-115
//   ell_rowmajor_group     cudaSpMVWarpsPerRowELLNTrasposed  src/SpMV_CUDA.cu:116-135
//   wave_sum               reduceWarpRegs                    src/include/cudaUtils.h:101-106 (32 lanes there, 64 here)
//
// All kernels take POD arguments by value (the reference passes a pointer to a
// device-resident struct and re-reads m->M, m->IRP... from global memory in
// every iteration, SpMV_CUDA.cu:35,44-45).
//
// The library is compiled with -ffp-contract=off: products are rounded before
// they are added, so every kernel that adds a row's products in ascending-j
// order is bit-identical to the serial oracle (sgemvSerial, gcc -O2 x86-64).
#pragma once
#include "device_mat.hpp"

namespace spmvhip {

// The AQL dispatch packet stores the grid size in WORK-ITEMS as 32 bits per
// dimension: blocks * threads must stay below 2^32 or the launch silently wraps.
// Launchers therefore fold large grids into (x, y) -- see grid2d() in
// device_mat.hpp -- and kernels recover the linear workgroup id here.
__device__ __forceinline__ uint64_t linear_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }

// streamed-once data: keep it out of the way of x in L2 / Infinity Cache
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
    return __builtin_nontemporal_load(p);          // (plain loads: equal or slower everywhere, c2 one-pass 0.345 vs 0.320 ms)
}

// value j of a matrix: from the stream, or -- UNIT: every stored value is the same double -- from a register
template <bool UNIT>
__device__ __forceinline__ double value_at(const double* __restrict__ AS, uint64_t j, double unitValue) {
    return UNIT ? unitValue : stream_load(AS + j);
}

// sum over the 64 lanes of a wavefront; result valid in lane 0
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}
// sum over aligned groups of G lanes (G power of two <= 64); valid in the group's lane 0
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_down(v, off, G);
    return v;
}
__device__ __forceinline__ double group_sum_rt(double v, int G) {
    for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_down(v, off, G);
    return v;
}

// ----------------------------------------------------------------------------------- CSR
// One thread walks one row directly in global memory.  Kept as the plain
// restatement of the reference kernel (variant 0) and as the A/B baseline for
// the LDS-stream kernel: lanes stride by the row length, so AS/JA reads are
// uncoalesced.
template <typename I>
__global__ __launch_bounds__(1024) void csr_scalar_kernel(
    uint32_t M, const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const double* __restrict__ x, double* __restrict__ y) {
    const uint64_t gid = linear_block() * blockDim.x + threadIdx.x;
    if (gid >= M) return;
    const uint32_t row = (uint32_t)gid;
    const I b = IRP[row], e = IRP[row + 1];
    double acc = 0;
    for (I j = b; j < e; ++j) acc += AS[j] * x[JA[j]];
    y[row] = acc;
}

// One wavefront per row, lanes stride the row, shuffle tree, lane 0 stores.
// rowsPerWg = blockDim.x / 64 wavefronts share a workgroup.
template <typename I>
__global__ __launch_bounds__(1024) void csr_vector_kernel(
    uint32_t M, const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const double* __restrict__ x, double* __restrict__ y) {
    const uint32_t wavesPerWg = blockDim.x / WAVE;
    const uint64_t grow = linear_block() * wavesPerWg + threadIdx.x / WAVE;   // wave-uniform
    const uint32_t lane = threadIdx.x % WAVE;
    if (grow >= M) return;
    const uint32_t row = (uint32_t)grow;
    const I b = IRP[row], e = IRP[row + 1];
    double acc = 0;
    for (I j = b + lane; j < e; j += WAVE) acc += stream_load(AS + j) * x[stream_load(JA + j)];
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc;
}

constexpr int STREAM_UNROLL = STREAM_NNZ / WG_THREADS;

// LDS-stream kernel (default CSR kernel).  Workgroup b owns the consecutive rows of block b, whose nnz span is
// contiguous in AS/JA:
//   1. every lane loads nnz tid, tid+256, ... of the span (fully coalesced, non-temporal), gathers x and parks the
//      rounded product in LDS;
//   2. the rows' segments of the LDS array are summed
//        SEQ: one thread per row, ascending j  -> bit-identical to the oracle
//        VEC: L lanes per row (L = largest power of two with L*rows <= 256), lane-strided partial sums + shuffle
//             tree (LDS segmented reduction)
// A block that is a single row longer than STREAM_NNZ takes the workgroup-per-row path instead (lane-strided walk,
// wave shuffle, LDS combine).  This is the second generation of the kernel; the first (dependent IRP loads in front
// of the stream) and a third (two row blocks per workgroup, 0-4 % slower) were measured against it and removed
// (profiles/r01_variants.md).  What the second generation shortened in the latency chain:
//   * the block table holds {first row, #rows, #nnz} and the block's nnz offset, read with scalar
//     loads -> no dependent IRP[r0] / IRP[r1] round trips before the stream can start;
//   * the block's row pointers are loaded together with the AS/JA span and parked in LDS as
//     16-bit local offsets -> the reduction phase never waits on global memory;
//   * blocks are ordered long rows first (their serial tail overlaps the rest), then by row;
//     the row-ordered part is dealt to the 8 XCDs in contiguous ranges (workgroups b, b+8, ...
//     share an XCD) so that neighbouring rows -- and their x window -- meet in one L2.
// (An LDS x tile -- staging a block's 2 Ki-column window of x in LDS and gathering from LDS -- was
//  built and measured: on a +-512 band it is SLOWER, 0.80 vs 0.56 ms, because the window is reloaded
//  per 2048-nnz block, costs a second barrier and halves occupancy, while the L1-resident gather it
//  replaces already runs at ~1 lane/clk/CU.  Not kept)
constexpr uint32_t STREAM2_MAX_ROWS = 2 * WG_THREADS;     // two row pointers per lane

template <typename I, bool SEQ, bool UNIT>
__global__ __launch_bounds__(WG_THREADS) void csr_stream2_kernel(
    uint32_t nBlk, uint32_t nLong, const uint4* __restrict__ blkInfo, const uint64_t* __restrict__ blkBase,
    const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, double unitValue, const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double   prod[STREAM_NNZ];
    __shared__ uint16_t rowOff[STREAM2_MAX_ROWS + 1];
    __shared__ double   wpart[WG_THREADS / WAVE];

    const uint32_t tid = threadIdx.x;
    uint64_t blk = linear_block();
    if (blk >= nBlk) return;
    if (blk >= nLong) {
        // XCD-contiguous deal of the row-ordered blocks
        // (bijection of [0,n): stripe x = q % 8 gets the a or a+1 consecutive blocks starting at x*a + min(x, rem))
        const uint64_t q = blk - nLong, n = nBlk - nLong;
        const uint64_t a = n / 8, rem = n % 8, xcd = q % 8;
        blk = nLong + xcd * a + (xcd < rem ? xcd : rem) + q / 8;
    }
    const uint4 info = blkInfo[blk];
    const uint32_t r0 = info.x, R = info.y, n = info.z;
    const uint64_t base = blkBase[blk];

    if (info.w) {
        // ---- long row (single row of n > 2048 entries)
        const uint64_t end = base + n;
        if (SEQ) {
            double acc = 0;
            for (uint64_t c = base; c < end; c += STREAM_NNZ) {
                const uint32_t cn = (uint32_t)(end - c < (uint64_t)STREAM_NNZ ? end - c : (uint64_t)STREAM_NNZ);
#pragma unroll
                for (int u = 0; u < STREAM_UNROLL; ++u) {
                    const uint32_t k = tid + u * WG_THREADS;
                    if (k < cn) prod[k] = value_at<UNIT>(AS, c + k, unitValue) * x[stream_load(JA + c + k)];
                }
                __syncthreads();
                if (tid == 0)
                    for (uint32_t j = 0; j < cn; ++j) acc += prod[j];
                __syncthreads();
            }
            if (tid == 0) y[r0] = acc;
            return;
        }
        double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        uint64_t j = base + tid;
        for (; j + 3 * WG_THREADS < end; j += 4 * WG_THREADS) {
            const uint32_t c0 = stream_load(JA + j), c1 = stream_load(JA + j + WG_THREADS),
                           c2 = stream_load(JA + j + 2 * WG_THREADS), c3 = stream_load(JA + j + 3 * WG_THREADS);
            const double a0 = value_at<UNIT>(AS, j, unitValue), a1 = value_at<UNIT>(AS, j + WG_THREADS, unitValue),
                         a2 = value_at<UNIT>(AS, j + 2 * WG_THREADS, unitValue), a3 = value_at<UNIT>(AS, j + 3 * WG_THREADS, unitValue);
            acc0 += a0 * x[c0]; acc1 += a1 * x[c1]; acc2 += a2 * x[c2]; acc3 += a3 * x[c3];
        }
        for (; j < end; j += WG_THREADS) acc0 += value_at<UNIT>(AS, j, unitValue) * x[stream_load(JA + j)];
        double acc = wave_sum((acc0 + acc1) + (acc2 + acc3));
        if (tid % WAVE == 0) wpart[tid / WAVE] = acc;
        __syncthreads();
        if (tid == 0) {
            double s = wpart[0];
#pragma unroll
            for (int w = 1; w < WG_THREADS / WAVE; ++w) s += wpart[w];
            y[r0] = s;
        }
        return;
    }

    // ---- 1. span + row pointers in flight together
    {
        uint32_t col[STREAM_UNROLL];
        double   val[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) {
            const uint32_t k = tid + u * WG_THREADS;
            const bool in = k < n;
            col[u] = in ? stream_load(JA + base + k) : 0u;
            val[u] = in ? value_at<UNIT>(AS, base + k, unitValue) : 0.0;
        }
        // rows r0 .. r0+R (R <= 512): two row pointers per lane, the closing one is base+n
        uint32_t rp0 = 0, rp1 = 0;
        if (tid < R) rp0 = (uint32_t)((uint64_t)IRP[r0 + tid] - base);
        if (tid + WG_THREADS < R) rp1 = (uint32_t)((uint64_t)IRP[r0 + tid + WG_THREADS] - base);
        double xv[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) xv[u] = x[col[u]];
        if (tid < R) rowOff[tid] = (uint16_t)rp0;
        if (tid + WG_THREADS < R) rowOff[tid + WG_THREADS] = (uint16_t)rp1;
        if (tid == 0) rowOff[R] = (uint16_t)n;
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) {
            const uint32_t k = tid + u * WG_THREADS;
            if (k < n) prod[k] = val[u] * xv[u];
        }
    }
    __syncthreads();

    // ---- 2. per-row reduction, everything in LDS
    if (SEQ) {
        for (uint32_t rr = tid; rr < R; rr += WG_THREADS) {
            const uint32_t s = rowOff[rr], e = rowOff[rr + 1];
            double acc = 0;
            for (uint32_t j = s; j < e; ++j) acc += prod[j];
            y[r0 + rr] = acc;
        }
    } else {
        int L = 1;
        while (L < WAVE && 2u * L * R <= (uint32_t)WG_THREADS) L <<= 1;
        const uint32_t rowsPerPass = WG_THREADS / L;
        const uint32_t g = tid / L, l = tid % L;
        const uint32_t passes = (R + rowsPerPass - 1) / rowsPerPass;
        for (uint32_t p = 0; p < passes; ++p) {
            const uint32_t rr = p * rowsPerPass + g;
            const bool live = rr < R;
            uint32_t s = 0, e = 0;
            if (live) { s = rowOff[rr]; e = rowOff[rr + 1]; }
            double acc = 0;
            for (uint32_t j = s + l; j < e; j += L) acc += prod[j];
            acc = group_sum_rt(acc, L);
            if (live && l == 0) y[r0 + rr] = acc;
        }
    }
}

// ----------------------------------------------------------------------------------- ELL
// Column-major ("transposed") + pitched, one thread per row: lane i of a wave
// reads element i of a 512-byte line in every slot.  USE_RL = stop at the row's
// own length; otherwise walk all slots incl. the {AS=0,JA=0} padding exactly
// like the reference kernel does.
// UNIT (with USE_RL only: padding cells are never touched then): every real cell holds `unitValue`, AS is not read.
template <bool USE_RL, bool UNIT = false>
__global__ __launch_bounds__(1024) void ell_colmajor_thread(
    uint32_t rows, uint32_t slots, size_t pitch, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const uint32_t* __restrict__ RL,
    const double* __restrict__ x, double* __restrict__ y, double unitValue = 0.0) {
    static_assert(USE_RL || !UNIT, "without row lengths the padding cells {0.0, column 0} are part of the sum");
    const uint64_t gid = linear_block() * blockDim.x + threadIdx.x;
    if (gid >= rows) return;
    const uint32_t row = (uint32_t)gid;
    const uint32_t n = USE_RL ? RL[row] : slots;
    double acc = 0;
    size_t idx = row;
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4, idx += 4 * pitch) {
        const uint32_t c0 = stream_load(JA + idx), c1 = stream_load(JA + idx + pitch),
                       c2 = stream_load(JA + idx + 2 * pitch), c3 = stream_load(JA + idx + 3 * pitch);
        const double a0 = value_at<UNIT>(AS, idx, unitValue), a1 = value_at<UNIT>(AS, idx + pitch, unitValue),
                     a2 = value_at<UNIT>(AS, idx + 2 * pitch, unitValue), a3 = value_at<UNIT>(AS, idx + 3 * pitch, unitValue);
        const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        acc += a0 * x0; acc += a1 * x1; acc += a2 * x2; acc += a3 * x3;   // ascending slot order
    }
    for (; i < n; ++i, idx += pitch) acc += value_at<UNIT>(AS, idx, unitValue) * x[stream_load(JA + idx)];
    y[row] = acc;
}

// Row-major + pitched, one thread per row (lane stride = pitch: uncoalesced by
// construction -- the reference's slowest kernel, kept for the A/B).
template <bool USE_RL>
__global__ __launch_bounds__(1024) void ell_rowmajor_thread(
    uint32_t rows, uint32_t slots, size_t pitch, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const uint32_t* __restrict__ RL,
    const double* __restrict__ x, double* __restrict__ y) {
    const uint64_t gid = linear_block() * blockDim.x + threadIdx.x;
    if (gid >= rows) return;
    const uint32_t row = (uint32_t)gid;
    const uint32_t n = USE_RL ? RL[row] : slots;
    const size_t p = (size_t)row * pitch;
    double acc = 0;
    for (uint32_t c = 0; c < n; ++c) acc += AS[p + c] * x[JA[p + c]];
    y[row] = acc;
}

// Row-major + pitched, G lanes per row (G = 64 is the reference's warp-per-row; smaller G packs
// 64/G short rows into one wavefront so lanes are not idle when slots < 64).  Lanes stride the row,
// shuffle tree, group lane 0 stores.  Every group owns ELL_GROUP_ROWS consecutive rows and issues
// the loads of all of them before the first gather: one row per wavefront is latency-bound
// (load -> gather -> 6 shuffles -> store per row), four rows in flight are not.
constexpr int ELL_GROUP_ROWS = 4;

template <bool USE_RL, int G>
__global__ __launch_bounds__(256) void ell_rowmajor_group(
    uint32_t rows, uint32_t slots, size_t pitch, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const uint32_t* __restrict__ RL,
    const double* __restrict__ x, double* __restrict__ y) {
    const uint64_t gid = linear_block() * blockDim.x + threadIdx.x;
    const uint64_t row0 = (gid / G) * ELL_GROUP_ROWS;
    const uint32_t l = (uint32_t)(gid % G);
    double acc[ELL_GROUP_ROWS];
    uint32_t n[ELL_GROUP_ROWS];
#pragma unroll
    for (int r = 0; r < ELL_GROUP_ROWS; ++r) {
        const uint64_t row = row0 + r;
        n[r] = row < rows ? (USE_RL ? RL[row] : slots) : 0u;
        acc[r] = 0;
    }
    if (slots <= (uint32_t)G) {
        // one access per row: all loads first, then all gathers
        uint32_t c[ELL_GROUP_ROWS];
        double a[ELL_GROUP_ROWS];
#pragma unroll
        for (int r = 0; r < ELL_GROUP_ROWS; ++r) {
            const size_t p = (size_t)(row0 + r) * pitch + l;
            const bool in = l < n[r];
            c[r] = in ? stream_load(JA + p) : 0u;
            a[r] = in ? stream_load(AS + p) : 0.0;
        }
#pragma unroll
        for (int r = 0; r < ELL_GROUP_ROWS; ++r) acc[r] = a[r] * x[c[r]];
    } else {
#pragma unroll
        for (int r = 0; r < ELL_GROUP_ROWS; ++r) {
            const size_t p = (size_t)(row0 + r) * pitch;
            for (uint32_t k = l; k < n[r]; k += G) acc[r] += stream_load(AS + p + k) * x[stream_load(JA + p + k)];
        }
    }
#pragma unroll
    for (int r = 0; r < ELL_GROUP_ROWS; ++r) acc[r] = group_sum<G>(acc[r]);
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < ELL_GROUP_ROWS; ++r)
            if (row0 + r < rows) y[row0 + r] = acc[r];
    }
}

// Row-major ELL through the LDS-stream machinery (the default of hipSpMVWarpsPerRowELLNTrasposed while a row fits a block).
// With the pitch equal to the slot count (rounded up to 2) the padded matrix IS one contiguous array: workgroup b owns
// Rb = floor(2048 / pitch) consecutive rows, i.e. the span of Rb * pitch cells starting at row b*Rb:
//   1. lanes load cells tid, tid + 256, ... of the span (fully coalesced, non-temporal), gather x and park the rounded
//      product in LDS -- padding cells {JA = 0, AS = 0.0} included: no test, no division in the streaming phase;
//   2. L lanes per row (largest power of two with L * rows <= 256) sum the row's first RL[row] cells (USE_RL: the row-length
//      early exit -- padding products are never added) or all `slots` cells (as the reference kernel does: padding adds
//      0.0 * x[0]) lane-strided, then a shuffle tree.
// The lanes-per-row kernel above reads a row with one wavefront instruction per 64/G rows and idles the lanes beyond the
// slot count (18 slots: 14 of 32); this one keeps every lane on a dense stream whatever the slot count, like its CSR
// sibling csr_stream2_kernel (stencil 500x100x100, 18 slots: 0.38 -> see profiles/r03_*).
// SEQ: one thread per row adds the row's cells in ascending slot order instead (the summation order of the thread-per-row
// kernels: bit-identical to them and to the serial oracle) -- hipSpMVRowsELLNNTransposed's default: the row-major
// thread-per-row kernel is uncoalesced by construction (a lane's loads are a pitch apart), the same sums fed from a
// coalesced span are not.
template <bool USE_RL, bool SEQ, bool UNIT = false>
__global__ __launch_bounds__(WG_THREADS) void ell_stream_kernel(
    uint32_t rows, uint32_t slots, uint32_t pitch, uint32_t rowsPerBlk, uint64_t nBlk, const uint32_t* __restrict__ JA,
    const double* __restrict__ AS, const uint32_t* __restrict__ RL, const double* __restrict__ x, double* __restrict__ y,
    double unitValue = 0.0) {
    static_assert(USE_RL || !UNIT, "without row lengths the padding cells {0.0, column 0} are part of the sum");
    __shared__ double prod[STREAM_NNZ];
    const uint32_t tid = threadIdx.x;
    uint64_t blk = linear_block();
    if (blk >= nBlk) return;
    {   // XCD-contiguous deal (workgroups b, b + 8, ... share an XCD): neighbouring rows -- and their x window -- meet in one L2
        const uint64_t a = nBlk / 8, rem = nBlk % 8, xcd = blk % 8;
        blk = xcd * a + (xcd < rem ? xcd : rem) + blk / 8;
    }
    const uint64_t r0 = blk * rowsPerBlk;
    const uint32_t R = (uint32_t)min((uint64_t)rowsPerBlk, rows - r0);
    const uint32_t n = R * pitch;                    // <= STREAM_NNZ
    const uint64_t base = r0 * pitch;
    {
        uint32_t col[STREAM_UNROLL];
        double   val[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) {
            const uint32_t k = tid + u * WG_THREADS;
            const bool in = k < n;
            col[u] = in ? stream_load(JA + base + k) : 0u;
            val[u] = in ? value_at<UNIT>(AS, base + k, unitValue) : 0.0;     // (UNIT: padding cells get the value too -- parked, never summed)
        }
        double xv[STREAM_UNROLL];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) xv[u] = x[col[u]];
#pragma unroll
        for (int u = 0; u < STREAM_UNROLL; ++u) {
            const uint32_t k = tid + u * WG_THREADS;
            if (k < n) prod[k] = val[u] * xv[u];
        }
    }
    __syncthreads();
    if (SEQ) {
        for (uint32_t rr = tid; rr < R; rr += WG_THREADS) {
            const uint32_t len = USE_RL ? RL[r0 + rr] : slots;
            const uint32_t s = rr * pitch;
            double acc = 0;
            for (uint32_t j = 0; j < len; ++j) acc += prod[s + j];
            y[r0 + rr] = acc;
        }
        return;
    }
    int L = 1;
    while (L < WAVE && 2u * L * R <= (uint32_t)WG_THREADS) L <<= 1;
    const uint32_t rowsPerPass = WG_THREADS / L;
    const uint32_t g = tid / L, l = tid % L;
    const uint32_t passes = (R + rowsPerPass - 1) / rowsPerPass;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t rr = p * rowsPerPass + g;
        const bool live = rr < R;
        uint32_t len = 0;
        if (live) len = USE_RL ? RL[r0 + rr] : slots;
        const uint32_t s = rr * pitch;
        double acc = 0;
        for (uint32_t j = l; j < len; j += L) acc += prod[s + j];
        acc = group_sum_rt(acc, L);
        if (live && l == 0) y[r0 + rr] = acc;
    }
}

}  // namespace spmvhip
