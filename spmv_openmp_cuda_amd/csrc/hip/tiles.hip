// tiles.hip -- column-sliced two-phase SpMV ("propagation blocking") for matrices
// whose x gather misses the caches.
//
// Why: a one-pass row-major SpMV moves a whole 128-B line per 8-byte x element
// that misses L1: 56 G nnz/s when x lives beyond L2, i.e.
// <= 9 % of the HBM roofline whatever the kernel.  This path replaces the random
// access by two streaming passes:
//   phase 1  pb_expand : the matrix is stored SLICE-MAJOR (slices of 16 Ki columns,
//            inside a slice sorted by row, then column).  A workgroup stages its
//            x slice in LDS (128 KiB), streams {value, 16-bit local column} and
//            writes the products back in the same order (coalesced).
//   phase 2  pb_reduce : a workgroup owns a bin of R consecutive rows whose partial
//            sums live in LDS (R x 8 B); it walks the bin's tile list
//            (slice 0..S-1; each tile is a contiguous run of the product array),
//            adds products with ds_add_f64 and finally stores the bin of y.
// ~28-30 B/nnz of pure streaming instead of 12 B/nnz + one line per nnz.  Measured ceilings on
// the box (scripts/microbench_copy.hip): a 10 B read : 8 B write stream runs at 5.0 TB/s, a pure
// read stream at 6.5 TB/s -> phase 1 >= 5.8 ms and phase 2 >= ~3.4 ms for 1.6 G nnz.
//
// Relation to the reference: its CPU variants spmvTilesCSR / spmvTilesAllocdCSR
// (src/SpMV_CSR_OMP.c:101-226) use the same 2-D decomposition -- column
// partitions, partial results per tile, final reduction; this is the GPU form
// of that idea with the partition sizes dictated by LDS.
//
// Summation order: products of one row are added by LDS atomics in arrival
// order -> results agree with the serial oracle to rounding (not bitwise, and
// not bitwise run-to-run).  The parity gate and the 1e-13 * sum|a x| check hold.
//
// The slice-major format is built ON THE DEVICE from the device CSR (one-time
// analysis, like the row-block table): rocPRIM's stable radix sort by slice id
// keeps the (row, column) order of the CSR inside every slice.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

namespace spmvhip {

constexpr uint32_t PB_C_SHIFT = 14;                 // 16 Ki columns per slice = 128 KiB of x in LDS
constexpr uint32_t PB_C       = 1u << PB_C_SHIFT;
constexpr uint32_t PB_R_MAX_SHIFT = 14;             // <= 16 Ki rows per bin = 128 KiB of y in LDS
constexpr uint32_t PB_CHUNK   = 1u << 17;           // entries of one slice handled by one phase-1 workgroup
constexpr int      PB_THREADS = 1024;

struct TileFormat {
    uint32_t S = 0, B = 0, rShift = 0;              // slices, bins, log2(rows per bin)
    uint64_t nnz = 0;
    double*   val = nullptr;                        // slice-major values
    uint16_t* lcol = nullptr;                       // column - slice*PB_C
    uint16_t* lrow = nullptr;                       // row - bin*R
    uint2*    desc = nullptr;                       // per bin: {start, len <= PB_SPLIT} runs of the slice-major arrays
    uint32_t* binDesc = nullptr;                    // [B+1] first descriptor of each bin
    uint32_t  nDesc = 0;
    uint3*    work = nullptr;                       // phase-1 work items {slice, begin, end}
    uint32_t  nWork = 0;
    double*   prod = nullptr;                       // products, slice-major (workspace)
    size_t    bytes = 0;
};

namespace {

__device__ __forceinline__ uint64_t lin_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }

// ---- build kernels ---------------------------------------------------------------------------
template <typename I>
__global__ __launch_bounds__(256) void pb_rowof_kernel(uint64_t M, const I* __restrict__ IRP, uint32_t* __restrict__ rowOf) {
    const uint64_t r = lin_block() * 4 + threadIdx.x / 64;
    if (r >= M) return;
    const uint64_t b = IRP[r], e = IRP[r + 1];
    for (uint64_t j = b + threadIdx.x % 64; j < e; j += 64) rowOf[j] = (uint32_t)r;
}

__global__ __launch_bounds__(256) void pb_keys_kernel(uint64_t nnz, const uint32_t* __restrict__ JA,
                                                      uint16_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t j = lin_block() * 256 + threadIdx.x;
    if (j >= nnz) return;
    keys[j] = (uint16_t)(JA[j] >> PB_C_SHIFT);
    idx[j] = (uint32_t)j;
}

// permute into slice-major order and mark where each tile starts.  tile id t = slice*B + bin is
// non-decreasing along the sorted order, so the first entry of a tile also fills the start of
// every empty tile before it.
__global__ __launch_bounds__(256) void pb_gather_kernel(
    uint64_t nnz, const uint32_t* __restrict__ perm, const uint16_t* __restrict__ skeys,
    const uint32_t* __restrict__ rowOf, const uint32_t* __restrict__ JA, const double* __restrict__ AS,
    uint32_t B, uint32_t rShift, uint64_t nTiles, double* __restrict__ val, uint16_t* __restrict__ lcol,
    uint16_t* __restrict__ lrow, uint32_t* __restrict__ tileStart) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t j = perm[p];
    const uint32_t row = rowOf[j];
    val[p] = AS[j];
    lcol[p] = (uint16_t)(JA[j] & (PB_C - 1));
    lrow[p] = (uint16_t)(row & ((1u << rShift) - 1));
    const uint64_t t = (uint64_t)skeys[p] * B + (row >> rShift);
    uint64_t tPrev;                                  // tile of the previous entry, or "-1"
    if (p == 0) tPrev = ~0ull;
    else tPrev = (uint64_t)skeys[p - 1] * B + (rowOf[perm[p - 1]] >> rShift);
    if (t != tPrev)
        for (uint64_t u = tPrev + 1; u <= t; ++u) tileStart[u] = (uint32_t)p;      // tPrev+1 wraps to 0 for p == 0
    if (p == nnz - 1)
        for (uint64_t u = t + 1; u <= nTiles; ++u) tileStart[u] = (uint32_t)nnz;
}

__global__ __launch_bounds__(256) void pb_fill_kernel(uint32_t* p, uint64_t n, uint32_t v) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- phase 1 -----------------------------------------------------------------------------------
// Each lane owns 4 consecutive entries per step (2 x 16-B value loads, one 8-B column load,
// 2 x 16-B product stores); the work item's range is peeled to a multiple of 4 so every
// vector access is naturally aligned.  Two steps are kept in flight; the first one is issued
// before the x slice is staged so the stream is already moving during the LDS fill.
typedef double   dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush4 __attribute__((ext_vector_type(4)));

constexpr int P1_DEPTH = 4;                         // vector steps in flight per lane (x2: current + next batch)
struct P1Regs { dbl2 a[P1_DEPTH], b[P1_DEPTH]; ush4 c[P1_DEPTH]; };

__device__ __forceinline__ void p1_load(P1Regs& r, uint32_t p, uint32_t ve, const double* __restrict__ val,
                                        const uint16_t* __restrict__ lcol) {
    constexpr uint32_t STEP = 4 * PB_THREADS;
#pragma unroll
    for (int u = 0; u < P1_DEPTH; ++u) {
        const uint32_t q = p + u * STEP;
        if (q < ve) {
            r.a[u] = __builtin_nontemporal_load((const dbl2*)(val + q));
            r.b[u] = __builtin_nontemporal_load((const dbl2*)(val + q + 2));
            r.c[u] = __builtin_nontemporal_load((const ush4*)(lcol + q));
        } else { r.a[u] = 0; r.b[u] = 0; r.c[u] = 0; }
    }
}

__global__ __launch_bounds__(PB_THREADS) void pb_expand_kernel(
    const uint3* __restrict__ work, const double* __restrict__ val, const uint16_t* __restrict__ lcol,
    const double* __restrict__ x, uint64_t N, double* __restrict__ prod) {
    extern __shared__ double xs[];                  // PB_C doubles
    const uint3 w = work[lin_block()];
    const uint32_t begin = w.y, end = w.z;
    const uint32_t vb = min(end, (begin + 3u) & ~3u);           // first multiple of 4 inside the range
    const uint32_t ve = vb + ((end - vb) & ~3u);                // end of the whole groups of 4
    constexpr uint32_t STEP = 4 * PB_THREADS;

    uint32_t p = vb + 4 * threadIdx.x;
    P1Regs cur, nxt;
    p1_load(cur, p, ve, val, lcol);                 // the stream starts moving before the x slice is staged

    const uint64_t col0 = (uint64_t)w.x << PB_C_SHIFT;
    {
        double xv[PB_C / PB_THREADS];
#pragma unroll
        for (uint32_t i = 0; i < PB_C / PB_THREADS; ++i) {
            const uint32_t k = threadIdx.x + i * PB_THREADS;
            xv[i] = (col0 + k < N) ? x[col0 + k] : 0.0;
        }
#pragma unroll
        for (uint32_t i = 0; i < PB_C / PB_THREADS; ++i) xs[threadIdx.x + i * PB_THREADS] = xv[i];
    }
    __syncthreads();

    // scalar head and tail (at most 3 entries each)
    if (threadIdx.x < vb - begin) { const uint32_t q = begin + threadIdx.x; prod[q] = val[q] * xs[lcol[q]]; }
    if (threadIdx.x < end - ve)   { const uint32_t q = ve + threadIdx.x;    prod[q] = val[q] * xs[lcol[q]]; }

    for (; p < ve; p += P1_DEPTH * STEP) {
        p1_load(nxt, p + P1_DEPTH * STEP, ve, val, lcol);
#pragma unroll
        for (int u = 0; u < P1_DEPTH; ++u) {
            const uint32_t q = p + u * STEP;
            if (q < ve) {
                dbl2 r0, r1;
                r0.x = cur.a[u].x * xs[cur.c[u].x]; r0.y = cur.a[u].y * xs[cur.c[u].y];
                r1.x = cur.b[u].x * xs[cur.c[u].z]; r1.y = cur.b[u].y * xs[cur.c[u].w];
                __builtin_nontemporal_store(r0, (dbl2*)(prod + q));
                __builtin_nontemporal_store(r1, (dbl2*)(prod + q + 2));
            }
        }
        cur = nxt;
    }
}

// ---- phase 2 -----------------------------------------------------------------------------------
// A bin's work is a list of descriptors {start, len <= PB_SPLIT}: the bin's non-empty tiles in
// slice order, long tiles cut into pieces.  Each wavefront takes 64 consecutive descriptors at a
// time (coalesced load, walked with shuffles) and keeps two groups of GROUP runs in flight:
// the loads of the next group are issued before the LDS atomics of the current one.
constexpr uint32_t PB_SPLIT = 64;                   // entries per descriptor = one wavefront-wide access
constexpr int      GROUP    = 8;

struct RunRegs { double pv[GROUP]; uint16_t rv[GROUP]; uint32_t ln[GROUP]; };

__device__ __forceinline__ void pb_fetch(RunRegs& r, uint2 mine, uint32_t t, uint32_t cnt, uint32_t lane,
                                         const double* __restrict__ prod, const uint16_t* __restrict__ lrow) {
#pragma unroll
    for (int q = 0; q < GROUP; ++q) {
        const uint32_t st = __shfl(mine.x, (int)((t + q) & 63));
        const uint32_t ln = (t + q < cnt) ? __shfl(mine.y, (int)((t + q) & 63)) : 0u;
        r.ln[q] = ln;
        const bool in = lane < ln;
        r.pv[q] = in ? __builtin_nontemporal_load(prod + st + lane) : 0.0;
        r.rv[q] = in ? __builtin_nontemporal_load(lrow + st + lane) : (uint16_t)0;
    }
}

__global__ __launch_bounds__(PB_THREADS) void pb_reduce_kernel(
    uint32_t rShift, uint64_t M, const uint32_t* __restrict__ binDesc, const uint2* __restrict__ desc,
    const double* __restrict__ prod, const uint16_t* __restrict__ lrow, double* __restrict__ y) {
    extern __shared__ double yb[];                  // R doubles
    const uint32_t R = 1u << rShift;
    const uint64_t bin = lin_block();
    for (uint32_t k = threadIdx.x; k < R; k += PB_THREADS) yb[k] = 0.0;
    __syncthreads();
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    constexpr uint32_t NW = PB_THREADS / 64;
    const uint32_t d0 = binDesc[bin], nD = binDesc[bin + 1] - d0;
    const uint2* dsc = desc + d0;
    for (uint32_t s0 = wave * 64; s0 < nD; s0 += NW * 64) {
        uint2 mine = make_uint2(0, 0);
        if (s0 + lane < nD) mine = dsc[s0 + lane];
        const uint32_t cnt = min(64u, nD - s0);
        RunRegs a, b;
        pb_fetch(a, mine, 0, cnt, lane, prod, lrow);
        for (uint32_t t = 0; t < cnt; t += GROUP) {
            if (t + GROUP < cnt) pb_fetch(b, mine, t + GROUP, cnt, lane, prod, lrow);
#pragma unroll
            for (int q = 0; q < GROUP; ++q)
                if (lane < a.ln[q]) atomicAdd(&yb[a.rv[q]], a.pv[q]);
            a = b;
        }
    }
    __syncthreads();
    const uint64_t r0 = bin << rShift;
    for (uint32_t k = threadIdx.x; k < R; k += PB_THREADS)
        if (r0 + k < M) y[r0 + k] = yb[k];
}

// ---- descriptor-list build ----------------------------------------------------------------------
// pieces[b*S+s] = number of descriptors of tile (s,b) = ceil(len / PB_SPLIT)
__global__ __launch_bounds__(256) void pb_pieces_kernel(uint32_t S, uint32_t B, const uint32_t* __restrict__ tileStart,
                                                        uint32_t* __restrict__ pieces) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i >= (uint64_t)S * B) return;
    const uint32_t b = (uint32_t)(i / S), s = (uint32_t)(i % S);
    const uint64_t t = (uint64_t)s * B + b;
    pieces[i] = (tileStart[t + 1] - tileStart[t] + PB_SPLIT - 1) / PB_SPLIT;
}
// offs = exclusive scan of pieces (bin-major) -> write the descriptors and the per-bin list starts
__global__ __launch_bounds__(256) void pb_desc_kernel(uint32_t S, uint32_t B, const uint32_t* __restrict__ tileStart,
                                                      const uint32_t* __restrict__ offs, uint32_t total,
                                                      uint2* __restrict__ desc, uint32_t* __restrict__ binDesc) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i >= (uint64_t)S * B) return;
    const uint32_t b = (uint32_t)(i / S), s = (uint32_t)(i % S);
    const uint64_t t = (uint64_t)s * B + b;
    uint32_t a = tileStart[t];
    const uint32_t e = tileStart[t + 1];
    uint32_t o = offs[i];
    if (s == 0) binDesc[b] = o;
    if (i + 1 == (uint64_t)S * B) binDesc[B] = total;
    for (; a < e; a += PB_SPLIT) desc[o++] = make_uint2(a, min(PB_SPLIT, e - a));
}

#define PB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "libspmvhip: tiles: %s: %s\n", #expr, hipGetErrorString(e_)); return EXIT_FAILURE; } } while (0)

struct TempBuf {
    void* p = nullptr;
    ~TempBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 1)); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

void freeTiles(TileFormat* t) {
    if (!t) return;
    (void)hipFree(t->val); (void)hipFree(t->lcol); (void)hipFree(t->lrow); (void)hipFree(t->desc); (void)hipFree(t->binDesc);
    (void)hipFree(t->work); (void)hipFree(t->prod);
    delete t;
}

int buildTiles(DevMat* d) {
    if (d->tiles) return EXIT_SUCCESS;
    if (d->kind != Kind::CSR) return EXIT_FAILURE;
    const uint64_t nnz = d->NZ, M = d->M, N = d->N;
    if (nnz >= IRP32_LIMIT || nnz == 0) { fprintf(stderr, "libspmvhip: tiles: nnz = %lu unsupported (needs 0 < nnz < 2^32)\n", (unsigned long)nnz); return EXIT_FAILURE; }
    const uint64_t S64 = (N + PB_C - 1) >> PB_C_SHIFT;
    if (S64 > 65535) { fprintf(stderr, "libspmvhip: tiles: %lu columns exceed 65535 slices\n", (unsigned long)N); return EXIT_FAILURE; }
    TileFormat* t = new TileFormat;
    t->S = (uint32_t)S64;
    // rows per bin: as large as LDS allows (longer tiles = longer contiguous runs in phase 2); halve it while
    // there are fewer than ~1024 bins (phase 2 should fill the chip several times over) AND the tiles would
    // still average >= 64 entries.  Measured: c3 (10 M x 10 M) 8 Ki rows 1.27 ms vs 16 Ki 1.35 ms; a
    // 10 M x 80 M shard of c5 16 Ki 1.44 ms, 8 Ki 1.50 ms, 4 Ki 1.95 ms.
    uint32_t rShift = PB_R_MAX_SHIFT;
    auto binsAt = [&](uint32_t sh) { return (M + (1ull << sh) - 1) >> sh; };
    while (rShift > 10 && binsAt(rShift) < 1024 && nnz / ((uint64_t)t->S * binsAt(rShift - 1)) >= 64) --rShift;
    if (const char* e = getenv("SPMV_PB_RSHIFT")) { const int v = atoi(e); if (v >= 10 && v <= (int)PB_R_MAX_SHIFT) rShift = (uint32_t)v; }   // tuning only
    t->rShift = rShift;
    t->B = (uint32_t)((M + (1ull << rShift) - 1) >> rShift);
    t->nnz = nnz;
    const uint64_t nTiles = (uint64_t)t->S * t->B;
    if (nTiles >= (1ull << 32) - 2) { fprintf(stderr, "libspmvhip: tiles: too many tiles\n"); delete t; return EXIT_FAILURE; }

    TempBuf rowOf, keys, keysOut, idx, perm, sortTmp, tileStart;
    auto fail = [&](const char* what) { fprintf(stderr, "libspmvhip: tiles: %s failed\n", what); freeTiles(t); return EXIT_FAILURE; };
    if (rowOf.alloc(nnz * 4) || keys.alloc(nnz * 2) || keysOut.alloc(nnz * 2) || idx.alloc(nnz * 4) || perm.alloc(nnz * 4) ||
        tileStart.alloc((nTiles + 2) * 4))
        return fail("temporary allocation");
    if (hipMalloc(&t->val, nnz * 8) || hipMalloc(&t->lcol, nnz * 2) || hipMalloc(&t->lrow, nnz * 2) ||
        hipMalloc(&t->prod, nnz * 8) || hipMalloc(&t->binDesc, ((size_t)t->B + 1) * 4))
        return fail("format allocation");

    if (d->irpBytes == 4)
        hipLaunchKernelGGL((pb_rowof_kernel<uint32_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint32_t*>(d->IRP), rowOf.as<uint32_t>());
    else
        hipLaunchKernelGGL((pb_rowof_kernel<uint64_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint64_t*>(d->IRP), rowOf.as<uint32_t>());
    hipLaunchKernelGGL(pb_keys_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, d->JA, keys.as<uint16_t>(), idx.as<uint32_t>());
    PB_TRY(hipGetLastError());

    unsigned bits = 1;
    while ((1u << bits) < t->S) ++bits;
    size_t tmpBytes = 0;
    PB_TRY(rocprim::radix_sort_pairs(nullptr, tmpBytes, keys.as<uint16_t>(), keysOut.as<uint16_t>(), idx.as<uint32_t>(),
                                     perm.as<uint32_t>(), (size_t)nnz, 0, bits, (hipStream_t) nullptr));
    if (sortTmp.alloc(tmpBytes)) return fail("sort workspace");
    PB_TRY(rocprim::radix_sort_pairs(sortTmp.p, tmpBytes, keys.as<uint16_t>(), keysOut.as<uint16_t>(), idx.as<uint32_t>(),
                                     perm.as<uint32_t>(), (size_t)nnz, 0, bits, (hipStream_t) nullptr));

    hipLaunchKernelGGL(pb_gather_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, perm.as<uint32_t>(),
                       keysOut.as<uint16_t>(), rowOf.as<uint32_t>(), d->JA, d->AS, t->B, rShift, nTiles, t->val, t->lcol,
                       t->lrow, tileStart.as<uint32_t>());
    PB_TRY(hipGetLastError());
    // per-bin descriptor lists: count pieces per tile (bin-major), scan, write
    {
        TempBuf pieces, offs, scanTmp;
        if (pieces.alloc(nTiles * 4) || offs.alloc(nTiles * 4)) return fail("descriptor workspace");
        hipLaunchKernelGGL(pb_pieces_kernel, grid2d((nTiles + 255) / 256, 256), dim3(256), 0, nullptr, t->S, t->B,
                           tileStart.as<uint32_t>(), pieces.as<uint32_t>());
        size_t scanBytes = 0;
        PB_TRY(rocprim::exclusive_scan(nullptr, scanBytes, pieces.as<uint32_t>(), offs.as<uint32_t>(), 0u, (size_t)nTiles,
                                       rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
        if (scanTmp.alloc(scanBytes)) return fail("scan workspace");
        PB_TRY(rocprim::exclusive_scan(scanTmp.p, scanBytes, pieces.as<uint32_t>(), offs.as<uint32_t>(), 0u, (size_t)nTiles,
                                       rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
        uint32_t lastOff = 0, lastCnt = 0;
        PB_TRY(hipMemcpy(&lastOff, offs.as<uint32_t>() + nTiles - 1, 4, hipMemcpyDeviceToHost));
        PB_TRY(hipMemcpy(&lastCnt, pieces.as<uint32_t>() + nTiles - 1, 4, hipMemcpyDeviceToHost));
        t->nDesc = lastOff + lastCnt;
        if (hipMalloc(&t->desc, std::max<size_t>(t->nDesc, 1) * sizeof(uint2))) return fail("descriptor allocation");
        hipLaunchKernelGGL(pb_desc_kernel, grid2d((nTiles + 255) / 256, 256), dim3(256), 0, nullptr, t->S, t->B,
                           tileStart.as<uint32_t>(), offs.as<uint32_t>(), t->nDesc, t->desc, t->binDesc);
        PB_TRY(hipGetLastError());
        PB_TRY(hipDeviceSynchronize());
    }
    t->bytes = nnz * 20 + (size_t)t->nDesc * sizeof(uint2) + ((size_t)t->B + 1) * 4;

    // phase-1 work list from the slice boundaries (tileStart[s*B])
    std::vector<uint32_t> sliceStart(t->S + 1);
    PB_TRY(hipMemcpy2D(sliceStart.data(), 4, tileStart.as<uint32_t>(), (size_t)t->B * 4, 4, t->S + 1, hipMemcpyDeviceToHost));
    std::vector<uint3> work;
    for (uint32_t s = 0; s < t->S; ++s) {
        // a slice is cut into equal pieces of at most PB_CHUNK entries (multiples of 4 keep the vector loads aligned);
        // a fixed chunk size + remainder left one short, fill-dominated work item per slice
        const uint32_t b0 = sliceStart[s], len = sliceStart[s + 1] - b0;
        if (!len) continue;
        const uint32_t pieces = (len + PB_CHUNK - 1) / PB_CHUNK;
        const uint32_t piece = ((len + pieces - 1) / pieces + 3) & ~3u;
        for (uint32_t b = b0; b < b0 + len; b += piece)
            work.push_back(make_uint3(s, b, std::min<uint32_t>(b + piece, b0 + len)));
    }
    t->nWork = (uint32_t)work.size();
    PB_TRY(hipMalloc(&t->work, std::max<size_t>(work.size(), 1) * sizeof(uint3)));
    PB_TRY(hipMemcpy(t->work, work.data(), work.size() * sizeof(uint3), hipMemcpyHostToDevice));
    PB_TRY(hipDeviceSynchronize());

    static bool attrSet = false;
    if (!attrSet) {
        PB_TRY(hipFuncSetAttribute((const void*)pb_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PB_C * 8));
        PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (1 << PB_R_MAX_SHIFT) * 8));
        attrSet = true;
    }
    d->tiles = t;
    return EXIT_SUCCESS;
}

size_t tilesBytes(const DevMat* d) { return d->tiles ? d->tiles->bytes : 0; }

// enqueue both phases on `stream`
int enqueueTiles(DevMat* d, const double* x, double* y, hipStream_t stream) {
    TileFormat* t = d->tiles;
    if (!t) return EXIT_FAILURE;
    if (t->nWork)
        hipLaunchKernelGGL(pb_expand_kernel, grid2d(t->nWork, PB_THREADS), dim3(PB_THREADS), PB_C * 8, stream, t->work, t->val,
                           t->lcol, x, d->N, t->prod);
    hipLaunchKernelGGL(pb_reduce_kernel, grid2d(t->B, PB_THREADS), dim3(PB_THREADS), (size_t)8 << t->rShift, stream,
                       t->rShift, d->M, t->binDesc, t->desc, t->prod, t->lrow, y);
    return hipGetLastError() == hipSuccess ? EXIT_SUCCESS : EXIT_FAILURE;
}

}  // namespace spmvhip
