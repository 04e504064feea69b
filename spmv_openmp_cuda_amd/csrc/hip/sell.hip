// sell.hip -- SELL-C-sigma ("sorted, sliced ELLPACK") on the device, C = 64 = one wavefront.
//
// pure ELL of the power-law matrix is impossible (10 M x 50 k slots = 6 TB, and the
// reference's loader refuses it, src/lib/parser.c:223-232); the reference author's own notes point at
// SELL-C-sigma as the way out.  This is that format, built on the device from the uploaded CSR:
//   * rows are sorted by length (descending) inside windows of SIGMA consecutive rows -- similar rows
//     meet in a slice, yet a slice still draws its rows from one neighbourhood, so a banded matrix
//     keeps its x locality;
//   * a slice = 64 sorted rows, stored column-major (entry k of lane l at sliceOff + k*64 + l) with
//     the slice's own width = its longest row: padding is a few per cent instead of ELL's M*maxRow;
//   * kernel: one wavefront per slice, one LANE per row, entries walked in ascending-j order ->
//     coalesced 512-B/256-B accesses, no LDS, no barrier, and bit-identical to the serial oracle
//     (the thread-per-row order of the reference's cudaSpMVRowsELL, src/SpMV_CUDA.cu:79-96);
//   * slice quadruples (one workgroup) are dealt to the 8 XCDs in contiguous ranges: with the default
//     round-robin placement every XCD's L2 sees every sorting window and thrashes (measured on a
//     +-512 band, sigma = 16 Ki: 3.8 ms round-robin vs 1.29 ms contiguous);
//   * (tried and removed, profiles/r02_sell_column_stripes_c2.log: storing the slices per COLUMN STRIPE of 1-4 MiB of x,
//     one launch per stripe with a lane continuing its row's sum from y, so that every launch gathers from a window of
//     x that stays in an XCD's L2.  Bit-identical and correct -- and no faster on config 2: 0.232 ms against 0.237.
//     With one lane per row the 64 lanes of a gather touch 64 different lines whatever cache serves them, and the rate
//     at which a CU's L1 takes lines from L2 (~0.3 per clock) is the bound, not where the lines come from;)
//   * (tried and removed, round 2: slices stored in groups of 4 slots -- entry k of lane l at (k/4)*256 + l*4 + k%4 -- so
//     that a lane fetches 4 columns with one 16-byte load and 4 values with two: 7 instead of 12 vector-memory
//     instructions per 4 entries, bit-identical, c2 0.244 ms against 0.238.  The coalesced loads are not what the L1
//     spends its time on; the 64-line gathers are;)
//   * (tried and removed, round 2: eight slices per wavefront with the loop over the slots outside the loop over the
//     slices, so that the whole launch walks slot 0, 1, 2, ... -- i.e. ascending columns -- together and gathers from one
//     moving window of x: c2 0.225 ms against 0.238, but 4.2 vs 3.8 ms on c3 and 2.5-2.7 vs 1.3-1.6 ms on the banded
//     matrices, whose locality lives inside a slice; profiles/r02_sell_sweep.log;)
//   * rows longer than SELL_MAX_ROW do not enter the slices (one of them would pin a wavefront for
//     a whole slice): they are processed workgroup-per-row from the CSR arrays (shuffle tree).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

namespace spmvhip {

constexpr uint32_t SELL_C       = 64;
constexpr uint32_t SELL_SIGMA   = 1u << 14;         // rows per sorting window (a multiple of SELL_C)
constexpr uint32_t SELL_MAX_ROW = 256;              // longer rows take the workgroup-per-row path (a long row would pin a lane, and pad its slice)

struct SellFormat {
    uint32_t  nSlices = 0, nLong = 0;
    uint64_t* sliceOff = nullptr;       // [nSlices+1] start of each slice in val/col
    uint32_t* perm = nullptr;           // sorted position -> original row (padded to nSlices*64 with 0xFFFFFFFF)
    uint32_t* slen = nullptr;           // sorted position -> entries of that row held in its slice
    double*   val = nullptr;
    uint32_t* col = nullptr;
    uint32_t* longRows = nullptr;       // original row ids of the rows longer than SELL_MAX_ROW
    size_t    bytes = 0;
};

namespace {

__device__ __forceinline__ uint64_t lin_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }

template <typename I>
__global__ __launch_bounds__(256) void sell_keys_kernel(uint64_t M, uint64_t padded, uint32_t sigma, const I* __restrict__ IRP,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ rows) {
    const uint64_t r = lin_block() * 256 + threadIdx.x;
    if (r >= padded) return;
    uint64_t len = 0;
    if (r < M) {
        len = (uint64_t)IRP[r + 1] - (uint64_t)IRP[r];
        if (len > SELL_MAX_ROW) len = 0;            // long rows occupy no slots (they sort to the window's end)
    }
    // ascending sort of (window, ~len): descending length inside each window; pad rows go last overall
    keys[r] = r < M ? ((r / sigma) << 32 | (0xFFFFFFFFull - len)) : ~0ull;
    rows[r] = r < M ? (uint32_t)r : 0xFFFFFFFFu;
}

// per sorted position: row length in the slice; per slice: width * 64 (for the scan)
template <typename I>
__global__ __launch_bounds__(256) void sell_widths_kernel(uint64_t M, uint32_t nSlices, const I* __restrict__ IRP,
                                                          const uint32_t* __restrict__ perm, uint32_t* __restrict__ slen,
                                                          uint64_t* __restrict__ sliceCells) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= (uint64_t)nSlices * SELL_C) return;
    const uint32_t row = perm[p];
    uint32_t len = 0;
    if (row != 0xFFFFFFFFu) {
        const uint64_t l = (uint64_t)IRP[row + 1] - (uint64_t)IRP[row];
        len = l > SELL_MAX_ROW ? 0u : (uint32_t)l;
    }
    slen[p] = len;
    if (p % SELL_C == 0) sliceCells[p / SELL_C] = (uint64_t)len * SELL_C;   // first row of a slice is its longest
}

// one wavefront per slice copies its rows into the column-major slice
template <typename I>
__global__ __launch_bounds__(256) void sell_fill_kernel(uint32_t nSlices, const uint64_t* __restrict__ sliceOff,
                                                        const uint32_t* __restrict__ perm, const uint32_t* __restrict__ slen,
                                                        const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                        const double* __restrict__ AS, double* __restrict__ val,
                                                        uint32_t* __restrict__ col) {
    const uint64_t s = lin_block() * 4 + threadIdx.x / 64;
    if (s >= nSlices) return;
    const uint32_t lane = threadIdx.x % 64;
    const uint64_t off = sliceOff[s];
    const uint32_t width = (uint32_t)((sliceOff[s + 1] - off) / SELL_C);
    const uint32_t row = perm[s * SELL_C + lane], n = slen[s * SELL_C + lane];
    const uint64_t b = row != 0xFFFFFFFFu ? (uint64_t)IRP[row] : 0;
    for (uint32_t k = 0; k < width; ++k) {
        const bool in = k < n;
        val[off + (uint64_t)k * SELL_C + lane] = in ? AS[b + k] : 0.0;
        col[off + (uint64_t)k * SELL_C + lane] = in ? JA[b + k] : 0u;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void sell_longlist_kernel(uint64_t M, const I* __restrict__ IRP,
                                                            uint32_t* __restrict__ longRows, uint32_t* __restrict__ counter) {
    const uint64_t r = lin_block() * 256 + threadIdx.x;
    if (r >= M) return;
    if ((uint64_t)IRP[r + 1] - (uint64_t)IRP[r] > SELL_MAX_ROW) longRows[atomicAdd(counter, 1u)] = (uint32_t)r;
}

// ---- the SpMV kernels ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sell_spmv_kernel(uint32_t nSlices, const uint64_t* __restrict__ sliceOff,
                                                        const uint32_t* __restrict__ perm, const uint32_t* __restrict__ slen,
                                                        const double* __restrict__ val, const uint32_t* __restrict__ col,
                                                        const double* __restrict__ x, double* __restrict__ y) {
    // workgroups b, b+8, ... share an XCD: deal the slice quadruples to the XCDs in contiguous ranges so that
    // the slices of one sorting window -- and their x window -- meet in one L2
    const uint64_t nWg = ((uint64_t)nSlices + 3) / 4, q = lin_block();
    if (q >= nWg) return;
    const uint64_t a = nWg / 8, rem = nWg % 8, xcd = q % 8;
    const uint64_t wg = xcd * a + (xcd < rem ? xcd : rem) + q / 8;      // bijection of [0, nWg)
    const uint64_t s = wg * 4 + threadIdx.x / 64;
    if (s >= nSlices) return;
    const uint32_t lane = threadIdx.x % 64;
    const uint32_t row = perm[s * SELL_C + lane], n = slen[s * SELL_C + lane];
    const double*   v = val + sliceOff[s] + lane;
    const uint32_t* c = col + sliceOff[s] + lane;
    double acc = 0;
    uint32_t k = 0;
    for (; k + 4 <= n; k += 4) {
        const uint32_t c0 = __builtin_nontemporal_load(c + (size_t)k * SELL_C), c1 = __builtin_nontemporal_load(c + (size_t)(k + 1) * SELL_C),
                       c2 = __builtin_nontemporal_load(c + (size_t)(k + 2) * SELL_C), c3 = __builtin_nontemporal_load(c + (size_t)(k + 3) * SELL_C);
        const double a0 = __builtin_nontemporal_load(v + (size_t)k * SELL_C), a1 = __builtin_nontemporal_load(v + (size_t)(k + 1) * SELL_C),
                     a2 = __builtin_nontemporal_load(v + (size_t)(k + 2) * SELL_C), a3 = __builtin_nontemporal_load(v + (size_t)(k + 3) * SELL_C);
        const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        acc += a0 * x0; acc += a1 * x1; acc += a2 * x2; acc += a3 * x3;         // ascending j
    }
    for (; k < n; ++k) acc += __builtin_nontemporal_load(v + (size_t)k * SELL_C) * x[__builtin_nontemporal_load(c + (size_t)k * SELL_C)];
    // pad lanes (0xFFFFFFFF) and rows owned by sell_long_kernel (top bit set at build time) store nothing
    if (!(row & 0x80000000u)) y[row] = acc;
}

template <typename I>
__global__ __launch_bounds__(256) void sell_long_kernel(uint32_t nLong, const uint32_t* __restrict__ longRows,
                                                        const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                        const double* __restrict__ AS, const double* __restrict__ x,
                                                        double* __restrict__ y) {
    __shared__ double wpart[4];
    const uint64_t i = lin_block();
    if (i >= nLong) return;
    const uint32_t row = longRows[i], tid = threadIdx.x;
    const uint64_t b = IRP[row], e = IRP[row + 1];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint64_t j = b + tid;
    for (; j + 768 < e; j += 1024) {
        a0 += __builtin_nontemporal_load(AS + j) * x[__builtin_nontemporal_load(JA + j)];
        a1 += __builtin_nontemporal_load(AS + j + 256) * x[__builtin_nontemporal_load(JA + j + 256)];
        a2 += __builtin_nontemporal_load(AS + j + 512) * x[__builtin_nontemporal_load(JA + j + 512)];
        a3 += __builtin_nontemporal_load(AS + j + 768) * x[__builtin_nontemporal_load(JA + j + 768)];
    }
    for (; j < e; j += 256) a0 += __builtin_nontemporal_load(AS + j) * x[__builtin_nontemporal_load(JA + j)];
    double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (tid % 64 == 0) wpart[tid / 64] = acc;
    __syncthreads();
    if (tid == 0) y[row] = (wpart[0] + wpart[1]) + (wpart[2] + wpart[3]);
}

// flag the long rows in perm (top bit) so the slice kernel leaves their y alone
template <typename I>
__global__ __launch_bounds__(256) void sell_flag_long_kernel(uint64_t n, const I* __restrict__ IRP, uint32_t* __restrict__ perm) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t row = perm[p];
    if (row == 0xFFFFFFFFu) return;
    if ((uint64_t)IRP[row + 1] - (uint64_t)IRP[row] > SELL_MAX_ROW) perm[p] = row | 0x80000000u;
}

#define SELL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: sell: %s: %s\n", #expr, hipGetErrorString(e_)); return EXIT_FAILURE; } } while (0)

struct Tmp {
    void* p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 1)); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

template <typename I>
int buildSellT(DevMat* d, SellFormat* f) {
    const uint64_t M = d->M;
    const I* IRP = static_cast<const I*>(d->IRP);
    const uint64_t padded = (M + SELL_C - 1) / SELL_C * SELL_C;
    f->nSlices = (uint32_t)(padded / SELL_C);
    Tmp keys, keysOut, rows, sortTmp, cells, scanTmp;
    if (keys.alloc(padded * 8) || keysOut.alloc(padded * 8) || rows.alloc(padded * 4) || cells.alloc(((size_t)f->nSlices + 1) * 8)) return EXIT_FAILURE;
    SELL_TRY(hipMalloc(&f->perm, std::max<size_t>(padded, 1) * 4));
    SELL_TRY(hipMalloc(&f->slen, std::max<size_t>(padded, 1) * 4));
    SELL_TRY(hipMalloc(&f->sliceOff, ((size_t)f->nSlices + 1) * 8));
    const dim3 gRows = grid2d((padded + 255) / 256, 256);
    const uint32_t sigma = SELL_SIGMA;              // whole slices: a window that ends inside a slice would break "first row of a slice is its longest"
    hipLaunchKernelGGL((sell_keys_kernel<I>), gRows, dim3(256), 0, nullptr, M, padded, sigma, IRP, keys.as<uint64_t>(), rows.as<uint32_t>());
    size_t tmpBytes = 0;
    SELL_TRY(rocprim::radix_sort_pairs(nullptr, tmpBytes, keys.as<uint64_t>(), keysOut.as<uint64_t>(), rows.as<uint32_t>(), f->perm,
                                       (size_t)padded, 0, 64, (hipStream_t) nullptr));
    if (sortTmp.alloc(tmpBytes)) return EXIT_FAILURE;
    SELL_TRY(rocprim::radix_sort_pairs(sortTmp.p, tmpBytes, keys.as<uint64_t>(), keysOut.as<uint64_t>(), rows.as<uint32_t>(), f->perm,
                                       (size_t)padded, 0, 64, (hipStream_t) nullptr));
    hipLaunchKernelGGL((sell_widths_kernel<I>), gRows, dim3(256), 0, nullptr, M, f->nSlices, IRP, f->perm, f->slen, cells.as<uint64_t>());
    SELL_TRY(hipMemsetAsync(cells.as<uint64_t>() + f->nSlices, 0, 8, nullptr));
    size_t scanBytes = 0;
    SELL_TRY(rocprim::exclusive_scan(nullptr, scanBytes, cells.as<uint64_t>(), f->sliceOff, (uint64_t)0, (size_t)f->nSlices + 1,
                                     rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
    if (scanTmp.alloc(scanBytes)) return EXIT_FAILURE;
    SELL_TRY(rocprim::exclusive_scan(scanTmp.p, scanBytes, cells.as<uint64_t>(), f->sliceOff, (uint64_t)0, (size_t)f->nSlices + 1,
                                     rocprim::plus<uint64_t>(), (hipStream_t) nullptr));
    uint64_t total = 0;
    SELL_TRY(hipMemcpy(&total, f->sliceOff + f->nSlices, 8, hipMemcpyDeviceToHost));
    SELL_TRY(hipMalloc(&f->val, std::max<uint64_t>(total, 1) * 8));
    SELL_TRY(hipMalloc(&f->col, std::max<uint64_t>(total, 1) * 4));
    if (f->nSlices)
        hipLaunchKernelGGL((sell_fill_kernel<I>), grid2d(((uint64_t)f->nSlices + 3) / 4, 256), dim3(256), 0, nullptr, f->nSlices,
                           f->sliceOff, f->perm, f->slen, IRP, d->JA, d->AS, f->val, f->col);
    // long rows
    Tmp counter;
    if (counter.alloc(4)) return EXIT_FAILURE;
    SELL_TRY(hipMemsetAsync(counter.p, 0, 4, nullptr));
    SELL_TRY(hipMalloc(&f->longRows, std::max<uint64_t>(std::min<uint64_t>(M, d->NZ / SELL_MAX_ROW + 1), 1) * 4));
    if (M) {
        hipLaunchKernelGGL((sell_longlist_kernel<I>), grid2d((M + 255) / 256, 256), dim3(256), 0, nullptr, M, IRP, f->longRows, counter.as<uint32_t>());
        hipLaunchKernelGGL((sell_flag_long_kernel<I>), gRows, dim3(256), 0, nullptr, padded, IRP, f->perm);
    }
    SELL_TRY(hipGetLastError());
    SELL_TRY(hipMemcpy(&f->nLong, counter.p, 4, hipMemcpyDeviceToHost));
    SELL_TRY(hipDeviceSynchronize());
    f->bytes = total * 12 + padded * 8 + ((size_t)f->nSlices + 1) * 8 + (size_t)f->nLong * 4;
    return EXIT_SUCCESS;
}

}  // namespace

void freeSell(SellFormat* f) {
    if (!f) return;
    (void)hipFree(f->sliceOff); (void)hipFree(f->perm); (void)hipFree(f->slen); (void)hipFree(f->val); (void)hipFree(f->col);
    (void)hipFree(f->longRows);
    delete f;
}

int buildSell(DevMat* d) {
    if (d->sell) return EXIT_SUCCESS;
    if (d->kind != Kind::CSR) return EXIT_FAILURE;
    if (d->M >= 0x7FFFFFFFull) { fprintf(stderr, "libspmvhip: sell: more than 2^31 rows unsupported\n"); return EXIT_FAILURE; }
    SellFormat* f = new SellFormat;
    const int rc = d->irpBytes == 4 ? buildSellT<uint32_t>(d, f) : buildSellT<uint64_t>(d, f);
    if (rc) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: sell: format build failed\n"); freeSell(f); return EXIT_FAILURE; }
    d->sell = f;
    return EXIT_SUCCESS;
}

size_t sellBytes(const DevMat* d) { return d->sell ? d->sell->bytes : 0; }

int enqueueSell(DevMat* d, const double* x, double* y, hipStream_t stream) {
    SellFormat* f = d->sell;
    if (!f) return EXIT_FAILURE;
    if (f->nSlices)
        hipLaunchKernelGGL(sell_spmv_kernel, grid2d(((uint64_t)f->nSlices + 3) / 4, 256), dim3(256), 0, stream, f->nSlices,
                           f->sliceOff, f->perm, f->slen, f->val, f->col, x, y);
    if (f->nLong) {
        if (d->irpBytes == 4)
            hipLaunchKernelGGL((sell_long_kernel<uint32_t>), grid2d(f->nLong, 256), dim3(256), 0, stream, f->nLong, f->longRows,
                               static_cast<const uint32_t*>(d->IRP), d->JA, d->AS, x, y);
        else
            hipLaunchKernelGGL((sell_long_kernel<uint64_t>), grid2d(f->nLong, 256), dim3(256), 0, stream, f->nLong, f->longRows,
                               static_cast<const uint64_t*>(d->IRP), d->JA, d->AS, x, y);
    }
    return hipGetLastError() == hipSuccess ? EXIT_SUCCESS : EXIT_FAILURE;
}

}  // namespace spmvhip
