// abi.hip -- the extern "C" boundary of libspmvhip.so (declared in include/spmvHip.h).
// Upload/free restate src/commons/cudaUtils.cu:20-98 + src/include/cudaUtils.h:70-78;
// the launchers replace `f<<<grid,block>>>(dMat,dVect,Conf,dOutV)` of
// src/main.cu:221-238 and test/SpMV_test.cu:103-146.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "spmvHip.h"
#include "kernels.hpp"

using namespace spmvhip;

namespace {

struct State {
    bool        inited = false;
    int         dev = 0;
    hipStream_t stream = nullptr;
    bool        sync = true;
    int         variantRowsCSR = 2;     // 0 scalar restatement, 1 LDS-stream kernel (sequential row sums), 2 the fastest serial-order
                                        // kernel for the matrix (LDS-stream / deterministic two-phase / deterministic stripes)
    int         variantWarpCSR = 2;     // 0 wavefront-per-row restatement, 1 LDS-stream kernel (LDS segmented reduction), 2 the fastest
                                        // reduction-order kernel for the matrix (LDS-stream / two-phase / stripes), measured at first use
    int         variantEllRowMajor = 1; // hipSpMVRowsELLNNTransposed: 0 a thread walks its row in global memory, 1 LDS-stream kernel, same sums
    int         ldsOrder = -1;          // lds_order_probe_kernel: -1 not run yet, 1 lane-ascending + in issue order, 0 anything else
    bool        ellRowLens = true;
    bool        unitValues = true;      // look for "every stored value is the same double" at upload (spmvHipSetUnitValues)
    double      lastSeconds = 0;
    spmvDim3    lastGrid{0, 0, 0}, lastBlock{0, 0, 0};
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
} S;

#define ERR(...) do { fprintf(stderr, "\33[31m\33[1m\33[44mlibspmvhip: "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\33[0m\n"); } while (0)

bool ready(const char* who) {
    if (S.inited) return true;
    ERR("%s: spmvHipInit() has not been called", who);
    return false;
}

DevMat* descOf(spmat* h, const char* who) {
    if (!h || !h->dev || h->dev == SPMAT_TAG_ELL_TRANSPOSED) {
        ERR("%s: not a device handle (upload with spMatCpyCSR/spMatCpyELL first)", who);
        return nullptr;
    }
    DevMat* d = static_cast<DevMat*>(h->dev);
    if (d->magic != 0x53504D56) { ERR("%s: corrupted device handle", who); return nullptr; }
    return d;
}

// ... for a launcher: the vectors are raw device pointers whose extent the library cannot know, but a NULL one would be
// dereferenced by every lane of the kernel -- a GPU page fault, which on a shared node is everybody's problem
DevMat* descOf(spmat* h, const double* x, const double* y, const char* who) {
    DevMat* d = descOf(h, who);
    if (d && (!x || !y)) { ERR("%s: %s is NULL", who, !x ? "x" : "y"); return nullptr; }
    return d;
}

// Block table of csr_stream2_kernel: rows packed while nnz <= STREAM_NNZ and rows <= STREAM2_MAX_ROWS;
// a longer row is a block of its own, flagged, and all such blocks come first (longest first) so that
// their serial tails overlap the rest of the grid.
template <typename I>
int buildRowBlocks2(DevMat* d, const I* IRP, uint64_t M) {
    std::vector<uint4> info, longs;
    std::vector<uint64_t> base, longBase;
    info.reserve(M / 64 + 2); base.reserve(M / 64 + 2);
    d->maxRowNnz = 0;
    for (uint64_t i = 0; i < M; ++i) d->maxRowNnz = std::max<uint64_t>(d->maxRowNnz, (uint64_t)IRP[i + 1] - (uint64_t)IRP[i]);
    uint64_t r = 0;
    while (r < M) {
        const uint64_t start = IRP[r];
        uint64_t e = r;
        while (e < M && (uint64_t)IRP[e + 1] - start <= (uint64_t)STREAM_NNZ && e - r < STREAM2_MAX_ROWS) ++e;
        if (e == r) {
            const uint64_t len = (uint64_t)IRP[r + 1] - start;
            if (len >= (1ull << 32)) { ERR("a single row with %lu entries is not supported", (unsigned long)len); return EXIT_FAILURE; }
            longs.push_back(make_uint4((uint32_t)r, 1u, (uint32_t)len, 1u));
            longBase.push_back(start);
            e = r + 1;
        } else {
            info.push_back(make_uint4((uint32_t)r, (uint32_t)(e - r), (uint32_t)((uint64_t)IRP[e] - start), 0u));
            base.push_back(start);
        }
        r = e;
    }
    std::vector<size_t> order(longs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return longs[a].z > longs[b].z; });
    std::vector<uint4> allInfo; std::vector<uint64_t> allBase;
    allInfo.reserve(longs.size() + info.size()); allBase.reserve(longs.size() + info.size());
    for (size_t i : order) { allInfo.push_back(longs[i]); allBase.push_back(longBase[i]); }
    allInfo.insert(allInfo.end(), info.begin(), info.end());
    allBase.insert(allBase.end(), base.begin(), base.end());
    d->nBlk2 = (uint32_t)allInfo.size();
    d->nLong2 = (uint32_t)longs.size();
    HIP_TRY(hipMalloc(&d->blkInfo, std::max<size_t>(allInfo.size(), 1) * sizeof(uint4)));
    HIP_TRY(hipMalloc(&d->blkBase, std::max<size_t>(allBase.size(), 1) * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(d->blkInfo, allInfo.data(), allInfo.size() * sizeof(uint4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->blkBase, allBase.data(), allBase.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    return EXIT_SUCCESS;
}

void freeDesc(DevMat* d) {
    if (!d) return;
    if (d->owns) {
        (void)hipFree(d->IRP); (void)hipFree(d->JA); (void)hipFree(d->AS); (void)hipFree(d->RL);
    }
    (void)hipFree(d->blkInfo); (void)hipFree(d->blkBase);
    freeTiles(d->tiles); freeTiles(d->tilesAlt);
    freeSell(d->sell);
    freeStripes(d->stripes); freeStripes(d->stripesAlt);
    d->magic = 0;
    delete d;
}

void publish(spmat* h, DevMat* d, ulong M, ulong N, ulong NZ, ulong maxRowNz) {
    memset(h, 0, sizeof *h);
    h->M = M; h->N = N; h->NZ = NZ; h->MAX_ROW_NZ = maxRowNz;
    h->JA = reinterpret_cast<ulong*>(d->JA);
    h->AS = d->AS;
    h->IRP = reinterpret_cast<ulong*>(d->IRP);
    h->RL = reinterpret_cast<ulong*>(d->RL);
    h->pitchJA = h->pitchAS = d->pitch;
    h->dev = d;
}

// timing bracket used by every launcher
struct Launch {
    bool timed;
    Launch(dim3 grid, dim3 block) : timed(S.sync) {
        (void)hipGetLastError();                    // finish() judges THIS launch, not whatever failed before it
        S.lastGrid = {grid.x, grid.y, grid.z};
        S.lastBlock = {block.x, block.y, block.z};
        if (timed) (void)hipEventRecord(S.ev0, S.stream);
    }
    void shape(dim3 grid, dim3 block) { S.lastGrid = {grid.x, grid.y, grid.z}; S.lastBlock = {block.x, block.y, block.z}; }
    int finish(const char* who) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ERR("%s: launch failed: %s", who, hipGetErrorString(e)); return EXIT_FAILURE; }
        if (!timed) return EXIT_SUCCESS;
        (void)hipEventRecord(S.ev1, S.stream);
        e = hipEventSynchronize(S.ev1);
        if (e != hipSuccess) { ERR("%s: kernel failed: %s", who, hipGetErrorString(e)); return EXIT_FAILURE; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, S.ev0, S.ev1);
        S.lastSeconds = ms * 1e-3;
        return EXIT_SUCCESS;
    }
};

// a launcher call that has nothing to launch (no rows, or no entries: y = 0): no stale time or shape is left behind
int nothingToLaunch(DevMat* d, double* dY) {
    S.lastSeconds = 0;
    S.lastGrid = {0, 0, 0}; S.lastBlock = {0, 0, 0};
    if (d->M && dY) {
        HIP_TRY(hipMemsetAsync(dY, 0, d->M * sizeof(double), S.stream));
        if (S.sync) HIP_TRY(hipStreamSynchronize(S.stream));
    }
    return EXIT_SUCCESS;
}

unsigned blockThreads(const CONFIG& cfg, unsigned dflt, unsigned maxThreads) {
    unsigned t = cfg.blockSize.x * std::max(1u, cfg.blockSize.y) * std::max(1u, cfg.blockSize.z);
    if (cfg.blockSize.x == 0) return dflt;
    if (t % WAVE != 0 || t > maxThreads) {
        ERR("CONFIG.blockSize %ux%ux%u is not a multiple of the 64-lane wavefront (or exceeds %u): using %u",
            cfg.blockSize.x, cfg.blockSize.y, cfg.blockSize.z, maxThreads, dflt);
        return dflt;
    }
    return t;
}

template <typename T>
int narrowUpload(T** dDst, const ulong* hSrc, size_t n, ulong limit, const char* what) {
    std::vector<T> tmp(n);
    for (size_t i = 0; i < n; ++i) {
        if (hSrc[i] > limit) { ERR("%s[%zu] = %lu does not fit the device index width", what, i, hSrc[i]); return EXIT_FAILURE; }
        tmp[i] = (T)hSrc[i];
    }
    HIP_TRY(hipMalloc(dDst, std::max<size_t>(n, 1) * sizeof(T)));
    HIP_TRY(hipMemcpy(*dDst, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return EXIT_SUCCESS;
}

__global__ void fill64_kernel(uint64_t* p, size_t n, uint64_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// device-side CSR -> ELL (row-major [rows][pitch] or column-major [slots][pitch]); one wavefront per row
template <typename I>
__global__ __launch_bounds__(256) void csr_to_ell_kernel(uint32_t M, uint32_t K, size_t pitch, int colMajor,
                                                         const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                         const double* __restrict__ AS, uint32_t* __restrict__ EJ,
                                                         double* __restrict__ EA, uint32_t* __restrict__ RL) {
    const uint64_t row = linear_block() * 4 + threadIdx.x / 64;
    if (row >= M) return;
    const uint32_t lane = threadIdx.x % 64;
    const I b = IRP[row];
    const uint32_t len = (uint32_t)(IRP[row + 1] - b);
    if (lane == 0) RL[row] = len;
    for (uint32_t c = lane; c < K; c += 64) {
        const size_t at = colMajor ? (size_t)c * pitch + row : (size_t)row * pitch + c;
        EJ[at] = c < len ? JA[b + c] : 0u;          // padding {0, 0.0} like the loader's calloc
        EA[at] = c < len ? AS[b + c] : 0.0;
    }
}

// What the deterministic format kernels rest on is measured behaviour of the LDS, not something the ISA manual promises:
// (a) lanes of ONE ds_add_f64 instruction that meet in an address are added in ascending lane order, (b) the LDS operations
// of one wavefront execute in issue order.  This probe checks both against sums computed on the host in that order (the
// values are chosen so that other orders give other bits); the serial-order selection offers the format kernels only
// where it passes, so the bitwise contract of hipSpMVRowsCSR never depends on the property silently.
// (The accumulator of every add is read from memory: with an address the compiler can prove wavefront-uniform its atomic
// optimizer would replace the 64 adds by a wavefront reduction and ONE add -- another summation order, and not what the
// SpMV kernels, whose addresses are per-lane rows, execute.)
__global__ __launch_bounds__(64) void lds_order_probe_kernel(const double* __restrict__ v, const uint32_t* __restrict__ slot,
                                                             double* __restrict__ out) {
    __shared__ double acc[4];
    const uint32_t lane = threadIdx.x;
    if (lane < 4) acc[lane] = 0.0;
    __syncthreads();
    atomicAdd(&acc[slot[lane]], v[lane]);                    // 64 lanes, one address
    atomicAdd(&acc[slot[64 + lane]], v[64 + lane]);          // two addresses, 32 lanes each
    atomicAdd(&acc[slot[128 + lane]], v[128 + lane]);        // a second instruction into the first address: behind the first one
    if (slot[192 + lane] < 4) atomicAdd(&acc[slot[192 + lane]], v[192 + lane]);   // a sparse lane mask
    __syncthreads();
    if (lane < 4) out[lane] = acc[lane];
}

// one wavefront per row: does any row hold a column smaller than its predecessor?  (the deterministic format kernels add a
// row's products in ascending COLUMN order, which is the serial oracle's ascending-j order only for such rows)
template <typename I>
__global__ __launch_bounds__(256) void csr_unsorted_kernel(uint64_t M, const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                           uint32_t* __restrict__ flag) {
    const uint64_t r = linear_block() * 4 + threadIdx.x / 64;
    if (r >= M) return;
    const uint64_t b = IRP[r], e = IRP[r + 1];
    bool bad = false;
    for (uint64_t j = b + threadIdx.x % 64; j + 1 < e; j += 64) bad |= JA[j] > JA[j + 1];
    if (bad) atomicOr(flag, 1u);
}

template <bool SEQ, bool UNIT>
static void launchStream2T(DevMat* d, double* x, double* y) {
    if (d->irpBytes == 4)
        hipLaunchKernelGGL((csr_stream2_kernel<uint32_t, SEQ, UNIT>), grid2d(d->nBlk2, WG_THREADS), dim3(WG_THREADS), 0, S.stream,
                           d->nBlk2, d->nLong2, d->blkInfo, d->blkBase, static_cast<const uint32_t*>(d->IRP), d->JA, d->AS, d->unitValue, x, y);
    else
        hipLaunchKernelGGL((csr_stream2_kernel<uint64_t, SEQ, UNIT>), grid2d(d->nBlk2, WG_THREADS), dim3(WG_THREADS), 0, S.stream,
                           d->nBlk2, d->nLong2, d->blkInfo, d->blkBase, static_cast<const uint64_t*>(d->IRP), d->JA, d->AS, d->unitValue, x, y);
}
template <bool SEQ>
static void launchStream2(DevMat* d, double* x, double* y) {
    if (d->unit) launchStream2T<SEQ, true>(d, x, y); else launchStream2T<SEQ, false>(d, x, y);
}

// 1 in *flag unless every value has the bit pattern `first` (bit patterns: -0.0 and 0.0, or two NaNs, are different values here)
__global__ __launch_bounds__(256) void values_differ_kernel(const uint64_t* __restrict__ AS, uint64_t n, uint64_t first, uint32_t* __restrict__ flag) {
    bool differ = false;
    for (uint64_t j = linear_block() * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * gridDim.y * 256) differ |= AS[j] != first;
    if (differ) atomicOr(flag, 1u);
}

// ELL: every REAL cell (slot < RL[row]) has the bit pattern `first`; padding cells are not looked at
__global__ __launch_bounds__(256) void ell_values_differ_kernel(uint64_t rows, size_t pitch, int colMajor, const uint64_t* __restrict__ AS,
                                                                const uint32_t* __restrict__ RL, uint64_t first, uint32_t* __restrict__ flag) {
    const uint64_t r = linear_block() * 256 + threadIdx.x;
    if (r >= rows) return;
    bool differ = false;
    for (uint32_t i = 0, n = RL[r]; i < n; ++i) differ |= AS[colMajor ? r + (uint64_t)i * pitch : r * pitch + i] != first;
    if (differ) atomicOr(flag, 1u);
}

// ... for an ELL upload that has row lengths; `first` = the bits of any real cell (found on the host side of the upload)
static int detectUnitValuesEll(DevMat* d, bool anyCell, uint64_t first) {
    d->unit = false;
    if (!S.unitValues || !anyCell || !d->RL || !d->AS || d->M == 0) return EXIT_SUCCESS;
    uint32_t differ = 1, *dFlag = nullptr;
    HIP_TRY(hipMalloc(&dFlag, 4));
    int rc = EXIT_FAILURE;
    if (hipOk(hipMemset(dFlag, 0, 4), "hipMemset")) {
        hipLaunchKernelGGL(ell_values_differ_kernel, grid2d((d->M + 255) / 256, 256), dim3(256), 0, nullptr, d->M, d->pitch,
                           d->kind == Kind::ELL_COLMAJOR ? 1 : 0, reinterpret_cast<const uint64_t*>(d->AS), d->RL, first, dFlag);
        if (hipOk(hipGetLastError(), "ell_values_differ_kernel") && hipOk(hipMemcpy(&differ, dFlag, 4, hipMemcpyDeviceToHost), "hipMemcpy")) rc = EXIT_SUCCESS;
    }
    (void)hipFree(dFlag);
    if (rc == EXIT_SUCCESS && !differ) { d->unit = true; memcpy(&d->unitValue, &first, 8); }
    return rc;
}

// sets d->unit / d->unitValue from the uploaded values (one pass over AS at upload; off with spmvHipSetUnitValues(0))
static int detectUnitValues(DevMat* d) {
    d->unit = false;
    if (!S.unitValues || d->NZ == 0 || !d->AS) return EXIT_SUCCESS;
    uint64_t first = 0;
    uint32_t differ = 1, *dFlag = nullptr;
    HIP_TRY(hipMemcpy(&first, d->AS, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMalloc(&dFlag, 4));
    int rc = EXIT_FAILURE;
    if (hipOk(hipMemset(dFlag, 0, 4), "hipMemset")) {
        const uint64_t blocks = std::min<uint64_t>((d->NZ + 255) / 256, 256 * 64);
        hipLaunchKernelGGL(values_differ_kernel, grid2d(blocks, 256), dim3(256), 0, nullptr, reinterpret_cast<const uint64_t*>(d->AS), d->NZ, first, dFlag);
        if (hipOk(hipGetLastError(), "values_differ_kernel") && hipOk(hipMemcpy(&differ, dFlag, 4, hipMemcpyDeviceToHost), "hipMemcpy")) rc = EXIT_SUCCESS;
    }
    (void)hipFree(dFlag);
    if (rc == EXIT_SUCCESS && !differ) { d->unit = true; memcpy(&d->unitValue, &first, 8); }
    return rc;
}

// row-major ELL through the LDS-stream kernel (a row must fit a block): seq = one thread sums a row in ascending slots
static int launchEllStream(DevMat* d, bool rl, bool seq, double* x, double* y, const char* who) {
    const uint32_t rowsPerBlk = std::min<uint32_t>((uint32_t)(STREAM_NNZ / d->pitch), WG_THREADS);
    const uint64_t nBlk = (d->M + rowsPerBlk - 1) / rowsPerBlk;
    const dim3 grid = grid2d(nBlk, WG_THREADS), block(WG_THREADS);
    Launch L(grid, block);
#define ELL_STREAM(RLV, SEQV, ...) hipLaunchKernelGGL((ell_stream_kernel<RLV, SEQV, ##__VA_ARGS__>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, \
                                                      (uint32_t)d->pitch, rowsPerBlk, nBlk, d->JA, d->AS, d->RL, x, y, d->unitValue)
    if (rl && d->unit) { if (seq) ELL_STREAM(true, true, true); else ELL_STREAM(true, false, true); }
    else if (rl) { if (seq) ELL_STREAM(true, true); else ELL_STREAM(true, false); }
    else    { if (seq) ELL_STREAM(false, true); else ELL_STREAM(false, false); }
#undef ELL_STREAM
    return L.finish(who);
}

template <int G>
static void launchEllGroup(DevMat* d, bool rl, dim3 grid, dim3 block, double* x, double* y) {
    if (rl) hipLaunchKernelGGL((ell_rowmajor_group<true, G>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, x, y);
    else    hipLaunchKernelGGL((ell_rowmajor_group<false, G>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, x, y);
}


}  // namespace

// row pointers must start at 0, never decrease and end at NZ: the kernels trust them for every AS/JA access
template <typename I>
static bool rowPointersOk(const I* IRP, uint64_t M, uint64_t NZ, const char* who) {
    if ((uint64_t)IRP[0] != 0 || (uint64_t)IRP[M] != NZ) {
        ERR("%s: inconsistent row pointers (IRP[0]=%lu IRP[M]=%lu NZ=%lu)", who, (unsigned long)IRP[0], (unsigned long)IRP[M], (unsigned long)NZ);
        return false;
    }
    for (uint64_t r = 0; r < M; ++r)
        if (IRP[r] > IRP[r + 1]) { ERR("%s: row pointers decrease at row %lu (%lu > %lu)", who, (unsigned long)r, (unsigned long)IRP[r], (unsigned long)IRP[r + 1]); return false; }
    return true;
}

// Upload an (nRows x nCols) row-major host array pair with a padded pitch.
static int uploadPitched(DevMat* d, const ulong* hJA, const double* hAS, size_t nRows, size_t nCols,
                         size_t pitch, ulong colLimit) {
    const size_t total = std::max<size_t>(nRows * pitch, 1);
    std::vector<uint32_t> ja(total, 0u);
    std::vector<double>   as(total, 0.0);
    for (size_t r = 0; r < nRows; ++r)
        for (size_t c = 0; c < nCols; ++c) {
            const ulong col = hJA[r * nCols + c];
            if (col > colLimit) { ERR("spMatCpyELL: column id %lu out of range", col); return EXIT_FAILURE; }
            ja[r * pitch + c] = (uint32_t)col;
            as[r * pitch + c] = hAS[r * nCols + c];
        }
    HIP_TRY(hipMalloc(&d->JA, total * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d->AS, total * sizeof(double)));
    HIP_TRY(hipMemcpy(d->JA, ja.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->AS, as.data(), total * sizeof(double), hipMemcpyHostToDevice));
    d->pitch = pitch;
    return EXIT_SUCCESS;
}

static int ellUpload(spmat* m, spmat* dst, bool transposed) {
    if (!ready("spMatCpyELL") || !m || !dst) return EXIT_FAILURE;
    if (!m->JA || !m->AS) { ERR("spMatCpyELL: host matrix has no ELL arrays"); return EXIT_FAILURE; }
    // reference field convention: a transposed matrix keeps slots in M and rows in MAX_ROW_NZ / N
    const ulong rows  = transposed ? m->MAX_ROW_NZ : m->M;
    const ulong slots = transposed ? m->M : m->MAX_ROW_NZ;
    // a transposed struct has lost the column count to the reference's field swap; this repo's ellTranspose (and
    // api.HostELL.transpose) keep it in the unused host field pitchJA -- 0 = unknown, column ids then cannot be checked
    const ulong cols  = transposed ? (ulong)m->pitchJA : m->N;
    if (rows >= (1ull << 32) - 1 || slots >= (1ull << 32) - 1) { ERR("spMatCpyELL: dimensions exceed 32-bit ids"); return EXIT_FAILURE; }
    DevMat* d = new DevMat;
    d->kind = transposed ? Kind::ELL_COLMAJOR : Kind::ELL_ROWMAJOR;
    d->M = rows; d->N = cols; d->NZ = m->NZ; d->K = slots;
    int rc;
    const ulong colLimit = transposed ? (cols ? cols - 1 : 0xFFFFFFFFul) : (m->N ? m->N - 1 : 0);
    if (transposed) rc = uploadPitched(d, m->JA, m->AS, slots, rows, (rows + 63) / 64 * 64, colLimit);
    else            rc = uploadPitched(d, m->JA, m->AS, rows, slots, (slots + 1) / 2 * 2, colLimit);   // (rows stay 16-B aligned; a wider pitch is only padding to stream)
    if (!rc && m->RL) rc = narrowUpload<uint32_t>(&d->RL, m->RL, rows, slots, "RL");
    if (!rc && m->RL && S.unitValues) {                // first real cell on the host: slot 0 of the first non-empty row
        ulong r = 0;
        while (r < rows && m->RL[r] == 0) ++r;
        uint64_t first = 0;
        if (r < rows) memcpy(&first, &m->AS[transposed ? r : r * slots], 8);      // host layouts: slot-major (pitch = rows) / row-major (pitch = slots)
        rc = detectUnitValuesEll(d, r < rows, first);
    }
    if (rc) { freeDesc(d); return EXIT_FAILURE; }
    publish(dst, d, m->M, m->N, m->NZ, m->MAX_ROW_NZ);
    return EXIT_SUCCESS;
}

namespace {
// Device copy of a host matrix behind the SPMV_INTERF-style wrappers.  The key is the host struct's address; the
// entry also remembers the shape and the array pointers it was uploaded from, and is re-uploaded when any of them
// differs (a freed struct whose address malloc handed out again, a matrix re-loaded in place).  Changing the VALUES
// inside the same arrays is invisible here: call spmvHipDropCache() before a cached host matrix is modified or freed.
struct Cached {
    spmat handle{}; double* dx = nullptr; double* dy = nullptr; int kind = 0;
    ulong M = 0, N = 0, NZ = 0, K = 0; const void *irp = nullptr, *ja = nullptr, *as = nullptr, *rl = nullptr;
    bool sameSource(const spmat* m, int k) const {
        return kind == k && M == m->M && N == m->N && NZ == m->NZ && K == m->MAX_ROW_NZ && irp == m->IRP && ja == m->JA && as == m->AS && rl == m->RL;
    }
    void release() { hipFreeSpmat(&handle); (void)hipFree(dx); (void)hipFree(dy); dx = dy = nullptr; }
};
std::map<const spmat*, Cached> g_cache;

int hostCall(spmat* mat, double* x, CONFIG* cfg, double* y, int kind, SPMV_HIP_INTERF fn) {
    if (!ready("spmvHip*") || !mat || !x || !y) return EXIT_FAILURE;
    auto it = g_cache.find(mat);
    if (it != g_cache.end() && !it->second.sameSource(mat, kind)) { it->second.release(); g_cache.erase(it); it = g_cache.end(); }
    if (it == g_cache.end()) {
        Cached c; c.kind = kind;
        c.M = mat->M; c.N = mat->N; c.NZ = mat->NZ; c.K = mat->MAX_ROW_NZ; c.irp = mat->IRP; c.ja = mat->JA; c.as = mat->AS; c.rl = mat->RL;
        int rc;
        if (kind == 0) rc = spMatCpyCSR(mat, &c.handle);
        else if (kind == 1) rc = spMatCpyELL(mat, &c.handle);
        else {  // column-major ELL: transposition is done on the fly from the row-major host matrix
            if (!mat->JA || !mat->AS) { ERR("spmvHipRowsELL: host matrix has no ELL arrays"); return EXIT_FAILURE; }
            spmat t = *mat;
            std::vector<ulong> ja(mat->M * mat->MAX_ROW_NZ);
            std::vector<double> as(mat->M * mat->MAX_ROW_NZ);
            for (ulong r = 0; r < mat->M; ++r)
                for (ulong c2 = 0; c2 < mat->MAX_ROW_NZ; ++c2) {
                    ja[c2 * mat->M + r] = mat->JA[r * mat->MAX_ROW_NZ + c2];
                    as[c2 * mat->M + r] = mat->AS[r * mat->MAX_ROW_NZ + c2];
                }
            t.JA = ja.data(); t.AS = as.data();
            t.M = mat->MAX_ROW_NZ; t.N = mat->M; t.MAX_ROW_NZ = mat->M;
            t.pitchJA = mat->N;                      // column count for the upload's range check
            rc = spMatCpyELLTransposed(&t, &c.handle);
        }
        if (rc) return EXIT_FAILURE;
        if (spmvHipVecAlloc(&c.dx, mat->N) || spmvHipVecAlloc(&c.dy, mat->M)) { c.release(); return EXIT_FAILURE; }
        it = g_cache.emplace(mat, c).first;
    }
    Cached& c = it->second;
    const bool wasSync = S.sync;
    S.sync = true;
    int rc = spmvHipVecUp(c.dx, x, mat->N);
    if (!rc) rc = spmvHipVecFill(c.dy, mat->M, 0x7FF8DEADDEADDEADull);     // poison y
    if (!rc) rc = fn(&c.handle, c.dx, cfg ? *cfg : CONFIG{}, c.dy);
    if (!rc) rc = spmvHipVecDown(y, c.dy, mat->M);
    S.sync = wasSync;
    return rc;
}
}  // namespace


namespace spmvhip { hipStream_t libraryStream() { return S.stream; } }

extern "C" {

// ------------------------------------------------------------------------ lifecycle
int spmvHipDeviceCount(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int spmvHipInit(int dev, size_t sizeofSpmat, size_t sizeofConfig) {
    if (sizeofSpmat != sizeof(spmat) || sizeofConfig != sizeof(CONFIG)) {
        ERR("spmvHipInit: layout mismatch: caller spmat/CONFIG = %zu/%zu bytes, library %zu/%zu "
            "(compile the host with this repo's include/spmv_types.h)",
            sizeofSpmat, sizeofConfig, sizeof(spmat), sizeof(CONFIG));
        return EXIT_FAILURE;
    }
    int n = spmvHipDeviceCount();
    if (n <= 0) { ERR("spmvHipInit: no HIP device visible -- this library has no CPU fallback"); return EXIT_FAILURE; }
    if (dev < 0 || dev >= n) { ERR("spmvHipInit: device %d out of range [0,%d)", dev, n); return EXIT_FAILURE; }
    HIP_TRY(hipSetDevice(dev));
    if (!S.ev0) HIP_TRY(hipEventCreate(&S.ev0));
    if (!S.ev1) HIP_TRY(hipEventCreate(&S.ev1));
    S.dev = dev;
    S.inited = true;
    return EXIT_SUCCESS;
}

// the push kernel's side stream (hipSpMVTilesReducePush) with its fork / join events: made on first use, on the device of
// that moment, and given back by spmvHipFinalize()
namespace { hipStream_t g_pushSide = nullptr; hipEvent_t g_pushFork = nullptr, g_pushJoin = nullptr; bool g_pushPending = false; }

int spmvHipFinalize(void) {
    if (!S.inited) return EXIT_SUCCESS;
    spmvHipDropCache();
    freeTilesWorkspace();
    peerFinalize();
    if (g_pushSide) {
        (void)hipStreamSynchronize(g_pushSide);
        (void)hipStreamDestroy(g_pushSide);
        (void)hipEventDestroy(g_pushFork);
        (void)hipEventDestroy(g_pushJoin);
        g_pushSide = nullptr; g_pushFork = g_pushJoin = nullptr; g_pushPending = false;
    }
    if (S.ev0) (void)hipEventDestroy(S.ev0);
    if (S.ev1) (void)hipEventDestroy(S.ev1);
    S.ev0 = S.ev1 = nullptr;
    S.inited = false;
    return EXIT_SUCCESS;
}

int spmvHipSetStream(void* stream) { S.stream = static_cast<hipStream_t>(stream); return EXIT_SUCCESS; }
int spmvHipSetSync(int sync) { S.sync = sync != 0; return EXIT_SUCCESS; }
double spmvHipLastKernelSeconds(void) { return S.lastSeconds; }
int spmvHipLastLaunch(spmvDim3* grid, spmvDim3* block) {
    if (grid) *grid = S.lastGrid;
    if (block) *block = S.lastBlock;
    return EXIT_SUCCESS;
}
int spmvHipDeviceSynchronize(void) { HIP_TRY(hipDeviceSynchronize()); return EXIT_SUCCESS; }

int spmvHipProbeLdsAtomicOrder(void) {
    if (!ready("spmvHipProbeLdsAtomicOrder")) return -1;
    if (S.ldsOrder >= 0) return S.ldsOrder;
    double h[256], expect[4] = {0, 0, 0, 0}, got[4] = {0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) h[i] = (i % 2 ? -1.0 : 1.0) / (3.0 + 7.0 * i) + (i % 5) * 0x1.0p-20;
    volatile double a0 = 0, a1 = 0, a2 = 0, a3 = 0;  // (volatile: plain sequential adds, whatever the host compiler would like to do)
    for (int l = 0; l < 64; ++l) a0 = a0 + h[l];
    for (int l = 0; l < 64; ++l) { if (l & 1) a2 = a2 + h[64 + l]; else a1 = a1 + h[64 + l]; }
    for (int l = 0; l < 64; ++l) a0 = a0 + h[128 + l];
    for (int l = 0; l < 64; l += 3) a3 = a3 + h[192 + l];
    expect[0] = a0; expect[1] = a1; expect[2] = a2; expect[3] = a3;
    uint32_t slot[256];
    for (int l = 0; l < 64; ++l) { slot[l] = 0; slot[64 + l] = 1 + (l & 1); slot[128 + l] = 0; slot[192 + l] = l % 3 == 0 ? 3 : 99; }
    double *dV = nullptr, *dOut = nullptr;
    uint32_t* dSlot = nullptr;
    int ok = 0;
    if (hipMalloc(&dV, sizeof h) == hipSuccess && hipMalloc(&dOut, sizeof got) == hipSuccess && hipMalloc(&dSlot, sizeof slot) == hipSuccess &&
        hipMemcpy(dV, h, sizeof h, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dSlot, slot, sizeof slot, hipMemcpyHostToDevice) == hipSuccess) {
        ok = 1;
        for (int rep = 0; rep < 4 && ok; ++rep) {    // a few launches: the answer must not depend on timing
            hipLaunchKernelGGL(lds_order_probe_kernel, dim3(1), dim3(64), 0, S.stream, dV, dSlot, dOut);
            if (hipStreamSynchronize(S.stream) != hipSuccess || hipMemcpy(got, dOut, sizeof got, hipMemcpyDeviceToHost) != hipSuccess) { ok = 0; break; }
            ok = memcmp(got, expect, sizeof got) == 0;
        }
    }
    (void)hipFree(dV); (void)hipFree(dOut); (void)hipFree(dSlot);
    (void)hipGetLastError();
    S.ldsOrder = ok;
    return ok;
}

int spmvHipSetVariant(const char* launcher, int variant) {
    if (!launcher) return EXIT_FAILURE;
    if (!strcmp(launcher, "hipSpMVRowsCSR") && variant >= 0 && variant <= 2) { S.variantRowsCSR = variant; return EXIT_SUCCESS; }
    if (!strcmp(launcher, "hipSpMVWarpPerRowCSR") && variant >= 0 && variant <= 2) { S.variantWarpCSR = variant; return EXIT_SUCCESS; }
    if (!strcmp(launcher, "hipSpMVRowsELLNNTransposed") && variant >= 0 && variant <= 1) { S.variantEllRowMajor = variant; return EXIT_SUCCESS; }
    ERR("spmvHipSetVariant: unknown (%s, %d)", launcher, variant);
    return EXIT_FAILURE;
}
int spmvHipSetEllRowLens(int use) { S.ellRowLens = use != 0; return EXIT_SUCCESS; }
int spmvHipSetUnitValues(int on) { S.unitValues = on != 0; return EXIT_SUCCESS; }
int spmvHipUnitValue(spmat* dMat, double* value) {
    DevMat* d = descOf(dMat, "spmvHipUnitValue");
    if (!d) return -1;
    if (d->unit && value) *value = d->unitValue;
    return d->unit ? 1 : 0;
}

// ------------------------------------------------------------------------ vectors
int spmvHipVecAlloc(double** dVec, size_t n) {
    if (!ready("spmvHipVecAlloc") || !dVec) return EXIT_FAILURE;
    HIP_TRY(hipMalloc(dVec, std::max<size_t>(n, 1) * sizeof(double)));
    return EXIT_SUCCESS;
}
int spmvHipVecFree(double* dVec) { HIP_TRY(hipFree(dVec)); return EXIT_SUCCESS; }
int spmvHipVecUp(double* dVec, const double* hVec, size_t n) {
    HIP_TRY(hipMemcpyAsync(dVec, hVec, n * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}
int spmvHipVecDown(double* hVec, const double* dVec, size_t n) {
    HIP_TRY(hipMemcpyAsync(hVec, dVec, n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}
int spmvHipVecFill(double* dVec, size_t n, uint64_t pattern) {
    if (n == 0) return EXIT_SUCCESS;
    unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fill64_kernel, dim3(grid), dim3(256), 0, S.stream, reinterpret_cast<uint64_t*>(dVec), n, pattern);
    HIP_TRY(hipGetLastError());
    if (S.sync) HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}

int spmvHipMalloc(void** dPtr, size_t bytes) {
    if (!ready("spmvHipMalloc") || !dPtr) return EXIT_FAILURE;
    HIP_TRY(hipMalloc(dPtr, std::max<size_t>(bytes, 1)));
    return EXIT_SUCCESS;
}
int spmvHipFree(void* dPtr) { HIP_TRY(hipFree(dPtr)); return EXIT_SUCCESS; }
int spmvHipMemcpyUp(void* dDst, const void* hSrc, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dDst, hSrc, bytes, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}
int spmvHipMemcpyDown(void* hDst, const void* dSrc, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(hDst, dSrc, bytes, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}

// ------------------------------------------------------------------------ upload
int spMatCpyCSR(spmat* m, spmat* dst) {
    if (!ready("spMatCpyCSR") || !m || !dst) return EXIT_FAILURE;
    if (!m->IRP || (m->NZ && (!m->JA || !m->AS))) { ERR("spMatCpyCSR: host matrix has no CSR arrays"); return EXIT_FAILURE; }
    if (m->M >= (1ull << 32) - 1 || m->N > (1ull << 32)) { ERR("spMatCpyCSR: %lu x %lu exceeds the 32-bit row/column ids of the device format", m->M, m->N); return EXIT_FAILURE; }
    if (!rowPointersOk(m->IRP, m->M, m->NZ, "spMatCpyCSR")) return EXIT_FAILURE;
    DevMat* d = new DevMat;
    d->kind = Kind::CSR;
    d->M = m->M; d->N = m->N; d->NZ = m->NZ;
    d->irpBytes = m->NZ < IRP32_LIMIT ? 4 : 8;
    int rc = EXIT_SUCCESS;
    if (d->irpBytes == 4) rc = narrowUpload<uint32_t>(reinterpret_cast<uint32_t**>(&d->IRP), m->IRP, m->M + 1, 0xFFFFFFFFul, "IRP");
    else                  rc = narrowUpload<uint64_t>(reinterpret_cast<uint64_t**>(&d->IRP), m->IRP, m->M + 1, ~0ul, "IRP");
    if (!rc) rc = narrowUpload<uint32_t>(&d->JA, m->JA, m->NZ, m->N ? m->N - 1 : 0, "JA");
    if (!rc) {
        if (!hipOk(hipMalloc(&d->AS, std::max<size_t>(m->NZ, 1) * sizeof(double)), "hipMalloc AS") ||
            !hipOk(hipMemcpy(d->AS, m->AS, m->NZ * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy AS")) rc = EXIT_FAILURE;
    }
    if (!rc && m->RL) rc = narrowUpload<uint32_t>(&d->RL, m->RL, m->M, 0xFFFFFFFFul, "RL");
    if (!rc) rc = buildRowBlocks2(d, m->IRP, m->M);
    if (!rc) rc = detectUnitValues(d);
    if (rc) { freeDesc(d); return EXIT_FAILURE; }
    publish(dst, d, m->M, m->N, m->NZ, 0);
    return EXIT_SUCCESS;
}

int spmvHipAdoptCSR(spmat* dst, ulong M, ulong N, ulong NZ, const void* dIRP, int irpBytes,
                    const uint32_t* dJA, const double* dAS, const void* hIRP) {
    if (!ready("spmvHipAdoptCSR") || !dst || !dIRP) return EXIT_FAILURE;
    if (irpBytes != 4 && irpBytes != 8) { ERR("spmvHipAdoptCSR: irpBytes must be 4 or 8"); return EXIT_FAILURE; }
    if (irpBytes == 4 && NZ >= IRP32_LIMIT) { ERR("spmvHipAdoptCSR: NZ=%lu needs 64-bit row pointers", NZ); return EXIT_FAILURE; }
    if (M >= (1ull << 32) - 1 || N > (1ull << 32)) { ERR("spmvHipAdoptCSR: dimensions exceed 32-bit ids"); return EXIT_FAILURE; }
    std::vector<unsigned char> tmp;
    if (!hIRP) {
        tmp.resize((M + 1) * (size_t)irpBytes);
        HIP_TRY(hipMemcpy(tmp.data(), dIRP, tmp.size(), hipMemcpyDeviceToHost));
        hIRP = tmp.data();
    }
    if (irpBytes == 4 ? !rowPointersOk(static_cast<const uint32_t*>(hIRP), M, NZ, "spmvHipAdoptCSR")
                      : !rowPointersOk(static_cast<const uint64_t*>(hIRP), M, NZ, "spmvHipAdoptCSR")) return EXIT_FAILURE;
    DevMat* d = new DevMat;
    d->kind = Kind::CSR; d->owns = false;
    d->M = M; d->N = N; d->NZ = NZ; d->irpBytes = irpBytes;
    d->IRP = const_cast<void*>(dIRP); d->JA = const_cast<uint32_t*>(dJA); d->AS = const_cast<double*>(dAS);
    const int rc2 = irpBytes == 4 ? buildRowBlocks2(d, static_cast<const uint32_t*>(hIRP), M)
                                  : buildRowBlocks2(d, static_cast<const uint64_t*>(hIRP), M);
    if (rc2 || detectUnitValues(d)) { freeDesc(d); return EXIT_FAILURE; }
    publish(dst, d, M, N, NZ, 0);
    return EXIT_SUCCESS;
}

int spMatCpyELL(spmat* m, spmat* dst) { return ellUpload(m, dst, m && m->dev == SPMAT_TAG_ELL_TRANSPOSED); }
int spMatCpyELLTransposed(spmat* m, spmat* dst) { return ellUpload(m, dst, true); }

int spmvHipCsrToEll(spmat* dCsr, int transposed, spmat* dEll) {
    DevMat* c = descOf(dCsr, "spmvHipCsrToEll");
    if (!c || !dEll) return EXIT_FAILURE;
    if (c->kind != Kind::CSR) { ERR("spmvHipCsrToEll: source handle is not CSR"); return EXIT_FAILURE; }
    const uint64_t K = c->maxRowNnz, rows = c->M;
    DevMat* d = new DevMat;
    d->kind = transposed ? Kind::ELL_COLMAJOR : Kind::ELL_ROWMAJOR;
    d->M = rows; d->N = c->N; d->NZ = c->NZ; d->K = K;
    d->pitch = transposed ? (rows + 63) / 64 * 64 : (K + 1) / 2 * 2;
    const size_t cells = std::max<size_t>((transposed ? K : rows) * d->pitch, 1);
    {   // ELL size guard.  The reference's loader refuses an ELL copy whose 2*M*maxRow padded cells exceed a fixed host
        // budget (src/lib/parser.c:223-232, config.h:69-70: 6*2^27 cells); on the device the budget is what the GPU has
        // free right now -- the unclipped power-law matrix (10 M rows x 50 k slots = 6 TB) is refused here, before any
        // allocation, the same matrix clipped to 64 slots (7.7 GB) passes.
        size_t freeB = 0, totalB = 0;
        const unsigned __int128 need128 = (unsigned __int128)(transposed ? K : rows) * d->pitch * 12 + (unsigned __int128)rows * 4;
        const size_t need = need128 > (unsigned __int128)~(size_t)0 ? ~(size_t)0 : (size_t)need128;
        // best effort: what is free at this moment (other processes and cached pools count as used); when the query itself
        // fails the guard is skipped and hipMalloc decides
        const bool known = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
        if (!known) (void)hipGetLastError();
        if (known && need > freeB) {
            ERR("spmvHipCsrToEll: ELL copy of %lu rows x %lu slots needs %.1f GB, device has %.1f GB free: refused "
                "(the reference refuses above 6*2^27 padded cells, parser.c:223-232)", (unsigned long)rows, (unsigned long)K,
                (double)need * 1e-9, (double)freeB * 1e-9);
            delete d;
            return EXIT_FAILURE;
        }
    }
    if (!hipOk(hipMalloc(&d->JA, cells * sizeof(uint32_t)), "hipMalloc ELL JA") ||
        !hipOk(hipMalloc(&d->AS, cells * sizeof(double)), "hipMalloc ELL AS") ||
        !hipOk(hipMalloc(&d->RL, std::max<size_t>(rows, 1) * sizeof(uint32_t)), "hipMalloc ELL RL") ||
        !hipOk(hipMemsetAsync(d->JA, 0, cells * sizeof(uint32_t), S.stream), "memset") ||
        !hipOk(hipMemsetAsync(d->AS, 0, cells * sizeof(double), S.stream), "memset")) { freeDesc(d); return EXIT_FAILURE; }
    if (rows) {
        const dim3 grid = grid2d((rows + 3) / 4, 256);
        if (c->irpBytes == 4) hipLaunchKernelGGL((csr_to_ell_kernel<uint32_t>), grid, dim3(256), 0, S.stream, (uint32_t)rows, (uint32_t)K, d->pitch, transposed, static_cast<const uint32_t*>(c->IRP), c->JA, c->AS, d->JA, d->AS, d->RL);
        else                  hipLaunchKernelGGL((csr_to_ell_kernel<uint64_t>), grid, dim3(256), 0, S.stream, (uint32_t)rows, (uint32_t)K, d->pitch, transposed, static_cast<const uint64_t*>(c->IRP), c->JA, c->AS, d->JA, d->AS, d->RL);
    }
    if (!hipOk(hipGetLastError(), "csr_to_ell launch") || !hipOk(hipStreamSynchronize(S.stream), "csr_to_ell")) { freeDesc(d); return EXIT_FAILURE; }
    d->unit = c->unit; d->unitValue = c->unitValue;   // the same values, and row lengths always
    // handle fields follow the reference's conventions (transposed: M = slots, MAX_ROW_NZ = rows)
    if (transposed) publish(dEll, d, K, rows, c->NZ, rows);
    else            publish(dEll, d, rows, c->N, c->NZ, K);
    return EXIT_SUCCESS;
}

int hipFreeSpmat(spmat* h) {
    if (!h || !h->dev) return EXIT_SUCCESS;
    DevMat* d = descOf(h, "hipFreeSpmat");
    if (!d) return EXIT_FAILURE;
    freeDesc(d);
    memset(h, 0, sizeof *h);
    return EXIT_SUCCESS;
}

// ------------------------------------------------------------------------ launchers
// enqueue-only entry used by shard.hip (explicit stream, no timing bracket, current device = the matrix')
int spmvHipEnqueueCSR(spmat* dMat, int warpPerRow, double* dX, double* dY, void* stream) {
    DevMat* d = descOf(dMat, dX, dY, "spmvHipEnqueueCSR");
    if (!d || d->kind != Kind::CSR) return EXIT_FAILURE;
    if (d->M == 0) return EXIT_SUCCESS;
    hipStream_t keep = S.stream;
    S.stream = static_cast<hipStream_t>(stream);
    if (warpPerRow) launchStream2<false>(d, dX, dY); else launchStream2<true>(d, dX, dY);
    S.stream = keep;
    HIP_TRY(hipGetLastError());
    return EXIT_SUCCESS;
}

// ---- CSR launchers --------------------------------------------------------------------------------------------
// the LDS-stream kernel: SEQ = one thread sums its row in ascending j (variant 1 of hipSpMVRowsCSR), otherwise the LDS
// segmented reduction (variant 1 of hipSpMVWarpPerRowCSR); candidate 0 of the two selections below
static int streamCSR(spmat* dMat, double* dX, double* dY, bool seq) {
    const char* who = seq ? "hipSpMVRowsCSR" : "hipSpMVWarpPerRowCSR";
    DevMat* d = descOf(dMat, dX, dY, who);
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("%s: handle is not CSR", who); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    Launch L(grid2d(d->nBlk2, WG_THREADS), dim3(WG_THREADS));
    if (seq) launchStream2<true>(d, dX, dY); else launchStream2<false>(d, dX, dY);
    return L.finish(who);
}
static int streamSerial(spmat* m, double* x, CONFIG, double* y)  { return streamCSR(m, x, y, true); }
static int streamReduce(spmat* m, double* x, CONFIG, double* y)  { return streamCSR(m, x, y, false); }

// the two-phase / stripes launchers on the given FORM of their format (built at the first call)
static int tilesForm(spmat* dMat, double* dX, double* dY, bool det, const char* who) {
    DevMat* d = descOf(dMat, dX, dY, who);
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("%s: handle is not CSR", who); return EXIT_FAILURE; }
    if (d->M == 0 || d->NZ == 0) return nothingToLaunch(d, dY);         // nothing to slice: y = 0
    useTiles(d, det);
    if (!d->tiles) {
        const spmvTilesOpts o{0, 0, -1, 0, 1};
        if (buildTiles(d, det ? &o : nullptr)) return EXIT_FAILURE;
    }
    uint32_t bins = 0, rowsPerBin = 0;
    tilesShape(d, &bins, &rowsPerBin);
    const uint32_t p2t = tilesPhase2Threads(d);
    Launch L(grid2d((uint64_t)((bins + 7) / 8) * 8, p2t), dim3(p2t));   // phase 2's shape (phase 1: one workgroup per slice piece)
    if (enqueueTiles(d, dX, dY, S.stream)) { ERR("%s: launch failed", who); return EXIT_FAILURE; }
    return L.finish(who);
}
// mode: 0 arrival order, 1 owner wavefronts (its own layout), 2 ordered tickets (the layout of mode 0)
static int stripesForm(spmat* dMat, double* dX, double* dY, int mode, const char* who) {
    DevMat* d = descOf(dMat, dX, dY, who);
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("%s: handle is not CSR", who); return EXIT_FAILURE; }
    if (d->M == 0 || d->NZ == 0) return nothingToLaunch(d, dY);         // nothing to sweep: y = 0
    useStripes(d, mode == 1);
    if (!d->stripes) {
        const spmvStripesOpts o{0, 0, -1, -1, 1};
        if (buildStripes(d, mode == 1 ? &o : nullptr)) return EXIT_FAILURE;
    }
    Launch L(dim3(1), dim3(1));
    dim3 grid, block;
    if (enqueueStripes(d, dX, dY, S.stream, mode, &grid, &block)) { ERR("%s: launch failed", who); return EXIT_FAILURE; }
    L.shape(grid, block);                            // the persistent grid that ran: min(bins, CUs) workgroups of 256 threads
    return L.finish(who);
}
static int tilesArrival(spmat* m, double* x, CONFIG, double* y)   { return tilesForm(m, x, y, false, "hipSpMVTilesCSR"); }
static int tilesSerial(spmat* m, double* x, CONFIG, double* y)    { return tilesForm(m, x, y, true, "hipSpMVTilesCSR (deterministic)"); }
static int stripesArrival(spmat* m, double* x, CONFIG, double* y) { return stripesForm(m, x, y, 0, "hipSpMVStripesCSR"); }
static int stripesOwner(spmat* m, double* x, CONFIG, double* y)   { return stripesForm(m, x, y, 1, "hipSpMVStripesCSR (deterministic: owner wavefronts)"); }
static int stripesOrdered(spmat* m, double* x, CONFIG, double* y) { return stripesForm(m, x, y, 2, "hipSpMVStripesCSR (deterministic: ordered tickets)"); }

static int autoRun(spmat* dMat, double* dX, CONFIG cfg, double* dY, int serial, const char* who);

int hipSpMVRowsCSR(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    if (S.variantRowsCSR == 2) return autoRun(dMat, dX, cfg, dY, 1, "hipSpMVRowsCSR");
    if (S.variantRowsCSR == 1) return streamSerial(dMat, dX, cfg, dY);
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVRowsCSR");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("hipSpMVRowsCSR: handle is not CSR"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    const unsigned bt = blockThreads(cfg, BLOCKS_1D, 1024);
    const dim3 grid = grid2d((d->M + bt - 1) / bt, bt), block(bt);
    Launch L(grid, block);
    const uint32_t M = (uint32_t)d->M;
    if (d->irpBytes == 4) hipLaunchKernelGGL((csr_scalar_kernel<uint32_t>), grid, block, 0, S.stream, M, static_cast<const uint32_t*>(d->IRP), d->JA, d->AS, dX, dY);
    else                  hipLaunchKernelGGL((csr_scalar_kernel<uint64_t>), grid, block, 0, S.stream, M, static_cast<const uint64_t*>(d->IRP), d->JA, d->AS, dX, dY);
    return L.finish("hipSpMVRowsCSR");
}

int hipSpMVWarpPerRowCSR(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    if (S.variantWarpCSR == 2) return autoRun(dMat, dX, cfg, dY, 0, "hipSpMVWarpPerRowCSR");
    if (S.variantWarpCSR == 1) return streamReduce(dMat, dX, cfg, dY);
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVWarpPerRowCSR");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("hipSpMVWarpPerRowCSR: handle is not CSR"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    const unsigned bt = blockThreads(cfg, WAVESIZE * BLOCKS_2D_WARP_R, 1024);
    const unsigned rowsPerWg = bt / WAVE;
    const dim3 grid = grid2d((d->M + rowsPerWg - 1) / rowsPerWg, bt), block(bt);
    Launch L(grid, block);
    const uint32_t M = (uint32_t)d->M;
    if (d->irpBytes == 4) hipLaunchKernelGGL((csr_vector_kernel<uint32_t>), grid, block, 0, S.stream, M, static_cast<const uint32_t*>(d->IRP), d->JA, d->AS, dX, dY);
    else                  hipLaunchKernelGGL((csr_vector_kernel<uint64_t>), grid, block, 0, S.stream, M, static_cast<const uint64_t*>(d->IRP), d->JA, d->AS, dX, dY);
    return L.finish("hipSpMVWarpPerRowCSR");
}

// explicit launchers and queries work on the form last asked for with spmvHipBuild*Opt (default: arrival order)
int hipSpMVTilesCSR(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    (void)cfg;
    DevMat* d = descOf(dMat, "hipSpMVTilesCSR");
    return d ? tilesForm(dMat, dX, dY, d->tilesPref, "hipSpMVTilesCSR") : EXIT_FAILURE;
}
int hipSpMVStripesCSR(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    (void)cfg;
    DevMat* d = descOf(dMat, "hipSpMVStripesCSR");
    return d ? stripesForm(dMat, dX, dY, d->stripesPref, "hipSpMVStripesCSR") : EXIT_FAILURE;
}

int spmvHipBuildTiles(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipBuildTiles");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("spmvHipBuildTiles: handle is not CSR"); return EXIT_FAILURE; }
    useTiles(d, d->tilesPref);
    if (d->tiles) return EXIT_SUCCESS;
    const spmvTilesOpts o{0, 0, -1, 0, 1};
    return buildTiles(d, d->tilesPref ? &o : nullptr);
}
size_t spmvHipTilesBytes(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipTilesBytes");
    return d ? tilesBytes(d) : 0;
}

int spmvHipBuildSell(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipBuildSell");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("spmvHipBuildSell: handle is not CSR"); return EXIT_FAILURE; }
    return buildSell(d);
}
size_t spmvHipSellBytes(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipSellBytes");
    return d ? sellBytes(d) : 0;
}
int hipSpMVRowsSELL(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    (void)cfg;
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVRowsSELL");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("hipSpMVRowsSELL: handle is not CSR (the SELL-C-sigma copy is derived from an uploaded CSR)"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    if (!d->sell && buildSell(d)) return EXIT_FAILURE;
    Launch L(grid2d((d->M + 255) / 256, 256), dim3(256));
    if (enqueueSell(d, dX, dY, S.stream)) { ERR("hipSpMVRowsSELL: launch failed"); return EXIT_FAILURE; }
    return L.finish("hipSpMVRowsSELL");
}

int spmvHipBuildStripes(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipBuildStripes");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("spmvHipBuildStripes: handle is not CSR"); return EXIT_FAILURE; }
    useStripes(d, d->stripesPref == 1);
    if (d->stripes) return EXIT_SUCCESS;
    const spmvStripesOpts o{0, 0, -1, -1, 1};
    return buildStripes(d, d->stripesPref == 1 ? &o : nullptr);
}
int spmvHipBuildStripesOpt(spmat* dMat, const spmvStripesOpts* opts) {
    DevMat* d = descOf(dMat, "spmvHipBuildStripesOpt");
    if (!d || !opts) return EXIT_FAILURE;
    if (d->kind != Kind::CSR || d->M == 0 || d->NZ == 0) { ERR("spmvHipBuildStripesOpt: needs a non-empty CSR handle"); return EXIT_FAILURE; }
    if (buildStripes(d, opts)) return EXIT_FAILURE;
    d->stripesPref = opts->deterministic;            // what hipSpMVStripesCSR and the queries use from now on
    return EXIT_SUCCESS;
}
size_t spmvHipStripesBytes(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipStripesBytes");
    return d ? stripesBytes(d) : 0;
}
int spmvHipStripesInfo(spmat* dMat, spmvStripesInfo* info) {
    DevMat* d = descOf(dMat, "spmvHipStripesInfo");
    if (!d || !info) return EXIT_FAILURE;
    useStripes(d, d->stripesPref == 1);
    stripesInfo(d, info);
    if (info->nBins && d->stripesPref == 2) info->deterministic = 2;     // the shared-stream layout, launched in ticket order
    return EXIT_SUCCESS;
}
int spmvHipStripesShape(spmat* dMat, unsigned* nBins, unsigned* rowsPerBin, int* wide, double* buildMs) {
    spmvStripesInfo i;
    if (spmvHipStripesInfo(dMat, &i)) return EXIT_FAILURE;
    if (nBins) *nBins = i.nBins;
    if (rowsPerBin) *rowsPerBin = i.rowsPerBin;
    if (wide) *wide = i.wide;
    if (buildMs) *buildMs = i.buildMs;
    return EXIT_SUCCESS;
}

// ---- the fastest CSR launcher for THIS matrix, found by timing --------------------------------------------------
// Which kernel wins depends on where x lives relative to the caches (DESIGN.md sections 4, 7, 8): the LDS-stream
// kernel when the columns of neighbouring rows meet in L1/L2 (narrow bands, small matrices), the stripes kernel
// while x fits the Infinity Cache, the two-phase kernel beyond.  A caller of the reference picks a kernel by name
// (CUDA_CSR_ROWS, CUDA_CSR_ROWS_WARP ...); here the two names stand for two CONTRACTS, and inside each contract the
// kernel is picked by measurement, once per handle, on the caller's own x:
//   selection 0, reduction order free (hipSpMVAutoCSR, hipSpMVWarpPerRowCSR variant 2): LDS-stream kernel with the LDS
//                segmented reduction / two-phase / stripes, sums in arrival order;
//   selection 1, serial order (hipSpMVRowsCSR variant 2): LDS-stream kernel with one thread per row / the deterministic
//                forms of the two-phase and the stripes kernel -- every candidate adds a row's products in ascending j,
//                so all of them give the bits of the serial oracle and the choice is invisible in y.
// Every eligible candidate computes y (one warm-up launch that also builds its format, then AUTO_REPS timed ones -- one
// if a launch takes milliseconds), the fastest
// stays, the formats of the others are released, and the chosen launcher runs once more so that y is its own.  The
// first call is a normal -- slow -- SpMV and synchronises the stream even in enqueue-only mode.
namespace {
constexpr int      AUTO_N = 4, AUTO_REPS = 3;
constexpr float    AUTO_LONG_MS = 2.0f;                 // a launch this long is timed once
constexpr uint64_t AUTO_MIN_NNZ = 1ull << 18;        // below this a launch is mostly latency: no private format pays
constexpr uint64_t AUTO_STRIPES_X_BYTES = 256ull << 20;   // the stripes kernel re-reads x once per XCD and round of bins
struct AutoCand { const char* name; SPMV_HIP* fn; };
const AutoCand AUTO_CAND[2][AUTO_N] = {
    {{"hipSpMVWarpPerRowCSR", &streamReduce}, {"hipSpMVTilesCSR", &tilesArrival}, {"hipSpMVStripesCSR", &stripesArrival}, {nullptr, nullptr}},
    {{"hipSpMVRowsCSR", &streamSerial}, {"hipSpMVTilesCSR(deterministic)", &tilesSerial}, {"hipSpMVStripesCSR(owner wavefronts)", &stripesOwner},
     {"hipSpMVStripesCSR(ordered tickets)", &stripesOrdered}}};

int autoSelect(spmat* dMat, DevMat* d, int serial, double* dX, CONFIG cfg, double* dY) {
    const bool fmtOk = d->NZ >= AUTO_MIN_NNZ && d->NZ < IRP32_LIMIT;
    const bool stripesOk = fmtOk && d->N * 8 <= AUTO_STRIPES_X_BYTES;
    bool eligible[AUTO_N] = {true, fmtOk, stripesOk, stripesOk && serial != 0};
    if (!fmtOk) { d->autoPick[serial] = 0; return EXIT_SUCCESS; }
    if (serial) {
        // the serial-order contract is ascending j; the deterministic format kernels deliver ascending COLUMNS: the same thing
        // only when no row holds a column below its predecessor (the reference's loader guarantees it, parser.c:195-202; a
        // caller's own device CSR may not) -- otherwise the LDS-stream kernel, which walks j, is the only candidate
        uint32_t* dFlag = nullptr;
        uint32_t unsorted = 1;
        if (hipMalloc(&dFlag, 4) == hipSuccess && hipMemsetAsync(dFlag, 0, 4, S.stream) == hipSuccess) {
            const dim3 grid = grid2d((d->M + 3) / 4, 256);
            if (d->irpBytes == 4) hipLaunchKernelGGL((csr_unsorted_kernel<uint32_t>), grid, dim3(256), 0, S.stream, d->M, static_cast<const uint32_t*>(d->IRP), d->JA, dFlag);
            else                  hipLaunchKernelGGL((csr_unsorted_kernel<uint64_t>), grid, dim3(256), 0, S.stream, d->M, static_cast<const uint64_t*>(d->IRP), d->JA, dFlag);
            if (hipMemcpyAsync(&unsorted, dFlag, 4, hipMemcpyDeviceToHost, S.stream) != hipSuccess || hipStreamSynchronize(S.stream) != hipSuccess) unsorted = 1;
        }
        (void)hipFree(dFlag);
        (void)hipGetLastError();
        if (unsorted || spmvHipProbeLdsAtomicOrder() != 1) { d->autoPick[serial] = 0; return EXIT_SUCCESS; }
    }
    // which formats exist already (the caller's, or the other selection's winner): those are never freed here
    useTiles(d, serial != 0);
    const bool hadTiles = d->tiles != nullptr;
    useStripes(d, false);
    const bool hadShared = d->stripes != nullptr;    // shared-stream layout: arrival order and ordered tickets
    useStripes(d, true);
    const bool hadOwner = d->stripes != nullptr;     // per-wavefront sub-streams
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); ERR("hipSpMVAutoCSR: event creation failed"); return EXIT_FAILURE; }
    const bool wasSync = S.sync;
    int best = -1;
    float bestMs = 0;
    // Lower bound of a format kernel's time: its bytes per entry at the rate this HBM streams (MI355X_MICROARCH.md: 6.3 TB/s).
    // A candidate whose BOUND is no better than what has already been measured cannot win: its format (12 B/nnz of memory,
    // 12 B/nnz of temporaries) is not built.  Order: no format, 12 B/nnz (stripes), 28 B/nnz (two-phase).
    const double vb = d->unit ? 8.0 : 0.0;           // a matrix whose values are all the same streams no values
    const double boundMs[AUTO_N] = {0.0, (double)d->NZ * (28.0 - vb) / 6.3e12 * 1e3, (double)d->NZ * (12.0 - vb) / 6.3e12 * 1e3,
                                    (double)d->NZ * (12.0 - vb) / 6.3e12 * 1e3};
    const int order[AUTO_N] = {0, 2, 3, 1};
    for (int k = 0; k < AUTO_N; ++k) {
        const int c = order[k];
        if (!eligible[c] || !AUTO_CAND[serial][c].fn) continue;
        if (best >= 0 && boundMs[c] >= bestMs) { d->autoMs[serial][c] = 0; continue; }
        // enqueue-only throughout (the library's own timing events belong to the device of spmvHipInit; this may run on another
        // device's stream, spmvHipEnqueueAuto) and no host round trip inside e0..e1
        S.sync = false;
        const int rcWarm = AUTO_CAND[serial][c].fn(dMat, dX, cfg, dY) || hipStreamSynchronize(S.stream) != hipSuccess;   // warm-up + format build
        if (rcWarm) {                                                     // a candidate that fails is not a candidate ...
            (void)hipGetLastError();                                      // ... and must not leave its error behind for the next one
            continue;
        }
        // one timed launch; AUTO_REPS - 1 more only when a launch is short enough for its timing to be noisy (on c5 the
        // LDS-stream candidate takes 31 ms a launch: measuring it three times more costs as much as building the winner's format)
        float ms = 0, more = 0;
        bool ok = hipEventRecord(e0, S.stream) == hipSuccess && AUTO_CAND[serial][c].fn(dMat, dX, cfg, dY) == EXIT_SUCCESS &&
                  hipEventRecord(e1, S.stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        int reps = 1;
        if (ok && ms < AUTO_LONG_MS) {
            ok = hipEventRecord(e0, S.stream) == hipSuccess;
            for (int r = 1; ok && r < AUTO_REPS; ++r) ok = AUTO_CAND[serial][c].fn(dMat, dX, cfg, dY) == EXIT_SUCCESS;
            ok = ok && hipEventRecord(e1, S.stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                 hipEventElapsedTime(&more, e0, e1) == hipSuccess;
            reps = AUTO_REPS;
        }
        if (!ok) { (void)hipGetLastError(); continue; }
        const float perLaunch = (ms + more) / reps;
        d->autoMs[serial][c] = perLaunch;
        if (best < 0 || perLaunch < bestMs) { best = c; bestMs = perLaunch; }
    }
    S.sync = wasSync;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (best < 0) { ERR("hipSpMVAutoCSR: no candidate kernel ran"); return EXIT_FAILURE; }
    // the losers' private copies of the matrix (12 B/nnz each) go; formats that existed before stay.  The shared-stream
    // stripes layout serves candidate 2 of the reduction-order selection and candidate 3 of the serial-order one.
    useTiles(d, serial != 0);
    if (best != 1 && d->tiles && !hadTiles) { freeTiles(d->tiles); d->tiles = nullptr; }
    const bool keepShared = serial ? best == 3 : best == 2, keepOwner = serial && best == 2;
    useStripes(d, false);
    if (!keepShared && d->stripes && !hadShared) { freeStripes(d->stripes); d->stripes = nullptr; }
    useStripes(d, true);
    if (!keepOwner && d->stripes && !hadOwner) { freeStripes(d->stripes); d->stripes = nullptr; }
    d->autoPick[serial] = best;
    return EXIT_SUCCESS;
}
}  // namespace

static int autoRun(spmat* dMat, double* dX, CONFIG cfg, double* dY, int serial, const char* who) {
    DevMat* d = descOf(dMat, dX, dY, who);
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::CSR) { ERR("%s: handle is not CSR", who); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    if (d->autoPick[serial] < 0 && autoSelect(dMat, d, serial, dX, cfg, dY)) return EXIT_FAILURE;
    return AUTO_CAND[serial][d->autoPick[serial]].fn(dMat, dX, cfg, dY);   // (also after the selection: y then is the chosen kernel's own)
}

int hipSpMVAutoCSR(spmat* dMat, double* dX, CONFIG cfg, double* dY) { return autoRun(dMat, dX, cfg, dY, 0, "hipSpMVAutoCSR"); }

static const char* autoChoice(spmat* dMat, int serial, double* msPerCandidate) {
    DevMat* d = descOf(dMat, "spmvHipAutoChoice");
    if (!d || d->autoPick[serial] < 0) return nullptr;
    if (msPerCandidate) for (int c = 0; c < AUTO_N; ++c) msPerCandidate[c] = d->autoMs[serial][c];
    return AUTO_CAND[serial][d->autoPick[serial]].name;
}
const char* spmvHipAutoChoice(spmat* dMat, double* msPerCandidate) { return autoChoice(dMat, 0, msPerCandidate); }
const char* spmvHipAutoChoiceRows(spmat* dMat, double* msPerCandidate) { return autoChoice(dMat, 1, msPerCandidate); }

// enqueue-only form of a selection's launcher on an explicit stream (shard.hip: one stream per device); the first call
// for a handle measures the candidates on that stream and synchronises it
static int enqueueAuto(spmat* dMat, double* dX, double* dY, void* stream, int serial, const char* who) {
    hipStream_t keepStream = S.stream;
    const bool keepSync = S.sync;
    S.stream = static_cast<hipStream_t>(stream);
    S.sync = false;
    const int rc = autoRun(dMat, dX, CONFIG{}, dY, serial, who);
    S.stream = keepStream;
    S.sync = keepSync;
    return rc;
}
int spmvHipEnqueueAuto(spmat* dMat, double* dX, double* dY, void* stream) { return enqueueAuto(dMat, dX, dY, stream, 0, "spmvHipEnqueueAuto"); }
int spmvHipEnqueueAutoRows(spmat* dMat, double* dX, double* dY, void* stream) { return enqueueAuto(dMat, dX, dY, stream, 1, "spmvHipEnqueueAutoRows"); }

// the two-phase format the explicit entry points below work on: the preferred form, built if missing
static DevMat* tilesReady(spmat* dMat, const char* who) {
    DevMat* d = descOf(dMat, who);
    if (!d) return nullptr;
    if (d->kind != Kind::CSR || d->M == 0 || d->NZ == 0) { ERR("%s: needs a non-empty CSR handle", who); return nullptr; }
    useTiles(d, d->tilesPref);
    if (!d->tiles) {
        const spmvTilesOpts o{0, 0, -1, 0, 1};
        if (buildTiles(d, d->tilesPref ? &o : nullptr)) return nullptr;
    }
    return d;
}

int spmvHipTilesShape(spmat* dMat, unsigned* nBins, unsigned* rowsPerBin) {
    DevMat* d = tilesReady(dMat, "spmvHipTilesShape");
    if (!d || !nBins || !rowsPerBin) return EXIT_FAILURE;
    uint32_t b = 0, r = 0;
    tilesShape(d, &b, &r);
    *nBins = b; *rowsPerBin = r;
    return EXIT_SUCCESS;
}

int spmvHipBuildTilesOpt(spmat* dMat, const spmvTilesOpts* opts) {
    DevMat* d = descOf(dMat, "spmvHipBuildTilesOpt");
    if (!d) return EXIT_FAILURE;
    if (!opts) return EXIT_FAILURE;
    if (d->kind != Kind::CSR || d->M == 0 || d->NZ == 0) { ERR("spmvHipBuildTilesOpt: needs a non-empty CSR handle"); return EXIT_FAILURE; }
    if (buildTiles(d, opts)) return EXIT_FAILURE;
    d->tilesPref = opts->deterministic != 0;         // what hipSpMVTilesCSR, Expand / Reduce and the queries use from now on
    return EXIT_SUCCESS;
}

int spmvHipTilesInfo(spmat* dMat, spmvTilesInfo* info) {
    DevMat* d = descOf(dMat, "spmvHipTilesInfo");
    if (!d || !info) return EXIT_FAILURE;
    useTiles(d, d->tilesPref);
    tilesInfo(d, info);
    return EXIT_SUCCESS;
}

int spmvHipTilesBinRow(spmat* dMat, unsigned bin, ulong* firstRow) {
    DevMat* d = tilesReady(dMat, "spmvHipTilesBinRow");
    if (!d || !firstRow) return EXIT_FAILURE;
    *firstRow = tilesBinRow(d, bin);
    return EXIT_SUCCESS;
}

int hipSpMVTilesExpand(spmat* dMat, double* dX) {
    DevMat* d = tilesReady(dMat, "hipSpMVTilesExpand");
    if (!d) return EXIT_FAILURE;
    if (!dX) { ERR("hipSpMVTilesExpand: x is NULL"); return EXIT_FAILURE; }
    Launch L(dim3(1), dim3(1024));
    if (enqueueTilesExpand(d, dX, S.stream)) { ERR("hipSpMVTilesExpand: launch failed"); return EXIT_FAILURE; }
    return L.finish("hipSpMVTilesExpand");
}

int hipSpMVTilesReduce(spmat* dMat, unsigned binBegin, unsigned binEnd, double* dY, int nExtra, double* const* dExtra) {
    DevMat* d = tilesReady(dMat, "hipSpMVTilesReduce");
    if (!d) return EXIT_FAILURE;
    uint32_t b = 0, r = 0;
    tilesShape(d, &b, &r);
    if (binBegin > binEnd || binEnd > b || nExtra < 0 || nExtra > SPMV_MAX_PEERS || (nExtra && !dExtra) || !dY) {
        ERR("hipSpMVTilesReduce: bins [%u,%u) of %u, %d extra destinations: invalid", binBegin, binEnd, b, nExtra);
        return EXIT_FAILURE;
    }
    Launch L(dim3(binEnd - binBegin ? binEnd - binBegin : 1), dim3(1024));
    if (enqueueTilesReduce(d, binBegin, binEnd, dY, nExtra, dExtra, S.stream)) { ERR("hipSpMVTilesReduce: launch failed"); return EXIT_FAILURE; }
    return L.finish("hipSpMVTilesReduce");
}

int hipSpMVTilesReducePush(spmat* dMat, double* dY, int nExtra, double* const* dExtra) {
    DevMat* d = tilesReady(dMat, "hipSpMVTilesReducePush");
    if (!d) return EXIT_FAILURE;
    if (nExtra < 1 || nExtra > SPMV_MAX_PEERS || !dExtra || !dY) { ERR("hipSpMVTilesReducePush: %d destinations: invalid", nExtra); return EXIT_FAILURE; }
    if (!g_pushSide) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&g_pushSide, hipStreamNonBlocking, hi));      // dispatched ahead of phase 2's later rounds
        HIP_TRY(hipEventCreateWithFlags(&g_pushFork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g_pushJoin, hipEventDisableTiming));
    }
    uint32_t b = 0, r = 0;
    tilesShape(d, &b, &r);
    const bool wasSync = S.sync;
    Launch L(dim3(b), dim3(1024));
    if (enqueueTilesReducePush(d, dY, nExtra, dExtra, S.stream, g_pushSide, g_pushFork, g_pushJoin)) { ERR("hipSpMVTilesReducePush: launch failed"); return EXIT_FAILURE; }
    g_pushPending = true;
    if (wasSync && spmvHipTilesPushJoin()) return EXIT_FAILURE;         // synchronous mode: everything delivered on return
    return L.finish("hipSpMVTilesReducePush");
}

int spmvHipTilesPushJoin(void) {
    if (g_pushPending) {
        HIP_TRY(hipStreamWaitEvent(S.stream, g_pushJoin, 0));        // push kernels run in order on one stream: the last event covers all
        g_pushPending = false;
    }
    return EXIT_SUCCESS;
}

int spmvHipTilesPushFailed(spmat* dMat) {
    DevMat* d = descOf(dMat, "spmvHipTilesPushFailed");
    return d ? tilesPushFailed(d) : 1;
}

int hipSpMVRowsELL(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVRowsELL");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::ELL_COLMAJOR) { ERR("hipSpMVRowsELL: expects the transposed (column-major) ELL upload: ellTranspose() + spMatCpyELL()"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    const unsigned bt = blockThreads(cfg, BLOCKS_1D, 1024);
    const dim3 grid = grid2d((d->M + bt - 1) / bt, bt), block(bt);
    Launch L(grid, block);
    const bool rl = S.ellRowLens && d->RL;
    if (rl && d->unit) hipLaunchKernelGGL((ell_colmajor_thread<true, true>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, dX, dY, d->unitValue);
    else if (rl) hipLaunchKernelGGL((ell_colmajor_thread<true>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, dX, dY, 0.0);
    else         hipLaunchKernelGGL((ell_colmajor_thread<false>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, dX, dY, 0.0);
    return L.finish("hipSpMVRowsELL");
}

int hipSpMVRowsELLNNTransposed(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVRowsELLNNTransposed");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::ELL_ROWMAJOR) { ERR("hipSpMVRowsELLNNTransposed: expects the row-major ELL upload (no ellTranspose)"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    if (S.variantEllRowMajor == 1 && cfg.blockSize.x == 0 && d->pitch && d->pitch <= (size_t)STREAM_NNZ)
        return launchEllStream(d, S.ellRowLens && d->RL, true, dX, dY, "hipSpMVRowsELLNNTransposed");
    const unsigned bt = blockThreads(cfg, BLOCKS_1D, 1024);
    const dim3 grid = grid2d((d->M + bt - 1) / bt, bt), block(bt);
    Launch L(grid, block);
    const bool rl = S.ellRowLens && d->RL;
    if (rl) hipLaunchKernelGGL((ell_rowmajor_thread<true>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, dX, dY);
    else    hipLaunchKernelGGL((ell_rowmajor_thread<false>), grid, block, 0, S.stream, (uint32_t)d->M, (uint32_t)d->K, d->pitch, d->JA, d->AS, d->RL, dX, dY);
    return L.finish("hipSpMVRowsELLNNTransposed");
}

int hipSpMVWarpsPerRowELLNTrasposed(spmat* dMat, double* dX, CONFIG cfg, double* dY) {
    DevMat* d = descOf(dMat, dX, dY, "hipSpMVWarpsPerRowELLNTrasposed");
    if (!d) return EXIT_FAILURE;
    if (d->kind != Kind::ELL_ROWMAJOR) { ERR("hipSpMVWarpsPerRowELLNTrasposed: expects the row-major ELL upload (no ellTranspose)"); return EXIT_FAILURE; }
    if (d->M == 0) return nothingToLaunch(d, nullptr);
    const bool rl = S.ellRowLens && d->RL;
    if (d->pitch && d->pitch <= (size_t)STREAM_NNZ)   // a row fits a block of the LDS-stream kernel: dense span loads, LDS segmented reduction
        return launchEllStream(d, rl, false, dX, dY, "hipSpMVWarpsPerRowELLNTrasposed");
    // longer rows: G lanes per row -- the smallest power of two covering the slots, 4..64
    int G = 4;
    while (G < WAVE && (uint64_t)G < d->K) G <<= 1;
    const unsigned bt = 256;
    (void)cfg;
    const uint64_t threads = (d->M + ELL_GROUP_ROWS - 1) / ELL_GROUP_ROWS * (uint64_t)G;     // a group of G lanes owns 4 rows
    const dim3 grid = grid2d((threads + bt - 1) / bt, bt), block(bt);
    Launch L(grid, block);
    switch (G) {
        case 4:  launchEllGroup<4>(d, rl, grid, block, dX, dY); break;
        case 8:  launchEllGroup<8>(d, rl, grid, block, dX, dY); break;
        case 16: launchEllGroup<16>(d, rl, grid, block, dX, dY); break;
        case 32: launchEllGroup<32>(d, rl, grid, block, dX, dY); break;
        default: launchEllGroup<64>(d, rl, grid, block, dX, dY); break;
    }
    return L.finish("hipSpMVWarpsPerRowELLNTrasposed");
}

// ------------------------------------------------------------------------ events
int spmvHipEventCreate(void** ev) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return EXIT_SUCCESS;
}
int spmvHipEventDestroy(void* ev) { HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(ev))); return EXIT_SUCCESS; }
int spmvHipEventRecord(void* ev) { HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), S.stream)); return EXIT_SUCCESS; }
int spmvHipEventElapsedMs(void* a, void* b, float* ms) {
    HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(b)));
    HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(a), static_cast<hipEvent_t>(b)));
    return EXIT_SUCCESS;
}

// ------------------------------------------------------------------------ SPMV_INTERF wrappers
int spmvHipRowsCSR(spmat* mat, double* x, CONFIG* cfg, double* y) { return hostCall(mat, x, cfg, y, 0, &hipSpMVRowsCSR); }
int spmvHipWarpPerRowCSR(spmat* mat, double* x, CONFIG* cfg, double* y) { return hostCall(mat, x, cfg, y, 0, &hipSpMVWarpPerRowCSR); }
int spmvHipRowsELL(spmat* mat, double* x, CONFIG* cfg, double* y) { return hostCall(mat, x, cfg, y, 2, &hipSpMVRowsELL); }
int spmvHipWarpsPerRowELL(spmat* mat, double* x, CONFIG* cfg, double* y) { return hostCall(mat, x, cfg, y, 1, &hipSpMVWarpsPerRowELLNTrasposed); }
int spmvHipDropCache(void) {
    for (auto& kv : g_cache) kv.second.release();
    g_cache.clear();
    return EXIT_SUCCESS;
}

// ------------------------------------------------------------------------ sharding helpers (host side)
int spmvHipPartitionRows(const ulong* IRP, ulong M, int nParts, ulong* bounds) {
    if (!IRP || !bounds || nParts <= 0) return EXIT_FAILURE;
    const ulong nnz = IRP[M] - IRP[0];
    bounds[0] = 0;
    for (int p = 1; p < nParts; ++p) {
        // first row whose starting offset reaches p/nParts of the nnz
        const ulong target = IRP[0] + (ulong)(((__uint128_t)nnz * (unsigned)p) / (unsigned)nParts);
        const ulong* it = std::lower_bound(IRP, IRP + M + 1, target);
        ulong r = (ulong)(it - IRP);
        if (r > M) r = M;
        // choose the closer of r-1 / r
        if (r > 0 && target - IRP[r - 1] < IRP[r] - target) --r;
        bounds[p] = std::max(r, bounds[p - 1]);
    }
    bounds[nParts] = M;
    return EXIT_SUCCESS;
}

int spmvHipCompactRows(double* dY, const double* dYPad, const ulong* bounds, int nParts, ulong maxRows) {
    if (!dY || !dYPad || !bounds || nParts <= 0) return EXIT_FAILURE;
    for (int p = 0; p < nParts; ++p) {
        const ulong rows = bounds[p + 1] - bounds[p];
        if (rows > maxRows) { ERR("spmvHipCompactRows: block %d has %lu rows > pad %lu", p, rows, maxRows); return EXIT_FAILURE; }
        if (rows)
            HIP_TRY(hipMemcpyAsync(dY + bounds[p], dYPad + (size_t)p * maxRows, rows * sizeof(double),
                                   hipMemcpyDeviceToDevice, S.stream));
    }
    if (S.sync) HIP_TRY(hipStreamSynchronize(S.stream));
    return EXIT_SUCCESS;
}

spmat* spmvHipRowBlockCSR(const spmat* host, ulong r0, ulong r1) {
    if (!host || !host->IRP || r0 > r1 || r1 > host->M) return nullptr;
    spmat* out = static_cast<spmat*>(calloc(1, sizeof(spmat)));
    if (!out) return nullptr;
    const ulong base = host->IRP[r0], nz = host->IRP[r1] - base, rows = r1 - r0;
    out->M = rows; out->N = host->N; out->NZ = nz;
    out->IRP = static_cast<ulong*>(malloc((rows + 1) * sizeof(ulong)));
    out->JA  = static_cast<ulong*>(malloc(std::max<ulong>(nz, 1) * sizeof(ulong)));
    out->AS  = static_cast<double*>(malloc(std::max<ulong>(nz, 1) * sizeof(double)));
    if (host->RL) out->RL = static_cast<ulong*>(malloc(std::max<ulong>(rows, 1) * sizeof(ulong)));
    if (!out->IRP || !out->JA || !out->AS || (host->RL && !out->RL)) {
        free(out->IRP); free(out->JA); free(out->AS); free(out->RL); free(out);
        return nullptr;
    }
    for (ulong r = 0; r <= rows; ++r) out->IRP[r] = host->IRP[r0 + r] - base;
    memcpy(out->JA, host->JA + base, nz * sizeof(ulong));
    memcpy(out->AS, host->AS + base, nz * sizeof(double));
    if (host->RL) memcpy(out->RL, host->RL + r0, rows * sizeof(ulong));
    return out;
}

}  // extern "C"
