// device_mat.hpp -- the library-private descriptor behind a device `spmat` handle.
//
// HBM layout (see DESIGN.md "Data layout in HBM"):
//   AS   fp64 values
//   JA   32-bit column ids (host keeps 64-bit `ulong`; narrowed during upload)
//   IRP  32-bit row pointers when NZ < 2^32, else 64-bit
//   RL   32-bit row lengths (optional)
//   ELL  row-major  [rows ][pitch(slots)]  pitch = slots rounded up to 16 elements
//        col-major  [slots][pitch(rows )]  pitch = rows  rounded up to 64 elements
//   blkInfo / blkBase  row blocks of the LDS-stream kernel (CSR only)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "spmvHip.h"

namespace spmvhip {

constexpr int      WAVE            = 64;
constexpr int      WG_THREADS      = 256;               // 4 wavefronts
constexpr int      STREAM_NNZ      = 2048;              // nnz staged in LDS per workgroup (16 KiB of fp64)
// 32-bit row pointers are used only below this nnz count, so that `j + stride`
// in the kernels can never wrap around 2^32
constexpr uint64_t IRP32_LIMIT     = (1ull << 32) - 65536;

enum class Kind : int { CSR = 0, ELL_ROWMAJOR = 1, ELL_COLMAJOR = 2 };

struct TileFormat;          // column-sliced two-phase format, tiles.hip
struct SellFormat;          // SELL-C-sigma, sell.hip
struct StripeFormat;        // bin-wise CSC (y bins in LDS, x from the XCD's L2), stripes.hip

struct DevMat {
    uint32_t magic = 0x53504D56;    // 'SPMV'
    Kind     kind  = Kind::CSR;
    uint64_t M = 0, N = 0, NZ = 0;  // logical rows, cols, true nnz
    uint64_t K = 0;                 // ELL slots per row (max row nnz)
    int      irpBytes = 4;
    void*     IRP = nullptr;
    uint32_t* JA  = nullptr;
    double*   AS  = nullptr;
    uint32_t* RL  = nullptr;
    size_t    pitch = 0;            // ELL pitch in elements (same for JA and AS)
    bool      owns = true;          // false for adopted arrays
    // every stored value is the same double (MatrixMarket `pattern` files are loaded as all 1.0 -- the graphs of the
    // reference's report, asia_osm and channel-500x100x100, are such files): found at upload; the CSR kernels then take
    // the value from a register instead of streaming 8 B per entry.  c * x[j] rounds exactly as AS[j] * x[j] does.
    bool      unit = false;
    double    unitValue = 0.0;
    uint64_t  maxRowNnz = 0;
    // row blocks of the LDS-stream kernel (CSR): consecutive rows packed while their nnz <= STREAM_NNZ; a longer row is a
    // block of its own; long rows first, then row order
    uint4*    blkInfo = nullptr;    // {first row, #rows, #nnz, long-row flag}
    uint64_t* blkBase = nullptr;    // nnz offset of the block
    uint32_t  nBlk2 = 0, nLong2 = 0;
    // The two-phase and the stripes format exist in two FORMS each -- arrival-order and deterministic (serial-order) sums --
    // and a handle may hold both: `tiles` / `stripes` is the ACTIVE one (what the functions of tiles.hip / stripes.hip work
    // on), `*Alt` the other.  useTiles / useStripes make the requested form the active one (possibly a null slot still to
    // be built).  `*Pref` is the form the explicit launchers and queries use: set by spmvHipBuildTilesOpt / ...StripesOpt.
    TileFormat* tiles = nullptr, *tilesAlt = nullptr;         // built lazily by hipSpMVTilesCSR / spmvHipBuildTiles / the selections
    StripeFormat* stripes = nullptr, *stripesAlt = nullptr;   // built lazily by hipSpMVStripesCSR / spmvHipBuildStripes / the selections
    bool      tilesPref = false;
    int       stripesPref = 0;      // 0 arrival order, 1 owner wavefronts, 2 ordered tickets (spmvStripesOpts.deterministic)
    SellFormat* sell = nullptr;     // built lazily by hipSpMVRowsSELL / spmvHipBuildSell
    // the selections: [0] among the reduction-order kernels (hipSpMVAutoCSR, hipSpMVWarpPerRowCSR), [1] among the
    // serial-order kernels (hipSpMVRowsCSR): index of the launcher chosen for this matrix (-1: not chosen yet) ...
    int       autoPick[2] = {-1, -1};
    float     autoMs[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};    // ... and what each candidate took (0 = not tried)
};

int  buildSell(DevMat* d);                                      // sell.hip
void freeSell(SellFormat* f);
int  enqueueSell(DevMat* d, const double* x, double* y, hipStream_t stream);
size_t sellBytes(const DevMat* d);

int  buildStripes(DevMat* d, const spmvStripesOpts* opts = nullptr);   // stripes.hip; explicit options replace an existing format
void freeStripes(StripeFormat* f);
void useStripes(DevMat* d, bool subStreams);                    // make that LAYOUT the active one (d->stripes may then be null)
int  enqueueStripes(DevMat* d, const double* x, double* y, hipStream_t stream, int mode, dim3* grid = nullptr, dim3* block = nullptr);
size_t stripesBytes(const DevMat* d);
void stripesInfo(const DevMat* d, spmvStripesInfo* out);

int  buildTiles(DevMat* d, const spmvTilesOpts* opts = nullptr); // tiles.hip; explicit options replace an existing format
void tilesInfo(const DevMat* d, spmvTilesInfo* out);
void freeTiles(TileFormat* t);
void useTiles(DevMat* d, bool deterministic);                   // make that form the active one (d->tiles may then be null)
void peerFinalize();                                            // peer.hip: the copy streams of the push exchange
void freeTilesWorkspace();                                      // the per-device product workspace (8 B/nnz of the largest matrix)
int  enqueueTiles(DevMat* d, const double* x, double* y, hipStream_t stream);
int  enqueueTilesExpand(DevMat* d, const double* x, hipStream_t stream);
int  enqueueTilesReduce(DevMat* d, uint32_t binBegin, uint32_t binEnd, double* y, int nExtra, double* const* extra, hipStream_t stream);
int  enqueueTilesReducePush(DevMat* d, double* y, int nExtra, double* const* extra, hipStream_t stream, hipStream_t side,
                            hipEvent_t evFork, hipEvent_t evJoin);
int  tilesPushFailed(DevMat* d);
void tilesShape(const DevMat* d, uint32_t* bins, uint32_t* rowsPerBin);
uint32_t tilesPhase2Threads(const DevMat* d);                   // workgroup size of phase 2 for the active form
uint64_t tilesBinRow(const DevMat* d, uint32_t bin);
hipStream_t libraryStream();                                    // abi.hip: the stream set with spmvHipSetStream
size_t tilesBytes(const DevMat* d);

// Fold `blocks` workgroups into an (x, y) grid whose x extent keeps
// x * threads < 2^32 (AQL grid_size is 32-bit work-items per dimension).
// Kernels use linear_block() and bounds-check against the real block count.
inline dim3 grid2d(uint64_t blocks, unsigned threads) {
    const uint64_t maxX = ((1ull << 32) - 1) / threads;         // >= 4 M blocks for <= 1024 threads
    if (blocks <= maxX) return dim3((unsigned)(blocks ? blocks : 1), 1, 1);
    const uint64_t gx = 1ull << 20;
    return dim3((unsigned)gx, (unsigned)((blocks + gx - 1) / gx), 1);
}

// a failed call is reported HERE, once: the runtime's sticky last-error is cleared so that the next, unrelated call's
// hipGetLastError() does not see it (a failed hipMalloc of a format build used to make the next spmvHipVecFill "fail")
inline bool hipOk(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    fprintf(stderr, "\33[31m\33[1m\33[44mlibspmvhip: %s\t%s\33[0m\n", what, hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr) do { if (!::spmvhip::hipOk((expr), #expr)) return EXIT_FAILURE; } while (0)

}  // namespace spmvhip
