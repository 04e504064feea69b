/*
 * spmvHip.h -- C-ABI of libspmvhip.so: the MI355X (gfx950) replacement for the
 * reference's CUDA path.  Plain C, plain pointers and sizes; no C++/torch types.
 *
 * Every entry point returns EXIT_SUCCESS (0) / EXIT_FAILURE (1) like the
 * reference's host functions (cudaUtils.cu:45,54; SpMV_CSR_OMP.c:35,61), prints
 * its diagnostics on stderr and NEVER calls exit() (the reference's
 * checkCudaErrors does, cudaUtils.h:26-34 -- not reproduced).
 *
 * What each group replaces in the reference (paths relative to its root):
 *   upload / free ......... src/include/cudaUtils.h:60-78, src/commons/cudaUtils.cu:20-98
 *   SpMV launchers ........ the five __global__ kernels of src/SpMV_CUDA.cu:33-135 as the
 *                           drivers invoke them: f<<<grid,block>>>(dMat,dVect,Conf,dOutV)
 *                           (src/main.cu:233, test/SpMV_test.cu:112)
 *   vectors / lifecycle ... the cudaMalloc/cudaMemcpy/cudaFree calls the drivers make inline
 *                           (src/main.cu:195-199,245-247,277-280)
 *   sharding .............. new (the reference is single-GPU); see DESIGN.md "Multi-GPU"
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include "spmv_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ lifecycle */
/* Select device `dev` for the calling thread and check the caller was compiled
 * against the same struct layouts.  Must be called once before anything else. */
int spmvHipInit(int dev, size_t sizeofSpmat, size_t sizeofConfig);
/* Gives back everything the library holds on its own account: the cached device copies of the host-pointer wrappers, the
 * per-device product workspace of the two-phase kernel, its side streams and events.  Handles made by spMatCpy* stay the
 * caller's to free (hipFreeSpmat).  spmvHipInit() may be called again afterwards. */
int spmvHipFinalize(void);
/* number of visible devices, or -1 */
int spmvHipDeviceCount(void);
/* Stream (hipStream_t passed as void*) used by all later launches/copies of the
 * calling process; NULL = the default stream.  The library uses a stream only INSIDE the calls made while it is set (or
 * passed): it may be destroyed afterwards without telling the library (the two-phase kernel's product workspace is handed
 * from stream to stream through an event recorded behind each use, never through the previous stream). */
int spmvHipSetStream(void* stream);
/* (With spmvHipSetSync(0) a launcher whose format exists only enqueues kernels on this stream -- no allocation, no
 * synchronisation, no timing events; the two-phase launcher records its one hand-over event unless the stream is being
 * captured -- so a solver's inner loop can be captured into a HIP graph and replayed:
 * tests/test_gpu_parity.py::test_launchers_capture_into_a_hip_graph.  A captured two-phase launch uses the device's product
 * workspace whenever the graph is replayed, which the library cannot see: keep replays in stream order with, or
 * synchronised against, two-phase launches of OTHER streams.)
 * Every launcher checks on the host what it can: the handle (kind, magic), NULL x / y (refused: a NULL would be a GPU page
 * fault in every lane); the EXTENT of x and y is the caller's promise, as in the reference. */
/* sync != 0 (default): every SpMV launcher waits for completion before it
 * returns and stores the device time in `ElapsedInternal`-style seconds
 * retrievable with spmvHipLastKernelSeconds() -- the behaviour of the
 * reference drivers (launch; cudaDeviceSynchronize; main.cu:233-238).
 * sync == 0: launchers only enqueue (for timed loops and graph capture). */
int spmvHipSetSync(int sync);
double spmvHipLastKernelSeconds(void);
/* launch shape of the most recent SpMV launcher call */
int spmvHipLastLaunch(spmvDim3* grid, spmvDim3* block);
int spmvHipDeviceSynchronize(void);

/* ------------------------------------------------------------- dense vectors */
int spmvHipVecAlloc(double** dVec, size_t n);
int spmvHipVecFree(double* dVec);
int spmvHipVecUp(double* dVec, const double* hVec, size_t n);
int spmvHipVecDown(double* hVec, const double* dVec, size_t n);
/* fill with a 64-bit pattern (e.g. a NaN payload to poison y before a launch) */
int spmvHipVecFill(double* dVec, size_t n, uint64_t pattern);

/* raw device bytes (device-format index arrays built on the GPU) */
int spmvHipMalloc(void** dPtr, size_t bytes);
int spmvHipFree(void* dPtr);
int spmvHipMemcpyUp(void* dDst, const void* hSrc, size_t bytes);
int spmvHipMemcpyDown(void* hDst, const void* dSrc, size_t bytes);

/* ------------------------------------------------------------- matrix upload */
/* Host CSR -> device (cudaUtils.cu:20-55).  `dMat` is caller-owned host memory
 * that becomes the device handle.  Column ids are narrowed to 32 bit, row
 * pointers to 32 bit when NZ < 2^32; row blocks for the LDS-stream kernel are
 * computed here.  host->RL may be NULL. */
int spMatCpyCSR(spmat* host, spmat* dMat);
/* Host ELL -> device (cudaUtils.cu:56-98).  Accepts the row-major matrix the
 * loader produces, or the output of ellTranspose() (recognised the same way
 * the reference kernels do: after transposition M holds the slot count and
 * MAX_ROW_NZ the row count, sparseUtils.c:168-171) -- pass `transposed` = 1 for
 * the latter.  Pitch = row length rounded up to 64 elements; pitchJA/pitchAS
 * are stored in elements as the reference does (cudaUtils.cu:81-83).  The
 * reference's field swap loses the column count of a transposed matrix; this
 * repo's ellTranspose() keeps it in the (otherwise unused) host field pitchJA
 * and the upload checks every column id against it (0 = unknown, unchecked). */
int spMatCpyELL(spmat* host, spmat* dMat);
int spMatCpyELLTransposed(spmat* hostT, spmat* dMat);
/* Device-side CSR -> ELL of an uploaded matrix (slots = its longest row, row
 * lengths kept): transposed = 0 gives the row-major handle spMatCpyELL(ell)
 * would, 1 the column-major one of ellTranspose + spMatCpyELL.  ELL size
 * guard: the reference's loader refuses an ELL copy above a fixed number of
 * padded cells (parser.c:223-232); here the copy is refused (EXIT_FAILURE, no
 * allocation attempted) when it does not fit the device memory that is free. */
int spmvHipCsrToEll(spmat* dCsr, int transposed, spmat* dEll);
/* Release the device arrays behind a handle (cudaUtils.h:70-78). */
int hipFreeSpmat(spmat* dMat);

/* Adopt arrays that are ALREADY on the device in device format (u32 columns,
 * u32 or u64 row pointers) -- used by the on-device synthetic generator and by
 * callers that assemble matrices on the GPU.  The handle takes ownership of
 * nothing: the caller frees the arrays after hipFreeSpmat(). `irpBytes` is 4 or 8.
 * `hIRP` is the same row-pointer array on the host (needed for the row-block
 * analysis); it may be NULL, in which case it is downloaded. */
int spmvHipAdoptCSR(spmat* dMat, ulong M, ulong N, ulong NZ,
                    const void* dIRP, int irpBytes, const uint32_t* dJA,
                    const double* dAS, const void* hIRP);

/* --------------------------------------------------------------- SpMV on GPU */
/* y = A x with A, x, y resident on the device.  (mat, x, CONFIG by value, y):
 * the parameter list of the reference's SPMV_CUDA typedef (SpMV.h:119-120). */
typedef int (SPMV_HIP)(spmat*, double*, CONFIG, double*);
typedef int (*SPMV_HIP_INTERF)(spmat*, double*, CONFIG, double*);

SPMV_HIP hipSpMVRowsCSR;                  /* <- cudaSpMVRowsCSR                  SpMV_CUDA.cu:33-49   */
SPMV_HIP hipSpMVWarpPerRowCSR;            /* <- cudaSpMVWarpPerRowCSR            SpMV_CUDA.cu:52-73   */
SPMV_HIP hipSpMVRowsELL;                  /* <- cudaSpMVRowsELL (transposed)     SpMV_CUDA.cu:79-96   */
SPMV_HIP hipSpMVRowsELLNNTransposed;      /* <- cudaSpMVRowsELLNNTransposed      SpMV_CUDA.cu:99-115  */
SPMV_HIP hipSpMVWarpsPerRowELLNTrasposed; /* <- cudaSpMVWarpsPerRowELLNTrasposed SpMV_CUDA.cu:116-135 */

/* Column-sliced two-phase SpMV for matrices whose x gather misses the caches
 * (DESIGN.md section 7): the GPU counterpart of the reference's 2-D decomposed
 * CPU variants spmvTilesCSR / spmvTilesAllocdCSR (src/SpMV_CSR_OMP.c:101-226:
 * column partitions, partial results, final reduction).  The slice-major copy
 * of the matrix (+12 B/nnz of device memory per matrix, plus ONE product
 * workspace of 8 B/nnz of the largest matrix, shared by all matrices of the
 * device and used in stream order) is built on the device at the first call, or
 * explicitly with spmvHipBuildTiles.  Row sums are added in
 * arrival order (LDS atomics): equal to the oracle to rounding, not bitwise -- unless the
 * deterministic form was asked for with spmvHipBuildTilesOpt (serial order, the oracle's bits;
 * +4 B/nnz).  The launcher runs the form last asked for (default: arrival order); a handle can
 * hold both. */
SPMV_HIP hipSpMVTilesCSR;
int    spmvHipBuildTiles(spmat* dMat);
size_t spmvHipTilesBytes(spmat* dMat);

/* One-pass SpMV with y bins in LDS and x served by the XCD's L2: rows are cut into
 * bins of <= 20 000 consecutive rows with equal entry counts, a workgroup owns a bin, and inside a bin the
 * entries are stored in COLUMN order, so that the workgroups resident on one XCD sweep x together and gather
 * from its L2.  12 B/nnz of streaming (fp64 value + {17-bit column offset, 15-bit local row}) and no second
 * pass -- the intent of cudaSpMVWarpPerRowCSR (src/SpMV_CUDA.cu:52-73: coalesced AS/JA, gathered x, on-chip
 * reduction) with the reduction in LDS accumulators.  The copy of the matrix (+12 B/nnz of device memory)
 * is built on the device at the first call or with spmvHipBuildStripes.  Pays off while x (N * 8 B) is small
 * against the entry stream; very wide matrices are the two-phase kernel's.  Row sums are added in arrival
 * order (LDS atomics): equal to the oracle to rounding, not bitwise -- unless a deterministic form was asked for with
 * spmvHipBuildStripesOpt (below).  The launcher runs the form last asked for (default: arrival order). */
SPMV_HIP hipSpMVStripesCSR;
int    spmvHipBuildStripes(spmat* dMat);
size_t spmvHipStripesBytes(spmat* dMat);
/* shape of the built format (zeros before the build): bins, rows of the highest bin, 1 if the 14 B/nnz
 * encoding with 32-bit columns had to be used, device time of the one-time build in ms */
int    spmvHipStripesShape(spmat* dMat, unsigned* nBins, unsigned* rowsPerBin, int* wide, double* buildMs);
/* Build options of the stripes format (all zero / -1 = automatic).  Arguments of the build, not process state:
 * spmvHipBuildStripesOpt(dMat, &opts) builds -- or REbuilds -- the format of this handle with them.
 *   rowsPerBin     0 or 1..20000: upper bound of the rows of a bin (y of a bin lives in LDS).
 *   grid           0 or 1..CUs of the device: persistent workgroups that walk the bins (the bin count is made a
 *                  multiple of it); fewer than the CU count leaves CUs to a kernel running beside this one.
 *   spread         -1 or 0..1024: 1/1024ths of a bin over which the column sweeps of one XCD's workgroups start
 *                  (default 6; ignored by the deterministic form).
 *   wide           1: force the 14 B/nnz encoding with 32-bit columns; 0 / -1: only when a step spans >= 2^17 columns.
 *   deterministic  1 or 2: a row's products are added in ascending column order, so the result is the same bits in every run,
 *                  on any number of row shards, and -- the products being rounded before they are added -- the bits of the
 *                  serial oracle (sgemvSerial, src/SpMV_CSR_OMP.c:229-250) when the columns of every row ascend, as the
 *                  reference's loader guarantees (src/lib/parser.c:195-202).  1 = "owner wavefronts": every row is added by
 *                  ONE wavefront (local row mod 4 owns it) walking its own column-ordered sub-stream -- a layout of its own;
 *                  nearly free on matrices with column locality, 1.5-2x on uniformly spread columns (a gather covers
 *                  neighbours of a quarter of the bin's entries).  2 = "ordered tickets": the layout and the shared stream
 *                  of the arrival-order kernel (options 0 and 2 build the same format), but the batches add in ticket order,
 *                  handed over through an LDS word: 1.2x on uniformly spread columns, 1.5x where many lanes of an
 *                  instruction meet in a row (narrow bands).  hipSpMVRowsCSR's default measures both. */
typedef struct { unsigned rowsPerBin; unsigned grid; int spread; int wide; int deterministic; } spmvStripesOpts;
int spmvHipBuildStripesOpt(spmat* dMat, const spmvStripesOpts* opts);
typedef struct {
    unsigned nBins, rowsPerBin, grid, spread;
    int      wide, deterministic;
    double   buildMs;          /* device time of the one-time build */
    size_t   bytes;            /* device memory of the format */
} spmvStripesInfo;
int spmvHipStripesInfo(spmat* dMat, spmvStripesInfo* info);   /* zeros when the format has not been built */

/* The fastest CSR launcher for THIS matrix, chosen by measurement at the first call for a handle (the reference's
 * callers choose a kernel by name -- CUDA_CSR_ROWS_WARP, ... -- src/main.cu:103-139; which of this library's kernels
 * wins depends on where x lives relative to the caches, DESIGN.md sections 4, 7, 8).  Candidates: hipSpMVWarpPerRowCSR
 * always; from 2^18 entries on hipSpMVTilesCSR and, while x (N * 8 B) fits the 256 MiB Infinity Cache,
 * hipSpMVStripesCSR.  The first call runs every eligible candidate on the caller's x (one launch that also builds its
 * format + 3 timed ones, a single one if it takes more than 2 ms; each leaves the complete y), keeps the fastest and frees the private formats of the others;
 * it synchronises the stream even after spmvHipSetSync(0).  Later calls go straight to the chosen launcher.
 * spmvHipAutoChoice: its name (NULL before the first call) and, if msPerCandidate != NULL, the three measured times in
 * ms in the order above (0 = not eligible / not tried; room for FOUR doubles, the fourth stays 0).  Arrival-order sums
 * when a format kernel wins. */
SPMV_HIP hipSpMVAutoCSR;
const char* spmvHipAutoChoice(spmat* dMat, double* msPerCandidate);
/* The same report for the selection hipSpMVRowsCSR (variant 2) makes among the SERIAL-ORDER kernels: "hipSpMVRowsCSR" (the
 * LDS-stream kernel, one thread per row), "hipSpMVTilesCSR(deterministic)", "hipSpMVStripesCSR(owner wavefronts)",
 * "hipSpMVStripesCSR(ordered tickets)"; FOUR times in that order.  Whichever is chosen, y is the bits of sgemvSerial. */
const char* spmvHipAutoChoiceRows(spmat* dMat, double* msPerCandidate);

/* SELL-C-sigma (C = 64 rows per slice = one wavefront, rows sorted by length inside 16 Ki-row
 * windows, column-major inside a slice) built on the device from an uploaded CSR handle at the
 * first call: the ELL-family kernel for matrices whose longest row makes plain ELL impossible
 * (the reference's loader refuses them, src/lib/parser.c:223-232).  One lane per
 * row, ascending-j order: bit-identical to the serial oracle for rows up to 256 entries;
 * longer rows are summed by a whole workgroup (shuffle tree). */
SPMV_HIP hipSpMVRowsSELL;
int    spmvHipBuildSell(spmat* dMat);
size_t spmvHipSellBytes(spmat* dMat);

/* Enqueue-only form of the two CSR launchers on an explicit stream (no timing
 * bracket, no synchronisation): warpPerRow = 0 -> hipSpMVRowsCSR semantics,
 * != 0 -> hipSpMVWarpPerRowCSR.  The current device must be the matrix'. */
int spmvHipEnqueueCSR(spmat* dMat, int warpPerRow, double* dX, double* dY, void* stream);
/* The same for the launcher hipSpMVAutoCSR / hipSpMVWarpPerRowCSR (variant 2) chose for this handle; the FIRST call for a
 * handle measures the candidates on `stream` and synchronises it (later calls only enqueue). */
int spmvHipEnqueueAuto(spmat* dMat, double* dX, double* dY, void* stream);
/* ... and for the launcher hipSpMVRowsCSR (variant 2) chose: serial-order sums, y bit-identical to sgemvSerial. */
int spmvHipEnqueueAutoRows(spmat* dMat, double* dX, double* dY, void* stream);

/* 1 when this device adds the lanes of one LDS atomic instruction that meet in an address in ascending lane order and runs a
 * wavefront's LDS operations in issue order -- measured by a probe kernel against host sums, cached; 0 otherwise, -1 before
 * spmvHipInit.  The deterministic forms of the two-phase and the stripes kernel give the serial oracle's bits BECAUSE of
 * this (it is observed behaviour of gfx950, not an ISA promise): hipSpMVRowsCSR's serial-order selection offers them only
 * where the probe returns 1 and otherwise stays with the kernel that sums a row in one thread. */
int spmvHipProbeLdsAtomicOrder(void);

/* Kernel variants behind each launcher (for A/B measurement; default = best):
 *   hipSpMVRowsCSR        0 = one thread walks its row in global memory: the plain restatement of
 *                             cudaSpMVRowsCSR (uncoalesced; 5-6x slower, profiles/r01_variants.md)
 *                         1 = LDS-stream kernel: coalesced span load, products parked in LDS, one
 *                             thread sums its row in ascending-j order (bit-identical to the serial oracle)
 *                         2 = (default) the fastest of this library's SERIAL-ORDER kernels for THIS matrix -- variant 1
 *                             and the deterministic forms of the two-phase and the stripes kernel (spmvTilesOpts /
 *                             spmvStripesOpts .deterministic) -- chosen by measurement at the first call for a handle,
 *                             like variant 2 of hipSpMVWarpPerRowCSR below (same first-call cost, same +12 B/nnz for
 *                             the winner's copy of the matrix; matrices below 2^18 entries go straight to variant 1).
 *                             Every candidate adds a row's products in ascending column order with the product rounded
 *                             first, so y is the bits of sgemvSerial whichever is chosen (for rows whose columns ascend,
 *                             which the reference's loader guarantees, src/lib/parser.c:195-202).  Reached by the
 *                             reference's names SpmvCUDA_CSRFuncs[0], CUDA_CSR_ROWS, spmvHipRowsCSR.
 *   hipSpMVWarpPerRowCSR  0 = one wavefront per row, __shfl_down tree (the reference kernel's intent, for
 *                             every row)
 *                         1 = LDS-stream kernel with the LDS segmented reduction for short rows and
 *                             wavefront-/workgroup-per-row sums for long ones
 *                         2 = (default) the fastest of this library's reduction-order kernels for THIS matrix --
 *                             variant 1, hipSpMVTilesCSR, hipSpMVStripesCSR -- chosen exactly as hipSpMVAutoCSR does:
 *                             the FIRST call for a handle runs the eligible candidates on the caller's x (each leaves
 *                             the complete y; matrices below 2^18 entries go straight to variant 1), keeps the
 *                             fastest and frees the others' formats.  Cost of that first call: a handful of SpMVs
 *                             plus the format builds (c3: ~0.15 s; c5: ~1-2 s, mostly allocation) and it synchronises
 *                             the stream; afterwards +12 B/nnz of device memory for the winner's copy of the matrix
 *                             (+ the shared 8 B/nnz product workspace when the two-phase kernel wins).  This is what
 *                             the reference's names reach: SpmvCUDA_CSRFuncs[SpmvCUDA_CSRFuncs_WarpPerRowIdx],
 *                             CUDA_CSR_ROWS_WARP, spmvHipWarpPerRowCSR (src/include/SpMV.h:130-134).  A caller that
 *                             wants the one-kernel behaviour of round 1/2 sets variant 1 (CLI: SPMV_VARIANT=1).
 *   hipSpMVRowsELLNNTransposed  0 = one thread walks its row of the row-major matrix in global memory: the plain
 *                             restatement of cudaSpMVRowsELLNNTransposed (a lane's loads lie a pitch apart: uncoalesced by
 *                             construction, the reference's slowest kernel); also what runs when CONFIG.blockSize is given
 *                         1 = (default) the same sums -- one thread adds its row's cells in ascending slot order, bit for
 *                             bit -- fed from a coalesced span parked in LDS (rows of up to 2048 slots; longer: variant 0)
 * Returns EXIT_FAILURE for an unknown (launcher, variant). */
int spmvHipSetVariant(const char* launcher, int variant);
/* Use the RL array for ELL early exit (1, default when RL was uploaded) or walk
 * all MAX_ROW_NZ slots like the reference's cudaSpMVRowsELL (0). */
int spmvHipSetEllRowLens(int useRowLens);
/* Matrices whose stored values are ALL THE SAME double -- MatrixMarket `pattern` files, which the loader fills with 1.0
 * (src/lib/parser.c:59-61): every graph of the DIMACS10 collection, among them asia_osm and channel-500x100x100-b050 of
 * the reference's report -- are recognised when a CSR matrix is uploaded or adopted (one pass over the values).  The CSR
 * kernels (LDS-stream, stripes, two-phase) then take the value from a register instead of streaming 8 B per entry: 4 B/nnz
 * of matrix traffic instead of 12.  y does not change by a bit (c * x[j] rounds as AS[j] * x[j] does).  Compared as bit
 * patterns.  spmvHipSetUnitValues(0) turns the recognition off for later uploads (A/B); spmvHipUnitValue: 1 and the value,
 * 0, or -1 for a bad handle. */
int spmvHipSetUnitValues(int on);
int spmvHipUnitValue(spmat* dMat, double* value);

/* SPMV_INTERF-compatible wrappers (host vectors in/out, matrix uploaded and
 * cached on first use, keyed by the host spmat address) so the GPU path can sit
 * in SpmvCSRFuncs[]-style tables next to the OpenMP variants (SpMV.h:146-159).
 * The cache entry also remembers the shape and the IRP/JA/AS/RL pointers it was
 * uploaded from and is re-uploaded when any of them differs; changing the values
 * INSIDE the same arrays is not seen: call spmvHipDropCache() before a cached
 * host matrix is modified or freed. */
int spmvHipRowsCSR(spmat* mat, double* x, CONFIG* cfg, double* y);
int spmvHipWarpPerRowCSR(spmat* mat, double* x, CONFIG* cfg, double* y);
int spmvHipRowsELL(spmat* mat, double* x, CONFIG* cfg, double* y);
int spmvHipWarpsPerRowELL(spmat* mat, double* x, CONFIG* cfg, double* y);
int spmvHipDropCache(void);

/* ------------------------------------------------------- events (measurement) */
int spmvHipEventCreate(void** ev);
int spmvHipEventDestroy(void* ev);
int spmvHipEventRecord(void* ev);                 /* on the stream set above */
int spmvHipEventElapsedMs(void* evStart, void* evStop, float* ms);  /* syncs on evStop */

/* ------------------------------------------------------------------ sharding */
/* nnz-balanced contiguous row blocks: bounds[p]..bounds[p+1] are the rows of
 * part p (bounds has nParts+1 entries).  IRP is a host row-pointer array. */
int spmvHipPartitionRows(const ulong* IRP, ulong M, int nParts, ulong* bounds);
/* Extract rows [r0,r1) of a host CSR matrix as a new host CSR (local IRP
 * rebased to 0, global column ids kept).  Free with freeSpmat(). */
spmat* spmvHipRowBlockCSR(const spmat* host, ulong r0, ulong r1);
/* After an all-gather of equally padded row blocks (dYPad = nParts blocks of
 * maxRows doubles, block p holding rows bounds[p]..bounds[p+1]) copy the rows
 * back to back into dY -- nParts device-to-device copies on the library stream. */
int spmvHipCompactRows(double* dY, const double* dYPad, const ulong* bounds, int nParts, ulong maxRows);
/* Single-process multi-device path (C drivers): shard, upload one block per
 * device, replicate x, run `mode` on every device concurrently, gather y with
 * an RCCL all-gather over xGMI.  bench.py uses one process per GPU instead and
 * calls the single-device entry points + torch.distributed (RCCL). */
int spmvHipShardCSR(spmat* host, int nDev, void** shardHandle);
/* ... with every device's rows cut into `groups` consecutive row groups (0 = automatic: 2 when nDev > 1): the
 * all-gather of group g runs on its own stream while group g+1 is computed. */
int spmvHipShardCSRGroups(spmat* host, int nDev, int groups, void** shardHandle);
/* mode 0: the kernel hipSpMVRowsCSR runs (variant 2: the fastest serial-order kernel per block; y bit-identical to the
 * 1-GPU result and to sgemvSerial); mode != 0: that of hipSpMVWarpPerRowCSR (variant 2: the fastest reduction-order kernel
 * per block).  Either selection is made by an untimed pass at the first call with that mode.  kernelSec = kernels of all row groups, slowest device; gatherSec = what the exchange adds after overlap. */
int spmvHipSpMVSharded(void* shardHandle, const double* hX, int mode, double* hY,
                       double* kernelSec, double* gatherSec);
int spmvHipShardFree(void* shardHandle);

/* ------------------------------------ peer windows: one process per GPU, xGMI */
/* bench.py's layout (one process per GPU) exchanges y with RCCL by default.  xGMI
 * is a point-to-point mesh, so the same exchange can also be written as every rank
 * PUSHING its rows straight into the other ranks' copies of y.  A "window" is a
 * device allocation other processes of the node can map: its 64-byte handle is
 * plain data the caller ships to them however it likes (bench.py: torch.distributed
 * all_gather_object).  New functionality -- the reference is single-GPU. */
#define SPMV_IPC_HANDLE_BYTES 64
#define SPMV_MAX_PEERS 15
int spmvHipWindowCreate(size_t bytes, void** dBase, unsigned char handle[SPMV_IPC_HANDLE_BYTES]);
int spmvHipWindowFree(void* dBase);
/* Map a window exported by ANOTHER process whose GPU is device `ownerDev` in this
 * process' numbering, readable/writable from the current device (peer access is
 * enabled when the devices differ).  Close before the owner frees it. */
int spmvHipWindowOpen(const unsigned char handle[SPMV_IPC_HANDLE_BYTES], int ownerDev, void** dPeer);
int spmvHipWindowClose(void* dPeer);
/* Copy bytes [offset, offset+bytes) of the own buffer `dSrcBase` to the same offset
 * of every peer mapping, one copy engine stream per peer, all ordered after what is
 * enqueued on the library stream so far; returns at once.  spmvHipPeerPushJoin()
 * makes the library stream wait for every push issued before it. */
int spmvHipPeerPush(const void* dSrcBase, size_t offset, size_t bytes, int nPeers, void* const* dPeerBases);
int spmvHipPeerPushJoin(void);

/* The two phases of hipSpMVTilesCSR as separate launches, for callers that overlap
 * the exchange of y with its production: Expand = phase 1 over the whole matrix;
 * Reduce = phase 2 for the bins [binBegin, binEnd) -- rows [binBegin*rowsPerBin,
 * min(binEnd*rowsPerBin, M)) -- stored to dY and, when nExtra > 0, also to the
 * nExtra further vectors dExtra[k] with the same row indexing (peer mappings: the
 * all-gather fused into the kernel as direct xGMI stores).  Enqueue-only when
 * spmvHipSetSync(0).  spmvHipTilesShape builds the format if needed. */
int spmvHipTilesShape(spmat* dMat, unsigned* nBins, unsigned* rowsPerBin);
/* Build options of the two-phase format (all zero / -1 = automatic).  They are arguments of the build, not
 * process state: spmvHipBuildTilesOpt(dMat, &opts) builds -- or REbuilds -- the format of this handle with them.
 *   rowsPerBin  0 or 64..20000.  Phase 2 finishes its bins in rounds of one workgroup per CU, and rows can only
 *               leave for the other ranks when their bin is finished: when the exchange is the longer part of a
 *               step, smaller bins (more rounds) let it start earlier, at the price of shorter tiles (N = 8 shard
 *               of c5: 1.19 ms automatic, 1.26 ms with half-size bins, 1.43 ms with quarter-size ones).
 *   taper       1: one round (one bin per CU) of quarter-height bins first and last, full-height bins between --
 *               the first round sets when rows start to travel, the last what is still to be sent when phase 2
 *               ends.  rowsPerBin of spmvHipTilesShape is then the height of the HIGHEST bin; spmvHipTilesBinRow
 *               gives the first row of any bin (bin == nBins: the row count).
 *   ntStore     1 / 0: phase 1 stores the products non-temporally / with the default policy; -1: by size
 *               (products that fit the 256 MiB Infinity Cache are kept there).
 *   chunk       entries per phase-1 work item, 0 or 4096..2^24.
 *   deterministic  1: every row is added by ONE wavefront in ascending column order (a bin is four sub-bins, each walked by
 *               one wavefront), so y is the same bits in every run and on any number of row shards, and -- products being
 *               rounded before they are added -- the bits of the serial oracle when the columns of every row ascend.
 *               Costs time: four wavefronts per CU instead of sixteen, tiles a quarter as long.  No
 *               tapered bins in this form. */
typedef struct { unsigned rowsPerBin; int taper; int ntStore; unsigned chunk; int deterministic; } spmvTilesOpts;
int spmvHipBuildTilesOpt(spmat* dMat, const spmvTilesOpts* opts);
typedef struct {
    unsigned nBins, rowsPerBin, nSlices;
    int      taper, ntStore;
    unsigned chunk;
    double   buildMs;          /* wall time of the one-time build, allocations included */
    size_t   bytes;            /* device memory of the format (the shared product workspace not included) */
    double   allocMs;          /* the part of buildMs the host spent in hipMalloc: format, product workspace, temporaries */
    size_t   tempBytes;        /* peak of the temporaries of the build (12 B per entry + the tile tables; freed when it returns) */
    int      deterministic;
} spmvTilesInfo;
int spmvHipTilesInfo(spmat* dMat, spmvTilesInfo* info);   /* zeros when the format has not been built */
int spmvHipTilesBinRow(spmat* dMat, unsigned bin, ulong* firstRow);
int hipSpMVTilesExpand(spmat* dMat, double* dX);
int hipSpMVTilesReduce(spmat* dMat, unsigned binBegin, unsigned binEnd, double* dY, int nExtra, double* const* dExtra);
/* Phase 2 over all bins with a PUSH KERNEL beside it (own high-priority stream, no LDS, a few wavefronts per CU):
 * every reduction workgroup sets a per-bin flag when its rows of y are stored (agent-scope release), the push
 * kernel copies each flagged bin to the nExtra destinations.  The reduction never waits for a link and the links
 * work from the first finished bin on.  With spmvHipSetSync(0) the library stream does NOT wait for the push
 * kernel (the next row group's phase 1 may run while these rows still travel): spmvHipTilesPushJoin() makes it
 * wait for every push kernel enqueued so far; in synchronous mode the call returns with everything delivered.
 * A flag that does not arrive within ~2 s makes the push kernel give up (spmvHipTilesPushFailed() == 1, y
 * incomplete) rather than hang. */
int hipSpMVTilesReducePush(spmat* dMat, double* dY, int nExtra, double* const* dExtra);
int spmvHipTilesPushJoin(void);
int spmvHipTilesPushFailed(spmat* dMat);

/* ------------------------------------------------- synthetic matrices on device */
/* Fill JA/AS of a CSR whose row pointers are given (device arrays, device
 * format), for global rows [rowOffset, rowOffset+M): DESIGN.md "Synthetic
 * inputs".  band == 0: columns stratified-uniform over [0,N); band > 0: columns
 * stratified over [r-band, r+band] clipped to [0,N). */
int spmvHipSynthFillCSR(ulong M, ulong N, ulong rowOffset, const void* dIRP, int irpBytes,
                        uint32_t* dJA, double* dAS, uint64_t seedStruct, uint64_t seedVal,
                        ulong band);
/* (the dense vector x is generated on the host -- it needs sin(), whose last
 * bit differs between libm and the device -- and uploaded with spmvHipVecUp) */

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_H */
