/*
 * spmv_types.h -- data model carried across the host <-> MI355X boundary.
 *
 * Names (spmat, CONFIG, ulong, IDX2D, DOUBLE_DIFF_THREASH, ...) follow the
 * reference's domain vocabulary so that its drivers keep compiling:
 *   spmat   <- reference src/include/sparseMatrix.h:25-42
 *   CONFIG  <- reference src/include/config.h:21-32
 *   helpers <- reference src/include/macros.h:24-66, config.h:33-36,83-121
 *
 * Differences that are deliberate (see DESIGN.md "Boundary"):
 *  - the struct layout does NOT depend on compile flags.  The reference adds
 *    `RL` under -DROWLENS and `pitchJA/pitchAS` under __CUDACC__, so the same
 *    name has four possible layouts; here every field is always present and
 *    "no row lengths" is RL == NULL at run time.  `spmvHipInit` receives
 *    sizeof(spmat)/sizeof(CONFIG) from the caller and refuses a mismatch.
 *  - CONFIG carries a plain 3 x unsigned launch shape instead of CUDA's dim3.
 */
#ifndef SPMV_TYPES_H
#define SPMV_TYPES_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* short unsigned aliases used all over the reference sources */
typedef unsigned char  uchar;
typedef unsigned short ushort;
typedef unsigned int   uint;
typedef unsigned long  ulong;     /* all host-side indices are 64-bit */

#ifndef FALSE
#define FALSE 0
#endif
#ifndef TRUE
#define TRUE  (!FALSE)
#endif

/* ---- small arithmetic helpers ------------------------------------------ */
#define ABS(a)                 ((a) > 0 ? (a) : -(a))
#define MIN(a, b)              ((a) < (b) ? (a) : (b))
#define MAX(a, b)              ((a) > (b) ? (a) : (b))
#define INT_DIV_CEIL(x, y)     (((x) - 1) / (y) + 1)
/* row-major 2-D index: element (i,j) of a matrix with nCols columns */
#define IDX2D(i, j, nCols)     ((j) + (i) * (nCols))
/* split `n = div*parts + rem` fairly: first `rem` parts get one extra */
#define UNIF_REMINDER_DISTRI(i, div, rem)          ((div) + ((i) < (rem) ? 1 : 0))
#define UNIF_REMINDER_DISTRI_STARTIDX(i, div, rem) ((i) * (div) + MIN((i), (rem)))
#define STATIC_ARR_ELEMENTS_N(arr) (sizeof((arr)) / sizeof(*(arr)))
#define _STRIFY(x) #x
#define STRIFY(x)  _STRIFY(x)

/* ---- diagnostics -------------------------------------------------------- */
#define CEND            "\33[0m"
#define CCC             "\33[1m\33[92m"
#define CCCERR          "\33[31m\33[1m\33[44m"
#define hprintf(str)            printf(CCC str CEND)
#define hprintsf(str, ...)      printf(CCC str CEND, __VA_ARGS__)
#define ERRPRINT(str)           fprintf(stderr, CCCERR str CEND)
#define ERRPRINTS(str, ...)     fprintf(stderr, CCCERR str CEND, __VA_ARGS__)

/* The reference gates blocks with statement-prefix macros (config.h:37-62).
 * Same spelling, same defaults, overridable with -D. */
#ifndef DEBUG
#define DEBUG               if (FALSE)
#endif
#ifndef DEBUGPRINT
#define DEBUGPRINT          if (FALSE)
#endif
#ifndef DEBUGCHECKS
#define DEBUGCHECKS         if (FALSE)
#endif
#ifndef VERBOSE
#define VERBOSE             if (FALSE)
#endif
#ifndef CONSISTENCY_CHECKS
#define CONSISTENCY_CHECKS  if (TRUE)
#endif
#ifndef AUDIT_INTERNAL_TIMES
#define AUDIT_INTERNAL_TIMES if (TRUE)
#endif

/* ---- numeric contract --------------------------------------------------- */
#define DOUBLE_DIFF_THREASH   7e-4      /* abs per-element gate, config.h:113 */
#define MAXRND                3e-5      /* |x_i| envelope,      config.h:115 */
#define DRNG_DEVFILE          "/dev/urandom"

/* ---- ELL size guard (parser.c:223-232, config.h:69-70) ------------------ */
#define ELL_MAX_ENTRIES       (6l << 27)
#ifndef SPMV_NO_ELL_LIMIT
#define LIMIT_ELL_SIZE
#endif

/* ---- timing / launch defaults ------------------------------------------- */
#ifndef AVG_TIMES_ITERATION
#define AVG_TIMES_ITERATION   5
#endif
#ifndef FAIR_CHUNKS_FOLDING
#define FAIR_CHUNKS_FOLDING   4
#endif
#ifndef SIMD_ROWS_REDUCTION
#define SIMD_ROWS_REDUCTION   TRUE
#endif
#ifndef BLOCKS_1D
#define BLOCKS_1D             (1u << 8)     /* threads per workgroup, 1-D kernels */
#endif
#ifndef BLOCKS_2D_WARP_R
#define BLOCKS_2D_WARP_R      (1u << 2)     /* wavefronts (rows) per workgroup    */
#endif
#define WAVESIZE              64            /* CDNA wavefront; the reference's WARPSIZE is 32 */

/* ---- scratch files kept for CLI compatibility (config.h:75-81,119-120) -- */
#ifndef TMPDIR
#define TMPDIR                "/tmp/"
#endif
#define RNDVECTORSIZE         100000
#define VECTOR_STEP_REALLOC   25
#define VECTOR_READ_BLOCK     50
#define RNDVECTORDUMP         TMPDIR "rndVectorDump"
#define RNDVECTORDUMPRAW      TMPDIR "rndVectorDumpRaw"
#define OUTVECTORDUMP         TMPDIR "outVectorDump"
#define OUTVECTORDUMPRAW      TMPDIR "outVectorDumpRaw"
#define TMP_EXTRACTED_MARTIX  TMPDIR "extractedMatrix"

/* wall-clock audit globals: defined once by the driver program, written by the
 * SpMV entry points (config.h:112).  GPU entry points store device seconds. */
extern double Start, End, Elapsed, ElapsedInternal;

/*
 * Sparse matrix, CSR or ELL (sparseMatrix.h:25-42).
 *   CSR : JA[NZ] column ids ascending per row, IRP[M+1] row pointers, AS[NZ]
 *   ELL : JA/AS are M x MAX_ROW_NZ row-major, padding {JA=0, AS=0.0}, IRP NULL
 *   RL  : per-row nnz count or NULL
 * A *device handle* is the same struct living in host memory whose `dev`
 * member points at the library's descriptor; its JA/AS/IRP/RL then hold device
 * addresses in the device's own (narrowed) index width and must not be
 * dereferenced by the host.
 */
typedef struct {
    ulong   NZ, M, N;
    ulong  *JA;
    ulong  *RL;
    ulong  *IRP;
    ulong   MAX_ROW_NZ;
    double *AS;
    size_t  pitchJA;     /* ELL on device: row pitch in ELEMENTS (cudaUtils.cu:81-83) */
    size_t  pitchAS;
    void   *dev;         /* opaque; NULL for host matrices */
} spmat;

/* ellTranspose() marks its output in the otherwise unused `dev` field of a HOST
 * matrix so that spMatCpyELL() knows which of M / MAX_ROW_NZ is the row count
 * (the reference re-uses the fields with swapped meaning, sparseUtils.c:168-171) */
#define SPMAT_TAG_ELL_TRANSPOSED ((void*)(uintptr_t)0x454C4C54u)

/* launch shape, laid out like CUDA's dim3 */
typedef struct { unsigned x, y, z; } spmvDim3;

/*
 * Run configuration (config.h:21-32).  gridRows/gridCols/threadNum/
 * chunkDistrbFunc drive the OpenMP variants; gridSize/blockSize are the GPU
 * launch shape the reference computes in its drivers (main.cu:221-226).  For
 * the HIP entry points a zero blockSize.x means "library default"; gridSize is
 * always derived by the library (the reference's x-only grid for its 2-D
 * blocks is the defect described in DESIGN.md) and is reported back through
 * spmvHipLastLaunch().
 */
typedef struct {
    ushort   gridRows;
    ushort   gridCols;
    uint     threadNum;
    void    *chunkDistrbFunc;
    spmvDim3 gridSize;
    spmvDim3 blockSize;
    size_t   sharedMemSize;
} CONFIG;

#ifdef __cplusplus
}
#endif
#endif /* SPMV_TYPES_H */
