/*
 * SpMV.h -- dispatch surface: mode names, the SPMV function-pointer type and the
 * implementation tables the drivers iterate over.  Mirrors the reference's
 * src/include/SpMV.h:27-64,118-159 so its drivers and log tooling keep working;
 * the CUDA __global__ pointers are replaced by host-callable HIP launchers
 * (a plain-C host cannot use <<<>>>), see spmvHip.h.
 */
#ifndef SPMV_DISPATCH_H
#define SPMV_DISPATCH_H

#include "spmv_types.h"
#include "sparseMatrix.h"
#include "spmvHip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* argv[3] of the CLI.  The reference matches these by PREFIX and therefore has
 * to test the longer names first (macros.h:43, main.cu:105-118); here they are
 * matched exactly.  The CUDA_* spellings stay valid and select the HIP path;
 * HIP_* are synonyms. */
#define CSR_ROWS                "CSR_ROWS"
#define CSR_ROWS_GROUPS         "CSR_ROWS_GROUPS"
#define CSR_TILES               "CSR_TILES"
#define CSR_TILES_ALLOCD        "CSR_TILES_ALLOCD"
#define ELL_ROWS                "ELL_ROWS"
#define ELL_ROWS_GROUPS         "ELL_ROWS_GROUPS"
#define ELL_TILES               "ELL_TILES"
#define CUDA_CSR_ROWS           "CUDA_CSR_ROWS"
#define CUDA_CSR_ROWS_WARP      "CUDA_CSR_ROWS_WARP"
#define CUDA_ELL_ROWS           "CUDA_ELL_ROWS"
#define CUDA_ELL_ROWS_NT        "CUDA_ELL_ROWS_NN_TRANSPOSED"      /* in the reference's table but not selectable from its CLI */
#define CUDA_ELL_ROWS_WARP      "CUDA_ELL_ROWS_WARP"
#define CUDA_ELL_ROWS_WARP_NT   "CUDA_ELL_ROWS_WARP_NN_TRANSPOSED"
#define CUDA_CSR_TILES          "CUDA_CSR_TILES"                   /* new: column-sliced two-phase kernel */
#define HIP_CSR_TILES           "HIP_CSR_TILES"
#define CUDA_SELL_ROWS          "CUDA_SELL_ROWS"                   /* new: SELL-C-sigma, one lane per row */
#define HIP_SELL_ROWS           "HIP_SELL_ROWS"
#define CUDA_CSR_STRIPES        "CUDA_CSR_STRIPES"                 /* new: one-pass stripes kernel (y bins in LDS, x from the XCD's L2) */
#define HIP_CSR_STRIPES         "HIP_CSR_STRIPES"
#define CUDA_CSR_AUTO           "CUDA_CSR_AUTO"                    /* new: the fastest CSR launcher for the matrix, by measurement */
#define HIP_CSR_AUTO            "HIP_CSR_AUTO"
#define HIP_CSR_ROWS            "HIP_CSR_ROWS"
#define HIP_CSR_ROWS_WARP       "HIP_CSR_ROWS_WARP"
#define HIP_ELL_ROWS            "HIP_ELL_ROWS"
#define HIP_ELL_ROWS_NT         "HIP_ELL_ROWS_NN_TRANSPOSED"
#define HIP_ELL_ROWS_WARP_NT    "HIP_ELL_ROWS_WARP_NN_TRANSPOSED"

typedef enum {              /* same order and values as SpMV.h:42-59 */
    _CSR_SORTED_ROWS,
    _CSR_ROWS,
    _CSR_ROWS_GROUPS,
    _CSR_TILES,
    _CSR_TILES_ALLOCD,
    _ELL_ROWS,
    _ELL_ROWS_GROUPS,
    _ELL_TILES,
    _CUDA_CSR_ROWS,
    _CUDA_CSR_ROWS_WARP,
    _CUDA_ELL_ROWS,
    _CUDA_ELL_ROWS_WARP,
    _CUDA_ELL_ROWS_WARP_NT,
    _CUDA_ELL_ROWS_NT,      /* appended: row-major thread-per-row ELL */
    _CUDA_CSR_TILES,        /* appended: column-sliced two-phase CSR */
    _CUDA_SELL_ROWS,        /* appended: SELL-C-sigma built from the CSR upload */
    _CUDA_CSR_STRIPES,      /* appended: one-pass stripes CSR */
    _CUDA_CSR_AUTO,         /* appended: hipSpMVAutoCSR */
    _COMPUTE_MODE_INVALID = -1
} COMPUTE_MODE;

/* exact-match lookup; _COMPUTE_MODE_INVALID when `name` is unknown */
COMPUTE_MODE spmvModeFromString(const char* name);
static inline int spmvModeIsGpu(COMPUTE_MODE m) { return m >= _CUDA_CSR_ROWS; }
static inline int spmvModeIsCsr(COMPUTE_MODE m) {
    return m <= _CSR_TILES_ALLOCD || m == _CUDA_CSR_ROWS || m == _CUDA_CSR_ROWS_WARP || m == _CUDA_CSR_TILES ||
           m == _CUDA_SELL_ROWS || m == _CUDA_CSR_STRIPES || m == _CUDA_CSR_AUTO;
}

/* y = A x: (matrix, x, run configuration, y) -> EXIT_SUCCESS / EXIT_FAILURE */
typedef int (SPMV)(spmat*, double*, CONFIG*, double*);
typedef int (*SPMV_INTERF)(spmat*, double*, CONFIG*, double*);

/* ---- GPU tables (SpMV.h:130-142), elements are the HIP launchers ---------- */
static const SPMV_HIP_INTERF SpmvCUDA_CSRFuncs[] = {
    &hipSpMVRowsCSR,
    &hipSpMVWarpPerRowCSR,
    &hipSpMVTilesCSR,           /* appended: no counterpart in the reference's GPU table */
    &hipSpMVRowsSELL,           /* appended: SELL-C-sigma copy of the CSR matrix */
    &hipSpMVStripesCSR,         /* appended: one-pass stripes kernel */
    &hipSpMVAutoCSR,            /* appended: the fastest of the above for the matrix at hand */
};
#define SpmvCUDA_CSRFuncs_WarpPerRowIdx     1
#define SpmvCUDA_CSRFuncs_TilesIdx          2
#define SpmvCUDA_CSRFuncs_SellIdx           3
#define SpmvCUDA_CSRFuncs_StripesIdx        4
#define SpmvCUDA_CSRFuncs_AutoIdx           5
static const SPMV_HIP_INTERF SpmvCUDA_ELLFuncs[] = {
    &hipSpMVRowsELL,
    &hipSpMVRowsELLNNTransposed,
    &hipSpMVWarpsPerRowELLNTrasposed,
};
#define SpmvCUDA_ELLFuncs_NN_TraposedImpl   1
#define SpmvCUDA_ELLFuncs_WarpPerRowIdx     2
#define SpmvHIP_CSRFuncs SpmvCUDA_CSRFuncs
#define SpmvHIP_ELLFuncs SpmvCUDA_ELLFuncs

/* ---- CPU side ---------------------------------------------------------------
 * The serial oracle and the OpenMP row-parallel baselines are NOT part of the
 * shipped GPU library: they live under oracle/ (checker + CPU baseline only).
 * Programs that link them (the parity harness under tests/) define
 * SPMV_WITH_OMP_TABLES to get the prototypes and the reference-shaped tables. */
#ifdef SPMV_WITH_OMP_TABLES
SPMV sgemvSerial;           /* SpMV_CSR_OMP.c:229-250 */
SPMV spmvRowsBasicCSR;      /* SpMV_CSR_OMP.c:34-63   */
SPMV spmvRowsBasicELL;      /* SpMV_ELL_OMP.c:33-67   */
SPMV spmvRowsBlocksCSR;     /* SpMV_CSR_OMP.c:65-99   -- CPU comparators, SURVEY 8f-1 */
SPMV spmvTilesCSR;          /* SpMV_CSR_OMP.c:101-162 */
SPMV spmvTilesAllocdCSR;    /* SpMV_CSR_OMP.c:165-226 */
SPMV spmvRowsBlocksELL;     /* SpMV_ELL_OMP.c:69-108  */
SPMV spmvTilesELL;          /* SpMV_ELL_OMP.c:110-174 */
/* same order as the reference's tables (SpMV.h:146-159) */
static const SPMV_INTERF SpmvCSRFuncs[] = { &sgemvSerial, &spmvRowsBasicCSR, &spmvRowsBlocksCSR, &spmvTilesCSR,
                                            &spmvTilesAllocdCSR };
static const SPMV_INTERF SpmvELLFuncs[] = { &spmvRowsBasicELL, &spmvRowsBlocksELL, &spmvTilesELL };
#endif

#ifdef __cplusplus
}
#endif
#endif /* SPMV_DISPATCH_H */
