/* modes.c -- exact-match lookup of the CLI compute-mode strings (the reference
 * matches by prefix and orders its tests longest-first, main.cu:103-119). */
#include <string.h>
#include "SpMV.h"

COMPUTE_MODE spmvModeFromString(const char* s) {
    static const struct { const char* name; COMPUTE_MODE mode; } table[] = {
        {CSR_ROWS, _CSR_ROWS}, {CSR_ROWS_GROUPS, _CSR_ROWS_GROUPS}, {CSR_TILES, _CSR_TILES},
        {CSR_TILES_ALLOCD, _CSR_TILES_ALLOCD}, {ELL_ROWS, _ELL_ROWS}, {ELL_ROWS_GROUPS, _ELL_ROWS_GROUPS},
        {ELL_TILES, _ELL_TILES},
        {CUDA_CSR_ROWS, _CUDA_CSR_ROWS}, {CUDA_CSR_ROWS_WARP, _CUDA_CSR_ROWS_WARP},
        {CUDA_ELL_ROWS, _CUDA_ELL_ROWS}, {CUDA_ELL_ROWS_NT, _CUDA_ELL_ROWS_NT},
        {CUDA_ELL_ROWS_WARP, _CUDA_ELL_ROWS_WARP}, {CUDA_ELL_ROWS_WARP_NT, _CUDA_ELL_ROWS_WARP_NT},
        {CUDA_CSR_TILES, _CUDA_CSR_TILES}, {HIP_CSR_TILES, _CUDA_CSR_TILES},
        {CUDA_SELL_ROWS, _CUDA_SELL_ROWS}, {HIP_SELL_ROWS, _CUDA_SELL_ROWS},
        {CUDA_CSR_STRIPES, _CUDA_CSR_STRIPES}, {HIP_CSR_STRIPES, _CUDA_CSR_STRIPES},
        {CUDA_CSR_AUTO, _CUDA_CSR_AUTO}, {HIP_CSR_AUTO, _CUDA_CSR_AUTO},
        {HIP_CSR_ROWS, _CUDA_CSR_ROWS}, {HIP_CSR_ROWS_WARP, _CUDA_CSR_ROWS_WARP},
        {HIP_ELL_ROWS, _CUDA_ELL_ROWS}, {HIP_ELL_ROWS_NT, _CUDA_ELL_ROWS_NT},
        {HIP_ELL_ROWS_WARP_NT, _CUDA_ELL_ROWS_WARP_NT},
    };
    if (!s) return _COMPUTE_MODE_INVALID;
    for (size_t i = 0; i < sizeof table / sizeof *table; ++i)
        if (!strcmp(s, table[i].name)) return table[i].mode;
    return _COMPUTE_MODE_INVALID;
}
