/*
 * oracle/ref_glue.c -- the few definitions the reference expects from its
 * DRIVER program, so that its own library sources can be linked into
 * oracle/_ref/libspmvref.so without either of its drivers (main.cu,
 * test/SpMV_test.cu):
 *   - the audit globals it declares `extern` in src/include/config.h:112 and
 *     defines in the drivers (src/main.cu:56, test/SpMV_test.cu:58);
 *   - out-of-line instances of its C99 `inline` helpers, which the drivers
 *     request exactly like this (test/SpMV_test.cu:40: `CHUNKS_DISTR
 *     chunksFair,chunksFairFolded,chunksNOOP;`); the alloc/free helpers already get
 *     theirs from src/commons/sparseUtils.c.
 * No reference source is copied: this file only #includes its headers from
 * /root/reference at build time (oracle/Makefile passes -I).
 */
#include <omp.h>
#include <stdlib.h>
#include <stdio.h>
#include "macros.h"
#include "sparseMatrix.h"
#include "SpMV.h"
#include "utils.h"
#include "ompChunksDivide.h"

double Start, End, Elapsed, ElapsedInternal;

extern inline void chunksNOOP(ulong r, spmat* mat, CONFIG* cfg);
extern inline void chunksFair(ulong r, spmat* mat, CONFIG* cfg);
extern inline void chunksFairFolded(ulong r, spmat* mat, CONFIG* cfg);
extern inline int BISECT_ARRAY(ulong target, ulong* arr, ulong len);
extern inline int IS_NNZ(spmat* smat, ulong i, ulong j);
extern inline int IS_NNZ_linear(spmat* smat, ulong i, ulong j);
extern inline void freeSpAcc(SPACC* r);

/* layout probes so the Python side can assert its ctypes mirror */
size_t refSizeofSpmat(void) { return sizeof(spmat); }
size_t refSizeofConfig(void) { return sizeof(CONFIG); }
void   refSetSchedule(int kind, int chunk) { omp_set_schedule((omp_sched_t)kind, chunk); }
int    refMaxThreads(void) { return omp_get_max_threads(); }
void*  refChunksNOOP(void) { return (void*)chunksNOOP; }
void*  refChunksFairFolded(void) { return (void*)chunksFairFolded; }
void*  refChunksFair(void) { return (void*)chunksFair; }
double refElapsedInternal(void) { return ElapsedInternal; }
