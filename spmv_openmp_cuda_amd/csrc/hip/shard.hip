// shard.hip -- single-process, multi-device SpMV for the plain-C drivers:
// nnz-balanced contiguous row blocks, x replicated, one RCCL all-gather of y
// over xGMI.  New functionality: the reference is single-GPU
// (no MPI/NCCL/cudaSetDevice anywhere in it).
//
// bench.py does NOT use this file: it runs one process per GPU and gathers with
// torch.distributed (RCCL).  RCCL is loaded lazily with dlopen so that
// libspmvhip.so carries no link-time dependency on it (a Python process that
// imported torch already holds torch's own librccl).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

extern "C" int spmvHipEnqueueCSR(spmat* dMat, int warpPerRow, double* dX, double* dY, void* stream);

namespace {

// the handful of RCCL entry points used, with the signatures of <rccl/rccl.h>
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclFloat64 = 8;      // ncclDouble in ncclDataType_t
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (so) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
        }
        if (!so) { fprintf(stderr, "libspmvhip: cannot load RCCL: %s\n", dlerror()); return false; }
#define SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(so, sym)); if (!field) { fprintf(stderr, "libspmvhip: RCCL lacks %s\n", sym); return false; }
        SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd") SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        return true;
    }
} rccl;

struct Shard {
    int nDev = 0;
    ulong M = 0, N = 0, maxRows = 0;
    std::vector<ulong> bounds;
    std::vector<spmat> mats;
    std::vector<double*> dX, dYpad;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> ev0, ev1, ev2;
    std::vector<ncclComm_t> comms;
    bool useRccl = false;
};

#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != 0) { fprintf(stderr, "libspmvhip: %s: %s\n", #expr, rccl.GetErrorString(r_)); return EXIT_FAILURE; } } while (0)

}  // namespace

extern "C" {

int spmvHipShardFree(void* handle) {
    Shard* sh = static_cast<Shard*>(handle);
    if (!sh) return EXIT_SUCCESS;
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (int d = 0; d < sh->nDev; ++d) {
        (void)hipSetDevice(d);
        if (d < (int)sh->mats.size()) hipFreeSpmat(&sh->mats[d]);
        if (d < (int)sh->dX.size()) (void)hipFree(sh->dX[d]);
        if (d < (int)sh->dYpad.size()) (void)hipFree(sh->dYpad[d]);
        if (d < (int)sh->ev0.size()) { (void)hipEventDestroy(sh->ev0[d]); (void)hipEventDestroy(sh->ev1[d]); (void)hipEventDestroy(sh->ev2[d]); }
        if (d < (int)sh->streams.size()) (void)hipStreamDestroy(sh->streams[d]);
        if (d < (int)sh->comms.size() && sh->comms[d]) rccl.CommDestroy(sh->comms[d]);
    }
    (void)hipSetDevice(keep);
    delete sh;
    return EXIT_SUCCESS;
}

int spmvHipShardCSR(spmat* host, int nDev, void** shardHandle) {
    if (!host || !shardHandle || nDev <= 0) return EXIT_FAILURE;
    int visible = spmvHipDeviceCount();
    if (nDev > visible) { fprintf(stderr, "libspmvhip: spmvHipShardCSR: %d devices requested, %d visible\n", nDev, visible); return EXIT_FAILURE; }
    int keep = 0;
    HIP_TRY(hipGetDevice(&keep));
    Shard* sh = new Shard;
    sh->nDev = nDev; sh->M = host->M; sh->N = host->N;
    sh->bounds.resize(nDev + 1);
    if (spmvHipPartitionRows(host->IRP, host->M, nDev, sh->bounds.data())) { delete sh; return EXIT_FAILURE; }
    for (int d = 0; d < nDev; ++d) sh->maxRows = std::max(sh->maxRows, sh->bounds[d + 1] - sh->bounds[d]);
    const char* force = getenv("SPMV_SHARD_FORCE_RCCL");
    sh->useRccl = nDev > 1 || (force && *force == '1');
    sh->mats.resize(nDev); sh->dX.assign(nDev, nullptr); sh->dYpad.assign(nDev, nullptr);
    sh->streams.assign(nDev, nullptr); sh->ev0.assign(nDev, nullptr); sh->ev1.assign(nDev, nullptr); sh->ev2.assign(nDev, nullptr);
    for (auto& m : sh->mats) memset(&m, 0, sizeof m);
    int rc = EXIT_SUCCESS;
    for (int d = 0; d < nDev && !rc; ++d) {
        if (hipSetDevice(d) != hipSuccess) { rc = EXIT_FAILURE; break; }
        spmat* blk = spmvHipRowBlockCSR(host, sh->bounds[d], sh->bounds[d + 1]);
        if (!blk) { rc = EXIT_FAILURE; break; }
        rc = spMatCpyCSR(blk, &sh->mats[d]);
        free(blk->IRP); free(blk->JA); free(blk->AS); free(blk->RL); free(blk);
        if (rc) break;
        if (hipMalloc(&sh->dX[d], std::max<size_t>(sh->N, 1) * sizeof(double)) != hipSuccess ||
            hipMalloc(&sh->dYpad[d], std::max<size_t>((size_t)nDev * sh->maxRows, 1) * sizeof(double)) != hipSuccess ||
            hipStreamCreateWithFlags(&sh->streams[d], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&sh->ev0[d]) != hipSuccess || hipEventCreate(&sh->ev1[d]) != hipSuccess ||
            hipEventCreate(&sh->ev2[d]) != hipSuccess)
            rc = EXIT_FAILURE;
    }
    if (!rc && sh->useRccl) {
        if (!rccl.load()) rc = EXIT_FAILURE;
        else {
            sh->comms.assign(nDev, nullptr);
            std::vector<int> devs(nDev);
            for (int d = 0; d < nDev; ++d) devs[d] = d;
            ncclResult_t r = rccl.CommInitAll(sh->comms.data(), nDev, devs.data());
            if (r != 0) { fprintf(stderr, "libspmvhip: ncclCommInitAll: %s\n", rccl.GetErrorString(r)); rc = EXIT_FAILURE; }
        }
    }
    (void)hipSetDevice(keep);
    if (rc) { spmvHipShardFree(sh); return EXIT_FAILURE; }
    *shardHandle = sh;
    return EXIT_SUCCESS;
}

// mode: 0 = thread-per-row semantics (hipSpMVRowsCSR), 1 = wavefront semantics (hipSpMVWarpPerRowCSR)
int spmvHipSpMVSharded(void* handle, const double* hX, int mode, double* hY, double* kernelSec, double* gatherSec) {
    Shard* sh = static_cast<Shard*>(handle);
    if (!sh || !hX || !hY) return EXIT_FAILURE;
    int keep = 0;
    HIP_TRY(hipGetDevice(&keep));
    const int n = sh->nDev;
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipMemcpyAsync(sh->dX[d], hX, sh->N * sizeof(double), hipMemcpyHostToDevice, sh->streams[d]));
        HIP_TRY(hipMemsetAsync(sh->dYpad[d], 0xFF, (size_t)n * sh->maxRows * sizeof(double), sh->streams[d]));   // NaN poison
        HIP_TRY(hipEventRecord(sh->ev0[d], sh->streams[d]));
        if (spmvHipEnqueueCSR(&sh->mats[d], mode != 0, sh->dX[d], sh->dYpad[d] + (size_t)d * sh->maxRows, sh->streams[d])) return EXIT_FAILURE;
        HIP_TRY(hipEventRecord(sh->ev1[d], sh->streams[d]));
    }
    if (sh->useRccl) {
        NCCL_TRY(rccl.GroupStart());
        for (int d = 0; d < n; ++d)
            NCCL_TRY(rccl.AllGather(sh->dYpad[d] + (size_t)d * sh->maxRows, sh->dYpad[d], sh->maxRows, kNcclFloat64,
                                    sh->comms[d], sh->streams[d]));
        NCCL_TRY(rccl.GroupEnd());
    }
    double kmax = 0, gmax = 0;
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipEventRecord(sh->ev2[d], sh->streams[d]));
        HIP_TRY(hipEventSynchronize(sh->ev2[d]));
        float k = 0, g = 0;
        HIP_TRY(hipEventElapsedTime(&k, sh->ev0[d], sh->ev1[d]));
        HIP_TRY(hipEventElapsedTime(&g, sh->ev1[d], sh->ev2[d]));
        kmax = std::max<double>(kmax, k * 1e-3);
        gmax = std::max<double>(gmax, g * 1e-3);
    }
    // every device now holds all blocks; read them back from device 0, compacted
    HIP_TRY(hipSetDevice(0));
    for (int p = 0; p < n; ++p) {
        const ulong rows = sh->bounds[p + 1] - sh->bounds[p];
        if (rows) HIP_TRY(hipMemcpy(hY + sh->bounds[p], sh->dYpad[0] + (size_t)p * sh->maxRows, rows * sizeof(double), hipMemcpyDeviceToHost));
    }
    (void)hipSetDevice(keep);
    if (kernelSec) *kernelSec = kmax;
    if (gatherSec) *gatherSec = gmax;
    return EXIT_SUCCESS;
}

}  // extern "C"
