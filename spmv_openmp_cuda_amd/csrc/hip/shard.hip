// shard.hip -- single-process, multi-device SpMV for the plain-C drivers:
// nnz-balanced contiguous row blocks, x replicated, RCCL all-gather of y over
// xGMI.  New functionality: the reference is single-GPU
// (no MPI/NCCL/cudaSetDevice anywhere in it).
//
// Every device's rows are cut into G consecutive row GROUPS (nDev x G blocks in
// all, still balanced by nnz).  A step runs, per device,
//     kernel(g) on the compute stream -> all-gather(g) on the gather stream
// for g = 0..G-1, so the gather of group g travels while group g+1 is computed
// (xGMI is point-to-point: at 2..8 devices the gather costs about as much as the
// kernel).  mode 1 runs the kernel hipSpMVWarpPerRowCSR would (the fastest
// reduction-order kernel for each block, measured at the first step); mode 0 the
// serial-order kernel of hipSpMVRowsCSR (bit-identical to the 1-GPU result).
//
// bench.py does NOT use this file: it runs one process per GPU and gathers with
// torch.distributed (RCCL) or peer windows.  RCCL is loaded lazily with dlopen so
// that libspmvhip.so carries no link-time dependency on it (a Python process
// that imported torch already holds torch's own librccl).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

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

// the calling thread's current device is put back on every way out
struct DeviceGuard {
    int keep = -1;
    DeviceGuard() { if (hipGetDevice(&keep) != hipSuccess) keep = -1; }
    ~DeviceGuard() { if (keep >= 0) (void)hipSetDevice(keep); }
};

struct Shard {
    int nDev = 0, G = 1;
    ulong M = 0, N = 0;
    std::vector<ulong> bounds;               // [nDev*G + 1]; block k = device k / G, group k % G
    std::vector<ulong> maxRows;              // [G] largest block of the group over the devices
    std::vector<spmat> mats;                 // [nDev*G]
    std::vector<double*> dX;                 // [nDev]
    std::vector<double*> dYpad;              // [nDev*G]: nDev * maxRows[g] doubles; device d's rows of group g sit in slot d
    std::vector<hipStream_t> compute, gather;
    std::vector<hipEvent_t> evStart, evEnd;  // [nDev] first kernel starts / last gather done
    std::vector<hipEvent_t> evK0, evK1;      // [nDev*G] around each group's kernel
    std::vector<ncclComm_t> comms;
    bool useRccl = false;
    bool selected[2] = {false, false};       // per mode: the per-block kernel selection has run
    ulong lo(int d, int g) const { return bounds[(size_t)d * G + g]; }
    ulong hi(int d, int g) const { return bounds[(size_t)d * G + g + 1]; }
};

#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != 0) { fprintf(stderr, "libspmvhip: %s: %s\n", #expr, rccl.GetErrorString(r_)); return EXIT_FAILURE; } } while (0)

int shardStep(Shard* sh, const double* hX, int mode, bool timed) {
    const int n = sh->nDev, G = sh->G;
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipMemcpyAsync(sh->dX[d], hX, sh->N * sizeof(double), hipMemcpyHostToDevice, sh->compute[d]));
        for (int g = 0; g < G; ++g)
            HIP_TRY(hipMemsetAsync(sh->dYpad[(size_t)d * G + g], 0xFF, (size_t)n * std::max<ulong>(sh->maxRows[g], 1) * sizeof(double), sh->compute[d]));   // NaN poison
        if (timed) HIP_TRY(hipEventRecord(sh->evStart[d], sh->compute[d]));
    }
    for (int g = 0; g < G; ++g) {
        const ulong pad = std::max<ulong>(sh->maxRows[g], 1);
        for (int d = 0; d < n; ++d) {
            HIP_TRY(hipSetDevice(d));
            const size_t k = (size_t)d * G + g;
            double* slot = sh->dYpad[k] + (size_t)d * pad;
            if (timed) HIP_TRY(hipEventRecord(sh->evK0[k], sh->compute[d]));
            const int rc = mode ? spmvHipEnqueueAuto(&sh->mats[k], sh->dX[d], slot, sh->compute[d])
                                : spmvHipEnqueueAutoRows(&sh->mats[k], sh->dX[d], slot, sh->compute[d]);
            if (rc) return EXIT_FAILURE;
            HIP_TRY(hipEventRecord(sh->evK1[k], sh->compute[d]));
            HIP_TRY(hipStreamWaitEvent(sh->gather[d], sh->evK1[k], 0));      // (the first group's wait also orders the poison before the gather)
        }
        if (sh->useRccl) {
            NCCL_TRY(rccl.GroupStart());
            for (int d = 0; d < n; ++d) {
                const size_t k = (size_t)d * G + g;
                NCCL_TRY(rccl.AllGather(sh->dYpad[k] + (size_t)d * pad, sh->dYpad[k], pad, kNcclFloat64, sh->comms[d], sh->gather[d]));
            }
            NCCL_TRY(rccl.GroupEnd());
        }
    }
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipEventRecord(sh->evEnd[d], sh->gather[d]));
        HIP_TRY(hipEventSynchronize(sh->evEnd[d]));
    }
    return EXIT_SUCCESS;
}

}  // namespace

extern "C" {

int spmvHipShardFree(void* handle) {
    Shard* sh = static_cast<Shard*>(handle);
    if (!sh) return EXIT_SUCCESS;
    DeviceGuard guard;
    for (int d = 0; d < sh->nDev; ++d) {
        (void)hipSetDevice(d);
        for (int g = 0; g < sh->G; ++g) {
            const size_t k = (size_t)d * sh->G + g;
            if (k < sh->mats.size()) hipFreeSpmat(&sh->mats[k]);
            if (k < sh->dYpad.size()) (void)hipFree(sh->dYpad[k]);
            if (k < sh->evK0.size() && sh->evK0[k]) (void)hipEventDestroy(sh->evK0[k]);
            if (k < sh->evK1.size() && sh->evK1[k]) (void)hipEventDestroy(sh->evK1[k]);
        }
        if (d < (int)sh->dX.size()) (void)hipFree(sh->dX[d]);
        if (d < (int)sh->evStart.size() && sh->evStart[d]) (void)hipEventDestroy(sh->evStart[d]);
        if (d < (int)sh->evEnd.size() && sh->evEnd[d]) (void)hipEventDestroy(sh->evEnd[d]);
        if (d < (int)sh->compute.size() && sh->compute[d]) (void)hipStreamDestroy(sh->compute[d]);
        if (d < (int)sh->gather.size() && sh->gather[d]) (void)hipStreamDestroy(sh->gather[d]);
        if (d < (int)sh->comms.size() && sh->comms[d]) rccl.CommDestroy(sh->comms[d]);
    }
    delete sh;
    return EXIT_SUCCESS;
}

int spmvHipShardCSRGroups(spmat* host, int nDev, int groups, void** shardHandle) {
    if (!host || !shardHandle || nDev <= 0 || groups < 0 || groups > 64) return EXIT_FAILURE;
    const int visible = spmvHipDeviceCount();
    if (nDev > visible) { fprintf(stderr, "libspmvhip: spmvHipShardCSR: %d devices requested, %d visible\n", nDev, visible); return EXIT_FAILURE; }
    DeviceGuard guard;
    Shard* sh = new Shard;
    const int G = groups ? groups : (nDev > 1 ? 2 : 1);
    sh->nDev = nDev; sh->G = G; sh->M = host->M; sh->N = host->N;
    const size_t nBlk = (size_t)nDev * G;
    sh->bounds.resize(nBlk + 1);
    if (spmvHipPartitionRows(host->IRP, host->M, (int)nBlk, sh->bounds.data())) { delete sh; return EXIT_FAILURE; }
    sh->maxRows.assign(G, 0);
    for (int d = 0; d < nDev; ++d)
        for (int g = 0; g < G; ++g) sh->maxRows[g] = std::max(sh->maxRows[g], sh->hi(d, g) - sh->lo(d, g));
    const char* force = getenv("SPMV_SHARD_FORCE_RCCL");
    sh->useRccl = nDev > 1 || (force && *force == '1');
    sh->mats.resize(nBlk); sh->dX.assign(nDev, nullptr); sh->dYpad.assign(nBlk, nullptr);
    sh->compute.assign(nDev, nullptr); sh->gather.assign(nDev, nullptr);
    sh->evStart.assign(nDev, nullptr); sh->evEnd.assign(nDev, nullptr); sh->evK0.assign(nBlk, nullptr); sh->evK1.assign(nBlk, nullptr);
    for (auto& m : sh->mats) memset(&m, 0, sizeof m);
    int rc = EXIT_SUCCESS;
    for (int d = 0; d < nDev && !rc; ++d) {
        if (hipSetDevice(d) != hipSuccess) { rc = EXIT_FAILURE; break; }
        if (hipMalloc(&sh->dX[d], std::max<size_t>(sh->N, 1) * sizeof(double)) != hipSuccess ||
            hipStreamCreateWithFlags(&sh->compute[d], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&sh->gather[d], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&sh->evStart[d]) != hipSuccess || hipEventCreate(&sh->evEnd[d]) != hipSuccess) { rc = EXIT_FAILURE; break; }
        for (int g = 0; g < G && !rc; ++g) {
            const size_t k = (size_t)d * G + g;
            spmat* blk = spmvHipRowBlockCSR(host, sh->lo(d, g), sh->hi(d, g));
            if (!blk) { rc = EXIT_FAILURE; break; }
            rc = spMatCpyCSR(blk, &sh->mats[k]);
            free(blk->IRP); free(blk->JA); free(blk->AS); free(blk->RL); free(blk);
            if (rc) break;
            if (hipMalloc(&sh->dYpad[k], (size_t)nDev * std::max<ulong>(sh->maxRows[g], 1) * sizeof(double)) != hipSuccess ||
                hipEventCreate(&sh->evK0[k]) != hipSuccess || hipEventCreate(&sh->evK1[k]) != hipSuccess)
                rc = EXIT_FAILURE;
        }
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
    if (rc) { spmvHipShardFree(sh); return EXIT_FAILURE; }
    *shardHandle = sh;
    return EXIT_SUCCESS;
}

int spmvHipShardCSR(spmat* host, int nDev, void** shardHandle) { return spmvHipShardCSRGroups(host, nDev, 0, shardHandle); }

// mode: 0 = hipSpMVRowsCSR (serial order, bit-identical to the 1-GPU y), != 0 = hipSpMVWarpPerRowCSR (the fastest
// reduction-order kernel per block; chosen by an untimed pass at the first call)
int spmvHipSpMVSharded(void* handle, const double* hX, int mode, double* hY, double* kernelSec, double* gatherSec) {
    Shard* sh = static_cast<Shard*>(handle);
    if (!sh || !hX || !hY) return EXIT_FAILURE;
    DeviceGuard guard;
    const int n = sh->nDev, G = sh->G;
    if (!sh->selected[mode != 0]) {
        if (shardStep(sh, hX, mode, false)) return EXIT_FAILURE;
        sh->selected[mode != 0] = true;
    }
    if (shardStep(sh, hX, mode, true)) return EXIT_FAILURE;
    double kmax = 0, tmax = 0;
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        float total = 0;
        double ksum = 0;
        HIP_TRY(hipEventElapsedTime(&total, sh->evStart[d], sh->evEnd[d]));
        for (int g = 0; g < G; ++g) {
            float k = 0;
            HIP_TRY(hipEventElapsedTime(&k, sh->evK0[(size_t)d * G + g], sh->evK1[(size_t)d * G + g]));
            ksum += k * 1e-3;
        }
        kmax = std::max(kmax, ksum);
        tmax = std::max<double>(tmax, total * 1e-3);
    }
    // every device now holds all blocks; read them back from device 0, compacted
    HIP_TRY(hipSetDevice(0));
    for (int g = 0; g < G; ++g) {
        const ulong pad = std::max<ulong>(sh->maxRows[g], 1);
        for (int p = 0; p < n; ++p) {
            if (p != 0 && !sh->useRccl) continue;
            const ulong rows = sh->hi(p, g) - sh->lo(p, g);
            if (rows) HIP_TRY(hipMemcpy(hY + sh->lo(p, g), sh->dYpad[g] + (size_t)p * pad, rows * sizeof(double), hipMemcpyDeviceToHost));
        }
    }
    if (kernelSec) *kernelSec = kmax;                       // kernels of all row groups, slowest device
    if (gatherSec) *gatherSec = std::max(0.0, tmax - kmax); // what the exchange adds to the step after overlap
    return EXIT_SUCCESS;
}

}  // extern "C"
