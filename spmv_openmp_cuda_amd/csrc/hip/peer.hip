// peer.hip -- peer windows: device buffers of one process mapped into the other processes of the node,
// and the push form of the all-gather of y (every rank copies its rows straight into the other ranks'
// vectors over its point-to-point xGMI links).  New functionality: the reference is single-GPU
// (no MPI/NCCL/cudaSetDevice anywhere in it); see DESIGN.md "Multi-GPU".
//
// The exchange of the 64-byte handles is the caller's business (bench.py ships them with
// torch.distributed); nothing here depends on torch or RCCL.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

using namespace spmvhip;

namespace {

static_assert(sizeof(hipIpcMemHandle_t) == SPMV_IPC_HANDLE_BYTES, "IPC handle size");

struct PushState {
    std::vector<hipStream_t> stream;        // one copy stream per peer slot
    std::vector<hipEvent_t>  done;
    std::vector<bool>        dirty;
    hipEvent_t               fence = nullptr;
} P;

int ensureSlots(int n) {
    if (!P.fence) HIP_TRY(hipEventCreateWithFlags(&P.fence, hipEventDisableTiming));
    while ((int)P.stream.size() < n) {
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        P.stream.push_back(s); P.done.push_back(e); P.dirty.push_back(false);
    }
    return EXIT_SUCCESS;
}

}  // namespace

namespace spmvhip {
void peerFinalize() {
    for (size_t k = 0; k < P.stream.size(); ++k) {
        (void)hipStreamSynchronize(P.stream[k]);
        (void)hipStreamDestroy(P.stream[k]);
        (void)hipEventDestroy(P.done[k]);
    }
    if (P.fence) (void)hipEventDestroy(P.fence);
    P = PushState{};
}
}  // namespace spmvhip

extern "C" {

int spmvHipWindowCreate(size_t bytes, void** dBase, unsigned char handle[SPMV_IPC_HANDLE_BYTES]) {
    if (!dBase || !handle) return EXIT_FAILURE;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 1));
    hipIpcMemHandle_t h;
    if (!hipOk(hipIpcGetMemHandle(&h, p), "hipIpcGetMemHandle")) { (void)hipFree(p); return EXIT_FAILURE; }
    memcpy(handle, &h, SPMV_IPC_HANDLE_BYTES);
    *dBase = p;
    return EXIT_SUCCESS;
}

int spmvHipWindowFree(void* dBase) {
    if (dBase) HIP_TRY(hipFree(dBase));
    return EXIT_SUCCESS;
}

int spmvHipWindowOpen(const unsigned char handle[SPMV_IPC_HANDLE_BYTES], int ownerDev, void** dPeer) {
    if (!handle || !dPeer) return EXIT_FAILURE;
    int cur = 0, n = 0;
    HIP_TRY(hipGetDevice(&cur));
    HIP_TRY(hipGetDeviceCount(&n));
    if (ownerDev < 0 || ownerDev >= n) {
        fprintf(stderr, "libspmvhip: spmvHipWindowOpen: owner device %d is not visible to this process (%d devices)\n", ownerDev, n);
        return EXIT_FAILURE;
    }
    if (ownerDev != cur) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, cur, ownerDev));
        if (!can) { fprintf(stderr, "libspmvhip: spmvHipWindowOpen: device %d cannot access device %d\n", cur, ownerDev); return EXIT_FAILURE; }
        hipError_t e = hipDeviceEnablePeerAccess(ownerDev, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();      // clear the sticky error
        else if (!hipOk(e, "hipDeviceEnablePeerAccess")) return EXIT_FAILURE;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, SPMV_IPC_HANDLE_BYTES);
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *dPeer = p;
    return EXIT_SUCCESS;
}

int spmvHipWindowClose(void* dPeer) {
    if (dPeer) HIP_TRY(hipIpcCloseMemHandle(dPeer));
    return EXIT_SUCCESS;
}

int spmvHipPeerPush(const void* dSrcBase, size_t offset, size_t bytes, int nPeers, void* const* dPeerBases) {
    if (nPeers < 0 || nPeers > SPMV_MAX_PEERS || (nPeers && (!dPeerBases || !dSrcBase))) return EXIT_FAILURE;
    if (!nPeers || !bytes) return EXIT_SUCCESS;
    if (ensureSlots(nPeers)) return EXIT_FAILURE;
    HIP_TRY(hipEventRecord(P.fence, libraryStream()));
    for (int k = 0; k < nPeers; ++k) {
        HIP_TRY(hipStreamWaitEvent(P.stream[k], P.fence, 0));
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(dPeerBases[k]) + offset, static_cast<const char*>(dSrcBase) + offset, bytes,
                               hipMemcpyDefault, P.stream[k]));
        HIP_TRY(hipEventRecord(P.done[k], P.stream[k]));
        P.dirty[k] = true;
    }
    return EXIT_SUCCESS;
}

int spmvHipPeerPushJoin(void) {
    for (size_t k = 0; k < P.stream.size(); ++k)
        if (P.dirty[k]) {
            HIP_TRY(hipStreamWaitEvent(libraryStream(), P.done[k], 0));
            P.dirty[k] = false;
        }
    return EXIT_SUCCESS;
}

}  // extern "C"
