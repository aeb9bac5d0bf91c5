// Peer-mapped exchange regions for the ONE-SHOT SyncBN exchange of a data-parallel step ("one-shot all-gather +
// local reduce / direct P2P write over the 7 links, which beats a ring at this size"): every rank owns one small region of
// fine-grained device memory, hands its IPC handle to the other ranks over the host control plane, and maps theirs.  The BatchNorm
// kernels then WRITE a rank's [G][2][C] partial sums straight into every peer's region over xGMI and read the peers' from their own
// (csrc/bn_fused.hip) -- no RCCL launch, no graph cut.  New functionality: the reference has no distributed path.
//
// Fine-grained (hipDeviceMallocFinegrained) memory: stores from a peer GPU and the flag that follows them become visible to a kernel
// that is already running on the owner (coarse-grained memory only promises that at kernel boundaries).
#include <string.h>

#include "aesr_common.h"
#include "../../include/aesr_hip.h"

extern "C" {

int aesr_p2p_alloc(size_t bytes, void** region) {
    AESR_CHECK_ARG(region && bytes > 0 && bytes % 128 == 0, "aesr_p2p_alloc: need a size that is a multiple of 128 bytes");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        aesr_set_error("aesr_p2p_alloc: hipExtMallocWithFlags(%zu bytes, fine-grained) failed: %s", bytes, hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(p);
        aesr_set_error("aesr_p2p_alloc: clearing the region failed: %s", hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    *region = p;
    return AESR_OK;
}

int aesr_p2p_free(void* region) {
    if (!region) return AESR_OK;
    const hipError_t e = hipFree(region);
    if (e != hipSuccess) {
        aesr_set_error("aesr_p2p_free: %s", hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    return AESR_OK;
}

int aesr_p2p_get_handle(void* region, char* handle64) {
    AESR_CHECK_ARG(region && handle64, "aesr_p2p_get_handle: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == AESR_P2P_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, region);
    if (e != hipSuccess) {
        aesr_set_error("aesr_p2p_get_handle: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on hosts whose driver "
                       "only supports dmabuf IPC)", hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    memcpy(handle64, &h, sizeof(h));
    return AESR_OK;
}

int aesr_p2p_open(const char* handle64, void** peer_region) {
    AESR_CHECK_ARG(handle64 && peer_region, "aesr_p2p_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        aesr_set_error("aesr_p2p_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    *peer_region = p;
    return AESR_OK;
}

int aesr_p2p_close(void* peer_region) {
    if (!peer_region) return AESR_OK;
    const hipError_t e = hipIpcCloseMemHandle(peer_region);
    if (e != hipSuccess) {
        aesr_set_error("aesr_p2p_close: %s", hipGetErrorString(e));
        return AESR_ERR_HIP;
    }
    return AESR_OK;
}

}  // extern "C"
