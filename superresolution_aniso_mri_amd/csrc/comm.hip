// RCCL communicator owned behind the C ABI: the data-parallel collectives of the ae_combined step (flat gradient all-reduce,
// SyncBN partial sums, parameter broadcast) issued on the CALLER's stream.
//
// Why not torch.distributed's ProcessGroupNCCL: it runs a watchdog thread that polls hipEventQuery on the end event of every
// collective it enqueued; when the step is being captured into a HIP graph that query hits an event recorded on the capturing
// stream and the process aborts (hipErrorStreamCaptureUnsupported -> terminate; profiles/r01_nccl_watchdog_abort.log).  A raw
// ncclAllReduce on the capturing stream is an ordinary capturable enqueue and has no watchdog: the whole step, collectives
// included, becomes ONE graph.
//
// librccl is bound at run time (dlopen by soname): inside a PyTorch process that resolves to the RCCL PyTorch already mapped
// (one RCCL per process), elsewhere to /opt/rocm/lib/librccl.so.1.  libaesr_hip.so therefore loads on machines without RCCL and
// the entry points below fail with a message instead.
//
// New functionality: the reference has no distributed path (kwatsch/trainer_ae.py:43-44 only moves the loss to 'cuda:1').
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "aesr_common.h"
#include "../../include/aesr_hip.h"

namespace {

// the slice of the NCCL API this file uses (rccl.h: ncclResult_t is an int enum, ncclSuccess == 0; ncclUniqueId is 128 bytes;
// ncclDataType_t: ncclFloat32 = 7, ncclFloat64 = 8; ncclRedOp_t: ncclSum = 0, ncclMax = 2)
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
enum { kFloat32 = 7, kFloat64 = 8, kSum = 0, kMax = 2 };

struct Api {
    void* handle = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* (*GetLastError)(Comm) = nullptr;
    char why[256] = {0};
};

Api g_api;
std::once_flag g_once;

void load_api() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.handle) break;
    }
    if (!g_api.handle) {
        snprintf(g_api.why, sizeof(g_api.why), "librccl.so.1 could not be loaded: %s", dlerror());
        return;
    }
#define AESR_SYM(field, name)                                                                      \
    *(void**)(&g_api.field) = dlsym(g_api.handle, name);                                           \
    if (!g_api.field && !g_api.why[0]) snprintf(g_api.why, sizeof(g_api.why), "librccl has no symbol %s", name);
    AESR_SYM(GetVersion, "ncclGetVersion")
    AESR_SYM(GetUniqueId, "ncclGetUniqueId")
    AESR_SYM(CommInitRank, "ncclCommInitRank")
    AESR_SYM(CommDestroy, "ncclCommDestroy")
    AESR_SYM(CommAbort, "ncclCommAbort")
    AESR_SYM(AllReduce, "ncclAllReduce")
    AESR_SYM(Broadcast, "ncclBroadcast")
    AESR_SYM(GroupStart, "ncclGroupStart")
    AESR_SYM(GroupEnd, "ncclGroupEnd")
    AESR_SYM(GetErrorString, "ncclGetErrorString")
#undef AESR_SYM
    *(void**)(&g_api.GetLastError) = dlsym(g_api.handle, "ncclGetLastError");      // optional
}

int api_ready(const char* what) {
    std::call_once(g_once, load_api);
    if (g_api.why[0]) {
        aesr_set_error("%s: %s", what, g_api.why);
        return AESR_ERR_UNSUPPORTED;
    }
    return AESR_OK;
}

struct CommState {
    Comm comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
};

int fail(const char* what, int rc, CommState* c) {
    const char* detail = (g_api.GetLastError && c && c->comm) ? g_api.GetLastError(c->comm) : "";
    aesr_set_error("%s: RCCL error %d (%s) %s", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?", detail ? detail : "");
    return AESR_ERR_HIP;
}

}  // namespace

extern "C" {

int aesr_comm_rccl_version(int* version_out) {
    if (int rc = api_ready("aesr_comm_rccl_version")) return rc;
    AESR_CHECK_ARG(version_out != nullptr, "aesr_comm_rccl_version: null output");
    const int r = g_api.GetVersion(version_out);
    return r == 0 ? AESR_OK : fail("ncclGetVersion", r, nullptr);
}

int aesr_comm_unique_id(void* id_host128) {
    if (int rc = api_ready("aesr_comm_unique_id")) return rc;
    AESR_CHECK_ARG(id_host128 != nullptr, "aesr_comm_unique_id: null output");
    UniqueId id;
    const int r = g_api.GetUniqueId(&id);
    if (r != 0) return fail("ncclGetUniqueId", r, nullptr);
    memcpy(id_host128, &id, AESR_COMM_ID_BYTES);
    return AESR_OK;
}

int aesr_comm_init(const void* id_host128, int nranks, int rank, void** comm_out) {
    if (int rc = api_ready("aesr_comm_init")) return rc;
    AESR_CHECK_ARG(id_host128 && comm_out, "aesr_comm_init: null argument");
    AESR_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "aesr_comm_init: rank %d of %d", rank, nranks);
    CommState* c = new CommState();
    c->nranks = nranks;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = -1;
    UniqueId id;
    memcpy(&id, id_host128, AESR_COMM_ID_BYTES);
    const int r = g_api.CommInitRank(&c->comm, nranks, id, rank);      // collective over the ranks: every rank calls it with the same id
    if (r != 0) {
        const int rc = fail("ncclCommInitRank", r, nullptr);
        delete c;
        return rc;
    }
    *comm_out = c;
    return AESR_OK;
}

int aesr_comm_destroy(void* comm) {
    if (!comm) return AESR_OK;
    CommState* c = (CommState*)comm;
    int r = 0;
    if (c->comm && g_api.CommDestroy) r = g_api.CommDestroy(c->comm);
    const int rc = r == 0 ? AESR_OK : fail("ncclCommDestroy", r, nullptr);
    delete c;
    return rc;
}

int aesr_comm_abort(void* comm) {
    if (!comm) return AESR_OK;
    CommState* c = (CommState*)comm;
    if (c->comm && g_api.CommAbort) (void)g_api.CommAbort(c->comm);
    delete c;
    return AESR_OK;
}

static int dtype_of(int dtype, int* nccl_t) {
    if (dtype == AESR_COMM_F32) { *nccl_t = kFloat32; return AESR_OK; }
    if (dtype == AESR_COMM_F64) { *nccl_t = kFloat64; return AESR_OK; }
    aesr_set_error("aesr_comm: dtype code %d (0 = f32, 1 = f64)", dtype);
    return AESR_ERR_ARG;
}

int aesr_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int op, void* stream) {
    AESR_CHECK_ARG(comm != nullptr, "aesr_comm_allreduce: null communicator");
    AESR_CHECK_ARG(buf != nullptr || count == 0, "aesr_comm_allreduce: null buffer");
    AESR_CHECK_ARG(op == AESR_COMM_SUM || op == AESR_COMM_MAX, "aesr_comm_allreduce: op code %d (0 = sum, 1 = max)", op);
    CommState* c = (CommState*)comm;
    int t;
    if (int rc = dtype_of(dtype, &t)) return rc;
    if (count == 0) return AESR_OK;
    const int r = g_api.AllReduce(buf, buf, count, t, op == AESR_COMM_SUM ? kSum : kMax, c->comm, (hipStream_t)stream);
    return r == 0 ? AESR_OK : fail("ncclAllReduce", r, c);
}

int aesr_comm_allreduce_many(void* comm, void* const* bufs_host, const size_t* counts_host, int nbufs, int dtype, int op, void* stream) {
    AESR_CHECK_ARG(comm != nullptr, "aesr_comm_allreduce_many: null communicator");
    AESR_CHECK_ARG(nbufs >= 0 && (nbufs == 0 || (bufs_host && counts_host)), "aesr_comm_allreduce_many: bad buffer list");
    AESR_CHECK_ARG(op == AESR_COMM_SUM || op == AESR_COMM_MAX, "aesr_comm_allreduce_many: op code %d", op);
    CommState* c = (CommState*)comm;
    int t;
    if (int rc = dtype_of(dtype, &t)) return rc;
    int r = g_api.GroupStart();
    if (r != 0) return fail("ncclGroupStart", r, c);
    int first_bad = 0;
    for (int i = 0; i < nbufs; ++i) {
        if (counts_host[i] == 0) continue;
        const int ri = g_api.AllReduce(bufs_host[i], bufs_host[i], counts_host[i], t, op == AESR_COMM_SUM ? kSum : kMax, c->comm, (hipStream_t)stream);
        if (ri != 0 && !first_bad) first_bad = ri;
    }
    r = g_api.GroupEnd();
    if (first_bad) return fail("ncclAllReduce (grouped)", first_bad, c);
    return r == 0 ? AESR_OK : fail("ncclGroupEnd", r, c);
}

int aesr_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
    AESR_CHECK_ARG(comm != nullptr, "aesr_comm_broadcast: null communicator");
    CommState* c = (CommState*)comm;
    AESR_CHECK_ARG(root >= 0 && root < c->nranks, "aesr_comm_broadcast: root %d of %d ranks", root, c->nranks);
    int t;
    if (int rc = dtype_of(dtype, &t)) return rc;
    if (count == 0) return AESR_OK;
    const int r = g_api.Broadcast(buf, buf, count, t, root, c->comm, (hipStream_t)stream);
    return r == 0 ? AESR_OK : fail("ncclBroadcast", r, c);
}

}  // extern "C"
