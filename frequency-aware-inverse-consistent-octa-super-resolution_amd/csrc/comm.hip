// Gradient exchange of the data-parallel train step: a thin C ABI over RCCL.
//
// The reference has no distributed code; the exchange sits where a DDP wrap of its loop would put it: after
// loss_G.backward() (train.py:238) and after the two discriminator backwards (train.py:255,267), one all-reduce
// over each flat gradient arena.  RCCL is bound at run time (dlopen of the librccl the process already holds --
// torch loads one -- else the system one), so libfaoctasr.so has no link-time dependency on it and a single-GPU
// host never touches it.  The collective is enqueued on the caller's stream and is capturable in a hipGraph.
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include <mutex>
#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;                 // written once under g_once, immutable afterwards
std::once_flag g_once;

void bind_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)                                   // the copy already mapped into this process, if any
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    Rccl r;
    r.handle = h;
#define BIND(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym))
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount");
    BIND(AllReduce, "ncclAllReduce");
    BIND(Broadcast, "ncclBroadcast");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.Broadcast && r.GetErrorString;
    g_rccl = r;
}

const Rccl* rccl() {
    std::call_once(g_once, bind_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int nccl_fail(const Rccl* r, const char* what, ncclResult_t e) {
    return faoctasr::fail(FAOCTASR_EHIP, "%s: %s", what, r->GetErrorString(e));
}

}  // namespace

extern "C" {

int faoctasr_comm_unique_id(void* id128) {
    const Rccl* r = rccl();
    if (!r) return faoctasr::fail(FAOCTASR_EUNSUPPORTED, "comm_unique_id: librccl.so could not be loaded (%s)", dlerror());
    if (!id128) return faoctasr::fail(FAOCTASR_EINVAL, "comm_unique_id: null buffer");
    ncclResult_t e = r->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128));
    return e == ncclSuccess ? FAOCTASR_OK : nccl_fail(r, "ncclGetUniqueId", e);
}

int faoctasr_comm_create(void** comm, int nranks, int rank, const void* id128) {
    const Rccl* r = rccl();
    if (!r) return faoctasr::fail(FAOCTASR_EUNSUPPORTED, "comm_create: librccl.so could not be loaded (%s)", dlerror());
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return faoctasr::fail(FAOCTASR_EINVAL, "comm_create: bad arguments (nranks %d, rank %d)", nranks, rank);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t e = r->CommInitRank(&c, nranks, id, rank);          // binds the calling thread's current device
    if (e != ncclSuccess) return nccl_fail(r, "ncclCommInitRank", e);
    *comm = c;
    return FAOCTASR_OK;
}

int faoctasr_comm_destroy(void* comm) {
    const Rccl* r = rccl();
    if (!r || !comm) return faoctasr::fail(FAOCTASR_EINVAL, "comm_destroy: no communicator");
    ncclResult_t e = r->CommDestroy(reinterpret_cast<ncclComm_t>(comm));
    return e == ncclSuccess ? FAOCTASR_OK : nccl_fail(r, "ncclCommDestroy", e);
}

int faoctasr_comm_size(void* comm) {
    const Rccl* r = rccl();
    if (!r || !comm) return faoctasr::fail(FAOCTASR_EINVAL, "comm_size: no communicator");
    int n = 0;
    ncclResult_t e = r->CommCount(reinterpret_cast<ncclComm_t>(comm), &n);
    return e == ncclSuccess ? n : nccl_fail(r, "ncclCommCount", e);
}

// In-place SUM all-reduce of one flat gradient bucket.  dtype 0 = fp32 (the arenas are fp32; the 1/world average is
// folded into faoctasr_adamw_step's grad_scale).
int faoctasr_grad_allreduce(float* bucket, long count, int dtype, void* comm, faoctasr_stream_t stream) {
    const Rccl* r = rccl();
    if (!r || !comm) return faoctasr::fail(FAOCTASR_EINVAL, "grad_allreduce: no communicator");
    if (!bucket || count < 0) return faoctasr::fail(FAOCTASR_EINVAL, "grad_allreduce: bad bucket");
    if (dtype != 0) return faoctasr::fail(FAOCTASR_EUNSUPPORTED, "grad_allreduce: dtype %d (only 0 = fp32)", dtype);
    if (count == 0) return FAOCTASR_OK;
    ncclResult_t e = r->AllReduce(bucket, bucket, (size_t)count, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                  reinterpret_cast<hipStream_t>(stream));
    return e == ncclSuccess ? FAOCTASR_OK : nccl_fail(r, "ncclAllReduce", e);
}

// Replicas start identical: broadcast rank `root`'s buffer (parameter arenas, BatchNorm buffers) in place.
int faoctasr_param_broadcast(float* buf, long count, int root, void* comm, faoctasr_stream_t stream) {
    const Rccl* r = rccl();
    if (!r || !comm) return faoctasr::fail(FAOCTASR_EINVAL, "param_broadcast: no communicator");
    if (!buf || count < 0) return faoctasr::fail(FAOCTASR_EINVAL, "param_broadcast: bad buffer");
    if (count == 0) return FAOCTASR_OK;
    ncclResult_t e = r->Broadcast(buf, buf, (size_t)count, ncclFloat32, root, reinterpret_cast<ncclComm_t>(comm),
                                  reinterpret_cast<hipStream_t>(stream));
    return e == ncclSuccess ? FAOCTASR_OK : nccl_fail(r, "ncclBroadcast", e);
}

}  // extern "C"
