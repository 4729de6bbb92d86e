// Gradient exchange of the sharded SVI step through the C ABI: RCCL (xGMI) bound at run time.
//
// The reference has no multi-device path (one MXNet context per Inference object); north_star shards the Monte-Carlo
// samples of StochasticVariationalInference.compute (inference/variational.py:15-26 -- an expectation over independent samples) over the
// GPUs and sums the flat gradient once per step.  mxfusion_amd's own loops use torch.distributed (backend "nccl" = RCCL); these entry
// points give a reference-side binder, which has no PyTorch, the same exchange.
//
// librccl is opened with dlopen on first use (the symbols of an RCCL already loaded into the process are preferred), so libmxf_gp.so has
// no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "common.h"
#include "internal.h"

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// resolved once; a function-local static initialised by a lambda is published only when fully populated (thread-safe by the language rules)
RcclApi load_rccl() {
    RcclApi api;
    void* src = RTLD_DEFAULT;                       // an RCCL the process already holds (e.g. the caller's framework)
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
        const char* env = getenv("MXF_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return api;
        src = api.lib;
    }
#define SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(src, name))
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Broadcast && api.GetErrorString;
    return api;
}

RcclApi* rccl() {
    static RcclApi api = load_rccl();
    return &api;
}

#define MXF_RCCL(h, api, call)                                                                              \
    do {                                                                                                    \
        ncclResult_t _r = (call);                                                                           \
        if (_r != ncclSuccess) MXF_FAIL(h, -7, "%s failed: %s", #call, (api)->GetErrorString(_r));          \
    } while (0)

int nccl_type(int dtype, ncclDataType_t* t) {
    if (dtype == MXF_F32) { *t = ncclFloat32; return 0; }
    if (dtype == MXF_F64) { *t = ncclFloat64; return 0; }
    return -1;
}

}  // namespace

void mxf_comm_release(mxf_ctx* h) {
    if (h && h->comm) {
        RcclApi* api = rccl();
        if (api->ok) (void)api->CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr; h->comm_nranks = 0; h->comm_rank = -1;
    }
}

extern "C" int mxf_comm_unique_id(mxf_handle h, void* id_out) {
    if (!h || !id_out) return -1;
    RcclApi* api = rccl();
    if (!api->ok) MXF_FAIL(h, -6, "mxf_comm_unique_id: librccl could not be loaded (set MXF_RCCL_LIB)");
    static_assert(sizeof(ncclUniqueId) == MXF_COMM_ID_BYTES, "MXF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    MXF_RCCL(h, api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int mxf_comm_init(mxf_handle h, int nranks, int rank, const void* id) {
    if (!h || !id) return -1;
    if (nranks < 1 || rank < 0 || rank >= nranks) MXF_FAIL(h, -2, "mxf_comm_init: bad rank %d of %d", rank, nranks);
    RcclApi* api = rccl();
    if (!api->ok) MXF_FAIL(h, -6, "mxf_comm_init: librccl could not be loaded (set MXF_RCCL_LIB)");
    if (h->comm) MXF_FAIL(h, -2, "mxf_comm_init: this handle already has a communicator (mxf_comm_destroy first)");
    MXF_HIP(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    MXF_RCCL(h, api, api->CommInitRank(&c, nranks, uid, rank));
    h->comm = c; h->comm_nranks = nranks; h->comm_rank = rank;
    return 0;
}

extern "C" int mxf_comm_destroy(mxf_handle h) {
    if (!h) return -1;
    mxf_comm_release(h);
    return 0;
}

extern "C" int mxf_allreduce_sum(mxf_handle h, int dtype, void* buf, int64_t count, void* stream) {
    if (!h) return -1;
    if (!h->comm) MXF_FAIL(h, -2, "mxf_allreduce_sum: no communicator on this handle (mxf_comm_init)");
    ncclDataType_t t;
    if (nccl_type(dtype, &t)) MXF_FAIL(h, -2, "mxf_allreduce_sum: bad dtype %d", dtype);
    if (count < 0 || (count > 0 && !buf)) MXF_FAIL(h, -2, "mxf_allreduce_sum: bad buffer");
    if (count == 0) return 0;
    RcclApi* api = rccl();
    MXF_RCCL(h, api, api->AllReduce(buf, buf, (size_t)count, t, ncclSum, (ncclComm_t)h->comm, (hipStream_t)stream));
    return 0;
}

extern "C" int mxf_bcast(mxf_handle h, int dtype, void* buf, int64_t count, int root, void* stream) {
    if (!h) return -1;
    if (!h->comm) MXF_FAIL(h, -2, "mxf_bcast: no communicator on this handle (mxf_comm_init)");
    ncclDataType_t t;
    if (nccl_type(dtype, &t)) MXF_FAIL(h, -2, "mxf_bcast: bad dtype %d", dtype);
    if (root < 0 || root >= h->comm_nranks) MXF_FAIL(h, -2, "mxf_bcast: bad root %d", root);
    if (count < 0 || (count > 0 && !buf)) MXF_FAIL(h, -2, "mxf_bcast: bad buffer");
    if (count == 0) return 0;
    RcclApi* api = rccl();
    MXF_RCCL(h, api, api->Broadcast(buf, buf, (size_t)count, t, root, (ncclComm_t)h->comm, (hipStream_t)stream));
    return 0;
}
