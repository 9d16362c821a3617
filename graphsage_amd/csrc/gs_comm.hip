// C1: gradient all-reduce over RCCL / xGMI inside the C ABI (ONE ncclAllReduce(sum, fp32) per step over
// the flat gradient buffer).  The reference is single-device (supervised_train.py:55-59), so there is no reference
// interface to mirror; this is the exchange step of the data-parallel hot path.
//
// The call is enqueued on the caller's HIP stream and is capturable into the step's hipGraph, so an N-GPU training
// step stays ONE graph launch (backward | all-reduce | clip+Adam) and several steps can be replayed per launch, as on
// one GPU.  RCCL is bound at run time with dlopen/dlsym: the copy torch has already mapped (SONAME librccl.so.1) is
// reused so that one process never holds two RCCL runtimes; the library itself has no link-time RCCL dependency.
#include "gs_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

typedef ncclResult_t (*fn_get_unique_id)(ncclUniqueId*);
typedef ncclResult_t (*fn_comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*fn_comm_destroy)(ncclComm_t);
typedef ncclResult_t (*fn_all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef const char* (*fn_error_string)(ncclResult_t);
typedef ncclResult_t (*fn_comm_count)(const ncclComm_t, int*);

static struct {
    void* handle;
    fn_get_unique_id get_unique_id;
    fn_comm_init_rank comm_init_rank;
    fn_comm_destroy comm_destroy;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
    fn_comm_count comm_count;
} g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

static int load_rccl() {
    if (g_rccl.handle) return GS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)          // pass 0: only a copy that is already mapped (torch's)
        for (int i = 0; i < 2 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!h) {
        gs_set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror());
        return GS_ENOTSUP;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) {
        gs_set_error("RCCL symbols missing in the mapped librccl");
        return GS_ENOTSUP;
    }
    g_rccl.handle = h;
    return GS_OK;
}

#define GS_RCCL(call)                                                                                      \
    do {                                                                                                   \
        ncclResult_t r__ = (call);                                                                         \
        if (r__ != ncclSuccess) {                                                                          \
            gs_set_error("%s failed: %s", #call, g_rccl.error_string ? g_rccl.error_string(r__) : "?");    \
            return GS_EHIP;                                                                                \
        }                                                                                                  \
    } while (0)

struct GsComm {
    ncclComm_t comm;
    int nranks, rank;
};

// Can this process bind RCCL at all?  Ranks agree on the answer (over their existing bootstrap transport) BEFORE any
// of them enters the collective ncclCommInitRank, so that a rank without RCCL cannot leave its peers blocked there.
extern "C" int gs_comm_available(void) { return load_rccl(); }

// Number of ranks of the communicator as RCCL itself reports it (ncclCommCount) -- lets a run describe itself.
extern "C" int gs_comm_count(void* comm, int32_t* n_out) {
    GS_REQUIRE(comm && n_out, "gs_comm_count: bad args");
    GsComm* c = (GsComm*)comm;
    int n = c->nranks;
    if (g_rccl.comm_count) GS_RCCL(g_rccl.comm_count(c->comm, &n));
    *n_out = n;
    return GS_OK;
}

extern "C" int gs_comm_unique_id(void* id_out_host, int32_t len) {
    GS_REQUIRE(id_out_host && len >= (int32_t)sizeof(ncclUniqueId), "gs_comm_unique_id: need a %d-byte host buffer",
               (int)sizeof(ncclUniqueId));
    int rc = load_rccl();
    if (rc != GS_OK) return rc;
    ncclUniqueId id;
    GS_RCCL(g_rccl.get_unique_id(&id));
    memset(id_out_host, 0, (size_t)len);
    memcpy(id_out_host, &id, sizeof(id));
    return GS_OK;
}

extern "C" int gs_comm_init_rank(void** comm_out, int32_t nranks, int32_t rank, const void* id_host, int32_t len) {
    GS_REQUIRE(comm_out && id_host && nranks > 0 && rank >= 0 && rank < nranks && len >= (int32_t)sizeof(ncclUniqueId),
               "gs_comm_init_rank: bad args");
    int rc = load_rccl();
    if (rc != GS_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    GsComm* c = new GsComm{nullptr, nranks, rank};
    ncclResult_t r = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);     // uses the calling thread's current device
    if (r != ncclSuccess) {
        gs_set_error("ncclCommInitRank failed: %s", g_rccl.error_string ? g_rccl.error_string(r) : "?");
        delete c;
        return GS_EHIP;
    }
    *comm_out = (void*)c;
    return GS_OK;
}

extern "C" int gs_comm_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream) {
    GS_REQUIRE(comm && buf && count > 0, "gs_comm_allreduce_sum_f32: bad args");
    GsComm* c = (GsComm*)comm;
    GS_RCCL(g_rccl.all_reduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
    return GS_OK;
}

extern "C" int gs_comm_destroy(void* comm) {
    if (!comm) return GS_OK;
    GsComm* c = (GsComm*)comm;
    if (g_rccl.comm_destroy && c->comm) (void)g_rccl.comm_destroy(c->comm);
    delete c;
    return GS_OK;
}
