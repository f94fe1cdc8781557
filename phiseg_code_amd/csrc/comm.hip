// Data-parallel gradient exchange over RCCL / xGMI behind the C ABI.
// The reference is single-process (phiseg_model.py:151-157); this is the one collective of the data-parallel design: a sum
// of the flat fp32 gradient arena, enqueued on the caller's HIP stream between the backward graph and the Adam graph.
// RCCL is reached through dlopen: libphx.so carries no link-time dependency on it (single-GPU users never load it), and a
// process that already holds an RCCL (torch.distributed's) shares that copy.
#include <dlfcn.h>

#include "phx_common.h"

namespace {

typedef struct { char internal[128]; } nccl_uid_t;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };                     // ncclDataType_t / ncclRedOp_t values (rccl.h)

struct Api {
    void* h = nullptr;
    int (*GetUniqueId)(nccl_uid_t*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
Api g_api;

int load_api() {
    if (g_api.h) return PHX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        phx_set_error("phx_comm: cannot load librccl.so (%s)", dlerror());
        return PHX_E_RUNTIME;
    }
#define SYM(field, name)                                                            \
    *(void**)(&g_api.field) = dlsym(h, name);                                       \
    if (!g_api.field) { phx_set_error("phx_comm: librccl lacks %s", name); return PHX_E_RUNTIME; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(AllReduce, "ncclAllReduce");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
#undef SYM
    g_api.h = h;
    return PHX_OK;
}

#define PHX_CHECK_NCCL(expr)                                                                             \
    do {                                                                                                 \
        const int _r = (expr);                                                                           \
        if (_r != 0) {                                                                                   \
            phx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_api.GetErrorString(_r));        \
            return PHX_E_COMM;                                                                           \
        }                                                                                                \
    } while (0)

struct Comm {
    nccl_comm_t comm;
    int world, rank;
};

}  // namespace

extern "C" {

/* dlopen librccl and bind the entry points (idempotent): lets every rank check that RCCL is usable BEFORE any collective step */
int phx_comm_load_api(void) { return load_api(); }

int phx_comm_unique_id(void* id128) {
    PHX_REQUIRE(id128 != nullptr, PHX_E_INVAL, "phx_comm_unique_id: null buffer");
    const int rc = load_api();
    if (rc != PHX_OK) return rc;
    nccl_uid_t id;
    PHX_CHECK_NCCL(g_api.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return PHX_OK;
}

int phx_comm_init(void** comm, int world, int rank, const void* id128) {
    PHX_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, PHX_E_INVAL, "phx_comm_init: bad arguments");
    const int rc = load_api();
    if (rc != PHX_OK) return rc;
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    Comm* c = new Comm{nullptr, world, rank};
    const int r = g_api.CommInitRank(&c->comm, world, id, rank);       // uses the calling thread's current HIP device
    if (r != 0) {
        phx_set_error("phx_comm_init: ncclCommInitRank -> %s", g_api.GetErrorString(r));
        delete c;
        return PHX_E_COMM;
    }
    *comm = c;
    return PHX_OK;
}

// In-place sum over the ranks of buf[0 .. n) (fp32), issued in `bucket_elems`-sized pieces inside one RCCL group so that they
// pipeline over the xGMI links; enqueued on `stream`: it runs after whatever that stream holds and before what follows.
int phx_comm_allreduce_sum_f32(void* comm, float* buf, size_t n, size_t bucket_elems, void* stream) {
    PHX_REQUIRE(comm != nullptr && buf != nullptr, PHX_E_INVAL, "phx_comm_allreduce_sum_f32: null argument");
    Comm* c = (Comm*)comm;
    if (n == 0) return PHX_OK;
    if (bucket_elems == 0) bucket_elems = n;
    PHX_CHECK_NCCL(g_api.GroupStart());
    for (size_t i = 0; i < n; i += bucket_elems) {
        const size_t m = n - i < bucket_elems ? n - i : bucket_elems;
        const int r = g_api.AllReduce(buf + i, buf + i, m, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream);
        if (r != 0) {                              // never leave the group open: close it, then report the first error
            phx_set_error("phx_comm_allreduce_sum_f32: ncclAllReduce -> %s", g_api.GetErrorString(r));
            g_api.GroupEnd();
            return PHX_E_COMM;
        }
    }
    PHX_CHECK_NCCL(g_api.GroupEnd());
    return PHX_OK;
}

int phx_comm_destroy(void* comm) {
    if (!comm) return PHX_OK;
    Comm* c = (Comm*)comm;
    const int r = g_api.h ? g_api.CommDestroy(c->comm) : 0;
    delete c;
    if (r != 0) {
        phx_set_error("phx_comm_destroy: ncclCommDestroy -> %s", g_api.GetErrorString(r));
        return PHX_E_COMM;
    }
    return PHX_OK;
}

}  // extern "C"
