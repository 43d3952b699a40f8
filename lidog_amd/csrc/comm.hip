// RCCL collectives behind the C ABI (`allreduce_f32(buf, n, comm, stream)`): what a host that does
// not go through torch.distributed binds for the two collectives of the path -- the gradient all-reduce of DDP
// (train_lidog.py:227-231, strategy='ddp') and the SyncBatchNorm statistics all-reduce (train_lidog.py:228).
// librccl is resolved with dlopen at first use: the library has no link-time dependency on it, and inside a torch
// process the RCCL that torch has already loaded is the one that gets used.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    LIDOG_REQUIRE(h != nullptr, "RCCL: cannot load librccl.so (%s)", dlerror());
#define SYM(field, name)                                                          \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                   \
    LIDOG_REQUIRE(g_rccl.field != nullptr, "RCCL: symbol %s missing", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.handle = h;
    return 0;
}
}  // namespace

#define LIDOG_CHECK_NCCL(expr)                                                                        \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess) {                                                                      \
            lidog_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));   \
            return 3;                                                                                 \
        }                                                                                             \
    } while (0)

extern "C" int32_t lidog_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int lidog_comm_unique_id(void *id_out) {
    if (int rc = rccl_load()) return rc;
    LIDOG_REQUIRE(id_out != nullptr, "comm_unique_id: output buffer missing");
    ncclUniqueId id;
    LIDOG_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int lidog_comm_init_rank(const void *id, int32_t nranks, int32_t rank, void **comm_out) {
    if (int rc = rccl_load()) return rc;
    LIDOG_REQUIRE(id != nullptr && comm_out != nullptr, "comm_init_rank: null argument");
    LIDOG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_rank: bad rank %d of %d", rank, nranks);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm;
    LIDOG_CHECK_NCCL(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    *comm_out = (void *)comm;
    return 0;
}

extern "C" int lidog_comm_destroy(void *comm) {
    if (comm == nullptr) return 0;
    if (int rc = rccl_load()) return rc;
    LIDOG_CHECK_NCCL(g_rccl.CommDestroy((ncclComm_t)comm));
    return 0;
}

extern "C" int lidog_allreduce_f32(float *buf, int64_t n, void *comm, void *stream) {
    if (n == 0) return 0;
    if (int rc = rccl_load()) return rc;
    LIDOG_REQUIRE(buf != nullptr && comm != nullptr && n > 0, "allreduce_f32: null buffer / communicator");
    LIDOG_CHECK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}

extern "C" int lidog_allreduce_f64(double *buf, int64_t n, void *comm, void *stream) {
    if (n == 0) return 0;
    if (int rc = rccl_load()) return rc;
    LIDOG_REQUIRE(buf != nullptr && comm != nullptr && n > 0, "allreduce_f64: null buffer / communicator");
    LIDOG_CHECK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}
