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
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
    SYM(CommCount, "ncclCommCount");
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

// ranks of the communicator as RCCL itself reports them (bench.py prints it: evidence of how many GPUs really took part)
extern "C" int32_t lidog_comm_count(void *comm) {
    if (comm == nullptr || rccl_load()) return -1;
    int n = -1;
    if (g_rccl.CommCount((ncclComm_t)comm, &n) != ncclSuccess) return -1;
    return n;
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

// ------------------------------------------------------------------ one-shot peer all-reduce for small messages
// The SyncBatchNorm statistics messages of the path (train_lidog.py:228) are <= 513 doubles, 241 of them per training
// step, every one on the dependent chain: a ring / tree collective pays its hop latency 241 times.  xGMI gives every GPU
// a direct link to every other one, so each rank PUSHES its vector into a mailbox in every peer's memory (one posted
// write per peer), raises a per-sender flag behind it, waits for the flags of all senders in its own mailbox, and adds
// the N vectors up in rank order -- the same order on every rank, so every rank computes the same bits.  One
// workgroup, one launch, on the caller's stream.
//
// Mailbox of a rank (fine-grained device memory, opened by the peers through hipIpc handles):
//   [2 slots][nranks][stride] doubles; a sender's data at [slot][sender][0 .. n), its flag (the call's sequence number,
//   uint64) at [slot][sender][stride - 1].  Slots alternate with the sequence number: call k + 2 cannot start before
//   every rank has contributed to call k + 1, i.e. has finished reading call k.
// Waiting is bounded: a rank that does not see a flag within the limit (minutes by default, seconds in the start-up
// self-test of lidog_amd.comm) sets the error word and leaves (the launch always terminates); lidog_peer_status reports it.
#define PEER_MAX_RANKS 16
#define PEER_SPIN_LIMIT_DEFAULT (1u << 27)   // x ~1-2 us of sleep + system-scope load: several minutes (a rank may be
                                            // busy writing a checkpoint while the others wait for its message)
// error word of a communicator: 0 ok, 1 = a sender's flag did not arrive within the wait limit, 2 = a sender is AHEAD
// of this rank (its flag carries a later sequence number: the ranks no longer make the same calls).  Once it is set
// no later call waits for anything (the step runs on, on invalid sums, at full speed; Transport.check() raises on
// every rank at the next boundary): a dead or desynchronised peer costs ONE wait limit, not one per call.
#define PEER_ERR_TIMEOUT 1
#define PEER_ERR_DESYNC 2

struct PeerComm {
    int rank, nranks, max_doubles, stride;
    unsigned spin_limit;
    uint64_t seq;
    double *local;
    double *peers[PEER_MAX_RANKS];
    int32_t *err_dev;
    hipStream_t stream;   // the stream of the first call; every later call must use it (slot reuse relies on stream order)
    bool stream_set;
    uint64_t skip_flag_at;  // fault injection (tests): the call with this sequence number raises no flags; 0 = never
};

struct PeerArgs {
    double *peers[PEER_MAX_RANKS];
    double *local;
    int rank, nranks, stride;
    unsigned spin_limit;
    uint64_t seq;
    int32_t *err;
    int skip_flag;
};

static inline int peer_stride(int max_doubles) {
    // data (rounded up to whole 16-byte granules) + the flag in the last double, whole 128-byte lines per sender
    return (int)(((int64_t)max_doubles + 2 + 15) / 16 * 16);
}

typedef double lidog_f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_peer_allreduce(PeerArgs a, double *__restrict__ buf, int n) {
    const int slot = (int)(a.seq & 1);
    const int tid = threadIdx.x;
    const int n2 = (n + 1) / 2;     // 16-byte granules per message (an odd message carries one pad double)
    const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    // 1. push my vector into every rank's mailbox (my own included: the sum below reads all N the same way), one 16-byte
    // store per granule: a posted write over the sender's direct xGMI link (the mailboxes are fine-grained memory:
    // stores go through to the owner's memory).  513 doubles to 8 ranks = 2 056 stores (was 4 104 scalar ones).
    for (int idx = tid; idx < n2 * a.nranks; idx += 256) {
        const int p = idx / n2, i = idx - p * n2;
        double *dst = a.peers[p] + ((size_t)slot * a.nranks + a.rank) * a.stride;
        lidog_f64x2 v;
        v.x = buf[2 * i];
        v.y = (2 * i + 1 < n) ? buf[2 * i + 1] : 0.0;
        *reinterpret_cast<lidog_f64x2 *>(dst + 2 * i) = v;
    }
    __threadfence_system();     // my stores are visible system-wide before any of my flags is
    __syncthreads();
    if (tid < a.nranks) {
        uint64_t *flag = reinterpret_cast<uint64_t *>(a.peers[tid] + ((size_t)slot * a.nranks + a.rank) * a.stride +
                                                      (a.stride - 1));
        if (!a.skip_flag) __hip_atomic_store(flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // 2. wait for sender `tid` in MY mailbox
        const uint64_t *mine = reinterpret_cast<const uint64_t *>(
            a.local + ((size_t)slot * a.nranks + tid) * a.stride + (a.stride - 1));
        unsigned spins = 0;
        for (;;) {
            const uint64_t f = __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (f == a.seq) break;
            if (f > a.seq) {    // the sender has moved on: this slot's data is gone
                __hip_atomic_store(a.err, PEER_ERR_DESYNC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            if (dead) break;    // an earlier call failed: nothing waits any more
            if (++spins > a.spin_limit) {
                __hip_atomic_store(a.err, PEER_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __threadfence_system();
    // 3. the sum, in rank order
    for (int i = tid; i < n2; i += 256) {
        double s0 = 0.0, s1 = 0.0;
        for (int r = 0; r < a.nranks; ++r) {
            const double *src = a.local + ((size_t)slot * a.nranks + r) * a.stride;
            const lidog_f64x2 v = *reinterpret_cast<const lidog_f64x2 *>(src + 2 * i);
            s0 += v.x;
            s1 += v.y;
        }
        buf[2 * i] = s0;
        if (2 * i + 1 < n) buf[2 * i + 1] = s1;
    }
}

extern "C" int32_t lidog_peer_handle_bytes(void) { return (int32_t)sizeof(hipIpcMemHandle_t); }

extern "C" int64_t lidog_peer_mailbox_bytes(int32_t nranks, int32_t max_doubles) {
    if (nranks < 1 || nranks > PEER_MAX_RANKS || max_doubles < 1) return -1;
    return 2 * (int64_t)nranks * peer_stride(max_doubles) * 8;
}

// fine-grained (uncached, coherent across agents) device memory for a mailbox, zeroed; handle_out: its hipIpc handle
extern "C" int lidog_peer_mailbox_alloc(int64_t bytes, void **ptr_out, void *handle_out) {
    LIDOG_REQUIRE(bytes > 0 && ptr_out && handle_out, "peer_mailbox_alloc: bad arguments");
    void *p = nullptr;
    LIDOG_CHECK_HIP(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    LIDOG_CHECK_HIP(hipMemset(p, 0, (size_t)bytes));
    LIDOG_CHECK_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    LIDOG_CHECK_HIP(hipIpcGetMemHandle(&h, p));
    memcpy(handle_out, &h, sizeof(h));
    *ptr_out = p;
    return 0;
}

extern "C" int lidog_peer_mailbox_open(const void *handle, void **ptr_out) {
    LIDOG_REQUIRE(handle && ptr_out, "peer_mailbox_open: bad arguments");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    LIDOG_CHECK_HIP(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
    // probe the mapping with a runtime copy before any kernel stores through it: a mapping that is not usable from
    // this device shows up here as an error code instead of as a memory fault inside the all-reduce kernel
    uint64_t probe = 0;
    hipError_t e = hipMemcpy(&probe, *ptr_out, sizeof(probe), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipIpcCloseMemHandle(*ptr_out);
        *ptr_out = nullptr;
        lidog_set_error("peer_mailbox_open: the peer's mailbox cannot be read from this device (%s)", hipGetErrorString(e));
        return 1;
    }
    return 0;
}

// peer_ptrs [nranks]: the mailbox of every rank as mapped into THIS process (entry `rank` = local)
extern "C" int lidog_peer_comm_create(int32_t rank, int32_t nranks, int32_t max_doubles, void *local,
                                      void *const *peer_ptrs, void **comm_out) {
    LIDOG_REQUIRE(nranks >= 1 && nranks <= PEER_MAX_RANKS && rank >= 0 && rank < nranks && local && peer_ptrs && comm_out,
                  "peer_comm_create: bad arguments (at most %d ranks)", PEER_MAX_RANKS);
    PeerComm *c = new PeerComm();
    c->rank = rank;
    c->nranks = nranks;
    c->max_doubles = max_doubles;
    c->stride = peer_stride(max_doubles);
    c->seq = 0;
    c->stream = nullptr;
    c->stream_set = false;
    c->skip_flag_at = 0;
    c->spin_limit = PEER_SPIN_LIMIT_DEFAULT;
    c->local = (double *)local;
    for (int r = 0; r < nranks; ++r) c->peers[r] = (double *)peer_ptrs[r];
    c->peers[rank] = c->local;
    // (hipMemset on device memory is asynchronous to the host and ordered on the NULL stream only: wait for it here, the
    // all-reduce kernels run on whatever stream the communicator is bound to later)
    if (hipMalloc((void **)&c->err_dev, sizeof(int32_t)) != hipSuccess || hipMemset(c->err_dev, 0, sizeof(int32_t)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        delete c;
        lidog_set_error("peer_comm_create: cannot allocate the error word");
        return 1;
    }
    *comm_out = c;
    return 0;
}

extern "C" int32_t lidog_peer_max_doubles(void *comm) { return comm ? ((PeerComm *)comm)->max_doubles : 0; }

// polls a waiting rank makes before it gives up on a sender (each ~1-2 us); 0 restores the default (several minutes)
extern "C" int lidog_peer_set_spin_limit(void *comm, int64_t polls) {
    LIDOG_REQUIRE(comm && polls >= 0 && polls < ((int64_t)1 << 32), "peer_set_spin_limit: bad arguments");
    ((PeerComm *)comm)->spin_limit = polls ? (unsigned)polls : PEER_SPIN_LIMIT_DEFAULT;
    return 0;
}

// sum over the ranks of buf[0 .. n) in place, identical bits on every rank; every rank calls it in the same order, and
// always on the same stream: the two mailbox slots alternate with the call number, and call k + 2 may only overwrite a
// slot after call k has been read -- which stream order guarantees and two streams would not
extern "C" int lidog_peer_allreduce_f64(void *comm, double *buf, int64_t n, void *stream) {
    PeerComm *c = (PeerComm *)comm;
    LIDOG_REQUIRE(c && buf && n >= 1 && n <= c->max_doubles, "peer_allreduce_f64: %lld doubles, the mailbox takes %d",
                  (long long)n, c ? c->max_doubles : 0);
    if (!c->stream_set) {
        c->stream = (hipStream_t)stream;
        c->stream_set = true;
    }
    LIDOG_REQUIRE(c->stream == (hipStream_t)stream,
                  "peer_allreduce_f64: called on another stream than the first call of this communicator (the mailbox "
                  "slots are reused in stream order; lidog_peer_rebind_stream after a synchronisation moves it)");
    PeerArgs a;
    for (int r = 0; r < c->nranks; ++r) a.peers[r] = c->peers[r];
    a.local = c->local;
    a.rank = c->rank;
    a.nranks = c->nranks;
    a.stride = c->stride;
    a.seq = ++c->seq;
    a.spin_limit = c->spin_limit;
    a.err = c->err_dev;
    a.skip_flag = (c->skip_flag_at != 0 && c->skip_flag_at == a.seq) ? 1 : 0;
    k_peer_allreduce<<<1, 256, 0, (hipStream_t)stream>>>(a, buf, (int)n);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// The communicator follows the caller to another stream: waits for everything queued on the old one first.
extern "C" int lidog_peer_rebind_stream(void *comm, void *stream) {
    PeerComm *c = (PeerComm *)comm;
    LIDOG_REQUIRE(c != nullptr, "peer_rebind_stream: null communicator");
    if (c->stream_set && c->stream != (hipStream_t)stream) LIDOG_CHECK_HIP(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)stream;
    c->stream_set = true;
    return 0;
}

// fault injection for the tests of the failure path: the call with sequence number `seq` (1-based) raises no flags
extern "C" int lidog_peer_inject_skip_flag(void *comm, int64_t seq) {
    PeerComm *c = (PeerComm *)comm;
    LIDOG_REQUIRE(c != nullptr && seq >= 0, "peer_inject_skip_flag: bad arguments");
    c->skip_flag_at = (uint64_t)seq;
    return 0;
}

// calls made on this communicator so far
extern "C" int64_t lidog_peer_calls(void *comm) { return comm ? (int64_t)((PeerComm *)comm)->seq : -1; }

// 0 = every wait so far was satisfied, 1 = a sender's flag did not arrive in time, 2 = a sender was ahead of this rank
// (synchronises with the device)
extern "C" int32_t lidog_peer_status(void *comm) {
    PeerComm *c = (PeerComm *)comm;
    if (!c) return -1;
    int32_t e = -1;
    if (hipMemcpy(&e, c->err_dev, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return e;
}

extern "C" int lidog_peer_mailbox_free(void *ptr) {
    if (ptr) LIDOG_CHECK_HIP(hipFree(ptr));
    return 0;
}

extern "C" int lidog_peer_mailbox_close(void *peer_ptr) {
    if (peer_ptr) LIDOG_CHECK_HIP(hipIpcCloseMemHandle(peer_ptr));
    return 0;
}

extern "C" int lidog_peer_comm_destroy(void *comm, int32_t close_peers) {
    PeerComm *c = (PeerComm *)comm;
    if (!c) return 0;
    if (close_peers)
        for (int r = 0; r < c->nranks; ++r)
            if (r != c->rank && c->peers[r]) (void)hipIpcCloseMemHandle(c->peers[r]);
    (void)hipFree(c->err_dev);
    (void)hipFree(c->local);
    delete c;
    return 0;
}
