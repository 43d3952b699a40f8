// Trunk executor: the MinkUNet encoder-decoder (utils/models/minkunet_bev.py:302-399, utils/models/minkunet.py:97-158)
// forward and backward as ONE host call each.
//
// The python operator path (lidog_amd/me.py) issues ~1 000 launches per training step through ~550 autograd nodes; the
// launch sequence itself is static (63 convolutions, 62 BatchNorms, 4 concatenations) -- only row counts and map
// pointers change per batch.  This file walks a table-driven program and calls the SAME entry points of this library
// in the same order with the same arguments as me._SparseConvFn / me._BatchNormFn / me._Cat2Fn do, so every result is
// bit-identical to the operator path (tests/test_gpu_trunk.py); what changes is the host cost of a step.
//
// No state is kept between calls except a pool of timing-less events for the second backward stream.  All memory comes
// from the caller: `arena` (activations that live until backward), `garena` (gradients; never reused inside one
// backward pass, so the weight gradients running on the lane stream may read them at any time), `scratch` (product
// rows, reduction workspaces: main stream only), `lane_scratch` (weight-gradient partial slabs: lane stream only).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

void lidog_gemm_multi_suspend(int delta);   // sconv_mfma.hip

namespace {

// ---- table layouts (int64 rows; python fills them as numpy arrays, lidog_amd/trunk.py)
enum { TC_KIND, TC_MAP, TC_CIN, TC_COUT, TC_K, TC_W, TC_WT, TC_GW, TC_BIAS, TC_GBIAS, TC_BNW, TC_BNB, TC_BNRM, TC_BNRV,
       TC_GBNW, TC_GBNB, TC_ITEMS, TC_NITEMS, TC_ITEMOFF, TC_COLS = 20 };
enum { TM_K, TM_NIN, TM_NOUT, TM_P, TM_PAIR_IN, TM_PAIR_OUT, TM_RP_OUT, TM_RL_OUT, TM_RP_IN, TM_RL_IN, TM_TILES,
       TM_NTILES, TM_NBR, TM_IDENT, TM_PERM, TM_WMASK, TM_ORDER, TM_COLS = 20 };
enum { TO_TYPE, TO_CONV, TO_IN, TO_OUT, TO_RELU, TO_RES, TO_FOLD, TO_B, TO_COLS = 8 };
enum { TB_LEVEL, TB_CH, TB_EXT, TB_COLS = 4 };
enum { REC_PRE, REC_MEAN, REC_INVSTD, REC_BITS, REC_COLS = 4 };   // REC_BITS: arena offset + 1 of the ReLU bit mask, 0 = none

enum { KIND_K3 = 0, KIND_DOWN = 1, KIND_UP = 2, KIND_1X1 = 3, KIND_STEM = 4 };
enum { OP_CONVBN = 0, OP_CAT = 1, OP_CONV = 2 };

struct Bump {
    char *base;
    int64_t off, cap, peak;
    bool dry;
    void reset() { off = 0; }
    void *take(int64_t bytes) {
        int64_t a = (bytes + 255) / 256 * 256;
        void *p = dry ? (void *)(uintptr_t)(4096 + off) : (void *)(base + off);
        off += a;
        if (off > peak) peak = off;
        return p;
    }
    bool ok() const { return dry || peak <= cap; }
};

struct Ctx {
    const int64_t *convs;
    const double *conv_f;
    int n_convs;
    const int64_t *maps;
    int n_maps;
    const int64_t *ops;
    int n_ops;
    const int64_t *bufs;
    int n_bufs;
    const int64_t *level_rows;
    const int64_t *ext;
    bool dry;
    int64_t rows(int b) const { return level_rows[bufs[b * TB_COLS + TB_LEVEL]]; }
    int ch(int b) const { return (int)bufs[b * TB_COLS + TB_CH]; }
    int64_t bytes(int b) const { return rows(b) * ch(b) * 4; }
};

template <typename T>
inline T *P(int64_t v) { return reinterpret_cast<T *>(static_cast<uintptr_t>(v)); }

// ---- data parallelism inside the executor (train_lidog.py:227-231: DDP + MinkowskiSyncBatchNorm).
// dp [DP_COLS] int64, NULL = single process.  Two transports: native RCCL communicators of this library
// (csrc/comm.hip; the statistics all-reduce is queued on the launch stream itself, between the reduction and the apply
// kernel, the gradient buckets on a stream of their own) or a host callback `int cb(int what, int64 a, int64 b)` that
// performs the collective with whatever the caller has (torch.distributed over gloo in the two-rank tests):
// what 0 = all-reduce (sum) of b doubles at device address a, in order on the launch stream; 1 = gradient bucket a has
// all its gradients queued.
enum { DP_SYNC_BN, DP_COMM_BN, DP_CALLBACK, DP_COMM_GRAD, DP_COMM_STREAM, DP_GRAD_BASE, DP_N_BUCKETS, DP_BUCKETS,
       DP_PENDING, DP_PARAM_BUCKET, DP_PEER, DP_COLS = 12 };
typedef int (*dp_callback_t)(int32_t what, int64_t a, int64_t b);

struct Dp {
    const int64_t *d;
    bool sync_bn() const { return d && d[DP_SYNC_BN] != 0; }
    bool buckets() const { return d && d[DP_N_BUCKETS] > 0 && d[DP_PENDING] && d[DP_PARAM_BUCKET]; }
    // the peer communicator's calls are bound to one stream (csrc/comm.hip): follow this pass's launch stream
    int bind(void *st) const {
        if (d && d[DP_PEER]) return lidog_peer_rebind_stream(P<void>(d[DP_PEER]), st);
        return 0;
    }
    int check() const {
        if (!d) return 0;
        LIDOG_REQUIRE(!d[DP_SYNC_BN] || d[DP_COMM_BN] || d[DP_CALLBACK] || d[DP_PEER],
                      "trunk: SyncBatchNorm statistics need a communicator or a callback");
        if (d[DP_N_BUCKETS] > 0) {
            LIDOG_REQUIRE(d[DP_BUCKETS] && d[DP_PENDING] && d[DP_PARAM_BUCKET], "trunk: gradient bucket tables missing");
            LIDOG_REQUIRE(d[DP_CALLBACK] || (d[DP_COMM_GRAD] && d[DP_COMM_STREAM] && d[DP_GRAD_BASE]),
                          "trunk: gradient buckets need a communicator with its stream, or a callback");
        }
        return 0;
    }
    // sum over the ranks of n doubles, in order on `st`
    int allreduce_f64(double *buf, int64_t n, void *st) const {
        // statistics messages: the one-shot peer all-reduce where it is set up (lidog_amd.comm), else the communicator
        if (d[DP_PEER] && n <= lidog_peer_max_doubles(P<void>(d[DP_PEER])))
            return lidog_peer_allreduce_f64(P<void>(d[DP_PEER]), buf, n, st);
        if (d[DP_COMM_BN]) return lidog_allreduce_f64(buf, n, P<void>(d[DP_COMM_BN]), st);
        LIDOG_REQUIRE(d[DP_CALLBACK], "trunk: no transport for the statistics all-reduce");
        int rc = reinterpret_cast<dp_callback_t>(static_cast<uintptr_t>(d[DP_CALLBACK]))(0, (int64_t)(uintptr_t)buf, n);
        LIDOG_REQUIRE(rc == 0, "trunk: the statistics all-reduce callback failed (%d)", rc);
        return 0;
    }
};

#define TRY(expr)                 \
    do {                          \
        if (!ctx.dry) {           \
            int _rc = (expr);     \
            if (_rc) return _rc;  \
        }                         \
    } while (0)

// events for forking the lane stream behind the main stream (one per convolution) and joining it again; one pool per
// process (one process drives one GPU), grown under a lock: backward passes run on autograd's device threads
hipEvent_t *event_pool(int n) {
    static std::vector<hipEvent_t> pool;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if ((int)pool.capacity() < 4096) pool.reserve(4096);   // the returned pointer stays valid while the pool grows
    while ((int)pool.size() < n) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        pool.push_back(e);
    }
    return pool.data();
}

int check_tables(const Ctx &ctx, bool grads) {
    LIDOG_REQUIRE(ctx.n_convs > 0 && ctx.n_ops > 0 && ctx.n_bufs > 0 && ctx.n_maps > 0, "trunk: empty tables");
    for (int b = 0; b < ctx.n_bufs; ++b)
        LIDOG_REQUIRE(ctx.rows(b) > 0 && ctx.ch(b) > 0, "trunk: buffer %d has no rows or channels", b);
    for (int o = 0; o < ctx.n_ops; ++o) {
        const int64_t *op = ctx.ops + (int64_t)o * TO_COLS;
        auto buf_ok = [&](int64_t b) { return b >= 0 && b < ctx.n_bufs; };
        LIDOG_REQUIRE(buf_ok(op[TO_IN]) && buf_ok(op[TO_OUT]), "trunk: op %d names a missing buffer", o);
        if (op[TO_TYPE] == OP_CAT) {
            LIDOG_REQUIRE(buf_ok(op[TO_B]), "trunk: op %d names a missing buffer", o);
            int a = (int)op[TO_IN], b = (int)op[TO_B], out = (int)op[TO_OUT];
            LIDOG_REQUIRE(ctx.rows(a) == ctx.rows(b) && ctx.rows(a) == ctx.rows(out) &&
                              ctx.ch(a) + ctx.ch(b) == ctx.ch(out) && ctx.ch(a) % 4 == 0 && ctx.ch(b) % 4 == 0,
                          "trunk: op %d concatenates mismatching buffers", o);
            continue;
        }
        LIDOG_REQUIRE(op[TO_CONV] >= 0 && op[TO_CONV] < ctx.n_convs, "trunk: op %d names a missing convolution", o);
        const int64_t *c = ctx.convs + op[TO_CONV] * TC_COLS;
        LIDOG_REQUIRE(c[TC_MAP] >= 0 && c[TC_MAP] < ctx.n_maps, "trunk: op %d names a missing map", o);
        const int64_t *m = ctx.maps + c[TC_MAP] * TM_COLS;
        const int kind = (int)c[TC_KIND];
        const int64_t n_in = kind == KIND_UP ? m[TM_NOUT] : m[TM_NIN], n_out = kind == KIND_UP ? m[TM_NIN] : m[TM_NOUT];
        LIDOG_REQUIRE(ctx.rows((int)op[TO_IN]) == n_in && ctx.rows((int)op[TO_OUT]) == n_out,
                      "trunk: op %d: buffers of %lld -> %lld rows on a map of %lld -> %lld rows", o,
                      (long long)ctx.rows((int)op[TO_IN]), (long long)ctx.rows((int)op[TO_OUT]), (long long)n_in,
                      (long long)n_out);
        LIDOG_REQUIRE(ctx.ch((int)op[TO_IN]) == c[TC_CIN] && ctx.ch((int)op[TO_OUT]) == c[TC_COUT],
                      "trunk: op %d: channel counts do not match its convolution", o);
        LIDOG_REQUIRE(m[TM_K] == c[TC_K] && m[TM_NTILES] > 0 && m[TM_TILES], "trunk: op %d: map / kernel mismatch", o);
        LIDOG_REQUIRE(c[TC_W] && (c[TC_GW] || !grads), "trunk: op %d: missing weights", o);
        if (op[TO_TYPE] == OP_CONVBN) {
            LIDOG_REQUIRE(c[TC_COUT] % 4 == 0, "trunk: op %d: BatchNorm width must be a multiple of 4", o);
            LIDOG_REQUIRE(c[TC_BNW] && c[TC_BNB] && c[TC_BNRM] && c[TC_BNRV] && ((c[TC_GBNW] && c[TC_GBNB]) || !grads),
                          "trunk: op %d: missing BatchNorm tensors", o);
            if (op[TO_RES] >= 0)
                LIDOG_REQUIRE(buf_ok(op[TO_RES]) && ctx.bytes((int)op[TO_RES]) == ctx.bytes((int)op[TO_OUT]),
                              "trunk: op %d: residual of another shape", o);
        }
        switch (kind) {
            case KIND_K3: {
                // sorted rows (output-stationary kernel) stand in for the per-row lists of the reduction pass
                const bool os = m[TM_PERM] && m[TM_WMASK] && m[TM_ORDER] && m[TM_NBR] && m[TM_NIN] == m[TM_NOUT] &&
                                c[TC_CIN] % 32 == 0 && c[TC_COUT] % 32 == 0 && lidog_get_sparse_core() == 1;
                LIDOG_REQUIRE(m[TM_PAIR_IN] && m[TM_PAIR_OUT] && c[TC_CIN] % 4 == 0 &&
                                  (os || (m[TM_RP_OUT] && m[TM_RL_OUT] && m[TM_RP_IN] && m[TM_RL_IN])),
                              "trunk: op %d: 3^3 map without pair or row lists", o);
                break;
            }
            case KIND_DOWN:
            case KIND_UP:
                LIDOG_REQUIRE(m[TM_PAIR_IN] && m[TM_PAIR_OUT] && m[TM_RP_OUT] && m[TM_RL_OUT] && c[TC_CIN] % 4 == 0,
                              "trunk: op %d: 2^3 map without pair or row lists", o);
                break;
            case KIND_1X1:
                LIDOG_REQUIRE(m[TM_IDENT] && m[TM_K] == 1, "trunk: op %d: 1x1 convolution without identity rows", o);
                break;
            case KIND_STEM:
                LIDOG_REQUIRE(m[TM_NBR] && m[TM_PAIR_IN] && m[TM_PAIR_OUT] && c[TC_CIN] == 1,
                              "trunk: op %d: stem without neighbour table", o);
                break;
            default:
                LIDOG_REQUIRE(false, "trunk: op %d: unknown convolution kind %d", o, kind);
        }
    }
    return 0;
}

inline const int32_t *tile_row(const int64_t *m, int r) { return P<const int32_t>(m[TM_TILES]) + r * m[TM_NTILES]; }

// Fusions the executor applies on top of the operator path's launch sequence (results stay bit-identical):
//   1 = BatchNorm-backward statistics in the epilogue of the data-gradient reduction that produces the gradient
//   2 = ReLU masks of the BatchNorm + residual + ReLU layers kept as bits (backward reads 1/32 of the saved output)
//   4 = BatchNorm + ReLU between the two convolutions of a block applied in the second one's staging (in_bn_of below)
int g_fusions = 7;
// the downsample branch of a layer's first block on the second stream in the forward pass (A/B: LIDOG_SIDE_FORWARD=0)
int g_side_forward = [] {
    const char *e = getenv("LIDOG_SIDE_FORWARD");
    return (e && e[0] == '0') ? 0 : 1;
}();
enum { SIDE_EVENT0 = 3000 };   // events of the forward pass's lane forks / joins, behind everything the backward pass uses
// ... and in the backward pass the same branch (BatchNorm backward + 1x1 data gradient of the downsample convolution) runs
// on a third stream of this library's, between the residual add's gradient and the block input's (A/B: LIDOG_SIDE_BACKWARD=0)
int g_side_backward = [] {
    const char *e = getenv("LIDOG_SIDE_BACKWARD");
    return (e && e[0] == '0') ? 0 : 1;
}();
enum { SIDE_BWD_EVENT0 = SIDE_EVENT0 + 64 };

hipStream_t side_stream() {
    static std::mutex lock;
    static std::vector<std::pair<int, hipStream_t>> streams;     // one per device this process drives
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    for (auto &p : streams)
        if (p.first == dev) return p.second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams.emplace_back(dev, st);
    return st;
}

// The BatchNorm + ReLU of op o needs no pass of its own when its output has exactly one reader and that reader is a 3^3
// convolution + BatchNorm on the matrix-core kernels (conv1 -> BN -> ReLU -> conv2 of a BasicBlock): the reader takes
// the raw convolution output and the BatchNorm vectors instead (lidog_sconv_*_in_bn), forward and weight gradient; the
// BatchNorm's own backward pass already works from the raw output (mask recomputed from it).  Returns the reader's op
// index, or -1.  Pure function of the tables: the forward and the backward pass decide alike.
int in_bn_reader(const Ctx &ctx, int o) {
    if (!(g_fusions & 4) || lidog_get_sparse_core() != 1) return -1;
    const int64_t *op = ctx.ops + (int64_t)o * TO_COLS;
    if (op[TO_TYPE] != OP_CONVBN || !op[TO_RELU] || op[TO_RES] >= 0) return -1;
    const int64_t out = op[TO_OUT];
    if (ctx.bufs[out * TB_COLS + TB_EXT] >= 0) return -1;
    int reader = -1, users = 0;
    for (int q = 0; q < ctx.n_ops; ++q) {
        const int64_t *u = ctx.ops + (int64_t)q * TO_COLS;
        if (u[TO_TYPE] == OP_CAT) {
            if (u[TO_IN] == out || u[TO_B] == out) ++users;
            continue;
        }
        if (u[TO_TYPE] == OP_CONVBN && u[TO_RES] == out) ++users;
        if (u[TO_IN] == out) {
            ++users;
            reader = q;
        }
    }
    if (users != 1 || reader < o) return -1;
    const int64_t *u = ctx.ops + (int64_t)reader * TO_COLS;
    const int64_t *c = ctx.convs + u[TO_CONV] * TC_COLS;
    if (u[TO_TYPE] != OP_CONVBN || c[TC_KIND] != KIND_K3 || c[TC_CIN] % 32 || c[TC_COUT] % 32) return -1;
    return reader;
}

// what the reader of a folded BatchNorm is handed instead of the normalised rows
struct InBnArgs {
    const float *pre, *mean, *invstd, *w, *b;
};

// Optional timing of the gathered GEMM's launches (bench.py's roofline figure): HIP events on the launch stream around
// every lidog_sconv_gemm call of the executor, with the launch's algorithmic FLOPs and bytes (every distinct input row
// and weight read once, every product row written once, the gather index read once: SURVEY.md 8(d)).
struct GemmRec {
    hipEvent_t e0, e1;
    double flops, bytes;
};
bool g_timing = false;
// algorithmic bytes of the two other HBM-heavy families while timing is on (DESIGN.md section 3 formulas; bench.py sets
// them against the PMC traffic of the same families): [0] MFMA weight-gradient launches, [1] their bytes
// 4 P (Cin + Cout) + 4 K Cin Cout (1 + 2 slabs / K), [2] per-row reduction launches, [3] their bytes
// 4 (P + N) C + 4 (P + N) (+ 8 N C for the saved input and mask / addend the backward-statistics form also reads)
double g_work[6] = {0, 0, 0, 0, 0, 0};   // [4] output-stationary launches, [5] their bytes 4 n (Cin + Cout) + 4 K Cin Cout + 4 K n
std::vector<GemmRec> g_recs;
std::vector<hipEvent_t> g_spare;

hipEvent_t timing_event() {
    hipEvent_t e = nullptr;
    if (!g_spare.empty()) {
        e = g_spare.back();
        g_spare.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

// n_src: rows of A (the gathered matrix)
int gemm(const Ctx &ctx, const int64_t *m, const float *A, int64_t n_src, const int32_t *gather, const float *B,
         const float *bias, int Cin, int Cout, float *out, const int32_t *scatter, void *st,
         const InBnArgs *in_bn = nullptr) {
    if (ctx.dry) return 0;
    GemmRec rec{nullptr, nullptr, 0, 0};
    if (g_timing) {
        rec.e0 = timing_event();
        rec.e1 = timing_event();
        LIDOG_REQUIRE(rec.e0 && rec.e1, "trunk: cannot create timing events");
        const double rows = (double)m[TM_P], src = (double)(n_src < m[TM_P] ? n_src : m[TM_P]);
        rec.flops = 2.0 * rows * Cin * Cout;
        rec.bytes = 4.0 * (src * Cin + rows * Cout + (double)m[TM_K] * Cin * Cout) + 4.0 * rows;
        LIDOG_CHECK_HIP(hipEventRecord(rec.e0, (hipStream_t)st));
    }
    int rc = in_bn ? lidog_sconv_gemm_in_bn(in_bn->pre, gather, B, bias, tile_row(m, 0), tile_row(m, 1), tile_row(m, 2),
                                            (int32_t)m[TM_NTILES], Cin, Cout, out, scatter, in_bn->mean, in_bn->invstd,
                                            in_bn->w, in_bn->b, 1, n_src, st)
                   : lidog_sconv_gemm(A, gather, B, bias, tile_row(m, 0), tile_row(m, 1), tile_row(m, 2),
                                      (int32_t)m[TM_NTILES], Cin, Cout, out, scatter, n_src, st);
    if (rc) return rc;
    if (g_timing) {
        LIDOG_CHECK_HIP(hipEventRecord(rec.e1, (hipStream_t)st));
        g_recs.push_back(rec);
    }
    return 0;
}

}  // namespace

// Forward pass.  rec [n_ops * 4 + n_bufs]: arena offsets of what backward needs (filled here, opaque to the caller).
// need [2]: bytes of arena / scratch this batch takes (always filled; with dry != 0 nothing is launched).
extern "C" int lidog_trunk_forward(const int64_t *convs, const double *conv_f, int32_t n_convs, const int64_t *maps,
                                   int32_t n_maps, const int64_t *ops, int32_t n_ops, const int64_t *bufs,
                                   int32_t n_bufs, const int64_t *level_rows, const int64_t *ext, void *arena,
                                   int64_t arena_bytes, void *scratch, int64_t scratch_bytes, int64_t *rec,
                                   int64_t *need, int32_t dry, const int64_t *dp_desc, void *stream, void *lane) {
    Ctx ctx{convs, conv_f, n_convs, maps, n_maps, ops, n_ops, bufs, n_bufs, level_rows, ext, dry != 0};
    Dp dp{dp_desc};
    if (int rc = check_tables(ctx, false)) return rc;
    if (int rc = dp.check()) return rc;
    if (!ctx.dry) {
        // sizes first: nothing is launched into an arena that is too small
        int64_t want[2];
        if (int rc = lidog_trunk_forward(convs, conv_f, n_convs, maps, n_maps, ops, n_ops, bufs, n_bufs, level_rows,
                                         ext, nullptr, 0, nullptr, 0, rec, want, 1, dp_desc, stream, lane))
            return rc;
        LIDOG_REQUIRE(arena && scratch && want[0] <= arena_bytes && want[1] <= scratch_bytes,
                      "trunk: forward needs %lld B of arena and %lld B of scratch, got %lld / %lld", (long long)want[0],
                      (long long)want[1], (long long)arena_bytes, (long long)scratch_bytes);
    }
    Bump ar{(char *)arena, 0, arena_bytes, 0, ctx.dry}, sc{(char *)scratch, 0, scratch_bytes, 0, ctx.dry};
    if (!ctx.dry && dp.sync_bn())
        if (int rc = dp.bind(stream)) return rc;
    int64_t *buf_off = rec + (int64_t)n_ops * REC_COLS;
    // activation buffers: external ones are the caller's tensors, the others live in the arena
    std::vector<float *> bp(n_bufs, nullptr);
    for (int b = 0; b < n_bufs; ++b) {
        int64_t e = bufs[b * TB_COLS + TB_EXT];
        if (e >= 0) {
            bp[b] = P<float>(ext[e]);
            buf_off[b] = -1;
            LIDOG_REQUIRE(ctx.dry || bp[b], "trunk: external buffer %d missing", b);
        } else {
            buf_off[b] = ar.off;
            bp[b] = (float *)ar.take(ctx.bytes(b));
        }
    }
    // One convolution (+ the statistics of its BatchNorm) and, separately, the BatchNorm's finalise + apply: under
    // SyncBatchNorm an all-reduce of the statistics sits between the two, and the first block of a layer sends the
    // statistics of conv1 and of its 1x1 downsample convolution (both read the block input) in ONE message
    // (me.BasicBlock._forward_joint_sync).
    struct Pending {
        int o;
        const int64_t *op, *c;
        float *pre, *mean, *invstd, *y;
        uint32_t *bits;
        double *sums;
        int64_t n;
        int Cout;
        float eps, mom;
    };
    const bool sync = dp.sync_bn();
    // The 1x1 downsample convolution + BatchNorm of a layer's first block (resnet_block.py:8-56; minkunet_bev.py:414-420)
    // reads the block input and is read again only by the residual add behind conv2: it runs on the second stream (idle in
    // the forward pass), next to conv1 / conv2 of its block.  `cur` / `scp`: the stream and the scratch allocator the two
    // lambdas below launch on -- the launch stream and the per-op scratch, or the lane and the ARENA (the per-op scratch is
    // recycled by the launch stream's next op while the lane may still be reading).
    void *cur = stream;
    Bump *scp = &sc;
    hipEvent_t *side_events = nullptr;
    int n_side = 0;
    std::vector<hipEvent_t> join_of(n_bufs, nullptr);     // buffer -> event after which its lane-made contents are complete
    // buffers whose BatchNorm + ReLU is applied by their reader (fusion 4): filled when the producer has run
    std::vector<InBnArgs> lazy(n_bufs, InBnArgs{nullptr, nullptr, nullptr, nullptr, nullptr});
    // sums_at: where this layer's (sum x, sum x^2, rows) go when they are part of a joint message, else NULL
    auto conv_part = [&](int o, double *sums_at, Pending &pd) -> int {
        const int64_t *op = ops + (int64_t)o * TO_COLS;
        int64_t *r = rec + (int64_t)o * REC_COLS;
        const int64_t *c = convs + op[TO_CONV] * TC_COLS;
        const int64_t *m = maps + c[TC_MAP] * TM_COLS;
        const int kind = (int)c[TC_KIND], Cin = (int)c[TC_CIN], Cout = (int)c[TC_COUT], K = (int)c[TC_K];
        const bool bn = op[TO_TYPE] == OP_CONVBN;
        const float *x = bp[op[TO_IN]];
        // an input made by a side branch on the lane (in MinkUNet34 such a buffer is only ever read as a residual, below;
        // a program that feeds it to a convolution is ordered here)
        if (join_of[op[TO_IN]] && cur == stream && !ctx.dry) {
            LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)cur, join_of[op[TO_IN]], 0));
            join_of[op[TO_IN]] = nullptr;
        }
        const InBnArgs *in_bn = lazy[op[TO_IN]].pre ? &lazy[op[TO_IN]] : nullptr;   // 3^3 convolution + BatchNorm only
        const int64_t n = ctx.rows((int)op[TO_OUT]);
        const float *W = P<const float>(c[TC_W]), *bias = P<const float>(c[TC_BIAS]);
        float *y = bp[op[TO_OUT]];
        float *pre = y, *mean = nullptr, *invstd = nullptr;
        const float eps = (float)conv_f[op[TO_CONV] * 2], mom = (float)conv_f[op[TO_CONV] * 2 + 1];
        if (bn) {
            r[REC_PRE] = ar.off;
            pre = (float *)ar.take(n * Cout * 4);
            r[REC_MEAN] = ar.off;
            mean = (float *)ar.take(Cout * 4);
            r[REC_INVSTD] = ar.off;
            invstd = (float *)ar.take(Cout * 4);
        }
        // ReLU after a residual add: the mask cannot be recomputed from the BatchNorm input alone; kept as bits
        uint32_t *bits = nullptr;
        r[REC_BITS] = 0;
        if (bn && (g_fusions & 2) && op[TO_RELU] && op[TO_RES] >= 0) {
            r[REC_BITS] = ar.off + 1;
            bits = (uint32_t *)ar.take(lidog_relu_bits_words(n, Cout) * 4);
        }
        float *rm = P<float>(c[TC_BNRM]), *rv = P<float>(c[TC_BNRV]);
        double *sums = nullptr;
        if (bn) sums = sums_at ? sums_at : (double *)scp->take((2 * Cout + 1) * 8);
        // local BatchNorm: the reduction's last kernel finalises mean / invstd / running statistics; SyncBatchNorm:
        // sums and row count only, finalised after the all-reduce
        float *f_mean = sync ? nullptr : mean, *f_invstd = sync ? nullptr : invstd;
        float *f_rm = sync ? nullptr : rm, *f_rv = sync ? nullptr : rv;
        const float f_eps = sync ? 0.f : eps, f_mom = sync ? 0.f : mom;
        bool stats_done = false;
        // sorted rows of a sparse symmetric 3^3 map: the output-stationary kernel (csrc/sconv_os.hip), no product rows
        const bool os = kind == KIND_K3 && m[TM_PERM] && m[TM_WMASK] && m[TM_ORDER] && m[TM_NBR] && m[TM_NIN] == m[TM_NOUT] &&
                        Cin % 32 == 0 && Cout % 32 == 0 && lidog_get_sparse_core() == 1;
        if (os && bn) {
            double *ws = (double *)scp->take(lidog_sconv_os_stats_ws(n, Cout) * 8);
            if (in_bn)
                TRY(lidog_sconv_os_stats_in_bn(in_bn->pre, P<const int32_t>(m[TM_NBR]), n, K, P<const int32_t>(m[TM_PERM]),
                                               P<const uint32_t>(m[TM_WMASK]), P<const int32_t>(m[TM_ORDER]), W, bias, Cin,
                                               Cout, pre, sums, ws, (double)n, f_eps, f_mom, f_mean, f_invstd, f_rm, f_rv,
                                               in_bn->mean, in_bn->invstd, in_bn->w, in_bn->b, 1, cur));
            else
            TRY(lidog_sconv_os_stats(x, P<const int32_t>(m[TM_NBR]), n, K, P<const int32_t>(m[TM_PERM]),
                                     P<const uint32_t>(m[TM_WMASK]), P<const int32_t>(m[TM_ORDER]), W, bias, Cin, Cout,
                                     pre, sums, ws, (double)n, f_eps, f_mom, f_mean, f_invstd, f_rm, f_rv, cur));
            stats_done = true;
            if (g_timing && !ctx.dry) {
                g_work[4] += 1;
                g_work[5] += 4.0 * n * (Cin + Cout) + 4.0 * K * Cin * Cout + 4.0 * K * n;
            }
        } else if (os) {
            TRY(lidog_sconv_os(x, P<const int32_t>(m[TM_NBR]), n, K, P<const int32_t>(m[TM_PERM]),
                               P<const uint32_t>(m[TM_WMASK]), P<const int32_t>(m[TM_ORDER]), W, 0, bias, nullptr, Cin,
                               Cout, pre, cur));
        } else if (kind == KIND_K3 || kind == KIND_DOWN) {
            // gathered GEMM into product rows, per-row reduction (+ BatchNorm statistics in its epilogue)
            float *T = (float *)scp->take(m[TM_P] * Cout * 4);
            if (int rc = gemm(ctx, m, x, ctx.rows((int)op[TO_IN]), P<const int32_t>(m[TM_PAIR_IN]), W, nullptr, Cin, Cout, T,
                              nullptr, cur, kind == KIND_K3 ? in_bn : nullptr))
                return rc;
            const int32_t *rp = P<const int32_t>(m[TM_RP_OUT]), *rl = P<const int32_t>(m[TM_RL_OUT]);
            if (bn) {
                double *ws = (double *)scp->take(lidog_sconv_reduce_stats_ws(n, Cout) * 8);
                TRY(lidog_sconv_reduce_rows_stats(T, rp, rl, n, Cout, bias, pre, sums, ws, (double)n, f_eps, f_mom,
                                                      f_mean, f_invstd, f_rm, f_rv, cur));
                stats_done = true;
            } else {
                TRY(lidog_sconv_reduce_rows(T, rp, rl, n, Cout, bias, nullptr, pre, cur));
            }
            if (g_timing && !ctx.dry) {
                g_work[2] += 1;
                g_work[3] += 4.0 * ((double)m[TM_P] + n) * Cout + 4.0 * ((double)m[TM_P] + n);
            }
        } else if (kind == KIND_UP) {
            // transposed 2^3 stride 2: every fine row has exactly one pair, the GEMM scatters straight into the output
            if (int rc = gemm(ctx, m, x, ctx.rows((int)op[TO_IN]), P<const int32_t>(m[TM_PAIR_OUT]), W, bias, Cin, Cout, pre,
                              P<const int32_t>(m[TM_PAIR_IN]), cur))
                return rc;
        } else if (kind == KIND_1X1) {
            if (int rc = gemm(ctx, m, x, ctx.rows((int)op[TO_IN]), nullptr, W, bias, Cin, Cout, pre, nullptr, cur)) return rc;
        } else {
            TRY(lidog_sconv_cin1(x, P<const int32_t>(m[TM_NBR]), W, bias, n, K, Cout, pre, cur));
        }
        if (bn && !stats_done) {
            int64_t wsn = lidog_bn_reduce_ws(Cout, 1);
            double *ws = wsn ? (double *)scp->take(wsn * 8) : nullptr;
            TRY(lidog_bn_stats(pre, n, Cout, 1, sums, ws, (double)n, f_eps, f_mom, f_mean, f_invstd, f_rm, f_rv, cur));
        }
        pd = Pending{o, op, c, pre, mean, invstd, y, bits, sums, n, Cout, eps, mom};
        return 0;
    };
    auto bn_part = [&](const Pending &pd) -> int {
        const int64_t *op = pd.op, *c = pd.c;
        const float *res = op[TO_RES] >= 0 ? bp[op[TO_RES]] : nullptr;
        if (res && join_of[op[TO_RES]] && !ctx.dry) {   // the residual branch was made on the lane
            LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)cur, join_of[op[TO_RES]], 0));
            join_of[op[TO_RES]] = nullptr;
        }
        if (in_bn_reader(ctx, pd.o) >= 0) {
            // no apply pass: the one reader of this output normalises the raw rows as it gathers them.  SyncBatchNorm:
            // mean / invstd / running statistics from the all-reduced sums (the apply pass would have derived them)
            if (sync)
                TRY(lidog_bn_finalize(pd.sums, -1.0, pd.Cout, pd.eps, pd.mom, pd.mean, pd.invstd, P<float>(c[TC_BNRM]),
                                      P<float>(c[TC_BNRV]), cur));
            lazy[op[TO_OUT]] = InBnArgs{pd.pre, pd.mean, pd.invstd, P<const float>(c[TC_BNW]), P<const float>(c[TC_BNB])};
            return 0;
        }
        if (sync) {   // mean / invstd / running statistics from the all-reduced sums (global count behind them) + the apply
                      // pass, one launch
            TRY(lidog_bn_apply_sync(pd.pre, pd.n, pd.Cout, pd.sums, pd.eps, pd.mom, pd.mean, pd.invstd,
                                    P<float>(c[TC_BNRM]), P<float>(c[TC_BNRV]), P<const float>(c[TC_BNW]),
                                    P<const float>(c[TC_BNB]), res, (int32_t)op[TO_RELU], pd.y, pd.bits, cur));
            return 0;
        }
        TRY(lidog_bn_apply_bits(pd.pre, pd.n, pd.Cout, 1, pd.mean, pd.invstd, P<const float>(c[TC_BNW]),
                                    P<const float>(c[TC_BNB]), res, (int32_t)op[TO_RELU], pd.y, pd.bits, cur));
        return 0;
    };
    for (int o = 0; o < n_ops; ++o) {
        const int64_t *op = ops + (int64_t)o * TO_COLS;
        sc.reset();
        if (op[TO_TYPE] == OP_CAT) {
            int a = (int)op[TO_IN], b = (int)op[TO_B];
            TRY(lidog_cat2(bp[a], ctx.ch(a), bp[b], ctx.ch(b), ctx.rows(a), bp[op[TO_OUT]], stream));
            continue;
        }
        Pending pd;
        // conv1 of a block followed by the block's downsample convolution on the same input: joint statistics message
        const int64_t *nx = o + 1 < n_ops ? ops + (int64_t)(o + 1) * TO_COLS : nullptr;
        const bool joint = sync && op[TO_TYPE] == OP_CONVBN && op[TO_FOLD] && nx && nx[TO_TYPE] == OP_CONVBN &&
                           nx[TO_IN] == op[TO_IN] && convs[nx[TO_CONV] * TC_COLS + TC_KIND] == KIND_1X1;
        if (joint) {
            const int Ca = (int)convs[op[TO_CONV] * TC_COLS + TC_COUT], Cd = (int)convs[nx[TO_CONV] * TC_COLS + TC_COUT];
            const int64_t msg_bytes = (int64_t)(2 * Ca + 1 + 2 * Cd + 1) * 8;
            // the downsample branch on the lane under SyncBatchNorm too: its convolution + statistics next to conv1's, the
            // joint message all-reduced ON THE LAUNCH STREAM as before (the order of the collectives does not change), its
            // apply pass back on the lane behind the all-reduce.  The message then lives in the arena (the lane still reads
            // its half after the launch stream's next op has recycled the per-op scratch).
            const bool jside = lane && g_side_forward && nx[TO_RES] < 0 && n_side < 16;
            double *msg = (double *)(jside ? ar.take(msg_bytes) : sc.take(msg_bytes));
            Pending pd2;
            if (jside) {
                hipEvent_t *ev = nullptr;   // fork, branch statistics ready, all-reduce done, branch output ready
                if (!ctx.dry) {
                    if (!side_events) {
                        side_events = event_pool(SIDE_EVENT0 + 64);
                        LIDOG_REQUIRE(side_events, "trunk: cannot create events");
                        side_events += SIDE_EVENT0;
                    }
                    ev = side_events + 4 * n_side;
                    LIDOG_CHECK_HIP(hipEventRecord(ev[0], (hipStream_t)stream));
                    LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)lane, ev[0], 0));
                }
                ++n_side;
                cur = lane;
                scp = &ar;
                int rc = conv_part(o + 1, msg + 2 * Ca + 1, pd2);
                cur = stream;
                scp = &sc;
                if (rc) return rc;
                if (!ctx.dry) LIDOG_CHECK_HIP(hipEventRecord(ev[1], (hipStream_t)lane));
                if (int rc1 = conv_part(o, msg, pd)) return rc1;
                if (!ctx.dry) LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ev[1], 0));
                TRY(dp.allreduce_f64(msg, 2 * Ca + 1 + 2 * Cd + 1, stream));
                if (!ctx.dry) {
                    LIDOG_CHECK_HIP(hipEventRecord(ev[2], (hipStream_t)stream));
                    LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)lane, ev[2], 0));
                }
                if (int rc1 = bn_part(pd)) return rc1;
                cur = lane;
                rc = bn_part(pd2);
                cur = stream;
                if (rc) return rc;
                if (!ctx.dry) {
                    LIDOG_CHECK_HIP(hipEventRecord(ev[3], (hipStream_t)lane));
                    join_of[nx[TO_OUT]] = ev[3];
                }
                ++o;
                continue;
            }
            if (int rc = conv_part(o, msg, pd)) return rc;
            if (int rc = conv_part(o + 1, msg + 2 * Ca + 1, pd2)) return rc;
            TRY(dp.allreduce_f64(msg, 2 * Ca + 1 + 2 * Cd + 1, stream));
            if (int rc = bn_part(pd)) return rc;
            if (int rc = bn_part(pd2)) return rc;
            ++o;
            continue;
        }
        // the block's downsample branch first, on the lane (local BatchNorm only: under SyncBatchNorm its statistics travel
        // in conv1's message, above)
        const bool side = lane && !sync && g_side_forward && op[TO_TYPE] == OP_CONVBN && op[TO_FOLD] && nx &&
                          nx[TO_TYPE] == OP_CONVBN && nx[TO_IN] == op[TO_IN] && nx[TO_RES] < 0 &&
                          convs[nx[TO_CONV] * TC_COLS + TC_KIND] == KIND_1X1 && n_side < 16;
        if (side) {
            hipEvent_t e_fork = nullptr, e_join = nullptr;
            if (!ctx.dry) {
                if (!side_events) {
                    side_events = event_pool(SIDE_EVENT0 + 64);
                    LIDOG_REQUIRE(side_events, "trunk: cannot create events");
                    side_events += SIDE_EVENT0;
                }
                e_fork = side_events[4 * n_side];
                e_join = side_events[4 * n_side + 3];
                LIDOG_CHECK_HIP(hipEventRecord(e_fork, (hipStream_t)stream));
                LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)lane, e_fork, 0));
            }
            ++n_side;
            Pending pd2;
            cur = lane;
            scp = &ar;
            int rc = conv_part(o + 1, nullptr, pd2);
            if (!rc) rc = bn_part(pd2);
            cur = stream;
            scp = &sc;
            if (rc) return rc;
            if (!ctx.dry) {
                LIDOG_CHECK_HIP(hipEventRecord(e_join, (hipStream_t)lane));
                join_of[nx[TO_OUT]] = e_join;
            }
        }
        if (int rc = conv_part(o, nullptr, pd)) return rc;
        if (op[TO_TYPE] != OP_CONVBN) continue;
        if (sync) TRY(dp.allreduce_f64(pd.sums, 2 * pd.Cout + 1, stream));
        if (int rc = bn_part(pd)) return rc;
        if (side) ++o;     // the downsample op has been done
    }
    // nothing made on the lane is left unjoined (every downsample output is a residual above; belt and braces)
    if (!ctx.dry)
        for (int b = 0; b < n_bufs; ++b)
            if (join_of[b]) LIDOG_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, join_of[b], 0));
    need[0] = ar.peak;
    need[1] = sc.peak;
    return 0;
}

// Backward pass.  ext_grad [n_ext]: incoming gradients of the external buffers (0 = none; read only).  Parameter
// gradients are written to the gW / g_bias / g_bn_* pointers of the convolution table.  lane: second stream for the
// weight gradients (NULL = in line); wgrad_first: queue a weight gradient before its data gradient's GEMM instead of
// behind it (me._WgradLane modes 1 / 2).  On return the main stream has been made to wait for the lane.
// need [3]: bytes of garena / scratch / lane_scratch.  conv_done [n_convs] (host): 1 for every convolution whose
// parameter gradients were written (a gradient reached its output), else 0.
extern "C" int lidog_trunk_backward(const int64_t *convs, const double *conv_f, int32_t n_convs, const int64_t *maps,
                                    int32_t n_maps, const int64_t *ops, int32_t n_ops, const int64_t *bufs,
                                    int32_t n_bufs, const int64_t *level_rows, const int64_t *ext,
                                    const int64_t *ext_grad, void *arena, const int64_t *rec, void *garena,
                                    int64_t garena_bytes, void *scratch, int64_t scratch_bytes, void *lane_scratch,
                                    int64_t lane_bytes, int64_t *need, int32_t *conv_done, int32_t dry,
                                    int32_t wgrad_first, const int64_t *dp_desc, void *stream, void *lane) {
    Ctx ctx{convs, conv_f, n_convs, maps, n_maps, ops, n_ops, bufs, n_bufs, level_rows, ext, dry != 0};
    Dp dp{dp_desc};
    if (int rc = check_tables(ctx, true)) return rc;
    if (int rc = dp.check()) return rc;
    if (!ctx.dry) {
        int64_t want[3];
        if (int rc = lidog_trunk_backward(convs, conv_f, n_convs, maps, n_maps, ops, n_ops, bufs, n_bufs, level_rows,
                                          ext, ext_grad, arena, rec, nullptr, 0, nullptr, 0, nullptr, 0, want,
                                          conv_done, 1, wgrad_first, dp_desc, stream, lane))
            return rc;
        LIDOG_REQUIRE(arena && garena && scratch && (lane_scratch || !lane) && want[0] <= garena_bytes &&
                          want[1] <= scratch_bytes && want[2] <= lane_bytes,
                      "trunk: backward needs %lld / %lld / %lld B (gradients / scratch / lane), got %lld / %lld / %lld",
                      (long long)want[0], (long long)want[1], (long long)want[2], (long long)garena_bytes,
                      (long long)scratch_bytes, (long long)lane_bytes);
    }
    Bump ga{(char *)garena, 0, garena_bytes, 0, ctx.dry}, sc{(char *)scratch, 0, scratch_bytes, 0, ctx.dry},
        ls{(char *)lane_scratch, 0, lane_bytes, 0, ctx.dry};
    hipStream_t main_st = (hipStream_t)stream, lane_st = (hipStream_t)lane;
    // the data gradients' gathered GEMMs run next to the lane's weight gradients: one unit per workgroup there
    // (sconv_mfma.hip:lidog_gemm_multi_suspend)
    struct MultiUnitGuard {
        bool on;
        explicit MultiUnitGuard(bool o) : on(o) { if (on) lidog_gemm_multi_suspend(1); }
        ~MultiUnitGuard() { if (on) lidog_gemm_multi_suspend(-1); }
    } multi_unit_guard(lane_st != nullptr && !ctx.dry);
    if (!ctx.dry && dp.sync_bn())
        if (int rc = dp.bind(stream)) return rc;
    hipEvent_t *events = nullptr;
    const bool sync = dp.sync_bn(), buckets = dp.buckets() && !ctx.dry;
    const int n_buckets = buckets ? (int)dp.d[DP_N_BUCKETS] : 0;
    if ((lane || buckets) && !ctx.dry) {
        // [0, n_convs): lane forks; n_convs: the final join; then two per gradient bucket (main / lane -> bucket stream)
        LIDOG_REQUIRE(n_convs + 1 + 2 * n_buckets < SIDE_EVENT0,
                      "trunk: %d convolutions + %d gradient buckets need more events than the %d in front of the side streams' range",
                      n_convs, n_buckets, (int)SIDE_EVENT0);
        events = event_pool(n_convs + 1 + 2 * n_buckets);
        LIDOG_REQUIRE(events, "trunk: cannot create events");
    }
    // A gradient bucket (a contiguous slice of the flat gradient buffer, lidog_amd.optim.GradientBuckets) is reduced as
    // soon as the last gradient of its parameters has been QUEUED: the bucket stream waits for what the launch stream
    // and the lane stream hold at that moment, nothing ever waits for the bucket stream here (the caller joins it
    // before the optimiser step).  pending[b] is shared with the caller, whose own parameters (the 2-D head) count
    // down in the same array from their gradient hooks.
    auto param_done = [&](int64_t conv, int slot) -> int {
        if (!buckets) return 0;
        const int b = P<const int32_t>(dp.d[DP_PARAM_BUCKET])[conv * 4 + slot];
        if (b < 0) return 0;
        LIDOG_REQUIRE(b < n_buckets, "trunk: parameter of convolution %lld in bucket %d of %d", (long long)conv, b, n_buckets);
        int32_t *pending = P<int32_t>(dp.d[DP_PENDING]);
        if (--pending[b] != 0) return 0;
        if (!dp.d[DP_COMM_GRAD]) {
            int rc = reinterpret_cast<dp_callback_t>(static_cast<uintptr_t>(dp.d[DP_CALLBACK]))(1, b, 0);
            LIDOG_REQUIRE(rc == 0, "trunk: the gradient bucket callback failed (%d)", rc);
            return 0;
        }
        hipStream_t cst = (hipStream_t)P<void>(dp.d[DP_COMM_STREAM]);
        hipEvent_t e_main = events[n_convs + 1 + 2 * b], e_lane = events[n_convs + 2 + 2 * b];
        LIDOG_CHECK_HIP(hipEventRecord(e_main, main_st));
        LIDOG_CHECK_HIP(hipStreamWaitEvent(cst, e_main, 0));
        if (lane) {   // every bucket waits for the lane as it stands: stream order then covers all earlier weight gradients
            LIDOG_CHECK_HIP(hipEventRecord(e_lane, lane_st));
            LIDOG_CHECK_HIP(hipStreamWaitEvent(cst, e_lane, 0));
        }
        const int64_t *sl = P<const int64_t>(dp.d[DP_BUCKETS]) + 2 * b;
        return lidog_allreduce_f32(P<float>(dp.d[DP_GRAD_BASE]) + sl[0], sl[1] - sl[0], P<void>(dp.d[DP_COMM_GRAD]), cst);
    };

    const int64_t *buf_off = rec + (int64_t)n_ops * REC_COLS;
    std::vector<float *> bp(n_bufs, nullptr);
    // gradient slot of every buffer: 0 = nothing yet, 1 = the caller's tensor (read only), 2 = ours (garena)
    std::vector<float *> gp(n_bufs, nullptr);
    std::vector<int> gs(n_bufs, 0);
    for (int b = 0; b < n_bufs; ++b) {
        int64_t e = bufs[b * TB_COLS + TB_EXT];
        if (e >= 0) {
            bp[b] = P<float>(ext[e]);
            if (ext_grad[e]) {
                gp[b] = P<float>(ext_grad[e]);
                gs[b] = 1;
            }
        } else {
            bp[b] = ctx.dry ? (float *)(uintptr_t)4096 : (float *)((char *)arena + buf_off[b]);
        }
    }
    // where a producer writes its contribution to buffer b's gradient, and what happens once it is queued
    auto target = [&](int b) { return (float *)ga.take(ctx.bytes(b)); };
    auto commit = [&](int b, float *p) -> int {
        if (gs[b] == 0) {
            gp[b] = p;
            gs[b] = 2;
            return 0;
        }
        float *dst = gs[b] == 2 ? gp[b] : (float *)ga.take(ctx.bytes(b));
        TRY(lidog_add(gp[b], p, ctx.rows(b) * ctx.ch(b), dst, stream));
        gp[b] = dst;
        gs[b] = 2;
        return 0;
    };
    int lane_used = 0;
    for (int i = 0; i < n_convs; ++i) conv_done[i] = 0;
    // producer of every buffer and its first consumer in forward order = the LAST one to add to its gradient here
    std::vector<int> producer(n_bufs, -1), first_consumer(n_bufs, n_ops);
    for (int o = n_ops - 1; o >= 0; --o) {
        const int64_t *op = ops + (int64_t)o * TO_COLS;
        producer[op[TO_OUT]] = o;
        first_consumer[op[TO_IN]] = o;
        if (op[TO_TYPE] == OP_CAT) first_consumer[op[TO_B]] = o;
        if (op[TO_TYPE] == OP_CONVBN && op[TO_RES] >= 0) first_consumer[op[TO_RES]] = o;
    }
    // BatchNorm-backward sums of op o already produced by the reduction that completed its output gradient
    std::vector<double *> bwd_sums(n_ops, nullptr);
    // Side branch: the downsample convolution of a layer's first block (1x1 + BatchNorm, no ReLU; its output is only the
    // residual of conv2's BatchNorm) has its backward -- BatchNorm reduce + apply, 1x1 data gradient -- on a third stream,
    // from the moment conv2's BatchNorm backward has written the residual's gradient until conv1's data gradient adds the
    // branch's contribution to the block input's gradient.  Under SyncBatchNorm the branch's statistics message is still
    // all-reduced ON THE LAUNCH STREAM, at the op's place in the order: the launch stream waits for the branch's reduction
    // (long finished: conv2's data gradient was queued in between), the branch's apply pass waits for the all-reduce.
    // Gradient buckets watch the launch stream and the lane only; that covers a side op's parameter gradients when the
    // launch stream has waited for its reduction (SyncBatchNorm), not otherwise (local BatchNorm + buckets: in line).
    // Same kernels, same arguments: same bits.
    const bool side_on = g_side_backward && lane && (sync || !dp.buckets());
    hipStream_t side_st = nullptr;
    hipEvent_t *side_ev = nullptr;
    int n_side = 0;
    std::vector<int> is_side(n_ops, 0);
    if (side_on)
        for (int o = 1; o < n_ops; ++o) {
            const int64_t *op = ops + (int64_t)o * TO_COLS, *pv = ops + (int64_t)(o - 1) * TO_COLS;
            if (op[TO_TYPE] == OP_CONVBN && pv[TO_TYPE] == OP_CONVBN && pv[TO_FOLD] && pv[TO_IN] == op[TO_IN] &&
                op[TO_RES] < 0 && !op[TO_RELU] && convs[op[TO_CONV] * TC_COLS + TC_KIND] == KIND_1X1 &&
                bufs[op[TO_OUT] * TB_COLS + TB_EXT] < 0 && bufs[op[TO_IN] * TB_COLS + TB_EXT] != 0)
                is_side[o] = 1;
        }
    std::vector<hipEvent_t> res_ready(n_bufs, nullptr), join_evt(n_bufs, nullptr);
    std::vector<char> res_ok(n_bufs, 0);     // the buffer's gradient is one residual gradient (decided alike in dry runs)
    hipEvent_t side_last = nullptr;
    auto join_side = [&](int b) -> int {      // the launch stream is about to read what the side branch wrote for buffer b
        if (join_evt[b] && !ctx.dry) LIDOG_CHECK_HIP(hipStreamWaitEvent(main_st, join_evt[b], 0));
        join_evt[b] = nullptr;
        return 0;
    };
    for (int o = n_ops - 1; o >= 0; --o) {
        const int64_t *op = ops + (int64_t)o * TO_COLS;
        const int64_t *r = rec + (int64_t)o * REC_COLS;
        sc.reset();
        const int out_b = (int)op[TO_OUT], in_b = (int)op[TO_IN];
        if (gs[out_b] == 0) continue;  // nothing reached this output: no gradients below it on this branch
        if (int rc = join_side(out_b)) return rc;
        // this op on the side stream?  (its output gradient is exactly one BatchNorm's residual gradient, and nothing has
        // reached the block input yet: the branch is the first contributor there, no add kernel)
        const bool side_op = is_side[o] && res_ok[out_b] && gs[in_b] == 0 && n_side < 16;
        void *cur = stream;
        Bump *scp = &sc;
        if (side_op) {
            if (!ctx.dry) {
                if (!side_st) {
                    side_st = side_stream();
                    side_ev = event_pool(SIDE_BWD_EVENT0 + 64);
                    LIDOG_REQUIRE(side_st && side_ev, "trunk: cannot create the side stream");
                    side_ev += SIDE_BWD_EVENT0;
                }
                LIDOG_CHECK_HIP(hipStreamWaitEvent(side_st, res_ready[out_b], 0));
                cur = (void *)side_st;
            }
            scp = &ga;      // per-op scratch is recycled by the launch stream's next op
        }
        if (op[TO_TYPE] == OP_CAT) {
            int a = in_b, b = (int)op[TO_B];
            float *ga_ = target(a), *gb_ = target(b);
            TRY(lidog_split2(gp[out_b], ctx.ch(a), ctx.ch(b), ctx.rows(a), ga_, gb_, stream));
            if (int rc = commit(a, ga_)) return rc;
            if (int rc = commit(b, gb_)) return rc;
            continue;
        }
        const int64_t *c = convs + op[TO_CONV] * TC_COLS;
        const int64_t *m = maps + c[TC_MAP] * TM_COLS;
        const int kind = (int)c[TC_KIND], Cin = (int)c[TC_CIN], Cout = (int)c[TC_COUT], K = (int)c[TC_K];
        const int64_t n = ctx.rows(out_b), n_in = ctx.rows(in_b);
        conv_done[op[TO_CONV]] = 1;
        const float *gout = gp[out_b];
        if (op[TO_TYPE] == OP_CONVBN) {
            const float *pre = ctx.dry ? nullptr : (const float *)((char *)arena + r[REC_PRE]);
            const float *mean = ctx.dry ? nullptr : (const float *)((char *)arena + r[REC_MEAN]);
            const float *invstd = ctx.dry ? nullptr : (const float *)((char *)arena + r[REC_INVSTD]);
            const bool relu = op[TO_RELU] != 0, has_res = op[TO_RES] >= 0;
            const bool mask_from_x = relu && !has_res;  // Cout % 4 == 0 checked above
            const uint32_t *mbits = (!ctx.dry && r[REC_BITS]) ? (const uint32_t *)((char *)arena + r[REC_BITS] - 1) : nullptr;
            const float *ymask = (relu && !mask_from_x && !r[REC_BITS]) ? bp[out_b] : nullptr;
            const float *bnw = P<const float>(c[TC_BNW]), *bnb = P<const float>(c[TC_BNB]);
            double *sums = bwd_sums[o];
            if (!sums) {
                sums = (double *)scp->take((2 * Cout + 1) * 8);
                int64_t wsn = lidog_bn_reduce_ws(Cout, 1);
                double *ws = wsn ? (double *)scp->take(wsn * 8) : nullptr;
                TRY(lidog_bn_bwd_reduce_bits(gout, pre, ymask, mbits, n, Cout, 1, mean, invstd, sums, ws, (double)n,
                                                 P<float>(c[TC_GBNW]), P<float>(c[TC_GBNB]), mask_from_x ? bnw : nullptr,
                                                 mask_from_x ? bnb : nullptr, cur));
            }
            float *dx = (float *)ga.take(n * Cout * 4);
            float *dres = has_res ? target((int)op[TO_RES]) : nullptr;
            // SyncBatchNorm: (sum dy', sum dy' xhat, rows) summed over the ranks; the apply kernel reads the global count
            if (sync && side_op && !ctx.dry) {
                hipEvent_t e1 = side_ev[4 * n_side + 2], e2 = side_ev[4 * n_side + 3];
                LIDOG_CHECK_HIP(hipEventRecord(e1, side_st));
                LIDOG_CHECK_HIP(hipStreamWaitEvent(main_st, e1, 0));
                TRY(dp.allreduce_f64(sums, 2 * Cout + 1, stream));
                LIDOG_CHECK_HIP(hipEventRecord(e2, main_st));
                LIDOG_CHECK_HIP(hipStreamWaitEvent(side_st, e2, 0));
            } else if (sync) {
                TRY(dp.allreduce_f64(sums, 2 * Cout + 1, stream));
            }
            TRY(lidog_bn_bwd_apply_bits(gout, pre, ymask, mbits, n, Cout, 1, mean, invstd, bnw, sums,
                                             sync ? -1.0 : (double)n, dx, dres, nullptr, nullptr,
                                             mask_from_x ? bnb : nullptr, cur));
            if (has_res) {
                const int rb = (int)op[TO_RES];
                const bool first = gs[rb] == 0;
                if (int rc = commit(rb, dres)) return rc;
                // the residual branch may start now (if it is a side branch and this was its whole gradient)
                if (side_on && first && producer[rb] >= 0 && is_side[producer[rb]] && n_side < 16) res_ok[rb] = 1;
                if (res_ok[rb] && !ctx.dry) {
                    if (!side_ev) {
                        side_st = side_stream();
                        side_ev = event_pool(SIDE_BWD_EVENT0 + 64);
                        LIDOG_REQUIRE(side_st && side_ev, "trunk: cannot create the side stream");
                        side_ev += SIDE_BWD_EVENT0;
                    }
                    res_ready[rb] = side_ev[4 * n_side];
                    LIDOG_CHECK_HIP(hipEventRecord(res_ready[rb], main_st));
                }
            }
            gout = dx;
        }
        // ---- convolution backward (me._SparseConvFn.backward)
        const float *x = bp[in_b];
        // input normalised on the fly in the forward pass (fusion 4): the weight gradient does the same from the
        // producer's raw output
        InBnArgs in_bn{nullptr, nullptr, nullptr, nullptr, nullptr};
        {
            const int po = producer[in_b];
            if (po >= 0 && in_bn_reader(ctx, po) == o) {
                const int64_t *pc = convs + ops[(int64_t)po * TO_COLS + TO_CONV] * TC_COLS;
                const int64_t *pr = rec + (int64_t)po * REC_COLS;
                in_bn = InBnArgs{ctx.dry ? (const float *)(uintptr_t)4096 : (const float *)((char *)arena + pr[REC_PRE]),
                                 ctx.dry ? nullptr : (const float *)((char *)arena + pr[REC_MEAN]),
                                 ctx.dry ? nullptr : (const float *)((char *)arena + pr[REC_INVSTD]),
                                 P<const float>(pc[TC_BNW]), P<const float>(pc[TC_BNB])};
            }
        }
        const int32_t *g_in, *g_out;
        if (kind == KIND_1X1) {
            g_in = g_out = P<const int32_t>(m[TM_IDENT]);
        } else if (kind == KIND_UP) {
            g_in = P<const int32_t>(m[TM_PAIR_OUT]);
            g_out = P<const int32_t>(m[TM_PAIR_IN]);
        } else {
            g_in = P<const int32_t>(m[TM_PAIR_IN]);
            g_out = P<const int32_t>(m[TM_PAIR_OUT]);
        }
        bool wgrad_done = false;
        auto queue_wgrad = [&]() -> int {
            wgrad_done = true;
            const int n_items = (int)c[TC_NITEMS];
            int slabs = lidog_sconv_wgrad_slabs(Cin, Cout, n_items);
            int64_t pbytes = (int64_t)(slabs > 1 ? slabs : 1) * Cin * Cout * 4;
            float *partial;
            void *st = cur;
            if (lane) {
                ls.reset();
                partial = (float *)ls.take(pbytes);
                if (!ctx.dry) {
                    hipEvent_t ev = events[op[TO_CONV]];
                    LIDOG_CHECK_HIP(hipEventRecord(ev, (hipStream_t)cur));
                    LIDOG_CHECK_HIP(hipStreamWaitEvent(lane_st, ev, 0));
                }
                st = lane;
                lane_used = 1;
            } else {
                partial = (float *)scp->take(pbytes);
            }
            if (in_bn.pre)
                TRY(lidog_sconv_wgrad_in_bn(in_bn.pre, g_in, gout, g_out, P<const int32_t>(c[TC_ITEMS]), n_items,
                                            P<const int32_t>(c[TC_ITEMOFF]), K, Cin, Cout, partial, P<float>(c[TC_GW]),
                                            in_bn.mean, in_bn.invstd, in_bn.w, in_bn.b, 1, st));
            else
            TRY(lidog_sconv_wgrad(x, g_in, gout, g_out, P<const int32_t>(c[TC_ITEMS]), n_items,
                                  P<const int32_t>(c[TC_ITEMOFF]), K, Cin, Cout, partial, P<float>(c[TC_GW]), st));
            if (g_timing && !ctx.dry && Cin % 32 == 0 && Cout % 32 == 0) {
                g_work[0] += 1;
                g_work[1] += 4.0 * (double)m[TM_P] * (Cin + Cout) + 4.0 * (double)Cin * Cout * (K + 2.0 * (slabs > 1 ? slabs : 1));
            }
            return 0;
        };
        const bool behind = lane && !wgrad_first;
        if (!behind)
            if (int rc = queue_wgrad()) return rc;
        const bool need_dgrad = bufs[in_b * TB_COLS + TB_EXT] != 0 && kind != KIND_STEM;  // ext slot 0 = input features
        if (need_dgrad) {
            if (int rc = join_side(in_b)) return rc;    // what a side branch has written to this buffer's gradient
            const float *Wt = P<const float>(c[TC_WT]);
            if (!Wt) {
                float *w = (float *)ga.take((int64_t)K * Cin * Cout * 4);
                TRY(lidog_transpose_kernel(P<const float>(c[TC_W]), K, Cin, Cout, w, cur));
                Wt = w;
            }
            if (kind == KIND_1X1 && Cout <= 8 && Cin % 4 == 0 && gs[in_b] != 0) {
                // the classifier's data gradient on rows that already hold a gradient (the BEV head's): product and sum
                // in one pass, the bits of the product pass followed by commit()'s lidog_add
                float *dst = gs[in_b] == 2 ? gp[in_b] : (float *)ga.take(ctx.bytes(in_b));
                TRY(lidog_sconv_gemm_addend(gout, nullptr, Wt, tile_row(m, 0), tile_row(m, 1), tile_row(m, 2),
                                            (int32_t)m[TM_NTILES], Cout, Cin, gp[in_b], dst, nullptr, cur));
                gp[in_b] = dst;
                gs[in_b] = 2;
            } else if (kind == KIND_1X1) {
                float *gx = target(in_b);
                if (int rc = gemm(ctx, m, gout, n, nullptr, Wt, nullptr, Cout, Cin, gx, nullptr, cur)) return rc;
                if (int rc = commit(in_b, gx)) return rc;
                if (side_op && !ctx.dry) {
                    join_evt[in_b] = side_ev[4 * n_side + 1];
                    LIDOG_CHECK_HIP(hipEventRecord(join_evt[in_b], side_st));
                    side_last = join_evt[in_b];
                }
            } else if (kind == KIND_DOWN) {
                // every fine (input) row has exactly one pair: the GEMM scatters straight into the gradient
                float *gx = target(in_b);
                if (int rc = gemm(ctx, m, gout, n, g_out, Wt, nullptr, Cout, Cin, gx, g_in, stream)) return rc;
                if (int rc = commit(in_b, gx)) return rc;
            } else if (kind == KIND_K3 && m[TM_PERM] && m[TM_WMASK] && m[TM_ORDER] && m[TM_NBR] && m[TM_NIN] == m[TM_NOUT] &&
                       Cin % 32 == 0 && Cout % 32 == 0 && lidog_get_sparse_core() == 1) {
                // output-stationary data gradient over the mirrored offsets of the map's sorted rows; what reached the
                // input buffer earlier (the residual branch of a block) is added in its epilogue, as the reduction pass
                // does.  The BatchNorm-backward sums of the producing layer then come from the stand-alone reduction
                // (the operator path's kernel: the same bits).
                const bool folds = op[TO_FOLD] && gs[in_b] != 0;
                float *gx = target(in_b);
                TRY(lidog_sconv_os(gout, P<const int32_t>(m[TM_NBR]), n_in, K, P<const int32_t>(m[TM_PERM]),
                                   P<const uint32_t>(m[TM_WMASK]), P<const int32_t>(m[TM_ORDER]), Wt, 1, nullptr,
                                   folds ? gp[in_b] : nullptr, Cout, Cin, gx, stream));
                if (g_timing && !ctx.dry) {
                    g_work[4] += 1;
                    g_work[5] += 4.0 * n_in * (Cin + Cout + (folds ? Cin : 0)) + 4.0 * K * Cin * Cout + 4.0 * K * n_in;
                }
                if (!wgrad_done)
                    if (int rc = queue_wgrad()) return rc;
                if (folds) {
                    gp[in_b] = gx;
                    gs[in_b] = 2;
                } else if (int rc = commit(in_b, gx)) {
                    return rc;
                }
            } else {
                float *T = (float *)sc.take(m[TM_P] * Cin * 4);
                if (int rc = gemm(ctx, m, gout, n, g_out, Wt, nullptr, Cout, Cin, T, nullptr, stream)) return rc;
                if (!wgrad_done)
                    if (int rc = queue_wgrad()) return rc;
                // K3: rows of the input side; UP: the coarse rows are the map's OUTPUT side
                const int32_t *rp = P<const int32_t>(kind == KIND_UP ? m[TM_RP_OUT] : m[TM_RP_IN]);
                const int32_t *rl = P<const int32_t>(kind == KIND_UP ? m[TM_RL_OUT] : m[TM_RL_IN]);
                float *gx = target(in_b);
                // This reduction completes the gradient of in_b when nothing reaches that buffer after it: it is the
                // buffer's first consumer in forward order, and whatever arrived before is either absent or enters
                // as the addend.  If a BatchNorm produced the buffer, its backward sums ride in the epilogue.
                const int po = producer[in_b];
                const bool folds = op[TO_FOLD] && gs[in_b] != 0;
                const int64_t *pop = po >= 0 ? ops + (int64_t)po * TO_COLS : nullptr;
                if ((g_fusions & 1) && pop && pop[TO_TYPE] == OP_CONVBN && first_consumer[in_b] == o &&
                    (folds || gs[in_b] == 0) && Cin % 4 == 0 && Cin / 4 <= 256) {
                    const int64_t *pc = convs + pop[TO_CONV] * TC_COLS;
                    const int64_t *pr = rec + (int64_t)po * REC_COLS;
                    const float *p_pre = ctx.dry ? nullptr : (const float *)((char *)arena + pr[REC_PRE]);
                    const float *p_mean = ctx.dry ? nullptr : (const float *)((char *)arena + pr[REC_MEAN]);
                    const float *p_invstd = ctx.dry ? nullptr : (const float *)((char *)arena + pr[REC_INVSTD]);
                    const bool p_relu = pop[TO_RELU] != 0, p_from_x = p_relu && pop[TO_RES] < 0;
                    const uint32_t *p_bits =
                        (!ctx.dry && pr[REC_BITS]) ? (const uint32_t *)((char *)arena + pr[REC_BITS] - 1) : nullptr;
                    const float *p_y = (p_relu && !p_from_x && !pr[REC_BITS]) ? bp[in_b] : nullptr;
                    double *sums = (double *)ga.take((2 * Cin + 1) * 8);   // lives until the producer's turn
                    double *ws = (double *)sc.take(lidog_bn_reduce_ws(Cin, 1) * 8);
                    TRY(lidog_sconv_reduce_rows_bwdstats(T, rp, rl, n_in, Cin, folds ? gp[in_b] : nullptr, gx, p_pre,
                                                              p_y, p_bits, p_mean, p_invstd,
                                                              p_from_x ? P<const float>(pc[TC_BNW]) : nullptr,
                                                              p_from_x ? P<const float>(pc[TC_BNB]) : nullptr, sums, ws,
                                                              (double)n_in, P<float>(pc[TC_GBNW]), P<float>(pc[TC_GBNB]),
                                                              stream));
                    bwd_sums[po] = sums;
                    gp[in_b] = gx;
                    gs[in_b] = 2;
                    if (g_timing && !ctx.dry) g_work[3] += 8.0 * (double)n_in * Cin;   // saved input + mask / addend
                } else if (folds) {
                    // the residual branch's gradient of the block input enters the sum in the reduction's epilogue
                    TRY(lidog_sconv_reduce_rows(T, rp, rl, n_in, Cin, nullptr, gp[in_b], gx, stream));
                    gp[in_b] = gx;
                    gs[in_b] = 2;
                } else {
                    TRY(lidog_sconv_reduce_rows(T, rp, rl, n_in, Cin, nullptr, nullptr, gx, stream));
                    if (int rc = commit(in_b, gx)) return rc;
                }
                if (g_timing && !ctx.dry) {
                    g_work[2] += 1;
                    g_work[3] += 4.0 * ((double)m[TM_P] + n_in) * Cin + 4.0 * ((double)m[TM_P] + n_in);
                }
            }
        }
        if (!wgrad_done)
            if (int rc = queue_wgrad()) return rc;
        if (c[TC_BIAS] && c[TC_GBIAS]) {
            double *ws = (double *)scp->take(lidog_colsum_ws(Cout) * 8);
            TRY(lidog_colsum(gout, n, Cout, P<float>(c[TC_GBIAS]), ws, cur));
        }
        if (side_op) {
            if (!ctx.dry && !join_evt[in_b]) {   // no data gradient was asked for: still order the parameter gradients
                side_last = side_ev[4 * n_side + 1];
                LIDOG_CHECK_HIP(hipEventRecord(side_last, side_st));
            }
            ++n_side;
        }
        // every parameter gradient of this convolution is queued now (kernel: lane or launch stream; bias and BatchNorm
        // gains / biases: launch stream)
        if (!ctx.dry) {
            if (int rc = param_done(op[TO_CONV], 0)) return rc;
            if (c[TC_BIAS] && c[TC_GBIAS])
                if (int rc = param_done(op[TO_CONV], 1)) return rc;
            if (op[TO_TYPE] == OP_CONVBN) {
                if (int rc = param_done(op[TO_CONV], 2)) return rc;
                if (int rc = param_done(op[TO_CONV], 3)) return rc;
            }
        }
    }
    if (side_last && !ctx.dry) LIDOG_CHECK_HIP(hipStreamWaitEvent(main_st, side_last, 0));   // BatchNorm gradients of the side ops
    if (lane_used && !ctx.dry) {
        hipEvent_t ev = events[n_convs];
        LIDOG_CHECK_HIP(hipEventRecord(ev, lane_st));
        LIDOG_CHECK_HIP(hipStreamWaitEvent(main_st, ev, 0));
    }
    need[0] = ga.peak;
    need[1] = sc.peak;
    need[2] = ls.peak;
    return 0;
}

// readers [n_ops]: for every op the op index of the convolution that normalises its output while staging it (fusion 4,
// in_bn_reader above), -1 where the BatchNorm keeps its own apply pass.  Host only; what the two passes will decide.
extern "C" int lidog_trunk_in_bn_readers(const int64_t *convs, int32_t n_convs, const int64_t *maps, int32_t n_maps,
                                         const int64_t *ops, int32_t n_ops, const int64_t *bufs, int32_t n_bufs,
                                         int32_t *readers) {
    LIDOG_REQUIRE(convs && maps && ops && bufs && readers, "trunk_in_bn_readers: null argument");
    Ctx ctx{convs, nullptr, n_convs, maps, n_maps, ops, n_ops, bufs, n_bufs, nullptr, nullptr, true};
    for (int o = 0; o < n_ops; ++o) readers[o] = in_bn_reader(ctx, o);
    return 0;
}

// Which fusions the executor applies (bit mask, see g_fusions; default: all).  Returns the previous mask; < 0 only reads.
extern "C" int32_t lidog_trunk_fusions(int32_t mask) {
    int32_t old = g_fusions;
    if (mask >= 0) g_fusions = mask;
    return old;
}

// Timing of the executor's gathered-GEMM launches (see GemmRec above).  on != 0: every launch from now on is bracketed
// by events; lidog_trunk_gemm_timing_read waits for the recorded launches and returns (launches, total ms, algorithmic
// FLOPs, algorithmic bytes) in out[0..3], then forgets them.  Not thread-safe: one timing client per process.
extern "C" int lidog_trunk_gemm_timing(int32_t on) {
    g_timing = on != 0;
    return 0;
}

// [0] MFMA weight-gradient launches, [1] their algorithmic bytes, [2] per-row reduction launches, [3] their algorithmic
// bytes, [4] output-stationary convolution launches, [5] theirs, accumulated by the executor while
// lidog_trunk_gemm_timing is on; reading resets them
extern "C" int lidog_trunk_work_read(double *out) {
    for (int i = 0; i < 6; ++i) {
        out[i] = g_work[i];
        g_work[i] = 0;
    }
    return 0;
}

extern "C" int lidog_trunk_gemm_timing_read(double *out) {
    out[0] = out[1] = out[2] = out[3] = 0;
    for (GemmRec &r : g_recs) {
        float ms = 0;
        LIDOG_CHECK_HIP(hipEventSynchronize(r.e1));
        LIDOG_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        out[0] += 1;
        out[1] += ms;
        out[2] += r.flops;
        out[3] += r.bytes;
        g_spare.push_back(r.e0);
        g_spare.push_back(r.e1);
    }
    g_recs.clear();
    return 0;
}
