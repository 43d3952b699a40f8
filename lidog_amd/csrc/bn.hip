// BatchNorm (training statistics in double, fused affine + residual + ReLU), ReLU and add on
// COO feature matrices [rows, C] and on NCHW images (hw = H*W).  Reference semantics:
// nn.BatchNorm1d inside ME.MinkowskiBatchNorm (utils/models/minkunet_bev.py:60,406-408),
// nn.BatchNorm2d of utils/models/conv2d.py:18,21; eps 1e-5, momentum 0.1, biased variance for
// normalisation, unbiased for running_var.
#include <stdlib.h>

#include <mutex>
#include <map>
#include <unordered_map>

#include "common.h"
#include "stats_tail.h"

// MODE 0: (x, x*x)            -> statistics
// MODE 1: (dy', dy' * xhat)   -> backward reductions, dy' = dy * (relu_y > 0) when relu_y given
template <int MODE>
__device__ __forceinline__ void red_terms(float x, float dy, float ry, bool has_relu, float mean, float invstd,
                                          double &t0, double &t1) {
    if (MODE == 0) {
        t0 += (double)x;
        t1 += (double)x * (double)x;
    } else {
        float g = (has_relu && !(ry > 0.f)) ? 0.f : dy;
        float xh = (x - mean) * invstd;
        t0 += (double)g;
        t1 += (double)g * (double)xh;
    }
}

// rw / rb (MODE 1, ry == NULL): BatchNorm weight and bias -- the ReLU mask is then recomputed from x with the forward
// pass's own expression ((x - mean) * invstd * w + b > 0, same operation order, contraction off: the same bits)
// instead of being read from the saved output: one tensor less to stream for a BatchNorm + ReLU without residual
template <int MODE>
__global__ __launch_bounds__(256) void k_colreduce_nc4(const float4 *__restrict__ x, const float4 *__restrict__ dy,
                                                       const float4 *__restrict__ ry, int64_t n, int C4,
                                                       const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, StatsTail tail,
                                                       const float *__restrict__ rw = nullptr,
                                                       const float *__restrict__ rb = nullptr,
                                                       const uint32_t *__restrict__ rbits = nullptr) {
    __shared__ double red[256 * 8];
    const int RB = 256 / C4;
    const int tid = threadIdx.x;
    const int r = tid / C4, c4 = tid % C4;
    const bool active = r < RB;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float m[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1}, gw[4] = {0, 0, 0, 0}, gb[4] = {1, 1, 1, 1};
    const bool from_x = MODE == 1 && ry == nullptr && rw != nullptr;
    if (MODE == 1 && active) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = mean[c4 * 4 + j]; is[j] = invstd[c4 * 4 + j]; }
        if (from_x) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { gw[j] = rw[c4 * 4 + j]; gb[j] = rb[c4 * 4 + j]; }
        }
    }
    if (active) {
        const int64_t step = (int64_t)gridDim.x * RB;
        // 4 independent rows in flight per lane: the pass is a pure HBM/L2 stream.  Loads are unconditional
        // (row clamped, value selected afterwards): a branch around a load costs a full wait per element
        for (int64_t row0 = (int64_t)blockIdx.x * RB + r; row0 < n; row0 += 4 * step) {
            float4 v[4], g[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t row = row0 + u * step;
                row = row < n ? row : n - 1;
                v[u] = x[row * C4 + c4];
                if (MODE == 1) {
                    g[u] = dy[row * C4 + c4];
                    y[u] = ry ? ry[row * C4 + c4] : make_float4(1, 1, 1, 1);
                    if (rbits) y[u] = lidog_relu_bits_as_float4(rbits, row * C4 + c4);
                } else {
                    g[u] = make_float4(0, 0, 0, 0);
                    y[u] = make_float4(1, 1, 1, 1);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = row0 + u * step < n;
                const float4 z = make_float4(0, 0, 0, 0);
                float4 vv = (MODE == 0 && !ok) ? z : v[u];
                float4 gg = ok ? g[u] : z;
                float4 yy = y[u];
                if (from_x) {   // the forward pass's pre-activation, bit for bit (k_bn_apply4)
                    yy.x = (vv.x - m[0]) * is[0] * gw[0] + gb[0];
                    yy.y = (vv.y - m[1]) * is[1] * gw[1] + gb[1];
                    yy.z = (vv.z - m[2]) * is[2] * gw[2] + gb[2];
                    yy.w = (vv.w - m[3]) * is[3] * gw[3] + gb[3];
                }
                const bool has_relu = ry != nullptr || from_x || rbits != nullptr;
                red_terms<MODE>(vv.x, gg.x, yy.x, has_relu, m[0], is[0], a[0], a[4]);
                red_terms<MODE>(vv.y, gg.y, yy.y, has_relu, m[1], is[1], a[1], a[5]);
                red_terms<MODE>(vv.z, gg.z, yy.z, has_relu, m[2], is[2], a[2], a[6]);
                red_terms<MODE>(vv.w, gg.w, yy.w, has_relu, m[3], is[3], a[3], a[7]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = a[j];
    __syncthreads();
    if (active && r == 0) {
        for (int rr = 1; rr < RB; ++rr)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += red[(rr * C4 + c4) * 8 + j];
    }
    // one partial row per workgroup, added in a fixed order by the last workgroup to arrive (stats_tail.h): no atomics
    // on the sums (512 workgroups adding to the same 2C addresses cost ~25 us of serialised L2 atomics),
    // bit-reproducible, and no finishing launch behind this one
    lidog_stats_tail(tail, active && r == 0, c4, a);
}

// sums[col] = sum over blocks (ascending, fixed tree) of partial[b][col] for the 2C columns (C first sums, C second
// sums).  One workgroup per channel pair, both sums of a channel in the same workgroup, so what follows the
// reduction rides in the same launch instead of a microsecond-sized kernel of its own per layer:
//   count > 0   -> sums[2C] = count   (SyncBatchNorm all-reduces the row count together with the sums)
//   fin.mean    -> mean / invstd / running statistics from (sum x, sum x^2) and count  (= k_bn_finalize)
//   fin.db / dw -> float copies of (first sums, second sums)                            (= k_bn_param_grads)
#define bn_finalize_channel lidog_bn_finalize_channel   // stats_tail.h

__global__ __launch_bounds__(256) void k_sums_finish(const double *__restrict__ partial, int nb, int C,
                                                     double *__restrict__ sums, double count, BnFinish fin) {
    __shared__ double red[256];
    const int cl = threadIdx.x & 3, bl = threadIdx.x >> 2;
    const int ch = blockIdx.x * 2 + (cl & 1), half = cl >> 1;
    const int C2 = 2 * C, col = half * C + ch;
    const bool valid = ch < C;
    double s0 = 0, s1 = 0;
    if (valid) {
        int b = bl;
        for (; b + 64 < nb; b += 128) {
            s0 += partial[(size_t)b * C2 + col];
            s1 += partial[(size_t)(b + 64) * C2 + col];
        }
        if (b < nb) s0 += partial[(size_t)b * C2 + col];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        if (bl < d) red[threadIdx.x] += red[threadIdx.x + d * 4];
        __syncthreads();
    }
    if (bl == 0 && valid) {
        sums[col] = red[cl];
        if (fin.db && half == 0) fin.db[ch] = (float)red[cl];
        if (fin.dw && half == 1) fin.dw[ch] = (float)red[cl];
        if (fin.mean && half == 0) bn_finalize_channel(red[cl], red[cl + 2], count, ch, fin);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && count > 0) sums[C2] = count;
}

int lidog_launch_sums_finish(const double *partial, int nb, int C, double *sums, double count, BnFinish fin,
                             hipStream_t st) {
    k_sums_finish<<<(unsigned)((C + 1) / 2), 256, 0, st>>>(partial, nb, C, sums, count, fin);
    return 0;
}

// the same three follow-ups for the reductions that accumulate with atomics (NCHW planes, C % 4 != 0)
__global__ void k_sums_post(double *__restrict__ sums, int C, double count, BnFinish fin) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && count > 0) sums[2 * C] = count;
    if (c >= C) return;
    if (fin.db) fin.db[c] = (float)sums[c];
    if (fin.dw) fin.dw[c] = (float)sums[C + c];
    if (fin.mean) bn_finalize_channel(sums[c], sums[C + c], count, c, fin);
}

// generic layout (NCHW planes, or [n,C] with C % 4 != 0 as hw = 1 "planes" of strided access)
template <int MODE>
__global__ __launch_bounds__(256) void k_colreduce_plane(const float *__restrict__ x, const float *__restrict__ dy,
                                                         const float *__restrict__ ry, int64_t hw, int C,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         double *__restrict__ partial) {
    __shared__ double red[2][4];
    const int64_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const int64_t base = plane * hw;
    const int64_t chunk = (hw + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = (i0 + chunk < hw) ? i0 + chunk : hw;
    double t0 = 0, t1 = 0;
    float m = 0.f, is = 1.f;
    if (MODE == 1) { m = mean[c]; is = invstd[c]; }
    // four elements of a thread in flight per round (1 024 consecutive floats per workgroup and round, every load
    // issued before the first use; tail elements read the chunk's last one and are masked): the pass is a pure stream,
    // and one 4-byte load per lane and iteration left it at 3 TB/s.  Per thread the terms are added in index order.
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) {
        float xv[4], gv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + 256 * u < i1 ? i + 256 * u : i1 - 1;
            xv[u] = x[base + j];
            gv[u] = MODE ? dy[base + j] : 0.f;
            yv[u] = (MODE && ry) ? ry[base + j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < i1) red_terms<MODE>(xv[u], gv[u], yv[u], ry != nullptr, m, is, t0, t1);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        t0 += __shfl_down(t0, d);
        t1 += __shfl_down(t1, d);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = t0; red[1][threadIdx.x >> 6] = t1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // one partial row per (image, chunk), two columns per channel: added in a fixed order by k_sums_finish -- no
        // atomics (round 2 added these with fp64 atomics: the only run-to-run irreproducible sums left on the path)
        double *row = partial + ((size_t)(plane / C) * gridDim.y + blockIdx.y) * 2 * C;
        row[c] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        row[C + c] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// [n, C] with C % 4 != 0: one block per channel, strided rows
template <int MODE>
__global__ __launch_bounds__(256) void k_colreduce_strided(const float *__restrict__ x, const float *__restrict__ dy,
                                                           const float *__restrict__ ry, int64_t n, int C,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           double *__restrict__ sums) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    double t0 = 0, t1 = 0;
    float m = 0.f, is = 1.f;
    if (MODE == 1) { m = mean[c]; is = invstd[c]; }
    for (int64_t i = threadIdx.x; i < n; i += 256)
        red_terms<MODE>(x[i * C + c], MODE ? dy[i * C + c] : 0.f, (MODE && ry) ? ry[i * C + c] : 1.f, ry != nullptr, m,
                        is, t0, t1);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        t0 += __shfl_down(t0, d);
        t1 += __shfl_down(t1, d);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = t0; red[1][threadIdx.x >> 6] = t1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[c] = red[0][0] + red[0][1] + red[0][2] + red[0][3];          // the only workgroup of this channel
        sums[C + c] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// Ticket words of the in-kernel finish (stats_tail.h): one block of 1 + STATS_MAX_GROUPS words per stream, zeroed when it
// is created and left zeroed by every launch that used it.  Launches that share a block are ordered by their stream.
static const size_t kTicketBytes = sizeof(unsigned) * (1 + STATS_MAX_GROUPS + 63) / 64 * 64;
static std::map<std::pair<int, hipStream_t>, unsigned *> g_ticket_blocks;   // (device, stream): the null stream's handle
static std::mutex g_ticket_lock;                                            // is the same on every device

unsigned *lidog_stats_tickets(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        lidog_set_error("stats tickets: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> guard(g_ticket_lock);
    const auto key = std::make_pair(dev, stream);
    auto it = g_ticket_blocks.find(key);
    if (it != g_ticket_blocks.end()) return it->second;
    unsigned *p = nullptr;
    // zeroed IN THE ORDER OF `stream`: hipMemset on device memory is asynchronous to the host and runs on the NULL stream,
    // which a non-blocking stream does not wait for -- with work queued on the NULL stream the first statistics kernel of a
    // new stream found garbage tickets, no workgroup saw itself as the last one and the sums were never written (round 5:
    // the first downsample branch on the side stream, 1 fresh process in 15; scripts/memset_order_probe.py,
    // tests/test_gpu_ops.py::test_first_statistics_launch_on_a_new_stream_while_the_default_stream_is_busy)
    if (hipMalloc((void **)&p, kTicketBytes) != hipSuccess || hipMemsetAsync(p, 0, kTicketBytes, stream) != hipSuccess) {
        lidog_set_error("stats tickets: cannot allocate %zu bytes of device memory", kTicketBytes);
        return nullptr;
    }
    g_ticket_blocks[key] = p;
    return p;
}

// A launch that ended in an error may have left arrivals in the ticket words of its stream (every later launch there would
// pick the wrong last workgroup and publish wrong sums): zero them again, in stream order.  Called by lidog_set_error's
// users through LIDOG_LAUNCH_CHECK failures of the statistics kernels (lidog_stats_tail_finish below).
int lidog_stats_tickets_reset(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    std::lock_guard<std::mutex> guard(g_ticket_lock);
    auto it = g_ticket_blocks.find(std::make_pair(dev, stream));
    if (it == g_ticket_blocks.end()) return 0;
    return hipMemsetAsync(it->second, 0, kTicketBytes, stream) == hipSuccess ? 0 : 1;
}

int lidog_stats_tail_make(StatsTail *tail, double *partial, double *sums, double count, int C, BnFinish fin,
                          hipStream_t st) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("LIDOG_STATS_TAIL");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    unsigned *tickets = nullptr;
    if (on) {
        tickets = lidog_stats_tickets(st);
        if (!tickets) return 1;
    }
    *tail = StatsTail{partial, tickets, sums, count, C, fin};
    return 0;
}

int lidog_stats_tail_finish(const StatsTail &tail, int nb, hipStream_t st) {
    if (tail.tickets) {
        // the kernel that carries the tail has just been launched: if that launch failed, some of its workgroups may never
        // arrive -- leave the stream's ticket words zero for whoever launches next
        if (hipPeekAtLastError() != hipSuccess) lidog_stats_tickets_reset(st);
        return 0;
    }
    return lidog_launch_sums_finish(tail.partial, nb, tail.C, tail.sums, tail.count, tail.fin, st);
}

static bool colreduce_uses_partials(int C, int64_t hw) { return hw == 1 && C % 4 == 0 && C / 4 <= 256; }
#define COLREDUCE_MAX_BLOCKS 512
#define COLREDUCE_PLANE_CHUNKS 64   // chunks of an NCHW plane (one workgroup each)
// The backward reduction over [rows, C] takes the grid of the per-row convolution reductions (one row per thread until
// 2048 workgroups are reached, then a grid-stride loop): the data-gradient reduction that produces dy can then
// accumulate the same partials in its epilogue, bit for bit (sconv.hip:k_sconv_reduce_rows4_bwdstats)
#define COLREDUCE_BWD_MAX_BLOCKS 2048

#include <stdlib.h>
// workgroups of the per-row statistics reductions (forward and backward); measured in the step: 1024 (two rows in flight
// per thread) 50.4 ms, 2048 50.8, 512 50.6; at bs 2 1536 / 2048 lose too (round 5)
extern "C" int32_t lidog_stats_max_blocks(void) { return 1024; }

extern "C" int64_t lidog_bn_bwd_reduce_blocks(int64_t n, int32_t C) {
    const int RB = 256 / (C / 4);
    int64_t nb = cdiv64(n, (int64_t)RB);
    const int64_t cap = lidog_stats_max_blocks();
    return nb > cap ? cap : (nb < 1 ? 1 : nb);
}

template <int MODE>
static int launch_colreduce(const float *x, const float *dy, const float *ry, int64_t n, int C, int64_t hw,
                            const float *mean, const float *invstd, double *sums, double *ws, double count,
                            BnFinish fin, hipStream_t st, const float *rw = nullptr, const float *rb = nullptr,
                            const uint32_t *rbits = nullptr) {
    if (n == 0) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * C + 1), st) == hipSuccess ? 0 : 1;
    if (colreduce_uses_partials(C, hw)) {
        LIDOG_REQUIRE(ws != nullptr, "bn reduce: workspace of lidog_bn_reduce_ws() doubles required");
        int C4 = C / 4, RB = 256 / C4;
        // 2 workgroups per CU, 4 rows in flight per lane: enough loads in flight to stream; small inputs get one
        // round of 4 rows per lane instead of four (the rounds of a lane are serial: latency, not bandwidth)
#ifndef COLREDUCE_ROUNDS
#define COLREDUCE_ROUNDS 1   // rounds of 4 rows per lane before the grid is capped (A/B switch; was 4)
#endif
        int64_t nb = cdiv64(n, (int64_t)RB * 4 * COLREDUCE_ROUNDS);
        if (nb > COLREDUCE_MAX_BLOCKS) nb = COLREDUCE_MAX_BLOCKS;
        if (MODE == 1) nb = lidog_bn_bwd_reduce_blocks(n, C);
        StatsTail tail;
        if (lidog_stats_tail_make(&tail, ws, sums, count, C, fin, st)) return 1;
        k_colreduce_nc4<MODE><<<(unsigned)nb, 256, 0, st>>>((const float4 *)x, (const float4 *)dy, (const float4 *)ry,
                                                            n, C4, mean, invstd, tail, rw, rb, rbits);
        lidog_stats_tail_finish(tail, (int)nb, st);
    } else {
        LIDOG_REQUIRE(rw == nullptr && rbits == nullptr,
                      "bn reduce: ReLU masks from x or from a bit mask only for [rows, C] with C %% 4 == 0");
        if (hw == 1) {
            // one workgroup per channel stores its result
            k_colreduce_strided<MODE><<<(unsigned)C, 256, 0, st>>>(x, dy, ry, n, C, mean, invstd, sums);
            if (count > 0 || fin.mean || fin.dw || fin.db)
                k_sums_post<<<(C + 127) / 128, 128, 0, st>>>(sums, C, count, fin);
        } else {
            // n = number of images B; planes = B*C; (image, chunk) partial rows summed in order by the finishing kernel
            LIDOG_REQUIRE(ws != nullptr, "bn reduce: NCHW input needs a workspace of n * lidog_bn_reduce_ws() doubles");
            int64_t planes = n * C;
            int chunks = (int)cdiv64(hw, 16384);
            if (chunks < 1) chunks = 1;
            if (chunks > COLREDUCE_PLANE_CHUNKS) chunks = COLREDUCE_PLANE_CHUNKS;
            k_colreduce_plane<MODE><<<dim3((unsigned)planes, (unsigned)chunks), 256, 0, st>>>(x, dy, ry, hw, C, mean,
                                                                                              invstd, ws);
            lidog_launch_sums_finish(ws, (int)(n * chunks), C, sums, count, fin, st);
        }
    }
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// doubles of workspace: [rows, C] with C % 4 == 0: the table of per-workgroup partials; NCHW (hw > 1): PER IMAGE
// (multiply by the number of images); 0 otherwise
extern "C" int64_t lidog_bn_reduce_ws(int32_t C, int64_t hw) {
    if (colreduce_uses_partials(C, hw)) return (int64_t)(COLREDUCE_BWD_MAX_BLOCKS + STATS_MAX_GROUPS) * 2 * C;
    return hw > 1 ? (int64_t)COLREDUCE_PLANE_CHUNKS * 2 * C : 0;
}

extern "C" int lidog_bn_stats(const float *x, int64_t n, int32_t C, int64_t hw, double *sums, double *ws, double count,
                              float eps, float momentum, float *mean, float *invstd, float *running_mean,
                              float *running_var, void *stream) {
    LIDOG_REQUIRE(mean == nullptr || count > 0, "bn_stats: finalising needs the row count");
    BnFinish fin = {eps, momentum, mean, invstd, running_mean, running_var, nullptr, nullptr};
    return launch_colreduce<0>(x, nullptr, nullptr, n, C, hw, nullptr, nullptr, sums, ws, count, fin,
                               (hipStream_t)stream);
}

extern "C" int lidog_bn_bwd_reduce(const float *dy, const float *x, const float *relu_y, int64_t n, int32_t C,
                                   int64_t hw, const float *mean, const float *invstd, double *sums, double *ws,
                                   double count, float *dw, float *db, const float *relu_w, const float *relu_b,
                                   void *stream) {
    return lidog_bn_bwd_reduce_bits(dy, x, relu_y, nullptr, n, C, hw, mean, invstd, sums, ws, count, dw, db, relu_w,
                                    relu_b, stream);
}

extern "C" int lidog_bn_bwd_reduce_bits(const float *dy, const float *x, const float *relu_y,
                                        const uint32_t *relu_bits, int64_t n, int32_t C, int64_t hw,
                                        const float *mean, const float *invstd, double *sums, double *ws, double count,
                                        float *dw, float *db, const float *relu_w, const float *relu_b, void *stream) {
    LIDOG_REQUIRE((relu_w == nullptr) == (relu_b == nullptr) && (relu_y != nullptr) + (relu_w != nullptr) +
                                                                        (relu_bits != nullptr) <= 1,
                  "bn_bwd_reduce: pass at most one of relu_y, relu_bits, (relu_w, relu_b)");
    BnFinish fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dw, db};
    return launch_colreduce<1>(x, dy, relu_y, n, C, hw, mean, invstd, sums, ws, count, fin, (hipStream_t)stream,
                               relu_w, relu_b, relu_bits);
}

__global__ void k_bn_finalize(const double *__restrict__ sums, double count, int C, float eps, float momentum,
                              float *__restrict__ mean, float *__restrict__ invstd, float *running_mean,
                              float *running_var) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!(count > 0)) count = sums[2 * C];  // SyncBatchNorm: the all-reduced row count rides behind the sums
    double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        double unb = (count > 1) ? var * count / (count - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

extern "C" int lidog_bn_finalize(const double *sums, double count, int32_t C, float eps, float momentum, float *mean,
                                 float *invstd, float *running_mean, float *running_var, void *stream) {
    k_bn_finalize<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(sums, count, C, eps, momentum, mean, invstd,
                                                                   running_mean, running_var);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ elementwise passes
__device__ __forceinline__ int chan_of(int64_t idx, int C, int64_t hw) {
    return (hw == 1) ? (int)(idx % C) : (int)((idx / hw) % C);
}

__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, int64_t total, int C, int64_t hw,
                                                  const float *__restrict__ mean, const float *__restrict__ invstd,
                                                  const float *__restrict__ w, const float *__restrict__ b,
                                                  const float *__restrict__ res, int relu, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int c = chan_of(i, C, hw);
        float v = (x[i] - mean[c]) * invstd[c] * w[c] + b[c];
        if (res) v += res[i];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_bn_apply4(const float4 *__restrict__ x, int64_t total4, int C4,
                                                   const float4 *__restrict__ mean, const float4 *__restrict__ invstd,
                                                   const float4 *__restrict__ w, const float4 *__restrict__ b,
                                                   const float4 *__restrict__ res, int relu, float4 *__restrict__ y,
                                                   uint32_t *__restrict__ bits) {
    // the loop is uniform per workgroup (the tail iteration masks lanes instead of leaving): the ReLU bit mask is
    // assembled from 8 neighbouring lanes with shuffles
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total4; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool ok = i < total4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            int c4 = (int)(i % C4);
            v = x[i];
            const float4 m = mean[c4], s = invstd[c4], ww = w[c4], bb = b[c4];
            v.x = (v.x - m.x) * s.x * ww.x + bb.x;
            v.y = (v.y - m.y) * s.y * ww.y + bb.y;
            v.z = (v.z - m.z) * s.z * ww.z + bb.z;
            v.w = (v.w - m.w) * s.w * ww.w + bb.w;
            if (res) { float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            y[i] = v;
        }
        if (bits) {   // 4 bits per float4 (element > 0), 8 float4 per 32-bit word: bit 4 * (i % 8) + component
            uint32_t nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            uint32_t wv = nib << (4 * (threadIdx.x & 7));
            wv |= __shfl_xor(wv, 1);
            wv |= __shfl_xor(wv, 2);
            wv |= __shfl_xor(wv, 4);
            if (ok && (threadIdx.x & 7) == 0) bits[i >> 3] = wv;
        }
    }
}

// The same pass for SyncBatchNorm: mean / invstd are derived from the ALL-REDUCED (sum x, sum x^2, rows) vector inside
// the launch -- k_bn_finalize's arithmetic, by the first C threads of every workgroup into LDS -- instead of by a
// launch of their own between the all-reduce and this pass (62 per step on the dependent chain of a data-parallel
// rank).  Workgroup 0 also stores mean / invstd (backward reads them) and updates the running statistics.
__global__ __launch_bounds__(256) void k_bn_apply4_sync(const float4 *__restrict__ x, int64_t total4, int C4,
                                                        const double *__restrict__ sums, float eps, float momentum,
                                                        float *__restrict__ mean_out, float *__restrict__ invstd_out,
                                                        float *running_mean, float *running_var,
                                                        const float4 *__restrict__ w, const float4 *__restrict__ b,
                                                        const float4 *__restrict__ res, int relu,
                                                        float4 *__restrict__ y, uint32_t *__restrict__ bits) {
    extern __shared__ float s_ms[];   // [C] mean, [C] invstd
    const int C = C4 * 4;
    const double count = sums[2 * C];
    for (int c = threadIdx.x; c < C; c += 256) {
        const double m = sums[c] / count;
        double var = sums[C + c] / count - m * m;
        if (var < 0) var = 0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
        s_ms[c] = mf;
        s_ms[C + c] = isf;
        if (blockIdx.x == 0) {
            mean_out[c] = mf;
            invstd_out[c] = isf;
            if (running_mean) {
                const double unb = (count > 1) ? var * count / (count - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const float4 *mean4 = reinterpret_cast<const float4 *>(s_ms), *is4 = reinterpret_cast<const float4 *>(s_ms + C);
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total4; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool ok = i < total4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            int c4 = (int)(i % C4);
            v = x[i];
            const float4 m = mean4[c4], s = is4[c4], ww = w[c4], bb = b[c4];
            v.x = (v.x - m.x) * s.x * ww.x + bb.x;
            v.y = (v.y - m.y) * s.y * ww.y + bb.y;
            v.z = (v.z - m.z) * s.z * ww.z + bb.z;
            v.w = (v.w - m.w) * s.w * ww.w + bb.w;
            if (res) { float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            y[i] = v;
        }
        if (bits) {
            uint32_t nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            uint32_t wv = nib << (4 * (threadIdx.x & 7));
            wv |= __shfl_xor(wv, 1);
            wv |= __shfl_xor(wv, 2);
            wv |= __shfl_xor(wv, 4);
            if (ok && (threadIdx.x & 7) == 0) bits[i >> 3] = wv;
        }
    }
}

static unsigned ew_grid(int64_t n) {
    int64_t g = cdiv64(n, 256);
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// NCHW images (hw > 1): workgroup (plane, chunk) -- the channel and its constants are fixed per workgroup, no
// per-element index arithmetic (the generic kernels spend a 64-bit division per element on it)
__global__ __launch_bounds__(256) void k_bn_apply_plane(const float *__restrict__ x, int64_t hw, int C,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ invstd,
                                                        const float *__restrict__ w, const float *__restrict__ b,
                                                        const float *__restrict__ res, int relu,
                                                        float *__restrict__ y) {
    const int64_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float m = mean[c], sc = invstd[c] * w[c], sh = b[c], is = invstd[c], ww = w[c];
    (void)sc;
    const int64_t chunk = (hw + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = plane * hw + (int64_t)blockIdx.y * chunk;
    const int64_t i1 = (blockIdx.y + 1) * chunk < hw ? i0 + chunk : plane * hw + hw;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) {   // four independent elements per thread and round
        float xv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + 256 * u < i1 ? i + 256 * u : i1 - 1;
            xv[u] = x[j];
            rv[u] = res ? res[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = (xv[u] - m) * is * ww + sh;   // same operation order as k_bn_apply
            if (res) v += rv[u];
            if (relu) v = v > 0.f ? v : 0.f;
            if (i + 256 * u < i1) y[i + 256 * u] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply_plane(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ ry, int64_t hw, int C,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ w,
                                                            const double *__restrict__ sums, double inv_count,
                                                            float *__restrict__ dx, float *__restrict__ dres) {
    const int64_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const double ic = (inv_count > 0) ? inv_count : 1.0 / sums[2 * C];
    const float m = mean[c], is = invstd[c], ww = w[c];
    const float m0 = (float)(sums[c] * ic), m1 = (float)(sums[C + c] * ic);
    const int64_t chunk = (hw + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = plane * hw + (int64_t)blockIdx.y * chunk;
    const int64_t i1 = (blockIdx.y + 1) * chunk < hw ? i0 + chunk : plane * hw + hw;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) {   // four independent elements per thread and round
        float gv[4], xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + 256 * u < i1 ? i + 256 * u : i1 - 1;
            gv[u] = dy[j];
            xv[u] = x[j];
            yv[u] = ry ? ry[j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float g = gv[u];
            if (ry && !(yv[u] > 0.f)) g = 0.f;
            const float xh = (xv[u] - m) * is;
            if (i + 256 * u < i1) {
                dx[i + 256 * u] = (g - m0 - xh * m1) * is * ww;   // same operation order as k_bn_bwd_apply
                if (dres) dres[i + 256 * u] = g;
            }
        }
    }
}

extern "C" int64_t lidog_relu_bits_words(int64_t n, int32_t C) { return (n * (C / 4) + 7) / 8; }

extern "C" int lidog_bn_apply(const float *x, int64_t n, int32_t C, int64_t hw, const float *mean,
                              const float *invstd, const float *w, const float *b, const float *residual,
                              int32_t relu, float *y, void *stream) {
    return lidog_bn_apply_bits(x, n, C, hw, mean, invstd, w, b, residual, relu, y, nullptr, stream);
}

extern "C" int lidog_bn_apply_bits(const float *x, int64_t n, int32_t C, int64_t hw, const float *mean,
                                   const float *invstd, const float *w, const float *b, const float *residual,
                                   int32_t relu, float *y, uint32_t *relu_bits, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int64_t total = n * C * hw;
    LIDOG_REQUIRE(relu_bits == nullptr || (hw == 1 && C % 4 == 0), "bn_apply: bit masks only for [rows, C], C %% 4 == 0");
    if (total == 0) return 0;
    if (hw == 1 && C % 4 == 0) {
        k_bn_apply4<<<ew_grid(total / 4), 256, 0, st>>>((const float4 *)x, total / 4, C / 4, (const float4 *)mean,
                                                        (const float4 *)invstd, (const float4 *)w, (const float4 *)b,
                                                        (const float4 *)residual, relu, (float4 *)y, relu_bits);
    } else if (hw >= 1024) {
        int chunks = (int)cdiv64(hw, 8192);
        if (chunks > 64) chunks = 64;
        k_bn_apply_plane<<<dim3((unsigned)(n * C), (unsigned)chunks), 256, 0, st>>>(x, hw, C, mean, invstd, w, b,
                                                                                   residual, relu, y);
    } else {
        k_bn_apply<<<ew_grid(total), 256, 0, st>>>(x, total, C, hw, mean, invstd, w, b, residual, relu, y);
    }
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// SyncBatchNorm forward, [rows, C] with C % 4 == 0: finalise (mean, invstd, running statistics) from the all-reduced
// sums AND apply (+ residual + ReLU + bit mask) in one launch; the same results as lidog_bn_finalize(sums, -1, ...)
// followed by lidog_bn_apply_bits.
extern "C" int lidog_bn_apply_sync(const float *x, int64_t n, int32_t C, const double *sums, float eps, float momentum,
                                   float *mean, float *invstd, float *running_mean, float *running_var, const float *w,
                                   const float *b, const float *residual, int32_t relu, float *y, uint32_t *relu_bits,
                                   void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(C % 4 == 0 && C >= 4 && C <= 4096 && sums && mean && invstd && w && b,
                  "bn_apply_sync: [rows, C] with C a multiple of 4, <= 4096; sums / mean / invstd / w / b required");
    LIDOG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_apply_sync: both running statistics or none");
    const int64_t total4 = n * C / 4;
    // an empty shard still has to store mean / invstd and move the running statistics (one workgroup does)
    const unsigned grid = total4 ? ew_grid(total4) : 1;
    k_bn_apply4_sync<<<grid, 256, (size_t)2 * C * sizeof(float), st>>>(
        (const float4 *)x, total4, C / 4, sums, eps, momentum, mean, invstd, running_mean, running_var, (const float4 *)w,
        (const float4 *)b, (const float4 *)residual, relu, (float4 *)y, relu_bits);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dy, const float *__restrict__ x,
                                                      const float *__restrict__ ry, int64_t total, int C, int64_t hw,
                                                      const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, const float *__restrict__ w,
                                                      const double *__restrict__ sums, double inv_count,
                                                      float *__restrict__ dx, float *__restrict__ dres) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int c = chan_of(i, C, hw);
        float g = dy[i];
        if (ry && !(ry[i] > 0.f)) g = 0.f;
        float is = invstd[c];
        float xh = (x[i] - mean[c]) * is;
        const double ic = (inv_count > 0) ? inv_count : 1.0 / sums[2 * C];
        float m0 = (float)(sums[c] * ic), m1 = (float)(sums[C + c] * ic);
        dx[i] = (g - m0 - xh * m1) * is * w[c];
        if (dres) dres[i] = g;
    }
}

// [rows, C] layout, C % 4 == 0: 16-B accesses; the per-channel constants (m0, m1, invstd*w) are computed once
// per thread, outside the row loop (each thread keeps its channel quad: the grid stride is a multiple of C4)
__global__ __launch_bounds__(256) void k_bn_bwd_apply4(const float4 *__restrict__ dy, const float4 *__restrict__ x,
                                                       const float4 *__restrict__ ry, int64_t total4, int C4,
                                                       const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ w,
                                                       const double *__restrict__ sums, double inv_count,
                                                       float4 *__restrict__ dx, float4 *__restrict__ dres,
                                                       const float *__restrict__ rb,
                                                       const uint32_t *__restrict__ rbits) {
    const int C = C4 * 4;
    const int64_t stride = (int64_t)gridDim.x * 256;  // launcher makes this a multiple of C4
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    const double ic = (inv_count > 0) ? inv_count : 1.0 / sums[2 * C];
    float mu[4], is[4], sc[4], m0[4], m1[4], gw[4], gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = c4 * 4 + j;
        mu[j] = mean[c];
        is[j] = invstd[c];
        gw[j] = w[c];
        gb[j] = rb ? rb[c] : 0.f;
        sc[j] = is[j] * w[c];
        m0[j] = (float)(sums[c] * ic);
        m1[j] = (float)(sums[C + c] * ic);
    }
    auto one = [&](int64_t at, float4 g, const float4 xv, const float4 y) {
        if (ry || rbits) {
            g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f;
            g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
        } else if (rb) {   // ReLU mask from the forward pass's pre-activation, recomputed bit for bit from x
            g.x = ((xv.x - mu[0]) * is[0] * gw[0] + gb[0]) > 0.f ? g.x : 0.f;
            g.y = ((xv.y - mu[1]) * is[1] * gw[1] + gb[1]) > 0.f ? g.y : 0.f;
            g.z = ((xv.z - mu[2]) * is[2] * gw[2] + gb[2]) > 0.f ? g.z : 0.f;
            g.w = ((xv.w - mu[3]) * is[3] * gw[3] + gb[3]) > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = (g.x - m0[0] - (xv.x - mu[0]) * is[0] * m1[0]) * sc[0];
        o.y = (g.y - m0[1] - (xv.y - mu[1]) * is[1] * m1[1]) * sc[1];
        o.z = (g.z - m0[2] - (xv.z - mu[2]) * is[2] * m1[2]) * sc[2];
        o.w = (g.w - m0[3] - (xv.w - mu[3]) * is[3] * m1[3]) * sc[3];
        dx[at] = o;
        if (dres) dres[at] = g;
    };
    const float4 none = make_float4(1.f, 1.f, 1.f, 1.f);
    // (Round 5: four grid strides of a thread in flight -- every load of a round issued before the first use -- make this
    // pass faster next to the weight gradients it co-runs with and the STEP slower: 48.41 / 48.52 -> 48.76 / 48.81 ms, same
    // box, alternating; the bandwidth it gains is taken from the second stream's gathers.  One stride in flight stays.)
    for (; i < total4; i += stride)
        one(i, dy[i], x[i], rbits ? lidog_relu_bits_as_float4(rbits, i) : (ry ? ry[i] : none));
}

__global__ void k_bn_param_grads(const double *__restrict__ sums, int C, float *dw, float *db) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (db) db[c] = (float)sums[c];
    if (dw) dw[c] = (float)sums[C + c];
}

extern "C" int lidog_bn_bwd_apply(const float *dy, const float *x, const float *relu_y, int64_t n, int32_t C,
                                  int64_t hw, const float *mean, const float *invstd, const float *w,
                                  const double *sums, double count, float *dx, float *dres, float *dw, float *db,
                                  const float *relu_b, void *stream) {
    return lidog_bn_bwd_apply_bits(dy, x, relu_y, nullptr, n, C, hw, mean, invstd, w, sums, count, dx, dres, dw, db,
                                   relu_b, stream);
}

extern "C" int lidog_bn_bwd_apply_bits(const float *dy, const float *x, const float *relu_y,
                                       const uint32_t *relu_bits, int64_t n, int32_t C, int64_t hw, const float *mean,
                                       const float *invstd, const float *w, const double *sums, double count,
                                       float *dx, float *dres, float *dw, float *db, const float *relu_b,
                                       void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int64_t total = n * C * hw;
    LIDOG_REQUIRE(relu_b == nullptr || (relu_y == nullptr && hw == 1 && C % 4 == 0),
                  "bn_bwd_apply: the ReLU mask can be recomputed from x only for [rows, C] with C %% 4 == 0");
    LIDOG_REQUIRE(relu_bits == nullptr || (relu_y == nullptr && relu_b == nullptr && hw == 1 && C % 4 == 0),
                  "bn_bwd_apply: a ReLU bit mask replaces relu_y / relu_b, [rows, C] with C %% 4 == 0 only");
    if (total > 0 && hw == 1 && C % 4 == 0) {
        int C4 = C / 4;
        int64_t total4 = total / 4;
        // grid stride = blocks * 256 must be a multiple of C4: blocks a multiple of C4 / gcd(C4, 256)
        int g = C4;
        for (int a = 256, b = C4; b;) { int t = a % b; a = b; b = t; g = a; }
        int unit = C4 / g;
        int64_t blocks = cdiv64(total4, 256 * 4);
        if (blocks > 4096) blocks = 4096;
        blocks = cdiv64(blocks, unit) * unit;
        k_bn_bwd_apply4<<<(unsigned)blocks, 256, 0, st>>>((const float4 *)dy, (const float4 *)x,
                                                          (const float4 *)relu_y, total4, C4, mean, invstd, w, sums,
                                                          count > 0 ? 1.0 / count : -1.0, (float4 *)dx,
                                                          (float4 *)dres, relu_b, relu_bits);
    } else if (total > 0 && hw >= 1024) {
        int chunks = (int)cdiv64(hw, 8192);
        if (chunks > 64) chunks = 64;
        k_bn_bwd_apply_plane<<<dim3((unsigned)(n * C), (unsigned)chunks), 256, 0, st>>>(
            dy, x, relu_y, hw, C, mean, invstd, w, sums, count > 0 ? 1.0 / count : -1.0, dx, dres);
    } else if (total > 0)
        k_bn_bwd_apply<<<ew_grid(total), 256, 0, st>>>(dy, x, relu_y, total, C, hw, mean, invstd, w, sums,
                                                       count > 0 ? 1.0 / count : -1.0,
                                                       dx, dres);
    if (dw || db) k_bn_param_grads<<<(C + 127) / 128, 128, 0, st>>>(sums, C, dw, db);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// column sums of a narrow [n, C] matrix (C <= 16): the bias gradient of the classifier (96 -> 7).  Every workgroup
// sums a contiguous slab of rows in double, the slabs are added in order by the second kernel (no atomics).
#define COLSUM_BLOCKS 512
__global__ __launch_bounds__(256) void k_colsum_narrow(const float *__restrict__ x, int64_t n, int C,
                                                       double *__restrict__ partial) {
    __shared__ double red[4][16];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const float *row = x + r * C;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < C) acc[j] += (double)row[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double v = acc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < C)
        partial[(size_t)blockIdx.x * C + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// 16 threads per column, each a strided chain over the slabs (b = part, part + 16, ...), the 16 chains added in order: the
// sums do not depend on timing.  (One thread per column walked 512 slabs one load at a time: 144 us on the launch stream at
// the start of every backward pass, for 28 KB.)
__global__ __launch_bounds__(256) void k_colsum_finish(const double *__restrict__ partial, int nb, int C,
                                                       float *__restrict__ out) {
    __shared__ double red[16][16];
    const int c = threadIdx.x & 15, part = threadIdx.x >> 4;
    double s = 0;
    if (c < C)
        for (int b = part; b < nb; b += 16) s += partial[(size_t)b * C + c];
    red[part][c] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        double t = red[0][c];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][c];
        out[c] = (float)t;
    }
}

// doubles of workspace lidog_colsum needs for C columns
extern "C" int64_t lidog_colsum_ws(int32_t C) {
    if (C <= 16) return (int64_t)COLSUM_BLOCKS * C;
    return (int64_t)(2 * C + 2) + lidog_bn_reduce_ws(C, 1);
}

extern "C" int lidog_colsum(const float *x, int64_t n, int32_t C, float *out, double *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(C >= 1, "colsum: C >= 1");
    LIDOG_REQUIRE(ws != nullptr, "colsum: needs a workspace of lidog_colsum_ws(C) doubles");
    if (n == 0) return hipMemsetAsync(out, 0, sizeof(float) * C, st) == hipSuccess ? 0 : 1;
    if (C > 16) {
        // wide case (a bias gradient of a wide convolution; `final` of the LiDOG path has 7 columns): the BatchNorm
        // statistics reduction, whose last kernel stores the float copy of the first sum
        BnFinish fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, out};
        return launch_colreduce<0>(x, nullptr, nullptr, n, C, 1, nullptr, nullptr, ws, ws + 2 * C + 2, 0.0, fin, st);
    }
    int nb = (int)(cdiv64(n, 1024) < COLSUM_BLOCKS ? cdiv64(n, 1024) : COLSUM_BLOCKS);
    k_colsum_narrow<<<nb, 256, 0, st>>>(x, n, C, ws);
    k_colsum_finish<<<1, 256, 0, st>>>(ws, nb, C, out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// invstd of an evaluation-mode BatchNorm (running statistics): 1 / sqrt(running_var + eps)
__global__ void k_bn_eval_invstd(const float *__restrict__ var, float eps, int C, float *__restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = (float)(1.0 / sqrt((double)var[c] + (double)eps));
}

extern "C" int lidog_bn_eval_invstd(const float *running_var, float eps, int32_t C, float *invstd, void *stream) {
    if (C == 0) return 0;
    k_bn_eval_invstd<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(running_var, eps, C, invstd);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ME.cat of two feature matrices on one coordinate map: out[i] = (a[i], b[i]); the backward pass splits the gradient
// back into two contiguous matrices (torch.cat's backward hands out strided views, which every consumer then copies)
__global__ __launch_bounds__(256) void k_cat2(const float4 *__restrict__ a, int Ca4, const float4 *__restrict__ b,
                                              int Cb4, int64_t n, float4 *__restrict__ out) {
    const int C4 = Ca4 + Cb4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C4) return;
    const int64_t i = idx / C4;
    const int c = (int)(idx - i * C4);
    out[idx] = c < Ca4 ? a[i * Ca4 + c] : b[i * Cb4 + (c - Ca4)];
}

__global__ __launch_bounds__(256) void k_split2(const float4 *__restrict__ g, int Ca4, int Cb4, int64_t n,
                                                float4 *__restrict__ ga, float4 *__restrict__ gb) {
    const int C4 = Ca4 + Cb4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C4) return;
    const int64_t i = idx / C4;
    const int c = (int)(idx - i * C4);
    const float4 v = g[idx];
    if (c < Ca4) ga[i * Ca4 + c] = v;
    else gb[i * Cb4 + (c - Ca4)] = v;
}

extern "C" int lidog_cat2(const float *a, int32_t Ca, const float *b, int32_t Cb, int64_t n, float *out, void *stream) {
    LIDOG_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "cat2: channel counts must be positive multiples of 4");
    if (n == 0) return 0;
    k_cat2<<<(unsigned)cdiv64(n * ((Ca + Cb) / 4), 256), 256, 0, (hipStream_t)stream>>>(
        (const float4 *)a, Ca / 4, (const float4 *)b, Cb / 4, n, (float4 *)out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_split2(const float *g, int32_t Ca, int32_t Cb, int64_t n, float *ga, float *gb, void *stream) {
    LIDOG_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "split2: channel counts must be positive multiples of 4");
    if (n == 0) return 0;
    k_split2<<<(unsigned)cdiv64(n * ((Ca + Cb) / 4), 256), 256, 0, (hipStream_t)stream>>>(
        (const float4 *)g, Ca / 4, Cb / 4, n, (float4 *)ga, (float4 *)gb);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void k_relu_fwd(const float *__restrict__ x, int64_t n, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void k_relu_bwd(const float *__restrict__ dy, const float *__restrict__ y,
                                                  int64_t n, float *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = (y[i] > 0.f) ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void k_add(const float *__restrict__ a, const float *__restrict__ b, int64_t n,
                                             float *__restrict__ o) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = a[i] + b[i];
}

extern "C" int lidog_relu_fwd(const float *x, int64_t n, float *y, void *stream) {
    if (n) k_relu_fwd<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(x, n, y);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
extern "C" int lidog_relu_bwd(const float *dy, const float *y, int64_t n, float *dx, void *stream) {
    if (n) k_relu_bwd<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(dy, y, n, dx);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
extern "C" int lidog_add(const float *a, const float *b, int64_t n, float *out, void *stream) {
    if (n) k_add<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(a, b, n, out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
