// In-kernel finish of a per-channel (sum, sum) reduction: the LAST workgroup to arrive adds the per-workgroup partial
// rows in a fixed order and does what bn.hip:k_sums_finish used to do in a launch of its own (write the sums and the
// row count, finalise mean / invstd / running statistics, emit the parameter gradients).  124 microsecond-sized
// launches per training step sat on the dependent chain between a reduction and the BatchNorm kernel that consumes it
// (MinkowskiBatchNorm / BatchNorm1d of every convolution, utils/models/minkunet_bev.py:60,406-408, forward and backward).
//
// Two levels, so that no workgroup reads more than STATS_GROUP rows: workgroups [g*32, g*32+32) form group g; the last
// of a group to arrive adds the group's rows in ascending workgroup order into group row g; the last group finisher adds
// the group rows in ascending group order.  Which workgroup does the adding depends on timing, WHAT is added in which
// order does not: the sums are bit-reproducible, and every kernel that ends in this tail (forward statistics, backward
// statistics fused or stand-alone) produces the same bits from the same partial rows.
//
// Inter-workgroup visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility":
// per-XCD L2s are not coherent, a CU's L1 is never refreshed): every handed-off double is stored write-through (sc1)
// and loaded sc1; every storing wave drains its stores (vmcnt 0) before the workgroup's barrier; ONE lane then adds to
// an agent-scope ticket, and the workgroup whose add returned the last ticket reads -- the guide's counter form.
// Ticket words are zero between launches: the final workgroup resets them (the buffer is zeroed when it is created).
#pragma once
#include "common.h"

#define STATS_GROUP 32
#define STATS_MAX_GROUPS 128  // rows of workspace behind the partial rows; 128 * 32 = 4096 partial rows at most

typedef __attribute__((address_space(1))) unsigned long long lidog_gu64;
typedef __attribute__((address_space(1))) unsigned int lidog_gu32;

__device__ __forceinline__ void lidog_store_sc1(double *p, double v) {
    __hip_atomic_store((lidog_gu64 *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lidog_load_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((lidog_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct StatsTail {
    double *partial;     // [gridDim.x + groups][2C]: a row per workgroup, the group rows behind them
    unsigned *tickets;   // [1 + groups]: word 0 counts finished groups, word 1 + g the arrivals of group g
    double *sums;        // [2C + 1] result (+ the row count when count > 0)
    double count;
    int C;
    BnFinish fin;
};

__device__ __forceinline__ void lidog_bn_finalize_channel(double sx, double sxx, double count, int c,
                                                          const BnFinish &fin) {
    double m = sx / count;
    double var = sxx / count - m * m;
    if (var < 0) var = 0;
    fin.mean[c] = (float)m;
    fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
    if (fin.running_mean) {
        double unb = (count > 1) ? var * count / (count - 1) : var;
        fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
        fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unb;
    }
}

// sum of rows [r0, r0 + n) of column `col`, ascending, eight loads in flight
__device__ __forceinline__ double lidog_rows_sum_sc1(const double *base, int r0, int n, int C2, int col) {
    double s = 0.0;
    const double *p = base + (size_t)r0 * C2 + col;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lidog_load_sc1(p + (size_t)(j + u) * C2);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < n; ++j) s += lidog_load_sc1(p + (size_t)j * C2);
    return s;
}

// Second half of the tail, for a workgroup whose slice of partial row `b` (of `nb` rows) has been stored with
// lidog_store_sc1 by its threads: `per_row` workgroups contribute to a row (column tiles of a 2-D grid; 1 otherwise).
// Called by EVERY thread of EVERY workgroup of the launch.
__device__ __forceinline__ void lidog_stats_tail_rows(const StatsTail &t, int b, int nb, int per_row) {
    __shared__ int s_last;
    const int C = t.C, C2 = 2 * C;
    const int ng = (nb + STATS_GROUP - 1) / STATS_GROUP;
    const int g = b / STATS_GROUP;
    const int g0 = g * STATS_GROUP;
    const int gsize = (nb - g0 < STATS_GROUP) ? nb - g0 : STATS_GROUP;
    if (!t.tickets) return;   // A/B switch (LIDOG_STATS_TAIL=0): bn.hip:k_sums_finish adds the rows in a launch of its own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the barrier
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 1;
        if (gsize * per_row > 1) {
            unsigned old = __hip_atomic_fetch_add((lidog_gu32 *)(t.tickets + 1 + g), 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(gsize * per_row - 1);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last workgroup of group g: the group's rows, ascending
    double *grow = t.partial + (size_t)(nb + g) * C2;
    for (int col = threadIdx.x; col < C2; col += blockDim.x)
        lidog_store_sc1(grow + col, lidog_rows_sum_sc1(t.partial, g0, gsize, C2, col));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 1;
        if (ng > 1) {
            unsigned old = __hip_atomic_fetch_add((lidog_gu32 *)t.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(ng - 1);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last group finisher: the group rows, ascending; then what follows the reduction
    const double *grows = t.partial + (size_t)nb * C2;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double s0 = lidog_rows_sum_sc1(grows, 0, ng, C2, c);
        const double s1 = lidog_rows_sum_sc1(grows, 0, ng, C2, C + c);
        t.sums[c] = s0;
        t.sums[C + c] = s1;
        if (t.fin.db) t.fin.db[c] = (float)s0;
        if (t.fin.dw) t.fin.dw[c] = (float)s1;
        if (t.fin.mean) lidog_bn_finalize_channel(s0, s1, t.count, c, t.fin);
    }
    if (threadIdx.x == 0 && t.count > 0) t.sums[C2] = t.count;
    // ticket words back to zero for the next launch on this stream (every add of this launch has returned by now)
    for (int i = threadIdx.x; i < 1 + ng; i += blockDim.x)
        __hip_atomic_store((lidog_gu32 *)(t.tickets + i), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by EVERY thread of EVERY workgroup (256 threads, 1-D grid) at the end of the kernel.  `writer`: this thread
// holds the workgroup's totals of channels 4 c4 .. 4 c4 + 3 (a[0..3] first sums, a[4..7] second sums).
__device__ __forceinline__ void lidog_stats_tail(const StatsTail &t, bool writer, int c4, const double (&a)[8]) {
    const int C = t.C, C2 = 2 * C;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    if (writer) {
        double *dst = t.partial + (size_t)b * C2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lidog_store_sc1(dst + c4 * 4 + j, a[j]);
            lidog_store_sc1(dst + C + c4 * 4 + j, a[4 + j]);
        }
    }
    lidog_stats_tail_rows(t, b, nb, 1);
}

// ticket words of the launches queued on `stream` (zeroed when created; launches that share them must be ordered, which
// launches on one stream are); NULL + error set on failure
unsigned *lidog_stats_tickets(hipStream_t stream);
int lidog_stats_tickets_reset(hipStream_t stream);   // zero them again (after a failed launch), in stream order

// Fills `tail` for a launch on `st` and returns 0; with LIDOG_STATS_TAIL=0 (same-box A/B runs) tail.tickets stays NULL
// and the caller launches bn.hip:k_sums_finish behind its kernel (lidog_stats_tail_finish does, when needed).
int lidog_stats_tail_make(StatsTail *tail, double *partial, double *sums, double count, int C, BnFinish fin,
                          hipStream_t st);
int lidog_stats_tail_finish(const StatsTail &tail, int nb, hipStream_t st);
