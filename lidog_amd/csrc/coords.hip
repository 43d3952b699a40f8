// Coordinate maps on the GPU: packed-key open-addressing hash table, strided maps with
// first-occurrence row order, neighbour tables and ballot/prefix-sum rule-book compaction.
// Semantics: SURVEY.md 8(b); restated on the CPU in oracle/me_oracle.c (orc_unique_first,
// orc_stride, orc_kernel_map, orc_pairs_from_nbr).
#include <limits.h>
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";
void lidog_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *lidog_last_error(void) { return g_err; }
// bumped whenever an entry point changes its arguments or their meaning (lidog_amd/_lib.py checks it at load time);
// 4 = round 4: in-kernel statistics finish (workspace sizes), peer all-reduce stream binding / fault hooks
extern "C" int lidog_abi_version(void) { return 8; }

extern "C" int64_t lidog_hash_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

// ------------------------------------------------------------------ block-level scans
__device__ __forceinline__ int block_excl_scan_256(int v, int *total) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();  // protect wsum reuse across calls
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < w) off += wsum[i];
        tot += wsum[i];
    }
    *total = tot;
    return off + x - v;
}

// exclusive scan of int32 data[n] in place; sums[nb+1] scratch (nb = ceil(n/1024)); sums[nb] = total
__global__ __launch_bounds__(256) void scan_block_sums(const int32_t *__restrict__ data, int64_t n,
                                                       int32_t *__restrict__ sums) {
    int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) v += data[base + j];
    int tot;
    block_excl_scan_256(v, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scan_block_offsets(int32_t *__restrict__ sums, int64_t nb) {
    // single block; chunked with carry
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t c = 0; c < nb; c += 256) {
        int64_t i = c + threadIdx.x;
        int v = (i < nb) ? sums[i] : 0;
        int tot;
        int ex = block_excl_scan_256(v, &tot);
        int carry = carry_s;
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry_s;
}

__global__ __launch_bounds__(256) void scan_apply(int32_t *__restrict__ data, int64_t n,
                                                  const int32_t *__restrict__ sums) {
    int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (base + j < n) ? data[base + j] : 0;
        s += v[j];
    }
    int tot;
    int ex = block_excl_scan_256(s, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) data[base + j] = ex;
        ex += v[j];
    }
}

static int device_exclusive_scan(int32_t *data, int64_t n, int32_t *sums, hipStream_t st) {
    int64_t nb = cdiv64(n, 1024);
    if (nb == 0) return 0;
    scan_block_sums<<<dim3((unsigned)nb), 256, 0, st>>>(data, n, sums);
    scan_block_offsets<<<1, 256, 0, st>>>(sums, nb);
    scan_apply<<<dim3((unsigned)nb), 256, 0, st>>>(data, n, sums);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ insert
__device__ __forceinline__ int32_t table_insert_min(uint64_t *keys, int32_t *vals, uint64_t mask, uint64_t key,
                                                    int32_t row) {
    uint64_t slot = lidog_mix(key) & mask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)LIDOG_EMPTY_KEY,
                                            (unsigned long long)key);
        if (prev == LIDOG_EMPTY_KEY || prev == key) {
            atomicMin(&vals[slot], row);
            return (int32_t)slot;
        }
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void k_insert(const int4 *__restrict__ coords, int64_t n, int32_t stride,
                                                uint64_t *keys, int32_t *vals, uint64_t mask,
                                                int32_t *__restrict__ slot_of, int32_t *err_flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    if (stride > 1) {  // floor toward -inf; stride is a power-of-two multiple in practice but stay general
        auto fl = [stride](int v) {
            int q = v / stride;
            if ((v % stride != 0) && (v < 0)) --q;
            return q * stride;
        };
        c.y = fl(c.y); c.z = fl(c.z); c.w = fl(c.w);
    }
    int bad = 0;
    uint64_t key = lidog_pack(c.x, c.y, c.z, c.w, &bad);
    if (bad) { *err_flag = 1; slot_of[i] = -1; return; }
    slot_of[i] = table_insert_min(keys, vals, mask, key, (int32_t)i);
}

__global__ __launch_bounds__(256) void k_first_row(int32_t *__restrict__ slot_then_first, int64_t n,
                                                   const int32_t *__restrict__ vals,
                                                   unsigned long long *n_unique) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int uniq = 0;
    if (i < n) {
        int s = slot_then_first[i];
        int f = (s >= 0) ? vals[s] : (int)i;
        slot_then_first[i] = f;
        uniq = (f == (int)i);
    }
    unsigned long long m = __ballot(uniq);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_unique, (unsigned long long)__popcll(m));
}

// ---- lidog_coords_insert + what the caller otherwise computes with a handful of device-wide reductions of its own
// (torch amin / amax / max, 0.8 ms per step on the map stream): info [9] int64 = (unique rows, error flag, largest
// batch index, lowest x, y, z, highest x, y, z), ONE read-back for the whole insert.
__global__ void k_info_init(long long *info) {
    const int i = threadIdx.x;
    if (i < 9) info[i] = (i < 2) ? 0 : (i == 2 || i >= 6) ? -(1ll << 40) : (1ll << 40);
}

__global__ __launch_bounds__(256) void k_insert_info(const int4 *__restrict__ coords, int64_t n, uint64_t *keys,
                                                     int32_t *vals, uint64_t mask, int32_t *__restrict__ slot_of,
                                                     int32_t *err_flag, long long *__restrict__ info) {
    __shared__ int s_lo[4][3], s_hi[4][4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
    if (i < n) {
        int4 c = coords[i];
        int bad = 0;
        uint64_t key = lidog_pack(c.x, c.y, c.z, c.w, &bad);
        if (bad) {
            *err_flag = 1;
            info[1] = 1;
            slot_of[i] = -1;
        } else {
            slot_of[i] = table_insert_min(keys, vals, mask, key, (int32_t)i);
        }
        lo[0] = hi[0] = c.y; lo[1] = hi[1] = c.z; lo[2] = hi[2] = c.w; hi[3] = c.x;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) lo[j] = min(lo[j], __shfl_xor(lo[j], d));
#pragma unroll
        for (int j = 0; j < 4; ++j) hi[j] = max(hi[j], __shfl_xor(hi[j], d));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_lo[w][j] = lo[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) s_hi[w][j] = hi[j];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int j = threadIdx.x;
        if (j < 3) {
            int v = min(min(s_lo[0][j], s_lo[1][j]), min(s_lo[2][j], s_lo[3][j]));
            if (v != INT_MAX) atomicMin(&info[3 + j], (long long)v);
        } else {
            const int k = j - 3;   // 0..2 = hi x, y, z; 3 = batch
            int v = max(max(s_hi[0][k], s_hi[1][k]), max(s_hi[2][k], s_hi[3][k]));
            if (v != INT_MIN) atomicMax(&info[k < 3 ? 6 + k : 2], (long long)v);
        }
    }
}

extern "C" int lidog_coords_insert_info(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int64_t cap,
                                        int32_t *first_row, int64_t *info, int32_t *err_flag, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "coords_insert: cap must be a power of two >= 2n");
    LIDOG_REQUIRE(info != nullptr, "coords_insert_info: info buffer missing");
    LIDOG_CHECK_HIP(hipMemsetAsync(keys, 0xff, sizeof(uint64_t) * cap, st));
    LIDOG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)vals, 0x7fffffff, cap, st));
    k_info_init<<<1, 16, 0, st>>>((long long *)info);
    if (n == 0) return 0;
    unsigned nb = (unsigned)cdiv64(n, 256);
    k_insert_info<<<nb, 256, 0, st>>>((const int4 *)coords, n, keys, vals, (uint64_t)(cap - 1), first_row, err_flag,
                                      (long long *)info);
    k_first_row<<<nb, 256, 0, st>>>(first_row, n, vals, (unsigned long long *)info);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_coords_insert(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int64_t cap,
                                   int32_t *first_row, int64_t *n_unique_dev, int32_t *err_flag, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "coords_insert: cap must be a power of two >= 2n");
    LIDOG_CHECK_HIP(hipMemsetAsync(keys, 0xff, sizeof(uint64_t) * cap, st));
    LIDOG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)vals, 0x7fffffff, cap, st));
    LIDOG_CHECK_HIP(hipMemsetAsync(n_unique_dev, 0, sizeof(int64_t), st));
    if (n == 0) return 0;
    unsigned nb = (unsigned)cdiv64(n, 256);
    k_insert<<<nb, 256, 0, st>>>((const int4 *)coords, n, 1, keys, vals, (uint64_t)(cap - 1), first_row, err_flag);
    k_first_row<<<nb, 256, 0, st>>>(first_row, n, vals, (unsigned long long *)n_unique_dev);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ compaction of a map with duplicates
__global__ __launch_bounds__(256) void k_flag_first(const int32_t *__restrict__ first_row, int64_t n,
                                                    int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (first_row[i] == (int)i);
}

__global__ __launch_bounds__(256) void k_compact_rows(const int32_t *__restrict__ first_row, int64_t n,
                                                      const int32_t *__restrict__ rank, const int4 *coords,
                                                      const uint64_t *keys, int32_t *vals, uint64_t mask,
                                                      int32_t *__restrict__ unique_rows,
                                                      int32_t *__restrict__ inverse) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int f = first_row[i];
    int r = rank[f];
    inverse[i] = r;
    if (f == (int)i) {
        unique_rows[r] = (int)i;
        int4 c = coords[i];
        int bad = 0;
        uint64_t key = lidog_pack(c.x, c.y, c.z, c.w, &bad);
        uint64_t slot = lidog_mix(key) & mask;
        while (keys[slot] != key) slot = (slot + 1) & mask;
        vals[slot] = r;
    }
}

extern "C" int lidog_coords_compact(const int32_t *first_row, int64_t n, const uint64_t *keys, int32_t *vals,
                                    int64_t cap, const int32_t *coords, int32_t *unique_rows, int32_t *inverse,
                                    int32_t *scan_ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    unsigned nb = (unsigned)cdiv64(n, 256);
    int32_t *rank = scan_ws, *sums = scan_ws + n;
    k_flag_first<<<nb, 256, 0, st>>>(first_row, n, rank);
    if (device_exclusive_scan(rank, n, sums, st)) return 1;
    // vals still hold first rows while every thread reads rank[first_row]; the rewrite of vals only
    // touches slots no other thread reads in this kernel
    k_compact_rows<<<nb, 256, 0, st>>>(first_row, n, rank, (const int4 *)coords, keys, vals, (uint64_t)(cap - 1),
                                       unique_rows, inverse);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ strided map
__global__ __launch_bounds__(256) void k_stride_flag(const int32_t *__restrict__ slot_of, int64_t n,
                                                     const int32_t *__restrict__ vals, int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        int s = slot_of[i];
        flag[i] = (s >= 0 && vals[s] == (int)i);
    }
}

__global__ __launch_bounds__(256) void k_stride_emit(const int4 *__restrict__ coords, int64_t n, int32_t stride,
                                                     const int32_t *__restrict__ slot_of,
                                                     const int32_t *__restrict__ vals,
                                                     const int32_t *__restrict__ rank, int4 *__restrict__ coords_out,
                                                     int32_t *__restrict__ parent2child, int64_t *n_out_dev) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int s = slot_of[i];
    int f = (s >= 0) ? vals[s] : (int)i;
    int r = rank[f];
    parent2child[i] = r;
    bool first = (f == (int)i);
    if (first) {
        int4 c = coords[i];
        auto fl = [stride](int v) {
            int q = v / stride;
            if ((v % stride != 0) && (v < 0)) --q;
            return q * stride;
        };
        c.y = fl(c.y); c.z = fl(c.z); c.w = fl(c.w);
        coords_out[r] = c;
    }
    if (i == n - 1) *n_out_dev = (int64_t)rank[i] + (first ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_stride_rewrite(int64_t n, const int32_t *__restrict__ slot_of,
                                                        int32_t *vals, const int32_t *__restrict__ rank,
                                                        const int32_t *__restrict__ parent2child) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int s = slot_of[i];
    // rank is the exclusive scan of the first-occurrence flags, so a row is a first occurrence
    // exactly when its child row equals its own rank (later rows of the same child have rank > child)
    if (s >= 0 && parent2child[i] == rank[i]) vals[s] = rank[i];
}

extern "C" int lidog_coords_stride(const int32_t *coords_in, int64_t n_in, int32_t new_stride, uint64_t *keys,
                                   int32_t *vals, int64_t cap, int32_t *parent2child, int32_t *coords_out,
                                   int64_t *n_out_dev, int32_t *ws, int32_t *err_flag, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(cap >= 2 * n_in && (cap & (cap - 1)) == 0, "coords_stride: cap must be a power of two >= 2n");
    LIDOG_REQUIRE(n_in < 0x3fffffff, "coords_stride: too many rows");
    LIDOG_CHECK_HIP(hipMemsetAsync(keys, 0xff, sizeof(uint64_t) * cap, st));
    LIDOG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)vals, 0x7fffffff, cap, st));
    LIDOG_CHECK_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), st));
    if (n_in == 0) return 0;
    unsigned nb = (unsigned)cdiv64(n_in, 256);
    int32_t *slot_of = ws, *rank = ws + n_in, *sums = ws + 2 * n_in;
    k_insert<<<nb, 256, 0, st>>>((const int4 *)coords_in, n_in, new_stride, keys, vals, (uint64_t)(cap - 1), slot_of,
                                 err_flag);
    k_stride_flag<<<nb, 256, 0, st>>>(slot_of, n_in, vals, rank);
    if (device_exclusive_scan(rank, n_in, sums, st)) return 1;
    k_stride_emit<<<nb, 256, 0, st>>>((const int4 *)coords_in, n_in, new_stride, slot_of, vals, rank,
                                      (int4 *)coords_out, parent2child, n_out_dev);
    k_stride_rewrite<<<nb, 256, 0, st>>>(n_in, slot_of, vals, rank, parent2child);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ kernel map (neighbour table)
struct KOffsets {
    int32_t d[125 * 3];
};

__global__ __launch_bounds__(256) void k_kernel_map(const int4 *__restrict__ coords_out, int64_t n_out,
                                                    const uint64_t *__restrict__ keys,
                                                    const int32_t *__restrict__ vals, uint64_t mask, KOffsets offs,
                                                    int32_t *__restrict__ nbr) {
    // the 3x3x3 (or 5^3, 2^3) offset neighbourhood is staged in LDS once per workgroup
    __shared__ int32_t s_off[125 * 3];
    const int K = gridDim.y;
    for (int t = threadIdx.x; t < K * 3; t += 256) s_off[t] = offs.d[t];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int k = blockIdx.y;
    int4 c = coords_out[o];
    int bad = 0;
    uint64_t key = lidog_pack(c.x, c.y + s_off[3 * k], c.z + s_off[3 * k + 1], c.w + s_off[3 * k + 2], &bad);
    nbr[(int64_t)k * n_out + o] = bad ? -1 : lidog_find(keys, vals, mask, key);
}

// ---- occupancy bitmap of a coordinate map: one bit per cell of its bounding box (x fastest, then y, z, batch).
// A kernel-map probe first tests the bit of the neighbour's cell: on LiDAR surfaces 84 % of the 3^3 and 90 % of the
// 5^3 neighbours do not exist, and each of those probes was a random 64-byte fetch from the hash table; the bits of a
// voxel's neighbourhood sit in a few cache lines that neighbouring voxels share.  Same table afterwards, bit for bit.
struct BitBox {
    int32_t x0, y0, z0;   // lowest cell (multiples of the map's tensor stride)
    int32_t nx, ny, nz;   // cells per axis
    int32_t stride, nb;   // tensor stride, batch count
};

__device__ __forceinline__ int64_t bit_index(const BitBox &bx, int b, int x, int y, int z) {
    // floor division is exact: coordinates of a map are multiples of its stride, and so are x0, y0, z0
    int ix = (x - bx.x0) / bx.stride, iy = (y - bx.y0) / bx.stride, iz = (z - bx.z0) / bx.stride;
    if ((unsigned)b >= (unsigned)bx.nb || x < bx.x0 || y < bx.y0 || z < bx.z0 || ix >= bx.nx || iy >= bx.ny || iz >= bx.nz)
        return -1;
    return (((int64_t)b * bx.nz + iz) * bx.ny + iy) * bx.nx + ix;
}

__global__ __launch_bounds__(256) void k_bitmap_set(const int4 *__restrict__ coords, int64_t n, BitBox bx,
                                                    uint32_t *__restrict__ bits, int32_t *__restrict__ err) {
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    int4 c = coords[o];
    int64_t i = bit_index(bx, c.x, c.y, c.z, c.w);
    if (i < 0) {
        *err = 2;   // a coordinate outside the box it was sized for: the bitmap cannot be trusted
        return;
    }
    atomicOr(&bits[i >> 5], 1u << (i & 31));
}

// One thread per output voxel, all K offsets: the bits of a voxel's neighbourhood sit in a few words (x is the fastest
// axis of both the offset list and the bitmap: the 3 or 5 x-neighbours of a (y, z) row share a word, and consecutive
// voxels share lines), so the bit tests are cache hits and only the existing neighbours (4.3 of 27, ~12 of 125) go to
// the hash table.  The neighbour table is written k-major as before (coalesced over the voxels of a wave).
__global__ __launch_bounds__(256) void k_kernel_map_bits(const int4 *__restrict__ coords_out, int64_t n_out,
                                                         const uint64_t *__restrict__ keys,
                                                         const int32_t *__restrict__ vals, uint64_t mask, KOffsets offs,
                                                         int K, BitBox bx, const uint32_t *__restrict__ bits,
                                                         int32_t *__restrict__ nbr) {
    __shared__ int32_t s_off[125 * 3];
    for (int t = threadIdx.x; t < K * 3; t += 256) s_off[t] = offs.d[t];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int4 c = coords_out[o];
    // blockIdx.y takes one slab of the offsets (a z-plane of a 5^3 or 3^3 kernel): more threads in flight for what is a
    // chain of dependent cache hits and probes per thread.  Five offsets per round (one x-row of a 5^3 kernel): the bit
    // words are requested together, then the existing neighbours are probed
    const int per = (K + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k_begin = (int)blockIdx.y * per, k_end = k_begin + per < K ? k_begin + per : K;
    for (int k0 = k_begin; k0 < k_end; k0 += 5) {
        int64_t idx[5];
        uint32_t word[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = k0 + j < k_end ? k0 + j : k_end - 1;
            idx[j] = bit_index(bx, c.x, c.y + s_off[3 * k], c.z + s_off[3 * k + 1], c.w + s_off[3 * k + 2]);
            word[j] = bits[idx[j] >= 0 ? idx[j] >> 5 : 0];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = k0 + j;
            if (k >= k_end) break;
            int32_t row = -1;
            if (idx[j] >= 0 && ((word[j] >> (idx[j] & 31)) & 1u)) {
                int bad = 0;
                uint64_t key = lidog_pack(c.x, c.y + s_off[3 * k], c.z + s_off[3 * k + 1], c.w + s_off[3 * k + 2], &bad);
                row = bad ? -1 : lidog_find(keys, vals, mask, key);
            }
            nbr[(int64_t)k * n_out + o] = row;
        }
    }
}

// words (uint32) of the bitmap of a box, or -1 when it would exceed max_bytes
extern "C" int64_t lidog_bitmap_words(int32_t nx, int32_t ny, int32_t nz, int32_t nb, int64_t max_bytes) {
    if (nx <= 0 || ny <= 0 || nz <= 0 || nb <= 0) return -1;
    double cells = (double)nx * ny * nz * nb;
    if (cells / 8.0 > (double)max_bytes) return -1;
    return ((int64_t)nx * ny * nz * nb + 31) / 32;
}

// bits [lidog_bitmap_words] must be zero on entry; err_flag (device int32) is set to 2 if a coordinate lies outside the box
extern "C" int lidog_bitmap_set(const int32_t *coords, int64_t n, int32_t x0, int32_t y0, int32_t z0, int32_t nx,
                                int32_t ny, int32_t nz, int32_t stride, int32_t nb, uint32_t *bits, int32_t *err_flag,
                                void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(stride >= 1 && nx > 0 && ny > 0 && nz > 0 && nb > 0, "bitmap_set: bad box");
    if (n == 0) return 0;
    BitBox bx{x0, y0, z0, nx, ny, nz, stride, nb};
    k_bitmap_set<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>((const int4 *)coords, n, bx, bits, err_flag);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// lidog_kernel_map with the occupancy bitmap of the INPUT map in front of the hash probes (bits == NULL: plain probes)
extern "C" int lidog_kernel_map_bits(const int32_t *coords_out, int64_t n_out, const uint64_t *in_keys,
                                     const int32_t *in_vals, int64_t in_cap, const int32_t *offsets_host, int32_t K,
                                     const uint32_t *bits, int32_t x0, int32_t y0, int32_t z0, int32_t nx, int32_t ny,
                                     int32_t nz, int32_t stride, int32_t nb, int32_t *nbr, void *stream) {
    if (bits == nullptr)
        return lidog_kernel_map(coords_out, n_out, in_keys, in_vals, in_cap, offsets_host, K, nbr, stream);
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && K <= 125, "kernel_map: K=%d out of range [1,125]", K);
    LIDOG_REQUIRE(stride >= 1 && nx > 0 && ny > 0 && nz > 0 && nb > 0, "kernel_map_bits: bad box");
    if (n_out == 0) return 0;
    KOffsets offs;
    for (int i = 0; i < K * 3; ++i) offs.d[i] = offsets_host[i];
    BitBox bx{x0, y0, z0, nx, ny, nz, stride, nb};
    int slabs = (K % 25 == 0) ? K / 25 : (K % 9 == 0) ? K / 9 : 1;     // z-planes of a 5^3 / 3^3 kernel
    dim3 grid((unsigned)cdiv64(n_out, 256), (unsigned)slabs);
    k_kernel_map_bits<<<grid, 256, 0, st>>>((const int4 *)coords_out, n_out, in_keys, in_vals, (uint64_t)(in_cap - 1),
                                            offs, K, bx, bits, nbr);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_kernel_map(const int32_t *coords_out, int64_t n_out, const uint64_t *in_keys,
                                const int32_t *in_vals, int64_t in_cap, const int32_t *offsets_host, int32_t K,
                                int32_t *nbr, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && K <= 125, "kernel_map: K=%d out of range [1,125]", K);
    if (n_out == 0) return 0;
    KOffsets offs;
    for (int i = 0; i < K * 3; ++i) offs.d[i] = offsets_host[i];
    dim3 grid((unsigned)cdiv64(n_out, 256), (unsigned)K);
    k_kernel_map<<<grid, 256, 0, st>>>((const int4 *)coords_out, n_out, in_keys, in_vals, (uint64_t)(in_cap - 1), offs,
                                       nbr);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ a kernel map as a subset of a larger one's offsets
// The 3^3 offsets of a coordinate map onto itself are 27 of the 125 offsets of its 5^3 map (same tensor stride, same
// dilation): the stem's neighbour table (conv0p1s1, utils/models/minkunet_bev.py:57) already holds every neighbour the
// stride-1 3^3 convolutions of block8 (:371) will ask for.  Row k of the small table = row sel[k] of the large one: one
// streaming copy instead of 27 n bitmap tests + hash probes, and the same table bit for bit by construction.
struct KSubset {
    int32_t sel[27];
};

__global__ __launch_bounds__(256) void k_nbr_subset(const int4 *__restrict__ big, int64_t n, KSubset sel,
                                                    int4 *__restrict__ nbr) {
    // one int4 (four rows) per thread; rows of the [K, n] tables are 16-byte aligned when n % 4 == 0
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (q * 4 >= n) return;
    nbr[(int64_t)k * (n / 4) + q] = big[(int64_t)sel.sel[k] * (n / 4) + q];
}

__global__ __launch_bounds__(256) void k_nbr_subset1(const int32_t *__restrict__ big, int64_t n, KSubset sel,
                                                     int32_t *__restrict__ nbr) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (r >= n) return;
    nbr[(int64_t)k * n + r] = big[(int64_t)sel.sel[k] * n + r];
}

extern "C" int lidog_kernel_map_subset(const int32_t *nbr_big, int64_t n, int32_t K_big, const int32_t *sel_host,
                                       int32_t K, int32_t *nbr, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && K <= 27 && K_big >= K && K_big <= 125, "kernel_map_subset: K=%d of K_big=%d", K, K_big);
    if (n == 0) return 0;
    LIDOG_REQUIRE(nbr_big && sel_host && nbr, "kernel_map_subset: null argument");
    KSubset sel;
    for (int k = 0; k < K; ++k) {
        LIDOG_REQUIRE(sel_host[k] >= 0 && sel_host[k] < K_big, "kernel_map_subset: offset %d selects row %d of %d", k,
                      sel_host[k], K_big);
        sel.sel[k] = sel_host[k];
    }
    if (n % 4 == 0 && ((uintptr_t)nbr_big % 16) == 0 && ((uintptr_t)nbr % 16) == 0)
        k_nbr_subset<<<dim3((unsigned)cdiv64(n / 4, 256), (unsigned)K), 256, 0, st>>>((const int4 *)nbr_big, n, sel,
                                                                                     (int4 *)nbr);
    else
        k_nbr_subset1<<<dim3((unsigned)cdiv64(n, 256), (unsigned)K), 256, 0, st>>>(nbr_big, n, sel, nbr);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ rule book (pairs) by ballot + prefix sum
// cnt[k * nbp + b]: number of valid neighbours of offset k among rows [1024 b, 1024 (b+1))
__global__ __launch_bounds__(256) void k_pairs_count(const int32_t *__restrict__ nbr, int64_t n_out, int nbp,
                                                     int32_t *__restrict__ cnt) {
    const int k = blockIdx.y, b = blockIdx.x;
    __shared__ int wsum[4];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t o = (int64_t)b * 1024 + j * 256 + threadIdx.x;
        bool v = (o < n_out) && (nbr[(int64_t)k * n_out + o] >= 0);
        c += __popcll(__ballot(v));  // wave-uniform count
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(int64_t)k * nbp + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup per offset k: exclusive scan of its row-block counts (chunks of 256 with a carry), total -> totals[k];
// a second, one-wave launch turns the K totals into k_off.  (Round 2 had ONE thread per offset walk its nbp counts
// serially: 84 us per map on the map stream, 0.95 ms per training step.)
__global__ __launch_bounds__(256) void k_pairs_scan(int32_t *__restrict__ cnt, int nbp, int32_t *__restrict__ totals) {
    const int k = blockIdx.x;
    int32_t *c = cnt + (int64_t)k * nbp;
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nbp; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < nbp ? c[b] : 0;
        int tot;
        const int ex = block_excl_scan_256(v, &tot);
        const int carry = carry_s;
        if (b < nbp) c[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[k] = carry_s;
}

__global__ __launch_bounds__(128) void k_pairs_koff(const int32_t *__restrict__ totals, int K, int64_t *__restrict__ k_off) {
    __shared__ int64_t s[128];
    const int t = threadIdx.x;
    s[t] = t < K ? totals[t] : 0;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int kk = 0; kk < K; ++kk) {
            k_off[kk] = run;
            run += s[kk];
        }
        k_off[K] = run;
    }
}

__global__ __launch_bounds__(256) void k_pairs_emit(const int32_t *__restrict__ nbr, int64_t n_out, int64_t n_in,
                                                    int nbp, const int32_t *__restrict__ cnt,
                                                    const int64_t *__restrict__ k_off, int32_t *__restrict__ pair_in,
                                                    int32_t *__restrict__ pair_out, int32_t *__restrict__ pos_out,
                                                    int32_t *__restrict__ pos_in) {
    const int k = blockIdx.y, b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ int wcnt[4][4];  // [j][wave]
    int nb_[4];
    unsigned long long masks[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t o = (int64_t)b * 1024 + j * 256 + threadIdx.x;
        nb_[j] = (o < n_out) ? nbr[(int64_t)k * n_out + o] : -1;
        masks[j] = __ballot(nb_[j] >= 0);
        if (lane == 0) wcnt[j][w] = __popcll(masks[j]);
    }
    __syncthreads();
    int64_t base = k_off[k] + cnt[(int64_t)k * nbp + b];
    int run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int before = run;
        for (int ww = 0; ww < 4; ++ww) {
            if (ww < w) before += wcnt[j][ww];
            run += wcnt[j][ww];
        }
        int64_t o = (int64_t)b * 1024 + j * 256 + threadIdx.x;
        if (o < n_out) {
            int32_t p = -1;
            if (nb_[j] >= 0) {
                int rank = before + __popcll(masks[j] & ((1ull << lane) - 1ull));
                p = (int32_t)(base + rank);
                pair_in[p] = nb_[j];
                pair_out[p] = (int32_t)o;
                if (pos_in) pos_in[(int64_t)k * n_in + nb_[j]] = p;
            }
            if (pos_out) pos_out[(int64_t)k * n_out + o] = p;
        }
    }
}

extern "C" int lidog_kernel_map_pairs(const int32_t *nbr, int64_t n_out, int64_t n_in, int32_t K,
                                      int64_t *k_off_dev, int32_t *pair_in, int32_t *pair_out, int32_t *pos_out,
                                      int32_t *pos_in, int32_t *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && K <= 125, "kernel_map_pairs: K=%d out of range [1,125]", K);
    LIDOG_REQUIRE(n_out * (int64_t)K < 0x7fffffff, "kernel_map_pairs: pair positions overflow int32");
    // pos_out / pos_in == NULL: the caller never walks this map by row (the 5^3 stem: 2 x 125 n ints not written)
    if (pos_in) LIDOG_CHECK_HIP(hipMemsetAsync(pos_in, 0xff, sizeof(int32_t) * n_in * K, st));
    if (n_out == 0) {
        LIDOG_CHECK_HIP(hipMemsetAsync(k_off_dev, 0, sizeof(int64_t) * (K + 1), st));
        return 0;
    }
    int nbp = (int)cdiv64(n_out, 1024);
    int32_t *cnt = ws, *totals = ws + (int64_t)nbp * K;
    dim3 grid((unsigned)nbp, (unsigned)K);
    k_pairs_count<<<grid, 256, 0, st>>>(nbr, n_out, nbp, cnt);
    k_pairs_scan<<<(unsigned)K, 256, 0, st>>>(cnt, nbp, totals);
    k_pairs_koff<<<1, 128, 0, st>>>(totals, K, k_off_dev);
    k_pairs_emit<<<grid, 256, 0, st>>>(nbr, n_out, n_in, nbp, cnt, k_off_dev, pair_in, pair_out, pos_out, pos_in);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ per-row lists of the rule book (for the reduction)
// pos [K][n] (pair position of (offset k, row o) or -1) -> row_ptr [n+1], row_list [P]: the pair positions of row o in
// ascending offset order.  The reduction pass then reads exactly the product rows a voxel has (4.3 on average for a
// 3^3 kernel on LiDAR surfaces) instead of probing all K offsets (27 index loads and 27 row loads per voxel, 84 % of
// them for missing neighbours).
__global__ __launch_bounds__(256) void k_rows_count(const int32_t *__restrict__ pos, int64_t n, int K,
                                                    int32_t *__restrict__ row_ptr) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o > n) return;
    int c = 0;
    if (o < n)
        for (int k = 0; k < K; ++k) c += pos[(int64_t)k * n + o] >= 0;
    row_ptr[o] = c;   // row_ptr[n] = 0: the exclusive scan leaves the total there
}

__global__ __launch_bounds__(256) void k_rows_fill(const int32_t *__restrict__ pos, int64_t n, int K, int mark_k,
                                                   const int32_t *__restrict__ row_ptr,
                                                   int32_t *__restrict__ row_list) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    int p = row_ptr[o];
    for (int k = 0; k < K; ++k) {
        const int v = pos[(int64_t)k * n + o];
        if (v >= 0) row_list[p++] = k == mark_k ? -1 : v;
    }
}

// ws: ceil((n + 1) / 1024) + 1 ints.  mark_k >= 0: the entry of that offset is stored as -1 (for a consumer that
// computes that offset's product itself; this library passes -1: every entry is a pair position).
extern "C" int lidog_kernel_map_rows(const int32_t *pos, int64_t n, int32_t K, int32_t mark_k, int32_t *row_ptr,
                                     int32_t *row_list, int32_t *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(n >= 0 && K >= 1 && n < ((int64_t)1 << 31) - 2, "kernel_map_rows: bad sizes");
    k_rows_count<<<(unsigned)cdiv64(n + 1, 256), 256, 0, st>>>(pos, n, K, row_ptr);
    if (device_exclusive_scan(row_ptr, n + 1, ws, st)) return 1;
    if (n) k_rows_fill<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(pos, n, K, mark_k, row_ptr, row_list);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
