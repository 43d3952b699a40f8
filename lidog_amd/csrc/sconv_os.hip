// Output-stationary sparse convolution for 3^3 kernels on one coordinate map (MinkowskiConvolution(kernel_size=3,
// stride=1) of the BasicBlocks, utils/models/minkunet_bev.py:425-439 / resnet_block.py:8-56): the product rows `T` of
// the two-pass path (gathered GEMM -> T [P, Cout] -> per-row reduction) never exist.
//
// A workgroup owns 128 OUTPUT rows x 32 NT columns and walks the kernel offsets k in ascending order: for every k it
// gathers the rows' neighbours nbr[k][row] (a missing neighbour is a zero row), multiplies by W[k] into a scratch
// accumulator -- the same k-ordered fmaf chain from zero as a product row of the two-pass path -- and adds that
// accumulator to the running sum of the rows: out = ((0 + T_k0) + T_k1) + ..., the additions of the reduction pass in
// the same order.  Results are therefore bit-identical to the two-pass path (and to oracle/me_oracle.c) up to the sign
// of a zero; what changes is the traffic (no 4 P Cout bytes written and read back: 1.2 GB per stride-1 96-channel
// layer at batch 4) and the length of a workgroup's matrix phase (every offset of the tile instead of one).
//
// Done naively this multiplies zero rows 2.2 x as often as real ones (a LiDAR voxel has 4.5 of its 27 neighbours).
// Rows are therefore SORTED by their neighbour mask first (lidog_kernel_map_sorted: 27-bit mask with the rarest offset
// as the most significant bit, stable radix sort), which puts rows with the same neighbour pattern into the same tile:
// a tile skips every offset none of its rows has, a wave skips the matrix work of every offset none of its 32 rows
// has.  Measured on the bench scans: 85 % of the 32-row x offset blocks that are multiplied are real pairs at tensor
// strides 1 and 2 (80 % at strides 4 / 8, 69 % at 16).
//
// The data gradient of such a convolution is the same sum over the MIRRORED offsets (the map is symmetric: row o has
// input neighbour i at offset k  <=>  row i has neighbour o at offset K-1-k), so the same sorted rows and masks serve
// both directions: `reverse` walks the bits from the top and takes the transposed weights of offset K-1-k.
#include <stdlib.h>

#include "common.h"
#include "clock_stamp.h"
#include "sconv_mfma.h"
#include "stats_tail.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OS_TM 128
#define OS_BK 32
#define OS_SA 33
#define OS_MAXK 27

// ------------------------------------------------------------------ sorted rows of a map
// pos[k]: bit position of offset k in the sort key = number of offsets that occur MORE often (ties: lower k first), so
// the rarest offset is the most significant bit
__global__ void k_os_bitpos(const int64_t *__restrict__ k_off, int K, int32_t *__restrict__ pos) {
    const int k = threadIdx.x;
    if (k >= K) return;
    const int64_t mine = k_off[k + 1] - k_off[k];
    int p = 0;
    for (int j = 0; j < K; ++j) {
        const int64_t c = k_off[j + 1] - k_off[j];
        if (c > mine || (c == mine && j < k)) ++p;
    }
    pos[k] = p;
}

__global__ __launch_bounds__(256) void k_os_keys(const int32_t *__restrict__ nbr, int64_t n, int K,
                                                 const int32_t *__restrict__ pos, uint32_t *__restrict__ keys,
                                                 int32_t *__restrict__ rows, uint32_t *__restrict__ masks) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t key = 0, m = 0;
    for (int k = 0; k < K; ++k) {
        const uint32_t has = nbr[(int64_t)k * n + r] >= 0 ? 1u : 0u;
        m |= has << k;
        key |= has << pos[k];
    }
    keys[r] = key;
    rows[r] = (int32_t)r;
    masks[r] = m;
}

// one wave per 32 sorted rows: the OR of their masks; rows past the end of the map are -1 in `perm`
__global__ __launch_bounds__(256) void k_os_wave_masks(const uint32_t *__restrict__ masks, int32_t *__restrict__ perm,
                                                       int64_t n, int64_t n_pad, uint32_t *__restrict__ wave_masks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t m = 0;
    if (i < n) m = masks[perm[i]];
    else if (i < n_pad) perm[i] = -1;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) m |= __shfl_xor(m, d);
    if (i < n_pad && (threadIdx.x & 31) == 0) wave_masks[i >> 5] = m;
}

// ---- stable LSD radix sort of (key, row) pairs, hand-written for wave64: RS_BITS bits per pass.  A workgroup is ONE
// wavefront that owns RS_CHUNK consecutive elements: (1) per-chunk digit histogram (+ per-digit totals), (2) exclusive scan
// of the digit-major [RS_BINS][chunks] table, one wavefront per digit, (3) scatter -- the chunk is walked 64 elements at a time in
// order; the lanes holding the same digit find each other with RS_BITS ballots, rank = popcount of the lower peers, the
// lowest peer advances the digit's cursor: equal keys keep their input order (what the numpy restatement in
// tests/test_gpu_sconv_os.py calls a stable argsort).  Replaces the CUB-compatibility wrapper the file used in round 4
// (~10 launches of rocPRIM's merge-sort fallback per sort; here 3 per pass).
#define RS_BITS 9
#define RS_BINS (1 << RS_BITS)
#define RS_CHUNK 1024
#define OS_TILE_ORDER_LDS 16384

// hist [RS_BINS][chunks] (digit-major) + totals [RS_BINS] (zeroed by the caller; integer atomics: order-independent)
__global__ __launch_bounds__(64) void k_rs_hist(const uint32_t *__restrict__ keys, int64_t n, int shift, int chunks,
                                                int32_t *__restrict__ hist, int32_t *__restrict__ totals) {
    __shared__ int32_t cnt[RS_BINS];
    const int lane = threadIdx.x;
    for (int i = lane; i < RS_BINS; i += 64) cnt[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
    uint32_t kreg[RS_CHUNK / 64];      // every load of the chunk in flight before the first use
#pragma unroll
    for (int r = 0; r < RS_CHUNK / 64; ++r) {
        const int64_t i = lo + 64 * r + lane;
        kreg[r] = keys[i < hi ? i : hi - 1];
    }
#pragma unroll
    for (int r = 0; r < RS_CHUNK / 64; ++r)
        if (lo + 64 * r + lane < hi) atomicAdd(&cnt[(kreg[r] >> shift) & (RS_BINS - 1)], 1);
    __syncthreads();
    for (int i = lane; i < RS_BINS; i += 64) {
        const int32_t c = cnt[i];
        hist[(int64_t)i * chunks + blockIdx.x] = c;
        if (c) atomicAdd(&totals[i], c);
    }
}

// one wavefront per digit d: hist[d][*] becomes the exclusive scan of the digit's chunk counts, offset by the number of
// keys with a smaller digit (the sum of totals[0 .. d)) -- the global position of the first key of (digit, chunk)
__global__ __launch_bounds__(64) void k_rs_scan(int32_t *__restrict__ hist, const int32_t *__restrict__ totals, int chunks) {
    const int d = blockIdx.x, lane = threadIdx.x;
    int32_t before = 0;
    for (int i = lane; i < d; i += 64) before += totals[i];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) before += __shfl_xor(before, s);
    int32_t *row = hist + (int64_t)d * chunks;
    int32_t run = before;
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        const int32_t v = c < chunks ? row[c] : 0;
        int32_t inc = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int32_t o = __shfl_up(inc, s);
            if (lane >= s) inc += o;
        }
        if (c < chunks) row[c] = run + inc - v;
        run += __shfl(inc, 63);
    }
}

__global__ __launch_bounds__(64) void k_rs_scatter(const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                                   int64_t n, int shift, int chunks, const int32_t *__restrict__ hist,
                                                   uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out) {
    __shared__ int32_t cur[RS_BINS];
    const int lane = threadIdx.x;
    for (int i = lane; i < RS_BINS; i += 64) cur[i] = hist[(int64_t)i * chunks + blockIdx.x];
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t kreg[RS_CHUNK / 64];
    int32_t vreg[RS_CHUNK / 64];
#pragma unroll
    for (int r = 0; r < RS_CHUNK / 64; ++r) {
        const int64_t i = lo + 64 * r + lane;
        kreg[r] = keys[i < hi ? i : hi - 1];
        vreg[r] = vals[i < hi ? i : hi - 1];
    }
#pragma unroll
    for (int r = 0; r < RS_CHUNK / 64; ++r) {
        const int64_t i = lo + 64 * r + lane;
        const bool live = i < hi;
        const uint32_t key = kreg[r];
        const int32_t val = vreg[r];
        const uint32_t d = (key >> shift) & (RS_BINS - 1);
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const int rank = __popcll(peers & below);
        const int32_t at = cur[d];
        __syncthreads();   // every lane has read its cursor before the leaders move them (one wave: costs nothing)
        if (live && rank == 0) cur[d] = at + __popcll(peers);
        __syncthreads();
        if (live) {
            keys_out[at + rank] = key;
            vals_out[at + rank] = val;
        }
    }
}

// launch order of the tiles: the ones with the most (wave, offset) blocks to multiply first, so that the grid ends on its
// shortest workgroups (a tile walks 5 to 27 offsets).  key = 127 - blocks, ascending, ties by tile id: a counting sort by
// ONE wavefront (a few thousand tiles; the same ballot ranking as k_rs_scatter).
__global__ __launch_bounds__(64) void k_os_tile_order(const uint32_t *__restrict__ wave_masks, int n_tiles,
                                                      int32_t *__restrict__ tile_order) {
    __shared__ int32_t cur[128];
    __shared__ uint8_t s_key[OS_TILE_ORDER_LDS];    // the tiles' keys (one coalesced 16-byte load per tile), if they fit
    const int lane = threadIdx.x;
    cur[lane] = 0;
    cur[lane + 64] = 0;
    __syncthreads();
    const bool in_lds = n_tiles <= OS_TILE_ORDER_LDS;
    auto key_global = [&](int t) {
        const uint4 m = reinterpret_cast<const uint4 *>(wave_masks)[t];
        return 127 - (__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));     // blocks <= 4 * 27
    };
    auto key_of = [&](int t) { return in_lds ? (int)s_key[t] : key_global(t); };
    for (int t = lane; t < n_tiles; t += 64) {
        const int k = key_global(t);
        if (in_lds) s_key[t] = (uint8_t)k;
        atomicAdd(&cur[k], 1);
    }
    __syncthreads();
    if (lane == 0) {
        int run = 0;
        for (int i = 0; i < 128; ++i) {
            const int c = cur[i];
            cur[i] = run;
            run += c;
        }
    }
    __syncthreads();
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int t0 = 0; t0 < n_tiles; t0 += 64) {
        const int t = t0 + lane;
        const bool live = t < n_tiles;
        const int d = live ? key_of(t) : 0;
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const int rank = __popcll(peers & below);
        const int at = cur[d];
        __syncthreads();
        if (live && rank == 0) cur[d] = at + __popcll(peers);
        __syncthreads();
        if (live) tile_order[at + rank] = t;
    }
}

static int64_t os_pad(int64_t n) { return (n + OS_TM - 1) / OS_TM * OS_TM; }

static int64_t rs_chunks(int64_t n) { return cdiv64(n, RS_CHUNK); }

// bytes of workspace lidog_kernel_map_sorted needs for a map of n rows
extern "C" int64_t lidog_kernel_map_sorted_ws(int64_t n) {
    if (n <= 0) return 256;
    return 256 + 4 * 4 * n + 4 * RS_BINS * (rs_chunks(n) + 4) + 4096;
}

// perm [pad128(n)] int32: the rows in sorted order (-1 behind the last one); wave_masks [pad128(n) / 32] uint32: the OR
// of the neighbour masks of every 32 sorted rows (four consecutive words = one 128-row tile); tile_order
// [pad128(n) / 128] int32: the tiles in launch order (most blocks first)
extern "C" int lidog_kernel_map_sorted(const int32_t *nbr, int64_t n, int32_t K, const int64_t *k_off, int32_t *perm,
                                       uint32_t *wave_masks, int32_t *tile_order, void *ws, int64_t ws_bytes,
                                       void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && K <= OS_MAXK, "kernel_map_sorted: 1 <= K <= %d", OS_MAXK);
    if (n == 0) return 0;
    LIDOG_REQUIRE(nbr && k_off && perm && wave_masks && tile_order && ws && n < ((int64_t)1 << 31),
                  "kernel_map_sorted: bad arguments");
    LIDOG_REQUIRE(ws_bytes >= lidog_kernel_map_sorted_ws(n), "kernel_map_sorted: workspace too small");
    char *p = (char *)ws;
    int32_t *pos = (int32_t *)p;                  p += 256;
    uint32_t *keys = (uint32_t *)p;               p += 4 * n;
    uint32_t *keys_out = (uint32_t *)p;           p += 4 * n;
    int32_t *rows = (int32_t *)p;                 p += 4 * n;
    uint32_t *masks = (uint32_t *)p;              p += 4 * n;
    int32_t *hist = (int32_t *)p;
    const int chunks = (int)rs_chunks(n);
    int32_t *totals = hist + (int64_t)RS_BINS * chunks;      // [3][RS_BINS], one row per pass (zeroed once)
    const int passes = (K + RS_BITS - 1) / RS_BITS;
    // the pairs ping-pong between (keys, a) and (keys_out, b); the row ids start where an odd / even number of passes
    // leaves them in `perm`
    int32_t *va = (passes & 1) ? rows : perm, *vb = (passes & 1) ? perm : rows;
    k_os_bitpos<<<1, 32, 0, st>>>(k_off, K, pos);
    k_os_keys<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(nbr, n, K, pos, keys, va, masks);
    uint32_t *ka = keys, *kb = keys_out;
    LIDOG_CHECK_HIP(hipMemsetAsync(totals, 0, sizeof(int32_t) * RS_BINS * passes, st));
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * RS_BITS;
        int32_t *tot = totals + pass * RS_BINS;
        k_rs_hist<<<chunks, 64, 0, st>>>(ka, n, shift, chunks, hist, tot);
        k_rs_scan<<<RS_BINS, 64, 0, st>>>(hist, tot, chunks);
        k_rs_scatter<<<chunks, 64, 0, st>>>(ka, va, n, shift, chunks, hist, kb, vb);
        uint32_t *tk = ka; ka = kb; kb = tk;
        int32_t *tv = va; va = vb; vb = tv;
    }
    const int64_t n_pad = os_pad(n);
    k_os_wave_masks<<<(unsigned)cdiv64(n_pad, 256), 256, 0, st>>>(masks, perm, n, n_pad, wave_masks);
    k_os_tile_order<<<1, 64, 0, st>>>(wave_masks, (int)(n_pad / OS_TM), tile_order);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ the convolution
template <int NT>
__device__ __forceinline__ void os_frag_load(const float *p, float (&f)[NT]) {
    if constexpr (NT == 4) {
        float4 v = *reinterpret_cast<const float4 *>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else if constexpr (NT == 2) {
        float2 v = *reinterpret_cast<const float2 *>(p);
        f[0] = v.x; f[1] = v.y;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) f[t] = p[t];
    }
}

// What the epilogue adds to the store of the rows: mode 1 = the BatchNorm statistics of the result, per-channel
// (sum x, sum x^2) in fp64, one partial row per tile, added in a fixed order and finalised by the last workgroup to
// arrive (stats_tail.h) -- the job of the reduction pass's statistics form (sconv.hip:k_sconv_reduce_rows4_stats).
// (The backward-statistics form of that pass was built here too and measured: its epilogue -- the producer's saved input
// and mask re-read per tile row -- costs what the stand-alone reduction costs, 0.047 vs 0.055 ms on a stride-1
// 96-channel layer, and its sums are not the operator path's bits; the data gradient therefore stays plain + addend.)
//   mode 3 = an evaluation-mode BatchNorm (+ residual + ReLU) applied to the result before it is stored: the validation
// path's fused form (sconv.hip:k_sconv_reduce_rows4_bn), same expression and operation order as bn.hip:k_bn_apply4.
struct OsStats {
    int mode;
    StatsTail tail;
    const float *bn_mean, *bn_invstd, *bn_w, *bn_b, *bn_res;
    int bn_relu;
};

// Epilogue: bias, addend, the rows' store through `s_row` (canonical row of every tile
// row, -1 behind the end of the map) and the statistics of OsStats.  `red`: >= 4 * 2 * TN doubles of LDS the caller no
// longer needs.  Called by all 256 threads of the workgroup.
template <int NT>
__device__ __forceinline__ void os_epilogue(f32x16 (&acc)[NT], const float (&bv)[NT], const int32_t *s_row, double *red,
                                            int tile, int col0, int Cout, const float *__restrict__ addend,
                                            float *__restrict__ out, const OsStats &st) {
    constexpr int TN = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the last, redundant prefetch is drained once (vmcnt 0), see sconv_mfma.hip
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] += bv[t];
    // per-lane statistics of this lane's 16 rows x NT columns (fp64), rows past the end of the map left out
    double s0[NT], s1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s0[t] = s1[t] = 0.0;
    float e_m[NT], e_is[NT], e_w[NT], e_b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        e_m[t] = 0.f; e_is[t] = 1.f; e_w[t] = 1.f; e_b[t] = 0.f;
    }
    if (st.mode == 3) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = col0 + li * NT + t;
            e_m[t] = st.bn_mean[c]; e_is[t] = st.bn_invstd[c]; e_w[t] = st.bn_w[c]; e_b[t] = st.bn_b[c];
        }
    }
    // rows in two batches of eight: the addend loads of a batch are issued before the first use -- unconditionally, a row
    // behind the end of the map reads row 0 and is masked afterwards (loads inside the per-row branch wait for memory
    // sixteen times in a row)
#pragma unroll
    for (int eb = 0; eb < 16; eb += 8) {
        int dst[8];
        float adv[8][NT], resv[8][NT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u;
            dst[u] = s_row[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
            const size_t at = (size_t)(dst[u] < 0 ? 0 : dst[u]) * Cout + col0 + li * NT;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                adv[u][t] = addend ? addend[at + t] : 0.f;
                resv[u][t] = (st.mode == 3 && st.bn_res) ? st.bn_res[at + t] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u;
            const bool ok = dst[u] >= 0;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = acc[t][e] + adv[u][t];   // no addend: + 0
            if (st.mode == 3) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float y = (v[t] - e_m[t]) * e_is[t] * e_w[t] + e_b[t];
                    if (st.bn_res) y += resv[u][t];
                    if (st.bn_relu) y = fmaxf(y, 0.f);
                    v[t] = y;
                }
            }
            if (ok) {
                float *o = out + (size_t)dst[u] * Cout + col0 + li * NT;
                if constexpr (NT == 4) {
                    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (NT == 2) {
                    *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = v[t];
                }
            }
            if (st.mode == 1) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float x = ok ? v[t] : 0.f;
                    s0[t] += (double)x;
                    s1[t] += (double)x * (double)x;
                }
            }
        }
    }
    if (st.mode == 1) {
        // the tile's column sums: the two row halves of a wave (lanes l, l + 32), then the four waves in order
        __syncthreads();   // every wave is done with the LDS that `red` aliases
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s0[t] += __shfl_xor(s0[t], 32);
            s1[t] += __shfl_xor(s1[t], 32);
            if (kh == 0) {
                red[(wave * 2 + 0) * TN + li * NT + t] = s0[t];
                red[(wave * 2 + 1) * TN + li * NT + t] = s1[t];
            }
        }
        __syncthreads();
        const int C2 = 2 * st.tail.C;
        if (tid < 2 * TN) {
            const int which = tid / TN, c = tid - which * TN;
            const double v = ((red[(0 * 2 + which) * TN + c] + red[(1 * 2 + which) * TN + c]) +
                              red[(2 * 2 + which) * TN + c]) + red[(3 * 2 + which) * TN + c];
            lidog_store_sc1(st.tail.partial + (size_t)tile * C2 + which * st.tail.C + col0 + c, v);
        }
        lidog_stats_tail_rows(st.tail, tile, (int)gridDim.x, (int)gridDim.y);
    }
}

// MFMA 32x32x2 lane maps as in sconv_mfma.hip: A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31],
// D[i][j]: j = l & 31, i = (e & 3) + 8 (e >> 2) + 4 (l >> 5).  MFMA column tile t of a wave = columns {li * NT + t}.
// FOLD: the input rows are the raw output of the layer before; its BatchNorm + ReLU (InBn, sconv_mfma.h) is applied while
// they are staged (a missing neighbour stays an exact zero).
template <int NT, int MINW, bool FOLD = false>
__global__ __launch_bounds__(256, MINW) void k_sconv_os_mfma(const float *__restrict__ A, const int32_t *__restrict__ nbr,
                                                       int64_t n, int K, const int32_t *__restrict__ perm,
                                                       const uint32_t *__restrict__ wave_masks,
                                                       const int32_t *__restrict__ tile_order,
                                                       const float *__restrict__ W, int reverse,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ addend, int Cin, int Cout,
                                                       float *__restrict__ out, OsStats st, InBn in_bn, int xcd_group) {
    constexpr int TN = 32 * NT;
    constexpr int BV = (OS_BK * TN / 4) / 256;
    __shared__ float As[OS_TM * OS_SA];
    __shared__ __attribute__((aligned(16))) float Bs[OS_BK * TN];
    __shared__ int32_t s_row[OS_TM];
    __shared__ int32_t s_nbr[OS_MAXK * OS_TM];   // neighbour row of (offset, tile row), -1 = none

    // heaviest tiles first; with xcd_group > 0 runs of that many consecutive tiles of the order (neighbouring rows of one
    // mask class: they gather overlapping sets of input rows) land on the same XCD and its L2 (workgroups go to the 8
    // XCDs round-robin)
    int slot = blockIdx.x;
    if (xcd_group > 0) {
        const int per_round = 8 * xcd_group, full = ((int)gridDim.x / per_round) * per_round;
        if (slot < full) {
            const int xcd = slot & 7, j = slot >> 3;
            slot = ((j / xcd_group) * 8 + xcd) * xcd_group + j % xcd_group;
        }
    }
    const int tile = tile_order[slot];
    const int col0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    LIDOG_STAMP_BEGIN()

    const uint32_t wm = wave_masks[tile * 4 + wave];
    const uint32_t tm = wave_masks[tile * 4] | wave_masks[tile * 4 + 1] | wave_masks[tile * 4 + 2] | wave_masks[tile * 4 + 3];
    if (tid < OS_TM) s_row[tid] = perm[(int64_t)tile * OS_TM + tid];
    __syncthreads();
    // neighbour table of the tile (one gather per (offset, row), all issued before use); offsets the tile does not
    // have are never read back
    for (int e = tid; e < K * OS_TM; e += 256) {
        const int k = e / OS_TM, r = e - k * OS_TM;
        const int row = s_row[r];
        s_nbr[e] = (row >= 0 && ((tm >> k) & 1u)) ? nbr[(int64_t)k * n + row] : -1;
    }
    __syncthreads();

    f32x16 acc[NT], part[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[t][e] = 0.f;
            part[t][e] = 0.f;
        }
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = -0.0f;
    if (bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bias[col0 + li * NT + t];
    }

    // staging: thread handles the float4 (tid & 7) of tile rows (tid >> 3) + 32 j
    float4 ra[4], rb0, rb1, rb2, rb3;
    float4 pm, ps, pw, pb;
    pm = ps = pw = pb = make_float4(0.f, 0.f, 0.f, 0.f);
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *a_row[4];
    bool ok_cur[4], ok_nxt[4];
    const int q4 = (tid & 7) * 4;
#define OS_LOADB(J, R)                                                                     \
    if constexpr (BV > J) {                                                                \
        int f = tid + 256 * J;                                                             \
        int kk = f / (TN / 4), c4 = f % (TN / 4);                                          \
        R = *reinterpret_cast<const float4 *>(Bk + (size_t)(kb + kk) * Cout + c4 * 4);      \
    }
#define OS_STOREB(J, R) \
    if constexpr (BV > J) *reinterpret_cast<float4 *>(&Bs[(tid + 256 * J) * 4]) = R;
    auto rows_of = [&](int k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int src = s_nbr[k * OS_TM + (tid >> 3) + 32 * j];
            ok_nxt[j] = src >= 0;
            a_row[j] = A + (size_t)(src < 0 ? 0 : src) * Cin + q4;
        }
    };
    auto load_chunk = [&](int k, int kb) {
        const float *Bk = W + (size_t)(reverse ? K - 1 - k : k) * Cin * Cout + col0;
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4 *>(a_row[j] + kb);
        if constexpr (FOLD) {   // the thread's four rows share the channel quad q4 of the chunk
            pm = *reinterpret_cast<const float4 *>(in_bn.mean + kb + q4);
            ps = *reinterpret_cast<const float4 *>(in_bn.invstd + kb + q4);
            pw = *reinterpret_cast<const float4 *>(in_bn.w + kb + q4);
            pb = *reinterpret_cast<const float4 *>(in_bn.b + kb + q4);
        }
        OS_LOADB(0, rb0) OS_LOADB(1, rb1) OS_LOADB(2, rb2) OS_LOADB(3, rb3)
    };
    auto next_offset = [&](uint32_t rem) { return reverse ? 31 - __builtin_clz(rem) : __builtin_ctz(rem); };

    // (Round 5: TWO chunks of gathered rows in flight -- a second set of staging registers, the weight loads of chunk c+1
    // issued before the row loads of chunk c+2 so that the in-order load counter lets the store of chunk c+1 go ahead;
    // 194 registers, bit-identical -- measured: 0.382 vs 0.383 ms forward and 0.423 vs 0.413 ms data gradient on the
    // stride-1 96-channel layer alone, 47.45 / 47.49 vs 47.43 / 47.49 ms in the step.  The 20 % this loop gains without
    // any global load is not the loads' latency.  Not kept; profiles/r05_ab_os_two_chunks_in_flight.txt.)
    uint32_t rem = tm;
    if (rem != 0) {
        int k_cur = next_offset(rem);
        rem &= ~(1u << k_cur);
        int kb_cur = 0;
        rows_of(k_cur);
#pragma unroll
        for (int j = 0; j < 4; ++j) ok_cur[j] = ok_nxt[j];
        load_chunk(k_cur, 0);
        for (;;) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = tid + 256 * j;
                const int o = (f >> 3) * OS_SA + (f & 7) * 4;
                const bool ok = ok_cur[j];
                float4 v = ra[j];
                if constexpr (FOLD) v = in_bn_apply(v, pm, ps, pw, pb, in_bn.relu);
                As[o] = ok ? v.x : 0.f;
                As[o + 1] = ok ? v.y : 0.f;
                As[o + 2] = ok ? v.z : 0.f;
                As[o + 3] = ok ? v.w : 0.f;
            }
            OS_STOREB(0, rb0) OS_STOREB(1, rb1) OS_STOREB(2, rb2) OS_STOREB(3, rb3)
            __syncthreads();
            // the chunk after this one: the next 32 channels of this offset, or the first 32 of the tile's next offset;
            // past the end the current chunk is fetched again (an unconditional load: a branch around the staging
            // registers sends them through scratch memory)
            const bool last_of_offset = kb_cur + OS_BK >= Cin;
            const bool more = !last_of_offset || rem != 0;
            int k_nxt = k_cur, kb_nxt = kb_cur + OS_BK;
            if (last_of_offset) {
                kb_nxt = 0;
                if (rem != 0) {
                    k_nxt = next_offset(rem);
                    rem &= ~(1u << k_nxt);
                    rows_of(k_nxt);
                }
            }
            load_chunk(k_nxt, kb_nxt);
            if ((wm >> k_cur) & 1u) {
                const float *arow = &As[(wave * 32 + li) * OS_SA + kh];
                const float *bcol = &Bs[kh * TN + li * NT];
                float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0, an1;
                os_frag_load<NT>(bcol, bq0);
                os_frag_load<NT>(bcol + 2 * TN, bq1);
                a0 = arow[0];
                a1 = arow[2];
#pragma unroll
                for (int j = 0; j < OS_BK / 4; ++j) {
                    if (j + 1 < OS_BK / 4) {
                        os_frag_load<NT>(bcol + (4 * j + 4) * TN, bn0);
                        os_frag_load<NT>(bcol + (4 * j + 6) * TN, bn1);
                        an0 = arow[4 * j + 4];
                        an1 = arow[4 * j + 6];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) part[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], part[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) part[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], part[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 1 < OS_BK / 4) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            bq0[t] = bn0[t];
                            bq1[t] = bn1[t];
                        }
                        a0 = an0;
                        a1 = an1;
                    }
                }
                if (last_of_offset) {   // the offset's product rows are complete: the reduction pass's addition
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            acc[t][e] += part[t][e];
                            part[t][e] = 0.f;
                        }
                }
            }
            if (!more) break;
            if (last_of_offset) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ok_cur[j] = ok_nxt[j];
            }
            k_cur = k_nxt;
            kb_cur = kb_nxt;
        }
    }
    os_epilogue<NT>(acc, bv, s_row, reinterpret_cast<double *>(As), tile, col0, Cout, addend, out, st);
    LIDOG_STAMP_END()
}

static int os_launch(const float *A, const int32_t *nbr, int64_t n, int K, const int32_t *perm,
                     const uint32_t *wave_masks, const int32_t *tile_order, const float *W, int reverse,
                     const float *bias, const float *addend, int Cin, int Cout, float *out, const OsStats &stats,
                     hipStream_t st, InBn in_bn = InBn{nullptr, nullptr, nullptr, nullptr, 0}) {
    LIDOG_REQUIRE(K >= 1 && K <= OS_MAXK && Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0,
                  "sconv_os: K <= %d, channel counts multiples of 32 (got K %d, %d -> %d)", OS_MAXK, K, Cin, Cout);
    LIDOG_REQUIRE(A && nbr && perm && wave_masks && tile_order && W && out, "sconv_os: null argument");
    int nt = (Cout % 128 == 0) ? 4 : (Cout % 96 == 0) ? 3 : (Cout % 64 == 0) ? 2 : 1;
    dim3 grid((unsigned)(os_pad(n) / OS_TM), (unsigned)(Cout / (32 * nt)));
    // runs of 4 consecutive tiles of the launch order per XCD (measured in the step, same box, alternating: 48.13 / 48.07 ->
    // 48.07 / 47.99 ms; 2: 48.11 / 47.95; 8: 48.14 / 48.20; 16: 48.23 / 48.35; 64: 50.3 -- the heaviest-first order must
    // survive)
    const int xcd_group = 4;
#define OS_LAUNCHF(NT_, MW_, F_)                                                                                 \
    k_sconv_os_mfma<NT_, MW_, F_><<<grid, 256, 0, st>>>(A, nbr, n, K, perm, wave_masks, tile_order, W, reverse, bias, \
                                                        addend, Cin, Cout, out, stats, in_bn, xcd_group)
#define OS_LAUNCH(NT_, MW_) OS_LAUNCHF(NT_, MW_, false)
    if (in_bn.mean) {   // 16 more registers: the 64-column kernel no longer fits four waves per SIMD
        switch (nt) {
            case 4: OS_LAUNCHF(4, 2, true); break;
            case 3: OS_LAUNCHF(3, 2, true); break;
            case 2: OS_LAUNCHF(2, 2, true); break;
            default: OS_LAUNCHF(1, 4, true);
        }
        LIDOG_LAUNCH_CHECK();
        return 0;
    }
    // (three waves per SIMD for the 96- / 128-column kernels spill: measured slower, round 4)
    switch (nt) {
        case 4: OS_LAUNCH(4, 2); break;
        case 3: OS_LAUNCH(3, 2); break;
        case 2: OS_LAUNCH(2, 4); break;
        default: OS_LAUNCH(1, 4);
    }
#undef OS_LAUNCH
#undef OS_LAUNCHF
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// out [n, Cout] = sum over k ascending of A[nbr[k][row]] W[k]  (+ bias) (+ addend), rows in canonical order.
// reverse != 0: the data gradient over a symmetric 3^3 map: offsets walked from the top, weights W[K-1-k] (pass the
// transposed kernels [K][Cout][Cin] with Cin / Cout swapped).  perm / wave_masks / tile_order: lidog_kernel_map_sorted.
extern "C" int lidog_sconv_os(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                              const uint32_t *wave_masks, const int32_t *tile_order, const float *W, int32_t reverse,
                              const float *bias, const float *addend, int32_t Cin, int32_t Cout, float *out,
                              void *stream) {
    if (n == 0) return 0;
    OsStats stats = {};
    return os_launch(A, nbr, n, K, perm, wave_masks, tile_order, W, reverse, bias, addend, Cin, Cout, out, stats,
                     (hipStream_t)stream);
}

// doubles of workspace of the two statistics forms: one partial row per 128-row tile + the group rows of the tail
extern "C" int64_t lidog_sconv_os_stats_ws(int64_t n, int32_t C) {
    return (os_pad(n) / OS_TM + STATS_MAX_GROUPS) * (int64_t)2 * C;
}

// Forward convolution + the BatchNorm statistics of its result (= lidog_sconv_gemm + lidog_sconv_reduce_rows_stats):
// sums [2 Cout + 1], ws: lidog_sconv_os_stats_ws doubles; count / eps / momentum / mean / ... as lidog_bn_stats.
static int os_stats(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                    const uint32_t *wave_masks, const int32_t *tile_order, const float *W, const float *bias, int32_t Cin,
                    int32_t Cout, float *out, double *sums, double *ws, double count, float eps, float momentum,
                    float *mean, float *invstd, float *running_mean, float *running_var, InBn in_bn, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(sums && ws, "sconv_os_stats: sums / workspace missing");
    LIDOG_REQUIRE(mean == nullptr || count > 0, "sconv_os_stats: finalising needs the row count");
    if (n == 0) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * Cout + 1), st) == hipSuccess ? 0 : 1;
    OsStats stats = {};
    stats.mode = 1;
    BnFinish fin = {eps, momentum, mean, invstd, running_mean, running_var, nullptr, nullptr};
    if (lidog_stats_tail_make(&stats.tail, ws, sums, count, Cout, fin, st)) return 1;
    // one partial row per tile: past what the two-level tail covers (4 096 tiles = 524 288 rows: six bench scans, or two
    // of the 0.02 m stress scans) the rows are added by bn.hip:k_sums_finish in a launch of its own, which takes any count
    if (os_pad(n) / OS_TM > (int64_t)STATS_MAX_GROUPS * STATS_GROUP) stats.tail.tickets = nullptr;
    int rc = os_launch(A, nbr, n, K, perm, wave_masks, tile_order, W, 0, bias, nullptr, Cin, Cout, out, stats, st, in_bn);
    if (rc) return rc;
    return lidog_stats_tail_finish(stats.tail, (int)(os_pad(n) / OS_TM), st);
}

extern "C" int lidog_sconv_os_stats(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                                    const uint32_t *wave_masks, const int32_t *tile_order, const float *W,
                                    const float *bias, int32_t Cin, int32_t Cout, float *out, double *sums, double *ws,
                                    double count, float eps, float momentum, float *mean, float *invstd,
                                    float *running_mean, float *running_var, void *stream) {
    return os_stats(A, nbr, n, K, perm, wave_masks, tile_order, W, bias, Cin, Cout, out, sums, ws, count, eps, momentum,
                    mean, invstd, running_mean, running_var, InBn{nullptr, nullptr, nullptr, nullptr, 0}, stream);
}

// The same with the BatchNorm (+ ReLU) of the layer BEFORE applied to the input rows as they are gathered: A is that
// layer's raw convolution output, in_* its batch statistics and affine parameters [Cin] -- conv -> BN -> ReLU -> conv
// without the normalised copy in between (= lidog_bn_apply_bits followed by lidog_sconv_os_stats, bit for bit).
extern "C" int lidog_sconv_os_stats_in_bn(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                                          const uint32_t *wave_masks, const int32_t *tile_order, const float *W,
                                          const float *bias, int32_t Cin, int32_t Cout, float *out, double *sums,
                                          double *ws, double count, float eps, float momentum, float *mean, float *invstd,
                                          float *running_mean, float *running_var, const float *in_mean,
                                          const float *in_invstd, const float *in_w, const float *in_b, int32_t in_relu,
                                          void *stream) {
    LIDOG_REQUIRE(in_mean && in_invstd && in_w && in_b, "sconv_os_stats_in_bn: input BatchNorm vectors missing");
    return os_stats(A, nbr, n, K, perm, wave_masks, tile_order, W, bias, Cin, Cout, out, sums, ws, count, eps, momentum,
                    mean, invstd, running_mean, running_var, InBn{in_mean, in_invstd, in_w, in_b, in_relu}, stream);
}

// Forward convolution with an evaluation-mode BatchNorm (+ residual + ReLU) in the epilogue: the validation path
// (= lidog_sconv_gemm + lidog_sconv_reduce_rows_bn; utils/models/minkunet_bev.py:376-393, running statistics).
extern "C" int lidog_sconv_os_bn(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                                 const uint32_t *wave_masks, const int32_t *tile_order, const float *W, const float *bias,
                                 int32_t Cin, int32_t Cout, const float *mean, const float *invstd, const float *w,
                                 const float *b, const float *residual, int32_t relu, float *out, void *stream) {
    if (n == 0) return 0;
    LIDOG_REQUIRE(mean && invstd && w && b, "sconv_os_bn: BatchNorm vectors missing");
    OsStats stats = {};
    stats.mode = 3;
    stats.bn_mean = mean; stats.bn_invstd = invstd; stats.bn_w = w; stats.bn_b = b; stats.bn_res = residual;
    stats.bn_relu = relu;
    return os_launch(A, nbr, n, K, perm, wave_masks, tile_order, W, 0, bias, nullptr, Cin, Cout, out, stats,
                     (hipStream_t)stream);
}
