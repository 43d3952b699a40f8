// Output-stationary sparse convolution (no product rows T, no reduction pass) for stride-1 odd kernels.
// A workgroup owns OS_BR consecutive output rows x up to 128 output columns, kept as an fp32 block in LDS; it walks
// the kernel offsets in order and, per offset, the pairs whose output row lies in its block (contiguous in the rule
// book, which is sorted by output row inside an offset; seg[k][b] = first such pair).  Those pairs are processed
// 32 at a time: gathered input rows -> LDS, one 32 x 32 MFMA tile per wave (wave w owns columns [32w, 32w+32) of
// the block, so nothing it accumulates is touched by another wave), product tile added to the block rows with LDS
// float adds.  Per output element the arithmetic is the same as the two-pass path: an ascending-ci fmaf chain per
// pair (v_mfma_f32_32x32x2_f32), pairs added in ascending offset order starting from +0 -> bit-identical results.
// The price is MFMA rows padded to 32 per (block, offset): 1.22-1.38 x the matrix work on LiDAR kernel maps
// (DESIGN.md section 8); what it saves is writing and re-reading P x Cout floats.
//
// STATUS: correct (bit-identical forward, data gradient and BatchNorm partial sums: tests/test_gpu_ops.py) but NOT
// used by lidog_amd.me: as written it runs at 0.6-0.7 x the speed of the two-pass path (scripts/bench_os.py; 96 -> 96
// at stride 1: 0.77 ms against 0.31 + 0.17 ms).  A 256 x 96 fp32 block is 100 KB of LDS, i.e. one workgroup = one
// wave per SIMD per CU, so index loads, row staging, operand reads, the block update and two barriers per 32-pair
// tile all run in series with the tile's 48 MFMAs (38 % matrix-pipe utilisation; 128-row blocks with two workgroups
// per CU: 0.71 ms).  Measured on the way: LDS float atomics for the block update cost 0.65 ms of the first version's
// 1.5 ms (plain read-add-write: 0.04 ms); branches around loads another 0.12 ms.
//
// The data gradient of a stride-1 odd kernel is the same computation on the same rule book: the map is symmetric
// (pair (i -> j) at offset k  <=>  pair (j -> i) at offset K-1-k), so gx[i] = sum_k gout[.] W_k^T runs over the
// list of offset K-1-k with weight k; `reverse` walks the list offsets downwards so that the sum keeps the
// ascending-k order of the two-pass path.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef OS_BR
#define OS_BR 256
#endif
#define OS_MAXT (27 * (OS_BR / 32) + 27)  // row tiles of one block: every offset adds at most one ragged tile

// seg[k][b] = first pair p of offset k (k_off[k] <= p <= k_off[k+1]) with pair_out[p] >= b * OS_BR
__global__ void k_os_segments(const int32_t *__restrict__ pair_out, const int64_t *__restrict__ k_off, int K, int nb,
                              int32_t *__restrict__ seg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * (nb + 1)) return;
    const int k = idx / (nb + 1), b = idx - k * (nb + 1);
    int64_t lo = k_off[k], hi = k_off[k + 1];
    const int64_t target = (int64_t)b * OS_BR;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (pair_out[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    seg[idx] = (int32_t)lo;
}

extern "C" int lidog_sconv_os_segments(const int32_t *pair_out, const int64_t *k_off, int32_t K, int64_t n_out,
                                       int32_t *seg, void *stream) {
    const int nb = (int)cdiv64(n_out, OS_BR);
    const int total = K * (nb + 1);
    if (total == 0) return 0;
    k_os_segments<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(pair_out, k_off, K, nb, seg);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t lidog_sconv_os_block_rows(void) { return OS_BR; }

// NCH = Cin / 32 (1..4), NW = waves = ceil(CT / 32) column slices
template <int NCH>
__global__ __launch_bounds__(256) void k_sconv_os(const float *__restrict__ A, const int32_t *__restrict__ pair_in,
                                                  const int32_t *__restrict__ pair_out,
                                                  const int32_t *__restrict__ seg, int K, int nb, int64_t n_out,
                                                  const float *__restrict__ W, int reverse, int Cout, int CT,
                                                  float *__restrict__ out, double *__restrict__ partial) {
    constexpr int Cin = 32 * NCH;
    constexpr int SA = Cin + 1;  // odd row stride: the 32 lanes of an A read hit 32 banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CTP = CT + 4;
    float *acc_blk = smem;                         // [OS_BR + 1][CTP]
    float *As = acc_blk + (OS_BR + 1) * CTP;       // [32][SA]; row OS_BR of the block absorbs padding rows
    int *s_idx = reinterpret_cast<int *>(As + 32 * SA);  // [3][64]: src rows / local dst rows of three tiles
    int2 *s_tile = reinterpret_cast<int2 *>(s_idx + 3 * 64);  // [OS_MAXT] (first pair, weight index | rows << 8)
    __shared__ int s_ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int rb = blockIdx.x, col0 = blockIdx.y * CT;
    const int row_base = rb * OS_BR;
    const bool wave_on = 32 * wave < CT;

    // ---- tile list of this block (wave 0: lane kk owns list offset kk, exclusive scan of the tile counts) and
    // zeroed accumulator block
    if (tid < 64) {
        int p0 = 0, p1 = 0;
        if (tid < K) {
            const int kl = reverse ? K - 1 - tid : tid;
            p0 = seg[(size_t)kl * (nb + 1) + rb];
            p1 = seg[(size_t)kl * (nb + 1) + rb + 1];
        }
        const int cnt = (p1 - p0 + 31) >> 5;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (tid >= d) incl += o;
        }
        int n = incl - cnt;
        for (int p = p0; p < p1; p += 32) {
            const int rows = p1 - p < 32 ? p1 - p : 32;
            s_tile[n++] = make_int2(p, tid | (rows << 8));
        }
        if (tid == K - 1) s_ntiles = incl;
    }
    for (int e = tid; e < OS_BR * CTP; e += 256) acc_blk[e] = 0.f;
    __syncthreads();
    const int n_tiles = s_ntiles;

    // indices of a tile: threads 0..31 hold the source row, 32..63 the local destination row (-1 = padding)
    // (every load below is unconditional: a branch around a load costs an exec-mask round trip per load and keeps
    // the scheduler from moving anything across it)
    auto load_idx = [&](int t) -> int {
        const int2 d = s_tile[t < n_tiles ? t : n_tiles - 1];
        const int r = tid & 31, rows = d.y >> 8;
        const int32_t *src = (tid & 32) ? pair_out : pair_in;
        const int v = src[d.x + (r < rows ? r : rows - 1)] - ((tid & 32) ? row_base : 0);
        return (t < n_tiles && r < rows) ? v : -1;
    };
    // the NCH float4 of the gathered rows this thread stages: row tid >> 3, float4 (tid & 7) + 8 j
    float4 ra[NCH];
    auto load_rows = [&](int slot) {
        const int src = s_idx[slot * 64 + (tid >> 3)];
        const float *row = A + (size_t)(src < 0 ? 0 : src) * Cin + (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < NCH; ++j) ra[j] = *reinterpret_cast<const float4 *>(row + 32 * j);
    };
    // B fragments of a whole tile: lane (li, kh) needs W[w][ci = 2 s + kh][col0 + 32 wave + li], s = 0..Cin/2-1
    float bc[16 * NCH], bn[16 * NCH];
    auto load_b = [&](int t, float (&b)[16 * NCH]) {
        const int widx = s_tile[t < n_tiles ? t : n_tiles - 1].y & 255;
        const float *wp = W + ((size_t)widx * Cin + kh) * Cout + col0 + 32 * (wave_on ? wave : 0) + li;
#pragma unroll
        for (int s = 0; s < 16 * NCH; ++s) b[s] = wp[(size_t)(2 * s) * Cout];
    };

    if (n_tiles > 0) {
        // prologue: indices of tiles 0 and 1 in the ring, tile 2's in registers; rows + B of tile 0 in flight
        {
            const int i0 = load_idx(0), i1 = load_idx(1);
            if (tid < 64) {
                s_idx[0 * 64 + tid] = i0;
                s_idx[1 * 64 + tid] = i1;
            }
        }
        __syncthreads();
        load_rows(0);
        load_b(0, bn);
        int idx_ahead = load_idx(2);

        for (int t = 0; t < n_tiles; ++t) {
            const int slot = t % 3;
            __syncthreads();  // every wave is through tile t - 1: its A image and its index slot are free
            if (tid < 64) s_idx[((t + 2) % 3) * 64 + tid] = idx_ahead;  // tile t + 2 -> the slot of tile t - 1
            {
                const int r = tid >> 3, q = tid & 7;
                const bool ok = s_idx[slot * 64 + r] >= 0;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    float *d = &As[r * SA + 32 * j + q * 4];
                    d[0] = ok ? ra[j].x : 0.f;
                    d[1] = ok ? ra[j].y : 0.f;
                    d[2] = ok ? ra[j].z : 0.f;
                    d[3] = ok ? ra[j].w : 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 16 * NCH; ++s) bc[s] = bn[s];
            __syncthreads();
            // in flight during this tile's MFMAs: rows and B fragments of tile t + 1, indices of tile t + 3
#ifndef OS_EXP_NOA
            if (t + 1 < n_tiles) load_rows((t + 1) % 3);
#endif
#ifndef OS_EXP_NOB
            load_b(t + 1, bn);
#endif
            idx_ahead = load_idx(t + 3);

            if (wave_on) {
                // all operands of the tile and its 16 destination rows requested before the first MFMA: with one
                // wave per SIMD nothing else hides an LDS round trip
                float av[16 * NCH];
                int dst[16];
                const float *arow = &As[li * SA + kh];
#pragma unroll
                for (int s = 0; s < 16 * NCH; ++s) av[s] = arow[2 * s];
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[e] = s_idx[slot * 64 + 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#ifndef OS_EXP_NOMFMA
#pragma unroll
                for (int s = 0; s < 16 * NCH; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bc[s], acc, 0, 0, 0);
#else
#pragma unroll
                for (int s = 0; s < 16 * NCH; ++s) acc[s & 15] += av[s] * bc[s];
#endif
                float *blk = acc_blk + 32 * wave + li;
                // the wave owns these columns and the rows of a tile are distinct: plain read-add-write, all reads
                // first (an LDS float atomic costs ~150 cycles per wave instruction here); padding rows go to the
                // spare row OS_BR
                float cur[16];
#ifndef OS_EXP_NORMW
#pragma unroll
                for (int e = 0; e < 16; ++e) cur[e] = blk[(dst[e] < 0 ? OS_BR : dst[e]) * CTP];
#pragma unroll
                for (int e = 0; e < 16; ++e) blk[(dst[e] < 0 ? OS_BR : dst[e]) * CTP] = cur[e] + acc[e];
#else
                if (acc[0] == 1.2345e30f) blk[0] = acc[1];
                (void)cur;
#endif
            }
        }
    }
    __syncthreads();

    // ---- block -> out, BatchNorm partial sums (same layout as k_sconv_reduce4_stats: partial[block][2C])
    const int64_t rows_here = n_out - row_base < OS_BR ? n_out - row_base : OS_BR;
    const int C4 = CT / 4;
    for (int e = tid; e < (int)rows_here * C4; e += 256) {
        const int r = e / C4, c4 = e - r * C4;
        const float4 v = *reinterpret_cast<const float4 *>(&acc_blk[r * CTP + c4 * 4]);
        *reinterpret_cast<float4 *>(&out[(size_t)(row_base + r) * Cout + col0 + c4 * 4]) = v;
    }
    if (partial && wave_on) {
        double s0 = 0, s1 = 0;
        for (int r = kh; r < (int)rows_here; r += 2) {
            const float v = acc_blk[r * CTP + 32 * wave + li];
            s0 += (double)v;
            s1 += (double)v * (double)v;
        }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (kh == 0) {
            double *dst = partial + (size_t)rb * 2 * Cout;
            dst[col0 + 32 * wave + li] = s0;
            dst[Cout + col0 + 32 * wave + li] = s1;
        }
    }
}

// out [n_out, Cout] = sum over offsets of A[pair_in] . W[k]   (forward, reverse = 0), or the data gradient with
// A = gout, W = transposed weights [K][Cout_fwd][Cin_fwd] and reverse = 1 (see the header of this file).
// seg: lidog_sconv_os_segments.  partial (may be NULL): ceil(n_out / 256) x 2*Cout doubles of BatchNorm partial sums.
extern "C" int lidog_sconv_os(const float *A, const int32_t *pair_in, const int32_t *pair_out, const int32_t *seg,
                              int32_t K, int64_t n_out, const float *W, int32_t reverse, int32_t Cin, int32_t Cout,
                              float *out, double *partial, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin % 32 == 0 && Cin >= 32 && Cin <= 128, "sconv_os: Cin must be 32, 64, 96 or 128");
    LIDOG_REQUIRE(Cout % 32 == 0 && Cout >= 32, "sconv_os: Cout must be a multiple of 32");
    LIDOG_REQUIRE(K >= 1 && K <= 27, "sconv_os: at most 27 offsets");
    if (n_out == 0) return 0;
    int CT = Cout <= 128 ? Cout : (Cout % 128 == 0 ? 128 : (Cout % 96 == 0 ? 96 : (Cout % 64 == 0 ? 64 : 32)));
    const int nb = (int)cdiv64(n_out, OS_BR);
    const size_t lds = sizeof(float) * ((size_t)(OS_BR + 1) * (CT + 4) + 32 * (Cin + 1)) + sizeof(int) * 3 * 64 +
                       sizeof(int2) * OS_MAXT;
    LIDOG_REQUIRE(lds <= 160 * 1024 - 64, "sconv_os: block does not fit the LDS (%zu bytes)", lds);
    dim3 grid((unsigned)nb, (unsigned)(Cout / CT));
#define OS_LAUNCH(NCH_)                                                                                          \
    do {                                                                                                         \
        static bool once = false;                                                                                \
        if (!once) {                                                                                             \
            LIDOG_CHECK_HIP(hipFuncSetAttribute((const void *)k_sconv_os<NCH_>,                                  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));   \
            once = true;                                                                                         \
        }                                                                                                        \
        k_sconv_os<NCH_><<<grid, 256, lds, st>>>(A, pair_in, pair_out, seg, K, nb, n_out, W, reverse, Cout, CT,  \
                                                 out, partial);                                                  \
    } while (0)
    switch (Cin / 32) {
        case 1: OS_LAUNCH(1); break;
        case 2: OS_LAUNCH(2); break;
        case 3: OS_LAUNCH(3); break;
        default: OS_LAUNCH(4);
    }
#undef OS_LAUNCH
    LIDOG_LAUNCH_CHECK();
    return 0;
}
