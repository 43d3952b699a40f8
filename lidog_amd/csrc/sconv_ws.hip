// Gathered GEMM with specialised waves (product-row path of the sparse convolutions: gather != NULL, no scatter, no bias).
//
// k_sconv_gemm_mfma (sconv_mfma.hip) has every wave do everything: fetch its share of the chunk, wait, write it to LDS,
// barrier, multiply, barrier.  Its time per tile fits a + b * chunks with a ~ b: one chunk's worth of per-tile latency
// (descriptor -> gather index -> first rows) that the other workgroups of the CU only partly cover, and two barriers
// per chunk that couple the four SIMDs.  Here a workgroup has 8 waves: 4 CONSUMERS (one per SIMD) that only read LDS
// operands, issue MFMAs and store finished product rows, and 4 PRODUCERS that only move data: global -> registers two
// chunks ahead (two register sets), registers -> LDS into the stage the consumers are not reading.  One barrier per chunk; the producers
// walk the workgroup's flat (tile, chunk) sequence, so the first chunk of the next tile is in flight while the last
// chunk of the current one is multiplied.  Workgroups are persistent (grid = 2 per CU), tiles dealt round-robin.
//
// Same arithmetic as k_sconv_gemm_mfma: per product row an fmaf chain over ascending input channel (v_mfma_f32_32x32x2
// with the k steps in order), so T is bit-identical.
#include <type_traits>

#include "common.h"
#include "sconv_mfma.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WS_TM 128
#define WS_BK 32
#define WS_SA 33   // odd row stride of the A image: the 32 lanes of an operand read hit 32 banks

namespace {

template <int NT>
__device__ __forceinline__ void ws_frag_load(const float *p, float (&f)[NT]) {
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

template <int NT>
__device__ __forceinline__ void ws_frag_store(float *p, const float (&f)[NT]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    if constexpr (NT == 4) {
        v4f v = {f[0], f[1], f[2], f[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
    } else if constexpr (NT == 2) {
        v2f v = {f[0], f[1]};
        __builtin_nontemporal_store(v, reinterpret_cast<v2f *>(p));
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) __builtin_nontemporal_store(f[t], p + t);
    }
}

template <int NT>
__global__ __launch_bounds__(512, 4) void k_sconv_gemm_ws(const float *__restrict__ A, const int32_t *__restrict__ gather,
                                                       const float *__restrict__ B,
                                                       const int32_t *__restrict__ tile_k,
                                                       const int32_t *__restrict__ tile_row0,
                                                       const int32_t *__restrict__ tile_rows, int n_tiles, int Cin,
                                                       int Cout, float *__restrict__ T) {
    constexpr int TN = 32 * NT;
    constexpr int BQ = (WS_BK * TN / 4) / 256;   // float4 of B per producer lane per chunk (1..4)
    __shared__ float As[2][WS_TM * WS_SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][WS_BK * TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int pw = wave & 3;                       // index inside the role
    const int ncol = Cout / TN, nchunk = Cin / WS_BK;
    // work items of this workgroup: w = blockIdx.x, blockIdx.x + gridDim.x, ...; item w = (tile w / ncol, column w % ncol)
    const int n_work = n_tiles * ncol;
    const int my_items = (n_work - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_items > 0 ? my_items * nchunk : 0;

    // ---- producer state: the 4 (row, 16-byte piece) pairs of this lane in the 32 rows of its wave, the B pieces
    const float *a_row[4] = {A, A, A, A};
    bool a_cur[4] = {false, false, false, false};      // validity of the rows the pointers stand for
    bool a_ok[2][4] = {{false, false, false, false}, {false, false, false, false}};   // ... of the rows in a register set
    float4 ra[2][4], rb[2][4];
    int32_t nsrc[4] = {0, 0, 0, 0};
    const float *Bk = B;
    // ---- consumer state
    f32x16 acc[NT];
    const int li = lane & 31, kh = lane >> 5;

    // descriptors of item index i (wave-uniform)
    auto item_of = [&](int i, int &k, int &row0, int &rows, int &col0) {
        int w = (int)blockIdx.x + i * (int)gridDim.x;
        int t = w / ncol;
        k = tile_k[t];
        row0 = tile_row0[t];
        rows = tile_rows[t];
        col0 = (w % ncol) * TN;
    };
    auto load_idx = [&](int i) {   // gather indices of item i for this producer lane
        int k, row0, rows, col0;
        item_of(i, k, row0, rows, col0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = pw * 32 + ((lane + 64 * j) >> 3);
            nsrc[j] = r < rows ? gather[row0 + r] : -1;
        }
    };
    auto set_rows = [&](int i) {   // pointers of item i from the indices fetched by load_idx(i)
        int k, row0, rows, col0;
        item_of(i, k, row0, rows, col0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_cur[j] = nsrc[j] >= 0;
            a_row[j] = A + (size_t)(nsrc[j] < 0 ? 0 : nsrc[j]) * Cin + ((lane + 64 * j) & 7) * 4;
        }
        Bk = B + (size_t)k * Cin * Cout + col0;
    };
    // global -> register set SET (unconditional loads: rows past the tile read row 0 and are zeroed at the LDS store)
    auto load_chunk = [&](auto SET, int c) {
        constexpr int S = decltype(SET)::value;
        const int kb = c * WS_BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[S][j] = *reinterpret_cast<const float4 *>(a_row[j] + kb);
            a_ok[S][j] = a_cur[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < BQ) {
                int f = pw * 64 * BQ + lane + 64 * j;      // this wave's quarter of the chunk's float4s
                int kk = f / (TN / 4), c4 = f % (TN / 4);
                rb[S][j] = *reinterpret_cast<const float4 *>(Bk + (size_t)(kb + kk) * Cout + c4 * 4);
            }
    };
    auto store_chunk = [&](auto SET, int s) {  // register set SET -> LDS stage s
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = lane + 64 * j;
            int r = pw * 32 + (f >> 3), q = f & 7;
            const int o = r * WS_SA + q * 4;
            const bool ok = a_ok[S][j];
            As[s][o] = ok ? ra[S][j].x : 0.f;
            As[s][o + 1] = ok ? ra[S][j].y : 0.f;
            As[s][o + 2] = ok ? ra[S][j].z : 0.f;
            As[s][o + 3] = ok ? ra[S][j].w : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < BQ) {
                int f = pw * 64 * BQ + lane + 64 * j;
                *reinterpret_cast<float4 *>(&Bs[s][f * 4]) = rb[S][j];
            }
    };
    // fetch chunk f of the flat sequence into register set SET; the indices of an item are requested while the last
    // chunk of the item before it is fetched, i.e. at least one iteration before set_rows() needs them
    auto fetch = [&](auto SET, int f) {
        if (f >= total) return;
        const int it = f / nchunk, c = f % nchunk;
        if (c == 0) set_rows(it);
        load_chunk(SET, c);
        if (c == nchunk - 1 && it + 1 < my_items) load_idx(it + 1);
    };

    // Iteration j (j = 0 .. total): producers write chunk j into stage j & 1 (fetched during iteration j - 1) and fetch
    // chunk j + 1; consumers multiply chunk j - 1 from stage (j - 1) & 1; one barrier closes the iteration.  The stage
    // written in iteration j was last read in iteration j - 1.  The two roles run separate loops (their registers are
    // then allocated separately: accumulators here, staging registers there) with the same number of barriers.
    if (producer) {
        std::integral_constant<int, 0> S0;
        std::integral_constant<int, 1> S1;
        // two chunks ahead: chunk j + 2 is requested in iteration j, into the register set chunk j has just left
        if (total > 0) {
            load_idx(0);
            fetch(S0, 0);
            fetch(S1, 1);
        }
        auto step = [&](auto SET, int j) {
            if (j < total) {
                store_chunk(SET, j & 1);     // waits for chunk j's loads (requested two iterations ago)
                fetch(SET, j + 2);
            }
        };
        int j = 0;
        for (; j + 1 <= total; j += 2) {
            step(S0, j);
            __syncthreads();
            step(S1, j + 1);
            __syncthreads();
        }
        if (j <= total) {
            step(S0, j);
            __syncthreads();
        }
        return;
    }
    __syncthreads();   // iteration 0: nothing to multiply yet
    for (int jc = 0; jc < total; ++jc) {
        const int it = jc / nchunk, c = jc % nchunk, s = jc & 1;
        if (c == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        }
        const float *arow = &As[s][(pw * 32 + li) * WS_SA + kh];
        const float *bcol = &Bs[s][kh * TN + li * NT];
        float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0 = 0.f, an1 = 0.f;
        ws_frag_load<NT>(bcol, bq0);
        ws_frag_load<NT>(bcol + 2 * TN, bq1);
        a0 = arow[0];
        a1 = arow[2];
#pragma unroll
        for (int r = 0; r < WS_BK / 4; ++r) {
            if (r + 1 < WS_BK / 4) {
                ws_frag_load<NT>(bcol + (4 * r + 4) * TN, bn0);
                ws_frag_load<NT>(bcol + (4 * r + 6) * TN, bn1);
                an0 = arow[4 * r + 4];
                an1 = arow[4 * r + 6];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < WS_BK / 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bq0[t] = bn0[t];
                    bq1[t] = bn1[t];
                }
                a0 = an0;
                a1 = an1;
            }
        }
        if (c == nchunk - 1) {     // the tile's product rows
            int k, row0, rows, col0;
            item_of(it, k, row0, rows, col0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int r = pw * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (r < rows) {
                    float *out = T + (size_t)(row0 + r) * Cout + col0 + li * NT;
                    float v[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = acc[t][e];
                    ws_frag_store<NT>(out, v);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

static int g_ws_cus = 0;

// product-row path only: gather != NULL; T rows row0 .. row0 + rows - 1 of every tile
int lidog_launch_gemm_ws(const float *A, const int32_t *gather, const float *B, const int32_t *tile_k,
                         const int32_t *tile_row0, const int32_t *tile_rows, int n_tiles, int Cin, int Cout, float *T,
                         hipStream_t st) {
    if (g_ws_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        g_ws_cus = cus;
    }
    int nt = (Cout % 128 == 0) ? 4 : (Cout % 96 == 0) ? 3 : (Cout % 64 == 0) ? 2 : 1;
    int n_work = n_tiles * (Cout / (32 * nt));
    int grid = 2 * g_ws_cus;
    if (grid > n_work) grid = n_work;
#define LAUNCH(NT_) \
    k_sconv_gemm_ws<NT_><<<grid, 512, 0, st>>>(A, gather, B, tile_k, tile_row0, tile_rows, n_tiles, Cin, Cout, T)
    switch (nt) {
        case 4: LAUNCH(4); break;
        case 3: LAUNCH(3); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(1);
    }
#undef LAUNCH
    return 0;
}
