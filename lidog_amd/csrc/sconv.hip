// Sparse convolution on the rule book: gathered GEMM (fp32 vector FMA, LDS-tiled for 64-wide
// waves) -> per-output-row reduction in ascending kernel-offset order (no atomics, bit-reproducible),
// weight gradient by split reduction over pairs.  CPU restatement: oracle/me_oracle.c
// (orc_conv_fwd / orc_conv_bwd_data / orc_conv_bwd_weight).
#include "common.h"
#include "stats_tail.h"
#include "sconv_mfma.h"

// 1 = exact-f32 MFMA cores (default), 0 = vector-FMA cores; identical results, see sconv_mfma.hip
static int g_sparse_core = 1;
extern "C" int lidog_set_sparse_core(int32_t core) {
    LIDOG_REQUIRE(core == 0 || core == 1, "sparse core must be 0 (vector FMA) or 1 (MFMA f32)");
    g_sparse_core = core;
    return 0;
}
extern "C" int lidog_get_sparse_core(void) { return g_sparse_core; }

// ------------------------------------------------------------------ gathered GEMM
// Tile: 128 pair-rows x (16*CN) columns, BK = 32 input channels per step, 256 threads as 16 (ty) x 16 (tx).
// Thread (ty,tx) owns rows ty + 16*i (i<8) and columns tx*CN .. tx*CN+CN-1.
// A tile in LDS row-major with stride 36 floats: the four ty of a wave read rows 36 floats apart ->
// distinct banks for ds_read_b128; the 16 tx of one ty read the same address (broadcast).
#define GM_TM 128
#define GM_BK 32
#define GM_SA 36

template <int CN>
__global__ __launch_bounds__(256) void k_sconv_gemm(const float *__restrict__ A, const int32_t *__restrict__ gather,
                                                    const float *__restrict__ B, const float *__restrict__ bias,
                                                    const int32_t *__restrict__ tile_k,
                                                    const int32_t *__restrict__ tile_row0,
                                                    const int32_t *__restrict__ tile_rows, int Cin, int Cout,
                                                    float *__restrict__ T, const int32_t *__restrict__ scatter) {
    constexpr int TN = 16 * CN;
    constexpr int BV = (GM_BK * TN / 4) / 256 > 0 ? (GM_BK * TN / 4) / 256 : 1;  // float4 of B per thread
    __shared__ __attribute__((aligned(16))) float As[GM_TM * GM_SA];
    __shared__ __attribute__((aligned(16))) float Bs[GM_BK * TN];
    __shared__ int32_t s_src[GM_TM];

    const int tile = blockIdx.x;
    const int k = tile_k[tile];
    const int row0 = tile_row0[tile];
    const int rows = tile_rows[tile];
    const int col0 = blockIdx.y * TN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

    if (tid < GM_TM) {
        int r = tid;
        int src = -1;
        if (r < rows) src = gather ? gather[row0 + r] : (row0 + r);
        s_src[r] = src;
    }
    __syncthreads();

    const float *Bk = B + (size_t)k * Cin * Cout + col0;

    // global -> register staging: A: 128 rows x 8 float4 = 1024 float4 -> 4 per thread (8 lanes per row)
    float4 ra[4];
    float4 rb[BV];
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            int r = f >> 3, q = f & 7;
            int src = s_src[r];
            // unconditional load + select (a branch around the load would serialise the gather)
            float4 v = *reinterpret_cast<const float4 *>(A + (size_t)(src < 0 ? 0 : src) * Cin + kb + q * 4);
            bool ok = src >= 0;
            ra[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            int f = tid + 256 * j;
            if (f < GM_BK * TN / 4) {
                int kk = f / (TN / 4), c4 = f % (TN / 4);
                rb[j] = *reinterpret_cast<const float4 *>(Bk + (size_t)(kb + kk) * Cout + c4 * 4);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            int r = f >> 3, q = f & 7;
            *reinterpret_cast<float4 *>(&As[r * GM_SA + q * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            int f = tid + 256 * j;
            if (f < GM_BK * TN / 4) *reinterpret_cast<float4 *>(&Bs[f * 4]) = rb[j];
        }
    };

    float acc[8][CN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = 0.f;

    load_chunk(0);
    for (int kb = 0; kb < Cin; kb += GM_BK) {
        __syncthreads();  // previous chunk fully consumed
        store_chunk();
        __syncthreads();
        if (kb + GM_BK < Cin) load_chunk(kb + GM_BK);  // overlaps with the FMAs below
#pragma unroll
        for (int k4 = 0; k4 < GM_BK; k4 += 4) {
            float4 a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const float4 *>(&As[(ty + 16 * i) * GM_SA + k4]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // column ownership is interleaved: vector v of this lane sits at column v*16*V + tx*V, so the
                // 16 lanes of one row group read 16*V contiguous floats (no LDS bank conflict)
                float b[CN];
                constexpr int V = (CN % 4 == 0) ? 4 : 2;
                if constexpr (V == 4) {
#pragma unroll
                    for (int j = 0; j < CN; j += 4) {
                        float4 v = *reinterpret_cast<const float4 *>(&Bs[(k4 + s) * TN + j * 16 + tx * 4]);
                        b[j] = v.x; b[j + 1] = v.y; b[j + 2] = v.z; b[j + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CN; j += 2) {
                        float2 v = *reinterpret_cast<const float2 *>(&Bs[(k4 + s) * TN + j * 16 + tx * 2]);
                        b[j] = v.x; b[j + 1] = v.y;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float av = (s == 0) ? a[i].x : (s == 1) ? a[i].y : (s == 2) ? a[i].z : a[i].w;
#pragma unroll
                    for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_fmaf(av, b[j], acc[i][j]);
                }
            }
        }
    }

    // epilogue: per (row, vector) one coalesced segment of 16 lanes
    constexpr int V = (CN % 4 == 0) ? 4 : 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int r = ty + 16 * i;
        if (r < rows) {
            size_t dst = scatter ? (size_t)scatter[row0 + r] : (size_t)(row0 + r);
            float *out = T + dst * Cout + col0;
#pragma unroll
            for (int j = 0; j < CN; j += V) {
                int c = j * 16 + tx * V;
                if constexpr (V == 4) {
                    float4 v = make_float4(acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3]);
                    if (bias) {
                        v.x += bias[col0 + c]; v.y += bias[col0 + c + 1]; v.z += bias[col0 + c + 2]; v.w += bias[col0 + c + 3];
                    }
                    *reinterpret_cast<float4 *>(out + c) = v;
                } else {
                    float2 v = make_float2(acc[i][j], acc[i][j + 1]);
                    if (bias) { v.x += bias[col0 + c]; v.y += bias[col0 + c + 1]; }
                    *reinterpret_cast<float2 *>(out + c) = v;
                }
            }
        }
    }
}

// any shape: one thread per (row, column); used for Cin=1 (stem) and Cout=7 (final)
__global__ __launch_bounds__(256) void k_sconv_gemm_small(const float *__restrict__ A,
                                                          const int32_t *__restrict__ gather,
                                                          const float *__restrict__ B,
                                                          const float *__restrict__ bias,
                                                          const int32_t *__restrict__ tile_k,
                                                          const int32_t *__restrict__ tile_row0,
                                                          const int32_t *__restrict__ tile_rows, int Cin, int Cout,
                                                          float *__restrict__ T,
                                                          const int32_t *__restrict__ scatter) {
    const int tile = blockIdx.x;
    const int k = tile_k[tile], row0 = tile_row0[tile], rows = tile_rows[tile];
    const float *Bk = B + (size_t)k * Cin * Cout;
    for (int e = threadIdx.x; e < rows * Cout; e += 256) {
        int r = e / Cout, c = e % Cout;
        int src = gather ? gather[row0 + r] : (row0 + r);
        const float *x = A + (size_t)src * Cin;
        float t = 0.f;
        for (int ci = 0; ci < Cin; ++ci) t = __builtin_fmaf(x[ci], Bk[(size_t)ci * Cout + c], t);
        if (bias) t += bias[c];
        size_t dst = scatter ? (size_t)scatter[row0 + r] : (size_t)(row0 + r);
        T[dst * Cout + c] = t;
    }
}

// Cout <= 8, Cin % 4 == 0 (the classifier, 96 -> 7): thread (row, half) reads its pair row as float4 and keeps
// every other output column; the weights of the tile's offset are broadcast from LDS.  Same ascending-ci fmaf chain
// per output as the other cores (bit-identical), without 96 dependent scalar loads per output.
__global__ __launch_bounds__(256) void k_sconv_gemm_cout8(const float *__restrict__ A,
                                                          const int32_t *__restrict__ gather,
                                                          const float *__restrict__ B,
                                                          const float *__restrict__ bias,
                                                          const int32_t *__restrict__ tile_k,
                                                          const int32_t *__restrict__ tile_row0,
                                                          const int32_t *__restrict__ tile_rows, int Cin, int Cout,
                                                          float *__restrict__ T,
                                                          const int32_t *__restrict__ scatter) {
    extern __shared__ float s_w[];  // [Cin][8]
    const int tile = blockIdx.x;
    const int k = tile_k[tile], row0 = tile_row0[tile], rows = tile_rows[tile];
    const float *Bk = B + (size_t)k * Cin * Cout;
    for (int e = threadIdx.x; e < Cin * 8; e += 256) {
        int ci = e >> 3, c = e & 7;
        s_w[e] = c < Cout ? Bk[(size_t)ci * Cout + c] : 0.f;
    }
    __syncthreads();
    const int r = threadIdx.x & 127, h = threadIdx.x >> 7;
    if (r >= rows) return;
    const int src = gather ? gather[row0 + r] : (row0 + r);
    const float4 *x = reinterpret_cast<const float4 *>(A + (size_t)src * Cin);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // columns h, h + 2, h + 4, h + 6
    for (int q = 0; q < Cin / 4; ++q) {
        const float4 v = x[q];
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *w = &s_w[(q * 4 + u) * 8 + h];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(xs[u], w[2 * j], acc[j]);
        }
    }
    const size_t dst = scatter ? (size_t)scatter[row0 + r] : (size_t)(row0 + r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = h + 2 * j;
        if (c < Cout) T[dst * Cout + c] = acc[j] + (bias ? bias[c] : 0.f);
    }
}

// Cin <= 8, Cout % 4 == 0 (the classifier's data gradient, 7 -> 96): thread (row, column quad); the row's <= 8
// inputs and the weights (LDS) give four outputs by the same ascending-ci fmaf chain as the other cores
__global__ __launch_bounds__(256) void k_sconv_gemm_cin8(const float *__restrict__ A,
                                                         const int32_t *__restrict__ gather,
                                                         const float *__restrict__ B,
                                                         const float *__restrict__ bias,
                                                         const int32_t *__restrict__ tile_k,
                                                         const int32_t *__restrict__ tile_row0,
                                                         const int32_t *__restrict__ tile_rows, int Cin, int Cout,
                                                         float *T,
                                                         const int32_t *__restrict__ scatter,
                                                         const float *addend) {
    // (T and addend are NOT restrict: the executor adds the classifier's data gradient onto the rows that already hold the
    // BEV head's gradient in place, addend == T; every thread reads its float4 of the addend before it writes it)
    extern __shared__ float s_w[];  // [Cin][Cout]
    const int tile = blockIdx.x;
    const int k = tile_k[tile], row0 = tile_row0[tile], rows = tile_rows[tile];
    const float *Bk = B + (size_t)k * Cin * Cout;
    for (int e = threadIdx.x; e < Cin * Cout; e += 256) s_w[e] = Bk[e];
    __syncthreads();
    const int C4 = Cout / 4;
    for (int e = threadIdx.x; e < rows * C4; e += 256) {
        const int r = e / C4, c4 = e - r * C4;
        const int src = gather ? gather[row0 + r] : (row0 + r);
        const float *x = A + (size_t)src * Cin;
        const size_t dst = scatter ? (size_t)scatter[row0 + r] : (size_t)(row0 + r);
        // `addend + product`, the operand order of lidog_add(addend, product): the same bits as the separate pass
        float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
        if (addend) ad = *reinterpret_cast<const float4 *>(&addend[dst * Cout + c4 * 4]);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = x[ci];
            const float4 w = *reinterpret_cast<const float4 *>(&s_w[ci * Cout + c4 * 4]);
            t.x = __builtin_fmaf(xv, w.x, t.x);
            t.y = __builtin_fmaf(xv, w.y, t.y);
            t.z = __builtin_fmaf(xv, w.z, t.z);
            t.w = __builtin_fmaf(xv, w.w, t.w);
        }
        if (bias) {
            t.x += bias[c4 * 4]; t.y += bias[c4 * 4 + 1]; t.z += bias[c4 * 4 + 2]; t.w += bias[c4 * 4 + 3];
        }
        if (addend) {
            t.x = ad.x + t.x; t.y = ad.y + t.y; t.z = ad.z + t.z; t.w = ad.w + t.w;
        }
        *reinterpret_cast<float4 *>(&T[dst * Cout + c4 * 4]) = t;
    }
}

// T = addend + (the gathered GEMM's product), for the narrow-input case (Cin <= 8, Cout % 4 == 0: the classifier's data
// gradient 7 -> 96, whose result is added to the gradient the BEV head left on the same rows): one pass instead of a
// product pass and an add pass; the same bits as lidog_sconv_gemm followed by lidog_add(addend, product).  Returns 3
// (without touching anything) for shapes it does not cover.
extern "C" int lidog_sconv_gemm_addend(const float *A, const int32_t *gather, const float *B, const int32_t *tile_k,
                                       const int32_t *tile_row0, const int32_t *tile_rows, int32_t n_tiles, int32_t Cin,
                                       int32_t Cout, const float *addend, float *T, const int32_t *scatter, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!(Cin >= 1 && Cin <= 8 && Cout % 4 == 0 && Cin * Cout * 4 <= 32 * 1024)) return 3;
    if (n_tiles == 0) return 0;
    LIDOG_REQUIRE(A && B && T && addend && tile_k && tile_row0 && tile_rows, "sconv_gemm_addend: null argument");
    k_sconv_gemm_cin8<<<dim3((unsigned)n_tiles), 256, (size_t)Cin * Cout * sizeof(float), st>>>(
        A, gather, B, nullptr, tile_k, tile_row0, tile_rows, Cin, Cout, T, scatter, addend);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// The gathered GEMM with the BatchNorm (+ ReLU) of the layer BEFORE applied to the rows as they are gathered: A is that
// layer's raw convolution output, in_* its batch statistics and affine parameters [Cin] (= lidog_bn_apply_bits followed by
// lidog_sconv_gemm, bit for bit).  Matrix-core kernels only: Cin, Cout multiples of 32, lidog_set_sparse_core(1).
extern "C" int lidog_sconv_gemm_in_bn(const float *A, const int32_t *gather, const float *B, const float *bias,
                                      const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows,
                                      int32_t n_tiles, int32_t Cin, int32_t Cout, float *T, const int32_t *scatter,
                                      const float *in_mean, const float *in_invstd, const float *in_w, const float *in_b,
                                      int32_t in_relu, int64_t a_rows, void *stream) {
    if (n_tiles == 0) return 0;
    LIDOG_REQUIRE(in_mean && in_invstd && in_w && in_b, "sconv_gemm_in_bn: input BatchNorm vectors missing");
    LIDOG_REQUIRE(g_sparse_core == 1 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0,
                  "sconv_gemm_in_bn: matrix-core kernels only (channel counts multiples of 32; got %d -> %d)", Cin, Cout);
    lidog_launch_gemm_mfma(A, gather, B, bias, tile_k, tile_row0, tile_rows, n_tiles, Cin, Cout, T, scatter,
                           InBn{in_mean, in_invstd, in_w, in_b, in_relu}, a_rows, (hipStream_t)stream);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_sconv_gemm(const float *A, const int32_t *gather, const float *B, const float *bias,
                                const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows,
                                int32_t n_tiles, int32_t Cin, int32_t Cout, float *T, const int32_t *scatter,
                                int64_t a_rows, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_tiles == 0) return 0;
    LIDOG_REQUIRE(Cin > 0 && Cout > 0, "sconv_gemm: bad channel counts %d %d", Cin, Cout);
    if (g_sparse_core == 1 && Cin % 32 == 0 && Cout % 32 == 0) {
        lidog_launch_gemm_mfma(A, gather, B, bias, tile_k, tile_row0, tile_rows, n_tiles, Cin, Cout, T, scatter,
                               InBn{nullptr, nullptr, nullptr, nullptr, 0}, a_rows, st);
        LIDOG_LAUNCH_CHECK();
        return 0;
    }
    int cn = 0;
    if (Cin % GM_BK == 0) {
        if (Cout % 128 == 0) cn = 8;
        else if (Cout % 96 == 0) cn = 6;
        else if (Cout % 64 == 0) cn = 4;
        else if (Cout % 32 == 0) cn = 2;
    }
#define LAUNCH_GEMM(CN_)                                                                                      \
    k_sconv_gemm<CN_><<<dim3((unsigned)n_tiles, (unsigned)(Cout / (16 * CN_))), 256, 0, st>>>(               \
        A, gather, B, bias, tile_k, tile_row0, tile_rows, Cin, Cout, T, scatter)
    switch (cn) {
        case 8: LAUNCH_GEMM(8); break;
        case 6: LAUNCH_GEMM(6); break;
        case 4: LAUNCH_GEMM(4); break;
        case 2: LAUNCH_GEMM(2); break;
        default:
            if (Cin <= 8 && Cout % 4 == 0 && Cin * Cout * 4 <= 32 * 1024)
                k_sconv_gemm_cin8<<<dim3((unsigned)n_tiles), 256, (size_t)Cin * Cout * sizeof(float), st>>>(
                    A, gather, B, bias, tile_k, tile_row0, tile_rows, Cin, Cout, T, scatter, nullptr);
            else if (Cout <= 8 && Cin % 4 == 0 && Cin <= 1024)
                k_sconv_gemm_cout8<<<dim3((unsigned)n_tiles), 256, (size_t)Cin * 8 * sizeof(float), st>>>(
                    A, gather, B, bias, tile_k, tile_row0, tile_rows, Cin, Cout, T, scatter);
            else
                k_sconv_gemm_small<<<dim3((unsigned)n_tiles), 256, 0, st>>>(A, gather, B, bias, tile_k, tile_row0,
                                                                            tile_rows, Cin, Cout, T, scatter);
    }
#undef LAUNCH_GEMM
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ per-row reduction over kernel offsets
// out row = sum over k ascending of T[pos[k][o]] (missing neighbours add an exact 0).  Loads are issued in
// independent batches of 9 / 4 (position first, then product rows, unconditionally -- a missing neighbour
// reads row 0, which stays in L1): a branch per offset would expose one full memory round trip per offset.
template <int G>
__device__ __forceinline__ void reduce_group(const float4 *__restrict__ T, const int32_t *__restrict__ pos, int64_t n,
                                             int C4, int64_t o, int c4, int k, float4 &acc) {
    int p[G];
    float4 t[G];
#pragma unroll
    for (int j = 0; j < G; ++j) p[j] = pos[(int64_t)(k + j) * n + o];
#pragma unroll
    for (int j = 0; j < G; ++j) t[j] = T[(int64_t)(p[j] < 0 ? 0 : p[j]) * C4 + c4];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        bool ok = p[j] >= 0;
        acc.x += ok ? t[j].x : 0.f;
        acc.y += ok ? t[j].y : 0.f;
        acc.z += ok ? t[j].z : 0.f;
        acc.w += ok ? t[j].w : 0.f;
    }
}

__device__ __forceinline__ float4 reduce_row(const float4 *__restrict__ T, const int32_t *__restrict__ pos, int64_t n,
                                             int K, int C4, int64_t o, int c4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 9 <= K; k += 9) reduce_group<9>(T, pos, n, C4, o, c4, k, acc);
    for (; k + 4 <= K; k += 4) reduce_group<4>(T, pos, n, C4, o, c4, k, acc);
    for (; k < K; ++k) reduce_group<1>(T, pos, n, C4, o, c4, k, acc);
    return acc;
}

__global__ __launch_bounds__(256) void k_sconv_reduce4(const float4 *__restrict__ T, const int32_t *__restrict__ pos,
                                                       int64_t n, int K, int C4, const float4 *__restrict__ bias,
                                                       const float4 *__restrict__ addend,
                                                       float4 *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C4) return;
    int64_t o = idx / C4;
    int c4 = (int)(idx % C4);
    float4 acc = reduce_row(T, pos, n, K, C4, o, c4);
    if (bias) {
        float4 b = bias[c4];
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    if (addend) {  // a second gradient of the same tensor (the residual branch): the add autograd would launch
        float4 a = addend[idx];
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    out[idx] = acc;
}

__global__ __launch_bounds__(256) void k_sconv_reduce1(const float *__restrict__ T, const int32_t *__restrict__ pos,
                                                       int64_t n, int K, int C, const float *__restrict__ bias,
                                                       float *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C) return;
    int64_t o = idx / C;
    int c = (int)(idx % C);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        int p = pos[(int64_t)k * n + o];
        if (p >= 0) acc += T[(int64_t)p * C + c];
    }
    if (bias) acc += bias[c];
    out[idx] = acc;
}

extern "C" int lidog_sconv_reduce(const float *T, const int32_t *pos, int64_t n, int32_t K, int32_t C,
                                  const float *bias, const float *addend, float *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    LIDOG_REQUIRE(addend == nullptr || C % 4 == 0, "sconv_reduce: an addend needs C % 4 == 0");
    if (C % 4 == 0) {
        int C4 = C / 4;
        k_sconv_reduce4<<<(unsigned)cdiv64(n * C4, 256), 256, 0, st>>>((const float4 *)T, pos, n, K, C4,
                                                                       (const float4 *)bias, (const float4 *)addend,
                                                                       (float4 *)out);
    } else {
        k_sconv_reduce1<<<(unsigned)cdiv64(n * C, 256), 256, 0, st>>>(T, pos, n, K, C, bias, out);
    }
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// Same reduction with the BatchNorm statistics of the result folded in: every workgroup also emits its partial
// (sum x, sum x^2) per channel in fp64 to partial[block][2C]; bn.hip:k_sums_finish adds the partials in block order
// (no atomics: the statistics are bit-reproducible) -- saves the separate statistics pass over `out`.
__global__ __launch_bounds__(256) void k_sconv_reduce4_stats(const float4 *__restrict__ T,
                                                             const int32_t *__restrict__ pos, int64_t n, int K, int C4,
                                                             const float4 *__restrict__ bias, float4 *__restrict__ out,
                                                             StatsTail tail) {
    __shared__ double red[256 * 8];
    const int RB = 256 / C4;
    const int tid = threadIdx.x;
    const int r = tid / C4, c4 = tid % C4;
    const bool active = r < RB;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        for (int64_t o = (int64_t)blockIdx.x * RB + r; o < n; o += (int64_t)gridDim.x * RB) {
            float4 acc = reduce_row(T, pos, n, K, C4, o, c4);
            if (bias) {
                float4 b = bias[c4];
                acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            }
            out[o * C4 + c4] = acc;
            a[0] += acc.x; a[1] += acc.y; a[2] += acc.z; a[3] += acc.w;
            a[4] += (double)acc.x * acc.x; a[5] += (double)acc.y * acc.y;
            a[6] += (double)acc.z * acc.z; a[7] += (double)acc.w * acc.w;
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
    lidog_stats_tail(tail, active && r == 0, c4, a);   // partial row + in-kernel finish by the last workgroup
}

// ---- the same reductions over per-row lists (lidog_kernel_map_rows): out row o = sum of T[row_list[p]] for
// p in [row_ptr[o], row_ptr[o+1]), i.e. over the offsets the voxel really has, in ascending offset order -- the same
// additions in the same order as the table walk above (a missing neighbour added an exact 0 there), so the results are
// bit-identical.  Loads in unconditional batches of 4 (index clamped to the row's last entry: an L1 hit).
__device__ __forceinline__ float4 reduce_row_list(const float4 *__restrict__ T, const int32_t *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ row_list, int C4, int64_t o, int c4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b = row_ptr[o], e = row_ptr[o + 1];
    for (int p = b; p < e; p += 4) {
        int idx[4];
        float4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = row_list[p + j < e ? p + j : e - 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = T[(int64_t)idx[j] * C4 + c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = p + j < e;
            acc.x += ok ? t[j].x : 0.f;
            acc.y += ok ? t[j].y : 0.f;
            acc.z += ok ? t[j].z : 0.f;
            acc.w += ok ? t[j].w : 0.f;
        }
    }
    return acc;
}

// two rows at once (the statistics kernels below walk several rows per thread: a row's walk is a chain of three
// dependent loads, and two independent chains in flight halve what a thread waits for); per row the additions are those
// of reduce_row_list in the same order
__device__ __forceinline__ void reduce_row_list2(const float4 *__restrict__ T, const int32_t *__restrict__ row_ptr,
                                                 const int32_t *__restrict__ row_list, int C4, int64_t o0, int64_t o1,
                                                 int c4, float4 &r0, float4 &r1) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b0 = row_ptr[o0], e0 = row_ptr[o0 + 1], b1 = row_ptr[o1], e1 = row_ptr[o1 + 1];
    const int len = (e0 - b0) > (e1 - b1) ? (e0 - b0) : (e1 - b1);
    for (int q = 0; q < len; q += 4) {
        int i0[4], i1[4];
        float4 t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            i0[j] = row_list[b0 + q + j < e0 ? b0 + q + j : (e0 > b0 ? e0 - 1 : 0)];   // an empty row reads entry 0: masked
            i1[j] = row_list[b1 + q + j < e1 ? b1 + q + j : (e1 > b1 ? e1 - 1 : 0)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = T[(int64_t)i0[j] * C4 + c4];
            t1[j] = T[(int64_t)i1[j] * C4 + c4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok0 = b0 + q + j < e0, ok1 = b1 + q + j < e1;
            a0.x += ok0 ? t0[j].x : 0.f; a0.y += ok0 ? t0[j].y : 0.f; a0.z += ok0 ? t0[j].z : 0.f; a0.w += ok0 ? t0[j].w : 0.f;
            a1.x += ok1 ? t1[j].x : 0.f; a1.y += ok1 ? t1[j].y : 0.f; a1.z += ok1 ? t1[j].z : 0.f; a1.w += ok1 ? t1[j].w : 0.f;
        }
    }
    r0 = a0;
    r1 = a1;
}

__global__ __launch_bounds__(256) void k_sconv_reduce_rows4(const float4 *__restrict__ T,
                                                            const int32_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ row_list, int64_t n, int C4,
                                                            const float4 *__restrict__ bias,
                                                            const float4 *__restrict__ addend,
                                                            float4 *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C4) return;
    int64_t o = idx / C4;
    int c4 = (int)(idx % C4);
    float4 acc = reduce_row_list(T, row_ptr, row_list, C4, o, c4);
    if (bias) {
        float4 b = bias[c4];
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    if (addend) {
        float4 a = addend[idx];
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    out[idx] = acc;
}

__global__ __launch_bounds__(256) void k_sconv_reduce_rows4_stats(const float4 *__restrict__ T,
                                                                  const int32_t *__restrict__ row_ptr,
                                                                  const int32_t *__restrict__ row_list, int64_t n,
                                                                  int C4, const float4 *__restrict__ bias,
                                                                  float4 *__restrict__ out, StatsTail tail) {
    __shared__ double red[256 * 8];
    const int RB = 256 / C4;
    const int tid = threadIdx.x;
    const int r = tid / C4, c4 = tid % C4;
    const bool active = r < RB;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const int64_t step = (int64_t)gridDim.x * RB;
        auto emit = [&](int64_t o, float4 acc) {
            if (bias) {
                float4 b = bias[c4];
                acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            }
            out[o * C4 + c4] = acc;
            a[0] += acc.x; a[1] += acc.y; a[2] += acc.z; a[3] += acc.w;
            a[4] += (double)acc.x * acc.x; a[5] += (double)acc.y * acc.y;
            a[6] += (double)acc.z * acc.z; a[7] += (double)acc.w * acc.w;
        };
        // two rows of this thread in flight (rows o and o + step), accumulated in row order
        for (int64_t o = (int64_t)blockIdx.x * RB + r; o < n; o += 2 * step) {
            if (o + step < n) {
                float4 acc0, acc1;
                reduce_row_list2(T, row_ptr, row_list, C4, o, o + step, c4, acc0, acc1);
                emit(o, acc0);
                emit(o + step, acc1);
            } else {
                emit(o, reduce_row_list(T, row_ptr, row_list, C4, o, c4));
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
    lidog_stats_tail(tail, active && r == 0, c4, a);   // partial row + in-kernel finish by the last workgroup
}

// Data-gradient reduction whose epilogue is the BatchNorm-backward reduction of the layer that PRODUCED the rows it
// writes: out[o] = sum of T rows (+ addend) is the complete gradient dy of that layer's BatchNorm output, so
// (sum dy', sum dy' * xhat) -- dy' = dy masked by the layer's ReLU, xhat from its saved pre-normalisation rows -- can be
// accumulated while dy is still in registers instead of by a second pass over dy (bn.hip:k_colreduce_nc4<1>, which then
// is not launched).  Same row -> (workgroup, thread) assignment, same per-thread order, same LDS tree and the same
// terms (red_terms<1>) as that kernel at the same grid size, and the partials go through the same finishing kernel:
// the sums are bit-identical to the two-pass path.
__global__ __launch_bounds__(256) void k_sconv_reduce_rows4_bwdstats(
    const float4 *__restrict__ T, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ row_list, int64_t n,
    int C4, const float4 *__restrict__ addend, float4 *__restrict__ out, const float4 *__restrict__ pre,
    const float4 *__restrict__ relu_y, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ rw, const float *__restrict__ rb, StatsTail tail,
    const uint32_t *__restrict__ rbits) {
    __shared__ double red[256 * 8];
    const int RB = 256 / C4;
    const int tid = threadIdx.x;
    const int r = tid / C4, c4 = tid % C4;
    const bool active = r < RB;
    const bool from_x = relu_y == nullptr && rw != nullptr;
    const bool has_relu = relu_y != nullptr || from_x || rbits != nullptr;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        float m[4], is[4], gw[4] = {0, 0, 0, 0}, gb[4] = {1, 1, 1, 1};
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = mean[c4 * 4 + j]; is[j] = invstd[c4 * 4 + j]; }
        if (from_x) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { gw[j] = rw[c4 * 4 + j]; gb[j] = rb[c4 * 4 + j]; }
        }
        const int64_t step = (int64_t)gridDim.x * RB;
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        auto emit = [&](int64_t idx, float4 g, float4 x, float4 y, float4 ad) {
            g.x += ad.x; g.y += ad.y; g.z += ad.z; g.w += ad.w;   // no addend: + 0 (a sum of products is never -0 ... and
            out[idx] = g;                                          // -0 + 0 = +0 compares equal anyway)
            if (from_x) {   // the forward pass's pre-activation, bit for bit (bn.hip:k_bn_apply4)
                y.x = (x.x - m[0]) * is[0] * gw[0] + gb[0];
                y.y = (x.y - m[1]) * is[1] * gw[1] + gb[1];
                y.z = (x.z - m[2]) * is[2] * gw[2] + gb[2];
                y.w = (x.w - m[3]) * is[3] * gw[3] + gb[3];
            }
            const float xv[4] = {x.x, x.y, x.z, x.w}, gv[4] = {g.x, g.y, g.z, g.w}, yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // = bn.hip:red_terms<1>
                const float gg = (has_relu && !(yv[j] > 0.f)) ? 0.f : gv[j];
                const float xh = (xv[j] - m[j]) * is[j];
                a[j] += (double)gg;
                a[4 + j] += (double)gg * (double)xh;
            }
        };
        // two rows of this thread in flight (rows o and o + step), accumulated in row order
        for (int64_t o = (int64_t)blockIdx.x * RB + r; o < n; o += 2 * step) {
            const int64_t i0 = o * C4 + c4;
            if (o + step < n) {
                const int64_t i1 = (o + step) * C4 + c4;
                const float4 x0 = pre[i0], x1 = pre[i1];
                const float4 y0 = rbits ? lidog_relu_bits_as_float4(rbits, i0) : relu_y ? relu_y[i0] : one;
                const float4 y1 = rbits ? lidog_relu_bits_as_float4(rbits, i1) : relu_y ? relu_y[i1] : one;
                const float4 ad0 = addend ? addend[i0] : zero, ad1 = addend ? addend[i1] : zero;
                float4 g0, g1;
                reduce_row_list2(T, row_ptr, row_list, C4, o, o + step, c4, g0, g1);
                emit(i0, g0, x0, y0, ad0);
                emit(i1, g1, x1, y1, ad1);
            } else {
                const float4 x0 = pre[i0];
                const float4 y0 = rbits ? lidog_relu_bits_as_float4(rbits, i0) : relu_y ? relu_y[i0] : one;
                const float4 ad0 = addend ? addend[i0] : zero;
                emit(i0, reduce_row_list(T, row_ptr, row_list, C4, o, c4), x0, y0, ad0);
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
    lidog_stats_tail(tail, active && r == 0, c4, a);   // partial row + in-kernel finish by the last workgroup
}

extern "C" int lidog_sconv_reduce_rows_bwdstats(const float *T, const int32_t *row_ptr, const int32_t *row_list,
                                                int64_t n, int32_t C, const float *addend, float *out,
                                                const float *pre, const float *relu_y, const uint32_t *relu_bits,
                                                const float *mean, const float *invstd, const float *relu_w,
                                                const float *relu_b, double *sums, double *partial_ws, double count,
                                                float *dw, float *db, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(C % 4 == 0 && C / 4 <= 256, "sconv_reduce_rows_bwdstats: C must be a multiple of 4, <= 1024");
    LIDOG_REQUIRE(pre && mean && invstd && sums && partial_ws, "sconv_reduce_rows_bwdstats: null argument");
    LIDOG_REQUIRE((relu_w == nullptr) == (relu_b == nullptr) &&
                      (relu_y != nullptr) + (relu_w != nullptr) + (relu_bits != nullptr) <= 1,
                  "sconv_reduce_rows_bwdstats: pass at most one of relu_y, relu_bits, (relu_w, relu_b)");
    if (n == 0) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * C + 1), st) == hipSuccess ? 0 : 1;
    const int C4 = C / 4;
    const int64_t nb = lidog_bn_bwd_reduce_blocks(n, C);   // the grid of lidog_bn_bwd_reduce: same partials
    BnFinish fin = {0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dw, db};
    StatsTail tail;
    if (lidog_stats_tail_make(&tail, partial_ws, sums, count, C, fin, st)) return 1;
    k_sconv_reduce_rows4_bwdstats<<<(unsigned)nb, 256, 0, st>>>(
        (const float4 *)T, row_ptr, row_list, n, C4, (const float4 *)addend, (float4 *)out, (const float4 *)pre,
        (const float4 *)relu_y, mean, invstd, relu_w, relu_b, tail, relu_bits);
    lidog_stats_tail_finish(tail, (int)nb, st);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// The same reduction with an evaluation-mode BatchNorm (+ residual + ReLU) applied in the epilogue: the validation path
// (minkunet_bev.py:376-393, running statistics) needs no statistics pass, so the separate BatchNorm kernel and one
// write + read of the convolution output go away.  Same expression and operation order as bn.hip:k_bn_apply4.
__global__ __launch_bounds__(256) void k_sconv_reduce_rows4_bn(const float4 *__restrict__ T,
                                                               const int32_t *__restrict__ row_ptr,
                                                               const int32_t *__restrict__ row_list, int64_t n, int C4,
                                                               const float4 *__restrict__ bias,
                                                               const float4 *__restrict__ mean,
                                                               const float4 *__restrict__ invstd,
                                                               const float4 *__restrict__ w,
                                                               const float4 *__restrict__ b,
                                                               const float4 *__restrict__ res, int relu,
                                                               float4 *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C4) return;
    int64_t o = idx / C4;
    int c4 = (int)(idx % C4);
    float4 v = reduce_row_list(T, row_ptr, row_list, C4, o, c4);
    if (bias) {
        float4 bb = bias[c4];
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    const float4 m = mean[c4], s = invstd[c4], ww = w[c4], bb = b[c4];
    v.x = (v.x - m.x) * s.x * ww.x + bb.x;
    v.y = (v.y - m.y) * s.y * ww.y + bb.y;
    v.z = (v.z - m.z) * s.z * ww.z + bb.z;
    v.w = (v.w - m.w) * s.w * ww.w + bb.w;
    if (res) { float4 r = res[idx]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    out[idx] = v;
}

extern "C" int lidog_sconv_reduce_rows_bn(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n,
                                          int32_t C, const float *bias, const float *mean, const float *invstd,
                                          const float *w, const float *b, const float *residual, int32_t relu,
                                          float *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    LIDOG_REQUIRE(C % 4 == 0 && C >= 4, "sconv_reduce_rows_bn: C must be a multiple of 4");
    LIDOG_REQUIRE(mean && invstd && w && b, "sconv_reduce_rows_bn: BatchNorm vectors missing");
    const int C4 = C / 4;
    k_sconv_reduce_rows4_bn<<<(unsigned)cdiv64(n * C4, 256), 256, 0, st>>>(
        (const float4 *)T, row_ptr, row_list, n, C4, (const float4 *)bias, (const float4 *)mean,
        (const float4 *)invstd, (const float4 *)w, (const float4 *)b, (const float4 *)residual, relu, (float4 *)out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_sconv_reduce_rows(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n,
                                       int32_t C, const float *bias, const float *addend, float *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    LIDOG_REQUIRE(C % 4 == 0 && C >= 4, "sconv_reduce_rows: C must be a multiple of 4");
    const int C4 = C / 4;
    k_sconv_reduce_rows4<<<(unsigned)cdiv64(n * C4, 256), 256, 0, st>>>((const float4 *)T, row_ptr, row_list, n, C4,
                                                                        (const float4 *)bias, (const float4 *)addend,
                                                                        (float4 *)out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_sconv_reduce_rows_stats(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n,
                                             int32_t C, const float *bias, float *out, double *sums,
                                             double *partial_ws, double count, float eps, float momentum, float *mean,
                                             float *invstd, float *running_mean, float *running_var, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(C % 4 == 0 && C / 4 <= 256, "sconv_reduce_rows_stats: C must be a multiple of 4, <= 1024");
    LIDOG_REQUIRE(mean == nullptr || count > 0, "sconv_reduce_rows_stats: finalising needs the row count");
    if (n == 0) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * C + 1), st) == hipSuccess ? 0 : 1;
    int C4 = C / 4, RB = 256 / C4;
    // one row per thread until the partial-sum table is full (2048 workgroups), then a grid-stride loop: a row's walk
    // is a chain of three dependent loads (row_ptr -> row_list -> product rows), and rows handled by the same thread run
    // one after the other -- on the small maps (deep layers, 8 k-point scans) four rows per thread were 4 x that latency
    // with the chip nearly empty (18 us instead of 8)
#ifndef RED_STATS_ROWS
#define RED_STATS_ROWS 1   // rows per thread before the grid-stride loop takes over (A/B switch; was 4)
#endif
    int64_t nb = cdiv64(n, (int64_t)RB * RED_STATS_ROWS);
    if (nb > lidog_stats_max_blocks()) nb = lidog_stats_max_blocks();
    BnFinish fin = {eps, momentum, mean, invstd, running_mean, running_var, nullptr, nullptr};
    StatsTail tail;
    if (lidog_stats_tail_make(&tail, partial_ws, sums, count, C, fin, st)) return 1;
    k_sconv_reduce_rows4_stats<<<(unsigned)nb, 256, 0, st>>>((const float4 *)T, row_ptr, row_list, n, C4,
                                                             (const float4 *)bias, (float4 *)out, tail);
    lidog_stats_tail_finish(tail, (int)nb, st);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// Cin == 1 (the 5^3 stem): no product rows at all.  out[o] = sum over k ascending of x[nbr[k][o]] * W[k][:] straight from
// the neighbour table (a product row of the two-pass path is fmaf(x, w, 0) = x * w, added in the same order: same
// bits).  One thread per output row (consecutive lanes read consecutive entries of a table row), all C4 <= 16
// float4 accumulators in registers, weights broadcast from LDS.  Saves writing and re-reading P x Cout floats
// (420 MB for the 125-offset stem at bs 4) and the [K, n] position table of the reduction.
template <int C4>
__global__ __launch_bounds__(256) void k_sconv_cin1(const float *__restrict__ x, const int32_t *__restrict__ nbr,
                                                    const float4 *__restrict__ W, int64_t n, int K,
                                                    const float4 *__restrict__ bias, float4 *__restrict__ out) {
    extern __shared__ float4 s_w4[];  // [K][C4]
    for (int e = threadIdx.x; e < K * C4; e += 256) s_w4[e] = W[e];
    __syncthreads();
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    float4 acc[C4];
#pragma unroll
    for (int c = 0; c < C4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 5 <= K; k += 5) {   // five neighbours per round: index loads, then the gathers, then the adds
        int idx[5];
        float xv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) idx[j] = nbr[(int64_t)(k + j) * n + o];
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[j] = x[idx[j] < 0 ? 0 : idx[j]];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (idx[j] < 0) continue;   // a missing neighbour adds an exact 0
#pragma unroll
            for (int c = 0; c < C4; ++c) {
                const float4 w = s_w4[(k + j) * C4 + c];
                acc[c].x += xv[j] * w.x;
                acc[c].y += xv[j] * w.y;
                acc[c].z += xv[j] * w.z;
                acc[c].w += xv[j] * w.w;
            }
        }
    }
    for (; k < K; ++k) {
        const int i = nbr[(int64_t)k * n + o];
        if (i < 0) continue;
        const float xv = x[i];
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            const float4 w = s_w4[k * C4 + c];
            acc[c].x += xv * w.x;
            acc[c].y += xv * w.y;
            acc[c].z += xv * w.z;
            acc[c].w += xv * w.w;
        }
    }
#pragma unroll
    for (int c = 0; c < C4; ++c) {
        if (bias) {
            const float4 b = bias[c];
            acc[c].x += b.x; acc[c].y += b.y; acc[c].z += b.z; acc[c].w += b.w;
        }
        out[o * C4 + c] = acc[c];
    }
}

extern "C" int lidog_sconv_cin1(const float *x, const int32_t *nbr, const float *W, const float *bias, int64_t n,
                                int32_t K, int32_t C, float *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE((C == 16 || C == 32 || C == 64) && (int64_t)K * C * 4 <= 48 * 1024,
                  "sconv_cin1: Cout must be 16, 32 or 64 with K*Cout*4 bytes of weights <= 48 KB");
    if (n == 0) return 0;
    const unsigned grid = (unsigned)cdiv64(n, 256);
    const size_t lds = (size_t)K * C * sizeof(float);
    if (C == 16) k_sconv_cin1<4><<<grid, 256, lds, st>>>(x, nbr, (const float4 *)W, n, K, (const float4 *)bias, (float4 *)out);
    else if (C == 32) k_sconv_cin1<8><<<grid, 256, lds, st>>>(x, nbr, (const float4 *)W, n, K, (const float4 *)bias, (float4 *)out);
    else k_sconv_cin1<16><<<grid, 256, lds, st>>>(x, nbr, (const float4 *)W, n, K, (const float4 *)bias, (float4 *)out);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t lidog_sconv_reduce_stats_ws(int64_t n, int32_t C) {
    // doubles of workspace needed by lidog_sconv_reduce_stats
    (void)n;
    return (int64_t)(2048 + STATS_MAX_GROUPS) * 2 * C;
}

extern "C" int lidog_sconv_reduce_stats(const float *T, const int32_t *pos, int64_t n, int32_t K, int32_t C,
                                        const float *bias, float *out, double *sums, double *partial_ws, double count,
                                        float eps, float momentum, float *mean, float *invstd, float *running_mean,
                                        float *running_var, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(C % 4 == 0 && C / 4 <= 256, "sconv_reduce_stats: C must be a multiple of 4, <= 1024");
    LIDOG_REQUIRE(mean == nullptr || count > 0, "sconv_reduce_stats: finalising needs the row count");
    // no rows (a rank whose shard is empty): the sums a SyncBatchNorm all-reduce will add must be zeros, not garbage
    if (n == 0) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * C + 1), st) == hipSuccess ? 0 : 1;
    int C4 = C / 4, RB = 256 / C4;
    int64_t nb = cdiv64(n, (int64_t)RB * 4);
    if (nb > 2048) nb = 2048;
    BnFinish fin = {eps, momentum, mean, invstd, running_mean, running_var, nullptr, nullptr};
    StatsTail tail;
    if (lidog_stats_tail_make(&tail, partial_ws, sums, count, C, fin, st)) return 1;
    k_sconv_reduce4_stats<<<(unsigned)nb, 256, 0, st>>>((const float4 *)T, pos, n, K, C4, (const float4 *)bias,
                                                        (float4 *)out, tail);
    lidog_stats_tail_finish(tail, (int)nb, st);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ weight gradient
// gW[k][ci][co] = sum_p A[pa[p]][ci] * G[pg[p]][co] over segment k.  Tile (16*RM) x (16*CN) of one k,
// pairs processed 32 at a time through LDS (rows stay row-major: no transpose needed for an outer product).
#define WG_R 32
template <int RM, int CN>
__global__ __launch_bounds__(256) void k_sconv_wgrad(const float *__restrict__ A, const int32_t *__restrict__ pa,
                                                     const float *__restrict__ G, const int32_t *__restrict__ pg,
                                                     const int32_t *__restrict__ items, int n_items, int Cin,
                                                     int Cout, float *__restrict__ partial) {
    constexpr int TM = 16 * RM, TN = 16 * CN;
    __shared__ __attribute__((aligned(16))) float As[WG_R * TM];
    __shared__ __attribute__((aligned(16))) float Gs[WG_R * TN];
    const int item = blockIdx.x;  // (offset, pair range) work item, equal-length ranges (see sconv_mfma.hip)
    const int tiles_n = Cout / TN;
    const int ci0 = (blockIdx.y / tiles_n) * TM, co0 = (blockIdx.y % tiles_n) * TN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t p0 = items[n_items + item], p1 = items[2 * n_items + item];

    float acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = 0.f;

    constexpr int AV = (WG_R * TM / 4 + 255) / 256, GV = (WG_R * TN / 4 + 255) / 256;
    // register-staged software pipeline: the gathered rows of chunk t+1 are in flight while chunk t is
    // multiplied out of LDS (the index load -> row load chain is two dependent HBM/L2 round trips)
    float4 ra[AV], rg[GV];
    auto load_chunk = [&](int64_t p) {
        // Unconditional loads (clamped pair index, zero-select afterwards), indices first: a branch around a
        // load makes hipcc wait vmcnt(0) per element and serialises the whole two-level gather.
        int ia[AV], ig[GV];
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + 256 * j;
            f = f < WG_R * TM / 4 ? f : WG_R * TM / 4 - 1;
            int64_t pr = p + f / (TM / 4);
            ia[j] = pa[pr < p1 ? pr : p1 - 1];
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + 256 * j;
            f = f < WG_R * TN / 4 ? f : WG_R * TN / 4 - 1;
            int64_t pr = p + f / (TN / 4);
            ig[j] = pg[pr < p1 ? pr : p1 - 1];
        }
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + 256 * j;
            f = f < WG_R * TM / 4 ? f : WG_R * TM / 4 - 1;
            int r = f / (TM / 4), c4 = f % (TM / 4);
            float4 v = *reinterpret_cast<const float4 *>(A + (size_t)ia[j] * Cin + ci0 + c4 * 4);
            bool ok = p + r < p1;
            ra[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + 256 * j;
            f = f < WG_R * TN / 4 ? f : WG_R * TN / 4 - 1;
            int r = f / (TN / 4), c4 = f % (TN / 4);
            float4 v = *reinterpret_cast<const float4 *>(G + (size_t)ig[j] * Cout + co0 + c4 * 4);
            bool ok = p + r < p1;
            rg[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };
    if (p0 < p1) load_chunk(p0);
    for (int64_t p = p0; p < p1; p += WG_R) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + 256 * j;
            if (f < WG_R * TM / 4) *reinterpret_cast<float4 *>(&As[f * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + 256 * j;
            if (f < WG_R * TN / 4) *reinterpret_cast<float4 *>(&Gs[f * 4]) = rg[j];
        }
        __syncthreads();
        if (p + WG_R < p1) load_chunk(p + WG_R);
#pragma unroll 8
        for (int r = 0; r < WG_R; ++r) {
            float a[RM], g[CN];
#pragma unroll
            for (int i = 0; i < RM; i += 2) {
                float2 v = *reinterpret_cast<const float2 *>(&As[r * TM + ty * RM + i]);
                a[i] = v.x; a[i + 1] = v.y;
            }
            if constexpr (CN % 4 == 0) {  // interleaved column ownership: conflict-free LDS reads (see k_sconv_gemm)
#pragma unroll
                for (int j = 0; j < CN; j += 4) {
                    float4 v = *reinterpret_cast<const float4 *>(&Gs[r * TN + j * 16 + tx * 4]);
                    g[j] = v.x; g[j + 1] = v.y; g[j + 2] = v.z; g[j + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < CN; j += 2) {
                    float2 v = *reinterpret_cast<const float2 *>(&Gs[r * TN + j * 16 + tx * 2]);
                    g[j] = v.x; g[j + 1] = v.y;
                }
            }
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_fmaf(a[i], g[j], acc[i][j]);
        }
    }
    float *dst = partial + (size_t)item * Cin * Cout;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        float *row = dst + (size_t)(ci0 + ty * RM + i) * Cout + co0;
        constexpr int V = (CN % 4 == 0) ? 4 : 2;
#pragma unroll
        for (int j = 0; j < CN; j += V) {
            if constexpr (V == 4)
                *reinterpret_cast<float4 *>(row + j * 16 + tx * 4) = make_float4(acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3]);
            else
                *reinterpret_cast<float2 *>(row + j * 16 + tx * 2) = make_float2(acc[i][j], acc[i][j + 1]);
        }
    }
}

// generic fallback: one thread per (k, ci, co) sweeping its split of the segment
__global__ __launch_bounds__(256) void k_sconv_wgrad_small(const float *__restrict__ A,
                                                           const int32_t *__restrict__ pa,
                                                           const float *__restrict__ G,
                                                           const int32_t *__restrict__ pg,
                                                           const int32_t *__restrict__ items, int n_items,
                                                           int Cin, int Cout, float *__restrict__ partial) {
    const int item = blockIdx.x;
    const int64_t p0 = items[n_items + item], p1 = items[2 * n_items + item];
    for (int e = blockIdx.y * 256 + threadIdx.x; e < Cin * Cout; e += 256 * gridDim.y) {
        int ci = e / Cout, co = e % Cout;
        float acc = 0.f;
        for (int64_t p = p0; p < p1; ++p)
            acc = __builtin_fmaf(A[(size_t)pa[p] * Cin + ci], G[(size_t)pg[p] * Cout + co], acc);
        partial[(size_t)item * Cin * Cout + e] = acc;
    }
}

// Cin == 1 (the 5^3 stem, in_channels = 1): gW[k][c] = sum_p a[pa[p]] * G[pg[p]][c]; one G row (Cout floats)
// per Cout consecutive lanes, 256 / Cout pairs in flight per sweep
__global__ __launch_bounds__(256) void k_sconv_wgrad_cin1(const float *__restrict__ A,
                                                          const int32_t *__restrict__ pa,
                                                          const float *__restrict__ G,
                                                          const int32_t *__restrict__ pg,
                                                          const int32_t *__restrict__ items, int n_items,
                                                          int Cout, float *__restrict__ partial) {
    __shared__ float red[256];
    const int item = blockIdx.x;
    const int64_t p0 = items[n_items + item], p1 = items[2 * n_items + item];
    const int c = threadIdx.x % Cout, r = threadIdx.x / Cout, RL = 256 / Cout;
    // eight pairs per round and thread, their (dependent) index -> value loads all in flight together: this launch is the
    // last one of the backward pass (the stem's output gradient is the last thing the main stream produces) and the
    // optimiser waits for it -- 172 -> 95 us (two pairs per round before).  Pairs past the end are clamped to the last
    // one and contribute a * 0.
    constexpr int U = 8;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < RL) {
        for (int64_t p = p0 + r; p < p1; p += U * RL) {
            float a[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pu = p + (int64_t)u * RL;
                const int64_t pc = pu < p1 ? pu : p1 - 1;
                a[u] = A[pa[pc]];
                g[u] = G[(size_t)pg[pc] * Cout + c];
                if (pu >= p1) a[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u & 3] = __builtin_fmaf(a[u], g[u], acc[u & 3]);
        }
    }
    red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (r == 0) {
        float s = red[c];
        for (int rr = 1; rr < RL; ++rr) s += red[rr * Cout + c];
        partial[(size_t)item * Cout + c] = s;
    }
}

// Cout <= 8 (the 1x1 classifier, 96 -> 7): one lane per input channel, Cout accumulators per lane
__global__ __launch_bounds__(256) void k_sconv_wgrad_cout8(const float *__restrict__ A,
                                                           const int32_t *__restrict__ pa,
                                                           const float *__restrict__ G,
                                                           const int32_t *__restrict__ pg,
                                                           const int32_t *__restrict__ items, int n_items,
                                                           int Cin, int Cout, float *__restrict__ partial) {
    __shared__ float red[8 * 256];
    const int item = blockIdx.x;
    const int64_t p0 = items[n_items + item], p1 = items[2 * n_items + item];
    const int ci = threadIdx.x % Cin, r = threadIdx.x / Cin, RL = 256 / Cin;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (r < RL) {
        // four pairs per round, all their (dependent) index -> row loads in flight together; pairs past the end are
        // clamped to the last one and contribute a * 0
        for (int64_t p = p0 + r; p < p1; p += 4 * RL) {
            float a[4], gv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pu = p + u * RL;
                const int64_t pc = pu < p1 ? pu : p1 - 1;
                a[u] = A[(size_t)pa[pc] * Cin + ci];
                const float *g = G + (size_t)pg[pc] * Cout;
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[u][j] = (j < Cout) ? g[j] : 0.f;
                if (pu >= p1) a[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < Cout) acc[j] = __builtin_fmaf(a[u], gv[u][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] = acc[j];
    __syncthreads();
    if (r == 0) {
        for (int j = 0; j < Cout; ++j) {
            float s = red[j * 256 + ci];
            for (int rr = 1; rr < RL; ++rr) s += red[j * 256 + rr * Cin + ci];
            partial[((size_t)item * Cin + ci) * Cout + j] = s;
        }
    }
}

// gW[k][e] = sum of the partial slots of the work items of offset k (item_off[k] .. item_off[k+1]), `per` slots
// per item.  64 elements x 4 slot lanes per workgroup; each lane adds every 4th slot in order, the four lane sums
// are combined in lane order (deterministic).
// n % 4 == 0: 16 slot lanes x 16 element lanes of float4 (64 elements per workgroup as below, four times the bytes
// in flight); slot lane l adds slots l, l + 16, ... in two chains, the 16 lane sums are combined in lane order
__global__ __launch_bounds__(256) void k_items_sum4(const float4 *__restrict__ partial,
                                                    const int32_t *__restrict__ item_off, int per, int64_t n4,
                                                    float4 *__restrict__ out) {
    __shared__ float4 red[256];
    const int lane_e = threadIdx.x & 15, lane_s = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + lane_e;
    const int k = blockIdx.y;
    const int64_t s0 = (int64_t)item_off[k] * per, s1 = (int64_t)item_off[k + 1] * per;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (e < n4) {
        int64_t sl = s0 + lane_s;
        for (; sl + 16 < s1; sl += 32) {
            const float4 u = partial[sl * n4 + e], v = partial[(sl + 16) * n4 + e];
            a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
            a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
        }
        if (sl < s1) {
            const float4 u = partial[sl * n4 + e];
            a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        }
    }
    red[threadIdx.x] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (lane_s == 0 && e < n4) {
        float4 t = red[lane_e];
#pragma unroll
        for (int l = 1; l < 16; ++l) {
            const float4 u = red[16 * l + lane_e];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        out[(int64_t)k * n4 + e] = t;
    }
}

__global__ __launch_bounds__(256) void k_items_sum(const float *__restrict__ partial,
                                                   const int32_t *__restrict__ item_off, int per, int64_t n,
                                                   float *__restrict__ out) {
    __shared__ float red[256];
    const int lane_e = threadIdx.x & 63, lane_s = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane_e;
    const int k = blockIdx.y;
    const int64_t s0 = (int64_t)item_off[k] * per, s1 = (int64_t)item_off[k + 1] * per;
    float acc = 0.f;
    if (e < n) {
        int64_t sl = s0 + lane_s;
        float a0 = 0.f, a1 = 0.f;  // two independent chains keep two loads in flight
        for (; sl + 4 < s1; sl += 8) {
            a0 += partial[sl * n + e];
            a1 += partial[(sl + 4) * n + e];
        }
        if (sl < s1) a0 += partial[sl * n + e];
        acc = a0 + a1;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (lane_s == 0 && e < n) out[(int64_t)k * n + e] = ((red[lane_e] + red[64 + lane_e]) + red[128 + lane_e]) + red[192 + lane_e];
}

static int pick_tile(int C) {
    if (C % 128 == 0) return 8;
    if (C % 96 == 0) return 6;
    if (C % 64 == 0) return 4;
    if (C % 32 == 0) return 2;
    return 0;
}

template <int RM>
static void launch_wgrad_rm(int cn, dim3 grid, hipStream_t st, const float *A, const int32_t *pa, const float *G,
                            const int32_t *pg, const int32_t *items, int n_items, int Cin, int Cout,
                            float *partial) {
    switch (cn) {
        case 8: k_sconv_wgrad<RM, 8><<<grid, 256, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout, partial); break;
        case 6: k_sconv_wgrad<RM, 6><<<grid, 256, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout, partial); break;
        case 4: k_sconv_wgrad<RM, 4><<<grid, 256, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout, partial); break;
        default: k_sconv_wgrad<RM, 2><<<grid, 256, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout, partial);
    }
}

// number of Cin*Cout-float slots the caller must provide in `partial` for n_items work items
extern "C" int lidog_sconv_wgrad_slabs(int32_t Cin, int32_t Cout, int32_t n_items) {
    if (g_sparse_core == 1 && pick_tile(Cin) && pick_tile(Cout)) return lidog_wgrad_mfma_slabs(Cin, Cout, n_items);
    return n_items;
}

// Slots of one weight-gradient launch: workgroups of the Cin x Cout kernel that are resident on the chip at a time
// (per-CU occupancy of the kernel x CUs); fold: the lidog_sconv_wgrad_in_bn form.  The caller (lidog_amd/me.py) cuts the
// rule book into work items so that a launch is a whole number of rounds of these slots.  0: not a matrix-core shape.
extern "C" int32_t lidog_sconv_wgrad_slots(int32_t Cin, int32_t Cout, int32_t fold) {
    if (!(g_sparse_core == 1 && pick_tile(Cin) && pick_tile(Cout))) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    int per = lidog_wgrad_mfma_wg_per_cu(Cin, Cout, fold);
    return per > 0 ? per * cus : 0;
}

static int sconv_wgrad(const float *A, const int32_t *pair_a, const float *G, const int32_t *pair_g,
                       const int32_t *items, int32_t n_items, const int32_t *item_off, int32_t K, int32_t Cin,
                       int32_t Cout, float *partial, float *gW, InBn in_bn, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(K >= 1 && n_items >= 0, "sconv_wgrad: bad K / n_items");
    LIDOG_REQUIRE(n_items == 0 || partial != nullptr, "sconv_wgrad: partial workspace missing");
    int rm = pick_tile(Cin), cn = pick_tile(Cout);
    int per = 1;
    if (n_items > 0) {
        if (g_sparse_core == 1 && rm && cn) {
            per = lidog_wgrad_mfma_slabs(Cin, Cout, 1);
            lidog_launch_wgrad_mfma(A, pair_a, G, pair_g, items, n_items, Cin, Cout, partial, in_bn, st);
        } else if (rm && cn) {
            dim3 grid((unsigned)n_items, (unsigned)((Cin / (16 * rm)) * (Cout / (16 * cn))));
            switch (rm) {
                case 8: launch_wgrad_rm<8>(cn, grid, st, A, pair_a, G, pair_g, items, n_items, Cin, Cout, partial); break;
                case 6: launch_wgrad_rm<6>(cn, grid, st, A, pair_a, G, pair_g, items, n_items, Cin, Cout, partial); break;
                case 4: launch_wgrad_rm<4>(cn, grid, st, A, pair_a, G, pair_g, items, n_items, Cin, Cout, partial); break;
                default: launch_wgrad_rm<2>(cn, grid, st, A, pair_a, G, pair_g, items, n_items, Cin, Cout, partial);
            }
        } else if (Cin == 1 && Cout <= 256) {
            k_sconv_wgrad_cin1<<<dim3((unsigned)n_items), 256, 0, st>>>(A, pair_a, G, pair_g, items, n_items, Cout,
                                                                        partial);
        } else if (Cout <= 8 && Cin <= 256) {
            k_sconv_wgrad_cout8<<<dim3((unsigned)n_items), 256, 0, st>>>(A, pair_a, G, pair_g, items, n_items, Cin,
                                                                         Cout, partial);
        } else {
            int by = (Cin * Cout + 255) / 256;
            if (by > 64) by = 64;
            k_sconv_wgrad_small<<<dim3((unsigned)n_items, (unsigned)by), 256, 0, st>>>(A, pair_a, G, pair_g, items,
                                                                                      n_items, Cin, Cout, partial);
        }
    }
    int64_t n = (int64_t)Cin * Cout;
    if (n % 4 == 0)
        k_items_sum4<<<dim3((unsigned)cdiv64(n / 4, 16), (unsigned)K), 256, 0, st>>>(
            (const float4 *)partial, item_off, per, n / 4, (float4 *)gW);
    else
        k_items_sum<<<dim3((unsigned)cdiv64(n, 64), (unsigned)K), 256, 0, st>>>(partial, item_off, per, n, gW);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_sconv_wgrad(const float *A, const int32_t *pair_a, const float *G, const int32_t *pair_g,
                                 const int32_t *items, int32_t n_items, const int32_t *item_off, int32_t K,
                                 int32_t Cin, int32_t Cout, float *partial, float *gW, void *stream) {
    return sconv_wgrad(A, pair_a, G, pair_g, items, n_items, item_off, K, Cin, Cout, partial, gW,
                       InBn{nullptr, nullptr, nullptr, nullptr, 0}, stream);
}

// The weight gradient of a convolution whose input is BatchNorm (+ ReLU) of A: A is the raw output of the layer before,
// in_* its batch statistics and affine parameters [Cin]; the normalised rows exist only in LDS (= the weight gradient
// over lidog_bn_apply_bits' output, bit for bit).  Matrix-core kernels only.
extern "C" int lidog_sconv_wgrad_in_bn(const float *A, const int32_t *pair_a, const float *G, const int32_t *pair_g,
                                       const int32_t *items, int32_t n_items, const int32_t *item_off, int32_t K,
                                       int32_t Cin, int32_t Cout, float *partial, float *gW, const float *in_mean,
                                       const float *in_invstd, const float *in_w, const float *in_b, int32_t in_relu,
                                       void *stream) {
    LIDOG_REQUIRE(in_mean && in_invstd && in_w && in_b, "sconv_wgrad_in_bn: input BatchNorm vectors missing");
    LIDOG_REQUIRE(g_sparse_core == 1 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0,
                  "sconv_wgrad_in_bn: matrix-core kernels only (channel counts multiples of 32; got %d -> %d)", Cin, Cout);
    return sconv_wgrad(A, pair_a, G, pair_g, items, n_items, item_off, K, Cin, Cout, partial, gW,
                       InBn{in_mean, in_invstd, in_w, in_b, in_relu}, stream);
}

// ------------------------------------------------------------------ Wt[k][co][ci] = W[k][ci][co]
__global__ __launch_bounds__(256) void k_transpose(const float *__restrict__ W, int Cin, int Cout,
                                                   float *__restrict__ Wt) {
    __shared__ float tile[32][33];
    const int k = blockIdx.z;
    const float *src = W + (size_t)k * Cin * Cout;
    float *dst = Wt + (size_t)k * Cin * Cout;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8)
        if (ci0 + r < Cin && co0 + tx < Cout) tile[r][tx] = src[(size_t)(ci0 + r) * Cout + co0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (co0 + r < Cout && ci0 + tx < Cin) dst[(size_t)(co0 + r) * Cin + ci0 + tx] = tile[tx][r];
}

extern "C" int lidog_transpose_kernel(const float *W, int32_t K, int32_t Cin, int32_t Cout, float *Wt,
                                      void *stream) {
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((Cout + 31) / 32), (unsigned)((Cin + 31) / 32), (unsigned)K);
    k_transpose<<<grid, 256, 0, st>>>(W, Cin, Cout, Wt);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// Every [K][Cin][Cout] kernel of a model transposed to [K][Cout][Cin] in ONE launch (the data gradients of a backward
// pass read the transposed weights; one launch after the optimiser step replaces one small kernel per convolution on
// the backward pass's dependent chain).  desc[m] = (src offset, dst offset, K, Cin, Cout, first tile), offsets in
// floats into `src` / `dst`; tiles are 32 x 32, enumerated per matrix as (k, ci tile, co tile).
__global__ __launch_bounds__(256) void k_transpose_batched(const float *__restrict__ src, float *__restrict__ dst,
                                                           const int64_t *__restrict__ desc, int n_mats) {
    __shared__ float tile[32][33];
    __shared__ int64_t s_d[6];
    const int64_t t = blockIdx.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_mats - 1;   // last matrix whose first tile is <= t
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (desc[(size_t)mid * 6 + 5] <= t) lo = mid;
            else hi = mid - 1;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) s_d[j] = desc[(size_t)lo * 6 + j];
    }
    __syncthreads();
    const int Cin = (int)s_d[3], Cout = (int)s_d[4];
    const int tco = (Cout + 31) / 32, tci = (Cin + 31) / 32;
    const int64_t local = t - s_d[5];
    const int k = (int)(local / (tci * tco));
    const int r = (int)(local - (int64_t)k * tci * tco);
    const int ci0 = (r / tco) * 32, co0 = (r % tco) * 32;
    const float *W = src + s_d[0] + (size_t)k * Cin * Cout;
    float *Wt = dst + s_d[1] + (size_t)k * Cin * Cout;
    if ((Cin & 31) == 0 && (Cout & 31) == 0 && (((uintptr_t)W | (uintptr_t)Wt) & 15) == 0) {
        // full tile, 16-byte accesses both ways: thread (row = t >> 3, quad = t & 7) moves one float4 in and one out
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        const float4 v = *reinterpret_cast<const float4 *>(W + (size_t)(ci0 + r) * Cout + co0 + 4 * q);
        tile[r][4 * q] = v.x; tile[r][4 * q + 1] = v.y; tile[r][4 * q + 2] = v.z; tile[r][4 * q + 3] = v.w;
        __syncthreads();
        const float4 o = make_float4(tile[4 * q][r], tile[4 * q + 1][r], tile[4 * q + 2][r], tile[4 * q + 3][r]);
        *reinterpret_cast<float4 *>(Wt + (size_t)(co0 + r) * Cin + ci0 + 4 * q) = o;
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int rr = ty; rr < 32; rr += 8)
        if (ci0 + rr < Cin && co0 + tx < Cout) tile[rr][tx] = W[(size_t)(ci0 + rr) * Cout + co0 + tx];
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8)
        if (co0 + rr < Cout && ci0 + tx < Cin) Wt[(size_t)(co0 + rr) * Cin + ci0 + tx] = tile[tx][rr];
}

extern "C" int lidog_transpose_batched(const float *src, float *dst, const int64_t *desc, int32_t n_mats,
                                       int64_t total_tiles, void *stream) {
    if (n_mats == 0 || total_tiles == 0) return 0;
    LIDOG_REQUIRE(total_tiles < ((int64_t)1 << 31), "transpose_batched: too many tiles");
    k_transpose_batched<<<(unsigned)total_tiles, 256, 0, (hipStream_t)stream>>>(src, dst, desc, n_mats);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
