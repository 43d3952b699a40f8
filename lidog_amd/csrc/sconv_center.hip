// Reduction pass of a stride-1 odd-kernel sparse convolution with the CENTRE offset fused in.
//
// The centre offset of such a kernel pairs every voxel with itself (its segment of the rule book is the identity map),
// so its products are a dense GEMM over consecutive rows.  Instead of writing them into the product rows T and reading
// them back (1 of the 4.3 product rows a LiDAR voxel has on average at stride 1), this kernel computes them on the
// matrix cores for a tile of 128 consecutive output rows and, in its epilogue, adds the product rows of the OTHER
// offsets from T -- in ascending offset order with the centre product taking its place in that order, i.e. exactly the
// additions of the two-pass path in the same order (bit-identical results; the centre product is the same
// ascending-ci fmaf chain starting from 0 that a product row is).  The gathered GEMM then runs over the rule book
// without its centre segment (23 % fewer tiles and product-row bytes at stride 1, 18 % at stride 2).
//
// Mainloop: csrc/sconv_mfma.hip:k_sconv_gemm_mfma (same staging, same operand pipeline) on consecutive rows.
// Epilogue: the tile's per-row lists (lidog_kernel_map_rows with the centre entry marked -1) are staged in LDS; a lane
// owns 16 rows x NT columns of the accumulators (MFMA 32x32x2 layout) and walks two rows at a time, four list entries
// per row in flight (eight 12..16-byte loads per lane, 3 workgroups per CU: ~75 KB in flight per CU).
// Optional BatchNorm statistics: per-tile (sum x, sum x^2) in fp64, partial[tile][2 Cout].
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CG_TM 128
#define CG_BK 32
#define CG_SA 33
#define CG_LMAX (CG_TM * 27)   // list entries of one tile (3^3 kernel: at most 27 per row)

template <int NT>
__device__ __forceinline__ void cg_frag_load(const float *p, float (&f)[NT]) {
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
__global__ __launch_bounds__(256) void k_sconv_center_reduce(const float *__restrict__ A, const float *__restrict__ Wc,
                                                            const float *__restrict__ T,
                                                            const int32_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ row_list, int64_t n, int Cin,
                                                            int Cout, const float *__restrict__ bias,
                                                            const float *__restrict__ addend, float *__restrict__ out,
                                                            double *__restrict__ partial) {
    constexpr int TN = 32 * NT;
    constexpr int BV = (CG_BK * TN / 4) / 256;
    __shared__ float As[CG_TM * CG_SA];
    __shared__ __attribute__((aligned(16))) float Bs[CG_BK * TN];
    __shared__ int32_t s_ptr[CG_TM + 1];
    __shared__ int32_t s_list[CG_LMAX];

    const int64_t row0 = (int64_t)blockIdx.x * CG_TM;
    const int rows = (int)(n - row0 < CG_TM ? n - row0 : CG_TM);
    const int col0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;

    // per-row list bounds of the tile (rows are consecutive, so the tile's list entries are one contiguous span)
    if (tid <= CG_TM) s_ptr[tid] = row_ptr[row0 + (tid < rows ? tid : rows)];
    __syncthreads();
    const int lbase = s_ptr[0];
    const int llen = s_ptr[rows] - lbase;
    for (int i = tid; i < llen; i += 256) s_list[i] = row_list[lbase + i];

    const float *Bk = Wc + col0;
    float4 ra[4], rb0, rb1, rb2, rb3;
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define CG_LOADB(J, R)                                                                      \
    if constexpr (BV > J) {                                                                 \
        int f = tid + 256 * J;                                                              \
        int kk = f / (TN / 4), c4 = f % (TN / 4);                                           \
        R = *reinterpret_cast<const float4 *>(Bk + (size_t)(kb + kk) * Cout + c4 * 4);      \
    }
#define CG_STOREB(J, R) \
    if constexpr (BV > J) *reinterpret_cast<float4 *>(&Bs[(tid + 256 * J) * 4]) = R;
    const float *a_row[4];
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = tid + 256 * j;
        int r = f >> 3;
        a_ok[j] = r < rows;
        a_row[j] = A + (size_t)(row0 + (r < rows ? r : 0)) * Cin + (f & 7) * 4;
    }
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4 *>(a_row[j] + kb);
        CG_LOADB(0, rb0) CG_LOADB(1, rb1) CG_LOADB(2, rb2) CG_LOADB(3, rb3)
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            int r = f >> 3, q = f & 7;
            const int o = r * CG_SA + q * 4;
            const bool ok = a_ok[j];
            As[o] = ok ? ra[j].x : 0.f;
            As[o + 1] = ok ? ra[j].y : 0.f;
            As[o + 2] = ok ? ra[j].z : 0.f;
            As[o + 3] = ok ? ra[j].w : 0.f;
        }
        CG_STOREB(0, rb0) CG_STOREB(1, rb1) CG_STOREB(2, rb2) CG_STOREB(3, rb3)
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = -0.0f;   // identity of IEEE addition
    if (bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bias[col0 + li * NT + t];
    }

    load_chunk(0);
    for (int kb = 0; kb < Cin; kb += CG_BK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        load_chunk(kb + CG_BK < Cin ? kb + CG_BK : kb);
        const float *arow = &As[(wave * 32 + li) * CG_SA + kh];
        const float *bcol = &Bs[kh * TN + li * NT];
        float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0, an1;
        cg_frag_load<NT>(bcol, bq0);
        cg_frag_load<NT>(bcol + 2 * TN, bq1);
        a0 = arow[0];
        a1 = arow[2];
#pragma unroll
        for (int j = 0; j < CG_BK / 4; ++j) {
            if (j + 1 < CG_BK / 4) {
                cg_frag_load<NT>(bcol + (4 * j + 4) * TN, bn0);
                cg_frag_load<NT>(bcol + (4 * j + 6) * TN, bn1);
                an0 = arow[4 * j + 4];
                an1 = arow[4 * j + 6];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < CG_BK / 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bq0[t] = bn0[t];
                    bq1[t] = bn1[t];
                }
                a0 = an0;
                a1 = an1;
            }
        }
    }
    // the redundant last prefetch is drained once, in straight-line code (sconv_mfma.hip explains why)
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // ---- epilogue: out row = sum over the row's list in ascending offset order, entry -1 = the centre product
    const size_t cbase = (size_t)col0 + (size_t)li * NT;
    double s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        // accumulator registers e0 = 2g, e0 + 1 -> rows wave*32 + (e & 3) + 8 (e >> 2) + 4 kh
        int rl[2], pb[2], pe[2];
        float v[2][NT];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 2 * g + j;
            rl[j] = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const bool valid = rl[j] < rows;
            pb[j] = valid ? s_ptr[rl[j]] - lbase : 0;
            pe[j] = valid ? s_ptr[rl[j] + 1] - lbase : 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) v[j][t] = 0.f;
        }
        for (int it = 0;; ++it) {
            const bool more = (pb[0] + 4 * it < pe[0]) | (pb[1] + 4 * it < pe[1]);
            if (!__builtin_amdgcn_ballot_w64(more)) break;   // wave-uniform exit
            int idx[2][4];
            float tv[2][4][NT];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = pb[j] + 4 * it + q;
                    // -2: no entry; LDS read clamped into the staged span
                    idx[j][q] = p < pe[j] ? s_list[p] : -2;
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // unconditional load (row 0 for "no entry" / the centre marker: value discarded below)
                    const float *src = T + (size_t)(idx[j][q] < 0 ? 0 : idx[j][q]) * Cout + cbase;
                    cg_frag_load<NT>(src, tv[j][q]);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ix = idx[j][q];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float add = ix >= 0 ? tv[j][q][t] : (ix == -1 ? acc[t][2 * g + j] : -0.0f);
                        v[j][t] += add;
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (rl[j] < rows) {
                const size_t o = (size_t)(row0 + rl[j]) * Cout + cbase;
                float r[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] = v[j][t] + bv[t];
                if (addend) {
                    float ad[NT];
                    cg_frag_load<NT>(addend + o, ad);
#pragma unroll
                    for (int t = 0; t < NT; ++t) r[t] += ad[t];
                }
                if constexpr (NT == 4) *reinterpret_cast<float4 *>(out + o) = make_float4(r[0], r[1], r[2], r[3]);
                else if constexpr (NT == 2) *reinterpret_cast<float2 *>(out + o) = make_float2(r[0], r[1]);
                else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) out[o + t] = r[t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    s1[t] += (double)r[t];
                    s2[t] += (double)r[t] * (double)r[t];
                }
            }
        }
    }
    if (partial) {
        // per-tile BatchNorm partial sums: lanes kh = 0 / 1 hold the same columns, the four waves different rows;
        // summed in a fixed order (bit-reproducible)
        __syncthreads();   // As is free now
        double *red = reinterpret_cast<double *>(As);   // [4 waves][TN][2]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s1[t] += __shfl_xor(s1[t], 32);
            s2[t] += __shfl_xor(s2[t], 32);
            if (kh == 0) {
                red[((wave * TN) + li * NT + t) * 2] = s1[t];
                red[((wave * TN) + li * NT + t) * 2 + 1] = s2[t];
            }
        }
        __syncthreads();
        if (tid < TN) {
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += red[((w * TN) + tid) * 2];
                b += red[((w * TN) + tid) * 2 + 1];
            }
            double *dst = partial + (size_t)blockIdx.x * 2 * Cout;
            dst[col0 + tid] = a;
            dst[Cout + col0 + tid] = b;
        }
    }
}

extern "C" int64_t lidog_sconv_center_reduce_ws(int64_t n, int32_t C) { return cdiv64(n, CG_TM) * 2 * C; }

// out [n, Cout] = sum over the per-row lists (entry -1 = A[o] . Wc computed here, other entries = rows of T) + bias +
// addend, optional BatchNorm statistics exactly as lidog_sconv_reduce_stats leaves them (sums may be NULL: none).
extern "C" int lidog_sconv_center_reduce(const float *A, const float *Wc, const float *T, const int32_t *row_ptr,
                                         const int32_t *row_list, int64_t n, int32_t Cin, int32_t Cout,
                                         const float *bias, const float *addend, float *out, double *sums,
                                         double *partial_ws, double count, float eps, float momentum, float *mean,
                                         float *invstd, float *running_mean, float *running_var, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && Cout >= 32,
                  "sconv_center_reduce: channel counts must be multiples of 32");
    LIDOG_REQUIRE(sums == nullptr || partial_ws != nullptr, "sconv_center_reduce: statistics need the workspace");
    LIDOG_REQUIRE(mean == nullptr || count > 0, "sconv_center_reduce: finalising needs the row count");
    if (n == 0) {
        if (sums) return hipMemsetAsync(sums, 0, sizeof(double) * (2 * Cout + 1), st) == hipSuccess ? 0 : 1;
        return 0;
    }
    const int nt = (Cout % 128 == 0) ? 4 : (Cout % 96 == 0) ? 3 : (Cout % 64 == 0) ? 2 : 1;
    const int64_t nb = cdiv64(n, CG_TM);
    dim3 grid((unsigned)nb, (unsigned)(Cout / (32 * nt)));
    double *partial = sums ? partial_ws : nullptr;
#define CG_LAUNCH(NT_)                                                                                          \
    k_sconv_center_reduce<NT_><<<grid, 256, 0, st>>>(A, Wc, T, row_ptr, row_list, n, Cin, Cout, bias, addend,   \
                                                     out, partial)
    switch (nt) {
        case 4: CG_LAUNCH(4); break;
        case 3: CG_LAUNCH(3); break;
        case 2: CG_LAUNCH(2); break;
        default: CG_LAUNCH(1);
    }
#undef CG_LAUNCH
    if (sums) {
        BnFinish fin = {eps, momentum, mean, invstd, running_mean, running_var, nullptr, nullptr};
        lidog_launch_sums_finish(partial_ws, (int)nb, Cout, sums, count, fin, st);
    }
    LIDOG_LAUNCH_CHECK();
    return 0;
}
