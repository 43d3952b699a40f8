// Exact-f32 matrix-core versions of the two sparse-convolution GEMMs (gathered GEMM and weight gradient).
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md, "FP32-input MFMA"),
// so results are IDENTICAL to the vector-FMA kernels of sconv.hip and to oracle/me_oracle.c; what changes is
// the operand traffic: one LDS dword per lane per MFMA instead of ~1 B of LDS per FMA, which is what caps the
// vector kernels near 40 % of the (common) 157 TF/s fp32 roof.
//
// MFMA 32x32x2 lane maps: A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31],
// D[i][j]: j = l & 31, i = (e & 3) + 8 (e >> 2) + 4 (l >> 5) for accumulator register e.
#include <stdlib.h>

#include "common.h"
#include "sconv_mfma.h"
#include "clock_stamp.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MG_TM 128
#define MG_BK 32
#define MG_SA 33  // odd row stride: the 32 lanes of an A read (one row each) hit 32 different banks

template <int NT>
__device__ __forceinline__ void frag_load(const float *p, float (&f)[NT]) {
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
__device__ __forceinline__ void frag_store(float *p, const float (&f)[NT]) {
    // the product rows are written once and read once by the reduction pass: streaming (nontemporal) stores keep them
    // from evicting the gathered feature rows and the weights from L2 (gathered GEMM 3-9 % faster)
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

// ------------------------------------------------------------------ gathered GEMM
// Tile 128 pair-rows x 32*NT columns; wave w owns rows [32w, 32w+32) and all NT column tiles.
// MINW: waves per SIMD the register allocation must leave room for (1 = whatever the kernel likes: 140-144 registers,
// three workgroups per CU; 4 = at most 128 registers, four workgroups per CU -- spill-free for the 96-column case only)
// FOLD: the producer's BatchNorm + ReLU applied to the gathered rows on their way into LDS (InBn, sconv_mfma.h); its four
// per-channel vectors for the chunk travel with the chunk's prefetch (16 registers: 160 of the 170 three workgroups allow)
template <int NT, int MINW = 1, bool FOLD = false>
__global__ __launch_bounds__(256, MINW) void k_sconv_gemm_mfma(const float *__restrict__ A,
                                                         const int32_t *__restrict__ gather,
                                                         const float *__restrict__ B,
                                                         const float *__restrict__ bias,
                                                         const int32_t *__restrict__ tile_k,
                                                         const int32_t *__restrict__ tile_row0,
                                                         const int32_t *__restrict__ tile_rows, int Cin, int Cout,
                                                         float *__restrict__ T,
                                                         const int32_t *__restrict__ scatter, InBn in_bn) {
    constexpr int TN = 32 * NT;
    constexpr int BV = (MG_BK * TN / 4) / 256;  // float4 of B per thread per chunk (1..4)
    __shared__ float As[MG_TM * MG_SA];
    __shared__ __attribute__((aligned(16))) float Bs[MG_BK * TN];
    __shared__ int32_t s_src[MG_TM];
    __shared__ int32_t s_dst[MG_TM];

    const int tile = blockIdx.x;
    const int k = tile_k[tile], row0 = tile_row0[tile], rows = tile_rows[tile];
    const int col0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    LIDOG_STAMP_BEGIN()

    // With a scatter index the destination rows go through LDS (every lane of the epilogue needs 16 of them); without
    // one they are row0 + r, and the gather indices are then fetched by the lanes that use them (8 lanes share an
    // address: one request) -- no LDS hop and no barrier between the tile descriptor and the first rows.
    if (scatter) {
        if (tid < MG_TM) {
            int src = -1, dst = -1;
            if (tid < rows) {
                src = gather ? gather[row0 + tid] : (row0 + tid);
                dst = scatter[row0 + tid];
            }
            s_src[tid] = src;
            s_dst[tid] = dst;
        }
        __syncthreads();
    }

    const float *Bk = B + (size_t)k * Cin * Cout + col0;
    // rb0..rb3 are separate named registers on purpose: as an array (float4 rb[BV]) hipcc keeps the B
    // staging values in scratch memory for BV >= 2
    float4 ra[4], rb0, rb1, rb2, rb3;
    float4 pm, ps, pw, pb;
    pm = ps = pw = pb = make_float4(0.f, 0.f, 0.f, 0.f);
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define MG_LOADB(J, R)                                                                      \
    if constexpr (BV > J) {                                                                 \
        int f = tid + 256 * J;                                                              \
        int kk = f / (TN / 4), c4 = f % (TN / 4);                                           \
        R = *reinterpret_cast<const float4 *>(Bk + (size_t)(kb + kk) * Cout + c4 * 4);      \
    }
#define MG_STOREB(J, R) \
    if constexpr (BV > J) *reinterpret_cast<float4 *>(&Bs[(tid + 256 * J) * 4]) = R;
    // the 4 gathered rows this thread stages (8 lanes fetch one 128-B line of a row), fixed for the tile
    const float *a_row[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = tid + 256 * j;
        int r = f >> 3;
        int src = scatter ? s_src[r] : (r < rows ? (gather ? gather[row0 + r] : row0 + r) : -1);
        a_row[j] = A + (size_t)(src < 0 ? 0 : src) * Cin + (f & 7) * 4;
    }
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // unconditional load (a branch around the load would serialise the gather); rows past the end of
            // the tile read row 0
            ra[j] = *reinterpret_cast<const float4 *>(a_row[j] + kb);
        }
        if constexpr (FOLD) {   // the thread's four rows share one channel quad: (tid & 7) * 4 of the chunk
            const int c = kb + (tid & 7) * 4;
            pm = *reinterpret_cast<const float4 *>(in_bn.mean + c);
            ps = *reinterpret_cast<const float4 *>(in_bn.invstd + c);
            pw = *reinterpret_cast<const float4 *>(in_bn.w + c);
            pb = *reinterpret_cast<const float4 *>(in_bn.b + c);
        }
        MG_LOADB(0, rb0) MG_LOADB(1, rb1) MG_LOADB(2, rb2) MG_LOADB(3, rb3)
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            int r = f >> 3, q = f & 7;
            const int o = r * MG_SA + q * 4;
            float4 v = ra[j];
            if constexpr (FOLD) v = in_bn_apply(v, pm, ps, pw, pb, in_bn.relu);
            // rows past the end of the tile hold row 0's values: a row of A only reaches the same row of the result, and
            // those rows are never stored (zeroing them cost 16 selects per chunk: 48.34 -> 48.28 ms per step)
            As[o] = v.x;
            As[o + 1] = v.y;
            As[o + 2] = v.z;
            As[o + 3] = v.w;
        }
        MG_STOREB(0, rb0) MG_STOREB(1, rb1) MG_STOREB(2, rb2) MG_STOREB(3, rb3)
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // bias of this lane's NT columns (+0 when there is none: x + 0 == x bit for bit, also for -0 results? no:
    // -0 + +0 = +0, so the no-bias case adds -0.0f, the identity of IEEE addition)
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = -0.0f;
    if (bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bias[col0 + li * NT + t];
    }

    load_chunk(0);
    for (int kb = 0; kb < Cin; kb += MG_BK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        // unconditional prefetch (the last iteration re-reads its own chunk): a branch around the staging
        // registers sends them through scratch memory
        load_chunk(kb + MG_BK < Cin ? kb + MG_BK : kb);
        const float *arow = &As[(wave * 32 + li) * MG_SA + kh];
        // MFMA column tile t of this wave = columns {li * NT + t}: the NT B operands of one k step are adjacent
        // in the row-major LDS image (one wide LDS read), and the NT results of a lane are adjacent in T (one
        // wide store).  The operands of step k2+1 are requested before the MFMAs of step k2 are issued.
        const float *bcol = &Bs[kh * TN + li * NT];
        // two k steps per round; sched_barrier pins the order "request round j+1, then multiply round j"
        // (left alone, the scheduler sinks the LDS reads below the MFMAs and every round waits a full LDS
        // latency with the matrix pipe idle)
        float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0, an1;
        frag_load<NT>(bcol, bq0);
        frag_load<NT>(bcol + 2 * TN, bq1);
        a0 = arow[0];
        a1 = arow[2];
#pragma unroll
        for (int j = 0; j < MG_BK / 4; ++j) {
            if (j + 1 < MG_BK / 4) {
                frag_load<NT>(bcol + (4 * j + 4) * TN, bn0);
                frag_load<NT>(bcol + (4 * j + 6) * TN, bn1);
                an0 = arow[4 * j + 4];
                an1 = arow[4 * j + 6];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < MG_BK / 4) {
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

    // Epilogue.  Nothing may be pending in vmcnt here except the stores themselves: with a load outstanding (the bias
    // used to be fetched at this point) hipcc puts `s_waitcnt vmcnt(0)` in front of EVERY predicated store of the
    // loop below, and each store then waits for the previous one to be acknowledged by memory (the 16 stores of a
    // wave serialised: 11 % of the whole kernel, 17 % on the 96-column layers).  The bias is therefore loaded before
    // the main loop; rows past the end of the tile are skipped by the exec mask, which needs no wait.
    // (the bias is also ADDED here, in straight-line code: a first use inside the predicated blocks below would bring
    // the per-store wait back; for the same reason the staging loads of the loop's last, redundant prefetch are
    // drained once, here: vmcnt(0), expcnt / lgkmcnt untouched)
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] += bv[t];
    // (Round 3 experiment: the 96-column case -- 3 dword stores at a 12-byte lane stride per accumulator register --
    // staged through LDS and written as 1 KB-contiguous 16-byte stores instead: bit-identical, 0.25 ms per training step
    // SLOWER (50.19 vs 49.95 ms, twice): the nontemporal dword stores already combine, the extra barrier and LDS hop do
    // not pay.  Not kept.)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int r = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        int dst = scatter ? s_dst[r] : (r < rows ? row0 + r : -1);
        if (dst >= 0) {
            float *out = T + (size_t)dst * Cout + col0 + li * NT;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = acc[t][e];
            frag_store<NT>(out, v);
        }
    }
    LIDOG_STAMP_END()
}

// ------------------------------------------------------------------ gathered GEMM, several units per workgroup
// Round 6 (profiles/r06_stalls_bs4_by_shape.txt): the waves of the kernel above spend 59-78 % of their cycles stalled at
// instruction issue behind the matrix pipe and only 13-25 % parked on memory or barriers, yet the pipe is busy 0.47-0.78
// of the time: what is missing is not bandwidth but matrix work to issue while a workgroup walks its dependent
// prologue (tile descriptor -> gather indices -> rows: three memory round trips before the first MFMA) and its epilogue.
// With one unit (a 128-row tile x 32 NT columns) per workgroup that is paid once per 4-12 chunks of MFMA work, by every
// workgroup of a round at the same time.  Here a workgroup owns units u, u + G, u + 2G, ... (G = gridDim.x) and the
// loop over 32-channel chunks simply runs on into the next unit: its descriptor and gather indices are fetched one unit
// ahead, its first chunk by the prefetch of the current unit's last chunk.  Per unit the arithmetic, its order and
// the stores are those of k_sconv_gemm_mfma: identical bits.  No scatter index, no bias (the launcher keeps the
// one-unit kernel for those).
template <int NT, int MINW = 1, bool FOLD = false>
__global__ __launch_bounds__(256, MINW) void k_sconv_gemm_mfma_ms(const float *__restrict__ A,
                                                            const int32_t *__restrict__ gather,
                                                            const float *__restrict__ B,
                                                            const int32_t *__restrict__ tile_k,
                                                            const int32_t *__restrict__ tile_row0,
                                                            const int32_t *__restrict__ tile_rows, int n_tiles,
                                                            int n_units, int Cin, int Cout, float *__restrict__ T,
                                                            InBn in_bn) {
    constexpr int TN = 32 * NT;
    constexpr int BV = (MG_BK * TN / 4) / 256;  // float4 of B per thread per chunk (1..4)
    __shared__ float As[MG_TM * MG_SA];
    __shared__ __attribute__((aligned(16))) float Bs[MG_BK * TN];

    const int G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    LIDOG_STAMP_BEGIN()
    const uint32_t q4 = (tid & 7) * 4;
    const char *Ab = reinterpret_cast<const char *>(A);

    // unit u = (tile u % n_tiles, column tile u / n_tiles): the order of the one-unit kernel's grid (x fastest)
    int u = blockIdx.x;
    int tile = u % n_tiles;
    int row0 = tile_row0[tile], rows = tile_rows[tile], col0 = (u / n_tiles) * TN;
    const float *Bk = B + (size_t)tile_k[tile] * Cin * Cout + col0;
    // BYTE offsets, 32 bits (the launcher takes this kernel only for an A below 4 GiB: 64-bit row pointers for two units
    // do not fit the registers of three workgroups per CU), of the 4 gathered rows this thread stages (8 lanes fetch one
    // 128-B line of a row); rows past the end of the tile read the tile's first row and are never stored
    uint32_t a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = (tid + 256 * j) >> 3;
        r = r < rows ? r : 0;
        uint32_t src = gather ? (uint32_t)gather[row0 + r] : (uint32_t)(row0 + r);
        a_off[j] = (src * (uint32_t)Cin + q4) * 4u;
    }
    // the unit after this one (the current one again when there is none: its loads are then redundant re-reads, as the
    // one-unit kernel's last prefetch)
    int un = u + G;
    bool has_next = un < n_units;
    int unc = has_next ? un : u;
    int ntile = unc % n_tiles;
    int nrow0 = tile_row0[ntile], nrows = tile_rows[ntile], ncol0 = (unc / n_tiles) * TN;
    const float *Bkn = B + (size_t)tile_k[ntile] * Cin * Cout + ncol0;
    uint32_t n_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = (tid + 256 * j) >> 3;
        r = r < nrows ? r : 0;
        uint32_t src = gather ? (uint32_t)gather[nrow0 + r] : (uint32_t)(nrow0 + r);
        n_off[j] = (src * (uint32_t)Cin + q4) * 4u;
    }

    float4 ra[4], rb0, rb1, rb2, rb3;
    float4 pm, ps, pw, pb;
    pm = ps = pw = pb = make_float4(0.f, 0.f, 0.f, 0.f);
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define MS_LOADB(J, R)                                                                  \
    if constexpr (BV > J) {                                                             \
        int f = tid + 256 * J;                                                          \
        int kk = f / (TN / 4), c4 = f % (TN / 4);                                       \
        R = *reinterpret_cast<const float4 *>(pB + (size_t)kk * Cout + c4 * 4);         \
    }
#define MS_STOREB(J, R) \
    if constexpr (BV > J) *reinterpret_cast<float4 *>(&Bs[(tid + 256 * J) * 4]) = R;
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            int r = f >> 3, q = f & 7;
            const int o = r * MG_SA + q * 4;
            float4 v = ra[j];
            if constexpr (FOLD) v = in_bn_apply(v, pm, ps, pw, pb, in_bn.relu);
            As[o] = v.x;
            As[o + 1] = v.y;
            As[o + 2] = v.z;
            As[o + 3] = v.w;
        }
        MS_STOREB(0, rb0) MS_STOREB(1, rb1) MS_STOREB(2, rb2) MS_STOREB(3, rb3)
    };
    // o0..o3: byte offsets of this thread's four rows at the chunk's first channel; pB: the chunk's first weight row;
    // chan: the chunk's first channel (FOLD vectors)
#define MS_LOAD_CHUNK(O0, O1, O2, O3, PB, CHAN)                                         \
    {                                                                                   \
        const float *pB = (PB);                                                         \
        ra[0] = *reinterpret_cast<const float4 *>(Ab + (O0));                           \
        ra[1] = *reinterpret_cast<const float4 *>(Ab + (O1));                           \
        ra[2] = *reinterpret_cast<const float4 *>(Ab + (O2));                           \
        ra[3] = *reinterpret_cast<const float4 *>(Ab + (O3));                           \
        if constexpr (FOLD) {                                                           \
            const int c = (CHAN) + (int)q4;                                             \
            pm = *reinterpret_cast<const float4 *>(in_bn.mean + c);                     \
            ps = *reinterpret_cast<const float4 *>(in_bn.invstd + c);                   \
            pw = *reinterpret_cast<const float4 *>(in_bn.w + c);                        \
            pb = *reinterpret_cast<const float4 *>(in_bn.b + c);                        \
        }                                                                               \
        MS_LOADB(0, rb0) MS_LOADB(1, rb1) MS_LOADB(2, rb2) MS_LOADB(3, rb3)             \
    }

    f32x16 acc[NT];
    MS_LOAD_CHUNK(a_off[0], a_off[1], a_off[2], a_off[3], Bk, 0)
    while (true) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int kb = 0; kb < Cin; kb += MG_BK) {
            __syncthreads();
            store_chunk();
            __syncthreads();
            // unconditional prefetch: the unit's next chunk, or -- behind its last chunk -- the first chunk of the next unit
            {
                const bool last = kb + MG_BK >= Cin;
                const int chan = last ? 0 : kb + MG_BK;
                const uint32_t cb = (uint32_t)chan * 4u;
                MS_LOAD_CHUNK(last ? n_off[0] : a_off[0] + cb, last ? n_off[1] : a_off[1] + cb,
                              last ? n_off[2] : a_off[2] + cb, last ? n_off[3] : a_off[3] + cb,
                              last ? Bkn : Bk + (size_t)chan * Cout, chan)
            }
            const float *arow = &As[(wave * 32 + li) * MG_SA + kh];
            const float *bcol = &Bs[kh * TN + li * NT];
            float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0, an1;
            frag_load<NT>(bcol, bq0);
            frag_load<NT>(bcol + 2 * TN, bq1);
            a0 = arow[0];
            a1 = arow[2];
#pragma unroll
            for (int j = 0; j < MG_BK / 4; ++j) {
                if (j + 1 < MG_BK / 4) {
                    frag_load<NT>(bcol + (4 * j + 4) * TN, bn0);
                    frag_load<NT>(bcol + (4 * j + 6) * TN, bn1);
                    an0 = arow[4 * j + 4];
                    an1 = arow[4 * j + 6];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < MG_BK / 4) {
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
        // epilogue of the unit (see k_sconv_gemm_mfma: nothing but the stores themselves may be pending in vmcnt; the next
        // unit's first chunk was requested a whole MFMA phase ago)
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int r = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (r < rows) {
                float *out = T + (size_t)(row0 + r) * Cout + col0 + li * NT;
                float v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = acc[t][e];
                frag_store<NT>(out, v);
            }
        }
        if (!has_next) break;
        // the next unit becomes the current one; fetch the descriptor and the gather indices of the one after it
        u = un;
        row0 = nrow0;
        rows = nrows;
        col0 = ncol0;
        Bk = Bkn;
#pragma unroll
        for (int j = 0; j < 4; ++j) a_off[j] = n_off[j];
        un = u + G;
        has_next = un < n_units;
        unc = has_next ? un : u;
        ntile = unc % n_tiles;
        nrow0 = tile_row0[ntile];
        nrows = tile_rows[ntile];
        ncol0 = (unc / n_tiles) * TN;
        Bkn = B + (size_t)tile_k[ntile] * Cin * Cout + ncol0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = (tid + 256 * j) >> 3;
            r = r < nrows ? r : 0;
            uint32_t src = gather ? (uint32_t)gather[nrow0 + r] : (uint32_t)(nrow0 + r);
            n_off[j] = (src * (uint32_t)Cin + q4) * 4u;
        }
    }
#undef MS_LOADB
#undef MS_STOREB
#undef MS_LOAD_CHUNK
    LIDOG_STAMP_END()
}

static int g_gemm_multi = -1, g_gemm_slots = 0, g_gemm_suspend = 0;
// Callers that are about to run the gathered GEMM NEXT TO another stream's matrix kernels (the executor's backward pass
// with its weight-gradient stream) suspend the several-units form for that span: its workgroups live as long as the
// launch, so the ones that find their slot taken by the other stream's workgroups start late and finish late -- measured
// in the two-stream training step 47.50 / 47.73 -> 47.91 / 47.85 ms with it on everywhere, against 6.6 % less kernel time
// alone (profiles/r06_ab_gemm_units_in_step.txt).  Nestable; host-side state like lidog_sconv_gemm_units.
void lidog_gemm_multi_suspend(int delta) { g_gemm_suspend += delta; }

// multi: 1 / 0 = several units per workgroup on / off (< 0: leave as it is); slots > 0: plan the units as if that many
// workgroups were resident (0 = ask the device) -- lets tests walk many units per workgroup on small inputs.  Returns the
// previous `multi`.
extern "C" int32_t lidog_sconv_gemm_units(int32_t multi, int32_t slots) {
    const int32_t old = g_gemm_multi < 0 ? 1 : g_gemm_multi;
    if (multi >= 0) g_gemm_multi = multi;
    g_gemm_slots = slots > 0 ? slots : 0;
    return old;
}

// resident workgroups of a kernel on the current device (occupancy x compute units), cached per kernel
template <typename K>
static int gemm_slots(K kernel, int *cache) {
    if (*cache <= 0) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        *cache = per_cu * cus;
    }
    return *cache;
}

int lidog_launch_gemm_mfma(const float *A, const int32_t *gather, const float *B, const float *bias,
                           const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows, int n_tiles,
                           int Cin, int Cout, float *T, const int32_t *scatter, InBn in_bn, int64_t a_rows, hipStream_t st) {
    int nt = (Cout % 128 == 0) ? 4 : (Cout % 96 == 0) ? 3 : (Cout % 64 == 0) ? 2 : 1;
    // (Round 5: 256-column workgroups for the 256-channel layers -- eight MFMA column tiles per wave, the 128 gathered rows
    // of a tile staged once instead of by two workgroups; 248 registers, two workgroups per CU, bit-identical -- measured in
    // the step, same box, alternating: 47.92 / 47.96 -> 48.21 / 48.15 ms.  Not kept.)
    dim3 grid((unsigned)n_tiles, (unsigned)(Cout / (32 * nt)));
    // several units per workgroup (k_sconv_gemm_mfma_ms) where a launch is more than one round of resident workgroups: as
    // many workgroups as are resident, unit u + i G for workgroup u (measured against "the fewest workgroups that need the
    // same number of rounds", which leaves CUs with unequal numbers of workgroups: bs 4 sum of 14 layer shapes 2.79 ->
    // 2.69 ms with that, 2.60 ms with every slot filled; profiles/r06_gemm_units.txt).  LIDOG_GEMM_MULTI=0: one unit per
    // workgroup.
    if (g_gemm_multi < 0) {
        const char *e = getenv("LIDOG_GEMM_MULTI");
        g_gemm_multi = e ? atoi(e) : 1;
    }
    const int multi = g_gemm_multi && g_gemm_suspend <= 0, slots_override = g_gemm_slots;
    if (multi && !scatter && !bias && a_rows > 0 && a_rows * (int64_t)Cin * 4 < ((int64_t)1 << 32)) {
        const int64_t n_units = (int64_t)n_tiles * (Cout / (32 * nt));
#define LAUNCHMS(NT_, MW_, F_)                                                                                       \
    {                                                                                                                \
        static int slots = 0;                                                                                        \
        const int s_ = slots_override > 0 ? slots_override : gemm_slots(k_sconv_gemm_mfma_ms<NT_, MW_, F_>, &slots); \
        if (n_units > s_) {                                                                                          \
            k_sconv_gemm_mfma_ms<NT_, MW_, F_><<<dim3((unsigned)s_), 256, 0, st>>>(                                   \
                A, gather, B, tile_k, tile_row0, tile_rows, n_tiles, (int)n_units, Cin, Cout, T, in_bn);             \
            return 0;                                                                                                \
        }                                                                                                            \
    }
        if (in_bn.mean) {
            switch (nt) {
                case 4: LAUNCHMS(4, 3, true); break;
                case 3: LAUNCHMS(3, 3, true); break;
                case 2: LAUNCHMS(2, 3, true); break;
                default: LAUNCHMS(1, 4, true);
            }
        } else {
            switch (nt) {
                case 4: LAUNCHMS(4, 3, false); break;
                case 3: LAUNCHMS(3, 4, false); break;
                case 2: LAUNCHMS(2, 4, false); break;
                default: LAUNCHMS(1, 4, false);
            }
        }
#undef LAUNCHMS
    }
#define LAUNCHF(NT_, MW_, F_)                                                                                     \
    k_sconv_gemm_mfma<NT_, MW_, F_><<<grid, 256, 0, st>>>(A, gather, B, bias, tile_k, tile_row0, tile_rows, Cin, Cout, \
                                                          T, scatter, in_bn)
#define LAUNCH(NT_, MW_) LAUNCHF(NT_, MW_, false)
    if (in_bn.mean) {
        switch (nt) {
            case 4: LAUNCHF(4, 1, true); break;
            case 3: LAUNCHF(3, 1, true); break;
            case 2: LAUNCHF(2, 1, true); break;
            default: LAUNCHF(1, 1, true);
        }
        return 0;
    }
    // the 96-column kernel fits 128 registers without spilling: four workgroups per CU instead of three, measured in
    // the training step (same box, alternating): 49.88 / 49.84 -> 49.76 / 49.74 ms
    switch (nt) {
        case 4: LAUNCH(4, 1); break;
        case 3: LAUNCH(3, 4); break;
        case 2: LAUNCH(2, 1); break;
        default: LAUNCH(1, 1);
    }
#undef LAUNCH
#undef LAUNCHF
    return 0;
}

// ------------------------------------------------------------------ weight gradient
// gW[k] tile (32 MT) x (32 NT) = sum over pairs of A_row^T (x) G_row.  Rows are the MFMA k dimension, taken
// straight from the row-major LDS images (no transpose).  NW waves share the MT*NT MFMA tiles; when there
// are fewer tiles than waves the waves form NGRP groups that each take every NGRP-th row pair and emit
// their own partial slab.
#define MW_R 32
// FOLD: the A rows are the raw output of the layer before; its BatchNorm + ReLU (InBn) is applied when they are written
// to LDS, from a copy of the four per-channel vectors of this workgroup's input-channel range kept in LDS
// (the 128 x 128 tile with FOLD is held to the 170 registers of three workgroups per CU, as without)
template <int MT, int NT, int NW, int NGRP, bool FOLD = false>
__global__ __launch_bounds__(64 * NW, (FOLD && MT * NT == 16) ? 3 : 1) void k_sconv_wgrad_mfma(const float *__restrict__ A,
                                                              const int32_t *__restrict__ pa,
                                                              const float *__restrict__ G,
                                                              const int32_t *__restrict__ pg,
                                                              const int32_t *__restrict__ items, int n_items,
                                                              int Cin, int Cout, float *__restrict__ partial,
                                                              InBn in_bn) {
    constexpr int TM = 32 * MT, TN = 32 * NT, NTH = 64 * NW;
    constexpr int TILES = MT * NT;
    __shared__ __attribute__((aligned(16))) float s_par[FOLD ? 4 * TM : 4];
    constexpr int WPG = NW / NGRP;            // waves per group
    constexpr int TPW = (TILES + WPG - 1) / WPG;  // tiles per wave
    __shared__ __attribute__((aligned(16))) float As[MW_R * TM];
    __shared__ __attribute__((aligned(16))) float Gs[MW_R * TN];
    // work item = (offset k, pair range [p0, p1)): ranges are cut to equal length on the host, so the offsets
    // with many pairs (the centre offset owns one pair per voxel) get proportionally more workgroups
    // row 3 of the item table = launch order: workgroup x runs item order[x] (me.py:_wgrad_items_host puts items that
    // read the same feature rows next to each other in time and on the same XCD)
    const int item = items[3 * n_items + blockIdx.x];
    const int tiles_n = Cout / TN;
    const int ci0 = (blockIdx.y / tiles_n) * TM, co0 = (blockIdx.y % tiles_n) * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    LIDOG_STAMP_BEGIN()
    const int grp = wave / WPG, wig = wave % WPG;
    const int64_t p0 = items[n_items + item], p1 = items[2 * n_items + item];

    int a_off[TPW], g_off[TPW];
    bool own[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int tt = wig * TPW + t;  // tiles enumerated column-major: tt = nj * MT + mi
        own[t] = tt < TILES;
        int mi = own[t] ? tt % MT : 0, nj = own[t] ? tt / MT : 0;
        a_off[t] = 32 * mi + li;
        g_off[t] = 32 * nj + li;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    constexpr int AV = (MW_R * TM / 4 + NTH - 1) / NTH, GV = (MW_R * TN / 4 + NTH - 1) / NTH;
    // Software pipeline, two levels deep: pair indices of chunk t+2 and gathered rows of chunk t+1 are in
    // flight while chunk t is multiplied.  All loads are unconditional (clamped index, zero-select after):
    // a branch around a load makes hipcc wait vmcnt(0) per element and serialises the two-level gather.
    float4 ra[AV], rg[GV];
    int ia[AV], ig[GV];
    auto load_idx = [&](int64_t p) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + NTH * j;
            f = f < MW_R * TM / 4 ? f : MW_R * TM / 4 - 1;
            int64_t pr = p + f / (TM / 4);
            ia[j] = pa[pr < p1 ? pr : p1 - 1];
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + NTH * j;
            f = f < MW_R * TN / 4 ? f : MW_R * TN / 4 - 1;
            int64_t pr = p + f / (TN / 4);
            ig[j] = pg[pr < p1 ? pr : p1 - 1];
        }
    };
    auto load_rows = [&](int64_t p) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + NTH * j;
            f = f < MW_R * TM / 4 ? f : MW_R * TM / 4 - 1;
            int r = f / (TM / 4), c4 = f % (TM / 4);
            (void)r;
            ra[j] = *reinterpret_cast<const float4 *>(A + (size_t)ia[j] * Cin + ci0 + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + NTH * j;
            f = f < MW_R * TN / 4 ? f : MW_R * TN / 4 - 1;
            int r = f / (TN / 4), c4 = f % (TN / 4);
            (void)r;
            rg[j] = *reinterpret_cast<const float4 *>(G + (size_t)ig[j] * Cout + co0 + c4 * 4);
        }
    };
    if (p0 < p1) {
        load_idx(p0);
        load_rows(p0);
        load_idx(p0 + MW_R);
    }
    if constexpr (FOLD) {   // visible to every thread behind the first barrier of the loop
        for (int i = tid; i < TM; i += NTH) {
            s_par[i] = in_bn.mean[ci0 + i];
            s_par[TM + i] = in_bn.invstd[ci0 + i];
            s_par[2 * TM + i] = in_bn.w[ci0 + i];
            s_par[3 * TM + i] = in_bn.b[ci0 + i];
        }
    }
    for (int64_t p = p0; p < p1; p += MW_R) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            int f = tid + NTH * j;
            // pair rows past the end of the item are zeroed here, not at load time: a select right after the
            // load would make the wave wait for the gather before its MFMA phase instead of after it
            // (componentwise selects: a ternary on a whole float4 goes through scratch memory)
            const bool ok = p + f / (TM / 4) < p1;
            float4 v = ra[j];
            if constexpr (FOLD) {
                const int c = ((f < MW_R * TM / 4 ? f : MW_R * TM / 4 - 1) % (TM / 4)) * 4;
                v = in_bn_apply(v, *reinterpret_cast<const float4 *>(&s_par[c]),
                                *reinterpret_cast<const float4 *>(&s_par[TM + c]),
                                *reinterpret_cast<const float4 *>(&s_par[2 * TM + c]),
                                *reinterpret_cast<const float4 *>(&s_par[3 * TM + c]), in_bn.relu);
            }
            if (f < MW_R * TM / 4) *reinterpret_cast<float4 *>(&As[f * 4]) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            int f = tid + NTH * j;
            const bool ok = p + f / (TN / 4) < p1;
            if (f < MW_R * TN / 4) *reinterpret_cast<float4 *>(&Gs[f * 4]) = make_float4(ok ? rg[j].x : 0.f, ok ? rg[j].y : 0.f, ok ? rg[j].z : 0.f, ok ? rg[j].w : 0.f);
        }
        __syncthreads();
        if (p + MW_R < p1) {
            load_rows(p + MW_R);
            load_idx(p + 2 * MW_R);
        }
        // compile-time trip count (the group only offsets the address); the operands of step q+1 are requested
        // before the MFMAs of step q are issued (sched_barrier pins that order: left alone, the scheduler sinks
        // the LDS reads below the MFMAs and every step waits an LDS latency with the matrix pipe idle)
        constexpr int Q = MW_R / 2 / NGRP;
        float af[TPW], gf[TPW], an[TPW], gn[TPW];
        {
            const int kk = 2 * grp + kh;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                af[t] = As[kk * TM + a_off[t]];
                gf[t] = Gs[kk * TN + g_off[t]];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q + 1 < Q) {
                const int kk = 2 * ((q + 1) * NGRP + grp) + kh;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    an[t] = As[kk * TM + a_off[t]];
                    gn[t] = Gs[kk * TN + g_off[t]];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], gf[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < Q) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    af[t] = an[t];
                    gf[t] = gn[t];
                }
            }
        }
    }
    float *dst = partial + (size_t)(item * NGRP + grp) * Cin * Cout;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!own[t]) continue;
        int tt = wig * TPW + t;
        int mi = tt % MT, nj = tt / MT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int ci = ci0 + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * kh;
            dst[(size_t)ci * Cout + co0 + 32 * nj + li] = acc[t][e];
        }
    }
    LIDOG_STAMP_END()
}

// Round 5, built and measured for the 96 x 96 case, not kept:
//  * four waves of v_mfma_f32_16x16x4_f32 (each a 48 x 48 quadrant = 3 x 3 tiles, LDS rows padded to 112 floats; 90-98
//    registers, four waves per SIMD instead of 164-174 / two): 0.368 / 0.262 ms alone on the stride-1 / stride-2 layers
//    against 0.380 / 0.292 for the three-wave 32x32x2 form below, and 48.91 / 48.92 against 48.72 / 48.78 ms IN the step
//    (same box, alternating): the kernel moves 1 158 MB of gathered rows in 0.37 ms either way (3.1 TB/s out of L2 /
//    Infinity Cache), and a faster lane kernel only takes that bandwidth from the launch stream sooner;
//  * the centre offset of a 3^3 map taken as identity pairs (row = pair number, no index loads, both operands streamed):
//    0.367 vs 0.368 ms alone, 48.95 / 48.91 vs 48.91 / 48.92 in the step -- the two-level software pipeline already hides
//    the index loads, and consecutive rows are no cheaper to fetch than gathered ones (locality is not the limiter).
// profiles/r05_ab_wgrad96.txt
static int tile32(int C) { return (C % 128 == 0) ? 4 : (C % 96 == 0) ? 3 : (C % 64 == 0) ? 2 : 1; }

// (NW, NGRP) per tile count: every wave gets the same number of MFMA tiles, or rows are split over groups
static void wg_shape(int tiles, int *nw, int *ngrp) {
    switch (tiles) {
        case 1: *nw = 4; *ngrp = 4; break;
        case 2: *nw = 4; *ngrp = 2; break;
        case 3: case 6: case 9: *nw = 3; *ngrp = 1; break;
        default: *nw = 4; *ngrp = 1;  // 4, 8, 12, 16
    }
}

int lidog_wgrad_mfma_slabs(int Cin, int Cout, int n_items) {
    int nw, ngrp;
    wg_shape(tile32(Cin) * tile32(Cout), &nw, &ngrp);
    return n_items * ngrp;
}

template <int MT, int NT, int NW, int NGRP>
static void launch_wg(dim3 grid, hipStream_t st, const float *A, const int32_t *pa, const float *G, const int32_t *pg,
                      const int32_t *items, int n_items, int Cin, int Cout, float *partial, InBn in_bn) {
    if (in_bn.mean)
        k_sconv_wgrad_mfma<MT, NT, NW, NGRP, true><<<grid, 64 * NW, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout,
                                                                             partial, in_bn);
    else
        k_sconv_wgrad_mfma<MT, NT, NW, NGRP, false><<<grid, 64 * NW, 0, st>>>(A, pa, G, pg, items, n_items, Cin, Cout,
                                                                              partial, in_bn);
}

template <int MT, int NT, int NW, int NGRP>
static int occ_wg(bool fold) {
    int n = 0;
    hipError_t e = fold ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sconv_wgrad_mfma<MT, NT, NW, NGRP, true>, 64 * NW, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sconv_wgrad_mfma<MT, NT, NW, NGRP, false>, 64 * NW, 0);
    return e == hipSuccess ? n : -1;
}

// workgroups of the weight-gradient kernel for Cin x Cout that fit one CU at a time (the caller sizes its work items so
// that a launch is a whole number of rounds of the chip's slots); -1 on error
int lidog_wgrad_mfma_wg_per_cu(int Cin, int Cout, int fold) {
    int mt = tile32(Cin), nt = tile32(Cout);
#define WG(MT_, NT_, NW_, NG_) return occ_wg<MT_, NT_, NW_, NG_>(fold != 0)
    switch (mt * 10 + nt) {
        case 11: WG(1, 1, 4, 4);
        case 12: WG(1, 2, 4, 2);
        case 21: WG(2, 1, 4, 2);
        case 13: WG(1, 3, 3, 1);
        case 31: WG(3, 1, 3, 1);
        case 14: WG(1, 4, 4, 1);
        case 41: WG(4, 1, 4, 1);
        case 22: WG(2, 2, 4, 1);
        case 23: WG(2, 3, 3, 1);
        case 32: WG(3, 2, 3, 1);
        case 24: WG(2, 4, 4, 1);
        case 42: WG(4, 2, 4, 1);
        case 33: WG(3, 3, 3, 1);
        case 34: WG(3, 4, 4, 1);
        case 43: WG(4, 3, 4, 1);
        default: WG(4, 4, 4, 1);
    }
#undef WG
}

int lidog_launch_wgrad_mfma(const float *A, const int32_t *pa, const float *G, const int32_t *pg,
                            const int32_t *items, int n_items, int Cin, int Cout, float *partial, InBn in_bn,
                            hipStream_t st) {
    int mt = tile32(Cin), nt = tile32(Cout);
    dim3 grid((unsigned)n_items, (unsigned)((Cin / (32 * mt)) * (Cout / (32 * nt))));
#define WG(MT_, NT_, NW_, NG_) launch_wg<MT_, NT_, NW_, NG_>(grid, st, A, pa, G, pg, items, n_items, Cin, Cout, partial, in_bn)
    switch (mt * 10 + nt) {
        case 11: WG(1, 1, 4, 4); break;
        case 12: WG(1, 2, 4, 2); break;
        case 21: WG(2, 1, 4, 2); break;
        case 13: WG(1, 3, 3, 1); break;
        case 31: WG(3, 1, 3, 1); break;
        case 14: WG(1, 4, 4, 1); break;
        case 41: WG(4, 1, 4, 1); break;
        case 22: WG(2, 2, 4, 1); break;
        case 23: WG(2, 3, 3, 1); break;
        case 32: WG(3, 2, 3, 1); break;
        case 24: WG(2, 4, 4, 1); break;
        case 42: WG(4, 2, 4, 1); break;
        case 33: WG(3, 3, 3, 1); break;
        case 34: WG(3, 4, 4, 1); break;
        case 43: WG(4, 3, 4, 1); break;
        default: WG(4, 4, 4, 1);
    }
#undef WG
    return 0;
}
