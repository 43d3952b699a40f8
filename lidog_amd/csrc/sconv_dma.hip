// Gathered GEMM of the product-row path as a deep asynchronous pipeline (experiment; cdna_hip_programming.md section 5,
// "Pipelining across barriers"): ONE 8-wave workgroup per CU on 256-row tiles, operands staged by LDS-DMA
// (global_load_lds_dwordx4: no registers, no LDS-write instructions) into a ring of 3 stages that stay in flight across
// raw s_barriers with counted vmcnt waits; the workgroup is persistent and walks a flat (tile, chunk) sequence, so the
// gather indices of the next tile and the first chunks of the next tile are in flight while the current one is
// multiplied.  Every wave issues its share of the DMAs and multiplies its 32 rows; two waves share a SIMD.
//
// LDS image of A: 256 rows x 32 floats, UNPADDED (a DMA writes 64 lanes x 16 B contiguously); the 16-B piece p of row i
// holds the global piece p ^ ((i >> 1) & 7) (swizzle on the source address, same involution on the read): an MFMA
// operand read of 32 rows at one k touches 32 banks twice instead of one bank 32 times.
//
// Same MFMA sequence per product row as k_sconv_gemm_mfma (k ascending): T is bit-identical.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DM_TM 256
#define DM_BK 32
#define DM_S 3
#define DM_A_BYTES (DM_TM * DM_BK * 4)

namespace {

// one LDS-DMA wave instruction: 64 lanes x 16 B from per-lane global addresses to lds_dst + lane * 16 (lds_dst uniform)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);   // wave-uniform by construction; tell the compiler
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most n vector-memory operations of this wave may still be outstanding (n is one of the sums below)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
#define C(N) case N: wait_vm<N>(); break;
        C(0) C(5) C(6) C(7) C(16) C(21) C(22) C(23) C(32) C(37) C(38) C(39)
#undef C
        default: wait_vm<0>();
    }
}

template <int NT>
__device__ __forceinline__ void dm_frag_load(const float *p, float (&f)[NT]) {
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
__device__ __forceinline__ void dm_frag_store(float *p, const float (&f)[NT]) {
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
__global__ __launch_bounds__(512, 2) void k_sconv_gemm_dma(const float *__restrict__ A,
                                                          const int32_t *__restrict__ gather,
                                                          const float *__restrict__ B,
                                                          const int32_t *__restrict__ tile_k,
                                                          const int32_t *__restrict__ tile_row0,
                                                          const int32_t *__restrict__ tile_rows, int n_tiles, int Cin,
                                                          int Cout, float *__restrict__ T) {
    constexpr int TN = 32 * NT;
    constexpr int B_BYTES = DM_BK * TN * 4;
    constexpr int STAGE = DM_A_BYTES + B_BYTES;
    constexpr int B_INSTR = B_BYTES / 1024;             // wave instructions per B chunk: 4, 8, 12, 16
    constexpr int B_OPS = B_INSTR > 8 ? 2 : 1;          // per wave (waves without a piece of their own repeat one)
    constexpr int IDX_OFF = DM_S * STAGE;               // two buffers of 256 gather indices
    __shared__ __attribute__((aligned(1024))) char lds[DM_S * STAGE + 2 * 1024];
    const unsigned lds0 = (unsigned)(uintptr_t)lds;     // LDS byte address of the array (addrspace 3 pointer value)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int ncol = Cout / TN, nchunk = Cin / DM_BK;
    const int n_work = n_tiles * ncol;
    const int my_items = (n_work - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_items * nchunk;
    if (total <= 0) return;

    auto item_of = [&](int i, int &k, int &row0, int &rows, int &col0) {
        int w = (int)blockIdx.x + i * (int)gridDim.x;
        int t = w / ncol;
        k = tile_k[t];
        row0 = tile_row0[t];
        rows = tile_rows[t];
        col0 = (w % ncol) * TN;
    };
    // gather indices of item i -> index buffer i & 1 (one DMA instruction per wave: 64 indices; waves 4-7 repeat)
    auto issue_idx = [&](int i) {
        int k, row0, rows, col0;
        item_of(i, k, row0, rows, col0);
        int r = 64 * (wave & 3) + lane;
        r = r < rows ? r : rows - 1;                     // rows past the tile repeat its last pair (never stored)
        glds4(gather + row0 + r, lds0 + IDX_OFF + (i & 1) * 1024 + (wave & 3) * 256);
    };
    // per-lane source rows of the current "issue" item: 4 (row, piece) pairs of this lane, piece swizzled
    const float *a_src[4] = {A, A, A, A};
    const float *b_src = B;
    auto set_item = [&](int i) {
        int k, row0, rows, col0;
        item_of(i, k, row0, rows, col0);
        const int32_t *idx = reinterpret_cast<const int32_t *>(lds + IDX_OFF + (i & 1) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int row = 8 * (wave * 4 + q) + (lane >> 3);
            int piece = (lane & 7) ^ ((row >> 1) & 7);
            a_src[q] = A + (size_t)idx[row] * Cin + piece * 4;
        }
        b_src = B + (size_t)k * Cin * Cout + col0;
    };
    // DMAs of flat chunk f (0 <= f < total) into stage f % 3; returns the number of operations this wave issued
    auto issue = [&](int f) -> int {
        const int i = f / nchunk, c = f % nchunk, st = f % DM_S;
        int ops = 4 + B_OPS;
        if (c == 0) {
            set_item(i);
            if (i + 1 < my_items) {
                issue_idx(i + 1);
                ++ops;
            }
        }
        const int kb = c * DM_BK;
        const unsigned sa = lds0 + st * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(a_src[q] + kb, sa + (wave * 4 + q) * 1024);
#pragma unroll
        for (int u = 0; u < B_OPS; ++u) {
            int h = wave + 8 * u;
            if (h >= B_INSTR) h = wave % (B_INSTR < 8 ? B_INSTR : 8) ;   // repeat a piece: same bytes to the same place
            int f4 = h * 64 + lane;
            int kk = f4 / (TN / 4), c4 = f4 % (TN / 4);
            glds16(b_src + (size_t)(kb + kk) * Cout + c4 * 4, sa + DM_A_BYTES + h * 1024);
        }
        return ops;
    };

    // ---- prologue: indices of item 0, then chunks 0 and 1
    issue_idx(0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int g_next = 0;          // operations of the newest group in flight (chunk j + 1 at the top of iteration j)
    int g_cur = issue(0);
    if (total > 1) g_next = issue(1);
    int st_old = 0, st_new = 0;   // stores issued between group j and group j + 1 / after group j + 1 (exact counts only)

    f32x16 acc[NT];
    const int rbase = (wave * 32 + li) * 32, sw = ((wave * 32 + li) >> 1) & 7;
    for (int j = 0; j < total; ++j) {
        // chunk j has landed once at most (stores after it + group j + 1 + stores after that) operations are outstanding
        wait_vm_dyn(st_old + g_next + st_new);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // stage (j + 2) % 3 held chunk j - 1: every wave is past it now
        int g_new = (j + 2 < total) ? issue(j + 2) : 0;
        const int i = j / nchunk, c = j % nchunk;
        const float *As = reinterpret_cast<const float *>(lds + (j % DM_S) * STAGE);
        const float *Bs = reinterpret_cast<const float *>(lds + (j % DM_S) * STAGE + DM_A_BYTES);
        if (c == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        }
        const float *arow = As + rbase + kh;
        const float *bcol = Bs + kh * TN + li * NT;
        // k step s (k = 2 s + kh): piece (s >> 1) ^ sw of the row, element 2 (s & 1) + kh of the piece
#define A_AT(S_) arow[((((S_) >> 1) ^ sw) << 2) + 2 * ((S_)&1)]
        float bq0[NT], bq1[NT], bn0[NT], bn1[NT], a0, a1, an0 = 0.f, an1 = 0.f;
        dm_frag_load<NT>(bcol, bq0);
        dm_frag_load<NT>(bcol + 2 * TN, bq1);
        a0 = A_AT(0);
        a1 = A_AT(1);
#pragma unroll
        for (int r = 0; r < DM_BK / 4; ++r) {
            if (r + 1 < DM_BK / 4) {
                dm_frag_load<NT>(bcol + (4 * r + 4) * TN, bn0);
                dm_frag_load<NT>(bcol + (4 * r + 6) * TN, bn1);
                an0 = A_AT(2 * r + 2);
                an1 = A_AT(2 * r + 3);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < DM_BK / 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bq0[t] = bn0[t];
                    bq1[t] = bn1[t];
                }
                a0 = an0;
                a1 = an1;
            }
        }
#undef A_AT
        int stored = 0;
        if (c == nchunk - 1) {
            int k, row0, rows, col0;
            item_of(i, k, row0, rows, col0);
            if (rows == DM_TM) {     // full tile: 16 stores per wave, all lanes (an exact count for the vmcnt arithmetic)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int r = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    float v[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = acc[t][e];
                    dm_frag_store<NT>(T + (size_t)(row0 + r) * Cout + col0 + li * NT, v);
                }
                stored = NT == 1 || NT == 2 || NT == 4 ? 16 : 16;   // one store instruction per row for every NT
            } else {                 // partial tile: predicated stores, counted as 0 (the waits then only get stricter)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int r = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    if (r < rows) {
                        float v[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) v[t] = acc[t][e];
                        dm_frag_store<NT>(T + (size_t)(row0 + r) * Cout + col0 + li * NT, v);
                    }
                }
            }
        }
        // next iteration waits for chunk j + 1: after its group came [stores of iteration j - 1 = st_new], group j + 2
        // (issued above, before this iteration's stores) and this iteration's stores
        st_old = st_new;
        st_new = stored;
        g_cur = g_next;
        g_next = g_new;
        (void)g_cur;
    }
    wait_vm<0>();
}

}  // namespace

// experiment entry: tiles of 256 rows (lidog_tiles_host with tile_rows = 256); gather != NULL; Cin, Cout multiples of 32
extern "C" int lidog_sconv_gemm_dma(const float *A, const int32_t *gather, const float *B, const int32_t *tile_k,
                                    const int32_t *tile_row0, const int32_t *tile_rows, int32_t n_tiles, int32_t Cin,
                                    int32_t Cout, float *T, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_tiles == 0) return 0;
    LIDOG_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 64, "sconv_gemm_dma: channel counts %d %d", Cin, Cout);
    int nt = (Cout % 128 == 0) ? 4 : (Cout % 96 == 0) ? 3 : (Cout % 64 == 0) ? 2 : 1;
    int n_work = n_tiles * (Cout / (32 * nt));
    int grid = 256 < n_work ? 256 : n_work;
#define LAUNCH(NT_) \
    k_sconv_gemm_dma<NT_><<<grid, 512, 0, st>>>(A, gather, B, tile_k, tile_row0, tile_rows, n_tiles, Cin, Cout, T)
    switch (nt) {
        case 4: LAUNCH(4); break;
        case 3: LAUNCH(3); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(1);
    }
#undef LAUNCH
    LIDOG_LAUNCH_CHECK();
    return 0;
}
