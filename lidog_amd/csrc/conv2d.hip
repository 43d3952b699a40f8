// Dense 2-D BEV head on the matrix cores: implicit-GEMM convolution forward / data gradient /
// weight gradient with exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain).
// Reference: Encoder2D = Conv2d(k3,s2,p1,bias=False) x2 + Conv2d(k1) (utils/models/conv2d.py:16-22,
// 116,184-185), NCHW float32.  CPU checker: oracle/ref_torch.py:Encoder2DRef (torch CPU conv).
//
// Every kernel computes D[i][j] = sum_k A(i,k) * B(k,j) on 128x128 (or 96x128) tiles, 32-deep LDS stages,
// 4 waves x MFMA 32x32 tiles.  j always runs over pixels so that the D columns (MFMA lane
// dimension) are contiguous in NCHW memory:
//   FWD    i = co, k = (ci,ky,kx), j = (b,yo,xo)   A = W[co][k]            B = im2col(X)
//   DGRAD  i = ci, k = (co,tap),   j = class pixel  A = Wd[class][ci][k]   B = gathered gY   (4 stride-2
//          parity classes, each with only its 1/2/2/4 contributing taps -> no wasted FLOPs)
//   WGRAD  i = co, k = (b,yo,xo),  j = (ci,ky,kx)   A = gY[co][k]          B = im2col(X)^T, split over k
#include "common.h"
#include "clock_stamp.h"

#include "conv2d.h"

// ------------------------------------------------------------------ WGRAD
// D[co][(ci,ky,kx)] = sum over pixels k = (b,yo,xo) of gY[co][k] * X[ci][2yo-1+ky][2xo-1+kx], split over k.
// 128 x 128 tile, 32 pixels per stage.  Thread (kk = tid & 31, rg = tid >> 5) stages pixel kk of 16 gY rows
// and 16 im2col columns; its (ci,ky,kx) columns are fixed for the whole kernel (packed offset + tap), its pixel
// advances by 32 per stage with adds only.  Loads are unconditional (invalid taps read element 0) and zeroed
// when the stage is written to LDS; LDS row pitch 129 makes the transposing stores conflict-free.
#define WG_KB 32
#define WG_LD 129
__global__ __launch_bounds__(256) void k_conv_wgrad(IgParams p) {
    __shared__ float As[WG_KB * WG_LD];
    __shared__ float Bs[WG_KB * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * IG_T, j0 = blockIdx.x * IG_T;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int k_begin = blockIdx.z * p.k_chunk;
    const int k_end = k_begin + p.k_chunk < p.Kd ? k_begin + p.k_chunk : p.Kd;
    const int kk = tid & 31, rg = tid >> 5;

    int cpk[16];  // ((element offset of the tap relative to the window corner) << 4) | tap, -1 past Nj
    unsigned a_mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int j = j0 + rg + 8 * r;
        cpk[r] = -1;
        if (j < p.Nj) {
            int ci = j / 9, t = j - ci * 9;
            int ty = t / 3, tx = t - ty * 3;
            cpk[r] = ((ci * HW + ty * p.W + tx) << 4) | t;
        }
        a_mask |= (unsigned)(i0 + rg + 8 * r < p.Mi) << r;
    }
    // pixel of this thread in the current stage (clamped to the last pixel once past the end)
    int m = k_begin + kk;
    int pb, yo, xo;
    {
        int mc = m < p.Kd ? m : p.Kd - 1;
        pb = mc / HoWo;
        int r_ = mc - pb * HoWo;
        yo = r_ / p.Wo;
        xo = r_ - yo * p.Wo;
    }

    float ra[16], rb[16];
    unsigned okb = 0;
    bool mv = false;
    auto load_stage = [&]() {
        mv = m < k_end;
        const int a_base = (pb * p.Cout + i0 + rg) * HoWo + yo * p.Wo + xo;
#pragma unroll
        for (int r = 0; r < 16; ++r) ra[r] = p.A[a_base + (((a_mask >> r) & 1u) ? r * 8 * HoWo : 0)];
        const int b_base = pb * p.Cin * HW + (2 * yo - 1) * p.W + 2 * xo - 1;  // window corner (may lie outside)
        const unsigned ym = (unsigned)(yo > 0) | 2u | ((unsigned)(2 * yo + 1 < p.H) << 2);
        const unsigned xm = (unsigned)(xo > 0) | 2u | ((unsigned)(2 * xo + 1 < p.W) << 2);
        const unsigned tapmask = mv ? (((ym & 1u) ? xm : 0u) | ((ym & 2u) ? xm << 3 : 0u) | ((ym & 4u) ? xm << 6 : 0u)) : 0u;
        okb = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = cpk[r];
            const unsigned ok = (c >= 0) ? ((tapmask >> (c & 15)) & 1u) : 0u;
            okb |= ok << r;
            rb[r] = p.Bm[ok ? b_base + (c >> 4) : 0];
        }
    };
    auto advance = [&]() {
        m += WG_KB;
        if (m < p.Kd) {
            xo += WG_KB;
            while (xo >= p.Wo) {
                xo -= p.Wo;
                ++yo;
            }
            while (yo >= p.Ho) {
                yo -= p.Ho;
                ++pb;
            }
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            As[kk * WG_LD + rg + 8 * r] = (mv && ((a_mask >> r) & 1u)) ? ra[r] : 0.f;
            Bs[kk * WG_LD + rg + 8 * r] = ((okb >> r) & 1u) ? rb[r] : 0.f;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    const int li = lane & 31, kh = lane >> 5;
    const float *a_rd = &As[kh * WG_LD + wi + li];
    const float *b_rd = &Bs[kh * WG_LD + wj + li];

    load_stage();
    for (int k0 = k_begin; k0 < k_end; k0 += WG_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        advance();
        load_stage();  // unconditional prefetch: past the end it re-reads the last pixel and is zeroed (mv false)
        float af[2], bf[2], an[2], bn[2];
        af[0] = a_rd[0]; af[1] = a_rd[32];
        bf[0] = b_rd[0]; bf[1] = b_rd[32];
#pragma unroll
        for (int k2 = 0; k2 < WG_KB / 2; ++k2) {
            if (k2 + 1 < WG_KB / 2) {
                an[0] = a_rd[(2 * k2 + 2) * WG_LD]; an[1] = a_rd[(2 * k2 + 2) * WG_LD + 32];
                bn[0] = b_rd[(2 * k2 + 2) * WG_LD]; bn[1] = b_rd[(2 * k2 + 2) * WG_LD + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k2 + 1 < WG_KB / 2) {
                af[0] = an[0]; af[1] = an[1];
                bf[0] = bn[0]; bf[1] = bn[1];
            }
        }
    }

    // D layout of 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        int j = j0 + wj + 32 * tj + li;
        if (j >= p.Nj) continue;
        float *d = p.D + (size_t)blockIdx.z * p.Mi * p.Nj + j;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + wi + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) d[(size_t)i * p.Nj] = acc[ti][tj][e];
            }
        }
    }
}

// ------------------------------------------------------------------ FWD / DGRAD, table-driven
// Same 128-pixel-wide tiles as k_igemm with 32-deep stages, built so that a stage costs ~150 instructions per
// wave besides its MFMAs (the first version spent ~530 on reduction-index arithmetic, as long as the MFMA phase):
//  - the reduction index kk -> (address offset, tap) map is computed once per workgroup into an LDS table;
//    per staged element only a table read, a tap-validity bit test (per-lane 9-bit / 4-bit mask), one add
//    and one select remain; all offsets are 32-bit;
//  - invalid taps load a known-good address; they (and rows past Mi) are zeroed when the stage is written to
//    LDS, not after the load: a select right behind a load makes the wave wait for it before its MFMA phase;
//  - the LDS operand reads of MFMA step k2+1 are issued before the MFMAs of step k2 (sched_barrier).
// WM x (4/WM) waves, each TI x TJ MFMA tiles: 128 x 128 (WM=2,TI=2,TJ=2) or 96 x 128 (WM=1,TI=3,TJ=1; the
// data gradient of a 96-channel input would waste a quarter of a 128-row tile).
template <int MODE, int WM, int TI, int TJ>
__device__ __forceinline__ void conv_s2_tile(const IgParams &p, const int bx, const int by, float *__restrict__ As,
                                             float *__restrict__ Bs) {
    constexpr int TMR = WM * TI * 32;              // tile rows (i)
    constexpr int AV = (TMR * C2_KB / 4) / 256;    // float4 of A per thread per stage
    static_assert((4 / WM) * TJ * 32 == IG_T, "tile is 128 pixels wide");
    static_assert((TMR * C2_KB / 4) % 256 == 0, "A stage must divide over the threads");
    extern __shared__ int2 s_tab[];  // [Kd] (offset, tap)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = by * TMR, j0 = bx * IG_T;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int nt = p.nky * p.nkx;

    for (int kk = tid; kk < p.Kd; kk += 256) {
        int koff, t;
        if (MODE == IG_FWD) {
            int ci = kk / 9;
            t = kk - ci * 9;
            int ty = t / 3, tx = t - ty * 3;
            koff = ci * HW + (ty - 1) * p.W + (tx - 1);
        } else {
            int co = kk / nt, tap = kk - co * nt;
            int ty = tap / p.nkx, tx = tap - ty * p.nkx;
            koff = co * HoWo - ty * p.Wo - tx;
            t = ty * 2 + tx;
        }
        s_tab[kk] = make_int2(koff, t);
    }

    // ---- the pixel of this lane (fixed for the whole kernel): 32-bit element offsets, tap validity mask
    const int j = j0 + (tid & 127);
    const bool jvalid = j < p.Nj;
    const int jj = jvalid ? j : 0;
    int base, safe;
    unsigned tapmask = 0;
    if (MODE == IG_FWD) {
        int pb = jj / HoWo, r = jj - pb * HoWo;
        int yo = r / p.Wo, xo = r - yo * p.Wo;
        base = pb * p.Cin * HW + (2 * yo) * p.W + 2 * xo;  // centre tap, always inside the image
        safe = base;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                int y = 2 * yo - 1 + ty, x = 2 * xo - 1 + tx;
                tapmask |= (unsigned)(y >= 0 && y < p.H && x >= 0 && x < p.W) << (ty * 3 + tx);
            }
    } else {
        int hw = p.Hc * p.Wc;
        int pb = jj / hw, r = jj - pb * hw;
        int pyy = (r / p.Wc) * 2 + p.py, pxx = (r % p.Wc) * 2 + p.px;
        int y0 = (pyy + 1 - p.ky0) >> 1, x0 = (pxx + 1 - p.kx0) >> 1;  // output pixel of the class's first tap
        safe = pb * p.Cout * HoWo;
        base = safe + y0 * p.Wo + x0;  // y0 - ty / x0 - tx for the later taps
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
                tapmask |= (unsigned)(y0 - ty >= 0 && y0 - ty < p.Ho && x0 - tx >= 0 && x0 - tx < p.Wo)
                           << (ty * 2 + tx);
    }
    if (!jvalid) tapmask = 0;

    // ---- the A rows this thread stages
    const float *a_ptr[AV];
    bool a_ok[AV];
#pragma unroll
    for (int v = 0; v < AV; ++v) {
        int f = tid + 256 * v;
        int i = i0 + (f >> 3);
        a_ok[v] = i < p.Mi;
        a_ptr[v] = p.A + (size_t)(a_ok[v] ? i : p.Mi - 1) * p.Kd + (f & 7) * 4;
    }
    __syncthreads();  // table complete

    float4 ra[AV];
    float rb[16];
    unsigned okbits = 0;
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int v = 0; v < AV; ++v) ra[v] = *reinterpret_cast<const float4 *>(a_ptr[v] + k0);
        okbits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int2 e = s_tab[k0 + kw + 2 * r];  // wave-uniform address: a broadcast read
            const unsigned ok = (tapmask >> e.y) & 1u;
            okbits |= ok << r;
            rb[r] = p.Bm[ok ? base + e.x : safe];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            int f = tid + 256 * v;
            int il = f >> 3, q = f & 7;
            const bool ok = a_ok[v];
            As[(q * 4 + 0) * IG_LD + il] = ok ? ra[v].x : 0.f;
            As[(q * 4 + 1) * IG_LD + il] = ok ? ra[v].y : 0.f;
            As[(q * 4 + 2) * IG_LD + il] = ok ? ra[v].z : 0.f;
            As[(q * 4 + 3) * IG_LD + il] = ok ? ra[v].w : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Bs[(kw + 2 * r) * IG_LD + (tid & 127)] = ((okbits >> r) & 1u) ? rb[r] : 0.f;
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int wi = (WM == 1 ? 0 : (wave >> 1)) * (TI * 32), wj = (WM == 1 ? wave : (wave & 1)) * (TJ * 32);
    const int li = lane & 31, kh = lane >> 5;
    const float *a_rd = &As[kh * IG_LD + wi + li];
    const float *b_rd = &Bs[kh * IG_LD + wj + li];

    load_stage(0);
    for (int k0 = 0; k0 < p.Kd; k0 += C2_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        load_stage(k0 + C2_KB < p.Kd ? k0 + C2_KB : k0);  // unconditional prefetch (last stage re-reads itself)
        float af[TI], bf[TJ], an[TI], bn[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) af[a] = a_rd[32 * a];
#pragma unroll
        for (int b = 0; b < TJ; ++b) bf[b] = b_rd[32 * b];
#pragma unroll
        for (int k2 = 0; k2 < C2_KB / 2; ++k2) {
            if (k2 + 1 < C2_KB / 2) {
#pragma unroll
                for (int a = 0; a < TI; ++a) an[a] = a_rd[(2 * k2 + 2) * IG_LD + 32 * a];
#pragma unroll
                for (int b = 0; b < TJ; ++b) bn[b] = b_rd[(2 * k2 + 2) * IG_LD + 32 * b];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k2 + 1 < C2_KB / 2) {
#pragma unroll
                for (int a = 0; a < TI; ++a) af[a] = an[a];
#pragma unroll
                for (int b = 0; b < TJ; ++b) bf[b] = bn[b];
            }
        }
    }

#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        int jo = j0 + wj + 32 * tj + li;
        if (jo >= p.Nj) continue;
        size_t col_off, i_stride;
        if (MODE == IG_FWD) {
            int b = jo / HoWo;
            col_off = (size_t)b * p.Cout * HoWo + (jo - b * HoWo);
            i_stride = HoWo;
        } else {
            int hw = p.Hc * p.Wc;
            int b = jo / hw, r = jo - b * hw;
            int y = (r / p.Wc) * 2 + p.py, x = (r % p.Wc) * 2 + p.px;
            col_off = (size_t)b * p.Cin * HW + (size_t)y * p.W + x;
            i_stride = HW;
        }
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + wi + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) p.D[col_off + (size_t)i * i_stride] = acc[ti][tj][e];
            }
        }
    }
}

// One launch over up to four problems (the stride-2 parity classes of a data gradient, longest reduction first) or one
// (forward): workgroup x -> (class, pixel tile, row tile) with the row tiles of a pixel tile next to each other (the
// second one finds the gathered pixels in L2).  Four launches of 1 744 / 1 734 / 1 734 / 1 724 equal workgroups on 512
// slots were 3.4 rounds each, i.e. four with the last one 40 % full; one launch with the 4-tap class first lets the short
// classes fill the long one's tail.
// (Tried and dropped, round 5: the forward pass's pixel tiles behind the last full round of 128-row workgroups as 64-row
// workgroups in the same launch -- 1 536 + 416 half-sized instead of 1 744 equal ones on 512 slots: 46.79 vs 46.80 ms.)
template <int MODE, int WM, int TI, int TJ>
__global__ __launch_bounds__(256) void k_conv_s2(IgClasses pc) {
    __shared__ float As[C2_KB * IG_LD];
    __shared__ float Bs[C2_KB * IG_LD];
    const int x = blockIdx.x;
    int cls = 0;
    while (cls + 1 < pc.n && x >= pc.first[cls + 1]) ++cls;
    const IgParams p = pc.c[cls];
    const int r = x - pc.first[cls];
    const int rt = pc.row_tiles[cls];
    LIDOG_STAMP_BEGIN()
    conv_s2_tile<MODE, WM, TI, TJ>(p, r / rt, r % rt, As, Bs);
    LIDOG_STAMP_END()
}

// ------------------------------------------------------------------ weight repack for DGRAD
// Wd[ci][co*nt + tap] = W[co][ci][ky(tap)][kx(tap)] per parity class;
// all four parity classes (py, px) in one launch: class cls = 2 py + px owns nt = (py ? 2 : 1) (px ? 2 : 1) taps and the
// slab behind the classes before it (Cin * Cout * {0, 1, 3, 5} floats) -- four launches of a few microseconds each sat
// between the data-gradient kernels of every 3x3 convolution
__global__ __launch_bounds__(256) void k_repack_dgrad_all(const float *__restrict__ W, int Cin, int Cout,
                                                          float *__restrict__ Wd) {
    const int64_t cc = (int64_t)Cin * Cout;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 9 * cc) return;
    const int cls = e < cc ? 0 : e < 3 * cc ? 1 : e < 5 * cc ? 2 : 3;
    const int64_t base = cls == 0 ? 0 : cls == 1 ? cc : cls == 2 ? 3 * cc : 5 * cc;
    const int py = cls >> 1, px = cls & 1;
    const int nky = py ? 2 : 1, nkx = px ? 2 : 1, ky0 = py ? 0 : 1, kx0 = px ? 0 : 1, nt = nky * nkx;
    const int64_t l = e - base;
    int kd = (int)(l % ((int64_t)Cout * nt));
    int ci = (int)(l / ((int64_t)Cout * nt));
    int co = kd / nt, tap = kd % nt;
    int ky = ky0 + (tap / nkx) * 2, kx = kx0 + (tap % nkx) * 2;
    Wd[e] = W[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
}

void lidog_launch_repack_dgrad_all(const float *W, int Cin, int Cout, float *Wd, hipStream_t st) {
    const int64_t total = 9 * (int64_t)Cin * Cout;
    k_repack_dgrad_all<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(W, Cin, Cout, Wd);
}

__global__ __launch_bounds__(256) void k_sum_splits(const float *__restrict__ partial, int64_t n, int splits,
                                                    int64_t stride, float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = partial[i];
    for (int s = 1; s < splits; ++s) acc += partial[(size_t)s * stride + i];
    out[i] = acc;
}

// ------------------------------------------------------------------ 1x1 convolution with few outputs (OutConv 256 -> 7)
#define PW_MAXCO 8
__global__ __launch_bounds__(256) void k_pw_fwd(const float *__restrict__ X, const float *__restrict__ W,
                                                const float *__restrict__ bias, int Cin, int Cout, int64_t HW,
                                                int64_t total, float *__restrict__ Y) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (b, pixel)
    if (e >= total) return;
    int64_t b = e / HW, pix = e - b * HW;
    float acc[PW_MAXCO];
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co) acc[co] = 0.f;
    const float *x = X + (size_t)b * Cin * HW + pix;
    for (int ci = 0; ci < Cin; ++ci) {
        float v = x[(size_t)ci * HW];
#pragma unroll
        for (int co = 0; co < PW_MAXCO; ++co)
            if (co < Cout) acc[co] = __builtin_fmaf(v, W[co * Cin + ci], acc[co]);
    }
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co)
        if (co < Cout) Y[((size_t)b * Cout + co) * HW + pix] = acc[co] + (bias ? bias[co] : 0.f);
}

__global__ __launch_bounds__(256) void k_pw_dgrad(const float *__restrict__ gY, const float *__restrict__ W, int Cin,
                                                  int Cout, int64_t HW, int64_t total, float *__restrict__ gX) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int64_t b = e / HW, pix = e - b * HW;
    float g[PW_MAXCO];
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co) g[co] = (co < Cout) ? gY[((size_t)b * Cout + co) * HW + pix] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < PW_MAXCO; ++co)
            if (co < Cout) acc = __builtin_fmaf(g[co], W[co * Cin + ci], acc);
        gX[((size_t)b * Cin + ci) * HW + pix] = acc;
    }
}

// gW[co][ci] = sum_{b,pix} gY[b][co][pix] * X[b][ci][pix], gbias[co] = sum gY[b][co][pix].
// Block (ci, s): slice s = (image b, quarter of its pixels); partial[s][co*Cin + ci] (and partial[s][Cout*Cin + co]
// from the ci == 0 blocks) in double per block, summed over s in order by k_sum_splits: no atomics.
#define PW_QUARTERS 4
__global__ __launch_bounds__(256) void k_pw_wgrad(const float *__restrict__ X, const float *__restrict__ gY, int Cin,
                                                  int Cout, int HW, float *__restrict__ partial) {
    __shared__ double red[2 * PW_MAXCO][4];
    const int ci = blockIdx.x, s = blockIdx.y;
    const int b = s / PW_QUARTERS, q = s % PW_QUARTERS;
    const int chunk = (HW + PW_QUARTERS - 1) / PW_QUARTERS;
    const int e0 = q * chunk, e1 = e0 + chunk < HW ? e0 + chunk : HW;
    const float *x = X + ((size_t)b * Cin + ci) * HW;
    const float *g = gY + (size_t)b * Cout * HW;
    double acc[PW_MAXCO], accb[PW_MAXCO];
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co) { acc[co] = 0; accb[co] = 0; }
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        float xv = x[e];
#pragma unroll
        for (int co = 0; co < PW_MAXCO; ++co)
            if (co < Cout) {
                float gv = g[(size_t)co * HW + e];
                acc[co] += (double)gv * (double)xv;
                accb[co] += (double)gv;
            }
    }
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co) {
        double a = acc[co], bsum = accb[co];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_down(a, d); bsum += __shfl_down(bsum, d); }
        if ((threadIdx.x & 63) == 0) { red[co][threadIdx.x >> 6] = a; red[PW_MAXCO + co][threadIdx.x >> 6] = bsum; }
    }
    __syncthreads();
    if (threadIdx.x < Cout) {
        int co = threadIdx.x;
        float *dst = partial + (size_t)s * (Cout * Cin + Cout);
        dst[co * Cin + ci] = (float)(red[co][0] + red[co][1] + red[co][2] + red[co][3]);
        if (ci == 0)
            dst[Cout * Cin + co] = (float)(red[PW_MAXCO + co][0] + red[PW_MAXCO + co][1] + red[PW_MAXCO + co][2] +
                                           red[PW_MAXCO + co][3]);
    }
}

// ------------------------------------------------------------------ C ABI
static int out_dim(int H, int k, int s, int p) { return (H + 2 * p - k) / s + 1; }

extern "C" int lidog_conv2d_fwd(const float *x, const float *w, const float *bias, int32_t B, int32_t Cin, int32_t H,
                                int32_t W, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *y,
                                void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 1) {
        LIDOG_REQUIRE(stride == 1 && pad == 0 && Cout <= PW_MAXCO, "conv2d_fwd: 1x1 path needs stride 1, Cout <= 8");
        int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
        if (total) k_pw_fwd<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(x, w, bias, Cin, Cout, HW, total, y);
        LIDOG_LAUNCH_CHECK();
        return 0;
    }
    LIDOG_REQUIRE(ksize == 3 && stride == 2 && pad == 1 && bias == nullptr,
                  "conv2d_fwd: MFMA path implements k3 s2 p1 without bias (Encoder2D)");
    LIDOG_REQUIRE((Cin * 9) % C2_KB == 0, "conv2d_fwd: Cin*9 must be a multiple of 32");
    IgParams p = {};
    p.A = w; p.Bm = x; p.D = y;
    p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Ho = out_dim(H, 3, 2, 1); p.Wo = out_dim(W, 3, 2, 1);
    p.Mi = Cout; p.Nj = B * p.Ho * p.Wo; p.Kd = Cin * 9;
    if (p.Nj == 0) return 0;
    LIDOG_REQUIRE((int64_t)B * Cin * H * W < ((int64_t)1 << 31) && (int64_t)p.Kd * 8 <= 24576,
                  "conv2d_fwd: tensor too large for 32-bit offsets / reduction table");
    IgClasses pc = {};
    pc.c[0] = p;
    pc.n = 1;
    pc.row_tiles[0] = (int)cdiv64(p.Mi, IG_T);
    pc.first[1] = (int)cdiv64(p.Nj, IG_T) * pc.row_tiles[0];
    k_conv_s2<IG_FWD, 2, 2, 2><<<(unsigned)pc.first[pc.n], 256, (size_t)p.Kd * sizeof(int2), st>>>(pc);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_conv2d_dgrad(const float *gy, const float *w, int32_t B, int32_t Cin, int32_t H, int32_t W,
                                  int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *gx, float *ws,
                                  void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 1) {
        LIDOG_REQUIRE(stride == 1 && pad == 0 && Cout <= PW_MAXCO, "conv2d_dgrad: 1x1 path needs stride 1, Cout <= 8");
        int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
        if (total) k_pw_dgrad<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(gy, w, Cin, Cout, HW, total, gx);
        LIDOG_LAUNCH_CHECK();
        return 0;
    }
    LIDOG_REQUIRE(ksize == 3 && stride == 2 && pad == 1, "conv2d_dgrad: MFMA path implements k3 s2 p1");
    LIDOG_REQUIRE(ws != nullptr, "conv2d_dgrad: needs a 9*Cin*Cout float workspace for the repacked weights");
    LIDOG_REQUIRE(Cout % C2_KB == 0, "conv2d_dgrad: Cout must be a multiple of 32");
    LIDOG_REQUIRE((int64_t)B * Cout * out_dim(H, 3, 2, 1) * out_dim(W, 3, 2, 1) < ((int64_t)1 << 31) &&
                      (int64_t)Cout * 4 * 8 <= 24576,
                  "conv2d_dgrad: tensor too large for 32-bit offsets / reduction table");
    int Ho = out_dim(H, 3, 2, 1), Wo = out_dim(W, 3, 2, 1);
    float *slab = ws;
    lidog_launch_repack_dgrad_all(w, Cin, Cout, ws, st);
    const bool rows96 = Cin % 128 != 0 && Cin % 96 == 0;   // 96-row tiles: no idle quarter of a 128-row tile
    IgClasses pc = {};
    IgParams cls_p[4];
    for (int py = 0; py < 2; ++py) {
        for (int px = 0; px < 2; ++px) {
            IgParams &p = cls_p[2 * py + px];
            p = IgParams{};
            p.Bm = gy; p.D = gx;
            p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo;
            p.py = py; p.px = px;
            // y + 1 - ky even: y even -> ky = 1; y odd -> ky in {0, 2}
            p.nky = py ? 2 : 1; p.ky0 = py ? 0 : 1; p.kystep = 2;
            p.nkx = px ? 2 : 1; p.kx0 = px ? 0 : 1; p.kxstep = 2;
            p.Hc = (H - py + 1) / 2; p.Wc = (W - px + 1) / 2;
            int nt = p.nky * p.nkx;
            p.Mi = Cin; p.Nj = B * p.Hc * p.Wc; p.Kd = Cout * nt;
            int64_t total = (int64_t)Cin * Cout * nt;
            p.A = slab;
            slab += total;
        }
    }
    // longest reduction first: (py, px) = (1, 1) has four taps, (0, 1) and (1, 0) two, (0, 0) one
    const int rt = rows96 ? Cin / 96 : (int)cdiv64(Cin, IG_T);
    size_t tab = 0;
    for (int cls : {3, 1, 2, 0}) {
        const IgParams &p = cls_p[cls];
        if (p.Nj <= 0) continue;
        pc.c[pc.n] = p;
        pc.row_tiles[pc.n] = rt;
        pc.first[pc.n + 1] = pc.first[pc.n] + (int)cdiv64(p.Nj, IG_T) * rt;
        ++pc.n;
        if ((size_t)p.Kd * sizeof(int2) > tab) tab = (size_t)p.Kd * sizeof(int2);
    }
    if (pc.n > 0) {
        if (rows96) k_conv_s2<IG_DGRAD, 1, 3, 1><<<(unsigned)pc.first[pc.n], 256, tab, st>>>(pc);
        else k_conv_s2<IG_DGRAD, 2, 2, 2><<<(unsigned)pc.first[pc.n], 256, tab, st>>>(pc);
    }
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_conv2d_wgrad(const float *x, const float *gy, int32_t B, int32_t Cin, int32_t H, int32_t W,
                                  int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *gw, float *gbias,
                                  float *ws, int64_t ws_floats, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 1) {
        LIDOG_REQUIRE(stride == 1 && pad == 0 && Cout <= PW_MAXCO, "conv2d_wgrad: 1x1 path needs stride 1, Cout <= 8");
        const int splits = B * PW_QUARTERS;
        const int64_t slab = (int64_t)Cout * Cin + Cout;
        LIDOG_REQUIRE(ws != nullptr && ws_floats >= splits * slab,
                      "conv2d_wgrad: 1x1 path needs a workspace of %lld floats", (long long)(splits * slab));
        LIDOG_REQUIRE((int64_t)H * W < ((int64_t)1 << 31), "conv2d_wgrad: image too large");
        if (B == 0 || H * W == 0) return 0;
        k_pw_wgrad<<<dim3((unsigned)Cin, (unsigned)splits), 256, 0, st>>>(x, gy, Cin, Cout, H * W, ws);
        k_sum_splits<<<(unsigned)cdiv64((int64_t)Cout * Cin, 256), 256, 0, st>>>(ws, (int64_t)Cout * Cin, splits, slab, gw);
        if (gbias) k_sum_splits<<<1, 256, 0, st>>>(ws + (int64_t)Cout * Cin, Cout, splits, slab, gbias);
        LIDOG_LAUNCH_CHECK();
        return 0;
    }
    LIDOG_REQUIRE(ksize == 3 && stride == 2 && pad == 1 && gbias == nullptr, "conv2d_wgrad: MFMA path implements k3 s2 p1");
    IgParams p = {};
    p.A = gy; p.Bm = x;
    p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Ho = out_dim(H, 3, 2, 1); p.Wo = out_dim(W, 3, 2, 1);
    p.Mi = Cout; p.Nj = Cin * 9; p.Kd = B * p.Ho * p.Wo;
    int64_t slab = (int64_t)p.Mi * p.Nj;
    int tiles = (int)(cdiv64(p.Nj, IG_T) * cdiv64(p.Mi, IG_T));
    // two full rounds of the chip's slots (211 registers: two workgroups per CU, 512 slots on 256 CUs), never a few
    // workgroups more: 36 tiles x 29 splits = 1 044 workgroups ran a third round for the last 20 (round 5)
    int splits = 1024 / tiles;
    if (splits < 1) splits = 1;
    LIDOG_REQUIRE(Cout % 8 == 0, "conv2d_wgrad: Cout must be a multiple of 8");
    LIDOG_REQUIRE((int64_t)Cin * H * W < ((int64_t)1 << 27) && (int64_t)B * Cin * H * W < ((int64_t)1 << 31) &&
                      (int64_t)B * Cout * p.Ho * p.Wo < ((int64_t)1 << 31),
                  "conv2d_wgrad: tensor too large for the packed 32-bit offsets");
    int64_t max_by_k = cdiv64(p.Kd, 4 * WG_KB);
    if (splits > max_by_k) splits = (int)(max_by_k < 1 ? 1 : max_by_k);
    if ((int64_t)splits * slab > ws_floats) splits = (int)(ws_floats / slab);
    LIDOG_REQUIRE(splits >= 1, "conv2d_wgrad: workspace too small (%lld floats needed per split)", (long long)slab);
    p.k_chunk = (int)(cdiv64(cdiv64(p.Kd, splits), WG_KB) * WG_KB);
    splits = (int)cdiv64(p.Kd, p.k_chunk);
    p.D = (splits == 1) ? gw : ws;
    dim3 grid((unsigned)cdiv64(p.Nj, IG_T), (unsigned)cdiv64(p.Mi, IG_T), (unsigned)splits);
    k_conv_wgrad<<<grid, 256, 0, st>>>(p);
    if (splits > 1) k_sum_splits<<<(unsigned)cdiv64(slab, 256), 256, 0, st>>>(ws, slab, splits, slab, gw);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ Adam on a flat buffer
// torch.optim.Adam semantics (L2 weight decay folded into the gradient; bias-corrected step)
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g,
                                              float *__restrict__ m, float *__restrict__ v, int64_t n, float lr_bc1,
                                              float beta1, float beta2, float eps, float wd, float bc2_sqrt,
                                              float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pi = p[i];
        float gi = g[i] * grad_scale + wd * pi;
        float mi = m[i];
        mi = mi + (gi - mi) * (1.f - beta1);
        float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - lr_bc1 * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int lidog_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                               float grad_scale, void *stream) {
    if (n == 0) return 0;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t g = cdiv64(n, 256);
    if (g > 8192) g = 8192;
    k_adam<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, (float)((double)lr / bc1),
                                                         beta1, beta2, eps, weight_decay, (float)sqrt(bc2), grad_scale);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
