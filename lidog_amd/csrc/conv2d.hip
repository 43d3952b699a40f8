// Dense 2-D BEV head on the matrix cores: implicit-GEMM convolution forward / data gradient /
// weight gradient with exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain).
// Reference: Encoder2D = Conv2d(k3,s2,p1,bias=False) x2 + Conv2d(k1) (utils/models/conv2d.py:16-22,
// 116,184-185), NCHW float32.  CPU checker: oracle/ref_torch.py:Encoder2DRef (torch CPU conv).
//
// One kernel template computes D[i][j] = sum_k A(i,k) * B(k,j) on 128x128 tiles, 16-deep LDS stages,
// 4 waves x (2x2) MFMA 32x32 tiles.  j always runs over pixels so that the D columns (MFMA lane
// dimension) are contiguous in NCHW memory:
//   FWD    i = co, k = (ci,ky,kx), j = (b,yo,xo)   A = W[co][k]            B = im2col(X)
//   DGRAD  i = ci, k = (co,tap),   j = class pixel  A = Wd[class][ci][k]   B = gathered gY   (4 stride-2
//          parity classes, each with only its 1/2/2/4 contributing taps -> no wasted FLOPs)
//   WGRAD  i = co, k = (b,yo,xo),  j = (ci,ky,kx)   A = gY[co][k]          B = im2col(X)^T, split over k
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IG_T 128
#define IG_KB 16
#define IG_LD 132

struct IgParams {
    const float *A;   // FWD: W [Cout][Cin*9]; DGRAD: Wd class slab [Cin][Cout*nt]; WGRAD: gY
    const float *Bm;  // FWD/WGRAD: X; DGRAD: gY
    float *D;         // FWD: Y; DGRAD: gX; WGRAD: partial slab [split][Cout][Cin*9]
    int Bn, Cin, H, W, Cout, Ho, Wo;
    int Mi, Nj, Kd;   // GEMM extents
    // DGRAD class description
    int py, px, nky, nkx, Hc, Wc;
    int ky0, kystep, kx0, kxstep;
    // WGRAD split
    int k_chunk;
};

enum { IG_FWD = 0, IG_DGRAD = 1, IG_WGRAD = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void k_igemm(IgParams p) {
    __shared__ float As[IG_KB * IG_LD];
    __shared__ float Bs[IG_KB * IG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * IG_T, j0 = blockIdx.x * IG_T;
    const int HoWo = p.Ho * p.Wo;
    int k_begin = 0, k_end = p.Kd;
    if (MODE == IG_WGRAD) {
        k_begin = blockIdx.z * p.k_chunk;
        k_end = k_begin + p.k_chunk < p.Kd ? k_begin + p.k_chunk : p.Kd;
    }

    // ---- per-thread invariant decode
    // FWD/DGRAD: B element (kk = tid/128 + 2r, jj = tid%128): the pixel is fixed per thread
    int pb = 0, pyy = 0, pxx = 0;
    bool jvalid = false;
    if (MODE != IG_WGRAD) {
        int j = j0 + (tid & 127);
        jvalid = j < p.Nj;
        int jj = jvalid ? j : 0;
        if (MODE == IG_FWD) {
            pb = jj / HoWo;
            int r = jj - pb * HoWo;
            pyy = r / p.Wo;
            pxx = r - pyy * p.Wo;
        } else {
            int hw = p.Hc * p.Wc;
            pb = jj / hw;
            int r = jj - pb * hw;
            pyy = (r / p.Wc) * 2 + p.py;  // input-image pixel of this class
            pxx = (r % p.Wc) * 2 + p.px;
        }
    }
    // WGRAD: B element (kk = tid%16, jj = tid/16 + 16r): the 8 (ci,ky,kx) columns are fixed per thread
    int w_off[8];
    int w_dy[8], w_dx[8];
    if (MODE == IG_WGRAD) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int j = j0 + (tid >> 4) + 16 * r;
            if (j < p.Nj) {
                int ci = j / 9, t = j - ci * 9;
                w_dy[r] = t / 3 - 1;
                w_dx[r] = t % 3 - 1;
                w_off[r] = ci * p.H * p.W;
            } else {
                w_off[r] = -1; w_dy[r] = 0; w_dx[r] = 0;
            }
        }
    }

    float ra[8], rb[8];
    auto load_stage = [&](int k0) {
        // ---- A
        if (MODE == IG_WGRAD) {
            int m = k0 + (tid & 15);
            bool mv = m < k_end;
            int b = mv ? m / HoWo : 0;
            int r_ = mv ? m - b * HoWo : 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int i = i0 + (tid >> 4) + 16 * r;
                ra[r] = (mv && i < p.Mi) ? p.A[((size_t)b * p.Cout + i) * HoWo + r_] : 0.f;
            }
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                int f = tid + 256 * v;
                int i = i0 + (f >> 2), q = f & 3;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < p.Mi) t = *reinterpret_cast<const float4 *>(p.A + (size_t)i * p.Kd + k0 + q * 4);
                ra[4 * v] = t.x; ra[4 * v + 1] = t.y; ra[4 * v + 2] = t.z; ra[4 * v + 3] = t.w;
            }
        }
        // ---- B
        if (MODE == IG_FWD) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int kk = k0 + (tid >> 7) + 2 * r;
                int ci = kk / 9, t = kk - ci * 9;
                int y = pyy * 2 - 1 + t / 3, x = pxx * 2 - 1 + t % 3;
                bool ok = jvalid && y >= 0 && y < p.H && x >= 0 && x < p.W;
                rb[r] = ok ? p.Bm[(((size_t)pb * p.Cin + ci) * p.H + y) * p.W + x] : 0.f;
            }
        } else if (MODE == IG_DGRAD) {
            const int nt = p.nky * p.nkx;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int kk = k0 + (tid >> 7) + 2 * r;
                int co = kk / nt, tap = kk - co * nt;
                int ky = p.ky0 + (tap / p.nkx) * p.kystep, kx = p.kx0 + (tap % p.nkx) * p.kxstep;
                int yo2 = pyy + 1 - ky, xo2 = pxx + 1 - kx;  // even by construction of the class
                int yo = yo2 >> 1, xo = xo2 >> 1;
                bool ok = jvalid && yo2 >= 0 && xo2 >= 0 && yo < p.Ho && xo < p.Wo;
                rb[r] = ok ? p.Bm[(((size_t)pb * p.Cout + co) * p.Ho + yo) * p.Wo + xo] : 0.f;
            }
        } else {
            int m = k0 + (tid & 15);
            bool mv = m < k_end;
            int b = mv ? m / HoWo : 0;
            int r_ = mv ? m - b * HoWo : 0;
            int yo = r_ / p.Wo, xo = r_ - yo * p.Wo;
            size_t base = (size_t)b * p.Cin * p.H * p.W;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int y = yo * 2 + w_dy[r], x = xo * 2 + w_dx[r];
                bool ok = mv && w_off[r] >= 0 && y >= 0 && y < p.H && x >= 0 && x < p.W;
                rb[r] = ok ? p.Bm[base + w_off[r] + (size_t)y * p.W + x] : 0.f;
            }
        }
    };
    auto store_stage = [&]() {
        if (MODE == IG_WGRAD) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                As[(tid & 15) * IG_LD + (tid >> 4) + 16 * r] = ra[r];
                Bs[(tid & 15) * IG_LD + (tid >> 4) + 16 * r] = rb[r];
            }
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                int f = tid + 256 * v;
                int il = f >> 2, q = f & 3;
#pragma unroll
                for (int s = 0; s < 4; ++s) As[(q * 4 + s) * IG_LD + il] = ra[4 * v + s];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) Bs[((tid >> 7) + 2 * r) * IG_LD + (tid & 127)] = rb[r];
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

    if (k_begin < k_end) load_stage(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += IG_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        if (k0 + IG_KB < k_end) load_stage(k0 + IG_KB);
#pragma unroll
        for (int k2 = 0; k2 < IG_KB / 2; ++k2) {
            float a0 = As[(2 * k2 + kh) * IG_LD + wi + li];
            float a1 = As[(2 * k2 + kh) * IG_LD + wi + 32 + li];
            float b0 = Bs[(2 * k2 + kh) * IG_LD + wj + li];
            float b1 = Bs[(2 * k2 + kh) * IG_LD + wj + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue.  D layout of 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        int j = j0 + wj + 32 * tj + li;
        if (j >= p.Nj) continue;
        size_t col_off;
        size_t i_stride;
        if (MODE == IG_FWD) {
            int b = j / HoWo;
            col_off = (size_t)b * p.Cout * HoWo + (j - b * HoWo);
            i_stride = HoWo;
        } else if (MODE == IG_DGRAD) {
            int hw = p.Hc * p.Wc;
            int b = j / hw;
            int r = j - b * hw;
            int y = (r / p.Wc) * 2 + p.py, x = (r % p.Wc) * 2 + p.px;
            col_off = (size_t)b * p.Cin * p.H * p.W + (size_t)y * p.W + x;
            i_stride = (size_t)p.H * p.W;
        } else {
            col_off = (size_t)blockIdx.z * p.Mi * p.Nj + j;
            i_stride = p.Nj;
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + wi + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) p.D[col_off + (size_t)i * i_stride] = acc[ti][tj][e];
            }
        }
    }
}

// ------------------------------------------------------------------ FWD / DGRAD, second generation
// Same tiling as k_igemm, but 32-deep stages and all reduction-index arithmetic on the scalar unit: the
// reduction row kk of a staged B element is wave-uniform (kw + 2r), so (ci,ky,kx) / (co,tap) and the address
// offset they imply are SGPR values; per lane only a validity bit (precomputed 3-bit row/column masks), one
// add and two selects remain.  Invalid taps load a known-good address and are zeroed by select (no branch
// around a load).
#define C2_KB 32
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_s2(IgParams p) {
    __shared__ float As[C2_KB * IG_LD];
    __shared__ float Bs[C2_KB * IG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * IG_T, j0 = blockIdx.x * IG_T;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 7);

    // ---- the pixel of this lane (fixed for the whole kernel)
    const int j = j0 + (tid & 127);
    const bool jvalid = j < p.Nj;
    const int jj = jvalid ? j : 0;
    size_t base, safe;
    unsigned ymask = 0, xmask = 0;
    if (MODE == IG_FWD) {
        int pb = jj / HoWo, r = jj - pb * HoWo;
        int yo = r / p.Wo, xo = r - yo * p.Wo;
        base = (size_t)pb * p.Cin * HW + (size_t)(2 * yo) * p.W + 2 * xo;  // centre tap, always inside the image
        safe = base;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int y = 2 * yo - 1 + t, x = 2 * xo - 1 + t;
            ymask |= (unsigned)(y >= 0 && y < p.H) << t;
            xmask |= (unsigned)(x >= 0 && x < p.W) << t;
        }
    } else {
        int hw = p.Hc * p.Wc;
        int pb = jj / hw, r = jj - pb * hw;
        int pyy = (r / p.Wc) * 2 + p.py, pxx = (r % p.Wc) * 2 + p.px;
        int y0 = (pyy + 1 - p.ky0) >> 1, x0 = (pxx + 1 - p.kx0) >> 1;  // output pixel of the class's first tap
        safe = (size_t)pb * p.Cout * HoWo;
        base = safe + (size_t)y0 * p.Wo + x0;   // y0 - t / x0 - t for the later taps
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ymask |= (unsigned)(y0 - t >= 0 && y0 - t < p.Ho) << t;
            xmask |= (unsigned)(x0 - t >= 0 && x0 - t < p.Wo) << t;
        }
    }
    if (!jvalid) ymask = 0;
    const int nt = p.nky * p.nkx;

    float ra[16], rb[16];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            int f = tid + 256 * v;
            int i = i0 + (f >> 3), q = f & 7;
            bool iok = i < p.Mi;
            float4 t = *reinterpret_cast<const float4 *>(p.A + (size_t)(iok ? i : p.Mi - 1) * p.Kd + k0 + q * 4);
            ra[4 * v] = iok ? t.x : 0.f; ra[4 * v + 1] = iok ? t.y : 0.f;
            ra[4 * v + 2] = iok ? t.z : 0.f; ra[4 * v + 3] = iok ? t.w : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + kw + 2 * r;  // wave-uniform
            int koff, ty, tx;
            if (MODE == IG_FWD) {
                int ci = kk / 9, t = kk - ci * 9;
                ty = t / 3; tx = t - ty * 3;
                koff = ci * HW + (ty - 1) * p.W + (tx - 1);
            } else {
                int co = kk / nt, tap = kk - co * nt;
                ty = tap / p.nkx; tx = tap - ty * p.nkx;
                koff = co * HoWo - ty * p.Wo - tx;
            }
            bool ok = ((ymask >> ty) & (xmask >> tx) & 1u) != 0;
            size_t addr = ok ? (size_t)((long long)base + koff) : safe;
            float v = p.Bm[addr];
            rb[r] = ok ? v : 0.f;
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            int f = tid + 256 * v;
            int il = f >> 3, q = f & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(q * 4 + e) * IG_LD + il] = ra[4 * v + e];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Bs[(kw + 2 * r) * IG_LD + (tid & 127)] = rb[r];
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

    load_stage(0);
    for (int k0 = 0; k0 < p.Kd; k0 += C2_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        load_stage(k0 + C2_KB < p.Kd ? k0 + C2_KB : k0);  // unconditional prefetch (last stage re-reads itself)
#pragma unroll
        for (int k2 = 0; k2 < C2_KB / 2; ++k2) {
            float a0 = As[(2 * k2 + kh) * IG_LD + wi + li];
            float a1 = As[(2 * k2 + kh) * IG_LD + wi + 32 + li];
            float b0 = Bs[(2 * k2 + kh) * IG_LD + wj + li];
            float b1 = Bs[(2 * k2 + kh) * IG_LD + wj + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
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
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + wi + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) p.D[col_off + (size_t)i * i_stride] = acc[ti][tj][e];
            }
        }
    }
}

// ------------------------------------------------------------------ weight repack for DGRAD
// Wd[ci][co*nt + tap] = W[co][ci][ky(tap)][kx(tap)] for one parity class
__global__ __launch_bounds__(256) void k_repack_dgrad(const float *__restrict__ W, int Cin, int Cout, int nky,
                                                      int nkx, int ky0, int kystep, int kx0, int kxstep,
                                                      float *__restrict__ Wd) {
    int nt = nky * nkx;
    int64_t total = (int64_t)Cin * Cout * nt;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int kd = (int)(e % ((int64_t)Cout * nt));
    int ci = (int)(e / ((int64_t)Cout * nt));
    int co = kd / nt, tap = kd % nt;
    int ky = ky0 + (tap / nkx) * kystep, kx = kx0 + (tap % nkx) * kxstep;
    Wd[e] = W[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
}

__global__ __launch_bounds__(256) void k_sum_splits(const float *__restrict__ partial, int64_t n, int splits,
                                                    float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = partial[i];
    for (int s = 1; s < splits; ++s) acc += partial[(size_t)s * n + i];
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

// block per ci: gW[co][ci] = sum_{b,pix} gY[b][co][pix] * X[b][ci][pix]; block ci == 0 also writes gbias
__global__ __launch_bounds__(256) void k_pw_wgrad(const float *__restrict__ X, const float *__restrict__ gY, int Bn,
                                                  int Cin, int Cout, int64_t HW, float *__restrict__ gW,
                                                  float *__restrict__ gbias) {
    __shared__ double red[2 * PW_MAXCO][4];
    const int ci = blockIdx.x;
    double acc[PW_MAXCO], accb[PW_MAXCO];
#pragma unroll
    for (int co = 0; co < PW_MAXCO; ++co) { acc[co] = 0; accb[co] = 0; }
    for (int64_t e = threadIdx.x; e < (int64_t)Bn * HW; e += 256) {
        int64_t b = e / HW, pix = e - b * HW;
        float x = X[((size_t)b * Cin + ci) * HW + pix];
#pragma unroll
        for (int co = 0; co < PW_MAXCO; ++co)
            if (co < Cout) {
                float g = gY[((size_t)b * Cout + co) * HW + pix];
                acc[co] += (double)g * (double)x;
                accb[co] += (double)g;
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
        gW[co * Cin + ci] = (float)(red[co][0] + red[co][1] + red[co][2] + red[co][3]);
        if (ci == 0 && gbias)
            gbias[co] = (float)(red[PW_MAXCO + co][0] + red[PW_MAXCO + co][1] + red[PW_MAXCO + co][2] +
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
    dim3 grid((unsigned)cdiv64(p.Nj, IG_T), (unsigned)cdiv64(p.Mi, IG_T), 1);
    k_conv_s2<IG_FWD><<<grid, 256, 0, st>>>(p);
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
    int Ho = out_dim(H, 3, 2, 1), Wo = out_dim(W, 3, 2, 1);
    float *slab = ws;
    for (int py = 0; py < 2; ++py) {
        for (int px = 0; px < 2; ++px) {
            IgParams p = {};
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
            k_repack_dgrad<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(w, Cin, Cout, p.nky, p.nkx, p.ky0, p.kystep,
                                                                         p.kx0, p.kxstep, slab);
            p.A = slab;
            slab += total;
            if (p.Nj > 0) {
                dim3 grid((unsigned)cdiv64(p.Nj, IG_T), (unsigned)cdiv64(p.Mi, IG_T), 1);
                k_conv_s2<IG_DGRAD><<<grid, 256, 0, st>>>(p);
            }
        }
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
        k_pw_wgrad<<<(unsigned)Cin, 256, 0, st>>>(x, gy, B, Cin, Cout, (int64_t)H * W, gw, gbias);
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
    int splits = (int)cdiv64(1024, tiles);
    int64_t max_by_k = cdiv64(p.Kd, 4 * IG_KB);
    if (splits > max_by_k) splits = (int)(max_by_k < 1 ? 1 : max_by_k);
    if ((int64_t)splits * slab > ws_floats) splits = (int)(ws_floats / slab);
    LIDOG_REQUIRE(splits >= 1, "conv2d_wgrad: workspace too small (%lld floats needed per split)", (long long)slab);
    p.k_chunk = (int)(cdiv64(cdiv64(p.Kd, splits), IG_KB) * IG_KB);
    splits = (int)cdiv64(p.Kd, p.k_chunk);
    p.D = (splits == 1) ? gw : ws;
    dim3 grid((unsigned)cdiv64(p.Nj, IG_T), (unsigned)cdiv64(p.Mi, IG_T), (unsigned)splits);
    k_igemm<IG_WGRAD><<<grid, 256, 0, st>>>(p);
    if (splits > 1) k_sum_splits<<<(unsigned)cdiv64(slab, 256), 256, 0, st>>>(ws, slab, splits, gw);
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
