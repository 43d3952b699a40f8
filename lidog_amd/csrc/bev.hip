// BEV projection of sparse voxel features, fused: winner map (last row wins) + the reference's
// [H,W,C] -> view(1,C,H,W) memory reinterpretation + MaxPool2d, without the dense [H,W,C] tensor.
// Reference: MinkUNetBaseBEV.filter_bounds / sparse2super, utils/models/minkunet_bev.py:158-230.
// CPU restatement: oracle/ref_torch.py:sparse2super_ref.
#include "common.h"

__global__ __launch_bounds__(256) void k_bev_winner(const int4 *__restrict__ coords, int64_t n,
                                                    const int32_t *__restrict__ lut_x,
                                                    const int32_t *__restrict__ lut_y, int lut_lo, int lut_n, int H,
                                                    int W, int32_t *winner, int32_t *__restrict__ pixel) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    int ix = c.y - lut_lo, iy = c.z - lut_lo;
    int pix = -1;
    if (ix >= 0 && ix < lut_n && iy >= 0 && iy < lut_n) {
        int px = lut_x[ix], py = lut_y[iy];
        if (px >= 0 && py >= 0 && px < W && py < H) {
            pix = (c.x * H + py) * W + px;
            atomicMax(&winner[pix], (int32_t)i);  // sequential index_put_: the LAST row wins
        }
    }
    pixel[i] = pix;
}

extern "C" int lidog_bev_winner(const int32_t *coords, int64_t n, const int32_t *lut_x, const int32_t *lut_y,
                                int32_t lut_lo, int32_t lut_n, int32_t H, int32_t W, int32_t *winner, int32_t *pixel,
                                void *stream) {
    if (n == 0) return 0;
    k_bev_winner<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, lut_x, lut_y,
                                                                          lut_lo, lut_n, H, W, winner, pixel);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// out[b][c'][yo][xo] = max over the pool window of V[c'][yy][xx], V = the [H,W,C] image of sample b
// read through view(C,H,W): flat f = c'*H*W + yy*W + xx  ->  pixel f / C, feature channel f % C.
// Ties keep the first cell in window scan order (torch max_pool2d uses a strict >).  PK >= pool kernel size.
struct PoolGeom {
    int C, H, W, pk, ps, pp, Ho, Wo, w_div_c, w_mod_c;
    uint64_t c_magic;  // ceil(2^40 / C): x / C = (x * c_magic) >> 40, exact while x * C < 2^40
};

template <int PK>
__device__ __forceinline__ void pool_one(const float *__restrict__ feats, const int32_t *__restrict__ win,
                                         const PoolGeom &g, int cp, int yo, int xo, float &best, int32_t &src) {
    const int C = g.C, H = g.H, W = g.W, pk = g.pk;
    const unsigned HW = (unsigned)H * (unsigned)W;
    best = -INFINITY;
    src = -1;
    // The window's cells of one view row are consecutive flat indices: they fall on one pixel (or two at a
    // multiple of C), so the pixel's winner is looked up once per row, not once per cell.  All rows' lookups are
    // issued first, unconditionally (index clamped): a load behind a data-dependent branch costs one memory
    // latency per window row.  One division per output (by multiplication); the rows below advance (pixel,
    // channel) by (W / C, W % C) with adds only -- 32-bit integer multiplies are quarter rate.
    int x_lo = xo * g.ps - g.pp, x_hi = x_lo + pk;
    if (x_lo < 0) x_lo = 0;
    if (x_hi > W) x_hi = W;
    const int seg = x_hi - x_lo;
    const int y0 = yo * g.ps - g.pp;
    const int dy_lo = y0 < 0 ? -y0 : 0;  // first window row inside the image
    const int n_rows = (pk < H - y0 ? pk : H - y0) - dy_lo;
    const unsigned f0 = (unsigned)cp * HW + __umul24((unsigned)(y0 + dy_lo), (unsigned)W) + (unsigned)x_lo;
    unsigned pix = (unsigned)(((uint64_t)f0 * g.c_magic) >> 40);
    int ch = (int)(f0 - __umul24(pix, (unsigned)C));
    int rows[PK], chs[PK];
    unsigned pixs[PK];
#pragma unroll
    for (int j = 0; j < PK; ++j) {
        pixs[j] = pix;
        chs[j] = ch;
        rows[j] = -1;
        if (j < pk) rows[j] = win[pix < HW ? pix : HW - 1u];  // uniform predicate
        pix += (unsigned)g.w_div_c;
        ch += g.w_mod_c;
        if (ch >= C) {
            ch -= C;
            ++pix;
        }
    }
#pragma unroll
    for (int j = 0; j < PK; ++j) {
        if (j >= n_rows || seg <= 0) continue;
        int row = rows[j];
        ch = chs[j];
        if (ch + seg <= C) {
            if (row < 0) {
                // empty pixel covering the whole row segment: its cells are zeros
                if (0.f > best) {
                    best = 0.f;
                    src = -1;
                }
            } else {
                // occupied pixel: its cells are fetched together, then compared in scan order
                const int32_t base = row * C + ch;
                float v[PK];
#pragma unroll
                for (int i = 0; i < PK; ++i) v[i] = feats[base + (i < seg ? i : 0)];
#pragma unroll
                for (int i = 0; i < PK; ++i)
                    if (i < seg && v[i] > best) {
                        best = v[i];
                        src = base + i;
                    }
            }
            continue;
        }
        // the segment straddles two pixels (rare): cell by cell
        pix = pixs[j];
        for (int xx = x_lo; xx < x_hi; ++xx) {
            float v = 0.f;
            int32_t s = -1;
            if (row >= 0) {
                s = row * C + ch;
                v = feats[(int64_t)s];
            }
            if (v > best) {
                best = v;
                src = s;
            }
            if (++ch == C) {
                ch = 0;
                ++pix;
                if (xx + 1 < x_hi) row = win[pix];
            }
        }
    }
}

// Sparse formulation: 98 % of the BEV pixels are empty, and a window without an occupied pixel pools to
// (0, -1).  The output is therefore filled with (0, -1) by two memsets and only the windows that contain a
// cell of an occupied pixel are computed: one wave per voxel row; the row that owns its pixel (the winner)
// enumerates the windows covering its C cells -- one or two runs along x in the viewed image -- and computes
// each of them completely (several winners may compute the same window: they write the same value).
template <int PK>
__global__ __launch_bounds__(256) void k_bev_pool_fwd_sparse(const float *__restrict__ feats,
                                                             const int32_t *__restrict__ winner,
                                                             const int32_t *__restrict__ pixel, int64_t n,
                                                             PoolGeom g, float *__restrict__ out,
                                                             int32_t *__restrict__ argsrc,
                                                             unsigned long long *__restrict__ rowbits, int words) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int pixi = pixel[i];
    if (pixi < 0 || winner[pixi] != (int32_t)i) return;
    const unsigned HW = (unsigned)g.H * (unsigned)g.W;
    const unsigned b = (unsigned)pixi / HW;
    const unsigned q = (unsigned)pixi - b * HW;
    const int32_t *win = winner + (size_t)b * HW;
    unsigned f = q * (unsigned)g.C;  // first cell of the pixel in the viewed image (C*H*W < 2^31)
    int left = g.C;
    while (left > 0) {
        const unsigned cp = f / HW;
        const unsigned r = f - cp * HW;
        const int yy = (int)(r / (unsigned)g.W);
        const int xx0 = (int)(r - (unsigned)yy * (unsigned)g.W);
        const int len = left < g.W - xx0 ? left : g.W - xx0;
        const int xx1 = xx0 + len - 1;
        // windows (yo, xo) with yo*ps-pp <= yy <= yo*ps-pp+pk-1, likewise in x
        int t0 = yy + g.pp - g.pk + 1, t1 = xx0 + g.pp - g.pk + 1;
        const int yo_lo = t0 > 0 ? (t0 + g.ps - 1) / g.ps : 0;
        const int xo_lo = t1 > 0 ? (t1 + g.ps - 1) / g.ps : 0;
        int yo_hi = (yy + g.pp) / g.ps, xo_hi = (xx1 + g.pp) / g.ps;
        if (yo_hi > g.Ho - 1) yo_hi = g.Ho - 1;
        if (xo_hi > g.Wo - 1) xo_hi = g.Wo - 1;
        const int nx = xo_hi - xo_lo + 1, ny = yo_hi - yo_lo + 1;
        if (nx > 0 && ny > 0) {
            if (rowbits && lane < ny) {
                // structural support of the image for the convolution that reads it (conv2d_sparse.hip): one bit per
                // computed window, row (b, cp, yo) = words of 64 columns; lane l marks row yo_lo + l
                unsigned long long *rw = rowbits + (((size_t)b * g.C + cp) * g.Ho + (yo_lo + lane)) * words;
                for (int w = xo_lo >> 6; w <= (xo_hi >> 6); ++w) {
                    unsigned long long m = ~0ull;
                    if (w == (xo_lo >> 6)) m &= ~0ull << (xo_lo & 63);
                    if (w == (xo_hi >> 6)) m &= ~0ull >> (63 - (xo_hi & 63));
                    atomicOr(&rw[w], m);
                }
            }
            float *o = out + ((size_t)b * g.C + cp) * ((size_t)g.Ho * g.Wo);
            int32_t *a = argsrc + ((size_t)b * g.C + cp) * ((size_t)g.Ho * g.Wo);
            for (int yo = yo_lo; yo <= yo_hi; ++yo)
                for (int xo = xo_lo + lane; xo <= xo_hi; xo += 64) {
                    float best;
                    int32_t src;
                    pool_one<PK>(feats, win, g, (int)cp, yo, xo, best, src);
                    o[(size_t)yo * g.Wo + xo] = best;
                    a[(size_t)yo * g.Wo + xo] = src;
                }
        }
        f += (unsigned)len;
        left -= len;
    }
}

extern "C" int lidog_bev_pool_fwd(const float *feats, int32_t C, const int32_t *winner, const int32_t *pixel,
                                  int64_t n, int32_t B, int32_t H, int32_t W, int32_t pk, int32_t ps, int32_t pp,
                                  int32_t Ho, int32_t Wo, float *out, int32_t *argsrc, uint64_t *rowbits,
                                  void *stream) {
    int64_t total = (int64_t)B * C * Ho * Wo;
    if (total == 0) return 0;
    LIDOG_REQUIRE((int64_t)H * W * C < ((int64_t)1 << 31), "bev_pool_fwd: C*H*W must stay below 2^31");
    LIDOG_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31), "bev_pool_fwd: B*H*W must stay below 2^31");
    LIDOG_REQUIRE(C >= 1 && C <= 512, "bev_pool_fwd: 1 <= C <= 512");
    LIDOG_REQUIRE(pk >= 1 && pk <= 8 && ps >= 1, "bev_pool_fwd: pool kernel size must be 1..8");
    LIDOG_REQUIRE(2 * pp <= pk, "bev_pool_fwd: padding must be at most half the kernel size (as torch requires)");
    LIDOG_REQUIRE((int64_t)H * W < ((int64_t)1 << 24), "bev_pool_fwd: H*W must stay below 2^24");
    hipStream_t st = (hipStream_t)stream;
    // windows without an occupied pixel: max over zeros = 0, no source cell
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)total, st) != hipSuccess) return 1;
    // with row bitmasks argsrc is only ever read where a bit is set (lidog_bev_pool_bwd), i.e. where it was written
    if (!rowbits && hipMemsetAsync(argsrc, 0xff, sizeof(int32_t) * (size_t)total, st) != hipSuccess) return 1;
    const int words = (Wo + 63) / 64;
    if (rowbits && hipMemsetAsync(rowbits, 0, sizeof(uint64_t) * (size_t)B * C * Ho * words, st) != hipSuccess) return 1;
    if (n == 0) return 0;
    LIDOG_REQUIRE(rowbits == nullptr || pk + ps <= 64, "bev_pool_fwd: row bitmasks need pool kernel + stride <= 64");
    PoolGeom g;
    g.C = C; g.H = H; g.W = W; g.pk = pk; g.ps = ps; g.pp = pp; g.Ho = Ho; g.Wo = Wo;
    g.w_div_c = W / C; g.w_mod_c = W % C;
    g.c_magic = (((uint64_t)1 << 40) + (uint64_t)C - 1) / (uint64_t)C;
    const unsigned grid = (unsigned)cdiv64(n, 4);
#define BEV_POOL_LAUNCH(PK)                                                                                     \
    k_bev_pool_fwd_sparse<PK><<<grid, 256, 0, st>>>(feats, winner, pixel, n, g, out, argsrc,                    \
                                                    reinterpret_cast<unsigned long long *>(rowbits), words)
    if (pk <= 3) BEV_POOL_LAUNCH(3);
    else if (pk <= 5) BEV_POOL_LAUNCH(5);
    else BEV_POOL_LAUNCH(8);
#undef BEV_POOL_LAUNCH
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// Backward of the fused scatter + view + max-pool, as a GATHER (no atomics: bit-reproducible).  One thread per
// (voxel row i, channel c): the row's pixel belongs to the winner row w (index_put's backward is a gather: EVERY row that
// targets a pixel receives that pixel's gradient); cell (w, c) sits at flat index f = q*C + c of the viewed image
// -> (c', yy, xx); it can be the arg-max of the at most ceil(pk/ps)^2 windows that cover it, whose gradients are
// added in ascending (yo, xo) order when argsrc says this cell was their maximum.  Every window covering a cell of an
// occupied pixel was computed by the forward pass, so argsrc is defined wherever it is read here.
__global__ __launch_bounds__(256) void k_bev_pool_bwd_gather(const float *__restrict__ gout,
                                                             const int32_t *__restrict__ argsrc,
                                                             const int32_t *__restrict__ winner,
                                                             const int32_t *__restrict__ pixel, int64_t n, PoolGeom g,
                                                             float *__restrict__ gfeats) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * g.C) return;
    const int64_t i = idx / g.C;
    const int c = (int)(idx - i * g.C);
    const int pixi = pixel[i];
    float acc = 0.f;
    if (pixi >= 0) {
        const unsigned HW = (unsigned)g.H * (unsigned)g.W;
        const unsigned b = (unsigned)pixi / HW;
        const unsigned q = (unsigned)pixi - b * HW;
        const int32_t cell = winner[pixi] * g.C + c;
        const unsigned f = q * (unsigned)g.C + (unsigned)c;
        const unsigned cp = f / HW;
        const unsigned r = f - cp * HW;
        const int yy = (int)(r / (unsigned)g.W);
        const int xx = (int)(r - (unsigned)yy * (unsigned)g.W);
        const int t0 = yy + g.pp - g.pk + 1, t1 = xx + g.pp - g.pk + 1;
        const int yo_lo = t0 > 0 ? (t0 + g.ps - 1) / g.ps : 0;
        const int xo_lo = t1 > 0 ? (t1 + g.ps - 1) / g.ps : 0;
        int yo_hi = (yy + g.pp) / g.ps, xo_hi = (xx + g.pp) / g.ps;
        if (yo_hi > g.Ho - 1) yo_hi = g.Ho - 1;
        if (xo_hi > g.Wo - 1) xo_hi = g.Wo - 1;
        const size_t plane = ((size_t)b * g.C + cp) * ((size_t)g.Ho * g.Wo);
        for (int yo = yo_lo; yo <= yo_hi; ++yo)
            for (int xo = xo_lo; xo <= xo_hi; ++xo) {
                const size_t e = plane + (size_t)yo * g.Wo + xo;
                if (argsrc[e] == cell) acc += gout[e];
            }
    }
    gfeats[idx] = acc;
}

extern "C" int lidog_bev_pool_bwd(const float *gout, const int32_t *argsrc, const int32_t *winner,
                                  const int32_t *pixel, int64_t n, int32_t C, int32_t B, int32_t H, int32_t W,
                                  int32_t pk, int32_t ps, int32_t pp, int32_t Ho, int32_t Wo, float *gfeats,
                                  void *stream) {
    hipStream_t st = (hipStream_t)stream;
    (void)B;
    if (n == 0) return 0;
    LIDOG_REQUIRE((int64_t)H * W * C < ((int64_t)1 << 31), "bev_pool_bwd: C*H*W must stay below 2^31");
    LIDOG_REQUIRE(pk >= 1 && pk <= 8 && ps >= 1 && 2 * pp <= pk, "bev_pool_bwd: bad pooling geometry");
    PoolGeom g;
    g.C = C; g.H = H; g.W = W; g.pk = pk; g.ps = ps; g.pp = pp; g.Ho = Ho; g.Wo = Wo;
    g.w_div_c = W / C; g.w_mod_c = W % C;
    g.c_magic = (((uint64_t)1 << 40) + (uint64_t)C - 1) / (uint64_t)C;
    k_bev_pool_bwd_gather<<<(unsigned)cdiv64(n * C, 256), 256, 0, st>>>(gout, argsrc, winner, pixel, n, g, gfeats);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
