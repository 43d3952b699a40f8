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
// Ties keep the first cell in window scan order (torch max_pool2d uses a strict >).
__global__ __launch_bounds__(256) void k_bev_pool_fwd(const float *__restrict__ feats, int C,
                                                      const int32_t *__restrict__ winner, int H, int W, int pk,
                                                      int ps, int pp, int Ho, int Wo, int64_t total,
                                                      float *__restrict__ out, int32_t *__restrict__ argsrc) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int xo = (int)(e % Wo);
    int64_t t = e / Wo;
    int yo = (int)(t % Ho);
    t /= Ho;
    int cp = (int)(t % C);
    int b = (int)(t / C);
    const int64_t HW = (int64_t)H * W;
    const int32_t *win = winner + (int64_t)b * HW;
    float best = -INFINITY;
    int32_t src = -1;
    for (int dy = 0; dy < pk; ++dy) {
        int yy = yo * ps - pp + dy;
        if (yy < 0 || yy >= H) continue;
        // the window's cells of one view row are consecutive flat indices: they fall on one pixel (or two at a
        // multiple of C), so the pixel's winner is looked up once per row, not once per cell
        int x_lo = xo * ps - pp, x_hi = x_lo + pk;
        if (x_lo < 0) x_lo = 0;
        if (x_hi > W) x_hi = W;
        // 32-bit index math (the launcher checks C*H*W < 2^31): a 64-bit division costs ~100 instructions
        unsigned f = (unsigned)cp * (unsigned)HW + (unsigned)yy * (unsigned)W + (unsigned)x_lo;
        unsigned pix = f / (unsigned)C;
        int ch = (int)(f - pix * (unsigned)C);
        int row = win[pix];
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
    out[e] = best;
    argsrc[e] = src;
}

extern "C" int lidog_bev_pool_fwd(const float *feats, int32_t C, const int32_t *winner, int32_t B, int32_t H,
                                  int32_t W, int32_t pk, int32_t ps, int32_t pp, int32_t Ho, int32_t Wo, float *out,
                                  int32_t *argsrc, void *stream) {
    int64_t total = (int64_t)B * C * Ho * Wo;
    if (total == 0) return 0;
    LIDOG_REQUIRE((int64_t)H * W * C < ((int64_t)1 << 31), "bev_pool_fwd: C*H*W must stay below 2^31");
    k_bev_pool_fwd<<<(unsigned)cdiv64(total, 256), 256, 0, (hipStream_t)stream>>>(feats, C, winner, H, W, pk, ps, pp, Ho,
                                                                                Wo, total, out, argsrc);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void k_bev_pool_bwd_cells(const float *__restrict__ gout,
                                                            const int32_t *__restrict__ argsrc, int64_t total,
                                                            float *gcell) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int32_t s = argsrc[e];
    if (s >= 0) atomicAdd(&gcell[s], gout[e]);
}

// index_put's backward is a gather: EVERY row that targets a pixel receives that pixel's gradient
__global__ __launch_bounds__(256) void k_bev_pool_bwd_rows(const float *__restrict__ gcell,
                                                           const int32_t *__restrict__ winner,
                                                           const int32_t *__restrict__ pixel, int64_t n, int C,
                                                           float *__restrict__ gfeats) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * C) return;
    int64_t i = idx / C;
    int c = (int)(idx % C);
    int pix = pixel[i];
    float g = 0.f;
    if (pix >= 0) g = gcell[(int64_t)winner[pix] * C + c];
    gfeats[idx] = g;
}

extern "C" int lidog_bev_pool_bwd(const float *gout, const int32_t *argsrc, int64_t n_out_elems,
                                  const int32_t *winner, const int32_t *pixel, int64_t n, int32_t C, float *gcell,
                                  float *gfeats, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_out_elems)
        k_bev_pool_bwd_cells<<<(unsigned)cdiv64(n_out_elems, 256), 256, 0, st>>>(gout, argsrc, n_out_elems, gcell);
    if (n)
        k_bev_pool_bwd_rows<<<(unsigned)cdiv64(n * C, 256), 256, 0, st>>>(gcell, winner, pixel, n, C, gfeats);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
