// Conv2d(k3, s2, p1) over a STRUCTURALLY SPARSE input: the first convolution of Encoder2D reads the image made by
// sparse2super (utils/models/minkunet_bev.py:158-230), in which 95 % of the cells are empty (no voxel under the
// pooling window: exact zeros, and no gradient is needed there).  Seen through the implicit-GEMM tiles of
// conv2d.hip, only ~21 % of the (128-pixel tile, input channel) pairs contain a non-empty cell (seed-0 synthetic
// scan; scripts/bev_sparsity.py), so each tile works on the compacted list of its active channels:
//   FWD    reduction index k = (ci, tap) runs over the active channels only (a skipped term is 0 * w: the fmaf
//          chain of the remaining terms is unchanged, so the result is bit-identical to the dense kernel);
//   DGRAD  rows i = ci of the output tile are the active channels only (the gradient of an empty cell is never
//          read: sparse2super's backward routes gradients to arg-max cells); rows of inactive channels are NOT
//          written.
// The activity comes from a `support` tensor [B,Cin,H,W] int32 (>= 0 where the cell is non-empty: the arg-max
// source map of the pooling): one pass turns it into a bit per cell, a second one into per-tile channel lists.
#include <vector>

#include "conv2d.h"

// ------------------------------------------------------------------ support -> row bitmasks -> tile lists
// one wave per image row (b, c, y): bit x of word x / 64 = support[b][c][y][x] >= 0
__global__ __launch_bounds__(256) void k_support_rowbits(const int32_t *__restrict__ sup, int64_t rows, int W,
                                                         int words, uint64_t *__restrict__ bits) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int32_t *s = sup + row * W;
    for (int w = 0; w < words; ++w) {
        const int x = w * 64 + lane;
        const bool v = x < W && s[x] >= 0;
        const uint64_t bal = __ballot(v);
        if (lane == 0) bits[row * words + w] = bal;
    }
}

// Tile = 128 consecutive pixels j = (b, yc, xc) of a pixel grid Hc x Wc per image.  Pixel (yc, xc) touches the
// input rows 2*yc + ya + [0, ny) and columns 2*xc + xlo .. 2*xc + xhi; `mask` selects the column parity.
//   forward / weight gradient: the output pixels, ya = -1, ny = 3, xlo = -1, xhi = +1, every column
//   data gradient, class (py, px): the input pixels of the class, ya = py, ny = 1, xlo = xhi = px, parity px
struct ListGeom {
    int C, H, W, Hc, Wc, Nj;
    int ya, ny, xlo, xhi;
    uint64_t mask;
};

// lists[tile][0] = number of active channels, lists[tile][1..] = their ids, ascending
__device__ __forceinline__ void tile_list(const uint64_t *__restrict__ bits, int words, const ListGeom &g,
                                          int32_t *__restrict__ lists, uint64_t *__restrict__ tbits, int tile) {
    __shared__ int s_cnt[2];
    const int c = threadIdx.x;
    int j = tile * IG_T;
    const int j1 = (j + IG_T - 1 < g.Nj - 1) ? j + IG_T - 1 : g.Nj - 1;
    const int hw = g.Hc * g.Wc;
    bool act = false;
    if (c < g.C) {
        while (j <= j1) {
            const int b = j / hw, r = j - b * hw;
            const int yc = r / g.Wc, xa = r - yc * g.Wc;
            int xb = xa + (j1 - j);
            if (xb > g.Wc - 1) xb = g.Wc - 1;
            int x_lo = 2 * xa + g.xlo, x_hi = 2 * xb + g.xhi;
            if (x_lo < 0) x_lo = 0;
            if (x_hi > g.W - 1) x_hi = g.W - 1;
            for (int dy = 0; dy < g.ny; ++dy) {
                const int y = 2 * yc + g.ya + dy;
                if (y < 0 || y >= g.H) continue;
                const uint64_t *row = bits + ((size_t)(b * g.C + c) * g.H + y) * words;
                for (int w = x_lo >> 6; w <= (x_hi >> 6); ++w) {
                    uint64_t m = g.mask;
                    if (w == (x_lo >> 6)) m &= ~0ull << (x_lo & 63);
                    if (w == (x_hi >> 6)) m &= ~0ull >> (63 - (x_hi & 63));
                    if (row[w] & m) act = true;
                }
            }
            j += xb - xa + 1;
        }
    }
    const uint64_t bal = __ballot(act);
    const int lane = c & 63, wv = c >> 6;
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int32_t *lst = lists + (size_t)tile * (g.C + 1);
    if (act) lst[1 + (wv ? s_cnt[0] : 0) + pre] = c;
    if (c == 0) lst[0] = s_cnt[0] + s_cnt[1];
    if (tbits && lane == 0) tbits[(size_t)tile * 2 + wv] = bal;  // bit c of the tile's 128-bit channel mask
}

__global__ __launch_bounds__(128) void k_tile_lists(const uint64_t *__restrict__ bits, int words, ListGeom g,
                                                    int32_t *__restrict__ lists, uint64_t *__restrict__ tbits) {
    tile_list(bits, words, g, lists, tbits, (int)blockIdx.x);
}

// the four data-gradient classes in one launch (they feed the backward pass only, but are built on the forward pass's
// stream: four launches of ~870 workgroups each cost four launch latencies there): workgroup x belongs to the class
// whose tile range [first[cls], first[cls + 1]) holds it
struct ListGeom4 {
    ListGeom g[4];
    int first[5];
    int64_t off[4];
};
__global__ __launch_bounds__(128) void k_tile_lists4(const uint64_t *__restrict__ bits, int words, ListGeom4 q,
                                                     int32_t *__restrict__ act) {
    const int x = blockIdx.x;
    const int cls = x < q.first[1] ? 0 : x < q.first[2] ? 1 : x < q.first[3] ? 2 : 3;
    tile_list(bits, words, q.g[cls], act + q.off[cls], nullptr, x - q.first[cls]);
}

// Weight gradient: the columns (ci, tap) of gW are cut into groups of WA_GC channels; group g works on the pixel
// tiles in which one of its channels is active.  glists[g][0] = number of such tiles, glists[g][1..] = their ids,
// ascending (ordered compaction: ballot + prefix over the four waves, 256 tiles per round).
#ifndef WA_GC
#define WA_GC 7   // 63 of 64 MFMA columns carry data; 57 % of the (group, tile) pairs are active on a LiDAR sweep
#endif
#define WA_NT ((9 * WA_GC + 31) / 32)  // 32-column MFMA tiles per group
__global__ __launch_bounds__(256) void k_group_lists(const uint64_t *__restrict__ tbits, int n_tiles, int C,
                                                     int32_t *__restrict__ glists) {
    __shared__ int s_cnt[4];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t m0 = 0, m1 = 0;
    for (int c = g * WA_GC; c < g * WA_GC + WA_GC && c < C; ++c) {
        if (c < 64) m0 |= 1ull << c;
        else m1 |= 1ull << (c - 64);
    }
    int32_t *gl = glists + (size_t)g * (n_tiles + 1);
    int base = 0;
    for (int t0 = 0; t0 < n_tiles; t0 += 256) {
        const int t = t0 + tid;
        const bool f = t < n_tiles && ((tbits[(size_t)t * 2] & m0) | (tbits[(size_t)t * 2 + 1] & m1)) != 0;
        const uint64_t bal = __ballot(f);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += s_cnt[w];
        if (f) gl[1 + off + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (tid == 0) gl[0] = base;
}

struct ActLayout {
    int words;
    int64_t bits_off, fwd_off, tbits_off, glists_off, dgrad_off[4], total;  // in int32 elements
    int fwd_tiles, dgrad_tiles[4], groups;
};

static int out_dim2(int H) { return (H + 2 - 3) / 2 + 1; }

static ActLayout act_layout(int B, int C, int H, int W) {
    ActLayout L;
    L.words = (W + 63) / 64;
    L.bits_off = 0;
    int64_t off = 2 * (int64_t)B * C * H * L.words;  // uint64 words as int32 pairs (offset stays even)
    L.fwd_tiles = (int)cdiv64((int64_t)B * out_dim2(H) * out_dim2(W), IG_T);
    L.fwd_off = off;
    off += (int64_t)L.fwd_tiles * (C + 1);
    off += off & 1;
    L.tbits_off = off;                       // uint64 [fwd_tiles][2]
    off += 4 * (int64_t)L.fwd_tiles;
    L.groups = (C + WA_GC - 1) / WA_GC;
    L.glists_off = off;
    off += (int64_t)L.groups * (L.fwd_tiles + 1);
    for (int cls = 0; cls < 4; ++cls) {
        int py = cls >> 1, px = cls & 1;
        int Hc = (H - py + 1) / 2, Wc = (W - px + 1) / 2;
        L.dgrad_tiles[cls] = (int)cdiv64((int64_t)B * Hc * Wc, IG_T);
        L.dgrad_off[cls] = off;
        off += (int64_t)L.dgrad_tiles[cls] * (C + 1);
    }
    L.total = off;
    return L;
}

extern "C" int64_t lidog_conv2d_support_ws(int32_t B, int32_t Cin, int32_t H, int32_t W) {
    return act_layout(B, Cin, H, W).total;
}

extern "C" int lidog_conv2d_support(const int32_t *support, int32_t B, int32_t Cin, int32_t H, int32_t W,
                                    int32_t *act, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin >= 1 && Cin <= 128, "conv2d_support: 1 <= Cin <= 128");
    LIDOG_REQUIRE((int64_t)B * Cin * H * W < ((int64_t)1 << 31), "conv2d_support: tensor too large");
    if ((int64_t)B * H * W == 0) return 0;
    ActLayout L = act_layout(B, Cin, H, W);
    uint64_t *bits = reinterpret_cast<uint64_t *>(act + L.bits_off);
    int64_t rows = (int64_t)B * Cin * H;
    if (support)  // NULL: lidog_bev_pool_fwd has already written the row bitmasks
        k_support_rowbits<<<(unsigned)cdiv64(rows, 4), 256, 0, st>>>(support, rows, W, L.words, bits);
    ListGeom g;
    g.C = Cin; g.H = H; g.W = W;
    g.Hc = out_dim2(H); g.Wc = out_dim2(W); g.Nj = B * g.Hc * g.Wc;
    g.ya = -1; g.ny = 3; g.xlo = -1; g.xhi = 1; g.mask = ~0ull;
    uint64_t *tbits = reinterpret_cast<uint64_t *>(act + L.tbits_off);
    k_tile_lists<<<(unsigned)L.fwd_tiles, 128, 0, st>>>(bits, L.words, g, act + L.fwd_off, tbits);
    k_group_lists<<<(unsigned)L.groups, 256, 0, st>>>(tbits, L.fwd_tiles, Cin, act + L.glists_off);
    ListGeom4 q;
    q.first[0] = 0;
    for (int cls = 0; cls < 4; ++cls) {
        int py = cls >> 1, px = cls & 1;
        g.Hc = (H - py + 1) / 2; g.Wc = (W - px + 1) / 2; g.Nj = B * g.Hc * g.Wc;
        g.ya = py; g.ny = 1; g.xlo = px; g.xhi = px;
        g.mask = px ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
        q.g[cls] = g;
        q.off[cls] = L.dgrad_off[cls];
        q.first[cls + 1] = q.first[cls] + L.dgrad_tiles[cls];
    }
    if (q.first[4] > 0) k_tile_lists4<<<(unsigned)q.first[4], 128, 0, st>>>(bits, L.words, q, act);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ FWD over the active channels of each pixel tile
// Same tile and MFMA schedule as k_conv_s2<IG_FWD, 2, 2, 2>: D[co][pixel] = sum_k W[co][k] * im2col(X)[k][pixel],
// 128 x 128, 32-deep stages.  The reduction table of the workgroup lists (offset, tap, weight row) of the ACTIVE
// channels' 9 taps, padded to a multiple of 32 with entries whose tap never validates; the weights are read from
// the transposed copy Wt[k][co], so that a stage row is contiguous whatever k it is.
__global__ __launch_bounds__(256) void k_conv_fwd_act(IgParams p, const int32_t *__restrict__ lists,
                                                      const float *__restrict__ Wt) {
    __shared__ __attribute__((aligned(16))) float As[C2_KB * IG_LD];
    __shared__ float Bs[C2_KB * IG_LD];
    extern __shared__ int2 s_tab[];  // [Kp] (x offset, tap | weight row << 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * IG_T, j0 = blockIdx.x * IG_T;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int32_t *lst = lists + (size_t)blockIdx.x * (p.Cin + 1);
    const int n_act = lst[0];
    const int Kd = n_act * 9, Kp = (Kd + C2_KB - 1) / C2_KB * C2_KB;

    for (int kk = tid; kk < Kp; kk += 256) {
        int koff = 0, t = 15, row = 0;
        if (kk < Kd) {
            int a = kk / 9;
            t = kk - a * 9;
            int ci = lst[1 + a];
            int ty = t / 3, tx = t - ty * 3;
            koff = ci * HW + (ty - 1) * p.W + (tx - 1);
            row = ci * 9 + t;
        }
        s_tab[kk] = make_int2(koff, t | (row << 4));
    }

    const int j = j0 + (tid & 127);
    const bool jvalid = j < p.Nj;
    const int jj = jvalid ? j : 0;
    int base;
    unsigned tapmask = 0;
    {
        int pb = jj / HoWo, r = jj - pb * HoWo;
        int yo = r / p.Wo, xo = r - yo * p.Wo;
        base = pb * p.Cin * HW + (2 * yo) * p.W + 2 * xo;  // centre tap, always inside the image
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                int y = 2 * yo - 1 + ty, x = 2 * xo - 1 + tx;
                tapmask |= (unsigned)(y >= 0 && y < p.H && x >= 0 && x < p.W) << (ty * 3 + tx);
            }
    }
    if (!jvalid) tapmask = 0;
    const int safe = base;
    __syncthreads();  // table complete

    float4 ra[4];
    float rb[16];
    unsigned okbits = 0;
    const int a_kr = tid >> 5, a_c4 = (tid & 31) * 4;  // stage row / column quad of float4 v: row a_kr + 8 v
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = s_tab[k0 + a_kr + 8 * v].y >> 4;
            ra[v] = *reinterpret_cast<const float4 *>(Wt + (size_t)row * p.Cout + i0 + a_c4);
        }
        okbits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int2 e = s_tab[k0 + kw + 2 * r];  // wave-uniform address: a broadcast read
            const unsigned ok = (tapmask >> (e.y & 15)) & 1u;
            okbits |= ok << r;
            rb[r] = p.Bm[ok ? base + e.x : safe];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<float4 *>(&As[(a_kr + 8 * v) * IG_LD + a_c4]) = ra[v];
#pragma unroll
        for (int r = 0; r < 16; ++r) Bs[(kw + 2 * r) * IG_LD + (tid & 127)] = ((okbits >> r) & 1u) ? rb[r] : 0.f;
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
    const float *a_rd = &As[kh * IG_LD + wi + li];
    const float *b_rd = &Bs[kh * IG_LD + wj + li];

    if (Kp > 0) load_stage(0);
    for (int k0 = 0; k0 < Kp; k0 += C2_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        load_stage(k0 + C2_KB < Kp ? k0 + C2_KB : k0);  // unconditional prefetch (last stage re-reads itself)
        float af[2], bf[2], an[2], bn[2];
        af[0] = a_rd[0]; af[1] = a_rd[32];
        bf[0] = b_rd[0]; bf[1] = b_rd[32];
#pragma unroll
        for (int k2 = 0; k2 < C2_KB / 2; ++k2) {
            if (k2 + 1 < C2_KB / 2) {
                an[0] = a_rd[(2 * k2 + 2) * IG_LD]; an[1] = a_rd[(2 * k2 + 2) * IG_LD + 32];
                bn[0] = b_rd[(2 * k2 + 2) * IG_LD]; bn[1] = b_rd[(2 * k2 + 2) * IG_LD + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k2 + 1 < C2_KB / 2) {
                af[0] = an[0]; af[1] = an[1];
                bf[0] = bn[0]; bf[1] = bn[1];
            }
        }
    }

#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        int jo = j0 + wj + 32 * tj + li;
        if (jo >= p.Nj) continue;
        int b = jo / HoWo;
        size_t col_off = (size_t)b * p.Cout * HoWo + (jo - b * HoWo);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + wi + 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) p.D[col_off + (size_t)i * HoWo] = acc[ti][tj][e];
            }
        }
    }
}

extern "C" int lidog_conv2d_fwd_sparse(const float *x, const float *w, const int32_t *act, int32_t B, int32_t Cin,
                                       int32_t H, int32_t W, int32_t Cout, float *y, float *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin >= 1 && Cin <= 128 && Cout % IG_T == 0, "conv2d_fwd_sparse: Cin <= 128, Cout a multiple of 128");
    LIDOG_REQUIRE(ws != nullptr, "conv2d_fwd_sparse: needs a Cin*9*Cout float workspace (transposed weights)");
    IgParams p = {};
    p.Bm = x; p.D = y;
    p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Ho = out_dim2(H); p.Wo = out_dim2(W);
    p.Mi = Cout; p.Nj = B * p.Ho * p.Wo; p.Kd = Cin * 9;
    if (p.Nj == 0) return 0;
    LIDOG_REQUIRE((int64_t)B * Cin * H * W < ((int64_t)1 << 31) && (int64_t)(p.Kd + C2_KB) * 8 <= 24576,
                  "conv2d_fwd_sparse: tensor too large for 32-bit offsets / reduction table");
    ActLayout L = act_layout(B, Cin, H, W);
    // Wt[k][co] = W[co][k]
    int rc = lidog_transpose_kernel(w, 1, Cout, p.Kd, ws, stream);
    if (rc) return rc;
    dim3 grid((unsigned)L.fwd_tiles, (unsigned)(Cout / IG_T), 1);
    k_conv_fwd_act<<<grid, 256, (size_t)(p.Kd + C2_KB) * sizeof(int2), st>>>(p, act + L.fwd_off, ws);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ DGRAD for the active channels of each pixel tile
// Same schedule as k_conv_s2<IG_DGRAD, 1, 3, 1> (rows i = ci, 128 class pixels per tile, k = (co, tap)), with the
// rows of the tile = the active channels of the tile: TIA = ceil(n_act / 32) row blocks instead of Cin / 32.
template <int TIA>
__device__ __forceinline__ void conv_dgrad_act_body(const IgParams &p, const int tile, const int *s_ch, int n_act,
                                                    float *As, float *Bs, const int2 *s_tab) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = tile * IG_T;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 7);

    const int j = j0 + (tid & 127);
    const bool jvalid = j < p.Nj;
    const int jj = jvalid ? j : 0;
    int base, safe;
    unsigned tapmask = 0;
    {
        int hw = p.Hc * p.Wc;
        int pb = jj / hw, r = jj - pb * hw;
        int pyy = (r / p.Wc) * 2 + p.py, pxx = (r % p.Wc) * 2 + p.px;
        int y0 = (pyy + 1 - p.ky0) >> 1, x0 = (pxx + 1 - p.kx0) >> 1;  // output pixel of the class's first tap
        safe = pb * p.Cout * HoWo;
        base = safe + y0 * p.Wo + x0;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
                tapmask |= (unsigned)(y0 - ty >= 0 && y0 - ty < p.Ho && x0 - tx >= 0 && x0 - tx < p.Wo)
                           << (ty * 2 + tx);
    }
    if (!jvalid) tapmask = 0;

    // A rows this thread stages: float4 v covers rows 32 v + (tid >> 3)
    const float *a_ptr[TIA];
    bool a_ok[TIA];
#pragma unroll
    for (int v = 0; v < TIA; ++v) {
        int il = 32 * v + (tid >> 3);
        a_ok[v] = il < n_act;
        a_ptr[v] = p.A + (size_t)(a_ok[v] ? s_ch[il] : 0) * p.Kd + (tid & 7) * 4;
    }

    float4 ra[TIA];
    float rb[16];
    unsigned okbits = 0;
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int v = 0; v < TIA; ++v) ra[v] = *reinterpret_cast<const float4 *>(a_ptr[v] + k0);
        okbits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int2 e = s_tab[k0 + kw + 2 * r];
            const unsigned ok = (tapmask >> e.y) & 1u;
            okbits |= ok << r;
            rb[r] = p.Bm[ok ? base + e.x : safe];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int v = 0; v < TIA; ++v) {
            int il = 32 * v + (tid >> 3), q = tid & 7;
            const bool ok = a_ok[v];
            As[(q * 4 + 0) * IG_LD + il] = ok ? ra[v].x : 0.f;
            As[(q * 4 + 1) * IG_LD + il] = ok ? ra[v].y : 0.f;
            As[(q * 4 + 2) * IG_LD + il] = ok ? ra[v].z : 0.f;
            As[(q * 4 + 3) * IG_LD + il] = ok ? ra[v].w : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Bs[(kw + 2 * r) * IG_LD + (tid & 127)] = ((okbits >> r) & 1u) ? rb[r] : 0.f;
    };

    f32x16 acc[TIA];
#pragma unroll
    for (int a = 0; a < TIA; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    const int wj = wave * 32;
    const int li = lane & 31, kh = lane >> 5;
    const float *a_rd = &As[kh * IG_LD + li];
    const float *b_rd = &Bs[kh * IG_LD + wj + li];

    load_stage(0);
    for (int k0 = 0; k0 < p.Kd; k0 += C2_KB) {
        __syncthreads();
        store_stage();
        __syncthreads();
        load_stage(k0 + C2_KB < p.Kd ? k0 + C2_KB : k0);
        float af[TIA], an[TIA], bf, bn = 0.f;
#pragma unroll
        for (int a = 0; a < TIA; ++a) af[a] = a_rd[32 * a];
        bf = b_rd[0];
#pragma unroll
        for (int k2 = 0; k2 < C2_KB / 2; ++k2) {
            if (k2 + 1 < C2_KB / 2) {
#pragma unroll
                for (int a = 0; a < TIA; ++a) an[a] = a_rd[(2 * k2 + 2) * IG_LD + 32 * a];
                bn = b_rd[(2 * k2 + 2) * IG_LD];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TIA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k2 + 1 < C2_KB / 2) {
#pragma unroll
                for (int a = 0; a < TIA; ++a) af[a] = an[a];
                bf = bn;
            }
        }
    }

    const int jo = j0 + wj + li;
    if (jo < p.Nj) {
        int hw = p.Hc * p.Wc;
        int b = jo / hw, r = jo - b * hw;
        int y = (r / p.Wc) * 2 + p.py, x = (r % p.Wc) * 2 + p.px;
        size_t col_off = (size_t)b * p.Cin * HW + (size_t)y * p.W + x;
#pragma unroll
        for (int ti = 0; ti < TIA; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int il = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (il < n_act) p.D[col_off + (size_t)s_ch[il] * HW] = acc[ti][e];
            }
        }
    }
}

// the four parity classes in one launch, the four-tap class first (conv2d.hip:k_conv_s2 has the reasoning)
__global__ __launch_bounds__(256) void k_conv_dgrad_act(IgClasses pc, const int32_t *__restrict__ act) {
    __shared__ float As[C2_KB * IG_LD];
    __shared__ float Bs[C2_KB * IG_LD];
    __shared__ int s_ch[128];
    extern __shared__ int2 s_tab[];  // [Kd] (offset, tap)
    const int tid = threadIdx.x;
    int cls = 0;
    while (cls + 1 < pc.n && (int)blockIdx.x >= pc.first[cls + 1]) ++cls;
    const IgParams p = pc.c[cls];
    const int tile = (int)blockIdx.x - pc.first[cls];
    const int32_t *lst = act + pc.list_off[cls] + (size_t)tile * (p.Cin + 1);
    const int n_act = lst[0];
    if (n_act == 0) return;  // workgroup-uniform: nothing of this tile is needed
    if (tid < n_act) s_ch[tid] = lst[1 + tid];
    const int nt = p.nky * p.nkx, HoWo = p.Ho * p.Wo;
    for (int kk = tid; kk < p.Kd; kk += 256) {
        int co = kk / nt, tap = kk - co * nt;
        int ty = tap / p.nkx, tx = tap - ty * p.nkx;
        s_tab[kk] = make_int2(co * HoWo - ty * p.Wo - tx, ty * 2 + tx);
    }
    __syncthreads();
    if (n_act <= 32) conv_dgrad_act_body<1>(p, tile, s_ch, n_act, As, Bs, s_tab);
    else if (n_act <= 64) conv_dgrad_act_body<2>(p, tile, s_ch, n_act, As, Bs, s_tab);
    else if (n_act <= 96) conv_dgrad_act_body<3>(p, tile, s_ch, n_act, As, Bs, s_tab);
    else conv_dgrad_act_body<4>(p, tile, s_ch, n_act, As, Bs, s_tab);
}

extern "C" int lidog_conv2d_dgrad_sparse(const float *gy, const float *w, const int32_t *act, int32_t B, int32_t Cin,
                                         int32_t H, int32_t W, int32_t Cout, float *gx, float *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin >= 1 && Cin <= 128 && Cout % C2_KB == 0, "conv2d_dgrad_sparse: Cin <= 128, Cout % 32 == 0");
    LIDOG_REQUIRE(ws != nullptr, "conv2d_dgrad_sparse: needs a 9*Cin*Cout float workspace for the repacked weights");
    int Ho = out_dim2(H), Wo = out_dim2(W);
    LIDOG_REQUIRE((int64_t)B * Cout * Ho * Wo < ((int64_t)1 << 31) && (int64_t)Cout * 4 * 8 <= 24576 &&
                      (int64_t)B * Cin * H * W < ((int64_t)1 << 31),
                  "conv2d_dgrad_sparse: tensor too large for 32-bit offsets / reduction table");
    ActLayout L = act_layout(B, Cin, H, W);
    float *slab = ws;
    lidog_launch_repack_dgrad_all(w, Cin, Cout, ws, st);   // conv2d.hip: the four classes' weight slabs, one launch
    IgParams cls_p[4];
    for (int cls = 0; cls < 4; ++cls) {
        int py = cls >> 1, px = cls & 1;
        IgParams &p = cls_p[cls];
        p = IgParams{};
        p.Bm = gy; p.D = gx;
        p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo;
        p.py = py; p.px = px;
        p.nky = py ? 2 : 1; p.ky0 = py ? 0 : 1; p.kystep = 2;
        p.nkx = px ? 2 : 1; p.kx0 = px ? 0 : 1; p.kxstep = 2;
        p.Hc = (H - py + 1) / 2; p.Wc = (W - px + 1) / 2;
        int nt = p.nky * p.nkx;
        p.Mi = Cin; p.Nj = B * p.Hc * p.Wc; p.Kd = Cout * nt;
        int64_t total = (int64_t)Cin * Cout * nt;
        p.A = slab;
        slab += total;
    }
    IgClasses pc = {};
    size_t tab = 0;
    for (int cls : {3, 1, 2, 0}) {
        const IgParams &p = cls_p[cls];
        if (p.Nj <= 0 || L.dgrad_tiles[cls] <= 0) continue;
        pc.c[pc.n] = p;
        pc.list_off[pc.n] = (long long)L.dgrad_off[cls];
        pc.first[pc.n + 1] = pc.first[pc.n] + (int)L.dgrad_tiles[cls];
        ++pc.n;
        if ((size_t)p.Kd * sizeof(int2) > tab) tab = (size_t)p.Kd * sizeof(int2);
    }
    if (pc.n > 0) k_conv_dgrad_act<<<(unsigned)pc.first[pc.n], 256, tab, st>>>(pc, act);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ WGRAD over the active pixel tiles of each channel group
// gW[co][(ci,tap)] = sum over pixels of gY[co][pixel] * X[ci][window(pixel) + tap].  Workgroup (g, co tile, split):
// the 9 * WA_GC columns of channel group g (padded to WA_NT MFMA tiles of 32) x 128 rows, accumulated over the
// pixel tiles of glists[g] that fall to this split, 32 pixels per LDS stage; splits are summed in order afterwards
// (no atomics).  A pixel tile in which none of the group's channels is active contributes exact zeros and is
// never visited.  Narrow groups skip more tiles but re-read gY once per group (WA_GC = 3: 41 % of the tiles, twice
// the gY traffic of the dense kernel, no faster; 7: 57 %, 1.2 x the traffic; 14: 69 %, 0.7 x).
#define WA_KB 32
#define WA_TPI 16  // pixel tiles per work item
#define WA_LDA 129
#define WA_LDB (32 * WA_NT + 1)
__global__ __launch_bounds__(256) void k_conv_wgrad_act(IgParams p, const int32_t *__restrict__ glists, int n_tiles,
                                                        int splits) {
    __shared__ float As[WA_KB * WA_LDA];
    __shared__ float Bs[WA_KB * WA_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, i0 = blockIdx.y * IG_T, sp = blockIdx.z;
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const int32_t *gl = glists + (size_t)g * (n_tiles + 1);
    const int n_g = gl[0];
    // work item = WA_TPI consecutive entries of the group's tile list: equal work per workgroup whatever the
    // number of active tiles of the group (equal splits per group left the busiest groups 1.75 x the average)
    const int t_begin = sp * WA_TPI;
    if (sp > 0 && t_begin >= n_g) return;  // slab sp of this group is not summed (k_sum_group_splits)
    const int t_end = t_begin + WA_TPI < n_g ? t_begin + WA_TPI : n_g;
    const int nq = t_end > t_begin ? 4 * (t_end - t_begin) : 0;  // stages of 32 pixels
    const int kk = tid & 31, rg = tid >> 5;

    int cpk[4 * WA_NT];  // ((element offset of the tap relative to the window corner) << 4) | tap, -1 = idle column
#pragma unroll
    for (int r = 0; r < 4 * WA_NT; ++r) {
        int j = rg + 8 * r;
        int ci = g * WA_GC + j / 9, t = j % 9;
        cpk[r] = -1;
        if (j < 9 * WA_GC && ci < p.Cin) {
            int ty = t / 3, tx = t - ty * 3;
            cpk[r] = ((ci * HW + ty * p.W + tx) << 4) | t;
        }
    }
    unsigned a_mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) a_mask |= (unsigned)(i0 + rg + 8 * r < p.Mi) << r;

    float ra[16], rb[4 * WA_NT];
    unsigned okb = 0;
    bool mv = false;
    // tile ids are fetched one tile (four stages) ahead: the address chain list -> gY / X of a stage would
    // otherwise be two dependent memory latencies long, more than the MFMA phase it hides behind
    const int n_my = t_end - t_begin;
    int tile_cur = n_my > 0 ? gl[1 + t_begin] : 0;
    int tile_nxt = n_my > 1 ? gl[2 + t_begin] : tile_cur;
    auto load_stage = [&](int q) {
        if ((q & 3) == 0 && q > 0) {
            tile_cur = tile_nxt;
            const int ahead = (q >> 2) + 1;
            tile_nxt = gl[1 + t_begin + (ahead < n_my ? ahead : n_my - 1)];
        }
        const int tile = tile_cur;
        const int m = tile * IG_T + (q & 3) * WA_KB + kk;
        mv = m < p.Kd;
        const int mc = mv ? m : p.Kd - 1;
        const int pb = mc / HoWo, r_ = mc - pb * HoWo;
        const int yo = r_ / p.Wo, xo = r_ - yo * p.Wo;
        const int a_base = (pb * p.Cout + i0 + rg) * HoWo + yo * p.Wo + xo;
#pragma unroll
        for (int r = 0; r < 16; ++r) ra[r] = p.A[a_base + (((a_mask >> r) & 1u) ? r * 8 * HoWo : 0)];
        const int b_base = pb * p.Cin * HW + (2 * yo - 1) * p.W + 2 * xo - 1;  // window corner (may lie outside)
        const unsigned ym = (unsigned)(yo > 0) | 2u | ((unsigned)(2 * yo + 1 < p.H) << 2);
        const unsigned xm = (unsigned)(xo > 0) | 2u | ((unsigned)(2 * xo + 1 < p.W) << 2);
        const unsigned tapmask = mv ? (((ym & 1u) ? xm : 0u) | ((ym & 2u) ? xm << 3 : 0u) | ((ym & 4u) ? xm << 6 : 0u)) : 0u;
        okb = 0;
#pragma unroll
        for (int r = 0; r < 4 * WA_NT; ++r) {
            const int c = cpk[r];
            const unsigned ok = (c >= 0) ? ((tapmask >> (c & 15)) & 1u) : 0u;
            okb |= ok << r;
            rb[r] = p.Bm[ok ? b_base + (c >> 4) : 0];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) As[kk * WA_LDA + rg + 8 * r] = (mv && ((a_mask >> r) & 1u)) ? ra[r] : 0.f;
#pragma unroll
        for (int r = 0; r < 4 * WA_NT; ++r) Bs[kk * WA_LDB + rg + 8 * r] = ((okb >> r) & 1u) ? rb[r] : 0.f;
    };

    f32x16 acc[WA_NT];
#pragma unroll
    for (int t = 0; t < WA_NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int li = lane & 31, kh = lane >> 5;
    const float *a_rd = &As[kh * WA_LDA + 32 * wave + li];
    const float *b_rd = &Bs[kh * WA_LDB + li];

    if (nq > 0) load_stage(0);
    for (int q = 0; q < nq; ++q) {
        __syncthreads();
        store_stage();
        __syncthreads();
        load_stage(q + 1 < nq ? q + 1 : q);  // unconditional prefetch (the last stage re-reads itself)
        float af = a_rd[0], an = 0.f, bf[WA_NT], bn[WA_NT];
#pragma unroll
        for (int t = 0; t < WA_NT; ++t) { bf[t] = b_rd[32 * t]; bn[t] = 0.f; }
#pragma unroll
        for (int k2 = 0; k2 < WA_KB / 2; ++k2) {
            if (k2 + 1 < WA_KB / 2) {
                an = a_rd[(2 * k2 + 2) * WA_LDA];
#pragma unroll
                for (int t = 0; t < WA_NT; ++t) bn[t] = b_rd[(2 * k2 + 2) * WA_LDB + 32 * t];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < WA_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            af = an;
#pragma unroll
            for (int t = 0; t < WA_NT; ++t) bf[t] = bn[t];
        }
    }

#pragma unroll
    for (int t = 0; t < WA_NT; ++t) {
        const int j = 32 * t + li;
        const int ci = g * WA_GC + j / 9;
        if (j < 9 * WA_GC && ci < p.Cin) {
            float *d = p.D + (size_t)sp * p.Mi * p.Nj + (size_t)ci * 9 + j % 9;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (i < p.Mi) d[(size_t)i * p.Nj] = acc[t][e];
            }
        }
    }
}

// gw[co][col] = sum of the slabs of the column's channel group, in order: ceil(n_g / WA_TPI) of them (at least one)
__global__ __launch_bounds__(256) void k_sum_group_splits(const float *__restrict__ partial, int64_t n, int Nj,
                                                          const int32_t *__restrict__ glists, int n_tiles,
                                                          float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % Nj);
    const int g = col / (9 * WA_GC);
    int ns = (glists[(size_t)g * (n_tiles + 1)] + WA_TPI - 1) / WA_TPI;
    if (ns < 1) ns = 1;
    float acc = partial[i];
    for (int s = 1; s < ns; ++s) acc += partial[(size_t)s * n + i];
    out[i] = acc;
}

extern "C" int64_t lidog_conv2d_wgrad_sparse_ws(int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout) {
    ActLayout L = act_layout(B, Cin, H, W);
    int64_t splits = cdiv64(L.fwd_tiles, WA_TPI);
    return (splits < 1 ? 1 : splits) * (int64_t)Cout * Cin * 9;
}

extern "C" int lidog_conv2d_wgrad_sparse(const float *x, const float *gy, const int32_t *act, int32_t B, int32_t Cin,
                                         int32_t H, int32_t W, int32_t Cout, float *gw, float *ws, int64_t ws_floats,
                                         void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(Cin >= 1 && Cin <= 128 && Cout % 8 == 0, "conv2d_wgrad_sparse: Cin <= 128, Cout % 8 == 0");
    IgParams p = {};
    p.A = gy; p.Bm = x;
    p.Bn = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Ho = out_dim2(H); p.Wo = out_dim2(W);
    p.Mi = Cout; p.Nj = Cin * 9; p.Kd = B * p.Ho * p.Wo;
    LIDOG_REQUIRE((int64_t)Cin * H * W < ((int64_t)1 << 27) && (int64_t)B * Cin * H * W < ((int64_t)1 << 31) &&
                      (int64_t)B * Cout * p.Ho * p.Wo < ((int64_t)1 << 31),
                  "conv2d_wgrad_sparse: tensor too large for the packed 32-bit offsets");
    const int64_t slab = (int64_t)p.Mi * p.Nj;
    if (p.Kd == 0) return hipMemsetAsync(gw, 0, sizeof(float) * slab, st) == hipSuccess ? 0 : 1;
    ActLayout L = act_layout(B, Cin, H, W);
    const int itiles = (int)cdiv64(p.Mi, IG_T);
    int splits = (int)cdiv64(L.fwd_tiles, WA_TPI);
    if (splits < 1) splits = 1;
    LIDOG_REQUIRE(ws != nullptr && (int64_t)splits * slab <= ws_floats,
                  "conv2d_wgrad_sparse: workspace too small (lidog_conv2d_wgrad_sparse_ws floats needed)");
    p.D = ws;
    dim3 grid((unsigned)L.groups, (unsigned)itiles, (unsigned)splits);
    k_conv_wgrad_act<<<grid, 256, 0, st>>>(p, act + L.glists_off, L.fwd_tiles, splits);
    k_sum_group_splits<<<(unsigned)cdiv64(slab, 256), 256, 0, st>>>(ws, slab, p.Nj, act + L.glists_off, L.fwd_tiles,
                                                                    gw);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// Matrix work the three support-restricted kernels EXECUTE for the lists in `act` (bench.py's `bev_mfma_frac`: executed
// FLOPs, not the dense-equivalent figure): out[0] forward, out[1] data gradient, out[2] weight gradient, in FLOPs.
//   forward: per pixel tile 2 * 128 pixels * Cout * (9 * active channels, padded to the 32-deep stage);
//   data gradient: per class tile 2 * 128 pixels * (active channels padded to a 32-row block) * Cout * taps of the class;
//   weight gradient: per (channel group, active pixel tile) 2 * 128 pixels * Cout * (WA_NT * 32 columns).
// Reads the list headers back (synchronises with `stream`): measurement only, never on the training path.
extern "C" int lidog_conv2d_support_work(const int32_t *act, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                                         double *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(act && out && Cin >= 1 && Cin <= 128, "conv2d_support_work: bad arguments");
    ActLayout L = act_layout(B, Cin, H, W);
    out[0] = out[1] = out[2] = 0.0;
    std::vector<int32_t> head;
    auto heads = [&](int64_t off, int tiles, int pitch) -> int {
        head.assign((size_t)(tiles > 0 ? tiles : 1), 0);
        if (tiles <= 0) return 0;
        LIDOG_CHECK_HIP(hipMemcpy2DAsync(head.data(), sizeof(int32_t), act + off, (size_t)pitch * sizeof(int32_t),
                                         sizeof(int32_t), (size_t)tiles, hipMemcpyDeviceToHost, st));
        LIDOG_CHECK_HIP(hipStreamSynchronize(st));
        return 0;
    };
    if (int rc = heads(L.fwd_off, L.fwd_tiles, Cin + 1)) return rc;
    for (int t = 0; t < L.fwd_tiles; ++t) {
        const int Kp = (head[t] * 9 + C2_KB - 1) / C2_KB * C2_KB;
        out[0] += 2.0 * IG_T * Cout * Kp;
    }
    for (int cls = 0; cls < 4; ++cls) {
        const int nt = ((cls >> 1) ? 2 : 1) * ((cls & 1) ? 2 : 1);
        if (int rc = heads(L.dgrad_off[cls], L.dgrad_tiles[cls], Cin + 1)) return rc;
        for (int t = 0; t < L.dgrad_tiles[cls]; ++t)
            out[1] += 2.0 * IG_T * ((head[t] + 31) / 32 * 32) * (double)Cout * nt;
    }
    if (int rc = heads(L.glists_off, L.groups, L.fwd_tiles + 1)) return rc;
    for (int g = 0; g < L.groups; ++g) out[2] += 2.0 * IG_T * (double)head[g] * Cout * (WA_NT * 32);
    return 0;
}
