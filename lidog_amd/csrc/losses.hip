// DICE losses of the LiDOG step, one pass forward and one pass backward over the logits.
// Reference: utils/losses/losses.py:56-97 (DICELoss), :100-187 (SoftDICELoss), called from
// utils/pipelines/trainer_lighting_2d.py:172-190 on the semantic logits [N,7] and on the BEV logits
// (NCHW read through .view(-1, 7)).  The torch formulation costs ~30 elementwise/reduction launches per loss,
// several of them column reductions of a 7-column matrix; here: softmax + the four per-class sums in one kernel
// (per-workgroup fp64 partials added in a fixed order: reproducible), a one-block finalize that also emits the
// per-class coefficients of the gradient, and one kernel for d loss / d logits.
//   loss = 1 - sum_c present_c * 2 I_c / U_c / (sum_c present_c + 1e-12),
//   I_c = sum_r p_rc t_rc,  U_c = sum_r (p_rc^2 or p_rc) + sum_r t_rc + 1e-12,
//   t = one-hot (DICE) or label-smoothed one-hot (soft: 1-eps / eps/(C-1)); rows with the ignore label are skipped.
#include "common.h"

#define DL_MAXC 20
#define DL_MAX_BLOCKS 1024

template <int C>
__device__ __forceinline__ void softmax_row(const float *__restrict__ x, float (&p)[C]) {
    float m = x[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        p[c] = expf(x[c] - m);
        s += p[c];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
}

// partial[block][4][C] = (I, sum p^2 or p, sum t, count of rows of the class)
template <int C>
__global__ __launch_bounds__(256) void k_dice_sums(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                   int64_t n, int64_t ignore, int has_ignore, float t_on, float t_off,
                                                   int powerize, double *__restrict__ partial) {
    __shared__ double red[4][4 * C];
    float acc[4 * C];
#pragma unroll
    for (int i = 0; i < 4 * C; ++i) acc[i] = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int64_t t = target[r];
        if (has_ignore && t == ignore) continue;
        float p[C];
        softmax_row<C>(logits + r * C, p);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const bool on = (t == c);
            const float tw = on ? t_on : t_off;
            acc[c] += p[c] * tw;
            acc[C + c] += powerize ? p[c] * p[c] : p[c];
            acc[2 * C + c] += tw;
            acc[3 * C + c] += on ? 1.f : 0.f;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4 * C; ++i) {
        double v = (double)acc[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4 * C)
        partial[(size_t)blockIdx.x * 4 * C + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one block: sums[4C] over the partials in block order; loss; coef[2C] = (a_c, b_c) with
// d loss / d p_rc = a_c * t_rc + b_c * (2 p_rc or 1)
__global__ __launch_bounds__(1024) void k_dice_finish(const double *__restrict__ partial, int nb, int C, int use_tmask,
                                                     float offset, float *__restrict__ loss, float *__restrict__ coef) {
    // 8 lanes per column: lane l adds the partials b = l, l + 8, ...; the 8 lane sums are combined in lane order
    // (fixed order: bit-reproducible).  One thread per column took 56 us on the step's forward -> backward seam.
    __shared__ double s[4 * DL_MAXC];
    __shared__ double part[8][4 * DL_MAXC];
    const int i = threadIdx.x >> 3, l = threadIdx.x & 7;
    if (i < 4 * C) {
        double v = 0;
        for (int b = l; b < nb; b += 8) v += partial[(size_t)b * 4 * C + i];
        part[l][i] = v;
    }
    __syncthreads();
    if (l == 0 && i < 4 * C) {
        double v = part[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) v += part[q][i];
        s[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double np = 0, acc = 0;
        for (int c = 0; c < C; ++c) {
            const double pres = use_tmask ? (s[3 * C + c] > 0 ? 1.0 : 0.0) : 1.0;
            const double uni = s[C + c] + s[2 * C + c] + 1e-12;
            np += pres;
            acc += pres * 2.0 * s[c] / uni;
        }
        const double den = np + 1e-12;
        loss[0] = (float)(1.0 - acc / den) + offset;
        for (int c = 0; c < C; ++c) {
            const double pres = use_tmask ? (s[3 * C + c] > 0 ? 1.0 : 0.0) : 1.0;
            const double uni = s[C + c] + s[2 * C + c] + 1e-12;
            coef[c] = (float)(-2.0 * pres / den / uni);
            coef[C + c] = (float)(2.0 * pres / den * s[c] / (uni * uni));
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_dice_bwd(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                  int64_t n, int64_t ignore, int has_ignore, float t_on, float t_off,
                                                  int powerize, const float *__restrict__ coef,
                                                  const float *__restrict__ gout, float *__restrict__ glogits) {
    float a[C], b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        a[c] = coef[c];
        b[c] = coef[C + c];
    }
    const float go = gout[0];
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int64_t t = target[r];
        float *g = glogits + r * C;
        if (has_ignore && t == ignore) {
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] = 0.f;
            continue;
        }
        float p[C], gp[C];
        softmax_row<C>(logits + r * C, p);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float tw = (t == c) ? t_on : t_off;
            gp[c] = a[c] * tw + b[c] * (powerize ? 2.f * p[c] : 1.f);
            dot += p[c] * gp[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = go * p[c] * (gp[c] - dot);
    }
}

static int dice_blocks(int64_t n) {
    int64_t nb = cdiv64(n, 256 * 4);
    if (nb < 1) nb = 1;
    return (int)(nb > DL_MAX_BLOCKS ? DL_MAX_BLOCKS : nb);
}

extern "C" int64_t lidog_dice_ws(int32_t C) { return (int64_t)DL_MAX_BLOCKS * 4 * C; }

// 7 classes in every configuration of the reference (configs/*: out_channels 7); the other counts (binary heads, 16-class
// nuScenes, 19 / 20-class SemanticKITTI label sets) cost one instantiation each
#define DICE_DISPATCH(CALL)                                                     \
    switch (C) {                                                                \
        case 2: CALL(2); break;                                                 \
        case 3: CALL(3); break;                                                 \
        case 4: CALL(4); break;                                                 \
        case 5: CALL(5); break;                                                 \
        case 6: CALL(6); break;                                                 \
        case 7: CALL(7); break;                                                 \
        case 8: CALL(8); break;                                                 \
        case 9: CALL(9); break;                                                 \
        case 10: CALL(10); break;                                               \
        case 11: CALL(11); break;                                               \
        case 12: CALL(12); break;                                               \
        case 13: CALL(13); break;                                               \
        case 14: CALL(14); break;                                               \
        case 15: CALL(15); break;                                               \
        case 16: CALL(16); break;                                               \
        case 17: CALL(17); break;                                               \
        case 18: CALL(18); break;                                               \
        case 19: CALL(19); break;                                               \
        case 20: CALL(20); break;                                               \
        default: LIDOG_REQUIRE(false, "dice: C must be in [2, 20]");            \
    }

extern "C" int lidog_dice_fwd(const float *logits, const int64_t *target, int64_t n, int32_t C, int64_t ignore_label,
                              int32_t has_ignore, float eps, int32_t soft, int32_t powerize, int32_t use_tmask,
                              float offset, double *ws, float *loss, float *coef, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    LIDOG_REQUIRE(n >= 0 && C >= 2 && C <= DL_MAXC, "dice: bad shape");
    const float t_on = soft ? 1.f - eps : 1.f, t_off = soft ? eps / (float)(C - 1) : 0.f;
    const int nb = dice_blocks(n);
#define CALL(C_) \
    k_dice_sums<C_><<<nb, 256, 0, st>>>(logits, target, n, ignore_label, has_ignore, t_on, t_off, powerize, ws)
    DICE_DISPATCH(CALL)
#undef CALL
    k_dice_finish<<<1, 1024, 0, st>>>(ws, nb, C, use_tmask, offset, loss, coef);   // 8 lanes x 4C <= 80 columns
    LIDOG_LAUNCH_CHECK();
    return 0;
}

extern "C" int lidog_dice_bwd(const float *logits, const int64_t *target, int64_t n, int32_t C, int64_t ignore_label,
                              int32_t has_ignore, float eps, int32_t soft, int32_t powerize, const float *coef,
                              const float *gout, float *glogits, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    const float t_on = soft ? 1.f - eps : 1.f, t_off = soft ? eps / (float)(C - 1) : 0.f;
    int64_t nb = cdiv64(n, 256);
    if (nb > 4096) nb = 4096;
#define CALL(C_)                                                                                                \
    k_dice_bwd<C_><<<(unsigned)nb, 256, 0, st>>>(logits, target, n, ignore_label, has_ignore, t_on, t_off, powerize, \
                                                 coef, gout, glogits)
    DICE_DISPATCH(CALL)
#undef CALL
    LIDOG_LAUNCH_CHECK();
    return 0;
}
