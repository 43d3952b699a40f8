// launchers of csrc/sconv_mfma.hip (internal to the library)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// BatchNorm (+ ReLU) of the layer that produced the gathered rows, applied while they are staged: the consumer reads the
// producer's raw convolution output and never needs the normalised copy (bn.hip:k_bn_apply4's expression and operation
// order, so the staged values are that kernel's bits).  mean == NULL: rows taken as they are.
struct InBn {
    const float *mean, *invstd, *w, *b;
    int relu;
};

__device__ __forceinline__ float4 in_bn_apply(float4 v, const float4 &m, const float4 &s, const float4 &w, const float4 &b,
                                              int relu) {
    v.x = (v.x - m.x) * s.x * w.x + b.x;
    v.y = (v.y - m.y) * s.y * w.y + b.y;
    v.z = (v.z - m.z) * s.z * w.z + b.z;
    v.w = (v.w - m.w) * s.w * w.w + b.w;
    const float lo = relu ? 0.f : -__builtin_inff();   // one max either way: no branch in the staging loop
    v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
    return v;
}

int lidog_launch_gemm_mfma(const float *A, const int32_t *gather, const float *B, const float *bias,
                           const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows, int n_tiles,
                           int Cin, int Cout, float *T, const int32_t *scatter, InBn in_bn, int64_t a_rows, hipStream_t st);
void lidog_gemm_multi_suspend(int delta);
int lidog_launch_wgrad_mfma(const float *A, const int32_t *pa, const float *G, const int32_t *pg,
                            const int32_t *items, int n_items, int Cin, int Cout, float *partial, InBn in_bn,
                            hipStream_t st);
int lidog_wgrad_mfma_slabs(int Cin, int Cout, int n_items);
int lidog_wgrad_mfma_wg_per_cu(int Cin, int Cout, int fold);
