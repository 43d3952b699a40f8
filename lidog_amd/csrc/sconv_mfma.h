// launchers of csrc/sconv_mfma.hip (internal to the library)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int lidog_launch_gemm_mfma(const float *A, const int32_t *gather, const float *B, const float *bias,
                           const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows, int n_tiles,
                           int Cin, int Cout, float *T, const int32_t *scatter, hipStream_t st);
int lidog_launch_wgrad_mfma(const float *A, const int32_t *pa, const float *G, const int32_t *pg,
                            const int32_t *items, int n_items, int Cin, int Cout, float *partial, hipStream_t st);
int lidog_wgrad_mfma_slabs(int Cin, int Cout, int n_items);
