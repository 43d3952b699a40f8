// Shared helpers for liblidog_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lidog_amd.h"

void lidog_set_error(const char *fmt, ...);

#define LIDOG_CHECK_HIP(expr)                                                                   \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            lidog_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

#define LIDOG_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            lidog_set_error(__VA_ARGS__); \
            return 2;                     \
        }                                 \
    } while (0)

#define LIDOG_LAUNCH_CHECK() LIDOG_CHECK_HIP(hipGetLastError())

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// What the last kernel of a per-channel (sum, sum) reduction does besides writing the sums (bn.hip:k_sums_finish):
// BatchNorm statistics finalisation (mean != NULL) and/or float copies of the two sums (dw / db != NULL).
struct BnFinish {
    float eps, momentum;
    float *mean, *invstd, *running_mean, *running_var;
    float *dw, *db;
};
int lidog_launch_sums_finish(const double *partial, int nb, int C, double *sums, double count, BnFinish fin,
                             hipStream_t st);

// 63-bit packed coordinate key: batch 12 bits, x/y/z 17 bits each (biased by 65536).
#define LIDOG_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t lidog_pack(int b, int x, int y, int z, int *bad) {
    unsigned ux = (unsigned)(x + 65536), uy = (unsigned)(y + 65536), uz = (unsigned)(z + 65536);
    if (((unsigned)b > 4095u) | (ux > 131071u) | (uy > 131071u) | (uz > 131071u)) *bad = 1;
    return ((uint64_t)(unsigned)b << 51) | ((uint64_t)ux << 34) | ((uint64_t)uy << 17) | (uint64_t)uz;
}

__device__ __forceinline__ uint64_t lidog_mix(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// row of `key` in an open-addressing table, or -1
__device__ __forceinline__ int lidog_find(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                          uint64_t mask, uint64_t key) {
    uint64_t slot = lidog_mix(key) & mask;
    for (;;) {
        uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == LIDOG_EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
}

// ReLU bit mask of a [rows, C] matrix (C % 4 == 0) written by lidog_bn_apply_bits: bit 4 * (q % 8) + j of word q / 8 is
// (element j of float4 number q) > 0.  Read back as a float4 of 1 / 0 so that the consumers keep their `y > 0` tests.
__device__ __forceinline__ float4 lidog_relu_bits_as_float4(const uint32_t *__restrict__ bits, int64_t q) {
    const uint32_t nib = bits[q >> 3] >> (4 * (int)(q & 7));
    return make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u));
}
