// Measured ceiling of the exact-f32 matrix-core instructions on this GPU: a register-only loop of independent
// MFMAs, 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float *out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float *out, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 4; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 4; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, sizeof(float) * 256 * 4096);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024, 2048}) {
        float ms = time_ms([&] { k_mfma32<4><<<blocks, 256>>>(out, iters); });
        double fl = 2.0 * 32 * 32 * 2 * 4 * (double)iters * blocks * 4;
        printf("mfma_f32_32x32x2f32  %4d WGs x4 waves, 4 acc: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
    }
    for (int blocks : {256, 1024}) {
        float ms = time_ms([&] { k_mfma16<8><<<blocks, 256>>>(out, iters); });
        double fl = 2.0 * 16 * 16 * 4 * 8 * (double)iters * blocks * 4;
        printf("mfma_f32_16x16x4f32  %4d WGs x4 waves, 8 acc: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
    }
    return 0;
}
