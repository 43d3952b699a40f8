// Which exact-f32 MFMA shape sustains more FLOP/s on RANDOM operands (the chip lowers its clock under matrix load, and
// MI355X_MICROARCH.md, DVFS give-back item 7, reports a 1.15 x gap between the two bf16 shapes at equal cycles)?
// Register-only loops and loops that re-read every operand from LDS, 1 wave per SIMD and 4 waves per SIMD, ~0.3 s each.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256) void k32(const float *__restrict__ rnd, float *out, int iters) {
    __shared__ float sm[8 * 256 * 2];
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x[8], y[8];
    for (int j = 0; j < 8; ++j) {
        x[j] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + j];
        y[j] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + j];
        sm[j * 256 + threadIdx.x] = x[j];
        sm[(8 + j) * 256 + threadIdx.x] = y[j];
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = x[j], b = y[j];
            if (LDS) {
                a = sm[j * 256 + threadIdx.x];
                b = sm[(8 + j) * 256 + ((threadIdx.x + it) & 255)];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void k16(const float *__restrict__ rnd, float *out, int iters) {
    __shared__ float sm[8 * 256 * 2];
    f32x4 acc[16];   // same 64 accumulator registers as 4 x 32x32
    for (int a = 0; a < 16; ++a)
        for (int e = 0; e < 4; ++e) acc[a][e] = 0.f;
    float x[8], y[8];
    for (int j = 0; j < 8; ++j) {
        x[j] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + j];
        y[j] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + j];
        sm[j * 256 + threadIdx.x] = x[j];
        sm[(8 + j) * 256 + threadIdx.x] = y[j];
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = x[j], b = y[j];
            if (LDS) {
                a = sm[j * 256 + threadIdx.x];
                b = sm[(8 + j) * 256 + ((threadIdx.x + it) & 255)];
            }
            // the same FLOPs as 4 x 32x32x2 (16384): 8 x 16x16x4 (2048 each)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[(2 * j + q) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(2 * j + q) & 15], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < 16; ++a)
        for (int e = 0; e < 4; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int maxb = 1024;
    std::vector<float> h((size_t)maxb * 256 * 16);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *rnd, *out;
    hipMalloc(&rnd, h.size() * sizeof(float)); hipMalloc(&out, sizeof(float) * 256 * maxb);
    hipMemcpy(rnd, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int blocks : {256, 1024}) {
        const int iters = blocks == 256 ? 60000 : 15000;   // ~0.2-0.3 s per launch
        const double fl = 2.0 * 32 * 32 * 2 * 4 * 8 * (double)iters * blocks * 4;
        float a = time_ms([&] { k32<false><<<blocks, 256>>>(rnd, out, iters); });
        float b = time_ms([&] { k16<false><<<blocks, 256>>>(rnd, out, iters); });
        float c = time_ms([&] { k32<true><<<blocks, 256>>>(rnd, out, iters); });
        float d = time_ms([&] { k16<true><<<blocks, 256>>>(rnd, out, iters); });
        printf("%4d WGs (%d wave/SIMD)  regs: 32x32x2 %.1f TF/s (%.0f ms)  16x16x4 %.1f TF/s | LDS operands: 32x32x2 %.1f TF/s  16x16x4 %.1f TF/s\n",
               blocks, blocks / 256, fl / a / 1e9, a, fl / b / 1e9, fl / c / 1e9, fl / d / 1e9);
    }
    return 0;
}
