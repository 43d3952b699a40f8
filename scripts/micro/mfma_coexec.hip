// What does work issued by OTHER waves of the same SIMD cost the f32 matrix pipe?  512-thread workgroups:
// waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (one per SIMD) a loop of one instruction class.
// Reported: time of the launch and the MFMA-side rate; "per op" = extra cycles per competing wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k_mix(float *out, const float *in, int iters, int per_iter, int mode,
                                             unsigned long long *ticks) {
    __shared__ float lds[4096];
    const bool mfma_role = threadIdx.x < 256;
    if (mode == 0 && !mfma_role) return;
    if (mfma_role) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a)
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
        const unsigned long long t0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        }
        float s = 0;
        for (int a = 0; a < 4; ++a)
            for (int e = 0; e < 16; ++e) s += acc[a][e];
        const unsigned long long t1 = wall_clock64();
        if ((threadIdx.x & 63) == 0) atomicMax(ticks, t1 - t0);
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const int l = threadIdx.x & 255;
    lds[l] = v[0];
    const float *gp = in + (threadIdx.x & 255);
    int sacc = blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < per_iter; j += 8) {
            if (mode == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 1e-4f);
            } else if (mode == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += ((volatile float *)lds)[l + 256 * i];
            } else if (mode == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ((volatile float *)lds)[l + 256 * i] = v[i];
            } else if (mode == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += __builtin_nontemporal_load(gp + 256 * i + ((it & 3) << 11));
            } else if (mode == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
            } else if (mode == 6) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(v[i]));
            } else if (mode == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int t = __float_as_int(v[i]);
                    asm volatile("v_add_u32 %0, %0, 3" : "+v"(t));
                    v[i] = __int_as_float(t);
                }
            }
        }
    }
    float s = (float)sacc;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, sizeof(float) * 512 * 4096);
    (void)hipMalloc(&in, sizeof(float) * 65536);
    (void)hipMemset(in, 0, sizeof(float) * 65536);
    unsigned long long *ticks;
    (void)hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;
    float base_ms = 0;
    auto run = [&](int per, int mode, const char *what) {
        k_mix<<<blocks, 512>>>(out, in, iters, per, mode, ticks);
        (void)hipDeviceSynchronize();
        (void)hipMemset(ticks, 0, 8);
        k_mix<<<blocks, 512>>>(out, in, iters, per, mode, ticks);
        (void)hipDeviceSynchronize();
        unsigned long long tk = 0;
        (void)hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
        float ms = (float)(tk / 100e6 * 1e3);   // wall_clock64 ticks at 100 MHz: slowest MFMA wave
        if (mode == 0) base_ms = ms;
        double fl = 2.0 * 32 * 32 * 2 * 4 * (double)iters * blocks * 4;
        double per_op = per ? (ms - base_ms) * 1e-3 * 2.4e9 / ((double)iters * per) : 0;
        printf("%-44s %2d per 4 MFMA: %7.3f ms  %6.1f TFLOP/s  %5.1f cycles per op\n", what, per, ms, fl / ms / 1e9, per_op);
    };
    run(0, 0, "MFMA waves alone (2 WGs/CU, 2 per SIMD)");
    const char *names[] = {"", "v_fma_f32", "ds_read_b32", "ds_write_b32", "global_load_dword (L2 hit)", "s_add_u32", "v_mov_b32",
                           "v_add_u32"};
    for (int mode = 1; mode <= 7; ++mode)
        for (int per : {8, 16, 32}) run(per, mode, names[mode]);
    return 0;
}
