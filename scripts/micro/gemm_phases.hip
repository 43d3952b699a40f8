// Where do the matrix-pipe cycles of the gathered GEMM go?  The loop of csrc/sconv_mfma.hip:k_sconv_gemm_mfma<4> (128 rows x
// 128 columns per workgroup, 32-channel chunks, 64 MFMAs per wave and chunk) with its parts switched off one by one:
//   mode 0  everything: gathered global loads (rows from a table in L2 / Infinity Cache), LDS staging, two barriers, MFMAs
//   mode 1  no global loads (the staging registers keep their values)
//   mode 2  no global loads, no LDS writes (barriers stay)
//   mode 3  no global loads, no LDS writes, no barriers: LDS operand reads + MFMAs only
//   mode 4  mode 0 with ONE barrier per chunk (LDS image double-buffered)
//   mode 5  LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write; two LDS images, one barrier per chunk; the A
//           image is lane-linear with the quad of a row XOR-swizzled on the SOURCE side, read back by ds_read_b128
//   mode 6  mode 0 on a 2 MB table (L2-resident rows): how much of mode 0 - mode 1 is load latency
// at 1, 2 and 3 workgroups per CU (unused dynamic LDS caps the residency).  Reported: TF/s of executed MFMA work and the
// share of the 157.3 TF/s peak.   Build: hipcc -O3 --offload-arch=gfx950 gemm_phases.hip -o gemm_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TM 128
#define BK 32
#define SA 33
#define NT 4
#define TN 128

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_gemm(const float *__restrict__ A, const int *__restrict__ gather,
                                                 const float *__restrict__ B, int Cin, float *__restrict__ T, int chunks) {
    constexpr int NBUF = (MODE == 4 || MODE == 5) ? 2 : 1;
    __shared__ float As[NBUF][TM * SA];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * TN];
    extern __shared__ float pad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const float *a_row[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = tid + 256 * j;
        a_row[j] = A + (size_t)gather[blockIdx.x * TM + (f >> 3)] * Cin + (f & 7) * 4;
    }
    float4 ra[4], rb[4];
    auto load = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4 *>(a_row[j] + kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            rb[j] = *reinterpret_cast<const float4 *>(B + (size_t)(kb + f / 32) * TN + (f % 32) * 4);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int f = tid + 256 * j;
            const int o = (f >> 3) * SA + (f & 7) * 4;
            As[buf][o] = ra[j].x; As[buf][o + 1] = ra[j].y; As[buf][o + 2] = ra[j].z; As[buf][o + 3] = ra[j].w;
            *reinterpret_cast<float4 *>(&Bs[buf][(tid + 256 * j) * 4]) = rb[j];
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    load(0);
    if (MODE >= 2) { store(0); __syncthreads(); }
    if (MODE == 4) { store(0); __syncthreads(); load(BK % Cin); }
    for (int c = 0; c < chunks; ++c) {
        const int kb = (c * BK) % Cin;
        int buf = 0;
        if (MODE <= 1) {
            __syncthreads();
            store(0);
            __syncthreads();
            if (MODE == 0) load((kb + BK) % Cin);
        } else if (MODE == 2) {
            __syncthreads();
            __syncthreads();
        } else if (MODE == 4) {
            // image c is in buffer c & 1 (written during chunk c - 1); write image c + 1 into the other one while this one is
            // read, one barrier per chunk
            buf = c & 1;
            store(buf ^ 1);
            load((kb + 2 * BK) % Cin);
        }
        const float *arow = &As[buf][(wave * 32 + li) * SA + kh];
        const float *bcol = &Bs[buf][kh * TN + li * NT];
        float4 bq0 = *reinterpret_cast<const float4 *>(bcol), bq1 = *reinterpret_cast<const float4 *>(bcol + 2 * TN), bn0, bn1;
        float a0 = arow[0], a1 = arow[2], an0, an1;
#pragma unroll
        for (int j = 0; j < BK / 4; ++j) {
            if (j + 1 < BK / 4) {
                bn0 = *reinterpret_cast<const float4 *>(bcol + (4 * j + 4) * TN);
                bn1 = *reinterpret_cast<const float4 *>(bcol + (4 * j + 6) * TN);
                an0 = arow[4 * j + 4];
                an1 = arow[4 * j + 6];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.w, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.w, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < BK / 4) { bq0 = bn0; bq1 = bn1; a0 = an0; a1 = an1; }
        }
        if (MODE == 4) __syncthreads();
    }
    // one store per lane keeps the accumulators alive
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    T[(size_t)blockIdx.x * 256 + tid] = MODE == 0 ? s : s + ra[0].x + rb[0].x;
}


__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const float *)p);
}
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(256, 1) void k_gemm_dma(const float *__restrict__ A, const int *__restrict__ gather,
                                                     const float *__restrict__ B, int Cin, float *__restrict__ T, int chunks) {
    __shared__ __attribute__((aligned(1024))) float As[2][TM * BK];
    __shared__ __attribute__((aligned(1024))) float Bs[2][BK * TN];
    extern __shared__ float pad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    // DMA instruction jj of this wave fills A rows 8 (4 wave + jj) .. + 7: lane -> row rr = lane >> 3, slot p = lane & 7 holds
    // quad p ^ ((row >> 1) & 7) of the row
    const float *a_src[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int r = 8 * (4 * wave + jj) + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        a_src[jj] = A + (size_t)gather[blockIdx.x * TM + r] * Cin + q * 4;
    }
    auto dma = [&](int kb, int buf) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * wave + jj;
            // inline asm: hipcc would put `s_waitcnt vmcnt(0)` between a glds it knows about and the next LDS read (it cannot
            // tell the two images apart); the waits are placed by hand below (cdna_hip_programming.md section 5)
            glds16(a_src[jj] + kb, lds_addr(&As[buf][j * 256]));
            glds16(B + (size_t)kb * TN + j * 256 + lane * 4, lds_addr(&Bs[buf][j * 256]));
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int i = wave * 32 + li;
    const int sw = (i >> 1) & 7;
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        dma(((c + 1) * BK) % Cin, buf ^ 1);
        const float4 *aq = reinterpret_cast<const float4 *>(&As[buf][i * BK]);
        const float *bcol = &Bs[buf][kh * TN + li * NT];
        float4 av = aq[0 ^ sw], avn;
        float4 bq0 = *reinterpret_cast<const float4 *>(bcol), bq1 = *reinterpret_cast<const float4 *>(bcol + 2 * TN), bn0, bn1;
#pragma unroll
        for (int j = 0; j < BK / 4; ++j) {
            if (j + 1 < BK / 4) {
                bn0 = *reinterpret_cast<const float4 *>(bcol + (4 * j + 4) * TN);
                bn1 = *reinterpret_cast<const float4 *>(bcol + (4 * j + 6) * TN);
                avn = aq[(j + 1) ^ sw];
            }
            const float a0 = kh ? av.y : av.x, a1 = kh ? av.w : av.z;
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq0.w, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq1.w, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < BK / 4) { bq0 = bn0; bq1 = bn1; av = avn; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA into the other image has landed ...
        __syncthreads();                                       // ... and so has everybody's; everybody is done reading this one
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    T[(size_t)blockIdx.x * 256 + tid] = s;
}

// the same product with plain loads (reference for mode 5's addressing): T2[wg][tid] as mode 5 writes it
#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

template <int MODE>
static void run(const float *A, const int *g, const float *B, int Cin, float *T, int chunks, int wgs, int per_cu) {
    void (*kern)(const float *, const int *, const float *, int, float *, int) =
        MODE == 5 ? k_gemm_dma : k_gemm<(MODE == 5 || MODE == 6) ? 0 : MODE>;
    hipFuncAttributes attr;
    CHECK(hipFuncGetAttributes(&attr, (const void *)kern));
    const size_t stat = attr.sharedSizeBytes, cu = 160 * 1024;
    // dynamic LDS so that per_cu workgroups fit in a CU's 160 KiB and per_cu + 1 do not
    size_t dyn = 0;
    if ((per_cu + 1) * stat <= cu) dyn = cu / (per_cu + 1) - stat + 1024;
    if (per_cu * (stat + dyn) > cu) { printf("mode %d: %d workgroups of %zu B do not fit\n", MODE, per_cu, stat); return; }
    if (stat + dyn > 64 * 1024)
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) kern<<<wgs, 256, dyn>>>(A, g, B, Cin, T, chunks);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) kern<<<wgs, 256, dyn>>>(A, g, B, Cin, T, chunks);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = (double)wgs * chunks * 4 * 64 * 2.0 * 32 * 32 * 2;
    const hipError_t err = hipGetLastError();
    printf("mode %d  %d WG/CU  %6d workgroups x %3d chunks: %8.3f ms  %6.1f TF/s  %.3f of 157.3%s\n", MODE, per_cu, wgs, chunks, ms,
           fl / ms / 1e9, fl / ms / 1e9 / 157.3, err == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
    const int rows = 41416, Cin = 256;        // the stride-8 map of the bench batch: 42 MB table
    std::vector<float> hA((size_t)rows * Cin), hB((size_t)Cin * TN);
    for (auto &v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hB) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    const int max_wgs = 768 * 8;
    std::vector<int> hg((size_t)max_wgs * TM);
    for (auto &v : hg) v = rand() % rows;
    std::vector<int> hg2(hg.size());
    for (auto &v : hg2) v = rand() % 2048;
    int *g2;
    CHECK(hipMalloc(&g2, hg2.size() * 4));
    CHECK(hipMemcpy(g2, hg2.data(), hg2.size() * 4, hipMemcpyHostToDevice));
    float *A, *B, *T; int *g;
    CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&B, hB.size() * 4)); CHECK(hipMalloc(&T, (size_t)max_wgs * 256 * 4));
    CHECK(hipMalloc(&g, hg.size() * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        // ONE round of resident workgroups, long loops: no tail, no relaunch -- the loop itself
        const int wgs = 256 * per_cu, chunks = 192;
        run<0>(A, g, B, Cin, T, chunks, wgs, per_cu);
        run<1>(A, g, B, Cin, T, chunks, wgs, per_cu);
        run<2>(A, g, B, Cin, T, chunks, wgs, per_cu);
        run<3>(A, g, B, Cin, T, chunks, wgs, per_cu);
        if (per_cu <= 2) run<4>(A, g, B, Cin, T, chunks, wgs, per_cu);
        if (per_cu <= 2) run<5>(A, g, B, Cin, T, chunks, wgs, per_cu);
        run<6>(A, g2, B, Cin, T, chunks, wgs, per_cu);
    }
    {   // mode 5 computes what mode 0 computes (same chunk sequence, same fmaf chains): compare the per-thread checksums
        std::vector<float> t0((size_t)768 * 256), t5(t0.size());
        k_gemm<0><<<768, 256>>>(A, g, B, Cin, T, 8);
        CHECK(hipMemcpy(t0.data(), T, t0.size() * 4, hipMemcpyDeviceToHost));
        k_gemm_dma<<<768, 256>>>(A, g, B, Cin, T, 8);
        CHECK(hipMemcpy(t5.data(), T, t5.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < t0.size(); ++i) bad += (t0[i] != t5[i]);
        printf("mode 5 against mode 0, 768 workgroups x 8 chunks: %zu of %zu per-thread sums differ\n", bad, t0.size());
    }
    // the real launch shape: 8 chunks per workgroup, 7 rounds of 768
    run<0>(A, g, B, Cin, T, 8, 768 * 7, 3);
    run<1>(A, g, B, Cin, T, 8, 768 * 7, 3);
    run<3>(A, g, B, Cin, T, 8, 768 * 7, 3);
    run<5>(A, g, B, Cin, T, 8, 768 * 7, 2);
    run<6>(A, g2, B, Cin, T, 8, 768 * 7, 3);
    return 0;
}
