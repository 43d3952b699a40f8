// Diagnostic build only (scripts/clock_stamps.sh): the shader clock a kernel really runs at, per workgroup, as
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The stamps go to a
// buffer no kernel reads; no output value depends on them.  Without -DLIDOG_CLOCK_STAMP every macro is empty and the
// shipped library contains none of this.
#pragma once
#ifdef LIDOG_CLOCK_STAMP
#include <hip/hip_runtime.h>
#define LIDOG_STAMP_SLOTS 4096
static __device__ unsigned long long g_lidog_stamps[4 * LIDOG_STAMP_SLOTS];
#define LIDOG_STAMP_BEGIN()                                                  \
    unsigned long long stamp_c0_ = 0, stamp_r0_ = 0;                         \
    if (threadIdx.x == 0) {                                                  \
        stamp_c0_ = __builtin_amdgcn_s_memtime();                            \
        stamp_r0_ = __builtin_amdgcn_s_memrealtime();                        \
    }
#define LIDOG_STAMP_END()                                                                         \
    if (threadIdx.x == 0) {                                                                       \
        const unsigned long long c1_ = __builtin_amdgcn_s_memtime();                              \
        const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();                          \
        const unsigned s_ = (blockIdx.x + gridDim.x * blockIdx.y) & (LIDOG_STAMP_SLOTS - 1);      \
        g_lidog_stamps[4 * s_ + 0] = c1_ - stamp_c0_;                                             \
        g_lidog_stamps[4 * s_ + 1] = r1_ - stamp_r0_;                                             \
        g_lidog_stamps[4 * s_ + 2] = r1_;                                                         \
        g_lidog_stamps[4 * s_ + 3] = 1;                                                           \
    }
// host: copies the 4 x LIDOG_STAMP_SLOTS words (cycles, 100 MHz ticks, end tick, valid) and clears them
extern "C" int lidog_debug_clock_stamps(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lidog_stamps), sizeof(unsigned long long) * 4 * LIDOG_STAMP_SLOTS) != hipSuccess)
        return 1;
    static unsigned long long zeros[4 * LIDOG_STAMP_SLOTS];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lidog_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
#else
#define LIDOG_STAMP_BEGIN()
#define LIDOG_STAMP_END()
#endif
