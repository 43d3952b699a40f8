// Fused optimiser steps on slices of the flat parameter / gradient buffers (lidog_amd/optim.py).
// SGD: torch.optim.SGD(lr, momentum, weight_decay, nesterov=True, dampening=0) as configured at
// utils/pipelines/trainer_lighting_2d.py:351-355 (momentum 0.98, :26).  Adam lives in conv2d.hip (k_adam).
#include "common.h"

// g' = g * grad_scale + wd * p;  buf = mu * buf + g'  (buf starts at 0: the first step gives buf = g' like
// torch's clone);  p -= lr * (g' + mu * buf)  [nesterov]  or  p -= lr * buf
__global__ __launch_bounds__(256) void k_sgd(float *__restrict__ p, const float *__restrict__ g,
                                             float *__restrict__ buf, int64_t n, float lr, float mu, float wd,
                                             int nesterov, float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * grad_scale + wd * pi;
        const float bi = buf[i] * mu + gi;
        const float step = nesterov ? gi + mu * bi : bi;
        p[i] = pi - lr * step;
        buf[i] = bi;
    }
}

extern "C" int lidog_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t n, float lr,
                              float momentum, float weight_decay, int32_t nesterov, float grad_scale,
                              void *stream) {
    if (n == 0) return 0;
    LIDOG_REQUIRE(momentum >= 0.f && lr >= 0.f, "sgd_step: lr and momentum must be non-negative");
    int64_t g = cdiv64(n, 256);
    if (g > 8192) g = 8192;
    k_sgd<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(param, grad, momentum_buf, n, lr, momentum, weight_decay,
                                                        nesterov, grad_scale);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
