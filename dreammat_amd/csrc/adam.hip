// Fused Adam step over one flat fp32 parameter buffer (hash grid + MLP) for gfx950.
// Semantics = torch.optim.Adam(lr, betas, eps) without weight decay / amsgrad, as configured by
// threestudio_dreammat/configs/dreammat.yaml:110-115 through threestudio/systems/utils.py:34-53.
// One pass: reads p,g,m,v, writes p,m,v and (optionally) zeroes g for the next step, float4-wide.
#include "dm_common.h"

__global__ __launch_bounds__(256) void k_adam(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                              float4* __restrict__ v, long long n4, float lr, float b1, float b2,
                                              float eps, float bc1, float bc2_sqrt, float gscale, int zero_grad) {
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        float* pp = (float*)&P; float* gg = (float*)&G; float* mm = (float*)&M; float* vv = (float*)&V;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = gg[k] * gscale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gr;
            vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
            float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] = pp[k] - (lr / bc1) * (mm[k] / denom);
        }
        p[i] = P; m[i] = M; v[i] = V;
        if (zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

extern "C" int dm_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step,
                            float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                            hipStream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return DM_ERR_ARG;
    if (n % 4 != 0) return DM_ERR_ARG;  // caller pads the flat buffer to a multiple of 4 floats (16 B)
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return DM_ERR_ARG;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long n4 = n / 4;
    int blocks = (int)std::min<long long>((n4 + 255) / 256, 256 * 8);
    DM_ENTER();
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, stream, (float4*)param, (float4*)grad, (float4*)exp_avg,
                       (float4*)exp_avg_sq, n4, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale,
                       zero_grad);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
