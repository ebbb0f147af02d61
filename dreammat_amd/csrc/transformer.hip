// Row-wise pieces of the UNet / ControlNet transformer blocks on bf16 token matrices for gfx950:
// LayerNorm and the GEGLU gate.  diffusers runs them as separate ATen kernels inside BasicTransformerBlock
// (reached from threestudio/models/guidance/dreammat_guidance.py:205-292); both are pure bandwidth
// (LayerNorm: 1 read + 1 write, GEGLU: 2 reads + 1 write of [tokens, C] bf16), so each is one pass with 16 B
// accesses.  Forward only: the diffusion nets run without autograd in score distillation (the SDS gradient is
// injected at the latents, dreammat_guidance.py:385-397).
#include <algorithm>

#include "dm_common.h"
#include "dm_elem.h"

namespace {


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row, NCH 16-byte chunks per lane held in registers between the two passes (mean, then the
// centred second moment: the same two-pass arithmetic as ATen's fp32-accumulating kernel).
template <int NCH>
__global__ __launch_bounds__(256) void k_layernorm(const elem_t* __restrict__ x, const elem_t* __restrict__ gamma,
                                                   const elem_t* __restrict__ beta, elem_t* __restrict__ y,
                                                   long long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    const int chunks = C / 8;
    float g[NCH][8], b[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int ch = lane + 64 * c;
        bool ok = ch < chunks;
        elem8 gv = *reinterpret_cast<const elem8*>(gamma + (ok ? ch : 0) * 8);
        elem8 bv = *reinterpret_cast<const elem8*>(beta + (ok ? ch : 0) * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { g[c][k] = (float)gv[k]; b[c][k] = (float)bv[k]; }
    }
    const float inv_c = 1.f / (float)C;
    for (long long r = wave0; r < rows; r += nwaves) {
        const elem_t* xr = x + r * C;
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int ch = lane + 64 * c;
            bool ok = ch < chunks;
            elem8 xv = *reinterpret_cast<const elem8*>(xr + (ok ? ch : 0) * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[c][k] = ok ? (float)xv[k] : 0.f; s += v[c][k]; }
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bool ok = lane + 64 * c < chunks;
#pragma unroll
            for (int k = 0; k < 8; ++k) { float d = v[c][k] - mean; q += ok ? d * d : 0.f; }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
        elem_t* yr = y + r * C;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int ch = lane + 64 * c;
            if (ch < chunks) {
                elem8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (elem_t)((v[c][k] - mean) * rstd * g[c][k] + b[c][k]);
                *reinterpret_cast<elem8*>(yr + ch * 8) = o;
            }
        }
    }
}

// y[r, c] = h[r, c] * gelu(h[r, inner + c])  (exact erf GELU).  The gate is rounded to bf16 before the product,
// which is what the two-kernel ATen sequence F.gelu(gate) -> mul produces.
__global__ __launch_bounds__(256) void k_geglu(const elem_t* __restrict__ h, elem_t* __restrict__ y, long long rows,
                                               int inner) {
    const int cpr = inner / 8;
    const long long total = rows * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i / cpr;
        int c = (int)(i - r * cpr);
        const elem_t* hr = h + r * 2LL * inner;
        elem8 xv = *reinterpret_cast<const elem8*>(hr + c * 8);
        elem8 gv = *reinterpret_cast<const elem8*>(hr + inner + c * 8);
        elem8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float gf = (float)gv[k];
            float ge = (float)(elem_t)(0.5f * gf * (1.f + erff(gf * 0.70710678118654752440f)));
            o[k] = (elem_t)((float)xv[k] * ge);
        }
        *reinterpret_cast<elem8*>(y + r * inner + c * 8) = o;
    }
}


// Row softmax of a score matrix that exists in memory (the VAE encoder's mid-block attention: one head of width 512 over
// 4096 tokens, differentiated -- dreammat_guidance.py:284-292 -> diffusers AttnProcessor: baddbmm, softmax, bmm).  The ATen
// sequence around the two matrix products was: scale (1 pass), bf16 -> fp32 copy, softmax, fp32 -> bf16 copy forward and
// softmax backward, two copies and a scale backward, all over a [B, 4096, 4096] tensor: 1.5 ms of the step.  Here: one pass
// each way, 16-byte accesses, fp32 arithmetic on bf16 storage.  One workgroup per row, NCH chunks of 8 columns per thread.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int NCH, bool BWD>
__global__ __launch_bounds__(256) void k_softmax_rows(const elem_t* __restrict__ a, const elem_t* __restrict__ b, elem_t* __restrict__ y,
                                                      long long rows, int cols, float scale) {
    // forward:  a = scores s,         y = softmax(scale * s)
    // backward: a = probabilities p,  b = dL/dp,  y = dL/ds = scale * p * (dp - sum_j p_j dp_j)
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunks = cols / 8;
    constexpr float kLog2e = 1.4426950408889634f;
    int it = 0;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x, ++it) {
        const elem_t* ar = a + r * cols;
        float v[NCH][8], w[NCH][8];
        float m = -INFINITY, acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = tid + 256 * c;
            const bool ok = ch < chunks;
            const elem8 av = *reinterpret_cast<const elem8*>(ar + (ok ? ch : 0) * 8);
            if (BWD) {
                const elem8 bv = *reinterpret_cast<const elem8*>(b + r * cols + (ok ? ch : 0) * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[c][k] = ok ? (float)av[k] : 0.f;
                    w[c][k] = (float)bv[k];
                    acc += v[c][k] * w[c][k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[c][k] = ok ? (float)av[k] * (scale * kLog2e) : -INFINITY;
                    m = fmaxf(m, v[c][k]);
                }
            }
        }
        float* rd = red[it & 1];              // (two slots: a fast wave may start the next row before a slow one has read)
        if (!BWD) {
            m = wave_max(m);
            if (lane == 0) rd[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(rd[0], rd[1]), fmaxf(rd[2], rd[3]));
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[c][k] = exp2f(v[c][k] - m); acc += v[c][k]; }
        }
        acc = wave_sum(acc);
        if (lane == 0) rd[wave] = acc;
        __syncthreads();
        const float tot = (rd[0] + rd[1]) + (rd[2] + rd[3]);
        const float inv = BWD ? 0.f : 1.f / tot;
        elem_t* yr = y + r * cols;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = tid + 256 * c;
            if (ch < chunks) {
                elem8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (elem_t)(BWD ? scale * v[c][k] * (w[c][k] - tot) : v[c][k] * inv);
                *reinterpret_cast<elem8*>(yr + ch * 8) = o;
            }
        }
    }
}

// 16-bit matrix transpose, batched: src [batch, R, C] -> dst [batch, C, R] (R % 64 == C % 64 == 0).  64 x 64 tiles through LDS,
// 16-byte loads and stores on both sides; the operands of the VAE mid-block attention's GEMM form that must be contracted along
// their rows (V^T for P V, K^T / Q^T / dO^T and the transposed probabilities / score gradients of its backward).
__global__ __launch_bounds__(256) void k_transpose16(const elem_t* __restrict__ src, elem_t* __restrict__ dst, int R, int C) {
    __shared__ unsigned short t[64][66];                       // (pitch 33 words: a column walk touches every bank once)
    const int tid = threadIdx.x;
    const long long base = (long long)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int rr = tid >> 3, ch = tid & 7;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = rr + 32 * k;
        const uint4 v = *reinterpret_cast<const uint4*>(src + base + (long long)(r0 + r) * C + c0 + ch * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[r][ch * 8 + 2 * e] = (unsigned short)(w[e] & 0xffffu);
            t[r][ch * 8 + 2 * e + 1] = (unsigned short)(w[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int oc = rr + 32 * k;                            // output row = source column
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (unsigned)t[ch * 8 + 2 * e][oc] | ((unsigned)t[ch * 8 + 2 * e + 1][oc] << 16);
        *reinterpret_cast<uint4*>(dst + base + (long long)(c0 + oc) * R + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Skip connection of a UNet up block with the ControlNet residual folded in (diffusers UNet2DConditionModel.forward:
// `down_block_res_samples = [s + r ...]` then `torch.cat([hidden, res_sample], dim=1)` in every up-block resnet):
// y[row, 0:Cx] = x[row], y[row, Cx:Cx+Cs] = s[row] (+ r[row] * r_scale) -- one pass instead of an add pass and a cat pass.
__global__ __launch_bounds__(256) void k_cat_add(const elem_t* __restrict__ x, const elem_t* __restrict__ s, const elem_t* __restrict__ r,
                                                 elem_t* __restrict__ y, long long rows, int Cx, int Cs, float r_scale) {
    const int cx8 = Cx / 8, c8 = (Cx + Cs) / 8;
    const long long total = rows * c8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c8;
        const int c = (int)(i - row * c8);
        elem8 v;
        if (c < cx8) {
            v = *reinterpret_cast<const elem8*>(x + row * Cx + c * 8);
        } else {
            const long long o = row * Cs + (long long)(c - cx8) * 8;
            v = *reinterpret_cast<const elem8*>(s + o);
            if (r) {
                const elem8 w = *reinterpret_cast<const elem8*>(r + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (elem_t)((float)v[k] + (float)w[k] * r_scale);
            }
        }
        *reinterpret_cast<elem8*>(y + row * (long long)(Cx + Cs) + c * 8) = v;
    }
}

}  // namespace

// y[m, n] = sum_k x[m, k] w[n, k] + bias[n] for a FEW channels (K = 8 per lane-row, N <= 16): AutoencoderKL's quant_conv (8 -> 8, 1 x 1)
// behind dreammat_guidance.py:284-292 -- the last Linear of the differentiated VAE encoder that ran on ATen; its data gradient is the
// same kernel on w^T.  One thread per row: a 16-byte load, N dot products of 8, one or two 16-byte stores.  fp32 accumulate, one rounding.
template <int N8>
__global__ __launch_bounds__(256) void k_linear_small(const elem_t* __restrict__ x, const elem_t* __restrict__ w, const elem_t* __restrict__ bias,
                                                      elem_t* __restrict__ y, long long M, int N) {
    __shared__ float sw[16 * 8 + 16];
    for (int i = threadIdx.x; i < 16 * 8; i += 256) sw[i] = i < N * 8 ? (float)w[i] : 0.f;
    for (int i = threadIdx.x; i < 16; i += 256) sw[128 + i] = (bias && i < N) ? (float)bias[i] : 0.f;
    __syncthreads();
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        const elem8 v = *reinterpret_cast<const elem8*>(x + m * 8);
        float xf[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xf[k] = (float)v[k];
#pragma unroll
        for (int c = 0; c < N8; ++c) {
            elem8 o;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                float a = sw[128 + 8 * c + n];
#pragma unroll
                for (int k = 0; k < 8; ++k) a = __builtin_fmaf(xf[k], sw[(8 * c + n) * 8 + k], a);
                o[n] = (elem_t)a;
            }
            *reinterpret_cast<elem8*>(y + m * (8 * N8) + 8 * c) = o;
        }
    }
}

extern "C" {

// x [M, 8], w [N, 8], bias [N] or NULL, y [M, N]; N = 8 | 16; 16-byte aligned rows.
int DM_T(dm_linear_small_, )(const void* x, const void* w, const void* bias, void* y, long long M, int K, int N, hipStream_t stream) {
    if (!x || !w || !y || M <= 0) return DM_ERR_ARG;
    if (K != 8 || (N != 8 && N != 16)) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return DM_ERR_ARG;
    const unsigned grid = (unsigned)std::min<long long>((M + 255) / 256, 256 * 8);
    DM_ENTER();
    if (N == 8) hipLaunchKernelGGL(k_linear_small<1>, dim3(grid), dim3(256), 0, stream, (const elem_t*)x, (const elem_t*)w, (const elem_t*)bias, (elem_t*)y, M, N);
    else hipLaunchKernelGGL(k_linear_small<2>, dim3(grid), dim3(256), 0, stream, (const elem_t*)x, (const elem_t*)w, (const elem_t*)bias, (elem_t*)y, M, N);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// x, y [rows, C] bf16 (row-contiguous), gamma/beta [C] bf16; C % 8 == 0, C <= 2048.
int DM_T(dm_layernorm_, )(const void* x, const void* gamma, const void* beta, void* y, long long rows, int C, float eps,
                      hipStream_t stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || C <= 0) return DM_ERR_ARG;
    if (C % 8 != 0 || C > 2048) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return DM_ERR_ARG;
    if (rows == 0) return DM_OK;
    const int nch = (C / 8 + 63) / 64;
    const unsigned grid = (unsigned)std::min<long long>((rows + 3) / 4, 256 * 16);
    DM_ENTER();
#define DM_LN(N) hipLaunchKernelGGL(k_layernorm<N>, dim3(grid), dim3(256), 0, stream, (const elem_t*)x, \
                                    (const elem_t*)gamma, (const elem_t*)beta, (elem_t*)y, rows, C, eps)
    switch (nch) {
    case 1: DM_LN(1); break;
    case 2: DM_LN(2); break;
    case 3: DM_LN(3); break;
    default: DM_LN(4); break;
    }
#undef DM_LN
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// h [rows, 2*inner] bf16 (value half | gate half), y [rows, inner] bf16; inner % 8 == 0.
int DM_T(dm_geglu_, )(const void* h, void* y, long long rows, int inner, hipStream_t stream) {
    if (!h || !y || rows < 0 || inner <= 0) return DM_ERR_ARG;
    if (inner % 8 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)h | (uintptr_t)y) & 15) return DM_ERR_ARG;
    if (rows == 0) return DM_OK;
    const long long total = rows * (inner / 8);
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    DM_ENTER();
    hipLaunchKernelGGL(k_geglu, dim3(grid), dim3(256), 0, stream, (const elem_t*)h, (elem_t*)y, rows, inner);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// s, p [rows, cols] bf16 row-contiguous: p = softmax(scale * s) over the columns, fp32 arithmetic.  cols % 8 == 0, cols <= 16384; p may alias s.
int DM_T(dm_softmax_rows_, )(const void* s, void* p, long long rows, int cols, float scale, hipStream_t stream) {
    if (!s || !p || rows < 0 || cols <= 0) return DM_ERR_ARG;
    if (cols % 8 != 0 || cols > 16384) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)s | (uintptr_t)p) & 15) return DM_ERR_ARG;
    if (rows == 0) return DM_OK;
    const int nch = (cols / 8 + 255) / 256;
    const unsigned grid = (unsigned)std::min<long long>(rows, 256 * 8);
    DM_ENTER();
#define DM_SM(N) hipLaunchKernelGGL((k_softmax_rows<N, false>), dim3(grid), dim3(256), 0, stream, (const elem_t*)s, \
                                    (const elem_t*)nullptr, (elem_t*)p, rows, cols, scale)
    switch (nch) {
    case 1: DM_SM(1); break;
    case 2: DM_SM(2); break;
    case 3: case 4: DM_SM(4); break;
    default: DM_SM(8); break;
    }
#undef DM_SM
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Backward of the above: ds = scale * p * (dp - rowsum(p * dp)); p, dp, ds [rows, cols] bf16 (ds may alias dp).
int DM_T(dm_softmax_rows_bwd_, )(const void* p, const void* dp, void* ds, long long rows, int cols, float scale, hipStream_t stream) {
    if (!p || !dp || !ds || rows < 0 || cols <= 0) return DM_ERR_ARG;
    if (cols % 8 != 0 || cols > 16384) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)p | (uintptr_t)dp | (uintptr_t)ds) & 15) return DM_ERR_ARG;
    if (rows == 0) return DM_OK;
    const int nch = (cols / 8 + 255) / 256;
    const unsigned grid = (unsigned)std::min<long long>(rows, 256 * 8);
    DM_ENTER();
#define DM_SM(N) hipLaunchKernelGGL((k_softmax_rows<N, true>), dim3(grid), dim3(256), 0, stream, (const elem_t*)p, \
                                    (const elem_t*)dp, (elem_t*)ds, rows, cols, scale)
    switch (nch) {
    case 1: DM_SM(1); break;
    case 2: DM_SM(2); break;
    case 3: case 4: DM_SM(4); break;
    default: DM_SM(8); break;
    }
#undef DM_SM
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// src [batch, R, C] -> dst [batch, C, R], 16-bit elements, R % 64 == C % 64 == 0, 16-byte aligned, src != dst.
int DM_T(dm_transpose_, )(const void* src, void* dst, int batch, int R, int C, hipStream_t stream) {
    if (!src || !dst || src == dst || batch <= 0 || R <= 0 || C <= 0) return DM_ERR_ARG;
    if (R % 64 != 0 || C % 64 != 0 || batch > 65535 || R / 64 > 65535) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return DM_ERR_ARG;
    DM_ENTER();
    hipLaunchKernelGGL(k_transpose16, dim3(C / 64, R / 64, batch), dim3(256), 0, stream, (const elem_t*)src, (elem_t*)dst, R, C);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// x [rows, Cx], s [rows, Cs], r [rows, Cs] or NULL, y [rows, Cx + Cs], all bf16 row-contiguous, Cx % 8 == Cs % 8 == 0.
// The sum is rounded to bf16 once (the unfused pair rounded s + r to bf16 as well: same values when r_scale == 1).
int DM_T(dm_cat_add_, )(const void* x, const void* s, const void* r, void* y, long long rows, int Cx, int Cs, float r_scale,
                    hipStream_t stream) {
    if (!x || !s || !y || rows < 0 || Cx <= 0 || Cs <= 0) return DM_ERR_ARG;
    if (Cx % 8 != 0 || Cs % 8 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)s | (uintptr_t)r | (uintptr_t)y) & 15) return DM_ERR_ARG;
    if (rows == 0) return DM_OK;
    const long long total = rows * ((Cx + Cs) / 8);
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    DM_ENTER();
    hipLaunchKernelGGL(k_cat_add, dim3(grid), dim3(256), 0, stream, (const elem_t*)x, (const elem_t*)s, (const elem_t*)r, (elem_t*)y,
                       rows, Cx, Cs, r_scale);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
