// Fused GroupNorm(32 groups) [+ SiLU] on NHWC bf16 activations for gfx950 (forward and backward).
// Covers every `F.silu(GroupNorm(x))` of the ResnetBlock2D / conv_norm_out layers and the plain GroupNorm
// in front of each Transformer2DModel / VAE attention that diffusers runs for
// threestudio/models/guidance/dreammat_guidance.py:205-292.  Bandwidth-bound: forward = 2 reads + 1 write
// of the activation (statistics pass + apply pass), all 16 B vector accesses; keeping the activations
// NHWC end-to-end removes the NCHW<->NHWC copies around the implicit-GEMM convolutions (csrc/conv.hip).
//   x [B, HW, C] bf16, gamma/beta [C] bf16, G = 32 groups of C/32 contiguous channels, eps.
// stats[b][g] = (sum, sumsq) fp32 accumulated with atomics (caller zeroes), then
//   y = act((x - mean) * rstd * gamma + beta),  act = identity | SiLU.
// Backward (the VAE encoder is differentiated through; weights are frozen => only dx):
//   dz = dy * act'(z);  dxhat = dz * gamma;  dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat)).
#include "dm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// workspace (fp32): stats [B,32,2] | bstats [B,32,2] | coef [B,7,C]
//   coef rows: 0 A = rstd*gamma   1 S = beta - mean*rstd*gamma   2 rstd   3 mean*rstd   4 gamma
//              5 rstd*mean_g(dxhat)   6 rstd*mean_g(dxhat*xhat)
struct GnArgs {
    const __bf16* x; const __bf16* gamma; const __bf16* beta; const __bf16* dy;
    __bf16* y;            // forward output / backward dx
    float* stats; float* bstats; float* coef;
    int B, HW, C, act;
    float eps;
    int rows_per_block;
};

__device__ __forceinline__ float siluf(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) {
    float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

__device__ __forceinline__ void load8(const float* p, float* o) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// per-(b, channel) coefficients from the group statistics.  PASS 0: rows 0-4, PASS 1: rows 5-6.
template <int PASS>
__global__ void k_gn_coef(GnArgs a) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const int cpg = a.C / 32, g = c / cpg;
    const float n = (float)a.HW * cpg;
    float m = a.stats[((long long)b * 32 + g) * 2] / n;
    float var = a.stats[((long long)b * 32 + g) * 2 + 1] / n - m * m;
    float rs = rsqrtf(fmaxf(var, 0.f) + a.eps);
    float* co = a.coef + (long long)b * 7 * a.C;
    if (PASS == 0) {
        float gm = (float)a.gamma[c], bt = (float)a.beta[c];
        co[c] = rs * gm; co[a.C + c] = bt - m * rs * gm; co[2 * a.C + c] = rs; co[3 * a.C + c] = m * rs;
        co[4 * a.C + c] = gm;
    } else {
        co[5 * a.C + c] = rs * a.bstats[((long long)b * 32 + g) * 2] / n;
        co[6 * a.C + c] = rs * a.bstats[((long long)b * 32 + g) * 2 + 1] / n;
    }
}

// MODE 0: forward statistics (sum x, sum x^2).  MODE 1: backward statistics (sum dxhat, sum dxhat*xhat).
template <int MODE>
__global__ __launch_bounds__(256) void k_gn_stats(GnArgs a) {
    extern __shared__ float sh[];          // [2*C] per-channel partial sums
    const int b = blockIdx.y;
    const int C = a.C, cpg = C / 32;
    const int chunks = C / 8;              // 16 B chunks per pixel row
    for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.f;
    __syncthreads();
    const long long row0 = (long long)blockIdx.x * a.rows_per_block;
    const long long row1 = min((long long)a.HW, row0 + a.rows_per_block);
    const __bf16* xb = a.x + (long long)b * a.HW * C;
    const __bf16* dyb = MODE ? a.dy + (long long)b * a.HW * C : nullptr;
    const float* co = a.coef + (long long)b * 7 * C;
    // thread -> fixed channel chunk(s); rows are strided over the threads that share a chunk
    const int lanes_per_row = min(chunks, 256);
    const int row_par = 256 / lanes_per_row;                 // rows processed concurrently by the block
    const int my_row = threadIdx.x / lanes_per_row;
    if (my_row < row_par) {
        for (int ch = threadIdx.x % lanes_per_row; ch < chunks; ch += lanes_per_row) {
            float s0[8], s1[8], A[8], S[8], rs[8], mrs[8], gm[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
            if (MODE) {
                load8(co + ch * 8, A); load8(co + C + ch * 8, S); load8(co + 2 * C + ch * 8, rs);
                load8(co + 3 * C + ch * 8, mrs); load8(co + 4 * C + ch * 8, gm);
            }
            for (long long r = row0 + my_row; r < row1; r += row_par) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + r * C + ch * 8);
                if (!MODE) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { float f = (float)v[k]; s0[k] += f; s1[k] += f * f; }
                } else {
                    bf16x8 d = *reinterpret_cast<const bf16x8*>(dyb + r * C + ch * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float xf = (float)v[k];
                        float xh = xf * rs[k] - mrs[k];
                        float dz = (float)d[k];
                        if (a.act) dz *= silu_grad(xf * A[k] + S[k]);
                        float dxh = dz * gm[k];
                        s0[k] += dxh; s1[k] += dxh * xh;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                atomicAdd(&sh[ch * 8 + k], s0[k]);
                atomicAdd(&sh[C + ch * 8 + k], s1[k]);
            }
        }
    }
    __syncthreads();
    float* out = MODE ? a.bstats : a.stats;
    if (threadIdx.x < 32) {
        int g = threadIdx.x;
        float t0 = 0.f, t1 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { t0 += sh[c]; t1 += sh[C + c]; }
        atomicAdd(&out[((long long)b * 32 + g) * 2], t0);
        atomicAdd(&out[((long long)b * 32 + g) * 2 + 1], t1);
    }
}

// MODE 0: y = act(x*A + S).  MODE 1: dx = rstd*dxhat - c1 - c2*xhat.
template <int MODE>
__global__ __launch_bounds__(256) void k_gn_apply(GnArgs a) {
    const int b = blockIdx.y;
    const int C = a.C, chunks = C / 8;
    const __bf16* xb = a.x + (long long)b * a.HW * C;
    const __bf16* dyb = MODE ? a.dy + (long long)b * a.HW * C : nullptr;
    __bf16* yb = a.y + (long long)b * a.HW * C;
    const float* co = a.coef + (long long)b * 7 * C;
    const int lanes_per_row = min(chunks, 256);
    const int row_par = 256 / lanes_per_row;
    const int my_row = threadIdx.x / lanes_per_row;
    if (my_row >= row_par) return;
    const long long row0 = (long long)blockIdx.x * a.rows_per_block;
    const long long row1 = min((long long)a.HW, row0 + a.rows_per_block);
    for (int ch = threadIdx.x % lanes_per_row; ch < chunks; ch += lanes_per_row) {
        float A[8], S[8], rs[8], mrs[8], gm[8], c1[8], c2[8];
        load8(co + ch * 8, A); load8(co + C + ch * 8, S);
        if (MODE) {
            load8(co + 2 * C + ch * 8, rs); load8(co + 3 * C + ch * 8, mrs); load8(co + 4 * C + ch * 8, gm);
            load8(co + 5 * C + ch * 8, c1); load8(co + 6 * C + ch * 8, c2);
        }
        for (long long r = row0 + my_row; r < row1; r += row_par) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + r * C + ch * 8);
            bf16x8 o;
            if (!MODE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float z = (float)v[k] * A[k] + S[k];
                    o[k] = (__bf16)(a.act ? siluf(z) : z);
                }
            } else {
                bf16x8 d = *reinterpret_cast<const bf16x8*>(dyb + r * C + ch * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float xf = (float)v[k];
                    float xh = xf * rs[k] - mrs[k];
                    float dz = (float)d[k];
                    if (a.act) dz *= silu_grad(xf * A[k] + S[k]);
                    o[k] = (__bf16)(rs[k] * (dz * gm[k]) - c1[k] - c2[k] * xh);
                }
            }
            *reinterpret_cast<bf16x8*>(yb + r * C + ch * 8) = o;
        }
    }
}

bool check_args(int B, int HW, int C) { return B > 0 && HW > 0 && C > 0 && C % 32 == 0 && C <= 8192; }

void launch_cfg(int B, int HW, int C, dim3& grid, int& rows_per_block) {
    // ~2048 workgroups over the chip, at least 8 rows each
    int blocks_per_b = (int)std::max<long long>(1, std::min<long long>((HW + 7) / 8, 2048 / std::max(1, B) + 1));
    rows_per_block = (HW + blocks_per_b - 1) / blocks_per_b;
    grid = dim3((unsigned)((HW + rows_per_block - 1) / rows_per_block), B);
}

}  // namespace

extern "C" {

size_t dm_groupnorm_workspace_floats(int B, int C) { return (size_t)B * (128 + 7 * (size_t)C); }

// ws: dm_groupnorm_workspace_floats(B,C) fp32 (kept by the caller for the backward).  act: 0 none, 1 SiLU.
int dm_groupnorm_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW,
                          int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const __bf16*)x; a.gamma = (const __bf16*)gamma; a.beta = (const __bf16*)beta; a.y = (__bf16*)y;
    a.stats = ws; a.bstats = ws + (size_t)B * 64; a.coef = ws + (size_t)B * 128;
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    DM_ENTER();
    DM_HIP(hipMemsetAsync(ws, 0, (size_t)B * 128 * sizeof(float), stream));
    hipLaunchKernelGGL(k_gn_stats<0>, g, dim3(256), 2 * C * sizeof(float), stream, a);
    hipLaunchKernelGGL(k_gn_coef<0>, dim3(dm_div_up(C, 256), B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_gn_apply<0>, g, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// dx from dy; ws = the workspace left by the matching dm_groupnorm_nhwc_fwd call (statistics + coefficients).
int dm_groupnorm_nhwc_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* ws,
                          int B, int HW, int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !dy || !dx || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const __bf16*)x; a.gamma = (const __bf16*)gamma; a.beta = (const __bf16*)beta; a.dy = (const __bf16*)dy;
    a.y = (__bf16*)dx; a.stats = ws; a.bstats = ws + (size_t)B * 64; a.coef = ws + (size_t)B * 128;
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    DM_ENTER();
    DM_HIP(hipMemsetAsync(a.bstats, 0, (size_t)B * 64 * sizeof(float), stream));
    hipLaunchKernelGGL(k_gn_stats<1>, g, dim3(256), 2 * C * sizeof(float), stream, a);
    hipLaunchKernelGGL(k_gn_coef<1>, dim3(dm_div_up(C, 256), B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_gn_apply<1>, g, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
