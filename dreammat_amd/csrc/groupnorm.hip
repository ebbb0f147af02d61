// Fused GroupNorm(32 groups) [+ SiLU] on NHWC bf16 activations for gfx950 (forward and backward).
// Covers every `F.silu(GroupNorm(x))` of the ResnetBlock2D / conv_norm_out layers and the plain GroupNorm
// in front of each Transformer2DModel / VAE attention that diffusers runs for
// threestudio/models/guidance/dreammat_guidance.py:205-292.  Bandwidth-bound: forward = 2 reads + 1 write
// of the activation (statistics pass + apply pass), all 16 B vector accesses; keeping the activations
// NHWC end-to-end removes the NCHW<->NHWC copies around the implicit-GEMM convolutions (csrc/conv.hip).
//   x [B, HW, C] bf16, gamma/beta [C] bf16, G = 32 groups of C/32 contiguous channels, eps.
// Group statistics (sum, sumsq) are reduced WITHOUT atomics: every workgroup writes one fp32 partial per group,
// the coefficient kernel adds the partials in a fixed order (deterministic, no memset, no same-address
// atomic traffic across the 8 XCDs -- the first version spent most of its time there), then
//   y = act((x - mean) * rstd * gamma + beta),  act = identity | SiLU.
// Backward (the VAE encoder is differentiated through; weights are frozen => only dx):
//   dz = dy * act'(z);  dxhat = dz * gamma;  dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat)).
#include "dm_common.h"
#include "dm_elem.h"

namespace {


// workspace (fp32): coef [B,7,C] | part [B*nblk,32,2] | bpart [B*nblk,32,2]   (nblk = workgroups per batch item)
//   coef rows: 0 A = rstd*gamma   1 S = beta - mean*rstd*gamma   2 rstd   3 mean*rstd   4 gamma
//              5 rstd*mean_g(dxhat)   6 rstd*mean_g(dxhat*xhat)
struct GnArgs {
    const elem_t* x; const elem_t* gamma; const elem_t* beta; const elem_t* dy;
    const elem_t* dres;   // backward only, may be NULL: a second gradient of x (the skip branch of a ResnetBlock2D), added into dx
    elem_t* y;            // forward output / backward dx
    float* part; float* bpart; float* coef;
    float* cpart;         // MODE 2 of k_gn_stats: per-(b, workgroup) channel sums [B*nblk][2][C] (dbeta, dgamma partials)
    int B, HW, C, act, nblk;
    float eps;
    int rows_per_block;
};

// sigmoid on the hardware's 2^x and 1/x (1 ulp each; the result is rounded to 16 bits).  `z / (1.f + __expf(-z))` compiled to an
// IEEE division -- v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup -- and made the apply kernels VALU-bound: 32 instructions
// per element forward, 50 backward, against 4-8 bytes of traffic (round 5: 1.2 T elements/s x 32 = the chip's whole VALU rate).
__device__ __forceinline__ float sigmoid_fast(float z) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));      // z -> -inf: rcp(inf) = 0
}
__device__ __forceinline__ float siluf(float z) { return z * sigmoid_fast(z); }
__device__ __forceinline__ float silu_grad(float z) {
    const float s = sigmoid_fast(z);
    return s * (1.f + z * (1.f - s));
}

__device__ __forceinline__ void load8(const float* p, float* o) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// per-(b, channel) coefficients from the group statistics.  PASS 0: rows 0-4, PASS 1: rows 5-6.
constexpr int GN_COEF_SLICES = 16;       // k_gn_coef: 1024 threads = 64 statistics x 16 interleaved slices of the partial rows

template <int PASS>
__global__ __launch_bounds__(64 * GN_COEF_SLICES) void k_gn_coef(GnArgs a) {
    // the 64 group statistics of this batch item: 64 statistics x 16 interleaved slices of the per-workgroup partials,
    // combined in a fixed order (a per-channel serial walk over up to 257 partials cost 26 us per call; 4 slices over the
    // 1537 rows of a one-image 512^2 tensor still 21 us, 44 times per step at one view per rank)
    __shared__ float red[GN_COEF_SLICES][64];
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    {
        const int st = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const float* pp = (PASS == 0 ? a.part : a.bpart) + (long long)b * a.nblk * 64 + st;
        float t = 0.f;
#pragma unroll 4
        for (int k = sl; k < a.nblk; k += GN_COEF_SLICES) t += pp[(long long)k * 64];
        red[sl][st] = t;
    }
    __syncthreads();
    if (c >= a.C) return;
    const int cpg = a.C / 32, g = c / cpg;
    const float n = (float)a.HW * cpg;
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int q = 0; q < GN_COEF_SLICES; q += 4) {
        t0 += (red[q][2 * g] + red[q + 1][2 * g]) + (red[q + 2][2 * g] + red[q + 3][2 * g]);
        t1 += (red[q][2 * g + 1] + red[q + 1][2 * g + 1]) + (red[q + 2][2 * g + 1] + red[q + 3][2 * g + 1]);
    }
    float* co = a.coef + (long long)b * 7 * a.C;
    if (PASS == 0) {
        float m = t0 / n;
        float var = t1 / n - m * m;
        float rs = rsqrtf(fmaxf(var, 0.f) + a.eps);
        float gm = (float)a.gamma[c], bt = (float)a.beta[c];
        co[c] = rs * gm; co[a.C + c] = bt - m * rs * gm; co[2 * a.C + c] = rs; co[3 * a.C + c] = m * rs;
        co[4 * a.C + c] = gm;
    } else {
        float rs = co[2 * a.C + c];
        co[5 * a.C + c] = rs * t0 / n;
        co[6 * a.C + c] = rs * t1 / n;
    }
}

// MODE 0: forward statistics (sum x, sum x^2).  MODE 1: backward statistics (sum dxhat, sum dxhat*xhat).
// MODE 2 (trainable affine parameters, ControlNet training): per-CHANNEL sums of dz and dz*xhat -- the partials of dbeta /
// dgamma -- written per workgroup (the caller adds them: fixed order, no atomics).
template <int MODE>
__global__ __launch_bounds__(256) void k_gn_stats(GnArgs a) {
    extern __shared__ float sh[];          // [row_par][2*C] per-thread partial sums (each slot written once)
    const int b = blockIdx.y;
    const int C = a.C, cpg = C / 32;
    const int chunks = C / 8;              // 16 B chunks per pixel row
    const long long row0 = (long long)blockIdx.x * a.rows_per_block;
    const long long row1 = min((long long)a.HW, row0 + a.rows_per_block);
    const elem_t* __restrict__ xb = a.x + (long long)b * a.HW * C;
    const elem_t* __restrict__ dyb = MODE ? a.dy + (long long)b * a.HW * C : nullptr;
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
            // 4 rows per trip: four independent 16 B loads in flight per thread (the single-load loop was
            // latency-bound at ~1.3 TB/s on the UNet-sized tensors)
            for (long long r = row0 + my_row; r < row1; r += 4LL * row_par) {
                elem8 v[4], d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    long long rr = r + (long long)u * row_par;
                    bool ok = rr < row1;
                    long long rc = ok ? rr : row0;
                    v[u] = *reinterpret_cast<const elem8*>(xb + rc * C + ch * 8);
                    if (MODE) d[u] = *reinterpret_cast<const elem8*>(dyb + rc * C + ch * 8);
                    if (!ok) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { v[u][k] = (elem_t)0.f; if (MODE) d[u][k] = (elem_t)0.f; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!MODE) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { float f = (float)v[u][k]; s0[k] += f; s1[k] += f * f; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float xf = (float)v[u][k];
                            float xh = xf * rs[k] - mrs[k];
                            float dz = (float)d[u][k];
                            if (a.act) dz *= silu_grad(xf * A[k] + S[k]);
                            float dxh = MODE == 2 ? dz : dz * gm[k];
                            s0[k] += dxh; s1[k] += dxh * xh;
                        }
                    }
                }
            }
            float* o = sh + (long long)my_row * 2 * C + ch * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[k] = s0[k]; o[C + k] = s1[k]; }
        }
    }
    __syncthreads();
    if (MODE == 2) {
        float* out = a.cpart + ((long long)b * a.nblk + blockIdx.x) * 2 * C;
        for (int c = threadIdx.x; c < 2 * C; c += 256) {
            float t = 0.f;
            for (int rp = 0; rp < row_par; ++rp) t += sh[(long long)rp * 2 * C + c];
            out[c] = t;
        }
        return;
    }
    // fixed-order reduction: thread t < 64 -> (group t>>1, statistic t&1)
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
        float t = 0.f;
        for (int rp = 0; rp < row_par; ++rp) {
            const float* src = sh + (long long)rp * 2 * C + which * C + g * cpg;
            for (int c = 0; c < cpg; ++c) t += src[c];
        }
        float* out = MODE ? a.bpart : a.part;
        out[((long long)b * a.nblk + blockIdx.x) * 64 + g * 2 + which] = t;
    }
}

// MODE 0: y = act(x*A + S).  MODE 1: dx = rstd*dxhat - c1 - c2*xhat.
// FUSED (forward only): no coefficient kernel in front -- every workgroup adds up the per-workgroup partials of its batch
// item itself (64 statistics x 4 interleaved slices, a fixed order; k_gn_coef uses 16 slices, so the two differ in the last
// bits of the sums) and forms A, S from gamma / beta on
// the fly.  One launch less per GroupNorm; at 1 view per rank the coefficient kernel was a 9 us launch in front of a 7 us
// apply, 110 times per step.
template <int MODE, bool FUSED = false>
__global__ __launch_bounds__(256) void k_gn_apply(GnArgs a) {
    const int b = blockIdx.y;
    const int C = a.C, chunks = C / 8;
    __shared__ float red[4][64];
    __shared__ float mr[32][2];                // mean, rstd per group
    if (FUSED) {
        const int st = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const float* pp = a.part + (long long)b * a.nblk * 64 + st;
        float t = 0.f;
#pragma unroll 4
        for (int k = sl; k < a.nblk; k += 4) t += pp[(long long)k * 64];
        red[sl][st] = t;
        __syncthreads();
        if (threadIdx.x < 32) {
            const int g = threadIdx.x;
            const float n = (float)a.HW * (C / 32);
            const float t0 = (red[0][2 * g] + red[1][2 * g]) + (red[2][2 * g] + red[3][2 * g]);
            const float t1 = (red[0][2 * g + 1] + red[1][2 * g + 1]) + (red[2][2 * g + 1] + red[3][2 * g + 1]);
            const float m = t0 / n;
            const float var = t1 / n - m * m;
            mr[g][0] = m;
            mr[g][1] = rsqrtf(fmaxf(var, 0.f) + a.eps);
        }
        __syncthreads();
    }
    const elem_t* __restrict__ xb = a.x + (long long)b * a.HW * C;
    const elem_t* __restrict__ dyb = MODE ? a.dy + (long long)b * a.HW * C : nullptr;
    const elem_t* __restrict__ drb = (MODE && a.dres) ? a.dres + (long long)b * a.HW * C : nullptr;
    elem_t* __restrict__ yb = a.y + (long long)b * a.HW * C;
    const float* co = a.coef + (long long)b * 7 * C;
    const int lanes_per_row = min(chunks, 256);
    const int row_par = 256 / lanes_per_row;
    const int my_row = threadIdx.x / lanes_per_row;
    if (my_row >= row_par) return;
    const long long row0 = (long long)blockIdx.x * a.rows_per_block;
    const long long row1 = min((long long)a.HW, row0 + a.rows_per_block);
    for (int ch = threadIdx.x % lanes_per_row; ch < chunks; ch += lanes_per_row) {
        float A[8], S[8], rs[8], mrs[8], gm[8], c1[8], c2[8];
        if (FUSED) {
            const elem8 gv = *reinterpret_cast<const elem8*>(a.gamma + ch * 8), bv = *reinterpret_cast<const elem8*>(a.beta + ch * 8);
            const int cpg = C / 32;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int g = (ch * 8 + k) / cpg;
                const float m = mr[g][0], rs = mr[g][1], gm = (float)gv[k], bt = (float)bv[k];
                A[k] = rs * gm;                                     // same expressions as k_gn_coef<0>
                S[k] = bt - m * rs * gm;
            }
        } else {
            load8(co + ch * 8, A); load8(co + C + ch * 8, S);
        }
        if (MODE) {
            load8(co + 2 * C + ch * 8, rs); load8(co + 3 * C + ch * 8, mrs); load8(co + 4 * C + ch * 8, gm);
            load8(co + 5 * C + ch * 8, c1); load8(co + 6 * C + ch * 8, c2);
        }
#pragma unroll 4
        for (long long r = row0 + my_row; r < row1; r += row_par) {
            elem8 v = *reinterpret_cast<const elem8*>(xb + r * C + ch * 8);
            elem8 o;
            if (!MODE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float z = (float)v[k] * A[k] + S[k];
                    o[k] = (elem_t)(a.act ? siluf(z) : z);
                }
            } else {
                elem8 d = *reinterpret_cast<const elem8*>(dyb + r * C + ch * 8);
                elem8 e = {};
                if (drb) e = *reinterpret_cast<const elem8*>(drb + r * C + ch * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float xf = (float)v[k];
                    float xh = xf * rs[k] - mrs[k];
                    float dz = (float)d[k];
                    if (a.act) dz *= silu_grad(xf * A[k] + S[k]);
                    o[k] = (elem_t)(rs[k] * (dz * gm[k]) - c1[k] - c2[k] * xh + (float)e[k]);
                }
            }
            *reinterpret_cast<elem8*>(yb + r * C + ch * 8) = o;
        }
    }
}

bool check_args(int B, int HW, int C) { return B > 0 && HW > 0 && C > 0 && C % 32 == 0 && C <= 8192; }

constexpr int GN_MAX_BLOCKS = 1536;      // workgroups over the chip (all batch items together), at least 8 rows each

void launch_cfg(int B, int HW, int C, dim3& grid, int& rows_per_block) {
    int blocks_per_b = (int)std::max<long long>(1, std::min<long long>((HW + 7) / 8, GN_MAX_BLOCKS / std::max(1, B) + 1));
    rows_per_block = (HW + blocks_per_b - 1) / blocks_per_b;
    grid = dim3((unsigned)((HW + rows_per_block - 1) / rows_per_block), B);
}

size_t part_floats(int B) { return ((size_t)GN_MAX_BLOCKS + (size_t)B) * 64; }      // >= B * blocks_per_b * 64

size_t stats_lds_bytes(int C) {
    int chunks = C / 8, lanes_per_row = std::min(chunks, 256), row_par = 256 / lanes_per_row;
    return (size_t)row_par * 2 * C * sizeof(float);
}

void bind_ws(GnArgs& a, float* ws, int B, int C) {
    a.coef = ws;
    a.part = ws + (size_t)B * 7 * C;
    a.bpart = a.part + part_floats(B);
}

}  // namespace

extern "C" {

#if !defined(DM_F16)      // (dtype-independent: exported once)
size_t dm_groupnorm_workspace_floats(int B, int C) { return (size_t)B * 7 * (size_t)C + 2 * part_floats(B); }

#endif

// ws: dm_groupnorm_workspace_floats(B,C) fp32 (kept by the caller for the backward).  act: 0 none, 1 SiLU.
int DM_S(dm_groupnorm_nhwc_fwd)(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW,
                          int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const elem_t*)x; a.gamma = (const elem_t*)gamma; a.beta = (const elem_t*)beta; a.y = (elem_t*)y;
    bind_ws(a, ws, B, C);
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    a.nblk = (int)g.x;
    DM_ENTER();
    hipLaunchKernelGGL(k_gn_stats<0>, g, dim3(256), stats_lds_bytes(C), stream, a);
    hipLaunchKernelGGL(k_gn_coef<0>, dim3(dm_div_up(C, 64 * GN_COEF_SLICES), B), dim3(64 * GN_COEF_SLICES), 0, stream, a);
    hipLaunchKernelGGL(k_gn_apply<0>, g, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Statistics + coefficients only (round 6): ws as dm_groupnorm_nhwc_fwd leaves it (coefficient rows 0-4 and the partial sums), no
// output tensor -- the apply pass rides in the consuming convolution (dm_conv3x3_gn_nhwc_*_fused reads rows 0 and 1); the workspace
// also serves dm_groupnorm_nhwc_bwd(_res) of the same GroupNorm.
int DM_S(dm_groupnorm_nhwc_stats)(const void* x, const void* gamma, const void* beta, float* ws, int B, int HW, int C, float eps,
                            hipStream_t stream) {
    if (!x || !gamma || !beta || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const elem_t*)x; a.gamma = (const elem_t*)gamma; a.beta = (const elem_t*)beta; a.y = nullptr;
    bind_ws(a, ws, B, C);
    a.B = B; a.HW = HW; a.C = C; a.act = 0; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    a.nblk = (int)g.x;
    DM_ENTER();
    hipLaunchKernelGGL(k_gn_stats<0>, g, dim3(256), stats_lds_bytes(C), stream, a);
    hipLaunchKernelGGL(k_gn_coef<0>, dim3(dm_div_up(C, 64 * GN_COEF_SLICES), B), dim3(64 * GN_COEF_SLICES), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Forward only (frozen nets under no_grad: nothing will ask for the backward): statistics + apply, the coefficient kernel
// folded into the apply kernel.  Same arithmetic as dm_groupnorm_nhwc_fwd; ws is scratch of the same size.
int DM_S(dm_groupnorm_nhwc_infer)(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW,
                            int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    if (C % 8 != 0 || (((uintptr_t)gamma | (uintptr_t)beta) & 15)) return DM_ERR_UNSUPPORTED;
    GnArgs a = {};
    a.x = (const elem_t*)x; a.gamma = (const elem_t*)gamma; a.beta = (const elem_t*)beta; a.y = (elem_t*)y;
    bind_ws(a, ws, B, C);
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    a.nblk = (int)g.x;
    DM_ENTER();
    hipLaunchKernelGGL(k_gn_stats<0>, g, dim3(256), stats_lds_bytes(C), stream, a);
    hipLaunchKernelGGL((k_gn_apply<0, true>), g, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// dx from dy; ws = the workspace left by the matching dm_groupnorm_nhwc_fwd call (statistics + coefficients).
// dres (may be NULL, [B,HW,C] bf16): dx = GroupNorm^T(dy) + dres in the same pass, one rounding -- x of a ResnetBlock2D feeds
// norm1 AND the skip connection, and the sum of its two gradients was an ATen add pass per block of the VAE encoder's backward.
int DM_S(dm_groupnorm_nhwc_bwd_res)(const void* x, const void* gamma, const void* beta, const void* dy, const void* dres, void* dx,
                              float* ws, int B, int HW, int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !dy || !dx || !ws || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const elem_t*)x; a.gamma = (const elem_t*)gamma; a.beta = (const elem_t*)beta; a.dy = (const elem_t*)dy;
    a.dres = (const elem_t*)dres;
    a.y = (elem_t*)dx;
    bind_ws(a, ws, B, C);
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    a.nblk = (int)g.x;
    DM_ENTER();
    hipLaunchKernelGGL(k_gn_stats<1>, g, dim3(256), stats_lds_bytes(C), stream, a);
    hipLaunchKernelGGL(k_gn_coef<1>, dim3(dm_div_up(C, 64 * GN_COEF_SLICES), B), dim3(64 * GN_COEF_SLICES), 0, stream, a);
    hipLaunchKernelGGL(k_gn_apply<1>, g, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int DM_S(dm_groupnorm_nhwc_bwd)(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* ws,
                          int B, int HW, int C, float eps, int act, hipStream_t stream) {
    return DM_S(dm_groupnorm_nhwc_bwd_res)(x, gamma, beta, dy, nullptr, dx, ws, B, HW, C, eps, act, stream);
}

#if !defined(DM_F16)      // (the ControlNet training loop, row f-4, runs in bf16)
// dbeta / dgamma partials of a GroupNorm with TRAINABLE affine parameters (the ControlNet copy in the training loop), after
// dm_groupnorm_nhwc_fwd left its workspace: cpart [B * dm_groupnorm_affine_rows(B, HW, C) / B][2][C] fp32 --
// dbeta = cpart[:, 0].sum(0), dgamma = cpart[:, 1].sum(0).  dx comes from dm_groupnorm_nhwc_bwd as before.
int dm_groupnorm_affine_rows(int B, int HW, int C) {
    if (!check_args(B, HW, C)) return 0;
    dim3 g; int rpb;
    launch_cfg(B, HW, C, g, rpb);
    return (int)g.x * B;
}
int dm_groupnorm_nhwc_bwd_affine(const void* x, const void* gamma, const void* beta, const void* dy, float* ws, float* cpart,
                                 int B, int HW, int C, float eps, int act, hipStream_t stream) {
    if (!x || !gamma || !beta || !dy || !ws || !cpart || !check_args(B, HW, C)) return DM_ERR_ARG;
    GnArgs a = {};
    a.x = (const elem_t*)x; a.gamma = (const elem_t*)gamma; a.beta = (const elem_t*)beta; a.dy = (const elem_t*)dy;
    bind_ws(a, ws, B, C);
    a.cpart = cpart;
    a.B = B; a.HW = HW; a.C = C; a.act = act; a.eps = eps;
    dim3 g;
    launch_cfg(B, HW, C, g, a.rows_per_block);
    a.nblk = (int)g.x;
    DM_ENTER();
    hipLaunchKernelGGL(k_gn_stats<2>, g, dim3(256), stats_lds_bytes(C), stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
#endif

}  // extern "C"
