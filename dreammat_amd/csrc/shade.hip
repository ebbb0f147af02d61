// Fused G-buffer -> split-sum PBR shade kernel (forward + backward) for gfx950.
// One thread per covered pixel; all views of the step in ONE launch.  HBM-bound by design:
// algorithmic traffic = n(12) + v(12) + features(20) in, colour(12) out = 56 B / covered pixel
// forward, 76 B backward; LUT / cubemap taps (1-texel face borders so a bilinear footprint never branches)
// are served by L2/MALL.  What bounds the kernel is the number of scattered cache lines per wave the texture
// address unit visits: 16 gather instructions per pixel with RGBA-fp32 texels and plain LUT taps, 8 with 8-byte
// texels (one 16 B load per bilinear row) and the FG x-pair table -- see shade_core.h.
// Reference: threestudio/models/materials/dreammat_material.py:679-711, 746-762.
#include "shade_core.h"

using namespace dm;

struct Strided {            // element (row i, channel c) at p[i*rs + c*cs]
    const float* p;
    long long rs, cs;
};
struct StridedOut {
    float* p;
    long long rs, cs;
};

struct ShadeArgs {
    EnvAtlas atlas;
    MatCfg mat;
    Strided nrm, view, feat;
    const int* pix_idx;      // [N] global pixel index (b*HW + y*W + x)
    const int* env_of_view;  // [B]
    const int* n_dev;        // device count of rows
    int HW;
    StridedOut color;        // [N,3]
    // optional debug outputs (null => skipped); rows of 3/3/3/3/1/1 floats, dense [N,C]
    float* albedo; float* spec_light; float* diff_light; float* spec_color; float* diff_color;
    float* metallic; float* roughness;
    // backward
    Strided dcolor;
    StridedOut dfeat;
};

// Both kernels are persistent grid-stride loops with a one-pixel software prefetch: the 11 input floats
// (+ pixel index) of the NEXT pixel are requested before the current pixel's ~600 VALU instructions and 16
// texel gathers run.  A one-thread-per-pixel launch has all waves streaming inputs, then all computing, then
// all gathering at the same time (measured: runtime ~ sum of the three phases); the prefetch overlaps them.
struct ShadeIn {
    F3 n, v;
    float f[5];
    F3 dc;
    int pix;
};

template <bool BWD>
__device__ __forceinline__ void shade_load(const ShadeArgs& a, long long i, ShadeIn& in) {
    in.n = f3(a.nrm.p[i * a.nrm.rs], a.nrm.p[i * a.nrm.rs + a.nrm.cs], a.nrm.p[i * a.nrm.rs + 2 * a.nrm.cs]);
    in.v = f3(a.view.p[i * a.view.rs], a.view.p[i * a.view.rs + a.view.cs], a.view.p[i * a.view.rs + 2 * a.view.cs]);
#pragma unroll
    for (int k = 0; k < 5; ++k) in.f[k] = a.feat.p[i * a.feat.rs + k * a.feat.cs];
    if (BWD)
        in.dc = f3(a.dcolor.p[i * a.dcolor.rs], a.dcolor.p[i * a.dcolor.rs + a.dcolor.cs],
                   a.dcolor.p[i * a.dcolor.rs + 2 * a.dcolor.cs]);
    in.pix = a.pix_idx[i];
}

template <int FMT>
__global__ __launch_bounds__(256) void k_shade_fwd(ShadeArgs a) {
    const long long N = *a.n_dev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    ShadeIn cur, nxt;
    shade_load<false>(a, i, cur);
    for (; i < N; i += stride) {
        const bool more = i + stride < N;
        if (more) shade_load<false>(a, i + stride, nxt);
        int env = a.env_of_view[cur.pix / a.HW];
        ShadeCtx c;
        shade_eval_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c);
        a.color.p[i * a.color.rs] = sat(c.pre.x);
        a.color.p[i * a.color.rs + a.color.cs] = sat(c.pre.y);
        a.color.p[i * a.color.rs + 2 * a.color.cs] = sat(c.pre.z);
        if (a.albedo) {
            F3 sl = lin2srgb(c.spec), dl = lin2srgb(c.diff), sc = lin2srgb(c.spec_albedo), dc = lin2srgb(c.albedo);
            a.albedo[3 * i] = c.albedo.x; a.albedo[3 * i + 1] = c.albedo.y; a.albedo[3 * i + 2] = c.albedo.z;
            a.spec_light[3 * i] = sl.x; a.spec_light[3 * i + 1] = sl.y; a.spec_light[3 * i + 2] = sl.z;
            a.diff_light[3 * i] = dl.x; a.diff_light[3 * i + 1] = dl.y; a.diff_light[3 * i + 2] = dl.z;
            a.spec_color[3 * i] = sc.x; a.spec_color[3 * i + 1] = sc.y; a.spec_color[3 * i + 2] = sc.z;
            a.diff_color[3 * i] = dc.x; a.diff_color[3 * i + 1] = dc.y; a.diff_color[3 * i + 2] = dc.z;
            a.metallic[i] = c.metallic;
            a.roughness[i] = c.roughness;
        }
        if (more) cur = nxt;
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_shade_bwd(ShadeArgs a) {
    const long long N = *a.n_dev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    ShadeIn cur, nxt;
    shade_load<true>(a, i, cur);
    for (; i < N; i += stride) {
        const bool more = i + stride < N;
        if (more) shade_load<true>(a, i + stride, nxt);
        int env = a.env_of_view[cur.pix / a.HW];
        ShadeCtx c;
        shade_eval_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c);
        float df[5];
        shade_backward(a.mat, c, cur.dc, df);
#pragma unroll
        for (int k = 0; k < 5; ++k) a.dfeat.p[i * a.dfeat.rs + k * a.dfeat.cs] = df[k];
        if (more) cur = nxt;
    }
}

// persistent launch: enough workgroups to fill every CU at the kernels' occupancy, never more than needed
static inline int shade_blocks(long long n_max) {
    long long need = (n_max + 255) / 256;
    return (int)std::min<long long>(need, 256 * 4);   // 106-108 VGPRs -> 4 waves per SIMD = 4 workgroups per CU
}

// Material smoothness regulariser (dreammat_material.py:110-123) fused: forward partial sums and
// analytic gradient wrt both feature sets.  loss = mean(kd_luma*kd_b)*0.25 + mean(ks_0*ks_1)*0.1
__global__ __launch_bounds__(256) void k_matreg(Strided feat, Strided featj, const int* n_dev, float gscale_in,
                                                float* __restrict__ loss_accum /*[1]*/, StridedOut dfeat,
                                                StridedOut dfeatj, int want_grad) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int N = *n_dev;
    float part = 0.f;
    if (i < N) {
        float s[5], sj[5], d[5], sg[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            s[k] = sigmoidf(feat.p[i * feat.rs + k * feat.cs]);
            sj[k] = sigmoidf(featj.p[i * featj.rs + k * featj.cs]);
            float df = s[k] - sj[k];
            d[k] = fabsf(df);
            sg[k] = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
        }
        float luma = (d[0] + d[1] + d[2]) / 3.f;
        float invN = 1.0f / (float)N;
        part = (luma * d[2] * 0.25f + d[3] * d[4] * 0.1f) * invN;
        if (want_grad) {
            float g = gscale_in * invN;
            float gd[5];
            gd[0] = g * 0.25f * d[2] / 3.f;
            gd[1] = g * 0.25f * d[2] / 3.f;
            gd[2] = g * 0.25f * (d[2] / 3.f + luma);
            gd[3] = g * 0.1f * d[4];
            gd[4] = g * 0.1f * d[3];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float gs = gd[k] * sg[k];
                dfeat.p[i * dfeat.rs + k * dfeat.cs] = gs * s[k] * (1.f - s[k]);
                dfeatj.p[i * dfeatj.rs + k * dfeatj.cs] = -gs * sj[k] * (1.f - sj[k]);
            }
        }
    }
    if (!want_grad) {
        for (int ofs = 32; ofs > 0; ofs >>= 1) part += __shfl_xor(part, ofs);
        __shared__ float wsum[4];
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_accum, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    }
}

extern "C" {

// Mirror of EnvAtlas for the C ABI (plain pointers and sizes).
struct dm_env_atlas {
    const float* spec; const float* diff; const float* fg_lut;
    long long spec_env_stride, diff_env_stride;
    long long mip_off[8];
    int mip_res[8];
    int n_mips, diff_res, lut_res;
    float min_rough_mip, max_rough_mip;
    int texel_format;
    const float* fg_pairs;
};
struct dm_mat_cfg { float min_metallic, max_metallic, min_roughness, max_roughness; };

static bool conv_atlas(const dm_env_atlas* in, EnvAtlas& A) {
    if (!in || !in->spec || !in->diff || !in->fg_lut || in->n_mips < 2 || in->n_mips > kMaxMips) return false;
    A.spec = (const float4*)in->spec; A.diff = (const float4*)in->diff; A.fg_lut = (const float2*)in->fg_lut;
    A.spec_env_stride = in->spec_env_stride; A.diff_env_stride = in->diff_env_stride;
    for (int i = 0; i < kMaxMips; ++i) { A.mip_off[i] = in->mip_off[i]; A.mip_res[i] = in->mip_res[i]; }
    A.n_mips = in->n_mips; A.diff_res = in->diff_res; A.lut_res = in->lut_res;
    A.min_rough_mip = in->min_rough_mip; A.max_rough_mip = in->max_rough_mip;
    if (in->texel_format < 0 || in->texel_format > 2) return false;
    A.texel_format = in->texel_format;
    A.fg_pairs = (const float4*)in->fg_pairs;
    return true;
}

// Forward.  Row tensors are addressed as p[i*row_stride + c*col_stride] (elements), so both the
// reference's [N,C] layout and the internal SoA [C,N] layout are accepted without copies.
int dm_shade_fwd(const dm_env_atlas* atlas, const dm_mat_cfg* mat, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, float* color, long long color_rs,
                 long long color_cs, float* dbg_albedo, float* dbg_spec_light, float* dbg_diff_light,
                 float* dbg_spec_color, float* dbg_diff_color, float* dbg_metallic, float* dbg_roughness,
                 hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !color || n_max <= 0 || HW <= 0)
        return DM_ERR_ARG;
    int ndbg = (dbg_albedo != 0) + (dbg_spec_light != 0) + (dbg_diff_light != 0) + (dbg_spec_color != 0) +
               (dbg_diff_color != 0) + (dbg_metallic != 0) + (dbg_roughness != 0);
    if (ndbg != 0 && ndbg != 7) return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW;
    a.color = {color, color_rs, color_cs};
    a.albedo = dbg_albedo; a.spec_light = dbg_spec_light; a.diff_light = dbg_diff_light;
    a.spec_color = dbg_spec_color; a.diff_color = dbg_diff_color; a.metallic = dbg_metallic;
    a.roughness = dbg_roughness;
    DM_ENTER();
    const dim3 grid(shade_blocks(n_max));
    switch (a.atlas.texel_format) {
        case kTexelRgb18e8: hipLaunchKernelGGL(k_shade_fwd<kTexelRgb18e8>, grid, dim3(256), 0, stream, a); break;
        case kTexelF16: hipLaunchKernelGGL(k_shade_fwd<kTexelF16>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(k_shade_fwd<kTexelF32>, grid, dim3(256), 0, stream, a); break;
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_shade_bwd(const dm_env_atlas* atlas, const dm_mat_cfg* mat, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, const float* dcolor, long long dcolor_rs,
                 long long dcolor_cs, float* dfeat, long long dfeat_rs, long long dfeat_cs, hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !dcolor || !dfeat || n_max <= 0 || HW <= 0)
        return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW;
    a.dcolor = {dcolor, dcolor_rs, dcolor_cs};
    a.dfeat = {dfeat, dfeat_rs, dfeat_cs};
    DM_ENTER();
    const dim3 grid(shade_blocks(n_max));
    switch (a.atlas.texel_format) {
        case kTexelRgb18e8: hipLaunchKernelGGL(k_shade_bwd<kTexelRgb18e8>, grid, dim3(256), 0, stream, a); break;
        case kTexelF16: hipLaunchKernelGGL(k_shade_bwd<kTexelF16>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(k_shade_bwd<kTexelF32>, grid, dim3(256), 0, stream, a); break;
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// loss_out (device float) must be zeroed by the caller before the forward call.
int dm_matreg_fwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float* loss_out, hipStream_t stream) {
    if (!feat || !featj || !n_dev || !loss_out || n_max <= 0) return DM_ERR_ARG;
    Strided f = {feat, f_rs, f_cs}, j = {featj, j_rs, j_cs};
    StridedOut z = {nullptr, 0, 0};
    DM_ENTER();
    hipLaunchKernelGGL(k_matreg, dim3(dm_div_up(n_max, 256)), dim3(256), 0, stream, f, j, n_dev, 0.f, loss_out, z, z, 0);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_matreg_bwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float grad_scale, float* dfeat,
                  long long df_rs, long long df_cs, float* dfeatj, long long dj_rs, long long dj_cs,
                  hipStream_t stream) {
    if (!feat || !featj || !n_dev || !dfeat || !dfeatj || n_max <= 0) return DM_ERR_ARG;
    Strided f = {feat, f_rs, f_cs}, j = {featj, j_rs, j_cs};
    StridedOut df = {dfeat, df_rs, df_cs}, dj = {dfeatj, dj_rs, dj_cs};
    DM_ENTER();
    hipLaunchKernelGGL(k_matreg, dim3(dm_div_up(n_max, 256)), dim3(256), 0, stream, f, j, n_dev, grad_scale, nullptr,
                       df, dj, 1);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
