// Fused G-buffer -> split-sum PBR shade kernel (forward + backward) for gfx950.
// One thread per covered pixel; all views of the step in ONE launch.  HBM-bound by design:
// algorithmic traffic = n(12) + v(12) + features(20) in, colour(12) out = 56 B / covered pixel
// forward, 76 B backward; LUT / cubemap taps (1-texel face borders so a bilinear footprint never branches)
// are served by L2/MALL.  What bounds the kernel is the number of scattered cache lines per wave the texture
// address unit visits: 16 gather instructions per pixel with RGBA-fp32 texels and plain LUT taps, 8 with 8-byte
// texels (one 16 B load per bilinear row) and the FG x-pair table -- see shade_core.h.
// Reference: threestudio/models/materials/dreammat_material.py:679-711, 746-762.
#include <cstdlib>

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
    int n_views;             // entries of env_of_view
    int offsets32;           // inputs are SoA (unit row stride) and every byte offset fits 31 bits: the fast loop applies
    StridedOut color;        // [N,3]
    // optional debug outputs (null => skipped); rows of 3/3/3/3/1/1 floats, dense [N,C]
    float* albedo; float* spec_light; float* diff_light; float* spec_color; float* diff_color;
    float* metallic; float* roughness;
    // backward
    Strided dcolor;
    StridedOut dfeat;
};

// Both kernels are persistent grid-stride loops with a software prefetch of the 11 input floats (+ pixel index) of later
// pixels (see shade_pixel_loop).  A one-thread-per-pixel launch has all waves streaming inputs, then all computing, then
// all gathering at the same time (measured: runtime ~ sum of the three phases); the prefetch overlaps them.
constexpr int kMaxViewsLds = 256;

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
// Pixel loop of both kernels.  8-byte texel formats (production): two-stage form with a prefetch distance of TWO pixels,
// the prefetch issued BETWEEN the gathers of the current pixel and their first use.  vmcnt retires in order, so with the
// round-1 order (prefetch first, gathers second) the wait in front of the first texel decode also waited for the OLDER HBM
// input stream of the next pixel, every iteration: the streaming latency was never hidden (rocprofv3 on the real
// G-buffer: 28 us with an atlas of ONE texel per face, i.e. without any gather divergence at all, against 31 us with the real
// atlas).  Now the only loads older than the gathers were issued a whole iteration earlier.
// `out`: the kernel's per-pixel result rows (forward: colour, 3 channels; backward: d loss / d features, 5 channels); `body`
// receives a store functor (channel, value) for them -- 64-bit pointer arithmetic on the general path, buffer stores with one
// 32-bit lane offset per pixel on the fast path (three v_mad_u64_u32 per store cost more issue slots than the shading of a texel)
template <int FMT, bool BWD, class Body>
__device__ __forceinline__ void shade_pixel_loop(const ShadeArgs& a, const StridedOut& out, Body&& body) {
    const long long N = *a.n_dev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (FMT == kTexelF32 || !a.atlas.fg_pairs || a.n_views > kMaxViewsLds || !a.offsets32) {   // legacy: monolithic, one-pixel prefetch,
                                                                                                // any row strides
        if (i >= N) return;
        ShadeIn cur, nxt;
        shade_load<BWD>(a, i, cur);
        for (; i < N; i += stride) {
            const bool more = i + stride < N;
            if (more) shade_load<BWD>(a, i + stride, nxt);
            int env = a.env_of_view[cur.pix / a.HW];
            ShadeCtx c;
            shade_eval_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c);
            const long long ii = i;
            body(i, cur.dc, c, [&](int k, float v) { out.p[ii * out.rs + k * out.cs] = v; });
            if (more) cur = nxt;
        }
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t does not exist in the host pass)
    // per-mip tables and the view -> environment table in LDS: a per-lane index into the kernel-argument copies (or into
    // env_of_view) is a global load whose latency sits in front of every gather of the pixel
    __shared__ int s_mip_off[kMaxMips], s_mip_res[kMaxMips], s_env[kMaxViewsLds];
    if (threadIdx.x < kMaxMips) {
        s_mip_off[threadIdx.x] = (int)a.atlas.mip_off[threadIdx.x];
        s_mip_res[threadIdx.x] = a.atlas.mip_res[threadIdx.x];
    }
    if ((int)threadIdx.x < a.n_views) s_env[threadIdx.x] = a.env_of_view[threadIdx.x];
    __syncthreads();
    // only now may a thread without a pixel leave: in the boundary workgroup of a small launch (N % 256 below the table
    // sizes) the threads that fill the upper table entries are exactly the ones past the end
    if (i >= N) return;
    const float inv_hw = 1.0f / (float)a.HW;
    const int hw_shift = (a.HW & (a.HW - 1)) == 0 ? __builtin_ctz(a.HW) : -1;     // 512^2, 1024^2, ...: a shift
    auto env_of = [&](int pix) {
        int view;
        if (hw_shift >= 0) {
            view = pix >> hw_shift;
        } else {
            view = (int)((float)pix * inv_hw);           // pix < 2^24 is exact in fp32; one step of correction covers the rounding
            view -= (view * a.HW > pix) ? 1 : 0;
            view += ((view + 1) * a.HW <= pix) ? 1 : 0;
        }
        return s_env[view];                              // (n_views <= kMaxViewsLds on this path)
    };
    // Everything is addressed as (uniform buffer descriptor) + (32-bit lane offset) [+ scalar offset]: no 64-bit lane
    // arithmetic.  The SoA rows of one tensor share ONE lane offset (4*pixel); the channel is a scalar offset.
    constexpr int kAll = 0x7ffffffc;
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.spec, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_diff = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.diff, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_fg = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.fg_pairs, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_nrm = __builtin_amdgcn_make_buffer_rsrc((void*)a.nrm.p, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_view = __builtin_amdgcn_make_buffer_rsrc((void*)a.view.p, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_feat = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat.p, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_pix = __builtin_amdgcn_make_buffer_rsrc((void*)a.pix_idx, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dcol = __builtin_amdgcn_make_buffer_rsrc((void*)(BWD ? a.dcolor.p : a.nrm.p), 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void*)out.p, 0, kAll, 0x00020000);
    const int out_rs = (int)out.rs, out_cs4 = (int)out.cs * 4;
    const int nc = (int)a.nrm.cs * 4, vc = (int)a.view.cs * 4, fc = (int)a.feat.cs * 4, dcs = BWD ? (int)a.dcolor.cs * 4 : 0;
    auto ldf = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    auto load_soa = [&](unsigned idx, ShadeIn& in) {
        const int vo = (int)(idx * 4u);
        in.n = f3(ldf(r_nrm, vo, 0), ldf(r_nrm, vo, nc), ldf(r_nrm, vo, 2 * nc));
        in.v = f3(ldf(r_view, vo, 0), ldf(r_view, vo, vc), ldf(r_view, vo, 2 * vc));
#pragma unroll
        for (int k = 0; k < 5; ++k) in.f[k] = ldf(r_feat, vo, k * fc);
        if (BWD) in.dc = f3(ldf(r_dcol, vo, 0), ldf(r_dcol, vo, dcs), ldf(r_dcol, vo, 2 * dcs));
        in.pix = (int)__builtin_amdgcn_raw_buffer_load_b32(r_pix, vo, 0, 0);
    };
    auto rows = [](__amdgpu_buffer_rsrc_t r) {
        return [r](unsigned off) { return __builtin_bit_cast(HalfRowBits, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
    };
    auto fg_rows = [r_fg](unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_fg, (int)off, 0, 0)); };
    // Two input register sets, two pixels per trip: the inputs of pixel i + 2*stride are loaded INTO the set whose pixel
    // has just issued its gathers (everything the second stage needs lives in ShadeCtx / ShadeTaps by then), so no set is
    // ever copied -- a copy of a freshly requested register is a wait for the whole stream in front of it.
    // The loop body is branch-free up to the stores: indices past the end are clamped (a redundant load of the last pixel)
    // instead of skipped, because a conditional load makes the compiler's vmcnt bookkeeping assume the load-free path and
    // wait for "all but the newest 7" at the first texel use -- which on the taken path includes the prefetch just issued.
    ShadeIn A, B;
    const unsigned n32 = (unsigned)N, s32 = (unsigned)stride, last = n32 - 1;
    unsigned j = (unsigned)i;
    load_soa(j, A);
    load_soa(min(j + s32, last), B);
    auto step = [&](ShadeIn& cur, unsigned idx) {
        const int env = env_of(cur.pix);
        ShadeCtx c;
        ShadeTaps t;
        shade_issue_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c, t, [](int l) { return s_mip_off[l]; },
                           [](int l) { return s_mip_res[l]; }, rows(r_spec), rows(r_diff), fg_rows);
        const F3 dc = cur.dc;
        __builtin_amdgcn_sched_barrier(0);
        load_soa(min(idx + 2 * s32, last), cur);
        __builtin_amdgcn_sched_barrier(0);
        shade_finish_t<FMT>(a.atlas, a.mat, t, c);
        if (idx < n32) {
            const int vo = DM_MUL24((int)idx, out_rs) * 4;
            body(idx, dc, c, [&](int k, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_out, vo, k * out_cs4, 0); });
        }
    };
    for (; j < n32; j += 2 * s32) {
        step(A, j);
        step(B, j + s32);
    }
#endif
}

template <int FMT, bool DBG>
__global__ __launch_bounds__(256, 3) void k_shade_fwd(ShadeArgs a) {
    shade_pixel_loop<FMT, false>(a, a.color, [&](long long i, F3, const ShadeCtx& c, auto&& st) {
        st(0, sat(c.pre.x));
        st(1, sat(c.pre.y));
        st(2, sat(c.pre.z));
        if (DBG) {
            F3 sl = lin2srgb(c.spec), dl = lin2srgb(c.diff), sc = lin2srgb(c.spec_albedo), dc = lin2srgb(c.albedo);
            a.albedo[3 * i] = c.albedo.x; a.albedo[3 * i + 1] = c.albedo.y; a.albedo[3 * i + 2] = c.albedo.z;
            a.spec_light[3 * i] = sl.x; a.spec_light[3 * i + 1] = sl.y; a.spec_light[3 * i + 2] = sl.z;
            a.diff_light[3 * i] = dl.x; a.diff_light[3 * i + 1] = dl.y; a.diff_light[3 * i + 2] = dl.z;
            a.spec_color[3 * i] = sc.x; a.spec_color[3 * i + 1] = sc.y; a.spec_color[3 * i + 2] = sc.z;
            a.diff_color[3 * i] = dc.x; a.diff_color[3 * i + 1] = dc.y; a.diff_color[3 * i + 2] = dc.z;
            a.metallic[i] = c.metallic;
            a.roughness[i] = c.roughness;
        }
    });
}

template <int FMT>
__global__ __launch_bounds__(256, 3) void k_shade_bwd(ShadeArgs a) {
    shade_pixel_loop<FMT, true>(a, a.dfeat, [&](long long i, F3 dc, const ShadeCtx& c, auto&& st) {
        float df[5];
        shade_backward(a.mat, c, dc, df);
#pragma unroll
        for (int k = 0; k < 5; ++k) st(k, df[k]);
    });
}

static inline int shade_offsets32(const ShadeArgs& a, long long n_max, bool bwd) {
    auto fits = [&](long long rs, long long cs, int ch) {      // SoA rows (unit row stride), every byte offset below 2^31
        return rs == 1 && cs >= 0 && (n_max - 1) + (ch - 1) * cs < 0x7fffffffLL / 4;
    };
    auto out_fits = [&](long long rs, long long cs, int ch) {   // outputs: any small row stride (24-bit multiply), 31-bit byte offsets
        return rs >= 1 && rs < (1 << 20) && cs >= 0 && n_max < (1 << 23) && (n_max - 1) * rs + (ch - 1) * cs < 0x7fffffffLL / 4;
    };
    return fits(a.nrm.rs, a.nrm.cs, 3) && fits(a.view.rs, a.view.cs, 3) && fits(a.feat.rs, a.feat.cs, 5) &&
           (bwd ? fits(a.dcolor.rs, a.dcolor.cs, 3) && out_fits(a.dfeat.rs, a.dfeat.cs, 5) : out_fits(a.color.rs, a.color.cs, 3)) &&
           n_max < 0x3fffffffLL;
}

// persistent launch: enough workgroups to fill every CU at the kernels' occupancy, never more than needed
static inline int shade_blocks(long long n_max, int wg_per_cu) {
    if (const char* e = getenv("DREAMMAT_SHADE_WGPCU")) wg_per_cu = atoi(e);     // development knob
    long long need = (n_max + 255) / 256;
    return (int)std::min<long long>(need, 256 * wg_per_cu);
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
                 const int32_t* n_dev, long long n_max, int HW, int n_views, float* color, long long color_rs,
                 long long color_cs, float* dbg_albedo, float* dbg_spec_light, float* dbg_diff_light,
                 float* dbg_spec_color, float* dbg_diff_color, float* dbg_metallic, float* dbg_roughness,
                 hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !color || n_max <= 0 || HW <= 0 || n_views <= 0)
        return DM_ERR_ARG;
    int ndbg = (dbg_albedo != 0) + (dbg_spec_light != 0) + (dbg_diff_light != 0) + (dbg_spec_color != 0) +
               (dbg_diff_color != 0) + (dbg_metallic != 0) + (dbg_roughness != 0);
    if (ndbg != 0 && ndbg != 7) return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW; a.n_views = n_views;
    a.color = {color, color_rs, color_cs};
    a.albedo = dbg_albedo; a.spec_light = dbg_spec_light; a.diff_light = dbg_diff_light;
    a.spec_color = dbg_spec_color; a.diff_color = dbg_diff_color; a.metallic = dbg_metallic;
    a.roughness = dbg_roughness;
    DM_ENTER();
    const dim3 grid(shade_blocks(n_max, 4));               // forward: <= 128 VGPRs, 4 workgroups per CU
    a.offsets32 = shade_offsets32(a, n_max, false);
    const bool dbg = ndbg != 0;
    switch (a.atlas.texel_format) {
        case kTexelRgb18e8:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelRgb18e8, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelRgb18e8, false>), grid, dim3(256), 0, stream, a);
            break;
        case kTexelF16:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelF16, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelF16, false>), grid, dim3(256), 0, stream, a);
            break;
        default:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelF32, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelF32, false>), grid, dim3(256), 0, stream, a);
            break;
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_shade_bwd(const dm_env_atlas* atlas, const dm_mat_cfg* mat, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, int n_views, const float* dcolor, long long dcolor_rs,
                 long long dcolor_cs, float* dfeat, long long dfeat_rs, long long dfeat_cs, hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !dcolor || !dfeat || n_max <= 0 || HW <= 0 || n_views <= 0)
        return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW; a.n_views = n_views;
    a.dcolor = {dcolor, dcolor_rs, dcolor_cs};
    a.dfeat = {dfeat, dfeat_rs, dfeat_cs};
    DM_ENTER();
    a.offsets32 = shade_offsets32(a, n_max, true);
    const dim3 grid(shade_blocks(n_max, 3));               // backward: <= 168 VGPRs, 3 workgroups per CU
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
