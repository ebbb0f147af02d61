// Monte-Carlo ray-traced shading (the reference's default material branch):
// DreamMatMaterial.forward(use_raytracing=True) + shade_raytracing (threestudio/models/materials/dreammat_material.py:
// 615-677, 726-744) with the BVH occlusion queries of raytracing_renderer.py:318-324 fused in.  Per covered pixel:
// nd cosine-weighted + ns GGX directions, one any-hit BVH query and one nearest-texel lat-long lookup per direction,
// BRDF / pdf arithmetic in registers (csrc/mc_shade_core.h).  The reference materialises [N, nd+ns, 3] intermediates
// (> 10 GB at 8 views x 512^2, which is why it only runs at batch 1); here nothing per-sample ever reaches HBM except
// one hit bit, kept so that the backward pass does not have to trace again.
//   forward : shade_pixel<float, trace + record hit bits>
//   backward: shade_pixel<Dual, reuse hit bits> (forward-mode d/d alpha) + analytic albedo / metallic terms
// STATUS: parity-tested against the reference's own outputs and autograd gradients on the CPU (tests/hostemu) and on
// the MI355X (tests/test_mc_gpu.py); one thread per pixel, 0.27 G rays/s -- correct, not yet fast.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "mc_shade_core.h"

extern "C" {
// mirrors of the public structs (include/dreammat_hip.h)
struct dm_mat_cfg { float min_metallic, max_metallic, min_roughness, max_roughness; };
struct dm_mc_scene {
    const void* bvh_nodes; const float* bvh_tris;
    const float* lights; int n_env, light_h, light_w;
    const float* samples_diffuse; const float* samples_specular;
    int n_diffuse, n_specular;
    int geometry_ggx_smith;
    const void* bvh_nodes4;
};
}

namespace {

using namespace dm::mc;

struct Str { const float* p; long long rs, cs; };
struct StrOut { float* p; long long rs, cs; };

struct McArgs {
    McCfg cfg;
    const DmBvhNode* nodes; const float* tris; const DmBvhNode4* nodes4;
    const float* lights; int n_env, light_h, light_w;
    const float* samples_d; const float* samples_s;
    Str pos, nrm, view, feat, dcolor;
    const int* pix_idx; const int* env_of_view; const int* n_dev; int HW;
    const float* rand_d; const float* rand_s;
    unsigned* hit_bits; int hit_words;
    StrOut color, dfeat;
    float *albedo, *spec_light, *diff_light, *spec_color, *diff_color, *metallic, *roughness;   // dense [N,3]/[N,1] or null
};

__device__ __forceinline__ void load3(const Str& s, long long i, float* o) {
    o[0] = s.p[i * s.rs]; o[1] = s.p[i * s.rs + s.cs]; o[2] = s.p[i * s.rs + 2 * s.cs];
}

template <bool BWD>
__global__ __launch_bounds__(128) void k_mc_shade(McArgs a) {
    const long long N = *a.n_dev;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float p[3], n[3], v[3], f[5];
    load3(a.pos, i, p); load3(a.nrm, i, n); load3(a.view, i, v);
#pragma unroll
    for (int k = 0; k < 5; ++k) f[k] = a.feat.p[i * a.feat.rs + k * a.feat.cs];
    const int env = a.env_of_view[a.pix_idx[i] / a.HW];
    McScene sc;
    sc.nodes = a.nodes; sc.tris = a.tris; sc.nodes4 = a.nodes4;
    sc.light = a.lights + (size_t)env * a.light_h * a.light_w * 3; sc.light_h = a.light_h; sc.light_w = a.light_w;
    sc.samples_d = a.samples_d; sc.samples_s = a.samples_s;
    const float rd = a.rand_d ? a.rand_d[i] : -1.f, rs = a.rand_s ? a.rand_s[i] : -1.f;
    // hit bits live in a small per-thread array (<= kMaxSamples / 32 words) and are copied to / from HBM once
    unsigned bits[kMaxSamples / 32];
    unsigned* gb = a.hit_bits + (size_t)i * a.hit_words;
    McPixel px;
    if (!BWD) {
        for (int w = 0; w < a.hit_words; ++w) bits[w] = 0u;
        shade_pixel<float, true>(a.cfg, sc, p, n, v, f, rd, rs, bits, px);
        for (int w = 0; w < a.hit_words; ++w) gb[w] = bits[w];
#pragma unroll
        for (int c = 0; c < 3; ++c) a.color.p[i * a.color.rs + c * a.color.cs] = lin2srgb_mc(px.pre[c]);
        if (a.albedo) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.albedo[3 * i + c] = lin2srgb_mc(px.albedo[c]);
                a.spec_light[3 * i + c] = lin2srgb_mc(px.Ls_mean[c]);
                a.diff_light[3 * i + c] = lin2srgb_mc(px.Ld_mean[c]);
                a.spec_color[3 * i + c] = lin2srgb_mc(px.specular[c]);
                a.diff_color[3 * i + c] = lin2srgb_mc(px.diffuse[c]);
            }
            a.metallic[i] = px.metallic;
            a.roughness[i] = sqrtf(px.alpha + 1e-7f);
        }
    } else {
        for (int w = 0; w < a.hit_words; ++w) bits[w] = gb[w];
        shade_pixel<Dual, false>(a.cfg, sc, p, n, v, f, rd, rs, bits, px);
        float dc[3], df[5];
        load3(a.dcolor, i, dc);
        finish_backward(a.cfg, px, dc, df);
#pragma unroll
        for (int k = 0; k < 5; ++k) a.dfeat.p[i * a.dfeat.rs + k * a.dfeat.cs] = df[k];
    }
}

// ---- one WAVE per pixel: the nd + ns sample directions are spread over the 64 lanes (6 rounds for 200 + 128), each
// lane traces its own occlusion rays and accumulates its share of A / B / Ld / Ls; a butterfly reduction combines the
// lanes, lane 0 finishes the pixel.  All rays of a wave start at the same surface point, so the top of the BVH is
// shared; the hit bits of a round are exactly one __ballot.  Default since round 2 (MI355X, 100 k points of the 50 880-triangle
// bench mesh, 200 + 128 directions: serial 130 ms / 0.25 G rays/s, wave 82 ms, wave + 4-wide BVH 68 ms / 0.48 G rays/s --
// profiles/r02_mc_probe.json); DREAMMAT_MC_KERNEL=serial selects the one-thread-per-pixel kernel.  The decomposition
// (strided samples, bit packing, finish on the combined sums) is checked on the CPU by tests/hostemu.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void wave_sum_inplace(float& v) { v = wave_sum(v); }
__device__ __forceinline__ void wave_sum_inplace(Dual& v) { v.v = wave_sum(v.v); v.d = wave_sum(v.d); }

template <bool BWD>
__global__ __launch_bounds__(256) void k_mc_shade_wave(McArgs a) {
    using S = typename std::conditional<BWD, Dual, float>::type;
    const long long N = *a.n_dev;
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int sn = a.cfg.n_diffuse + a.cfg.n_specular;
    for (long long i = wave0; i < N; i += nwaves) {
        float p[3], n[3], v[3], f[5];
        load3(a.pos, i, p); load3(a.nrm, i, n); load3(a.view, i, v);
#pragma unroll
        for (int k = 0; k < 5; ++k) f[k] = a.feat.p[i * a.feat.rs + k * a.feat.cs];
        const int env = a.env_of_view[a.pix_idx[i] / a.HW];
        McScene sc;
        sc.nodes = a.nodes; sc.tris = a.tris; sc.nodes4 = a.nodes4;
        sc.light = a.lights + (size_t)env * a.light_h * a.light_w * 3; sc.light_h = a.light_h; sc.light_w = a.light_w;
        sc.samples_d = a.samples_d; sc.samples_s = a.samples_s;
        McFrame fr;
        McPixel px;
        pixel_setup(a.cfg, p, n, v, f, a.rand_d ? a.rand_d[i] : -1.f, a.rand_s ? a.rand_s[i] : -1.f, fr, px);
        const S al = seed(S(), px.alpha);
        McAcc<S> acc;
        acc_clear(al, acc);
        unsigned* gb = a.hit_bits + (size_t)i * a.hit_words;
        for (int base = 0; base < sn; base += 64) {
            const int s = base + lane;
            const bool active = s < sn;
            bool hit = false;
            if (BWD && active) hit = (gb[s >> 5] >> (s & 31)) & 1u;
            if (active) sample_eval<S, !BWD>(a.cfg, sc, fr, al, s, hit, acc);
            if (!BWD) {
                const unsigned long long m = __ballot(active && hit);
                if (lane == 0) {
                    gb[base >> 5] = (unsigned)m;
                    if ((base >> 5) + 1 < a.hit_words) gb[(base >> 5) + 1] = (unsigned)(m >> 32);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wave_sum_inplace(acc.A[c]); wave_sum_inplace(acc.B[c]);
            acc.Ld[c] = wave_sum(acc.Ld[c]); acc.Ls[c] = wave_sum(acc.Ls[c]);
        }
        if (lane != 0) continue;
        pixel_finish(a.cfg, acc, px);
        if (!BWD) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.color.p[i * a.color.rs + c * a.color.cs] = lin2srgb_mc(px.pre[c]);
            if (a.albedo) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a.albedo[3 * i + c] = lin2srgb_mc(px.albedo[c]);
                    a.spec_light[3 * i + c] = lin2srgb_mc(px.Ls_mean[c]);
                    a.diff_light[3 * i + c] = lin2srgb_mc(px.Ld_mean[c]);
                    a.spec_color[3 * i + c] = lin2srgb_mc(px.specular[c]);
                    a.diff_color[3 * i + c] = lin2srgb_mc(px.diffuse[c]);
                }
                a.metallic[i] = px.metallic;
                a.roughness[i] = sqrtf(px.alpha + 1e-7f);
            }
        } else {
            float dc[3], df[5];
            load3(a.dcolor, i, dc);
            finish_backward(a.cfg, px, dc, df);
#pragma unroll
            for (int k = 0; k < 5; ++k) a.dfeat.p[i * a.dfeat.rs + k * a.dfeat.cs] = df[k];
        }
    }
}

bool use_wave_kernel() {
    const char* e = getenv("DREAMMAT_MC_KERNEL");      // read per call: tests toggle it
    return !(e && !strcmp(e, "serial"));
}

template <bool BWD>
void launch_mc(const McArgs& a, long long n_max, hipStream_t stream) {
    if (use_wave_kernel()) {
        const unsigned grid = (unsigned)std::min<long long>((n_max + 3) / 4, 256 * 8);
        hipLaunchKernelGGL(k_mc_shade_wave<BWD>, dim3(grid), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_mc_shade<BWD>, dim3(dm_div_up(n_max, 128)), dim3(128), 0, stream, a);
    }
}

bool fill(McArgs& a, const dm_mc_scene* s, const dm_mat_cfg* mat) {
    if (!s || !mat || !s->bvh_nodes || !s->bvh_tris || !s->lights || !s->samples_diffuse || !s->samples_specular) return false;
    if (s->n_env <= 0 || s->light_h <= 0 || s->light_w <= 0 || s->n_diffuse <= 0 || s->n_specular <= 0) return false;
    if (s->n_diffuse + s->n_specular > kMaxSamples) return false;
    a.cfg = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness, s->n_diffuse, s->n_specular,
             s->geometry_ggx_smith ? 1 : 0};
    a.nodes = (const DmBvhNode*)s->bvh_nodes; a.tris = s->bvh_tris; a.nodes4 = (const DmBvhNode4*)s->bvh_nodes4;
    a.lights = s->lights; a.n_env = s->n_env; a.light_h = s->light_h; a.light_w = s->light_w;
    a.samples_d = s->samples_diffuse; a.samples_s = s->samples_specular;
    return true;
}

}  // namespace

extern "C" {

int dm_mc_hit_words(int n_diffuse, int n_specular) { return (n_diffuse + n_specular + 31) / 32; }

int dm_mc_shade_fwd(const dm_mc_scene* scene, const dm_mat_cfg* mat, const float* pos, long long pos_rs, long long pos_cs,
                    const float* nrm, long long nrm_rs, long long nrm_cs, const float* view, long long view_rs,
                    long long view_cs, const float* feat, long long feat_rs, long long feat_cs, const int32_t* pix_idx,
                    const int32_t* env_of_view, const int32_t* n_dev, long long n_max, int HW, const float* rand_diffuse,
                    const float* rand_specular, uint32_t* hit_bits, float* color, long long color_rs, long long color_cs,
                    float* dbg_albedo, float* dbg_spec_light, float* dbg_diff_light, float* dbg_spec_color,
                    float* dbg_diff_color, float* dbg_metallic, float* dbg_roughness, hipStream_t stream) {
    McArgs a = {};
    if (!fill(a, scene, mat) || !pos || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev || !hit_bits || !color ||
        n_max <= 0 || HW <= 0)
        return DM_ERR_ARG;
    const bool any_dbg = dbg_albedo || dbg_spec_light || dbg_diff_light || dbg_spec_color || dbg_diff_color || dbg_metallic || dbg_roughness;
    const bool all_dbg = dbg_albedo && dbg_spec_light && dbg_diff_light && dbg_spec_color && dbg_diff_color && dbg_metallic && dbg_roughness;
    if (any_dbg && !all_dbg) return DM_ERR_ARG;
    a.pos = {pos, pos_rs, pos_cs}; a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW;
    a.rand_d = rand_diffuse; a.rand_s = rand_specular;
    a.hit_bits = hit_bits; a.hit_words = dm_mc_hit_words(a.cfg.n_diffuse, a.cfg.n_specular);
    a.color = {color, color_rs, color_cs};
    a.albedo = dbg_albedo; a.spec_light = dbg_spec_light; a.diff_light = dbg_diff_light; a.spec_color = dbg_spec_color;
    a.diff_color = dbg_diff_color; a.metallic = dbg_metallic; a.roughness = dbg_roughness;
    DM_ENTER();
    launch_mc<false>(a, n_max, stream);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_mc_shade_bwd(const dm_mc_scene* scene, const dm_mat_cfg* mat, const float* pos, long long pos_rs, long long pos_cs,
                    const float* nrm, long long nrm_rs, long long nrm_cs, const float* view, long long view_rs,
                    long long view_cs, const float* feat, long long feat_rs, long long feat_cs, const int32_t* pix_idx,
                    const int32_t* env_of_view, const int32_t* n_dev, long long n_max, int HW, const float* rand_diffuse,
                    const float* rand_specular, const uint32_t* hit_bits, const float* dcolor, long long dcolor_rs,
                    long long dcolor_cs, float* dfeat, long long dfeat_rs, long long dfeat_cs, hipStream_t stream) {
    McArgs a = {};
    if (!fill(a, scene, mat) || !pos || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev || !hit_bits || !dcolor ||
        !dfeat || n_max <= 0 || HW <= 0)
        return DM_ERR_ARG;
    a.pos = {pos, pos_rs, pos_cs}; a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW;
    a.rand_d = rand_diffuse; a.rand_s = rand_specular;
    a.hit_bits = const_cast<unsigned*>(hit_bits); a.hit_words = dm_mc_hit_words(a.cfg.n_diffuse, a.cfg.n_specular);
    a.dcolor = {dcolor, dcolor_rs, dcolor_cs};
    a.dfeat = {dfeat, dfeat_rs, dfeat_cs};
    DM_ENTER();
    launch_mc<true>(a, n_max, stream);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
