// TEST INFRASTRUCTURE: runs the product's per-thread __host__ __device__ arithmetic
// (dreammat_amd/csrc/raster_core.h, shade_core.h, bvh_core.h, mc_shade_core.h) in plain host loops so the CPU-only test suite
// can check it against the oracle without a GPU.  Never loaded by the product.
#include <cstring>
#include <vector>

#include "../../dreammat_amd/csrc/bvh_core.h"
#include "../../dreammat_amd/csrc/grid_core.h"
#include "../../dreammat_amd/csrc/mc_shade_core.h"
#include "../../dreammat_amd/csrc/raster_core.h"
#include "../../dreammat_amd/csrc/shade_core.h"

#pragma clang fp contract(off)
using namespace dm;

extern "C" {

int emu_rasterize(const float* pos, int B, int Nv, const int* tri, int Nf, int H, int W, float* rast) {
    size_t npix = (size_t)H * W;
    std::vector<float> best(npix);
    std::vector<int> bt(npix);
    for (int b = 0; b < B; ++b) {
        const float4* P = (const float4*)pos + (size_t)b * Nv;
        float4* R = (float4*)rast + (size_t)b * npix;
        for (size_t i = 0; i < npix; ++i) { best[i] = 2.0f; bt[i] = 0x7fffffff; R[i] = make_float4(0, 0, 0, 0); }
        // reversed triangle order on purpose: the result must not depend on bin order
        for (int t = Nf - 1; t >= 0; --t) {
            float4 p0 = P[tri[3 * t]], p1 = P[tri[3 * t + 1]], p2 = P[tri[3 * t + 2]];
            TriSetup s;
            if (!tri_setup(p0, p1, p2, H, W, s)) continue;
            EdgeEq e0 = edge_eq(s.x[1], s.y[1], s.x[2], s.y[2], s.sgn);
            EdgeEq e1 = edge_eq(s.x[2], s.y[2], s.x[0], s.y[0], s.sgn);
            EdgeEq e2 = edge_eq(s.x[0], s.y[0], s.x[1], s.y[1], s.sgn);
            for (int py = s.py0; py <= s.py1; ++py)
                for (int px = s.px0; px <= s.px1; ++px) {
                    int cx = (2 * px + 1 - W) * kSubpix, cy = (2 * py + 1 - H) * kSubpix;
                    if (edge_eval(e0, cx, cy) <= 0 || edge_eval(e1, cx, cy) <= 0 || edge_eval(e2, cx, cy) <= 0) continue;
                    float b0, b1, zw;
                    if (!frag_bary(p0, p1, p2, px, py, H, W, b0, b1, zw)) continue;
                    size_t pi = (size_t)py * W + px;
                    if (zw < best[pi] || (zw == best[pi] && t < bt[pi])) {
                        best[pi] = zw; bt[pi] = t;
                        R[pi] = make_float4(clamp01(b0), clamp01(b1), zw, (float)(t + 1));
                    }
                }
        }
    }
    return 0;
}

int emu_aa_plan(const float* pos, int B, int Nv, const int* tri, const int* opp, const float* rast, int H, int W,
                float* plan) {
    for (int b = 0; b < B; ++b) {
        const float4* P = (const float4*)pos + (size_t)b * Nv;
        const float4* R = (const float4*)rast + (size_t)b * H * W;
        float2* A = (float2*)plan + (size_t)b * H * W;
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                size_t pi = (size_t)py * W + px;
                float2 a = make_float2(0.f, 0.f);
                if (px + 1 < W && R[pi].w != R[pi + 1].w) a.x = aa_pair(P, tri, opp, R[pi], R[pi + 1], H, W, px, py, 0);
                if (py + 1 < H && R[pi].w != R[pi + W].w) a.y = aa_pair(P, tri, opp, R[pi], R[pi + W], H, W, px, py, 1);
                A[pi] = a;
            }
    }
    return 0;
}

struct emu_atlas {
    const float* spec; const float* diff; const float* fg_lut;
    long long spec_env_stride, diff_env_stride;
    long long mip_off[8];
    int mip_res[8];
    int n_mips, diff_res, lut_res;
    float min_rough_mip, max_rough_mip;
    int texel_format;
    const float* fg_pairs;
};

static EnvAtlas conv(const emu_atlas* in) {
    EnvAtlas A;
    A.spec = (const float4*)in->spec; A.diff = (const float4*)in->diff; A.fg_lut = (const float2*)in->fg_lut;
    A.spec_env_stride = in->spec_env_stride; A.diff_env_stride = in->diff_env_stride;
    for (int i = 0; i < 8; ++i) { A.mip_off[i] = in->mip_off[i]; A.mip_res[i] = in->mip_res[i]; }
    A.n_mips = in->n_mips; A.diff_res = in->diff_res; A.lut_res = in->lut_res;
    A.min_rough_mip = in->min_rough_mip; A.max_rough_mip = in->max_rough_mip;
    A.texel_format = in->texel_format;
    A.fg_pairs = (const float4*)in->fg_pairs;
    return A;
}

// nrm/view [N,3], feat [N,5], env [N]; outputs color [N,3], dbg [N,17] = albedo3 spec_light3 diff_light3
// spec_color3 diff_color3 metallic roughness; if dcolor != null also dfeat [N,5].
int emu_shade(const emu_atlas* atlas, const float* matcfg, const float* nrm, const float* view, const float* feat,
              const int* env, long long N, float* color, float* dbg, const float* dcolor, float* dfeat) {
    EnvAtlas A = conv(atlas);
    MatCfg M = {matcfg[0], matcfg[1], matcfg[2], matcfg[3]};
    for (long long i = 0; i < N; ++i) {
        ShadeCtx c;
        shade_eval(A, M, env[i], f3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]),
                   f3(view[3 * i], view[3 * i + 1], view[3 * i + 2]), feat + 5 * i, c);
        color[3 * i] = sat(c.pre.x); color[3 * i + 1] = sat(c.pre.y); color[3 * i + 2] = sat(c.pre.z);
        if (dbg) {
            float* d = dbg + 17 * i;
            F3 sl = lin2srgb(c.spec), dl = lin2srgb(c.diff), sc = lin2srgb(c.spec_albedo), dc = lin2srgb(c.albedo);
            d[0] = c.albedo.x; d[1] = c.albedo.y; d[2] = c.albedo.z;
            d[3] = sl.x; d[4] = sl.y; d[5] = sl.z; d[6] = dl.x; d[7] = dl.y; d[8] = dl.z;
            d[9] = sc.x; d[10] = sc.y; d[11] = sc.z; d[12] = dc.x; d[13] = dc.y; d[14] = dc.z;
            d[15] = c.metallic; d[16] = c.roughness;
        }
        if (dcolor) shade_backward(M, c, f3(dcolor[3 * i], dcolor[3 * i + 1], dcolor[3 * i + 2]), dfeat + 5 * i);
    }
    return 0;
}

// any-hit BVH traversal (bvh_core.h) over host copies of dm_bvh_build's outputs
int emu_bvh_any_hit(const void* nodes, const float* tris, const float* org, const float* dir, long long n, float t_max,
                    unsigned char* hit) {
    for (long long i = 0; i < n; ++i)
        hit[i] = dm_bvh_any_hit((const DmBvhNode*)nodes, tris, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i],
                                dir[3 * i + 1], dir[3 * i + 2], t_max) ? 1 : 0;
    return 0;
}

// Monte-Carlo shading (mc_shade_core.h).  lanes == 1: the serial kernel's path -- forward = shade_pixel<float, true>
// (traces, records the hit bits), backward = shade_pixel<Dual, false> on the recorded bits + finish_backward.
// lanes == 64: the wave kernel's decomposition (k_mc_shade_wave) replayed lane by lane -- sample s of a round goes to
// lane s % 64, each lane keeps its own partial sums, the hit bits of a round are packed like __ballot, the partial
// sums are combined before pixel_finish.  out [N,25] = color3 albedo3 roughness1 metalness1 specular_lights3
// diffuse_lights3 specular_colors3 diffuse_colors3 pre3 (linear colour) alpha1 pad1.
}  // extern "C"

template <class S, bool TRACE>
static void emu_pixel_lanes(const dm::mc::McCfg& cfg, const dm::mc::McScene& sc, const float* p, const float* n, const float* v,
                            const float* f, float rd, float rs, unsigned* hb, int words, int lanes, dm::mc::McPixel& px) {
    using namespace dm::mc;
    McFrame fr;
    pixel_setup(cfg, p, n, v, f, rd, rs, fr, px);
    const S al = seed(S(), px.alpha);
    std::vector<McAcc<S>> part(lanes);
    for (auto& a : part) acc_clear(al, a);
    const int sn = cfg.n_diffuse + cfg.n_specular;
    for (int base = 0; base < sn; base += lanes) {
        unsigned long long ballot = 0;
        for (int lane = 0; lane < lanes; ++lane) {
            const int s = base + lane;
            if (s >= sn) continue;
            bool hit = TRACE ? false : ((hb[s >> 5] >> (s & 31)) & 1u);
            sample_eval<S, TRACE>(cfg, sc, fr, al, s, hit, part[lane]);
            if (TRACE && hit) ballot |= 1ull << lane;
        }
        if (TRACE) {
            hb[base >> 5] = (unsigned)ballot;
            if ((base >> 5) + 1 < words) hb[(base >> 5) + 1] = (unsigned)(ballot >> 32);
        }
    }
    McAcc<S> acc;
    acc_clear(al, acc);
    for (auto& a : part)
        for (int c = 0; c < 3; ++c) {
            acc.A[c] = acc.A[c] + a.A[c]; acc.B[c] = acc.B[c] + a.B[c];
            acc.Ld[c] += a.Ld[c]; acc.Ls[c] += a.Ls[c];
        }
    pixel_finish(cfg, acc, px);
}

extern "C" {

int emu_mc_shade(const float* cfg4, int nd, int ns, int ggx_smith, const void* nodes, const void* nodes4, const float* tris, const float* light,
                 int lh, int lw, const float* samples_d, const float* samples_s, long long N, const float* p, const float* n,
                 const float* v, const float* feat, const float* rand_d, const float* rand_s, unsigned* hit_bits, float* out,
                 const float* dcolor, float* dfeat, int lanes) {
    using namespace dm::mc;
    McCfg cfg = {cfg4[0], cfg4[1], cfg4[2], cfg4[3], nd, ns, ggx_smith};
    McScene sc = {(const DmBvhNode*)nodes, tris, (const DmBvhNode4*)nodes4, nullptr, {nullptr, nullptr, nullptr, nullptr}, light, lh, lw, samples_d, samples_s};
    const int words = kMaxSamples / 32;
    if (nd + ns > kMaxSamples || (lanes != 1 && lanes != 64)) return -1;
    const int used_words = (nd + ns + 31) / 32;
    for (long long i = 0; i < N; ++i) {
        unsigned* hb = hit_bits + (size_t)i * words;
        McPixel px;
        const float rd = rand_d ? rand_d[i] : -1.f, rs = rand_s ? rand_s[i] : -1.f;
        if (!dcolor) {
            std::memset(hb, 0, words * sizeof(unsigned));
            if (lanes == 1) shade_pixel<float, true>(cfg, sc, p + 3 * i, n + 3 * i, v + 3 * i, feat + 5 * i, rd, rs, hb, px);
            else emu_pixel_lanes<float, true>(cfg, sc, p + 3 * i, n + 3 * i, v + 3 * i, feat + 5 * i, rd, rs, hb, used_words, lanes, px);
            float* o = out + 25 * i;
            for (int c = 0; c < 3; ++c) {
                o[c] = lin2srgb_mc(px.pre[c]); o[3 + c] = lin2srgb_mc(px.albedo[c]);
                o[8 + c] = lin2srgb_mc(px.Ls_mean[c]); o[11 + c] = lin2srgb_mc(px.Ld_mean[c]);
                o[14 + c] = lin2srgb_mc(px.specular[c]); o[17 + c] = lin2srgb_mc(px.diffuse[c]);
                o[20 + c] = px.pre[c];
            }
            o[6] = sqrtf(px.alpha + 1e-7f); o[7] = px.metallic; o[23] = px.alpha; o[24] = 0.f;
        } else {
            if (lanes == 1) shade_pixel<Dual, false>(cfg, sc, p + 3 * i, n + 3 * i, v + 3 * i, feat + 5 * i, rd, rs, hb, px);
            else emu_pixel_lanes<Dual, false>(cfg, sc, p + 3 * i, n + 3 * i, v + 3 * i, feat + 5 * i, rd, rs, hb, used_words, lanes, px);
            finish_backward(cfg, px, dcolor + 3 * i, dfeat + 5 * i);
        }
    }
    return 0;
}

// traversal statistics of the any-hit query (same control flow as dm_bvh_any_hit): nodes popped and triangles tested per
// ray -- a CPU-measurable proxy for the GPU traversal cost, used to tune the builder (tools/bvh_stats.py)
int emu_bvh_stats(const void* nodes_v, const float* tris, const float* org, const float* dir, long long n, float t_max,
                  long long* nodes_visited, long long* tris_tested, long long* hits) {
    const DmBvhNode* nodes = (const DmBvhNode*)nodes_v;
    long long nv = 0, tt = 0, nh = 0;
    for (long long i = 0; i < n; ++i) {
        const float ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2], dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
        const float big = 3.0e38f;
        const float ix = fabsf(dx) > 1e-30f ? 1.0f / dx : (dx < 0.f ? -big : big);
        const float iy = fabsf(dy) > 1e-30f ? 1.0f / dy : (dy < 0.f ? -big : big);
        const float iz = fabsf(dz) > 1e-30f ? 1.0f / dz : (dz < 0.f ? -big : big);
        int stack[64], sp = 0;
        stack[sp++] = 0;
        bool hit = false;
        while (sp > 0 && !hit) {
            const DmBvhNode nd = nodes[stack[--sp]];
            ++nv;
            float t0 = (nd.bmin[0] - ox) * ix, t1 = (nd.bmax[0] - ox) * ix;
            float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
            t0 = (nd.bmin[1] - oy) * iy; t1 = (nd.bmax[1] - oy) * iy;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            t0 = (nd.bmin[2] - oz) * iz; t1 = (nd.bmax[2] - oz) * iz;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            if (!(tf >= fmaxf(tn, 0.f)) || tn > t_max) continue;
            if (nd.b > 0) {
                for (int k = 0; k < nd.b && !hit; ++k) { ++tt; hit = dm_bvh_ray_triangle(tris + 12 * (size_t)(nd.a + k), ox, oy, oz, dx, dy, dz, t_max); }
            } else if (sp + 2 <= 64) { stack[sp++] = nd.a; stack[sp++] = nd.a + 1; }
        }
        nh += hit;
    }
    *nodes_visited = nv; *tris_tested = tt; *hits = nh;
    return 0;
}

int emu_bvh4_any_hit(const void* nodes4, const float* tris, const float* org, const float* dir, long long n, float t_max,
                     unsigned char* hit) {
    for (long long i = 0; i < n; ++i)
        hit[i] = dm_bvh4_any_hit((const DmBvhNode4*)nodes4, tris, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i],
                                 dir[3 * i + 1], dir[3 * i + 2], t_max) ? 1 : 0;
    return 0;
}

// the occupancy-grid traversal (grid_core.h): grid = a dm_grid whose pointers address host memory.  stats (optional, [n][3]):
// walk moves (steps + leaps) / occupied cells / triangles tested per ray, from a replay with dm_grid_any_hit's control flow
int emu_grid_any_hit(const void* grid_v, const float* org, const float* dir, long long n, float t_max, unsigned char* hit, int* stats) {
    const DmGrid g = *(const DmGrid*)grid_v;
    const DmGridTables tb = {g.bits, g.sbase, g.off16, g.dist4};
    for (long long i = 0; i < n; ++i)
        hit[i] = dm_grid_any_hit(g, tb, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], t_max) ? 1 : 0;
    if (!stats) return 0;
    for (long long i = 0; i < n; ++i) {
        int moves = 0, occ = 0, tests = 0;
        const float ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2], dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
        DmDda s;
        bool alive = dm_dda_init(g, s, ox, oy, oz, dx, dy, dz, t_max);
        while (alive) {
            ++moves;
            const int c = dm_dda_cell(g, s);
            const uint32_t w = tb.bits[c >> 5];
            if ((w >> (c & 31)) & 1u) {
                ++occ;
                const uint32_t r = dm_grid_rank(tb, c, w);
                bool h = false;
                for (uint32_t e = g.occ_start[r]; e < g.occ_start[r + 1] && !h; ++e) {
                    ++tests;
                    h = dm_bvh_ray_triangle(g.cell_tris + 12 * (size_t)e, ox, oy, oz, dx, dy, dz, t_max);
                }
                if (h) break;
                alive = dm_dda_step(g, s);
            } else {
                alive = dm_dda_advance(g, tb, s, ox, oy, oz, dx, dy, dz);
            }
        }
        stats[3 * i] = moves; stats[3 * i + 1] = occ; stats[3 * i + 2] = tests;
    }
    return 0;
}

// node fetches / triangle tests per ray of the 4-wide traversal (same control flow as dm_bvh4_any_hit)
int emu_bvh4_stats(const void* nodes_v, const float* tris, const float* org, const float* dir, long long n, float t_max,
                   long long* nodes_visited, long long* tris_tested, long long* hits) {
    const DmBvhNode4* nodes = (const DmBvhNode4*)nodes_v;
    long long nv = 0, tt = 0, nh = 0;
    for (long long i = 0; i < n; ++i) {
        const float ox = org[3 * i], oy = org[3 * i + 1], oz = org[3 * i + 2], dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
        const float big = 3.0e38f;
        const float ix = fabsf(dx) > 1e-30f ? 1.0f / dx : (dx < 0.f ? -big : big);
        const float iy = fabsf(dy) > 1e-30f ? 1.0f / dy : (dy < 0.f ? -big : big);
        const float iz = fabsf(dz) > 1e-30f ? 1.0f / dz : (dz < 0.f ? -big : big);
        int stack[48], sp = 0;
        stack[sp++] = 0;
        bool hit = false;
        while (sp > 0 && !hit) {
            const DmBvhNode4 nd = nodes[stack[--sp]];
            ++nv;
            for (int k = 0; k < 4 && !hit; ++k) {
                if (nd.b[k] < 0) continue;
                float t0 = (nd.lo[0][k] - ox) * ix, t1 = (nd.hi[0][k] - ox) * ix;
                float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                t0 = (nd.lo[1][k] - oy) * iy; t1 = (nd.hi[1][k] - oy) * iy;
                tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                t0 = (nd.lo[2][k] - oz) * iz; t1 = (nd.hi[2][k] - oz) * iz;
                tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                if (!(tf >= fmaxf(tn, 0.f)) || tn > t_max) continue;
                if (nd.b[k] > 0) {
                    for (int j = 0; j < nd.b[k] && !hit; ++j) { ++tt; hit = dm_bvh_ray_triangle(tris + 12 * (size_t)(nd.a[k] + j), ox, oy, oz, dx, dy, dz, t_max); }
                } else if (sp < 48) stack[sp++] = nd.a[k];
            }
        }
        nh += hit;
    }
    *nodes_visited = nv; *tris_tested = tt; *hits = nh;
    return 0;
}

}  // extern "C"
