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
    const void* grid;                 // const dm_grid* (host struct, device pointers) or null
};
}

namespace {

using namespace dm::mc;

struct Str { const float* p; long long rs, cs; };
struct StrOut { float* p; long long rs, cs; };

struct McArgs {
    McCfg cfg;
    const DmBvhNode* nodes; const float* tris; const DmBvhNode4* nodes4;
    DmGrid grid; int use_grid; int grid_in_lds; int grid_table_words;
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
    sc.grid = a.use_grid ? &a.grid : nullptr; sc.grid_tb = {a.grid.bits, a.grid.sbase, a.grid.off16, a.grid.dist4};
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

// ---- wave-cooperative occupancy-grid traversal (the one-wave-per-pixel forward kernel).  Same cells, same triangle tests and
// the same boolean per ray as dm_grid_any_hit (csrc/grid_core.h), reorganised for 64 lanes: measured with in-kernel counters
// on the bench scene, the per-ray form ran the cell loop with 21 of 64 lanes active and the triangle loop with 9 -- 163
// triangle-loop trips per 64 rays for 28 tests per ray -- because every lane reaches its occupied cells at its own step and
// the lists have their own lengths.  Here a round has three wave-uniform parts:
//   (1) every live ray walks empty cells (LDS reads + VALU) until it stands on an occupied cell or is gone;
//   (2) the lanes' triangle lists are laid end to end (prefix sum of the lengths) and the (ray, triangle) pairs are dealt out
//       64 at a time -- every lane with a list writes its number over its stretch of an owner table in LDS, pair k reads its
//       owner there and the owner's ray next to it: full lanes whatever the individual list lengths.  (The first version found
//       the owner by a 6-step search over the prefixes: six DEPENDENT LDS round trips per batch, SQ_WAIT_ANY 45 % of the
//       wave-cycles -- profiles/r03_pmc_mc.json);
//   (3) rays that were hit stop, the others step past their cell.
// development counters (tools/build_variant.sh st mc_shade.hip -DDM_MC_STATS; tools/mc_probe.py --stats): wave-level trips
#ifdef DM_MC_STATS
__device__ unsigned long long dm_mc_stats[8];
#define DM_STAT(i, n) do { if (lane == 0) atomicAdd(&dm_mc_stats[i], (unsigned long long)(n)); } while (0)
#else
#define DM_STAT(i, n) do { } while (0)
#endif

constexpr unsigned kOwnerWindow = 512;   // pairs per pass of part (2) (a round of the bench scene has ~200)

struct WaveScratch {
    unsigned start[64];           // exclusive prefix of this round's list lengths
    unsigned e0[64];              // first record of each lane's list
    float ray[64][6];             // origin, direction
    unsigned hit[64];
    unsigned char owner[kOwnerWindow];   // pair k of the current window -> lane that owns it
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the four tables as LDS pointers (32-bit addresses, ds_read): through generic pointers the walk loop read them with flat
// loads and 64-bit address arithmetic, 119 VALU per cell
#define DM_LDS __attribute__((address_space(3)))
struct LdsTables {
    const DM_LDS uint32_t* bits; const DM_LDS uint32_t* sbase; const DM_LDS uint16_t* off16; const DM_LDS uint8_t* dist4;
};

__device__ __forceinline__ bool grid_trace_wave(const DmGrid& g, const LdsTables& tb, bool active, float ox, float oy, float oz,
                                                float dx, float dy, float dz, float t_max, WaveScratch* ws, int lane) {
    DmDda s;
    bool alive = active && dm_dda_init(g, s, ox, oy, oz, dx, dy, dz, t_max);
    bool hit = false;
    ws->ray[lane][0] = ox; ws->ray[lane][1] = oy; ws->ray[lane][2] = oz;
    ws->ray[lane][3] = dx; ws->ray[lane][4] = dy; ws->ray[lane][5] = dz;
    ws->hit[lane] = 0u;
    DM_STAT(0, 1);                                       // wave-rays
    const int bd0 = (g.dim[0] + 1) >> 1, bd1 = (g.dim[1] + 1) >> 1;
    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        DM_STAT(1, 1);                                   // rounds
        int c = 0;
        uint32_t w = 0;
        bool found = false;
        while (__builtin_amdgcn_ballot_w64(alive && !found) != 0ull) {
            DM_STAT(2, 1);                               // cell-walk trips
            DM_STAT(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(alive && !found)));
            if (alive && !found) {
                // both table reads up front (one wait): the block distance decides how to move on from an empty cell
                c = dm_dda_cell(g, s);
                const int bi = ((s.iz >> 1) * bd1 + (s.iy >> 1)) * bd0 + (s.ix >> 1);
                w = tb.bits[c >> 5];
                const int D = (tb.dist4[bi >> 1] >> ((bi & 1) * 4)) & 15;
                if ((w >> (c & 31)) & 1u) found = true;
                else alive = dm_dda_advance_d(g, s, D, ox, oy, oz, dx, dy, dz);
            }
        }
        unsigned e0 = 0, cnt = 0;
        if (found) {
            const int wi = c >> 5;
            const uint32_t r = tb.sbase[wi >> 6] + (uint32_t)tb.off16[wi] + (uint32_t)__builtin_popcount(w & ((1u << (c & 31)) - 1u));
            e0 = g.occ_start[r];
            cnt = g.occ_start[r + 1] - e0;
        }
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const unsigned total = __shfl(incl, 63, 64);
        if (total == 0u) break;                          // nobody found a cell: every ray is gone
        ws->start[lane] = incl - cnt;
        ws->e0[lane] = e0;
        wave_lds_sync();
        DM_STAT(5, total);                               // (ray, triangle) pairs
        const unsigned my0 = incl - cnt, my1 = incl;
        for (unsigned cbase = 0; cbase < total; cbase += kOwnerWindow) {
            const unsigned cend = min(total, cbase + kOwnerWindow);
            for (unsigned k = max(my0, cbase); k < min(my1, cend); ++k) ws->owner[k - cbase] = (unsigned char)lane;
            wave_lds_sync();
            for (unsigned base = cbase; base < cend; base += 64) {
                DM_STAT(4, 1);                           // pair batches
                const unsigned k = base + lane;
                if (k < cend) {
                    const int lo = ws->owner[k - cbase];
                    const unsigned e = ws->e0[lo] + (k - ws->start[lo]);
                    const float* r6 = ws->ray[lo];
                    if (dm_bvh_ray_triangle(g.cell_tris + 12 * (size_t)e, r6[0], r6[1], r6[2], r6[3], r6[4], r6[5], t_max)) ws->hit[lo] = 1u;
                }
            }
            if (cend < total) wave_lds_sync();           // (the next window rewrites the owner table)
        }
        wave_lds_sync();
        if (found) {
            if (ws->hit[lane]) { hit = true; alive = false; }
            else alive = dm_dda_step(g, s);
        }
        wave_lds_sync();                                 // (the next round rewrites start / e0)
    }
    return hit;
}

template <bool BWD, int BLOCK, bool COOP = false>
__global__ __launch_bounds__(BLOCK) void k_mc_shade_wave(McArgs a) {
    using S = typename std::conditional<BWD, Dual, float>::type;
    // occupancy tables of the mesh grid (bits | sbase | off16, contiguous and 16-byte padded in the blob): one LDS copy per
    // workgroup, shared by its 16 waves -- see csrc/grid_core.h
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_grid[];
    DmGridTables gtb = {a.grid.bits, a.grid.sbase, a.grid.off16, a.grid.dist4};
    if (!BWD && a.use_grid && a.grid_in_lds) {
        const int nw = a.grid_table_words;
        for (int w = threadIdx.x; w < nw; w += blockDim.x) lds_grid[w] = a.grid.bits[w];
        __syncthreads();
        gtb.bits = lds_grid;
        gtb.sbase = lds_grid + (a.grid.sbase - a.grid.bits);
        gtb.off16 = reinterpret_cast<const uint16_t*>(lds_grid + (reinterpret_cast<const uint32_t*>(a.grid.off16) - a.grid.bits));
        gtb.dist4 = reinterpret_cast<const uint8_t*>(lds_grid + (reinterpret_cast<const uint32_t*>(a.grid.dist4) - a.grid.bits));
    }
    WaveScratch* ws = nullptr;
    LdsTables ltb = {};
    if (COOP) {                                          // (launched only with the tables in LDS)
        ws = reinterpret_cast<WaveScratch*>(lds_grid + (a.grid_table_words + 3) / 4 * 4) + (threadIdx.x >> 6);
        const DM_LDS uint32_t* lb = (const DM_LDS uint32_t*)lds_grid;
        ltb.bits = lb;
        ltb.sbase = lb + (a.grid.sbase - a.grid.bits);
        ltb.off16 = (const DM_LDS uint16_t*)(lb + (reinterpret_cast<const uint32_t*>(a.grid.off16) - a.grid.bits));
        ltb.dist4 = (const DM_LDS uint8_t*)(lb + (reinterpret_cast<const uint32_t*>(a.grid.dist4) - a.grid.bits));
    }
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
        sc.grid = a.use_grid ? &a.grid : nullptr; sc.grid_tb = gtb;
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
            if (!BWD && COOP) {
                // the occlusion rays of the round, traced by the whole wave together (grid_trace_wave)
                V3<float> d = {0.f, 0.f, 1.f};
                if (active) d = sample_dir<float>(a.cfg, sc, fr, px.alpha, s);
                const float eps = 1e-5f;                   // get_lights (:490-507), as in occluded()
                hit = grid_trace_wave(a.grid, ltb, active, fr.p[0] + d.x * eps, fr.p[1] + d.y * eps, fr.p[2] + d.z * eps, d.x, d.y, d.z,
                                      10.0f, ws, lane);
                if (active) sample_eval<S, false>(a.cfg, sc, fr, al, s, hit, acc);
            } else if (active) {
                sample_eval<S, !BWD>(a.cfg, sc, fr, al, s, hit, acc);
            }
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
        if constexpr (!BWD) {
            if (a.use_grid && a.grid_in_lds) {
                // one workgroup of 16 waves per CU around ONE LDS copy of the occupancy tables + a scratch block per wave
                // (8 waves with twice the registers, no spills: 27.5 vs 18.1 ms on the bench scene -- occupancy wins)
                const size_t lds = (size_t)((a.grid_table_words + 3) / 4 * 4) * 4 + 16 * sizeof(WaveScratch);
                static bool attr_set = false;
                if (!attr_set) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mc_shade_wave<false, 1024, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr_set = true;
                }
                const unsigned grid = (unsigned)std::min<long long>((n_max + 15) / 16, 256);
                hipLaunchKernelGGL((k_mc_shade_wave<false, 1024, true>), dim3(grid), dim3(1024), lds, stream, a);
                return;
            }
        }
        {
            const unsigned grid = (unsigned)std::min<long long>((n_max + 3) / 4, 256 * 8);
            hipLaunchKernelGGL((k_mc_shade_wave<BWD, 256>), dim3(grid), dim3(256), 0, stream, a);
        }
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
    a.use_grid = 0; a.grid_in_lds = 0;
    if (s->grid) {
        a.grid = *(const DmGrid*)s->grid;
        if (!a.grid.bits || !a.grid.sbase || !a.grid.off16 || !a.grid.dist4 || !a.grid.occ_start || !a.grid.cell_tris || a.grid.n_words <= 0)
            return false;
        a.use_grid = 1;
        // the four tables are the first sections of dm_grid_build's blob (contiguous, in this order): copy them to LDS when they fit
        const long long n_blocks = (long long)((a.grid.dim[0] + 1) / 2) * ((a.grid.dim[1] + 1) / 2) * ((a.grid.dim[2] + 1) / 2);
        const long long tw = (reinterpret_cast<const uint32_t*>(a.grid.dist4) - a.grid.bits) + (n_blocks + 7) / 8;
        const bool contiguous = a.grid.sbase > a.grid.bits && reinterpret_cast<const uint32_t*>(a.grid.off16) > a.grid.sbase &&
                                reinterpret_cast<const uint32_t*>(a.grid.dist4) > reinterpret_cast<const uint32_t*>(a.grid.off16) && tw < (1 << 20);
        a.grid_table_words = contiguous ? (int)tw : 0;
        a.grid_in_lds = contiguous && ((size_t)tw + 3) / 4 * 16 + 16 * sizeof(WaveScratch) <= 160 * 1024;
    }
    return true;
}

}  // namespace

extern "C" {

#ifdef DM_MC_STATS
int dm_mc_debug_stats(unsigned long long* out8) {       // development: read and clear the counters
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(dm_mc_stats), sizeof(z)) != hipSuccess) return DM_ERR_UNSUPPORTED;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dm_mc_stats), z, sizeof(z));
    return DM_OK;
}
#endif

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
