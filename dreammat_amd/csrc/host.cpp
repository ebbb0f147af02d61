// Host-side entry points of libdreammat_hip.so (no device code): mesh topology for the antialias
// kernels (nvdiffrast builds this edge->opposite-vertex hash inside dr.antialias on every call;
// the DreamMat mesh is fixed, so it is built once per mesh) and library introspection.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "bvh_core.h"
#include "dm_common.h"
#include "grid_core.h"

namespace {

struct Aabb {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Aabb& o) { grow(o.lo); grow(o.hi); }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0.f) return 0.f;
        return 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct BvhBuilder {
    const float* v; const int32_t* tri; int n;
    std::vector<Aabb> box;            // per triangle
    std::vector<float> cen;           // per triangle centroid [n][3]
    std::vector<int32_t> order;       // triangle ids, partitioned in place
    std::vector<DmBvhNode> nodes;
    int kLeaf = 4;                    // max triangles per leaf (bvh_core.h assumes nothing about it)
    static constexpr int kBins = 16;

    void set_box(int node, const Aabb& b) {
        for (int k = 0; k < 3; ++k) { nodes[node].bmin[k] = b.lo[k]; nodes[node].bmax[k] = b.hi[k]; }
    }
    // binned surface-area heuristic; falls back to a median split when no bin boundary separates the centroids
    void build(int node, int first, int count) {
        Aabb bb, cb;
        for (int i = first; i < first + count; ++i) { bb.grow(box[order[i]]); cb.grow(&cen[3 * (size_t)order[i]]); }
        set_box(node, bb);
        if (count <= kLeaf) { nodes[node].a = first; nodes[node].b = count; return; }
        int axis = 0;
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        int mid = first + count / 2;
        if (ext[axis] > 0.f) {
            Aabb bin_box[kBins]; int bin_cnt[kBins] = {0};
            const float scale = kBins / ext[axis];
            auto bin_of = [&](int t) { return std::min(kBins - 1, (int)((cen[3 * (size_t)t + axis] - cb.lo[axis]) * scale)); };
            for (int i = first; i < first + count; ++i) { int b = bin_of(order[i]); bin_box[b].grow(box[order[i]]); ++bin_cnt[b]; }
            float right_area[kBins]; Aabb acc; int best = -1; float best_cost = FLT_MAX;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bin_box[b]); right_area[b] = acc.area(); }
            Aabb left; int nl = 0;
            for (int b = 0; b + 1 < kBins; ++b) {
                left.grow(bin_box[b]); nl += bin_cnt[b];
                if (nl == 0 || nl == count) continue;
                float cost = left.area() * nl + right_area[b + 1] * (count - nl);
                if (cost < best_cost) { best_cost = cost; best = b; }
            }
            if (best >= 0) {
                auto it = std::partition(order.begin() + first, order.begin() + first + count,
                                         [&](int32_t t) { return bin_of(t) <= best; });
                mid = (int)(it - order.begin());
            }
        }
        if (mid == first || mid == first + count || ext[axis] <= 0.f) {        // degenerate: split the list in half
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int32_t x, int32_t y) { return cen[3 * (size_t)x + axis] < cen[3 * (size_t)y + axis]; });
        }
        const int left_node = (int)nodes.size();
        nodes.emplace_back(); nodes.emplace_back();
        nodes[node].a = left_node; nodes[node].b = 0;
        build(left_node, first, mid - first);
        build(left_node + 1, mid, first + count - mid);
    }
};

}  // namespace

extern "C" {

int dm_abi_version(void) { return 14; }    // 4: + dm_grid_build / dm_grid_any_hit_rays / dm_host_free, dm_mc_scene.grid; 5: + dm_attention_fwd_lse_bf16 / dm_attention_bwd_bf16; 6: + dm_conv3x3_wgrad_*; 7: + dm_groupnorm_nhwc_bwd_affine; 8: dm_attention_fwd_lse_bf16 takes V untransposed; 9: + dm_gbuffer_compact_tiled; 10: + dm_softmax_rows_bf16 / _bwd, dm_groupnorm_nhwc_bwd_res, dm_conv3x3_small_res_nhwc_bf16, dm_conv2x2_nhwc_bf16; 11: + the _f16 instantiations of the net kernels (dm_elem.h), dm_attention_fwd_fp8; 12: + dm_conv2x2_subpixel_nhwc_bf16 / _f16; 13: + dm_groupnorm_nhwc_stats, dm_conv3x3_gn_ok, dm_conv3x3_gn_nhwc_bf16_fused / _f16 (GroupNorm apply folded into the halo-patch convolution); 14: + dm_transpose_bf16 / _f16, dm_gemm_bf16_batched / _f16, dm_softmax_rows_* up to 16384 columns (the VAE mid-block attention on dm_gemm_*_fused)

// opp[t][i] = vertex opposite to edge i of triangle t in the other triangle sharing that edge,
// -1 if none.  Edge 0 = (v1,v2), edge 1 = (v2,v0), edge 2 = (v0,v1).  Host pointers.
int dm_mesh_build_topology(const int32_t* tri, int32_t n_tri, int32_t* opp) {
    if (!tri || !opp || n_tri <= 0) return DM_ERR_ARG;
    struct Rec { int32_t a, b, t, slot; };
    std::vector<Rec> e((size_t)n_tri * 3);
    for (int32_t t = 0; t < n_tri; ++t)
        for (int i = 0; i < 3; ++i) {
            int32_t va = tri[3 * t + (i + 1) % 3], vb = tri[3 * t + (i + 2) % 3];
            e[(size_t)3 * t + i] = {std::min(va, vb), std::max(va, vb), t, i};
            opp[(size_t)3 * t + i] = -1;
        }
    std::sort(e.begin(), e.end(), [](const Rec& x, const Rec& y) {
        if (x.a != y.a) return x.a < y.a;
        if (x.b != y.b) return x.b < y.b;
        if (x.t != y.t) return x.t < y.t;
        return x.slot < y.slot;
    });
    for (size_t i = 0; i < e.size();) {
        size_t j = i + 1;
        while (j < e.size() && e[j].a == e[i].a && e[j].b == e[i].b) ++j;
        if (j - i >= 2) {
            int32_t o0 = tri[3 * e[i].t + e[i].slot], o1 = tri[3 * e[i + 1].t + e[i + 1].slot];
            opp[3 * e[i].t + e[i].slot] = o1;
            opp[3 * e[i + 1].t + e[i + 1].slot] = o0;
            for (size_t k = i + 2; k < j; ++k) {
                int32_t self = tri[3 * e[k].t + e[k].slot];
                opp[3 * e[k].t + e[k].slot] = (o0 != self) ? o0 : o1;
            }
        }
        i = j;
    }
    return DM_OK;
}

// Bounding-volume hierarchy over the (fixed) DreamMat mesh for the Monte-Carlo shading branch: the host-side
// counterpart of `_raytracing.create_raytracer(vertices, triangles)` (raytracing_renderer.py:31).  Host pointers.
//   nodes_out [2*n_tri] DmBvhNode (32 B each), tris_out [n_tri*12] floats in leaf order, order_out [n_tri] (or NULL) =
//   original triangle id of each leaf slot, *n_nodes_out = nodes used.  Layout: csrc/bvh_core.h.
int dm_bvh_build(const float* v_pos, int32_t n_vert, const int32_t* tri, int32_t n_tri, void* nodes_out, float* tris_out,
                 int32_t* order_out, int32_t* n_nodes_out) {
    if (!v_pos || !tri || !nodes_out || !tris_out || !n_nodes_out || n_vert <= 0 || n_tri <= 0) return DM_ERR_ARG;
    for (size_t i = 0; i < (size_t)n_tri * 3; ++i)
        if (tri[i] < 0 || tri[i] >= n_vert) return DM_ERR_ARG;
    BvhBuilder b;
    b.v = v_pos; b.tri = tri; b.n = n_tri;
    if (const char* e = getenv("DREAMMAT_BVH_LEAF")) b.kLeaf = std::max(1, std::min(16, atoi(e)));   // tuning knob (tools/bvh_stats.py)
    b.box.resize(n_tri); b.cen.resize((size_t)n_tri * 3); b.order.resize(n_tri);
    for (int32_t t = 0; t < n_tri; ++t) {
        b.order[t] = t;
        float c[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) {
            const float* p = v_pos + 3 * (size_t)tri[3 * (size_t)t + k];
            b.box[t].grow(p);
            for (int d = 0; d < 3; ++d) c[d] += p[d] * (1.0f / 3.0f);
        }
        for (int d = 0; d < 3; ++d) b.cen[3 * (size_t)t + d] = c[d];
    }
    b.nodes.reserve((size_t)2 * n_tri);
    b.nodes.emplace_back();
    b.build(0, 0, n_tri);
    if (b.nodes.size() > (size_t)2 * n_tri) return DM_ERR_WORKSPACE;
    std::memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(DmBvhNode));
    for (int32_t j = 0; j < n_tri; ++j) {
        const int32_t t = b.order[j];
        const float* p0 = v_pos + 3 * (size_t)tri[3 * (size_t)t];
        const float* p1 = v_pos + 3 * (size_t)tri[3 * (size_t)t + 1];
        const float* p2 = v_pos + 3 * (size_t)tri[3 * (size_t)t + 2];
        float* o = tris_out + 12 * (size_t)j;
        for (int d = 0; d < 3; ++d) { o[d] = p0[d]; o[4 + d] = p1[d] - p0[d]; o[8 + d] = p2[d] - p0[d]; }
        o[3] = o[7] = o[11] = 0.f;
        if (order_out) order_out[j] = t;
    }
    *n_nodes_out = (int32_t)b.nodes.size();
    return DM_OK;
}

// 4-wide collapse of dm_bvh_build's binary tree (csrc/bvh_core.h, DmBvhNode4): nodes2 = the n_nodes2 binary nodes,
// nodes4_out must hold n_nodes2 entries (upper bound), *n_nodes4_out = entries used.  Triangle order is unchanged.
int dm_bvh_collapse4(const void* nodes2_v, int32_t n_nodes2, void* nodes4_out, int32_t* n_nodes4_out) {
    if (!nodes2_v || !nodes4_out || !n_nodes4_out || n_nodes2 <= 0) return DM_ERR_ARG;
    const DmBvhNode* n2 = (const DmBvhNode*)nodes2_v;
    std::vector<DmBvhNode4> out;
    out.reserve(n_nodes2);
    // work list of (binary node -> wide node) pairs; children of a wide node = grandchildren of the binary node where
    // possible (a binary child that is a leaf stays one slot)
    struct Item { int bin, wide; };
    std::vector<Item> todo;
    out.emplace_back();
    todo.push_back({0, 0});
    auto set_child = [&](DmBvhNode4& w, int k, const DmBvhNode& c, int a, int b) {
        for (int d = 0; d < 3; ++d) { w.lo[d][k] = c.bmin[d]; w.hi[d][k] = c.bmax[d]; }
        w.a[k] = a; w.b[k] = b;
    };
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        DmBvhNode4 w;
        for (int k = 0; k < 4; ++k) { w.a[k] = 0; w.b[k] = -1; for (int d = 0; d < 3; ++d) { w.lo[d][k] = 1.f; w.hi[d][k] = -1.f; } }
        const DmBvhNode& root = n2[it.bin];
        int slots[4], ns = 0;
        if (root.b > 0) {
            slots[ns++] = it.bin;                                   // a leaf root (tiny mesh): one slot
        } else {
            for (int c = 0; c < 2; ++c) {
                const int ci = root.a + c;
                if (n2[ci].b > 0) slots[ns++] = ci;                 // leaf child
                else { slots[ns++] = n2[ci].a; slots[ns++] = n2[ci].a + 1; }   // its two children
            }
        }
        for (int k = 0; k < ns; ++k) {
            const DmBvhNode& c = n2[slots[k]];
            if (c.b > 0) set_child(w, k, c, c.a, c.b);
            else {
                const int wi = (int)out.size();
                out.emplace_back();
                set_child(w, k, c, wi, 0);
                todo.push_back({slots[k], wi});
            }
        }
        out[it.wide] = w;
    }
    if ((int32_t)out.size() > n_nodes2) return DM_ERR_WORKSPACE;
    std::memcpy(nodes4_out, out.data(), out.size() * sizeof(DmBvhNode4));
    *n_nodes4_out = (int32_t)out.size();
    return DM_OK;
}

}  // extern "C"

// ---- uniform occupancy grid of the same triangles (csrc/grid_core.h) ---------------------------------------------------
namespace {

// Akenine-Moeller triangle / box overlap (separating axes: 3 box normals, the triangle normal, 9 edge cross products),
// box centred at c with half size h (already inflated by the caller); v* are absolute positions.
bool tri_box_overlap(const float c[3], float h, const float v0a[3], const float v1a[3], const float v2a[3]) {
    double v0[3], v1[3], v2[3];
    for (int d = 0; d < 3; ++d) { v0[d] = (double)v0a[d] - c[d]; v1[d] = (double)v1a[d] - c[d]; v2[d] = (double)v2a[d] - c[d]; }
    const double e[3][3] = {{v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]},
                            {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]},
                            {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]}};
    for (int d = 0; d < 3; ++d) {                                   // box normals
        const double lo = std::min(v0[d], std::min(v1[d], v2[d])), hi = std::max(v0[d], std::max(v1[d], v2[d]));
        if (lo > h || hi < -h) return false;
    }
    for (int i = 0; i < 3; ++i)                                     // edge i x axis a
        for (int a = 0; a < 3; ++a) {
            double ax[3] = {0, 0, 0};
            const int b = (a + 1) % 3, cc = (a + 2) % 3;            // axis = unit_a x e[i]
            ax[b] = -e[i][cc]; ax[cc] = e[i][b];
            const double p0 = ax[0] * v0[0] + ax[1] * v0[1] + ax[2] * v0[2];
            const double p1 = ax[0] * v1[0] + ax[1] * v1[1] + ax[2] * v1[2];
            const double p2 = ax[0] * v2[0] + ax[1] * v2[1] + ax[2] * v2[2];
            const double r = h * (std::fabs(ax[0]) + std::fabs(ax[1]) + std::fabs(ax[2]));
            if (std::min(p0, std::min(p1, p2)) > r || std::max(p0, std::max(p1, p2)) < -r) return false;
        }
    const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                         e[0][0] * e[1][1] - e[0][1] * e[1][0]};   // triangle plane
    const double dist = n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2];
    const double r = h * (std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2]));
    return std::fabs(dist) <= r;
}

}  // namespace

extern "C" {

// Conservative voxelisation of dm_bvh_build's triangle array (tris12 = its `tris_out`, [n_tri][12]) into a uniform grid of at
// most `res` cells along the longest axis of the mesh box (res <= 0: chosen from the triangle count, <= 96 so that the
// occupancy tables fit LDS).  HOST function.  *blob_out (malloc'd, release with dm_host_free) holds the five sections of
// csrc/grid_core.h, each padded to 16 bytes:
//   bits[n_words] u32 | sbase[ceil(n_words / 64)] u32 | off16[n_words] u16 | dist4[one nibble per 2x2x2 block] |
//   occ_start[n_occ + 1] u32 | cell_tris[n_entries][12] f32
// and `grid` receives the scalars (its pointers are left NULL: the caller copies the blob to the device and fills them).
int dm_grid_build(const float* tris12, int32_t n_tri, int32_t res, void* grid_v, uint32_t** blob_out, int64_t* blob_words) {
    if (!tris12 || !grid_v || !blob_out || !blob_words || n_tri <= 0) return DM_ERR_ARG;
    DmGrid& g = *(DmGrid*)grid_v;
    if (res <= 0) {
        // a closed surface of n triangles occupies ~6 res^2 cells: ~2 triangles per occupied cell (the bench mesh, 50 880
        // triangles: res 64 traced fastest of 48 / 56 / 64 / 76 / 84, profiles/r03_mc_probe.json)
        res = std::min(76, (int)std::lround(std::sqrt((double)n_tri / 12.5)));
        if (const char* e = getenv("DREAMMAT_GRID_RES")) res = atoi(e);          // tuning knob (tools/mc_probe.py)
    }
    res = std::max(4, std::min(96, res));                           // (<= 76: tables + per-wave scratch of the shading kernel fit LDS)
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    auto vert = [&](int32_t t, int k, float* p) {
        const float* s = tris12 + 12 * (size_t)t;
        for (int d = 0; d < 3; ++d) p[d] = k == 0 ? s[d] : s[d] + s[4 * k + d];
    };
    for (int32_t t = 0; t < n_tri; ++t)
        for (int k = 0; k < 3; ++k) {
            float p[3];
            vert(t, k, p);
            for (int d = 0; d < 3; ++d) {
                if (!std::isfinite(p[d])) return DM_ERR_ARG;
                lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[d]);
            }
        }
    float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0.f)) ext = 1.f;
    const float cell = ext * 1.02f / (float)res;                    // 1 % margin on either side of the longest axis
    for (int d = 0; d < 3; ++d) {
        const float mid = 0.5f * (lo[d] + hi[d]);
        int n = (int)std::ceil((hi[d] - lo[d]) / cell + 0.04f);
        n = std::max(1, std::min(res, n));
        g.dim[d] = n;
        g.gmin[d] = mid - 0.5f * (float)n * cell;
    }
    g.cell = cell; g.inv_cell = 1.0f / cell;
    const long long n_cells = (long long)g.dim[0] * g.dim[1] * g.dim[2];
    g.n_words = (int)((n_cells + 31) / 32);
    const float infl = 1e-3f * cell;
    std::vector<std::pair<uint32_t, uint32_t>> ent;                  // (cell, triangle)
    ent.reserve((size_t)n_tri * 8);
    for (int32_t t = 0; t < n_tri; ++t) {
        float v[3][3];
        for (int k = 0; k < 3; ++k) vert(t, k, v[k]);
        int c0[3], c1[3];
        for (int d = 0; d < 3; ++d) {
            const float a = std::min(v[0][d], std::min(v[1][d], v[2][d])) - infl, b = std::max(v[0][d], std::max(v[1][d], v[2][d])) + infl;
            c0[d] = std::max(0, std::min(g.dim[d] - 1, (int)std::floor((a - g.gmin[d]) * g.inv_cell)));
            c1[d] = std::max(0, std::min(g.dim[d] - 1, (int)std::floor((b - g.gmin[d]) * g.inv_cell)));
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int x = c0[0]; x <= c1[0]; ++x) {
                    const float c[3] = {g.gmin[0] + ((float)x + 0.5f) * cell, g.gmin[1] + ((float)y + 0.5f) * cell,
                                        g.gmin[2] + ((float)z + 0.5f) * cell};
                    if (tri_box_overlap(c, 0.5f * cell + infl, v[0], v[1], v[2]))
                        ent.emplace_back((uint32_t)(((long long)z * g.dim[1] + y) * g.dim[0] + x), (uint32_t)t);
                }
    }
    std::sort(ent.begin(), ent.end());
    std::vector<uint32_t> bits(g.n_words, 0u), occ_start;
    for (size_t i = 0; i < ent.size(); ++i)
        if (i == 0 || ent[i].first != ent[i - 1].first) {
            bits[ent[i].first >> 5] |= 1u << (ent[i].first & 31);
            occ_start.push_back((uint32_t)i);
        }
    g.n_occ = (int)occ_start.size();
    occ_start.push_back((uint32_t)ent.size());
    g.n_entries = (long long)ent.size();
    // block distance field: Chebyshev distance (in 2x2x2 blocks) to the nearest block that holds an occupied cell, <= 15
    const int bd[3] = {(g.dim[0] + 1) / 2, (g.dim[1] + 1) / 2, (g.dim[2] + 1) / 2};
    const long long n_blocks = (long long)bd[0] * bd[1] * bd[2];
    std::vector<uint8_t> dist((size_t)n_blocks, 255);
    for (long long c = 0; c < n_cells; ++c)
        if ((bits[c >> 5] >> (c & 31)) & 1u) {
            const int x = (int)(c % g.dim[0]), y = (int)((c / g.dim[0]) % g.dim[1]), z = (int)(c / ((long long)g.dim[0] * g.dim[1]));
            dist[((size_t)(z >> 1) * bd[1] + (y >> 1)) * bd[0] + (x >> 1)] = 0;
        }
    for (int r = 1; r <= 15; ++r) {
        std::vector<size_t> grow;
        for (int z = 0; z < bd[2]; ++z)
            for (int y = 0; y < bd[1]; ++y)
                for (int x = 0; x < bd[0]; ++x) {
                    const size_t b = ((size_t)z * bd[1] + y) * bd[0] + x;
                    if (dist[b] != 255) continue;
                    bool near = false;
                    for (int dz = -1; dz <= 1 && !near; ++dz)
                        for (int dy = -1; dy <= 1 && !near; ++dy)
                            for (int dx = -1; dx <= 1 && !near; ++dx) {
                                const int X = x + dx, Y = y + dy, Z = z + dz;
                                if (X < 0 || Y < 0 || Z < 0 || X >= bd[0] || Y >= bd[1] || Z >= bd[2]) continue;
                                near = dist[((size_t)Z * bd[1] + Y) * bd[0] + X] == r - 1;
                            }
                    if (near) grow.push_back(b);
                }
        for (size_t b : grow) dist[b] = (uint8_t)r;
    }
    // sections, each padded to 16 bytes: bits | sbase | off16 | dist4 | occ_start | cell_tris
    auto pad4 = [](int64_t w) { return (w + 3) / 4 * 4; };
    const int64_t nsb = (g.n_words + 63) / 64;
    const int64_t w_bits = pad4(g.n_words), w_sb = pad4(nsb), w_off = pad4((g.n_words + 1) / 2), w_dist = pad4((n_blocks + 7) / 8),
                  w_occ = pad4((int64_t)occ_start.size());
    const int64_t words = w_bits + w_sb + w_off + w_dist + w_occ + 12 * (int64_t)ent.size();
    uint32_t* blob = (uint32_t*)calloc((size_t)words, 4);
    if (!blob) return DM_ERR_WORKSPACE;
    uint32_t* p = blob;
    std::memcpy(p, bits.data(), (size_t)g.n_words * 4);
    uint32_t* sb = blob + w_bits;
    uint16_t* off = (uint16_t*)(blob + w_bits + w_sb);
    uint32_t run = 0, in_block = 0;
    for (int w = 0; w < g.n_words; ++w) {
        if ((w & 63) == 0) { sb[w >> 6] = run; in_block = 0; }
        off[w] = (uint16_t)in_block;                                // < 64 * 32
        const uint32_t pc = (uint32_t)__builtin_popcount(bits[w]);
        run += pc; in_block += pc;
    }
    uint8_t* d4 = (uint8_t*)(blob + w_bits + w_sb + w_off);
    for (long long b = 0; b < n_blocks; ++b) d4[b >> 1] |= (uint8_t)(std::min<int>(dist[(size_t)b], 15) << ((b & 1) * 4));
    std::memcpy(blob + w_bits + w_sb + w_off + w_dist, occ_start.data(), occ_start.size() * 4);
    float* ct = (float*)(blob + w_bits + w_sb + w_off + w_dist + w_occ);
    for (size_t i = 0; i < ent.size(); ++i) {
        std::memcpy(ct + 12 * i, tris12 + 12 * (size_t)ent[i].second, 48);
        const int32_t id = (int32_t)ent[i].second;
        std::memcpy(ct + 12 * i + 3, &id, 4);                       // (slot 3 is unused by the intersection test)
    }
    g.bits = g.sbase = g.occ_start = nullptr; g.off16 = nullptr; g.dist4 = nullptr; g.cell_tris = nullptr;
    *blob_out = blob; *blob_words = words;
    return DM_OK;
}

void dm_host_free(void* p) { free(p); }

}  // extern "C"
