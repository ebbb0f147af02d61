// Host-side entry points of libdreammat_hip.so (no device code): mesh topology for the antialias
// kernels (nvdiffrast builds this edge->opposite-vertex hash inside dr.antialias on every call;
// the DreamMat mesh is fixed, so it is built once per mesh) and library introspection.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "bvh_core.h"
#include "dm_common.h"

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

int dm_abi_version(void) { return 3; }     // 3: + dm_attention_selected, dm_cat_add_bf16; dm_attention_select names changed

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
