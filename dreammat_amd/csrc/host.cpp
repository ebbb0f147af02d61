// Host-side entry points of libdreammat_hip.so (no device code): mesh topology for the antialias
// kernels (nvdiffrast builds this edge->opposite-vertex hash inside dr.antialias on every call;
// the DreamMat mesh is fixed, so it is built once per mesh) and library introspection.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "dm_common.h"

extern "C" {

int dm_abi_version(void) { return 1; }

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

}  // extern "C"
