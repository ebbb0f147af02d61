// Uniform-grid any-hit traversal for the Monte-Carlo shading branch: the same question as bvh_core.h --
// "is the direction occluded", `_raytracing.create_raytracer(v, f).trace(...)` as DreamMatMaterial.get_lights uses it
// (threestudio/models/materials/dreammat_material.py:490-507, raytracing_renderer.py:318-324) -- answered by a 3-D DDA over
// an occupancy bit grid of the fixed mesh instead of a tree walk.
//
// Why a grid here.  The shading rays all start ON the surface and 4 of 5 leave the object without hitting anything.  A BVH
// walk pays ~19 dependent 128-byte node fetches and ~30 triangle tests for such a ray (tools/bvh_stats.py), every one of
// them a scattered L1/L2 access: the wave-per-pixel kernel sat at 0.48 G rays/s (profiles/r02_mc_probe.json).  The grid's
// occupancy bits (<= 96^3 bits = 108 KB) live in LDS, so the empty space a ray crosses costs a few VALU instructions and
// one ds_read per cell and NO memory traffic; global memory is touched only in occupied cells (rank table -> triangle list
// -> triangles), i.e. around the origin and at real occluders.
//
// Exactness.  The answer is a boolean over the SAME triangle test (dm_bvh_ray_triangle), so it equals the BVH's as long as
// no triangle the ray hits is skipped: the voxelisation is conservative (exact triangle / box overlap, boxes inflated by
// 1e-3 cell -- dm_grid_build in host.cpp), which also absorbs the DDA's rounding at cell faces.
// Host + device: the same code runs in tests/hostemu on the CPU.
//
// Layout (built on the host by dm_grid_build):
//   cell (x, y, z) -> c = (z * dim[1] + y) * dim[0] + x;  bits[c >> 5] bit (c & 31) = some triangle overlaps the cell
//   rank of an occupied cell = sbase[w >> 6] + off16[w] + popcount(bits[w] & lower bits), w = c >> 5  (sbase: occupied cells
//                  before the 64-word block, off16: occupied cells in the block's words before w -- 16 bits suffice, and
//                  bits + off16 + sbase of a 96^3 grid are 152 KB: all three live in LDS, so finding a cell's triangle list
//                  costs no global access)
//   dist4          = one nibble per 2x2x2 block of cells (block (bx, by, bz) -> nibble (bz * bd[1] + by) * bd[0] + bx, bd =
//                  ceil(dim / 2), low nibble first): Chebyshev distance in BLOCKS to the nearest block with an occupied cell,
//                  capped at 15.  A ray standing in a block of distance D >= 2 knows that the cube of blocks within D - 1
//                  around it is empty and jumps to the face through which it leaves that cube: open space costs a few
//                  leaps instead of a step per cell (the per-cell walk was 64 % of the kernel's instructions)
//   occ_start[r]   = first record of the r-th occupied cell in cell_tris (occ_start[n_occ] = n_entries)
//   cell_tris[e]   = 48 B {v0.xyz, id, e1.xyz, 0, e2.xyz, 0}: the triangles of the cells INLINE, cell after cell (a triangle
//                  that overlaps k cells is stored k times; id = its number in dm_bvh_build's leaf order, as int bits) -- one
//                  dependent global access (occ_start) between the LDS tables and the triangle data
#pragma once
#include <stdint.h>

#include "bvh_core.h"

struct DmGrid {
    float gmin[3];
    float cell, inv_cell;
    int dim[3];
    int n_words, n_occ;
    long long n_entries;
    const uint32_t* bits;
    const uint32_t* sbase;        // [ceil(n_words / 64)]
    const uint16_t* off16;        // [n_words] (+ padding to a whole uint32)
    const uint8_t* dist4;         // [ceil(bd0 * bd1 * bd2 / 2)] nibbles
    const uint32_t* occ_start;
    const float* cell_tris;       // [n_entries][12]
};

// the three tables a traversal reads per cell: the kernel's LDS copies or the global / host arrays
struct DmGridTables {
    const uint32_t* bits; const uint32_t* sbase; const uint16_t* off16; const uint8_t* dist4;
};

// 3-D DDA state of one ray (scalars, not arrays: an axis chosen at run time would index them through scratch memory)
struct DmDda {
    int ix, iy, iz;
    float tx, ty, tz;             // ray parameter of the next cell face per axis
    float ivx, ivy, ivz;          // 1 / direction (+-3e38 for a zero component)
    float t1;                     // end of the walk: exit from the grid's box or t_max
};

// put the walk at ray parameter t: the cell that contains the point and the next faces
DM_HD void dm_dda_seek(const DmGrid& g, DmDda& s, float ox, float oy, float oz, float dx, float dy, float dz, float t) {
    const float big = 3.0e38f;
    auto cell_of = [&](float ok, float dk, float gm, int dim) {
        int c = (int)floorf((ok + dk * t - gm) * g.inv_cell);
        return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
    };
    s.ix = cell_of(ox, dx, g.gmin[0], g.dim[0]); s.iy = cell_of(oy, dy, g.gmin[1], g.dim[1]); s.iz = cell_of(oz, dz, g.gmin[2], g.dim[2]);
    s.tx = fabsf(dx) > 1e-30f ? (g.gmin[0] + (float)(s.ix + (dx > 0.f ? 1 : 0)) * g.cell - ox) * s.ivx : big;
    s.ty = fabsf(dy) > 1e-30f ? (g.gmin[1] + (float)(s.iy + (dy > 0.f ? 1 : 0)) * g.cell - oy) * s.ivy : big;
    s.tz = fabsf(dz) > 1e-30f ? (g.gmin[2] + (float)(s.iz + (dz > 0.f ? 1 : 0)) * g.cell - oz) * s.ivz : big;
}

// false: the ray misses the grid's box (or is NaN)
DM_HD bool dm_dda_init(const DmGrid& g, DmDda& s, float ox, float oy, float oz, float dx, float dy, float dz, float t_max) {
    const float big = 3.0e38f;
    s.ivx = fabsf(dx) > 1e-30f ? 1.0f / dx : (dx < 0.f ? -big : big);
    s.ivy = fabsf(dy) > 1e-30f ? 1.0f / dy : (dy < 0.f ? -big : big);
    s.ivz = fabsf(dz) > 1e-30f ? 1.0f / dz : (dz < 0.f ? -big : big);
    float t0 = 0.f, t1 = t_max;
    {
        float ta = (g.gmin[0] - ox) * s.ivx, tb_ = (g.gmin[0] + g.dim[0] * g.cell - ox) * s.ivx;
        t0 = fmaxf(t0, fminf(ta, tb_)); t1 = fminf(t1, fmaxf(ta, tb_));
        ta = (g.gmin[1] - oy) * s.ivy; tb_ = (g.gmin[1] + g.dim[1] * g.cell - oy) * s.ivy;
        t0 = fmaxf(t0, fminf(ta, tb_)); t1 = fminf(t1, fmaxf(ta, tb_));
        ta = (g.gmin[2] - oz) * s.ivz; tb_ = (g.gmin[2] + g.dim[2] * g.cell - oz) * s.ivz;
        t0 = fmaxf(t0, fminf(ta, tb_)); t1 = fminf(t1, fmaxf(ta, tb_));
    }
    if (!(t1 >= t0)) return false;
    s.t1 = t1;
    dm_dda_seek(g, s, ox, oy, oz, dx, dy, dz, t0);
    return true;
}

// one step to the next cell along the ray; false = the ray has left the box / passed t_max.  Branch-free: the 64 rays of a
// wave cross faces of all three axes in the same step, a three-way branch would run all three arms one after the other.
DM_HD bool dm_dda_step(const DmGrid& g, DmDda& s) {
    const bool xm = s.tx <= s.ty && s.tx <= s.tz;
    const bool ym = !xm && s.ty <= s.tz;
    const bool zm = !xm && !ym;
    const float tmin = xm ? s.tx : (ym ? s.ty : s.tz);
    if (tmin > s.t1) return false;                     // the next face lies beyond the box exit / t_max
    s.ix += xm ? (s.ivx > 0.f ? 1 : -1) : 0;
    s.iy += ym ? (s.ivy > 0.f ? 1 : -1) : 0;
    s.iz += zm ? (s.ivz > 0.f ? 1 : -1) : 0;
    s.tx += xm ? g.cell * fabsf(s.ivx) : 0.f;
    s.ty += ym ? g.cell * fabsf(s.ivy) : 0.f;
    s.tz += zm ? g.cell * fabsf(s.ivz) : 0.f;
    return (unsigned)s.ix < (unsigned)g.dim[0] && (unsigned)s.iy < (unsigned)g.dim[1] && (unsigned)s.iz < (unsigned)g.dim[2];
}

// the move from an EMPTY cell whose block distance is D: a plain step next to the surface (D < 2), a leap to the exit of the
// empty cube of blocks around the cell where the distance field allows one (see the layout notes).  false = the ray is gone.
DM_HD bool dm_dda_advance_d(const DmGrid& g, DmDda& s, int D, float ox, float oy, float oz, float dx, float dy, float dz) {
    if (D < 2) return dm_dda_step(g, s);
    const int bx = s.ix >> 1, by = s.iy >> 1, bz = s.iz >> 1;
    // faces of the cube of blocks [b - (D - 1), b + (D - 1)] on the sides the ray travels towards
    const float two = 2.0f * g.cell;
    const float px = g.gmin[0] + (float)(dx > 0.f ? bx + D : bx - D + 1) * two;
    const float py = g.gmin[1] + (float)(dy > 0.f ? by + D : by - D + 1) * two;
    const float pz = g.gmin[2] + (float)(dz > 0.f ? bz + D : bz - D + 1) * two;
    const float big = 3.0e38f;
    const float ex = fabsf(dx) > 1e-30f ? (px - ox) * s.ivx : big;
    const float ey = fabsf(dy) > 1e-30f ? (py - oy) * s.ivy : big;
    const float ez = fabsf(dz) > 1e-30f ? (pz - oz) * s.ivz : big;
    const float t = fminf(ex, fminf(ey, ez));
    if (!(t <= s.t1)) return false;                    // out of the box / past t_max before the cube ends
    dm_dda_seek(g, s, ox, oy, oz, dx, dy, dz, t);
    return true;
}

DM_HD bool dm_dda_advance(const DmGrid& g, const DmGridTables& tb, DmDda& s, float ox, float oy, float oz, float dx, float dy, float dz) {
    const int bd0 = (g.dim[0] + 1) >> 1, bd1 = (g.dim[1] + 1) >> 1;
    const int bi = ((s.iz >> 1) * bd1 + (s.iy >> 1)) * bd0 + (s.ix >> 1);
    return dm_dda_advance_d(g, s, (tb.dist4[bi >> 1] >> ((bi & 1) * 4)) & 15, ox, oy, oz, dx, dy, dz);
}

DM_HD int dm_dda_cell(const DmGrid& g, const DmDda& s) { return (s.iz * g.dim[1] + s.iy) * g.dim[0] + s.ix; }

// rank of occupied cell c (w = its occupancy word)
DM_HD uint32_t dm_grid_rank(const DmGridTables& tb, int c, uint32_t w) {
    const int wi = c >> 5;
    return tb.sbase[wi >> 6] + (uint32_t)tb.off16[wi] + (uint32_t)__builtin_popcount(w & ((1u << (c & 31)) - 1u));
}

// One ray on its own: the reference form of the query (tests/hostemu, the stand-alone ray kernel, the one-thread-per-pixel
// shading kernel).  The one-wave-per-pixel kernel walks the same cells with the same tests, cooperatively
// (grid_trace_wave in mc_shade.hip).  (Bounded by the grid: every step moves one cell along one axis.)
DM_HD bool dm_grid_any_hit(const DmGrid& g, const DmGridTables& tb, float ox, float oy, float oz, float dx, float dy, float dz,
                           float t_max) {
    DmDda s;
    if (!dm_dda_init(g, s, ox, oy, oz, dx, dy, dz, t_max)) return false;
    for (int guard = 0; guard < 3 * 1024; ++guard) {
        const int c = dm_dda_cell(g, s);
        const uint32_t w = tb.bits[c >> 5];
        if ((w >> (c & 31)) & 1u) {
            const uint32_t r = dm_grid_rank(tb, c, w);
            const uint32_t e0 = g.occ_start[r], e1 = g.occ_start[r + 1];
            for (uint32_t e = e0; e < e1; ++e)
                if (dm_bvh_ray_triangle(g.cell_tris + 12 * (size_t)e, ox, oy, oz, dx, dy, dz, t_max)) return true;
            if (!dm_dda_step(g, s)) return false;
        } else if (!dm_dda_advance(g, tb, s, ox, oy, oz, dx, dy, dz)) {
            return false;
        }
    }
    return false;
}
