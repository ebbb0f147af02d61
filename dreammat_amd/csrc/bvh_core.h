// BVH any-hit traversal core for the Monte-Carlo shading branch (groundwork): replaces
// `_raytracing.create_raytracer(v, f).trace(...)` (threestudio/models/renderers/raytracing_renderer.py:31,61,318-324)
// as the reference uses it from DreamMatMaterial.get_lights (dreammat_material.py:490-507): only "is the direction
// occluded" matters -- hit <=> some intersection with 0 < t < t_max (the reference tests the traced depth < 10).
// Host + device: the same code runs in tests/hostemu on the CPU.
//
// Layout (built on the host by dm_bvh_build, csrc/host.cpp):
//   nodes[i]  = 32 B {bmin.xyz, a, bmax.xyz, b}: leaf when b > 0: triangles [a, a+b) of `tris`; else children a, a+1
//   tris[j]   = 48 B {v0.xyz, 0, e1.xyz, 0, e2.xyz, 0} in leaf order (e1 = v1-v0, e2 = v2-v0)
#pragma once
#include "dm_common.h"

struct DmBvhNode {
    float bmin[3]; int a;
    float bmax[3]; int b;
};

DM_HD bool dm_bvh_ray_triangle(const float* __restrict__ t12, float ox, float oy, float oz, float dx, float dy, float dz,
                               float t_max) {
    // double-sided Moeller-Trumbore
    const float v0x = t12[0], v0y = t12[1], v0z = t12[2];
    const float e1x = t12[4], e1y = t12[5], e1z = t12[6];
    const float e2x = t12[8], e2y = t12[9], e2z = t12[10];
    const float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    if (fabsf(det) < 1e-20f) return false;
    const float inv = 1.0f / det;
    const float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    const float u = (tx * px + ty * py + tz * pz) * inv;
    if (u < 0.f || u > 1.f) return false;
    const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    const float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (v < 0.f || u + v > 1.f) return false;
    const float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    return t > 0.f && t < t_max;
}

// true if the ray o + t d, 0 < t < t_max, hits any triangle.  Stack depth 64 covers any tree dm_bvh_build produces
// (it splits at the median when the SAH split degenerates, so depth <= ~2 log2(n)).
DM_HD bool dm_bvh_any_hit(const DmBvhNode* __restrict__ nodes, const float* __restrict__ tris, float ox, float oy, float oz,
                          float dx, float dy, float dz, float t_max) {
    const float big = 3.0e38f;
    const float ix = fabsf(dx) > 1e-30f ? 1.0f / dx : (dx < 0.f ? -big : big);
    const float iy = fabsf(dy) > 1e-30f ? 1.0f / dy : (dy < 0.f ? -big : big);
    const float iz = fabsf(dz) > 1e-30f ? 1.0f / dz : (dz < 0.f ? -big : big);
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const DmBvhNode nd = nodes[stack[--sp]];
        // slab test against [0, t_max]
        float t0 = (nd.bmin[0] - ox) * ix, t1 = (nd.bmax[0] - ox) * ix;
        float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
        t0 = (nd.bmin[1] - oy) * iy; t1 = (nd.bmax[1] - oy) * iy;
        tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
        t0 = (nd.bmin[2] - oz) * iz; t1 = (nd.bmax[2] - oz) * iz;
        tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
        if (!(tf >= fmaxf(tn, 0.f)) || tn > t_max) continue;
        if (nd.b > 0) {
            for (int k = 0; k < nd.b; ++k)
                if (dm_bvh_ray_triangle(tris + 12 * (size_t)(nd.a + k), ox, oy, oz, dx, dy, dz, t_max)) return true;
        } else if (sp + 2 <= 64) {
            stack[sp++] = nd.a;
            stack[sp++] = nd.a + 1;
        }
    }
    return false;
}

// ---- 4-wide variant (opt-in, DREAMMAT_BVH=4): the binary tree collapsed two levels at a time, the four child boxes
// stored IN the parent (SoA, 128 B = 8 x 16 B loads).  One node fetch decides four subtrees, so a ray makes a quarter to
// a third of the DEPENDENT memory round trips of the pop-then-test binary layout (tools/bvh_stats.py counts them on
// the CPU).  child k: b[k] > 0 leaf = triangles [a[k], a[k]+b[k]); b[k] == 0 internal node a[k]; b[k] < 0 empty slot.
struct DmBvhNode4 {
    float lo[3][4];
    float hi[3][4];
    int a[4];
    int b[4];
};

DM_HD bool dm_bvh4_any_hit(const DmBvhNode4* __restrict__ nodes, const float* __restrict__ tris, float ox, float oy, float oz,
                           float dx, float dy, float dz, float t_max) {
    const float big = 3.0e38f;
    const float ix = fabsf(dx) > 1e-30f ? 1.0f / dx : (dx < 0.f ? -big : big);
    const float iy = fabsf(dy) > 1e-30f ? 1.0f / dy : (dy < 0.f ? -big : big);
    const float iz = fabsf(dz) > 1e-30f ? 1.0f / dz : (dz < 0.f ? -big : big);
    int stack[48];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const DmBvhNode4 nd = nodes[stack[--sp]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (nd.b[k] < 0) continue;
            float t0 = (nd.lo[0][k] - ox) * ix, t1 = (nd.hi[0][k] - ox) * ix;
            float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
            t0 = (nd.lo[1][k] - oy) * iy; t1 = (nd.hi[1][k] - oy) * iy;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            t0 = (nd.lo[2][k] - oz) * iz; t1 = (nd.hi[2][k] - oz) * iz;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            if (!(tf >= fmaxf(tn, 0.f)) || tn > t_max) continue;
            if (nd.b[k] > 0) {
                for (int j = 0; j < nd.b[k]; ++j)
                    if (dm_bvh_ray_triangle(tris + 12 * (size_t)(nd.a[k] + j), ox, oy, oz, dx, dy, dz, t_max)) return true;
            } else if (sp < 48) {
                stack[sp++] = nd.a[k];
            }
        }
    }
    return false;
}

