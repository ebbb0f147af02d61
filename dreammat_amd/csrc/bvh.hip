// Any-hit ray queries against the mesh BVH (csrc/bvh_core.h): the occlusion test behind DreamMatMaterial.get_lights
// (threestudio/models/materials/dreammat_material.py:490-507 -> raytracing_renderer.py:318-324).  Groundwork for
// SURVEY row f-1 (Monte-Carlo shading): the fused shading kernel will call dm_bvh_any_hit per sample direction; this
// entry point exposes the traversal on its own.  One thread per ray; rays of a wave start at neighbouring pixels
// and diverge with the sample direction, so node fetches are L2 hits rather than coalesced loads.
#include "bvh_core.h"
#include "grid_core.h"

namespace {

__global__ __launch_bounds__(256) void k_bvh_any_hit(const DmBvhNode* __restrict__ nodes, const float* __restrict__ tris,
                                                     const float* __restrict__ org, const float* __restrict__ dir, long long n,
                                                     float t_max, unsigned char* __restrict__ hit) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hit[i] = dm_bvh_any_hit(nodes, tris, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i], dir[3 * i + 1],
                            dir[3 * i + 2], t_max) ? 1 : 0;
}

// the same query through the occupancy grid (csrc/grid_core.h), bits read from global memory: the stand-alone entry point is
// for tests and measurements, the shading kernel keeps the bits in LDS
__global__ __launch_bounds__(256) void k_grid_any_hit(DmGrid g, const float* __restrict__ org, const float* __restrict__ dir,
                                                      long long n, float t_max, unsigned char* __restrict__ hit) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DmGridTables tb = {g.bits, g.sbase, g.off16, g.dist4};
    hit[i] = dm_grid_any_hit(g, tb, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], t_max) ? 1 : 0;
}

}  // namespace

extern "C" {

// grid: host struct (dm_grid) with device pointers.
int dm_grid_any_hit_rays(const void* grid, const float* origins, const float* dirs, long long n, float t_max, unsigned char* hit,
                         hipStream_t stream) {
    if (!grid || !origins || !dirs || !hit || n < 0 || !(t_max > 0.f)) return DM_ERR_ARG;
    const DmGrid g = *(const DmGrid*)grid;
    if (!g.bits || !g.sbase || !g.off16 || !g.dist4 || !g.occ_start || !g.cell_tris || g.n_words <= 0) return DM_ERR_ARG;
    if (n == 0) return DM_OK;
    if (n > 0x7fffffffLL * 256LL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL(k_grid_any_hit, dim3((unsigned)dm_div_up(n, 256)), dim3(256), 0, stream, g, origins, dirs, n, t_max, hit);
    DM_LAUNCH_CHECK();
    return DM_OK;
}


// nodes / tris: device copies of dm_bvh_build's outputs; origins, dirs [n,3] fp32; hit [n] bytes (1 = occluded).
int dm_bvh_any_hit_rays(const void* nodes, const float* tris, const float* origins, const float* dirs, long long n,
                        float t_max, unsigned char* hit, hipStream_t stream) {
    if (!nodes || !tris || !origins || !dirs || !hit || n < 0 || !(t_max > 0.f)) return DM_ERR_ARG;
    if (n == 0) return DM_OK;
    if (n > 0x7fffffffLL * 256LL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL(k_bvh_any_hit, dim3((unsigned)dm_div_up(n, 256)), dim3(256), 0, stream, (const DmBvhNode*)nodes, tris,
                       origins, dirs, n, t_max, hit);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
