/*
 * dreammat_hip.h -- C ABI of libdreammat_hip.so (gfx950 / MI355X only).
 *
 * The reference (zzzyuqing/DreamMat) has NO native code and no FFI of its own: every kernel on
 * its SDS hot path lives in CUDA-only pip dependencies that it calls through their Python API.
 * Each entry point below replaces one of those calls; the citation names the reference call site
 * (paths relative to threestudio_dreammat/threestudio/).  INTEGRATION.md shows the ctypes stubs a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host
 *   - sizes are explicit; nothing is allocated inside: outputs and scratch come from the caller
 *     (scratch sizes from the *_workspace_bytes queries)
 *   - `stream` is the hipStream_t the work is enqueued on (0 = null stream); calls are async
 *   - return: 0 = ok, negative = DM_ERR_* argument/workspace/unsupported, positive = hipError_t
 *   - "row tensors" [N,C] are addressed as p[i*row_stride + c*col_stride] (strides in elements) so
 *     the reference's [N,C] layout and the internal coalesced [C,N] layout both work copy-free
 *   - `n_dev` arguments are DEVICE int32 row counts (produced by dm_gbuffer_compact) so that no
 *     host synchronisation is needed between kernels; `n_max` only sizes the launch
 */
#ifndef DREAMMAT_HIP_H
#define DREAMMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dm_stream_t; /* == hipStream_t */

#define DM_OK 0
#define DM_ERR_ARG (-1)
#define DM_ERR_WORKSPACE (-2)
#define DM_ERR_UNSUPPORTED (-3)

int dm_abi_version(void);

/* ---- mesh (host) ------------------------------------------------------------------------- */
/* Edge -> opposite-vertex table used by the antialias kernels.  nvdiffrast rebuilds this hash
 * inside every dr.antialias call (utils/rasterize.py:56); the DreamMat mesh is fixed
 * (models/renderers/raytracing_renderer.py:101), so it is built once.  tri_host/opp_host: [n_tri,3]. */
int dm_mesh_build_topology(const int32_t* tri_host, int32_t n_tri, int32_t* opp_host);

/* ---- rasterize ---------------------------------------------------------------------------- */
/* NVDiffRasterizerContext.vertex_transform (utils/rasterize.py:22-28):
 * pos_clip[B,n_vert,4] = [v_pos,1] @ mvp[b]^T.  v_pos [n_vert,3], mvp [B,4,4] row-major. */
int dm_vertex_transform(const float* v_pos, int n_vert, const float* mvp, int B, float* pos_clip,
                        dm_stream_t stream);

/* dr.rasterize(ctx, pos, tri, (H,W)) (utils/rasterize.py:37 <- raytracing_renderer.py:124).
 * rast[B,H,W,4] = (u, v, z/w, tri_id+1), zeros where empty.  ws from dm_raster_workspace_bytes
 * (a larger ws gives more bin capacity). */
size_t dm_raster_workspace_bytes(int B, int n_tri, int H, int W);
int dm_rasterize(const float* pos_clip, int B, int n_vert, const int32_t* tri, int n_tri, int H, int W,
                 float* rast, void* ws, size_t ws_bytes, dm_stream_t stream);
/* Blocking: *overflow_host = 1 if the last dm_rasterize on `ws` ran out of triangle-bin capacity
 * (result then incomplete: retry with a larger workspace). */
int dm_raster_overflowed(const void* ws, dm_stream_t stream, int* overflow_host);

/* dr.interpolate(attr[None], rast, tri) (utils/rasterize.py:66-68 <- raytracing_renderer.py:136,150,178).
 * attr [n_vert,C] shared by all views; out [n_pix,C], zeros where empty. */
int dm_interpolate(const float* attr, int n_vert, int C, const int32_t* tri, const float* rast, long long n_pix,
                   float* out, dm_stream_t stream);

/* ---- antialias ---------------------------------------------------------------------------- */
/* dr.antialias(color, rast, pos, tri) (utils/rasterize.py:56 <- raytracing_renderer.py:127,147,199)
 * split into the per-step pair analysis (plan[B,H,W,2]) and its application, so the three calls of
 * one step and the backward pass share one analysis.  Only the colour gradient exists (fixed mesh). */
int dm_antialias_plan(const float* pos_clip, int B, int n_vert, const int32_t* tri, const int32_t* opp,
                      const float* rast, int H, int W, float* plan, dm_stream_t stream);
int dm_antialias_apply(const float* color, const float* plan, int B, int H, int W, int C, float* out,
                       dm_stream_t stream); /* C in {1,3,4} */
int dm_antialias_grad(const float* dout, const float* plan, int B, int H, int W, int C, float* dcolor,
                      dm_stream_t stream);

/* ---- G-buffer ----------------------------------------------------------------------------- */
/* The `x[selector]` compaction + interpolate + normalize + tangent-plane jitter of
 * raytracing_renderer.py:136-173 (get_orthogonal_directions :306-316) in one pass.
 * Outputs are SoA with pitch `cap` rows: pix_idx[cap], pos[3,cap], pos_jitter[3,cap], nrm[3,cap],
 * view[3,cap] (= -rays_d); *n_out = number of covered pixels (row-major order).
 * jitter_u in [0,1), jitter_n ~ N(0,1), both [n_pix] (pass NULL for both to skip the jitter). */
size_t dm_gbuffer_workspace_bytes(long long n_pix);
int dm_gbuffer_compact(const float* rast, long long n_pix, const int32_t* tri, const float* v_pos,
                       const float* v_nrm, const float* rays_d, const float* jitter_u, const float* jitter_n,
                       float jitter_eps, long long cap, int32_t* pix_idx, float* pos, float* pos_jitter, float* nrm,
                       float* view, int32_t* n_out, void* ws, size_t ws_bytes, dm_stream_t stream);
/* The same rows enumerated in TILE order (ABI v9; the product's default): one workgroup per 16 x 16 macro tile (row-major
 * inside a view, views in order), one wave per 8 x 8 sub-tile, lanes in Morton order inside it -- so the 64 rows any later
 * kernel's wave works on are an 8 x 8 pixel block (neighbouring normals / reflection vectors / positions: cube-map lines and
 * hash-grid cells shared across the wave) instead of a 64-pixel scanline run.  rast is [B,H,W,4]; pix_idx[i] is still the
 * row-major global pixel index b*H*W + y*W + x of compacted row i, so scatter / gather / antialias do not change. */
size_t dm_gbuffer_tiled_workspace_bytes(int B, int H, int W);
int dm_gbuffer_compact_tiled(const float* rast, int B, int H, int W, const int32_t* tri, const float* v_pos,
                             const float* v_nrm, const float* rays_d, const float* jitter_u, const float* jitter_n,
                             float jitter_eps, long long cap, int32_t* pix_idx, float* pos, float* pos_jitter,
                             float* nrm, float* view, int32_t* n_out, void* ws, size_t ws_bytes, dm_stream_t stream);

/* ControlNet depth [B,H,W,1] / view-normal [B,H,W,3] maps before antialias
 * (raytracing_renderer.py:129-147, compute_controlnet_normals :326-331).  minmax_ws: >= 8*B bytes. */
int dm_control_maps(const float* rast, int B, int H, int W, const int32_t* tri, const float* v_nrm,
                    const float* w2c, float* depth, float* normal, void* minmax_ws, dm_stream_t stream);

/* color[selector] = values / values = dense[selector] (raytracing_renderer.py:198,201-207) */
int dm_scatter_rows(const int32_t* pix_idx, const int32_t* n_dev, long long n_max, const float* src,
                    long long src_row_stride, long long src_col_stride, int C, float* dst, dm_stream_t stream);
int dm_gather_rows(const int32_t* pix_idx, const int32_t* n_dev, long long n_max, const float* src, int C,
                   float* dst, long long dst_row_stride, long long dst_col_stride, dm_stream_t stream);

/* ---- feature field ------------------------------------------------------------------------ */
/* tcnn.Encoding(3, HashGrid...)(contract_to_unisphere(x)) (models/networks.py:55-64 <-
 * models/geometry/dreammat_mesh.py:239-254; geometry/base.py:20-32 bounded branch with bbox +-radius).
 * table [sum(lv_size), 2] fp32; enc [M, 2*n_levels]. Level tables are host arrays. */
int dm_hashgrid_fwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                    const float* table, int n_levels, const float* lv_scale_host, const uint32_t* lv_res_host,
                    const uint32_t* lv_size_host, const uint32_t* lv_offset_host, float radius, float* enc,
                    long long enc_rs, long long enc_cs, dm_stream_t stream);
/* Backward of the above wrt the table; ADDS into dtable (caller zeroes it once per step). */
/* The same gradient with the HASHED levels (resolution^3 > table size) routed through bins instead of one global atomic pair per
 * corner (fp32 atomics retire at ~20 G/s on MI355X): tuples are binned by table region, summed in LDS, added once.  Workspace:
 * dm_hashgrid_bwd_workspace_bytes() bytes of device memory, 256 B aligned, contents irrelevant.  ADDS into dtable. */
size_t dm_hashgrid_bwd_workspace_bytes(long long m_max, int n_levels, const uint32_t* lv_res, const uint32_t* lv_size);
int dm_hashgrid_bwd_binned(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                           const float* denc, long long denc_rs, long long denc_cs, int n_levels, const float* lv_scale,
                           const uint32_t* lv_res, const uint32_t* lv_size, const uint32_t* lv_offset, float radius,
                           float* dtable, void* workspace, size_t workspace_bytes, dm_stream_t stream);
int dm_hashgrid_bwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                    const float* denc, long long denc_rs, long long denc_cs, int n_levels,
                    const float* lv_scale_host, const uint32_t* lv_res_host, const uint32_t* lv_size_host,
                    const uint32_t* lv_offset_host, float radius, float* dtable, dm_stream_t stream);
/* The 2-D grid of the uv-space field (DreamMatMesh n_input_dims = 2, dreammat_mesh.py:128-135, 246-250): x [M,2] texture
 * coordinates by strides, contracted with the same +-radius box; level sizes min(next_multiple(res^2, 8), 2^log2_hashmap_size);
 * bwd scatter-adds into dtable (zeroed by the caller) with one atomic pair per corner. */
int dm_hashgrid2d_fwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max, const float* table,
                      int n_levels, const float* lv_scale, const uint32_t* lv_res, const uint32_t* lv_size,
                      const uint32_t* lv_offset, float radius, float* enc, long long enc_rs, long long enc_cs, dm_stream_t stream);
int dm_hashgrid2d_bwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max, const float* denc,
                      long long denc_rs, long long denc_cs, int n_levels, const float* lv_scale, const uint32_t* lv_res,
                      const uint32_t* lv_size, const uint32_t* lv_offset, float radius, float* dtable, dm_stream_t stream);
/* The feature network behind the encoding (dreammat_mesh.py:246-254 -> networks.py:150-187, VanillaMLP: bias-free
 * Linear(n_in, 64) -> ReLU -> Linear(64, n_out), fp32), fused: x [n_in][M] and y / dx likewise by feature stride (points
 * contiguous: the layout dm_hashgrid_fwd writes), w1 [64][n_in], w2 [n_out][64]; n_in = 16 | 32, n_out <= 8.  bwd writes dx
 * and ADDS into dw1 / dw2 (float atomics, one set per workgroup); dy[m * dy_rs + k * dy_cs].  The backward runs its five
 * products on the bf16 matrix pipe with SPLIT operands (a = hi + lo, 16 significant bits, three MFMAs per product; 24 bits / six
 * MFMAs for the pre-activation whose sign gates the ReLU): results within 1e-5 of fp32 math (csrc/field_mlp.hip, round 6). */
int dm_field_mlp_fwd(const float* x, long long x_fs, long long M, const float* w1, const float* w2, int n_in, int n_out, float* y,
                     long long y_fs, dm_stream_t stream);
int dm_field_mlp_bwd(const float* x, long long x_fs, long long M, const float* w1, const float* w2, int n_in, int n_out,
                     const float* dy, long long dy_rs, long long dy_cs, float* dx, long long dx_fs, float* dw1, float* dw2,
                     dm_stream_t stream);

/* ---- material / shading ------------------------------------------------------------------- */
/* Pre-filtered environment atlas (envlight.EnvLight equivalents for all env maps, built once at
 * configure time) + the FG LUT of models/materials/dreammat_material.py:399-404.  Texel format per
 * `texel_format`; every cube face carries a 1-texel border.  Passed by pointer to a HOST struct. */
typedef struct dm_env_atlas {
    const float* spec;         /* [n_env][mips][6][(R+2)][(R+2)][4] */
    const float* diff;         /* [n_env][6][(Rd+2)][(Rd+2)][4] */
    const float* fg_lut;       /* [lut_res][lut_res][2] */
    long long spec_env_stride; /* in texels */
    long long diff_env_stride;
    long long mip_off[8];      /* texel offset of each mip inside one env */
    int mip_res[8];
    int n_mips, diff_res, lut_res;
    float min_rough_mip, max_rough_mip; /* envlight's 0.08 / 0.5 */
    int texel_format;          /* spec / diff texels: 0 = RGBA fp32 (16 B); 1 = RGBA fp16 (8 B); 2 = RGB18E8 (8 B: three 18-bit
                                * mantissas R[0,18) G[18,36) B[36,54) + shared exponent E[55,63), value = m * 2^(E-127)) */
    const float* fg_pairs;     /* optional [lut_res][lut_res+1][4]: entry (row, x0+1) = {lut[row][max(x0,0)], lut[row][min(x0+1,
                                * lut_res-1)]}, the clamped x-pair of a bilinear row as one 16 B load; NULL = plain LUT taps */
} dm_env_atlas;
typedef struct dm_mat_cfg { float min_metallic, max_metallic, min_roughness, max_roughness; } dm_mat_cfg;

/* pix_idx[i] = view * HW + pixel; env_of_view [n_views] maps the view to its environment map.
 * DreamMatMaterial.forward, use_raytracing=False branch (dreammat_material.py:746-762) fused with
 * shade_splitsum (:679-711): sigmoid activation -> albedo/metallic/roughness -> FG LUT fetch ->
 * diffuse + specular env lookups -> clamp.  One thread per covered pixel, all views in one launch.
 * The 7 dbg_* outputs (dense [N,3]/[N,1]) are the reference's logging buffers; pass all NULL to skip. */
int dm_shade_fwd(const dm_env_atlas* atlas_host, const dm_mat_cfg* mat_host, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, int n_views, float* color, long long color_rs,
                 long long color_cs, float* dbg_albedo, float* dbg_spec_light, float* dbg_diff_light,
                 float* dbg_spec_color, float* dbg_diff_color, float* dbg_metallic, float* dbg_roughness,
                 dm_stream_t stream);
/* d loss/d features from d loss/d color (recomputes the forward; nothing is saved). */
int dm_shade_bwd(const dm_env_atlas* atlas_host, const dm_mat_cfg* mat_host, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, int n_views, const float* dcolor, long long dcolor_rs,
                 long long dcolor_cs, float* dfeat, long long dfeat_rs, long long dfeat_cs, dm_stream_t stream);

/* material_smoothness_grad (dreammat_material.py:110-123) on sigmoid(features) / sigmoid(features_jitter).
 * fwd ADDS the loss into *loss_out (device float, caller zeroes); bwd writes both feature gradients. */
int dm_matreg_fwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float* loss_out, dm_stream_t stream);
int dm_matreg_bwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float grad_scale, float* dfeat,
                  long long df_rs, long long df_cs, float* dfeatj, long long dj_rs, long long dj_cs,
                  dm_stream_t stream);

/* ---- ray queries for the Monte-Carlo shading branch (groundwork) ------------------- */
/* `_raytracing.create_raytracer(vertices, triangles)` (models/renderers/raytracing_renderer.py:31): HOST function,
 * host pointers.  nodes_out: 2*n_tri nodes of 32 B {bmin.xyz, a, bmax.xyz, b} (leaf when b > 0: triangles [a, a+b) of
 * tris_out; else children a and a+1); tris_out [n_tri*12] = {v0,0,e1,0,e2,0} in leaf order; order_out [n_tri] (may be
 * NULL) = original triangle id per leaf slot; *n_nodes_out = nodes used.  Copy nodes / tris to the device once. */
int dm_bvh_build(const float* v_pos_host, int32_t n_vert, const int32_t* tri_host, int32_t n_tri, void* nodes_out,
                 float* tris_out, int32_t* order_out, int32_t* n_nodes_out);
/* Optional 4-wide form of the same tree (the four child boxes stored in the parent, 128 B nodes: a quarter to a third of
 * the dependent fetches per ray).  HOST function; nodes4_out holds n_nodes2 entries, *n_nodes4_out = entries used. */
int dm_bvh_collapse4(const void* nodes2_host, int32_t n_nodes2, void* nodes4_out, int32_t* n_nodes4_out);
/* Uniform occupancy grid over the same triangles (csrc/grid_core.h): the occlusion query as a 3-D DDA whose empty-space
 * steps touch no memory (the kernels keep the bits in LDS).  HOST function.  tris12 = dm_bvh_build's tris_out; res = cells
 * along the longest axis of the mesh box (<= 0: chosen from n_tri, at most 96).  *blob_out is malloc'd (dm_host_free) and
 * holds six sections, each padded to 16 bytes: bits[n_words] u32 | sbase[ceil(n_words / 64)] u32 | off16[n_words] u16 |
 * dist4[a nibble per 2x2x2 block of cells: distance in blocks to the nearest occupied one] | occ_start[n_occ + 1] u32 | cell_tris[n_entries][12] f32 (the triangles of every occupied cell inline; slot 3 = the triangle's
 * number in leaf order as int bits).  `grid` receives the scalars, its pointers are left NULL for the caller to fill with
 * the device addresses of the sections. */
typedef struct dm_grid {
    float gmin[3]; float cell, inv_cell;
    int32_t dim[3];
    int32_t n_words, n_occ;
    long long n_entries;
    const uint32_t* bits; const uint32_t* sbase; const uint16_t* off16; const uint8_t* dist4; const uint32_t* occ_start;
    const float* cell_tris;
} dm_grid;
int dm_grid_build(const float* tris12, int32_t n_tri, int32_t res, dm_grid* grid, uint32_t** blob_out, int64_t* blob_words);
void dm_host_free(void* p);
/* dm_bvh_any_hit_rays through the grid (same answers: a boolean over the same triangle test); grid = host struct with
 * device pointers. */
int dm_grid_any_hit_rays(const dm_grid* grid_host, const float* origins, const float* dirs, long long n, float t_max,
                         unsigned char* hit, dm_stream_t stream);
/* `RayTracer.trace` as DreamMatMaterial.get_lights consumes it (dreammat_material.py:490-507,
 * raytracing_renderer.py:318-324): hit[i] = 1 iff ray origins[i] + t*dirs[i] meets the mesh for some 0 < t < t_max
 * (double-sided).  Device pointers; origins, dirs [n,3] fp32. */
int dm_bvh_any_hit_rays(const void* nodes, const float* tris, const float* origins, const float* dirs, long long n,
                        float t_max, unsigned char* hit, dm_stream_t stream);

/* Monte-Carlo ray-traced shading: DreamMatMaterial.forward, use_raytracing=True branch (dreammat_material.py:726-744)
 * + shade_raytracing (:615-677) + get_lights (:490-507) with the occlusion rays fused in.  Host struct, device data:
 * lights [n_env][light_h][light_w][3] fp32 lat-long radiance (get_envirmentlight_blender :452-470, nearest texel);
 * samples_* [n][2] = the (azimuth, elevation) Fibonacci tables of configure() (:389-398) in [0,1]^2.
 * mat->min/max_roughness carry the SQUARED range here (cfg.min/max_roughness_squre).  rand_* [N] = the per-point
 * azimuth rotations in [0,1) (torch.rand in the reference; NULL = none).  hit_bits [N][dm_mc_hit_words()] u32 is
 * written by fwd and read by bwd (one bit per sample direction, diffuse first).  n_diffuse + n_specular <= 1024.
 * Round-1 status: parity-tested on CPU and GPU against the reference, not yet optimised. */
typedef struct dm_mc_scene {
    const void* bvh_nodes; const float* bvh_tris;       /* device copies of dm_bvh_build's outputs */
    const float* lights; int n_env, light_h, light_w;
    const float* samples_diffuse; const float* samples_specular;
    int n_diffuse, n_specular;
    int geometry_ggx_smith;                             /* cfg.geometry_type: 0 = 'schlick', 1 = 'ggx_smith' */
    const void* bvh_nodes4;                             /* optional: device copy of dm_bvh_collapse4's output, else NULL */
    const dm_grid* grid;                                /* optional: occupancy grid with device pointers (then the occlusion
                                                         * queries walk the grid instead of the tree), else NULL */
} dm_mc_scene;
int dm_mc_hit_words(int n_diffuse, int n_specular);
int dm_mc_shade_fwd(const dm_mc_scene* scene_host, const dm_mat_cfg* mat_host, const float* pos, long long pos_rs,
                    long long pos_cs, const float* nrm, long long nrm_rs, long long nrm_cs, const float* view,
                    long long view_rs, long long view_cs, const float* feat, long long feat_rs, long long feat_cs,
                    const int32_t* pix_idx, const int32_t* env_of_view, const int32_t* n_dev, long long n_max, int HW,
                    const float* rand_diffuse, const float* rand_specular, uint32_t* hit_bits, float* color,
                    long long color_rs, long long color_cs, float* dbg_albedo, float* dbg_spec_light,
                    float* dbg_diff_light, float* dbg_spec_color, float* dbg_diff_color, float* dbg_metallic,
                    float* dbg_roughness, dm_stream_t stream);
int dm_mc_shade_bwd(const dm_mc_scene* scene_host, const dm_mat_cfg* mat_host, const float* pos, long long pos_rs,
                    long long pos_cs, const float* nrm, long long nrm_rs, long long nrm_cs, const float* view,
                    long long view_rs, long long view_cs, const float* feat, long long feat_rs, long long feat_cs,
                    const int32_t* pix_idx, const int32_t* env_of_view, const int32_t* n_dev, long long n_max, int HW,
                    const float* rand_diffuse, const float* rand_specular, const uint32_t* hit_bits,
                    const float* dcolor, long long dcolor_rs, long long dcolor_cs, float* dfeat, long long dfeat_rs,
                    long long dfeat_cs, dm_stream_t stream);

/* ---- attention ---------------------------------------------------------------------------- */
/* The QK^T.softmax.V of every transformer block diffusers runs inside ControlNetModel /
 * UNet2DConditionModel (models/guidance/dreammat_guidance.py:205-241, 261-282), bf16, MFMA.
 * q [B,Sq,Hh,D], k [B,Skv,Hh,D], out [B,Sq,Hh,D] by strides (d contiguous);
 * vt = V transposed [B,Hh,D,Skv_pad] by strides (kv contiguous, rows zero-padded to a multiple of 8).
 * D % 8 == 0, D <= 160; pointers 16 B aligned; strides multiples of 8 (out: 4) elements. */
int dm_attention_fwd_bf16(const void* q, const void* k, const void* vt, void* out, int B, int Hh, int Sq, int Skv,
                          int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss,
                          long long k_hs, long long vt_bs, long long vt_hs, long long vt_ds, long long o_bs,
                          long long o_ss, long long o_hs, float scale, dm_stream_t stream);
/* Differentiated attention (the trainable transformer blocks of the ControlNet training loop,
 * controlnet_train/diffusers_train_controlnet.py:858-915 -- there torch autograd runs diffusers' attention processors).
 * Forward: as dm_attention_fwd_bf16 but v is [B,Skv,Hh,D] with k's strides (not transposed), plus lse [B,Hh,Sq] fp32 =
 * rowmax + log2(rowsum) of the scaled scores (log2 domain).
 * Backward: q, out, dout, dq share (q_bs,q_ss,q_hs); k, v, dk, dv share (k_bs,k_ss,k_hs); v is [B,Skv,Hh,D] (not
 * transposed); delta [B,Hh,Sq] fp32 is scratch.  Nothing S x S is stored, no atomics (bit-reproducible).
 * D % 8 == 0, D <= 128 (DM_ERR_UNSUPPORTED above); pointers 16 B aligned, strides multiples of 8 elements. */
int dm_attention_fwd_lse_bf16(const void* q, const void* k, const void* v, void* out, float* lse, int B, int Hh, int Sq,
                              int Skv, int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs,
                              long long k_ss, long long k_hs, long long o_bs, long long o_ss, long long o_hs, float scale,
                              dm_stream_t stream);
int dm_attention_bwd_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                          float* delta, void* dq, void* dk, void* dv, int B, int Hh, int Sq, int Skv, int D,
                          long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss, long long k_hs,
                          float scale, dm_stream_t stream);
/* Kernel family for every later dm_attention_fwd_bf16 call of the process: "auto" (default: one wave per SIMD with 256
 * query rows per workgroup for 64-wide heads at S >= 1024, the 4 x 32-row LDS-DMA kernel otherwise, the register-staged
 * generic kernel for head sizes 40 / 80 / 160), or "w64" / "v3l" / "staged" to force one (shapes outside its domain fall
 * through to the next); NULL = back to the DREAMMAT_ATTN_KERNEL environment variable / "auto".  For A/B measurements and
 * the parity tests, which run every family.  DM_ERR_ARG for an unknown name. */
int dm_attention_select(const char* name);
/* name of the family in force ("auto", "w64", "v3l", "staged"): what reports should quote */
const char* dm_attention_selected(void);

/* ---- convolution ------------------------------------------------------------------------- */
/* 3x3 convolutions of UNet2DConditionModel / ControlNetModel / AutoencoderKL (the F.conv2d calls diffusers
 * makes under models/guidance/dreammat_guidance.py:205-292) as an implicit GEMM on MFMA: NHWC bf16,
 * x [B,Hin,Win,Cin], w [Cout,3,3,Cin] (tap-major, K contiguous), bias [Cout] or NULL, y [B,Hout,Wout,Cout].
 * pad_y/pad_x = leading zero padding, trailing padding implied by Hout/Wout.  Cin % 32 == 0, Cout % 64 == 0.
 * The data gradient (the VAE encoder is differentiated through) is the same call with w' = taps flipped and
 * Cin/Cout swapped. */
int dm_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                         int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, dm_stream_t stream);
/* Same convolution with the two adds diffusers' ResnetBlock2D performs around it folded into the epilogue:
 * rowbias [B,Cout] bf16 or NULL (`hidden_states + temb[:, :, None, None]`), residual [B,Hout,Wout,Cout] bf16 or
 * NULL (`input_tensor + hidden_states`); y = conv + bias + rowbias + residual, rounded once.  Cin % 64 == 0. */
int dm_conv3x3_nhwc_bf16_fused(const void* x, const void* w, const void* bias, const void* rowbias, const void* residual,
                               void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride,
                               int pad_y, int pad_x, dm_stream_t stream);

/* Weight gradient of a 3x3 / pad 1 convolution with stride 1 or 2 -- the trainable convolutions of the ControlNet training
 * loop (controlnet_train/diffusers_train_controlnet.py:858-915: torch autograd through F.conv2d there).
 * x [B,H,W,Cin], dy [B,Ho,Wo,Cout] bf16 NHWC with Ho = (H - 1) / stride + 1; part [splits][Cout][3][3][Cin] fp32 where
 * splits = dm_conv3x3_wgrad_splits(B, Ho, Wo, Cin, Cout): dW = the sum over the leading dimension (every split writes all of
 * its slice: no zero fill, no atomics).  The data gradient is dm_conv3x3_nhwc_bf16 on the flipped, channel-swapped weights.
 * Cin % 64 == 0, Cout % 64 == 0, Wo a power of two <= 64, Ho * Wo % 64 == 0 (DM_ERR_UNSUPPORTED otherwise; splits = 0). */
int dm_conv3x3_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout);
int dm_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* part, int B, int H, int W, int Cin, int Cout, int stride,
                               dm_stream_t stream);

/* 2 x 2 window "convolution" (ABI v10): y[b,yo,xo,n] = sum_{dy,dx in {0,1}} sum_c x[b, yo - pad_y + dy, xo - pad_x + dx, c] w[n][2 dy + dx][c],
 * zero outside the image; x [B,Hin,Win,Cin], w [Cout,4,Cin], y [B,Hout,Wout,Cout] NHWC bf16, Cin % 64 == 0, Cout % 256 == 0.
 * The sub-pixel form of the DATA GRADIENT of a stride-2 3x3 convolution (AutoencoderKL's Downsample2D,
 * `F.pad(x, (0,1,0,1))` + `conv(stride=2)`, differentiated at dreammat_guidance.py:284-292): the four output parities are four
 * blocks of Cin_x output channels over the gradient's own resolution (pad 1), see csrc/conv.hip; the caller interleaves them.
 * Also nearest-2x upsampling + 3x3 convolution (diffusers Upsample2D, UNet up blocks via dreammat_guidance.py:261-282) at the
 * SOURCE resolution: weights = sums of the 3x3 taps that fall on one source pixel, (h + 1) x (w + 1) output grid, parity (py, px)
 * of output (u, v) = channel block 2 py + px at grid (u + py, v + px).  bias [Cout] bf16 or NULL. */
int dm_conv2x2_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                         int Cout, int pad_y, int pad_x, dm_stream_t stream);

/* The same products (ABI v12) with the four Cout / 4-channel blocks of an output pixel stored as the four SUB-PIXELS of the 2x finer
 * tensor both callers want, instead of as 4 Cs channels of one pixel (hipops interleaved the blocks with a copy of the whole tensor:
 * 0.48 ms of the step for the three stride-2 data gradients of the VAE encoder, dreammat_guidance.py:284-292).  (Cout / 4) % 16 == 0.
 *   mode 1: y [B, 2 Hout, 2 Wout, Cout / 4]; block 2 py + px of pixel (u, v) -> (2u + py, 2v + px): the data gradient of a stride-2 conv.
 *   mode 2: y [B, 2 (Hout - 1), 2 (Wout - 1), Cout / 4]; block 2 py + px of grid position (u, v) -> (2u - py, 2v - px) where that lies
 *           inside: nearest-2x upsampling + 3x3 convolution evaluated on its (h + 1) x (w + 1) grid (Hout = h + 1, Wout = w + 1). */
int dm_conv2x2_subpixel_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout,
                                  int Wout, int Cout, int pad_y, int pad_x, int mode, dm_stream_t stream);

/* The few-channel stem convolutions of the same nets (ControlNetConditioningEmbedding 22->16, 16->16, 16->32 s2, 32->32,
 * 32->96 s2; conv_in 4->320): direct form, one thread per output pixel x 16 output channels, no im2col.  Same tensor
 * layouts as dm_conv3x3_nhwc_bf16; Cin in {4, 8, 16, 22, 32}, Cout % 16 == 0; act = 1 applies the SiLU that follows these
 * layers in ControlNetConditioningEmbedding.forward before the rounding to bf16. */
int dm_conv3x3_small_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                               int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act, dm_stream_t stream);
/* the same with a residual added before the rounding (ABI v10): residual [res_B, Hout, Wout, Cout] bf16 or NULL, image b takes
 * residual image b % res_B -- ControlNetModel.forward `sample = conv_in(sample) + controlnet_cond_embedding(cond)` with the
 * embedding of the B views shared by the three guidance branches (dreammat_guidance.py:205-241 -> diffusers ControlNetModel). */
int dm_conv3x3_small_res_nhwc_bf16(const void* x, const void* w, const void* bias, const void* residual, int res_B, void* y, int B,
                                   int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act,
                                   dm_stream_t stream);

/* Linear / 1x1-convolution layers of the same nets (diffusers Attention.to_q/to_k/to_out, FeedForward, Transformer2DModel
 * proj_in/proj_out, ResnetBlock2D.conv_shortcut: the F.linear / 1x1 F.conv2d calls under
 * models/guidance/dreammat_guidance.py:205-292) on the 1-tap instantiation of the same kernel:
 * y[M,N] = x[M,K] w[N,K]^T + bias[N] (+ residual[M,N]), bf16 row-major, fp32 accumulate, rounded once.
 * geglu != 0 fuses diffusers' GEGLU (`hidden, gate = proj(x).chunk(2); hidden * gelu(gate)`): w and bias rows interleaved in
 * blocks of 32 (32 value rows, then their 32 gate rows), y is [M, N/2]; value and gate stay in fp32 up to the product (one
 * rounding -- the unfused pair rounds both to 16 bits first) and gelu is the exact erf form, evaluated as gate * Phi(gate) with
 * an erfc polynomial (Abramowitz-Stegun 7.1.26, |error of Phi| < 3e-7).  M % 16 == 0, K % 64 == 0, N % 64 == 0 (geglu: N % 128 == 0, no residual). */
int dm_gemm_bf16_fused(const void* x, const void* w, const void* bias, const void* residual, void* y, long long M, int K,
                       int N, int geglu, dm_stream_t stream);
/* `batch` independent products in one launch (ABI v14): y[i] = x[i] w[i]^T, x [batch,M,K], w [batch,N,K], y [batch,M,N] contiguous,
 * no bias / residual; M % 256 == 0, K % 64 == 0, N % 64 == 0.  The per-image products of the VAE mid-block attention (one head of
 * 512, differentiated: dreammat_guidance.py:284-292 -> diffusers' Attention in AutoencoderKL's mid block), a few images per launch. */
int dm_gemm_bf16_batched(const void* x, const void* w, void* y, int batch, long long M, int K, int N, dm_stream_t stream);

/* ---- normalisation ----------------------------------------------------------------------- */
/* Forward only (frozen nets under no_grad): the coefficient kernel folded into the apply kernel -- 2 launches instead of
 * 3, same arithmetic; ws = scratch of dm_groupnorm_workspace_floats(B,C), nothing is kept in it. */
int dm_groupnorm_nhwc_infer(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW, int C,
                            float eps, int act, dm_stream_t stream);
/* ABI v13: GroupNorm with its APPLY pass folded into the consuming 3 x 3 convolution -- diffusers ResnetBlock2D.forward
 * `conv(nonlinearity(norm(x)))` behind models/guidance/dreammat_guidance.py:261-292.  dm_groupnorm_nhwc_stats runs the
 * statistics and coefficient kernels only and leaves ws as dm_groupnorm_nhwc_fwd does (it also serves dm_groupnorm_nhwc_bwd(_res)
 * of the same GroupNorm); dm_conv3x3_gn_nhwc_bf16_fused is dm_conv3x3_nhwc_bf16_fused (stride 1, pad 1) on the UN-normalised x with
 * act(x * A + S), rounded to 16 bits, formed in LDS (gn_coef = that ws; gn_act 0 | 1 = SiLU) -- same results as the two calls up
 * to the summation order of the taps.  dm_conv3x3_gn_ok: 1 when the halo-patch kernel serves the shape, else the fused entry
 * returns DM_ERR_UNSUPPORTED and the caller runs apply pass + convolution. */
int dm_groupnorm_nhwc_stats(const void* x, const void* gamma, const void* beta, float* ws, int B, int HW, int C, float eps,
                            dm_stream_t stream);
int dm_conv3x3_gn_ok(int B, int H, int W, int Cin, int Cout);
int dm_conv3x3_gn_nhwc_bf16_fused(const void* x, const float* gn_coef, int gn_act, const void* w, const void* bias, const void* rowbias,
                                  const void* residual, void* y, int B, int H, int W, int Cin, int Cout, dm_stream_t stream);
/* GroupNorm(32) [+ SiLU] of the ResnetBlock2D / Transformer2DModel / conv_norm_out layers of the same nets,
 * NHWC bf16: x,y [B,HW,C], gamma/beta [C] bf16.  ws: dm_groupnorm_workspace_floats(B,C) fp32, kept by the caller
 * between fwd and bwd.  act: 0 = none, 1 = SiLU.  bwd returns dx (the frozen nets of the SDS step need nothing else). */
size_t dm_groupnorm_workspace_floats(int B, int C);
int dm_groupnorm_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW,
                          int C, float eps, int act, dm_stream_t stream);
int dm_groupnorm_nhwc_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* ws,
                          int B, int HW, int C, float eps, int act, dm_stream_t stream);
/* the same with a second gradient of x added in the pass (ABI v10; dres [B,HW,C] bf16 or NULL): x of a ResnetBlock2D feeds norm1
 * and the skip connection -- diffusers ResnetBlock2D.forward `output = input_tensor + hidden_states` under AutoencoderKL's
 * encoder, differentiated at dreammat_guidance.py:284-292. */
int dm_groupnorm_nhwc_bwd_res(const void* x, const void* gamma, const void* beta, const void* dy, const void* dres, void* dx,
                              float* ws, int B, int HW, int C, float eps, int act, dm_stream_t stream);
/* GroupNorm with TRAINABLE affine parameters (the ControlNet copy in controlnet_train/diffusers_train_controlnet.py:858-915):
 * per-workgroup partials of dbeta / dgamma after dm_groupnorm_nhwc_fwd, cpart [dm_groupnorm_affine_rows(B,HW,C)][2][C] fp32:
 * dbeta = cpart[:, 0].sum(0), dgamma = cpart[:, 1].sum(0) (the caller's fixed-order sum: no atomics). */
int dm_groupnorm_affine_rows(int B, int HW, int C);
int dm_groupnorm_nhwc_bwd_affine(const void* x, const void* gamma, const void* beta, const void* dy, float* ws, float* cpart,
                                 int B, int HW, int C, float eps, int act, dm_stream_t stream);

/* LayerNorm and the GEGLU gate of diffusers' BasicTransformerBlock (norm1/2/3, ff.net.0) inside the same nets,
 * forward only (the diffusion nets run without autograd in SDS, dreammat_guidance.py:385-397).
 * layernorm: x,y [rows,C] bf16, gamma/beta [C] bf16, C % 8 == 0, C <= 2048.
 * geglu:     h [rows, 2*inner] bf16 = (value | gate), y [rows, inner] = value * gelu_erf(gate), inner % 8 == 0. */
int dm_layernorm_bf16(const void* x, const void* gamma, const void* beta, void* y, long long rows, int C, float eps,
                      dm_stream_t stream);
int dm_geglu_bf16(const void* h, void* y, long long rows, int inner, dm_stream_t stream);
/* ABI v13: a Linear / 1 x 1 layer of a FEW channels -- AutoencoderKL's quant_conv (8 -> 8) behind dreammat_guidance.py:284-292, the last
 * Linear of the differentiated VAE encoder that ran on ATen; y [M, N] = x [M, 8] w [N, 8]^T + bias, K = 8, N = 8 | 16, fp32 accumulate.
 * Its data gradient is the same call on w^T. */
int dm_linear_small_bf16(const void* x, const void* w, const void* bias, void* y, long long M, int K, int N, dm_stream_t stream);
/* Skip connection of a UNet up block with the ControlNet residual folded in (diffusers' `down_block_res_samples = [s + r]`
 * followed by `torch.cat([hidden, res_sample], dim=1)`, reached from dreammat_guidance.py:261-282): y[row] = x[row] | (s[row] +
 * r_scale * r[row]); x [rows,Cx], s/r [rows,Cs] (r may be NULL), y [rows,Cx+Cs], bf16, Cx % 8 == Cs % 8 == 0. */
int dm_cat_add_bf16(const void* x, const void* s, const void* r, void* y, long long rows, int Cx, int Cs, float r_scale,
                    dm_stream_t stream);

/* Row softmax over a materialised score matrix, forward and backward (ABI v10): the VAE encoder's mid-block attention (one
 * head of width 512, differentiated: dreammat_guidance.py:284-292 -> diffusers' Attention in AutoencoderKL's mid block) keeps
 * its two matrix products on the GEMM library and replaces the scale / cast / softmax / cast chain between them.
 * fwd: p = softmax(scale * s) per row; bwd: ds = scale * p * (dp - rowsum(p * dp)).  [rows, cols] bf16 row-contiguous, fp32
 * arithmetic, cols % 8 == 0, cols <= 16384; p may alias s, ds may alias dp.
 * ABI v14: the two products (and the four of the backward) run on dm_gemm_*_fused one image at a time over one reusable
 * [S, S] score buffer -- no [B, S, S] tensor, nothing left on the GEMM library; operands that must be contracted along their
 * rows are transposed by dm_transpose_*: src [batch, R, C] -> dst [batch, C, R], 16-bit elements, R % 64 == C % 64 == 0. */
int dm_softmax_rows_bf16(const void* s, void* p, long long rows, int cols, float scale, dm_stream_t stream);
int dm_softmax_rows_bwd_bf16(const void* p, const void* dp, void* ds, long long rows, int cols, float scale, dm_stream_t stream);
int dm_transpose_bf16(const void* src, void* dst, int batch, int R, int C, dm_stream_t stream);

/* ---- IEEE-half instantiations of the net kernels (ABI v11) --------------------------------- */
/* The reference's nets run in fp16 (`half_precision_weights`, threestudio/models/guidance/dreammat_guidance.py:56,92-94;
 * BASELINE.json configs[4] "fp16 UNet").  Every entry point of the sections above whose tensors are bf16 exists a second time for
 * `_Float16` tensors -- the same sources compiled with -DDM_F16 (csrc/dm_elem.h), v_mfma_f32_32x32x16_f16 in place of the bf16
 * instruction, identical tiles / layouts / arguments / error behaviour.  Names: f16 in place of bf16, or an _f16 suffix where the
 * bf16 name carries no dtype.  (The attention kernels that keep un-normalised probabilities in 16 bits bound them by half's
 * exponent range and send a workgroup whose rows leave it to their exact path.)  The training-only entry points
 * (dm_attention_bwd_bf16, dm_conv3x3_wgrad_nhwc_bf16, dm_groupnorm_nhwc_bwd_affine) stay bf16: row f-4 trains in bf16. */
int dm_attention_fwd_f16(const void* q, const void* k, const void* vt, void* out, int B, int Hh, int Sq, int Skv,
                          int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss,
                          long long k_hs, long long vt_bs, long long vt_hs, long long vt_ds, long long o_bs,
                          long long o_ss, long long o_hs, float scale, dm_stream_t stream);
int dm_attention_fwd_lse_f16(const void* q, const void* k, const void* v, void* out, float* lse, int B, int Hh, int Sq,
                              int Skv, int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs,
                              long long k_ss, long long k_hs, long long o_bs, long long o_ss, long long o_hs, float scale,
                              dm_stream_t stream);
int dm_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                         int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, dm_stream_t stream);
int dm_conv3x3_nhwc_f16_fused(const void* x, const void* w, const void* bias, const void* rowbias, const void* residual,
                               void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride,
                               int pad_y, int pad_x, dm_stream_t stream);
int dm_conv2x2_nhwc_f16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                         int Cout, int pad_y, int pad_x, dm_stream_t stream);
int dm_conv2x2_subpixel_nhwc_f16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout,
                                 int Wout, int Cout, int pad_y, int pad_x, int mode, dm_stream_t stream);
int dm_conv3x3_small_nhwc_f16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                               int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act, dm_stream_t stream);
int dm_conv3x3_small_res_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, int res_B, void* y, int B,
                                   int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act,
                                   dm_stream_t stream);
int dm_gemm_f16_fused(const void* x, const void* w, const void* bias, const void* residual, void* y, long long M, int K,
                       int N, int geglu, dm_stream_t stream);
int dm_gemm_f16_batched(const void* x, const void* w, void* y, int batch, long long M, int K, int N, dm_stream_t stream);
int dm_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, long long rows, int C, float eps,
                      dm_stream_t stream);
int dm_geglu_f16(const void* h, void* y, long long rows, int inner, dm_stream_t stream);
int dm_linear_small_f16(const void* x, const void* w, const void* bias, void* y, long long M, int K, int N, dm_stream_t stream);
int dm_cat_add_f16(const void* x, const void* s, const void* r, void* y, long long rows, int Cx, int Cs, float r_scale,
                    dm_stream_t stream);
int dm_softmax_rows_f16(const void* s, void* p, long long rows, int cols, float scale, dm_stream_t stream);
int dm_softmax_rows_bwd_f16(const void* p, const void* dp, void* ds, long long rows, int cols, float scale, dm_stream_t stream);
int dm_transpose_f16(const void* src, void* dst, int batch, int R, int C, dm_stream_t stream);
int dm_groupnorm_nhwc_fwd_f16(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW,
                          int C, float eps, int act, dm_stream_t stream);
int dm_groupnorm_nhwc_infer_f16(const void* x, const void* gamma, const void* beta, void* y, float* ws, int B, int HW, int C,
                            float eps, int act, dm_stream_t stream);
int dm_groupnorm_nhwc_stats_f16(const void* x, const void* gamma, const void* beta, float* ws, int B, int HW, int C, float eps,
                            dm_stream_t stream);
int dm_conv3x3_gn_nhwc_f16_fused(const void* x, const float* gn_coef, int gn_act, const void* w, const void* bias, const void* rowbias,
                                  const void* residual, void* y, int B, int H, int W, int Cin, int Cout, dm_stream_t stream);
int dm_groupnorm_nhwc_bwd_f16(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* ws,
                          int B, int HW, int C, float eps, int act, dm_stream_t stream);
int dm_groupnorm_nhwc_bwd_res_f16(const void* x, const void* gamma, const void* beta, const void* dy, const void* dres, void* dx,
                              float* ws, int B, int HW, int C, float eps, int act, dm_stream_t stream);

/* ---- MX-FP8 attention (ABI v11) ------------------------------------------------------------- */
/* BASELINE.json configs[4] "fp8 MFMA attention": softmax(q k^T scale) v of the 64-wide SD-2.1 heads with both matrix products
 * on v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 elements, one E8M0 scale per 32 elements of the contracted dimension).
 * q, k, vt, out and their strides: as dm_attention_fwd_bf16 with D = 64; elem_f16 = 0: bf16 tensors, 1: IEEE half.
 * Sq % 256 == 0, Skv % 64 == 0 (else DM_ERR_UNSUPPORTED: the caller keeps the 16-bit kernels).
 * ws: dm_attention_fp8_workspace_bytes(B, Hh, Sq, Skv) bytes of scratch (the quantised operands), 256-byte aligned. */
size_t dm_attention_fp8_workspace_bytes(int B, int Hh, int Sq, int Skv);
int dm_attention_fwd_fp8(const void* q, const void* k, const void* vt, void* out, int B, int Hh, int Sq, int Skv, int D,
                         long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss, long long k_hs,
                         long long vt_bs, long long vt_hs, long long vt_ds, long long o_bs, long long o_ss, long long o_hs,
                         float scale, int elem_f16, void* ws, size_t ws_bytes, dm_stream_t stream);

/* ---- optimiser ---------------------------------------------------------------------------- */
/* torch.optim.Adam step (configs/dreammat.yaml:110-115 via systems/utils.py:34-53) over one flat
 * fp32 buffer; grad is multiplied by grad_scale first (1/world after a sum all-reduce) and
 * optionally zeroed for the next step.  n % 4 == 0, 16 B aligned. */
int dm_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                 float beta1, float beta2, float eps, float grad_scale, int zero_grad, dm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DREAMMAT_HIP_H */
