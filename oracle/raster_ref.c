/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the three nvdiffrast ops the DreamMat renderer calls
 * (reference: threestudio/utils/rasterize.py:30-37 `dr.rasterize`, :49-56 `dr.antialias`,
 * :66-68 `dr.interpolate`; call sites threestudio/models/renderers/raytracing_renderer.py:124-199).
 *
 * nvdiffrast itself is an un-vendored git dependency (requirements.txt:15, unpinned HEAD) and
 * cannot be built here (CUDA only).  The rules below restate its published behaviour
 * ("upstream-documented, unverified here" => PARITY UNPINNED against real nvdiffrast):
 *
 *   rasterize : clip -> NDC -> window, pixel centres at (x+.5, y+.5), image row 0 = NDC y=-1,
 *               coverage decided on snapped fixed-point vertices (4 sub-pixel bits, as
 *               CudaRaster's CR_SUBPIXEL_LOG2) with a top-left tie-break, nearest z/w wins,
 *               output (u, v, z/w, tri_id+1) with u,v the perspective-correct barycentrics of
 *               vertices 0 and 1 computed from clip-space edge functions, 0 where empty.
 *   antialias : for every horizontally/vertically adjacent pixel pair with different ids take
 *               the nearer triangle, keep only its silhouette edges (opposite "wing" vertex on
 *               the same side), find the edge crossing the segment between the two pixel
 *               centres and blend by the crossing position.
 *
 * Every float expression is evaluated one IEEE-754 binary32 operation at a time, in the order
 * written (build with -ffp-contract=off, no -ffast-math), so that the HIP kernels -- which use
 * the same operation order -- are BIT-exact on (tri_id, u, v, z/w).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DM_SUBPIX 8      /* snapped units per half NDC-pixel: 16 units == 1 pixel */
#define DM_SNAP_MAX 4194304 /* 2^22 guard band */
#define DM_W_EPS 1e-8f

static inline int32_t snap_coord(float v) {
    /* v already multiplied by (dim*8); rint = round-half-even like v_rndne_f32 */
    float r = rintf(v);
    if (!(r > -(float)DM_SNAP_MAX)) r = -(float)DM_SNAP_MAX; /* also catches NaN */
    if (r > (float)DM_SNAP_MAX) r = (float)DM_SNAP_MAX;
    return (int32_t)r;
}

static inline int64_t floordiv16(int64_t a) { return a >> 4; }               /* arithmetic shift */
static inline int64_t ceildiv16(int64_t a) { return -((-a) >> 4); }

typedef struct {
    int32_t x[3], y[3];   /* snapped */
    int32_t px0, px1, py0, py1;
    int32_t sgn;          /* +1 / -1, 0 = culled */
} tri_setup_t;

static void setup_tri(const float *p0, const float *p1, const float *p2, int H, int W, tri_setup_t *s) {
    const float *p[3] = {p0, p1, p2};
    s->sgn = 0;
    for (int i = 0; i < 3; ++i) {
        float w = p[i][3];
        if (!(w > DM_W_EPS)) return;            /* no near-plane clipping: discard */
        float rw = 1.0f / w;
        float xn = p[i][0] * rw;
        float yn = p[i][1] * rw;
        s->x[i] = snap_coord(xn * (float)(W * DM_SUBPIX));
        s->y[i] = snap_coord(yn * (float)(H * DM_SUBPIX));
    }
    int64_t ax = (int64_t)s->x[1] - s->x[0], ay = (int64_t)s->y[1] - s->y[0];
    int64_t bx = (int64_t)s->x[2] - s->x[0], by = (int64_t)s->y[2] - s->y[0];
    int64_t area2 = ax * by - ay * bx;
    if (area2 == 0) return;
    int32_t minx = s->x[0], maxx = s->x[0], miny = s->y[0], maxy = s->y[0];
    for (int i = 1; i < 3; ++i) {
        if (s->x[i] < minx) minx = s->x[i];
        if (s->x[i] > maxx) maxx = s->x[i];
        if (s->y[i] < miny) miny = s->y[i];
        if (s->y[i] > maxy) maxy = s->y[i];
    }
    /* pixel centre cx = (2*px + 1 - W) * 8  =>  px = (cx + 8W - 8) / 16 */
    int64_t offx = (int64_t)DM_SUBPIX * W - DM_SUBPIX, offy = (int64_t)DM_SUBPIX * H - DM_SUBPIX;
    int64_t a = ceildiv16((int64_t)minx + offx), b = floordiv16((int64_t)maxx + offx);
    int64_t c = ceildiv16((int64_t)miny + offy), d = floordiv16((int64_t)maxy + offy);
    if (a < 0) a = 0;
    if (c < 0) c = 0;
    if (b > W - 1) b = W - 1;
    if (d > H - 1) d = H - 1;
    if (a > b || c > d) return;
    s->px0 = (int32_t)a; s->px1 = (int32_t)b; s->py0 = (int32_t)c; s->py1 = (int32_t)d;
    s->sgn = area2 > 0 ? 1 : -1;
}

/* edge a->b, orientation normalised by sgn; inside if E>0 or (E==0 and top-left) */
static inline int edge_inside(int32_t xa, int32_t ya, int32_t xb, int32_t yb, int32_t cx, int32_t cy, int32_t sgn) {
    int64_t dx = (int64_t)sgn * ((int64_t)xb - xa), dy = (int64_t)sgn * ((int64_t)yb - ya);
    int64_t e = dx * ((int64_t)cy - ya) - dy * ((int64_t)cx - xa);
    if (e > 0) return 1;
    if (e < 0) return 0;
    return (dy > 0) || (dy == 0 && dx < 0);
}

/* nvdiffrast's fragment shader arithmetic (perspective-correct barycentrics from clip space) */
static inline int frag_bary(const float *p0, const float *p1, const float *p2, int px, int py, int H, int W,
                            float *b0, float *b1, float *zw) {
    float rW = 1.0f / (float)W, rH = 1.0f / (float)H;
    float fx = (float)(2 * px + 1 - W) * rW;
    float fy = (float)(2 * py + 1 - H) * rH;
    float p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
    float p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
    float p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
    float a0 = p1x * p2y - p1y * p2x;
    float a1 = p2x * p0y - p2y * p0x;
    float a2 = p0x * p1y - p0y * p1x;
    float asum = (a0 + a1) + a2;
    float iw = 1.0f / asum;
    float z = (p0[2] * a0 + p1[2] * a1) + p2[2] * a2;
    float w = (p0[3] * a0 + p1[3] * a1) + p2[3] * a2;
    float q = z / w;
    if (!(q >= -1.0f && q <= 1.0f)) return 0; /* depth clip; also rejects NaN/inf */
    *b0 = a0 * iw; *b1 = a1 * iw; *zw = q;
    return 1;
}

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* pos [B,Nv,4] f32 clip space, tri [Nf,3] i32, out rast [B,H,W,4] f32 */
int dmo_rasterize(const float *pos, int B, int Nv, const int32_t *tri, int Nf, int H, int W, float *rast) {
    size_t npix = (size_t)H * W;
    float *best = (float *)malloc(npix * sizeof(float));
    if (!best) return -1;
    for (int b = 0; b < B; ++b) {
        const float *P = pos + (size_t)b * Nv * 4;
        float *R = rast + (size_t)b * npix * 4;
        memset(R, 0, npix * 4 * sizeof(float));
        for (size_t i = 0; i < npix; ++i) best[i] = 2.0f;
        for (int t = 0; t < Nf; ++t) {
            int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= Nv || i1 >= Nv || i2 >= Nv) continue;
            const float *p0 = P + 4 * (size_t)i0, *p1 = P + 4 * (size_t)i1, *p2 = P + 4 * (size_t)i2;
            tri_setup_t s;
            setup_tri(p0, p1, p2, H, W, &s);
            if (!s.sgn) continue;
            for (int py = s.py0; py <= s.py1; ++py) {
                int32_t cy = (2 * py + 1 - H) * DM_SUBPIX;
                for (int px = s.px0; px <= s.px1; ++px) {
                    int32_t cx = (2 * px + 1 - W) * DM_SUBPIX;
                    if (!edge_inside(s.x[1], s.y[1], s.x[2], s.y[2], cx, cy, s.sgn)) continue;
                    if (!edge_inside(s.x[2], s.y[2], s.x[0], s.y[0], cx, cy, s.sgn)) continue;
                    if (!edge_inside(s.x[0], s.y[0], s.x[1], s.y[1], cx, cy, s.sgn)) continue;
                    float b0, b1, zw;
                    if (!frag_bary(p0, p1, p2, px, py, H, W, &b0, &b1, &zw)) continue;
                    size_t pi = (size_t)py * W + px;
                    /* nearest wins; ties keep the lower triangle index (loop order) */
                    if (zw < best[pi]) {
                        best[pi] = zw;
                        R[4 * pi + 0] = clampf(b0, 0.f, 1.f);
                        R[4 * pi + 1] = clampf(b1, 0.f, 1.f);
                        R[4 * pi + 2] = zw;
                        R[4 * pi + 3] = (float)(t + 1);
                    }
                }
            }
        }
    }
    free(best);
    return 0;
}

/* ---------------------------------------------------------------- topology */
/* opp[t][i] = vertex opposite to edge i of triangle t in the other triangle sharing that edge,
 * or -1.  Edge 0 = (v1,v2), edge 1 = (v2,v0), edge 2 = (v0,v1).  Like nvdiffrast's edge-vertex
 * hash, at most the first two triangles (in index order) that use an edge are recorded. */
typedef struct { int32_t a, b, t, slot; } edge_rec_t;
static int edge_cmp(const void *x, const void *y) {
    const edge_rec_t *p = (const edge_rec_t *)x, *q = (const edge_rec_t *)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    if (p->t != q->t) return p->t < q->t ? -1 : 1;
    return p->slot - q->slot;
}
int dmo_build_topology(const int32_t *tri, int Nf, int32_t *opp) {
    edge_rec_t *e = (edge_rec_t *)malloc((size_t)Nf * 3 * sizeof(edge_rec_t));
    if (!e) return -1;
    for (int t = 0; t < Nf; ++t) {
        for (int i = 0; i < 3; ++i) {
            int32_t va = tri[3 * t + (i + 1) % 3], vb = tri[3 * t + (i + 2) % 3];
            edge_rec_t *r = &e[3 * t + i];
            r->a = va < vb ? va : vb; r->b = va < vb ? vb : va; r->t = t; r->slot = i;
            opp[3 * t + i] = -1;
        }
    }
    qsort(e, (size_t)Nf * 3, sizeof(edge_rec_t), edge_cmp);
    size_t n = (size_t)Nf * 3, i = 0;
    while (i < n) {
        size_t j = i + 1;
        while (j < n && e[j].a == e[i].a && e[j].b == e[i].b) ++j;
        if (j - i >= 2) {
            const edge_rec_t *r0 = &e[i], *r1 = &e[i + 1];
            int32_t o0 = tri[3 * r0->t + r0->slot], o1 = tri[3 * r1->t + r1->slot];
            opp[3 * r0->t + r0->slot] = o1;
            opp[3 * r1->t + r1->slot] = o0;
            /* further sharers (non-manifold) see the first recorded opposite that is not themselves */
            for (size_t k = i + 2; k < j; ++k) {
                int32_t self = tri[3 * e[k].t + e[k].slot];
                opp[3 * e[k].t + e[k].slot] = (o0 != self) ? o0 : o1;
            }
        }
        i = j;
    }
    free(e);
    return 0;
}

/* ---------------------------------------------------------------- antialias */
static inline int same_sign(float a, float b) {
    int32_t ia, ib;
    memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
    return (ia ^ ib) >= 0;
}
#define NEG_MAX (-3.402823466e38f)
/* n0/d0 > n1/d1 with -FLT_MAX sentinels */
static inline int rational_gt(float n0, float n1, float d0, float d1) {
    if (n0 == NEG_MAX) return 0;
    if (n1 == NEG_MAX) return 1;
    float l = n0 * d1, r = n1 * d0;
    return same_sign(d0, d1) ? (l > r) : (l < r);
}
static inline int max_idx3(float n0, float n1, float n2, float d0, float d1, float d2) {
    int g10 = rational_gt(n1, n0, d1, d0);
    int g20 = rational_gt(n2, n0, d2, d0);
    int g21 = rational_gt(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}

/* Analyse the pair (pixel0=(px,py), pixel1 = +1 in x if d==0, +1 in y if d==1).
 * Returns alpha: >0 => pixel0 += alpha*(c1-c0); <0 => pixel1 += alpha*(c1-c0); 0 => nothing. */
static float aa_pair(const float *P, const int32_t *tri, const int32_t *opp, const float *R,
                     int H, int W, int px, int py, int d) {
    size_t pix0 = (size_t)py * W + px;
    size_t pix1 = pix0 + (d ? (size_t)W : 1);
    int tri0 = (int)R[4 * pix0 + 3] - 1, tri1 = (int)R[4 * pix1 + 3] - 1;
    if (tri0 == tri1) return 0.f;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (R[4 * pix0 + 2] < R[4 * pix1 + 2]) ? tri0 : tri1;
    if (t == tri1) { px += 1 - d; py += d; }
    int vi0 = tri[3 * t], vi1 = tri[3 * t + 1], vi2 = tri[3 * t + 2];
    int op0 = opp[3 * t], op1 = opp[3 * t + 1], op2 = opp[3 * t + 2];
    if (op0 < 0) op0 = vi0;   /* no neighbour: the vertex itself => always silhouette */
    if (op1 < 0) op1 = vi1;
    if (op2 < 0) op2 = vi2;
    const float *p0 = P + 4 * (size_t)vi0, *p1 = P + 4 * (size_t)vi1, *p2 = P + 4 * (size_t)vi2;
    const float *o0 = P + 4 * (size_t)op0, *o1 = P + 4 * (size_t)op1, *o2 = P + 4 * (size_t)op2;
    float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    float fx = ((float)px + 0.5f) - xh, fy = ((float)py + 0.5f) - yh;
    float w0 = 1.0f / p0[3], w1 = 1.0f / p1[3], w2 = 1.0f / p2[3];
    float ow0 = 1.0f / o0[3], ow1 = 1.0f / o1[3], ow2 = 1.0f / o2[3];
    float x0 = (p0[0] * w0) * xh - fx, y0 = (p0[1] * w0) * yh - fy;
    float x1 = (p1[0] * w1) * xh - fx, y1 = (p1[1] * w1) * yh - fy;
    float x2 = (p2[0] * w2) * xh - fx, y2 = (p2[1] * w2) * yh - fy;
    float ox0 = (o0[0] * ow0) * xh - fx, oy0 = (o0[1] * ow0) * yh - fy;
    float ox1 = (o1[0] * ow1) * xh - fx, oy1 = (o1[1] * ow1) * yh - fy;
    float ox2 = (o2[0] * ow2) * xh - fx, oy2 = (o2[1] * ow2) * yh - fy;
    float bb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    float a0 = (x1 - ox0) * (y2 - oy0) - (x2 - ox0) * (y1 - oy0);
    float a1 = (x2 - ox1) * (y0 - oy1) - (x0 - ox1) * (y2 - oy1);
    float a2 = (x0 - ox2) * (y1 - oy2) - (x1 - ox2) * (y0 - oy2);
    int s0 = same_sign(a0, bb), s1 = same_sign(a1, bb), s2 = same_sign(a2, bb);
    if (!(s0 || s1 || s2)) return 0.f;
    if (d) { float tmp; tmp = x0; x0 = y0; y0 = tmp; tmp = x1; x1 = y1; y1 = tmp; tmp = x2; x2 = y2; y2 = tmp; }
    float dx0 = x2 - x1, dx1 = x0 - x2, dx2 = x1 - x0;
    float dy0 = y2 - y1, dy1 = y0 - y2, dy2 = y1 - y0;
    float ds = (t == tri0) ? 1.f : -1.f;
    float d0 = ds * (x1 * dy0 - y1 * dx0);
    float d1 = ds * (x2 * dy1 - y2 * dx1);
    float d2 = ds * (x0 * dy2 - y0 * dx2);
    if (same_sign(y1, y2)) { d0 = NEG_MAX; dy0 = 1.f; }
    if (same_sign(y2, y0)) { d1 = NEG_MAX; dy1 = 1.f; }
    if (same_sign(y0, y1)) { d2 = NEG_MAX; dy2 = 1.f; }
    int di = max_idx3(d0, d1, d2, dy0, dy1, dy2);
    float dc = NEG_MAX;
    if (di == 0 && s0 && fabsf(dy0) >= fabsf(dx0)) dc = d0 / dy0;
    if (di == 1 && s1 && fabsf(dy1) >= fabsf(dx1)) dc = d1 / dy1;
    if (di == 2 && s2 && fabsf(dy2) >= fabsf(dx2)) dc = d2 / dy2;
    const float eps = 0.0625f;
    if (dc > -eps && dc < 1.f + eps) {
        dc = clampf(dc, 0.f, 1.f);
        return ds * (0.5f - dc);
    }
    return 0.f;
}

/* plan [B,H,W,2]: alpha of the (x,x+1) pair and of the (y,y+1) pair anchored at each pixel */
int dmo_antialias_plan(const float *pos, int B, int Nv, const int32_t *tri, const int32_t *opp, int Nf,
                       const float *rast, int H, int W, float *plan) {
    (void)Nf;
    for (int b = 0; b < B; ++b) {
        const float *P = pos + (size_t)b * Nv * 4;
        const float *R = rast + (size_t)b * H * W * 4;
        float *A = plan + (size_t)b * H * W * 2;
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                size_t pi = (size_t)py * W + px;
                A[2 * pi + 0] = (px + 1 < W) ? aa_pair(P, tri, opp, R, H, W, px, py, 0) : 0.f;
                A[2 * pi + 1] = (py + 1 < H) ? aa_pair(P, tri, opp, R, H, W, px, py, 1) : 0.f;
            }
    }
    return 0;
}

/* out = color + sum over pairs; color/out [B,H,W,C] */
int dmo_antialias_apply(const float *color, const float *plan, int B, int H, int W, int C, float *out) {
    size_t n = (size_t)B * H * W * C;
    memcpy(out, color, n * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                size_t pi = ((size_t)b * H + py) * W + px;
                for (int d = 0; d < 2; ++d) {
                    float alpha = plan[2 * pi + d];
                    if (alpha == 0.f) continue;
                    size_t pj = pi + (d ? (size_t)W : 1);
                    size_t tgt = alpha > 0.f ? pi : pj;
                    for (int c = 0; c < C; ++c)
                        out[tgt * C + c] += alpha * (color[pj * C + c] - color[pi * C + c]);
                }
            }
    return 0;
}

/* gradient wrt color (the only gradient the DreamMat path uses: the mesh is fixed) */
int dmo_antialias_grad(const float *dout, const float *plan, int B, int H, int W, int C, float *dcolor) {
    size_t n = (size_t)B * H * W * C;
    memcpy(dcolor, dout, n * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                size_t pi = ((size_t)b * H + py) * W + px;
                for (int d = 0; d < 2; ++d) {
                    float alpha = plan[2 * pi + d];
                    if (alpha == 0.f) continue;
                    size_t pj = pi + (d ? (size_t)W : 1);
                    size_t tgt = alpha > 0.f ? pi : pj;
                    for (int c = 0; c < C; ++c) {
                        float g = alpha * dout[tgt * C + c];
                        dcolor[pj * C + c] += g;
                        dcolor[pi * C + c] -= g;
                    }
                }
            }
    return 0;
}

/* interpolate: attr [Nv,C] shared by all views; out [B,H,W,C] */
int dmo_interpolate(const float *attr, int Nv, int C, const int32_t *tri, const float *rast, int B, int H, int W,
                    float *out) {
    (void)Nv;
    size_t npix = (size_t)B * H * W;
    for (size_t pi = 0; pi < npix; ++pi) {
        int t = (int)rast[4 * pi + 3] - 1;
        float *o = out + pi * C;
        if (t < 0) { for (int c = 0; c < C; ++c) o[c] = 0.f; continue; }
        float b0 = rast[4 * pi], b1 = rast[4 * pi + 1];
        float b2 = (1.0f - b0) - b1;
        const float *a0 = attr + (size_t)tri[3 * t] * C, *a1 = attr + (size_t)tri[3 * t + 1] * C,
                    *a2 = attr + (size_t)tri[3 * t + 2] * C;
        for (int c = 0; c < C; ++c) o[c] = (b0 * a0[c] + b1 * a1[c]) + b2 * a2[c];
    }
    return 0;
}
