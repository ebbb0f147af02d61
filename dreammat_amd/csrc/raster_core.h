// Per-thread arithmetic of the rasterizer / antialias kernels, shared by the device kernels
// (raster.hip) and the host emulation used by CPU-side unit tests (tests/hostemu).
//
// Contract (documented nvdiffrast behaviour, see DESIGN.md "Rasterizer rules"):
//  * window coords from clip space, pixel centres at (x+.5,y+.5), row 0 = NDC y=-1
//  * coverage on vertices snapped to 1/16 pixel, integer edge functions, top-left tie-break
//  * nearest z/w wins, ties -> lower triangle index
//  * (u,v,z/w) from clip-space edge functions evaluated at the NDC pixel centre
// Every float expression is one IEEE binary32 op at a time in the written order (this file is
// compiled with -ffp-contract=off), which makes coverage ids AND barycentrics reproducible bit
// for bit by the CPU oracle (oracle/raster_ref.c restates the same rules independently).
#pragma once
#include "dm_common.h"

#pragma clang fp contract(off)

namespace dm {

constexpr int kSubpix = 8;            // snapped units per half pixel (16 units = 1 pixel)
constexpr int kSnapMax = 1 << 22;     // guard band for snapped coordinates
constexpr float kWEps = 1e-8f;
constexpr int kTile = 8;              // 8x8 pixel tiles, one wave per tile

struct TriSetup {
    int x[3], y[3];
    int px0, px1, py0, py1;
    int sgn;  // 0 => culled
};

DM_HD int snap(float v) {
    float r = rintf(v);
    if (!(r > -(float)kSnapMax)) r = -(float)kSnapMax;
    if (r > (float)kSnapMax) r = (float)kSnapMax;
    return (int)r;
}

DM_HD bool tri_setup(const float4& p0, const float4& p1, const float4& p2, int H, int W, TriSetup& s) {
    s.sgn = 0;
    const float4 p[3] = {p0, p1, p2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!(p[i].w > kWEps)) return false;
        float rw = 1.0f / p[i].w;
        float xn = p[i].x * rw;
        float yn = p[i].y * rw;
        s.x[i] = snap(xn * (float)(W * kSubpix));
        s.y[i] = snap(yn * (float)(H * kSubpix));
    }
    long long ax = (long long)s.x[1] - s.x[0], ay = (long long)s.y[1] - s.y[0];
    long long bx = (long long)s.x[2] - s.x[0], by = (long long)s.y[2] - s.y[0];
    long long area2 = ax * by - ay * bx;
    if (area2 == 0) return false;
    int minx = min(s.x[0], min(s.x[1], s.x[2])), maxx = max(s.x[0], max(s.x[1], s.x[2]));
    int miny = min(s.y[0], min(s.y[1], s.y[2])), maxy = max(s.y[0], max(s.y[1], s.y[2]));
    // pixel centre cx = (2 px + 1 - W) * 8  <=>  px = (cx + 8W - 8) / 16
    long long offx = (long long)kSubpix * W - kSubpix, offy = (long long)kSubpix * H - kSubpix;
    long long a = -((-((long long)minx + offx)) >> 4), b = ((long long)maxx + offx) >> 4;
    long long c = -((-((long long)miny + offy)) >> 4), d = ((long long)maxy + offy) >> 4;
    if (a < 0) a = 0;
    if (c < 0) c = 0;
    if (b > W - 1) b = W - 1;
    if (d > H - 1) d = H - 1;
    if (a > b || c > d) return false;
    s.px0 = (int)a; s.px1 = (int)b; s.py0 = (int)c; s.py1 = (int)d;
    s.sgn = area2 > 0 ? 1 : -1;
    return true;
}

// Edge a->b in the orientation-normalised triangle:  E(c) = dx*(cy-ya) - dy*(cx-xa),
// written as E(c) = A*cx + B*cy + C with the top-left bias folded in: inside <=> E' > 0.
struct EdgeEq {
    int A, B;          // |A|,|B| <= 2^23
    long long C;       // includes the +1 bias of a top-left edge
};

DM_HD EdgeEq edge_eq(int xa, int ya, int xb, int yb, int sgn) {
    long long dx = (long long)sgn * ((long long)xb - xa), dy = (long long)sgn * ((long long)yb - ya);
    EdgeEq e;
    e.A = (int)(-dy);
    e.B = (int)dx;
    e.C = dy * xa - dx * ya;
    bool tl = (dy > 0) || (dy == 0 && dx < 0);
    e.C += tl ? 1 : 0;
    return e;
}

DM_HD long long edge_eval(const EdgeEq& e, int cx, int cy) {
    return (long long)e.A * cx + (long long)e.B * cy + e.C;
}

// nvdiffrast fragment-shader arithmetic.  Returns false when the fragment is depth-clipped / degenerate.
DM_HD bool frag_bary(const float4& p0, const float4& p1, const float4& p2, int px, int py, int H, int W,
                     float& b0, float& b1, float& zw) {
    float rW = 1.0f / (float)W, rH = 1.0f / (float)H;
    float fx = (float)(2 * px + 1 - W) * rW;
    float fy = (float)(2 * py + 1 - H) * rH;
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    float a0 = p1x * p2y - p1y * p2x;
    float a1 = p2x * p0y - p2y * p0x;
    float a2 = p0x * p1y - p0y * p1x;
    float asum = (a0 + a1) + a2;
    float iw = 1.0f / asum;
    float z = (p0.z * a0 + p1.z * a1) + p2.z * a2;
    float w = (p0.w * a0 + p1.w * a1) + p2.w * a2;
    float q = z / w;
    if (!(q >= -1.0f && q <= 1.0f)) return false;
    b0 = a0 * iw;
    b1 = a1 * iw;
    zw = q;
    return true;
}

DM_HD float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

// ---------------------------------------------------------------------------- antialias
DM_HD bool same_sign(float a, float b) {
    return ((__builtin_bit_cast(int, a) ^ __builtin_bit_cast(int, b)) >= 0);
}
constexpr float kNegMax = -3.402823466e38f;

DM_HD bool rational_gt(float n0, float n1, float d0, float d1) {
    if (n0 == kNegMax) return false;
    if (n1 == kNegMax) return true;
    float l = n0 * d1, r = n1 * d0;
    return same_sign(d0, d1) ? (l > r) : (l < r);
}
DM_HD int max_idx3(float n0, float n1, float n2, float d0, float d1, float d2) {
    bool g10 = rational_gt(n1, n0, d1, d0);
    bool g20 = rational_gt(n2, n0, d2, d0);
    bool g21 = rational_gt(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}

// Pair (pixel0 = (px,py), pixel1 = +x if d==0 else +y).  r0/r1 = their rast texels.
// alpha > 0: pixel0 += alpha*(c1-c0);  alpha < 0: pixel1 += alpha*(c1-c0);  0: nothing.
DM_HD float aa_pair(const float4* __restrict__ P, const int* __restrict__ tri, const int* __restrict__ opp,
                    float4 r0, float4 r1, int H, int W, int px, int py, int d) {
    int tri0 = (int)r0.w - 1, tri1 = (int)r1.w - 1;
    if (tri0 == tri1) return 0.f;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (r0.z < r1.z) ? tri0 : tri1;
    if (t == tri1) { px += 1 - d; py += d; }
    int vi0 = tri[3 * t], vi1 = tri[3 * t + 1], vi2 = tri[3 * t + 2];
    int op0 = opp[3 * t], op1 = opp[3 * t + 1], op2 = opp[3 * t + 2];
    if (op0 < 0) op0 = vi0;
    if (op1 < 0) op1 = vi1;
    if (op2 < 0) op2 = vi2;
    float4 p0 = P[vi0], p1 = P[vi1], p2 = P[vi2], o0 = P[op0], o1 = P[op1], o2 = P[op2];
    float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    float fx = ((float)px + 0.5f) - xh, fy = ((float)py + 0.5f) - yh;
    float w0 = 1.0f / p0.w, w1 = 1.0f / p1.w, w2 = 1.0f / p2.w;
    float ow0 = 1.0f / o0.w, ow1 = 1.0f / o1.w, ow2 = 1.0f / o2.w;
    float x0 = (p0.x * w0) * xh - fx, y0 = (p0.y * w0) * yh - fy;
    float x1 = (p1.x * w1) * xh - fx, y1 = (p1.y * w1) * yh - fy;
    float x2 = (p2.x * w2) * xh - fx, y2 = (p2.y * w2) * yh - fy;
    float ox0 = (o0.x * ow0) * xh - fx, oy0 = (o0.y * ow0) * yh - fy;
    float ox1 = (o1.x * ow1) * xh - fx, oy1 = (o1.y * ow1) * yh - fy;
    float ox2 = (o2.x * ow2) * xh - fx, oy2 = (o2.y * ow2) * yh - fy;
    float bb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    float a0 = (x1 - ox0) * (y2 - oy0) - (x2 - ox0) * (y1 - oy0);
    float a1 = (x2 - ox1) * (y0 - oy1) - (x0 - ox1) * (y2 - oy1);
    float a2 = (x0 - ox2) * (y1 - oy2) - (x1 - ox2) * (y0 - oy2);
    bool s0 = same_sign(a0, bb), s1 = same_sign(a1, bb), s2 = same_sign(a2, bb);
    if (!(s0 || s1 || s2)) return 0.f;
    if (d) {
        float tmp;
        tmp = x0; x0 = y0; y0 = tmp;
        tmp = x1; x1 = y1; y1 = tmp;
        tmp = x2; x2 = y2; y2 = tmp;
    }
    float dx0 = x2 - x1, dx1 = x0 - x2, dx2 = x1 - x0;
    float dy0 = y2 - y1, dy1 = y0 - y2, dy2 = y1 - y0;
    float ds = (t == tri0) ? 1.f : -1.f;
    float d0 = ds * (x1 * dy0 - y1 * dx0);
    float d1 = ds * (x2 * dy1 - y2 * dx1);
    float d2 = ds * (x0 * dy2 - y0 * dx2);
    if (same_sign(y1, y2)) { d0 = kNegMax; dy0 = 1.f; }
    if (same_sign(y2, y0)) { d1 = kNegMax; dy1 = 1.f; }
    if (same_sign(y0, y1)) { d2 = kNegMax; dy2 = 1.f; }
    int di = max_idx3(d0, d1, d2, dy0, dy1, dy2);
    float dc = kNegMax;
    if (di == 0 && s0 && fabsf(dy0) >= fabsf(dx0)) dc = d0 / dy0;
    if (di == 1 && s1 && fabsf(dy1) >= fabsf(dx1)) dc = d1 / dy1;
    if (di == 2 && s2 && fabsf(dy2) >= fabsf(dx2)) dc = d2 / dy2;
    const float eps = 0.0625f;
    if (dc > -eps && dc < 1.f + eps) {
        dc = clamp01(dc);
        return ds * (0.5f - dc);
    }
    return 0.f;
}

}  // namespace dm
