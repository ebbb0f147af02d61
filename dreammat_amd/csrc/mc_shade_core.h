// Per-pixel arithmetic of the Monte-Carlo ray-traced shading branch (the reference's DEFAULT
// material path).  Follows threestudio/models/materials/dreammat_material.py:
//   :726-744 forward (use_raytracing branch: sigmoid activation, albedo / metallic / roughness ranges, roughness is
//            alpha "already squared"),  :615-677 shade_raytracing,  :554-596 sample_{diffuse,specular}_directions,
//   :543-553 get_orthogonal_directions, :509-541 fresnel / geometry terms, :599-604 distribution_ggx,
//   :490-507 get_lights (occluded => 0, else nearest lat-long texel :452-470), utils/ops.py:83-88 lin2srgb.
// One pixel = nd cosine-weighted + ns GGX directions; every direction contributes to the specular estimator with the
// one-sample-MIS pdf of ITS OWN sampler, the first nd also to the diffuse estimator.
//
// Gradients: the radiance along a direction (ray hit, nearest texel) is piecewise constant, so the reference's autograd
// graph reaches the material only through the BRDF terms, the pdfs and the roughness-dependent GGX sample directions.
// albedo and metallic enter linearly through F0:  spec_c = F0_c * A_c + B_c  with
//   A_c = mean_s (1 - k_s) L_c(s) w_s,   B_c = mean_s k_s L_c(s) w_s,   k = clamp(1 - HoV, 0, 1)^5,
// so only roughness needs a derivative THROUGH the sample loop: it is carried in forward mode (value, d/d alpha) by
// the same templated code that computes the values (S = float: forward pass, S = Dual: backward pass).
// Host + device: tests/hostemu runs this file on the CPU against oracle/mc_shading.py (which is pinned to the
// reference's own method bodies by tests/golden/mc_shading.npz).
#pragma once
#include "bvh_core.h"
#include "dm_common.h"
#include "grid_core.h"

namespace dm {
namespace mc {

constexpr float kPi = 3.14159265358979323846f;
constexpr int kMaxSamples = 1024;                  // hit bits per pixel: kMaxSamples / 32 words

// ---- forward-mode scalar: value + derivative with respect to alpha
struct Dual { float v, d; };
DM_HD Dual mk(float v, float d = 0.f) { Dual r; r.v = v; r.d = d; return r; }
DM_HD float val(float x) { return x; }
DM_HD float val(Dual x) { return x.v; }
DM_HD float der(float) { return 0.f; }
DM_HD float der(Dual x) { return x.d; }
DM_HD float cst(float, float v) { return v; }                     // constant of the same scalar type as the tag
DM_HD Dual cst(Dual, float v) { return mk(v, 0.f); }
DM_HD float seed(float, float v) { return v; }                    // the differentiation variable itself
DM_HD Dual seed(Dual, float v) { return mk(v, 1.f); }
DM_HD Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.d + b.d); }
DM_HD Dual operator+(Dual a, float b) { return mk(a.v + b, a.d); }
DM_HD Dual operator+(float a, Dual b) { return mk(a + b.v, b.d); }
DM_HD Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.d - b.d); }
DM_HD Dual operator-(Dual a, float b) { return mk(a.v - b, a.d); }
DM_HD Dual operator-(float a, Dual b) { return mk(a - b.v, -b.d); }
DM_HD Dual operator*(Dual a, Dual b) { return mk(a.v * b.v, a.d * b.v + a.v * b.d); }
DM_HD Dual operator*(Dual a, float b) { return mk(a.v * b, a.d * b); }
DM_HD Dual operator*(float a, Dual b) { return mk(a * b.v, a * b.d); }
DM_HD Dual operator/(Dual a, Dual b) { float q = a.v / b.v; return mk(q, (a.d - q * b.d) / b.v); }
DM_HD Dual operator/(float a, Dual b) { float q = a / b.v; return mk(q, -q * b.d / b.v); }
DM_HD Dual operator/(Dual a, float b) { return mk(a.v / b, a.d / b); }
DM_HD float sqrt_(float x) { return sqrtf(x); }
DM_HD Dual sqrt_(Dual x) { float s = sqrtf(x.v); return mk(s, 0.5f * x.d / s); }
DM_HD float sat_(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
DM_HD Dual sat_(Dual x) { return (x.v < 0.f) ? mk(0.f) : (x.v > 1.f) ? mk(1.f) : x; }   // torch.clamp: grad where 0 <= x <= 1
DM_HD float pow5_(float x) { float x2 = x * x; return x2 * x2 * x; }
DM_HD Dual pow5_(Dual x) { float x2 = x.v * x.v, x4 = x2 * x2; return mk(x4 * x.v, 5.f * x4 * x.d); }

template <class S> struct V3 { S x, y, z; };
template <class S> DM_HD S dotf(const V3<S>& a, const float* b) { return a.x * b[0] + a.y * b[1] + a.z * b[2]; }
// F.normalize(v + d): divide by max(norm, 1e-12)
template <class S> DM_HD V3<S> half_vector(const V3<S>& d, const float* v) {
    V3<S> h;
    h.x = d.x + v[0]; h.y = d.y + v[1]; h.z = d.z + v[2];
    S hn = sqrt_(h.x * h.x + h.y * h.y + h.z * h.z);
    if (val(hn) < 1e-12f) hn = cst(hn, 1e-12f);
    h.x = h.x / hn; h.y = h.y / hn; h.z = h.z / hn;
    return h;
}
template <class S> DM_HD S ggx_d(S NoH, S a2) {                   // distribution_ggx (:599-604)
    S den = NoH * NoH * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * (den * den) + 1e-4f);
}

struct McCfg {
    float min_metallic, max_metallic, min_rough_sq, max_rough_sq;    // dreammat_material.py:352-356
    int n_diffuse, n_specular;                                       // cfg.diffuse_sample_num / specular_sample_num
    int geometry_ggx_smith;                                          // cfg.geometry_type: 0 'schlick', 1 'ggx_smith'
};

struct McScene {
    const DmBvhNode* nodes; const float* tris;                       // dm_bvh_build outputs
    const DmBvhNode4* nodes4;                                        // optional 4-wide form (dm_bvh_collapse4), else null
    const DmGrid* grid; DmGridTables grid_tb;                        // optional occupancy grid (dm_grid_build) + where its tables are (LDS / global)
    const float* light; int light_h, light_w;                        // lat-long radiance [h][w][3] of this pixel's env
    const float* samples_d; const float* samples_s;                  // [n][2] (azimuth, elevation) tables in [0,1]^2
};

// get_orthogonal_directions (:543-553): the longer of (y,-x,0) and (-z,0,x), normalised; y = z cross x
DM_HD void ortho_frame(const float* z, float* x, float* y) {
    const float n0 = sqrtf(z[1] * z[1] + z[0] * z[0]), n1 = sqrtf(z[2] * z[2] + z[0] * z[0]);
    float o[3];
    if (n0 > n1) { o[0] = z[1]; o[1] = -z[0]; o[2] = 0.f; } else { o[0] = -z[2]; o[1] = 0.f; o[2] = z[0]; }
    const float inv = 1.0f / fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
    x[0] = o[0] * inv; x[1] = o[1] * inv; x[2] = o[2] * inv;
    y[0] = z[1] * x[2] - z[2] * x[1]; y[1] = z[2] * x[0] - z[0] * x[2]; y[2] = z[0] * x[1] - z[1] * x[0];
}

// get_envirmentlight_blender (:452-470): nearest texel, z-up lat-long
DM_HD void env_lookup(const McScene& sc, float dx, float dy, float dz, float* rgb) {
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    const float theta = acosf(z);
    float phi = fmodf(atan2f(y, x), 2.f * kPi);
    if (phi < 0.f) phi += 2.f * kPi;                                 // python % takes the divisor's sign
    const float u = -phi / (2.f * kPi) + 0.5f, v = theta / kPi;
    float px = fmodf(u * sc.light_w, (float)sc.light_w), py = fmodf(v * sc.light_h, (float)sc.light_h);
    if (px < 0.f) px += sc.light_w;
    if (py < 0.f) py += sc.light_h;
    int ix = (int)px, iy = (int)py;
    ix = ix >= sc.light_w ? sc.light_w - 1 : ix;
    iy = iy >= sc.light_h ? sc.light_h - 1 : iy;
    const float* t = sc.light + 3 * ((size_t)iy * sc.light_w + ix);
    rgb[0] = t[0]; rgb[1] = t[1]; rgb[2] = t[2];
}

// get_lights (:490-507): the ray starts 1e-5 along the direction; miss <=> no hit closer than 10
DM_HD bool occluded(const McScene& sc, const float* p, float dx, float dy, float dz) {
    const float eps = 1e-5f;
    if (sc.grid) return dm_grid_any_hit(*sc.grid, sc.grid_tb, p[0] + dx * eps, p[1] + dy * eps, p[2] + dz * eps, dx, dy, dz, 10.0f);
    if (sc.nodes4) return dm_bvh4_any_hit(sc.nodes4, sc.tris, p[0] + dx * eps, p[1] + dy * eps, p[2] + dz * eps, dx, dy, dz, 10.0f);
    return dm_bvh_any_hit(sc.nodes, sc.tris, p[0] + dx * eps, p[1] + dy * eps, p[2] + dz * eps, dx, dy, dz, 10.0f);
}

template <class S> struct McAcc {
    S A[3], B[3];          // sums over all sn samples of (1-k) L w and k L w
    float Ld[3], Ls[3];    // sums of L over the diffuse / specular samples
};

// one sample direction d with sampler pdf `prob`: BRDF weight w = D G / (4 NoV pdf + 1e-5), Fresnel split into A / B
template <class S>
DM_HD void add_sample(const McCfg& cfg, const float* n, const float* v, float NoV, S a, const V3<S>& d, const V3<S>& h, S prob,
                      const float* L, McAcc<S>& acc) {
    S HoV = sat_(dotf(h, v));
    S k = pow5_(sat_(1.0f - HoV));
    S NoL = sat_(dotf(d, n));
    S NoH = sat_(dotf(h, n));
    S a2 = a * a;
    S D = ggx_d(NoH, a2);
    S G;
    if (cfg.geometry_ggx_smith) {                                   // geometry_ggx_smith_correlated (:532-541)
        const float cv = NoV * NoV;
        S cl = NoL * NoL;
        S fv = 0.5f * sqrt_(1.0f + a2 * ((1.0f - cv) / (cv + 1e-7f))) - 0.5f;
        S fl = 0.5f * sqrt_(1.0f + a2 * ((1.0f - cl) / (cl + 1e-7f))) - 0.5f;
        G = 1.0f / (1.0f + fv + fl);
    } else {                                                        // geometry_schlick (:519-530), k = alpha / 2
        S kk = a / 2.0f;
        S gv = NoV / (NoV * (1.0f - kk) + kk + 1e-5f);
        S gl = NoL / (NoL * (1.0f - kk) + kk + 1e-5f);
        G = gv * gl;
    }
    S w = D * G / (4.0f * NoV * prob + 1e-5f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S lw = L[c] * w;
        acc.A[c] = acc.A[c] + (1.0f - k) * lw;
        acc.B[c] = acc.B[c] + k * lw;
    }
}

struct McPixel {
    float m[5];                                   // sigmoid(features)
    float albedo[3], metallic, alpha;
    float Ld_mean[3], Ls_mean[3];                 // mean radiance of the diffuse / specular sample sets
    float diffuse[3], specular[3], pre[3];        // colours before lin2srgb
    float dpre_dalbedo[3], dpre_dmetallic[3], dpre_dalpha[3];   // per colour channel (dalpha only when S = Dual)
};

// ---- a pixel in three pieces, so that the samples can be walked serially by one thread (shade_pixel) or spread over
// the lanes of a wave (mc_shade.hip, k_mc_shade_wave): setup (material activation, frames) / one sample / finish.
struct McFrame {
    float n[3], v[3], p[3], r[3];                 // normal, view direction, surface point, mirror direction
    float xd[3], yd[3], xs[3], ys[3];             // tangent frames of n and r
    float NoV, rand_d, rand_s;
};

DM_HD void pixel_setup(const McCfg& cfg, const float* p, const float* n, const float* v, const float* feat, float rand_d,
                       float rand_s, McFrame& fr, McPixel& out) {
#pragma unroll
    for (int k = 0; k < 5; ++k) out.m[k] = 1.0f / (1.0f + expf(-feat[k]));
#pragma unroll
    for (int c = 0; c < 3; ++c) out.albedo[c] = sat_(out.m[c]);
    out.metallic = out.m[3] * (cfg.max_metallic - cfg.min_metallic) + cfg.min_metallic;
    out.alpha = out.m[4] * (cfg.max_rough_sq - cfg.min_rough_sq) + cfg.min_rough_sq;
    const float ndv = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    fr.NoV = sat_(ndv);
#pragma unroll
    for (int c = 0; c < 3; ++c) { fr.n[c] = n[c]; fr.v[c] = v[c]; fr.p[c] = p[c]; fr.r[c] = ndv * n[c] * 2.f - v[c]; }
    ortho_frame(fr.n, fr.xd, fr.yd);
    ortho_frame(fr.r, fr.xs, fr.ys);
    fr.rand_d = rand_d; fr.rand_s = rand_s;
}

template <class S> DM_HD void acc_clear(S a, McAcc<S>& acc) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc.A[c] = cst(a, 0.f); acc.B[c] = cst(a, 0.f); acc.Ld[c] = 0.f; acc.Ls[c] = 0.f; }
}

// direction of sample s of the pixel: s < nd = cosine-weighted direction s around the normal (:554-573), else GGX direction
// s - nd around the mirror direction (:575-596; it depends on alpha, hence S)
template <class S>
DM_HD V3<S> sample_dir(const McCfg& cfg, const McScene& sc, const McFrame& fr, S a, int s) {
    const int nd = cfg.n_diffuse;
    V3<S> d;
    if (s < nd) {
        float az = sc.samples_d[2 * s] * kPi * 2.f;
        const float el = sc.samples_d[2 * s + 1];
        if (fr.rand_d >= 0.f) az = fmodf(az + fr.rand_d * kPi * 2.f, 2.f * kPi);
        const float el_sqrt = sqrtf(el + 1e-7f), cz = sqrtf(1.f - el + 1e-7f);
        const float cx = el_sqrt * cosf(az), cy = el_sqrt * sinf(az);
        d.x = cst(a, cx * fr.xd[0] + cy * fr.yd[0] + cz * fr.n[0]);
        d.y = cst(a, cx * fr.xd[1] + cy * fr.yd[1] + cz * fr.n[1]);
        d.z = cst(a, cx * fr.xd[2] + cy * fr.yd[2] + cz * fr.n[2]);
    } else {
        const int j = s - nd;
        float phi = kPi * 2.f * sc.samples_s[2 * j];
        const float el = sc.samples_s[2 * j + 1];
        if (fr.rand_s >= 0.f) phi = fmodf(phi + fr.rand_s * kPi * 2.f, 2.f * kPi);
        S cos_t = sqrt_((1.0f - el + 1e-6f) / (1.0f + (a * a - 1.0f) * el + 1e-6f) + 1e-6f);
        S sin_t = sqrt_(1.0f - cos_t * cos_t + 1e-6f);
        const float cph = cosf(phi), sph = sinf(phi);
        d.x = (cph * sin_t) * fr.xs[0] + (sph * sin_t) * fr.ys[0] + cos_t * fr.r[0];
        d.y = (cph * sin_t) * fr.xs[1] + (sph * sin_t) * fr.ys[1] + cos_t * fr.r[1];
        d.z = (cph * sin_t) * fr.xs[2] + (sph * sin_t) * fr.ys[2] + cos_t * fr.r[2];
    }
    return d;
}

// sample s of the pixel: pdf NoL/pi * nd/sn (cosine-weighted) or D NoH / (4 VoH + 1e-5) * ns/sn (GGX).
// TRACE: shoot the occlusion ray and return the result in `hit`; else take `hit` as given (recorded by the forward, or found
// by the wave-cooperative traversal of the one-wave-per-pixel kernel).
template <class S, bool TRACE>
DM_HD void sample_eval(const McCfg& cfg, const McScene& sc, const McFrame& fr, S a, int s, bool& hit, McAcc<S>& acc) {
    const int nd = cfg.n_diffuse, ns = cfg.n_specular, sn = nd + ns;
    const V3<S> d = sample_dir<S>(cfg, sc, fr, a, s);
    if (TRACE) hit = occluded(sc, fr.p, val(d.x), val(d.y), val(d.z));
    float L[3] = {0.f, 0.f, 0.f};
    if (!hit) env_lookup(sc, val(d.x), val(d.y), val(d.z), L);
    if (s < nd) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc.Ld[c] += L[c];
        const float NoL_d = sat_(val(d.x) * fr.n[0] + val(d.y) * fr.n[1] + val(d.z) * fr.n[2]);
        add_sample<S>(cfg, fr.n, fr.v, fr.NoV, a, d, half_vector(d, fr.v), cst(a, NoL_d / kPi * ((float)nd / (float)sn)), L, acc);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc.Ls[c] += L[c];
        const V3<S> h = half_vector(d, fr.v);
        S NoH = sat_(dotf(h, fr.n)), VoH = sat_(dotf(h, fr.v));
        S prob = ggx_d(NoH, a * a) * NoH / (4.0f * VoH + 1e-5f) * ((float)ns / (float)sn);
        add_sample<S>(cfg, fr.n, fr.v, fr.NoV, a, d, h, prob, L, acc);
    }
}

// sums over all samples -> colours and their sensitivities (out.m / albedo / metallic / alpha set by pixel_setup)
template <class S> DM_HD void pixel_finish(const McCfg& cfg, const McAcc<S>& acc, McPixel& out) {
    const int nd = cfg.n_diffuse, ns = cfg.n_specular, sn = nd + ns;
    const float inv_sn = 1.0f / (float)sn, inv_nd = 1.0f / (float)nd, inv_ns = 1.0f / (float)ns;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float F0 = 0.04f * (1.f - out.metallic) + out.metallic * out.albedo[c];
        const float A = val(acc.A[c]) * inv_sn, B = val(acc.B[c]) * inv_sn;
        out.Ld_mean[c] = acc.Ld[c] * inv_nd;
        out.Ls_mean[c] = acc.Ls[c] * inv_ns;
        out.diffuse[c] = out.albedo[c] * out.Ld_mean[c];
        out.specular[c] = F0 * A + B;
        out.pre[c] = out.diffuse[c] + out.specular[c];
        out.dpre_dalbedo[c] = out.Ld_mean[c] + out.metallic * A;
        out.dpre_dmetallic[c] = A * (out.albedo[c] - 0.04f);
        out.dpre_dalpha[c] = (F0 * der(acc.A[c]) + der(acc.B[c])) * inv_sn;
    }
}

// One thread walks all samples.  S = float: values.  S = Dual: additionally d(pre)/d(alpha).
// hit_bits: one bit per sample (diffuse first, kMaxSamples/32 words, zeroed by the caller when WRITE_HITS).
// rand_d / rand_s: the per-point azimuth rotations in [0,1) (torch.rand in the reference); < 0 = no rotation.
template <class S, bool WRITE_HITS>
DM_HD void shade_pixel(const McCfg& cfg, const McScene& sc, const float* p, const float* n, const float* v, const float* feat,
                       float rand_d, float rand_s, unsigned* hit_bits, McPixel& out) {
    McFrame fr;
    pixel_setup(cfg, p, n, v, feat, rand_d, rand_s, fr, out);
    const S a = seed(S(), out.alpha);
    McAcc<S> acc;
    acc_clear(a, acc);
    const int sn = cfg.n_diffuse + cfg.n_specular;
    for (int s = 0; s < sn; ++s) {
        bool hit = WRITE_HITS ? false : ((hit_bits[s >> 5] >> (s & 31)) & 1u);
        sample_eval<S, WRITE_HITS>(cfg, sc, fr, a, s, hit, acc);
        if (WRITE_HITS && hit) hit_bits[s >> 5] |= 1u << (s & 31);
    }
    pixel_finish(cfg, acc, out);
}

DM_HD float lin2srgb_mc(float x) {
    float r = (x > 0.0031308f) ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * x;
    return sat_(r);
}
// d lin2srgb / dx with torch's where / clamp gradient rules (clamp passes where 0 <= y <= 1)
DM_HD float lin2srgb_grad(float x) {
    float y, dy;
    if (x > 0.0031308f) { y = powf(x, 1.0f / 2.4f) * 1.055f - 0.055f; dy = 1.055f / 2.4f * powf(x, 1.0f / 2.4f - 1.0f); }
    else { y = 12.92f * x; dy = 12.92f; }
    return (y >= 0.f && y <= 1.f) ? dy : 0.f;
}

// d loss / d features from d loss / d color (color = lin2srgb(pre)); `px` from shade_pixel<Dual, ...>
DM_HD void finish_backward(const McCfg& cfg, const McPixel& px, const float* dcolor, float* dfeat) {
    float dalb[3], dmet = 0.f, dalpha = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = dcolor[c] * lin2srgb_grad(px.pre[c]);
        dalb[c] = g * px.dpre_dalbedo[c];
        dmet += g * px.dpre_dmetallic[c];
        dalpha += g * px.dpre_dalpha[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)                                      // albedo = clamp(sigmoid, 0, 1): always inside
        dfeat[c] = dalb[c] * px.m[c] * (1.f - px.m[c]);
    dfeat[3] = dmet * (cfg.max_metallic - cfg.min_metallic) * px.m[3] * (1.f - px.m[3]);
    dfeat[4] = dalpha * (cfg.max_rough_sq - cfg.min_rough_sq) * px.m[4] * (1.f - px.m[4]);
}

}  // namespace mc
}  // namespace dm
