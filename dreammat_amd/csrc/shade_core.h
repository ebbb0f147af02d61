// Per-pixel arithmetic of the fused G-buffer -> split-sum PBR shade kernel (forward + backward).
// Follows threestudio/models/materials/dreammat_material.py:746-762 (material activation, split-sum
// branch of forward) and :679-711 (shade_splitsum); lin2srgb = threestudio/utils/ops.py:83-88.
// Texture fetch rules (FG LUT bilinear/clamp; cube bilinear; trilinear over the roughness mip chain)
// restate nvdiffrast's dr.texture / ashawkey/envlight (see DESIGN.md "Texture rules").
// Shared by shade.hip and the host emulation under tests/hostemu.
#pragma once
#include "dm_common.h"

namespace dm {

constexpr int kMaxMips = 8;

struct EnvAtlas {
    const float4* spec;          // [n_env][mips...] each mip = 6 faces x (R+2) x (R+2) RGBA texels (1-texel border)
    const float4* diff;          // [n_env][6][(Rd+2)][(Rd+2)]
    const float2* fg_lut;        // [lut_res][lut_res] (row = roughness, col = n.v)
    long long spec_env_stride;   // texels
    long long diff_env_stride;
    long long mip_off[kMaxMips]; // texel offset of mip l inside one env
    int mip_res[kMaxMips];
    int n_mips;
    int diff_res;
    int lut_res;
    float min_rough_mip, max_rough_mip;   // envlight MIN/MAX_ROUGHNESS (0.08, 0.5)
    int texel_format;            // spec / diff texels: 0 = RGBA fp32 (16 B), 1 = RGBA fp16 (8 B), 2 = RGB18E8 (8 B, default):
                                 // with 8-byte texels the two texels of a bilinear ROW are one 16 B load
    const float4* fg_pairs;      // optional [lut_res][lut_res+1] x-pairs of the FG LUT (see fg_fetch); null = 4 plain taps
};

enum { kTexelF32 = 0, kTexelF16 = 1, kTexelRgb18e8 = 2 };

struct MatCfg {
    float min_metallic, max_metallic, min_roughness, max_roughness;
};

// 24-bit integer multiply: full rate on the VALU (v_mul_i32_i24 / v_mad_i32_i24), where a 32-bit v_mul_lo_u32 takes four
// issue slots.  Every product in this file is an index into an atlas far below 2^24 texels per factor.
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_MUL24(a, b) __mul24((a), (b))
#else
#define DM_MUL24(a, b) ((a) * (b))
#endif

struct F3 { float x, y, z; };
DM_HD F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
DM_HD F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DM_HD F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DM_HD F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DM_HD F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
DM_HD float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DM_HD float sat(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// nvdiffrast indexCubeMap: direction -> face, (u,v) in [0,1]
DM_HD int cube_index(F3 d, float& u, float& v) {
    // branch-free form of: |z| > max(|x|,|y|) -> z faces (4,5); else |y| > |x| -> y faces (2,3); else x faces (0,1)
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const bool zmaj = az > fmaxf(ax, ay);
    const bool ymaj = !zmaj && ay > ax;
    const float c = zmaj ? d.z : (ymaj ? d.y : d.x);
    const bool pos = c > 0.f;
    const float xs = pos ? d.x : -d.x, zs = pos ? d.z : -d.z;          // x faces: a = -sign(c) z; y faces: b = sign(c) z
    const float a = zmaj ? xs : (ymaj ? d.x : -zs);
    const float b = ymaj ? zs : -d.y;
    const int face = (zmaj ? 4 : (ymaj ? 2 : 0)) + (c < 0.f ? 1 : 0);
    const float m = 0.5f / fabsf(c);
    u = sat(a * m + 0.5f);
    v = sat(b * m + 0.5f);
    return face;
}

// Cube lookup split in two so that the (branchy) direction -> face/uv step is shared by the two mip
// levels of the specular fetch.  All texel indexing is 32-bit (an atlas is far below 2^31 texels).
struct CubeCoord { int face; float u, v; };

DM_HD CubeCoord cube_coord(F3 d) {
    CubeCoord c;
    c.face = cube_index(d, c.u, c.v);
    return c;
}

// bilinear fetch from one padded cube mip (texel (ix,iy) of face f at ((f*(R+2) + iy+1)*(R+2) + ix+1))
DM_HD F3 cube_fetch(const float4* __restrict__ tex, int R, CubeCoord cc) {
    float x = cc.u * (float)R - 0.5f, y = cc.v * (float)R - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int P = R + 2;
    int idx = (cc.face * P + (int)y0 + 1) * P + (int)x0 + 1;       // +1: border
    float4 t00 = tex[idx], t10 = tex[idx + 1], t01 = tex[idx + P], t11 = tex[idx + P + 1];
    float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    return f3(t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11,
              t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11,
              t00.z * w00 + t10.z * w10 + t01.z * w01 + t11.z * w11);
}

// fp16 atlas (opt-in): texel = 4 halves, so the two texels of a bilinear row are ONE 16-byte load (8-byte aligned).
// The shade kernels are bound by the number of scattered gather instructions per pixel (16 with fp32 texels, rocprofv3:
// SQ_WAIT_INST_ANY 54 %, VALU 9 %); this halves the cube-map gathers (12 -> 6).  Values are clamped to +-65504 when the
// atlas is packed; relative texel error <= 2^-11.
typedef unsigned int HalfRowBits __attribute__((ext_vector_type(4), aligned(4)));   // 8 halves = one 16 B load
DM_HD float half_lo(unsigned w) { unsigned short b = (unsigned short)(w & 0xffffu); _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }
DM_HD float half_hi(unsigned w) { unsigned short b = (unsigned short)(w >> 16); _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }
DM_HD F3 cube_fetch_half(const void* __restrict__ tex, long long texel_base, int R, CubeCoord cc) {
    float x = cc.u * (float)R - 0.5f, y = cc.v * (float)R - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int P = R + 2;
    long long idx = texel_base + (long long)(cc.face * P + (int)y0 + 1) * P + (int)x0 + 1;
    const char* base = (const char*)tex;
    // texel x0 = words 0,1 (r g | b a), texel x0+1 = words 2,3
    const HalfRowBits r0 = *reinterpret_cast<const HalfRowBits*>(base + idx * 8);
    const HalfRowBits r1 = *reinterpret_cast<const HalfRowBits*>(base + (idx + P) * 8);
    float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    return f3(half_lo(r0.x) * w00 + half_lo(r0.z) * w10 + half_lo(r1.x) * w01 + half_lo(r1.z) * w11,
              half_hi(r0.x) * w00 + half_hi(r0.z) * w10 + half_hi(r1.x) * w01 + half_hi(r1.z) * w11,
              half_lo(r0.y) * w00 + half_lo(r0.w) * w10 + half_lo(r1.y) * w01 + half_lo(r1.w) * w11);
}

// RGB18E8 atlas: 8-byte texel = three 18-bit mantissas sharing one 8-bit exponent (bits R[0,18) G[18,36) B[36,54) E[55,63):
// the exponent occupies the fp32 exponent field of the high word, so the scale 2^(E-127) is `hi & 0x7f800000`;
// value = mantissa * 2^(E-127)).  Radiance is non-negative, so no sign bits are needed; the largest channel of a texel keeps
// 18 significant bits (relative error <= 2^-18, the others the same ABSOLUTE error), i.e. fp32-class accuracy at half the
// bytes and -- the point -- half the gather instructions of RGBA fp32: the shade kernels are bound by the number of
// scattered cache lines the texture-address unit has to visit per wave, not by bytes or by VALU work.
DM_HD F3 rgb18e8_decode(unsigned lo, unsigned hi) {
    const unsigned r = lo & 0x3ffffu, g = ((lo >> 18) | (hi << 14)) & 0x3ffffu, b = (hi >> 4) & 0x3ffffu;
    const unsigned ebits = hi & 0x7f800000u;             // 2^(E-127) as fp32 bits (E in [1,254])
    float sc;
    __builtin_memcpy(&sc, &ebits, 4);
    return f3((float)r * sc, (float)g * sc, (float)b * sc);
}
DM_HD F3 cube_fetch_rgbe(const void* __restrict__ tex, long long texel_base, int R, CubeCoord cc) {
    float x = cc.u * (float)R - 0.5f, y = cc.v * (float)R - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int P = R + 2;
    long long idx = texel_base + (long long)(cc.face * P + (int)y0 + 1) * P + (int)x0 + 1;
    const char* base = (const char*)tex;
    const HalfRowBits r0 = *reinterpret_cast<const HalfRowBits*>(base + idx * 8);
    const HalfRowBits r1 = *reinterpret_cast<const HalfRowBits*>(base + (idx + P) * 8);
    F3 t00 = rgb18e8_decode(r0.x, r0.y), t10 = rgb18e8_decode(r0.z, r0.w);
    F3 t01 = rgb18e8_decode(r1.x, r1.y), t11 = rgb18e8_decode(r1.z, r1.w);
    float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    return f3(t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11,
              t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11,
              t00.z * w00 + t10.z * w10 + t01.z * w01 + t11.z * w11);
}
// Split form of the 8-byte-texel lookups, so that a kernel can put other memory operations BETWEEN the issue of the gathers
// and their first use: cube_tap_addr -> two 16 B row loads -> cube_tap_blend.
struct CubeTap { unsigned off0, off1; float fx, fy; };      // byte offsets of the two 16 B rows (an atlas is far below 4 GB)
DM_HD CubeTap cube_tap_addr(int texel_base, int R, CubeCoord cc) {
    float x = cc.u * (float)R - 0.5f, y = cc.v * (float)R - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    CubeTap t;
    t.fx = x - x0; t.fy = y - y0;
    const int P = R + 2;
    const int idx = texel_base + DM_MUL24(DM_MUL24(cc.face, P) + (int)y0 + 1, P) + (int)x0 + 1;
    t.off0 = (unsigned)idx * 8u;
    t.off1 = (unsigned)(idx + P) * 8u;
    return t;
}
// default 16-byte row fetch: plain pointer + byte offset (host emulation, legacy paths); the kernels pass buffer loads
struct PlainRows {
    const void* base;
    DM_HD HalfRowBits operator()(unsigned byte_off) const {
        return *reinterpret_cast<const HalfRowBits*>((const char*)base + byte_off);
    }
};
struct PlainFgRows {
    const float4* base;
    DM_HD float4 operator()(unsigned byte_off) const { return *reinterpret_cast<const float4*>((const char*)base + byte_off); }
};
// shared-exponent texels: the exponent scales the bilinear WEIGHT (one multiply per texel), the three 18-bit mantissas are
// converted and accumulated as they are -- 10 instructions per texel instead of 13 + 3 (the kernels are VALU-bound: ~450
// instructions per pixel, 12 texels per pixel)
DM_HD void rgb18e8_accum(unsigned lo, unsigned hi, float w, F3& acc) {
    const unsigned r = lo & 0x3ffffu, g = ((lo >> 18) | (hi << 14)) & 0x3ffffu, b = (hi >> 4) & 0x3ffffu;
    const unsigned ebits = hi & 0x7f800000u;
    float sc;
    __builtin_memcpy(&sc, &ebits, 4);
    const float ws = w * sc;
    acc.x += (float)r * ws; acc.y += (float)g * ws; acc.z += (float)b * ws;
}
DM_HD F3 cube_tap_blend(int fmt, HalfRowBits r0, HalfRowBits r1, float fx, float fy) {
    float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    if (fmt == kTexelRgb18e8) {
        F3 acc = f3(0.f, 0.f, 0.f);
        rgb18e8_accum(r0.x, r0.y, w00, acc);
        rgb18e8_accum(r0.z, r0.w, w10, acc);
        rgb18e8_accum(r1.x, r1.y, w01, acc);
        rgb18e8_accum(r1.z, r1.w, w11, acc);
        return acc;
    }
    return f3(half_lo(r0.x) * w00 + half_lo(r0.z) * w10 + half_lo(r1.x) * w01 + half_lo(r1.z) * w11,
              half_hi(r0.x) * w00 + half_hi(r0.z) * w10 + half_hi(r1.x) * w01 + half_hi(r1.z) * w11,
              half_lo(r0.y) * w00 + half_lo(r0.w) * w10 + half_lo(r1.y) * w01 + half_lo(r1.w) * w11);
}

// one bilinear cube lookup in whichever texel format the atlas carries (`texel_off` in texels)
DM_HD F3 cube_fetch_any(int fmt, const float4* __restrict__ tex, long long texel_off, int R, CubeCoord cc) {
    if (fmt == kTexelRgb18e8) return cube_fetch_rgbe(tex, texel_off, R, cc);
    if (fmt == kTexelF16) return cube_fetch_half(tex, texel_off, R, cc);
    return cube_fetch(tex + texel_off, R, cc);
}

DM_HD F3 cube_bilinear(const float4* __restrict__ tex, int R, F3 d) { return cube_fetch(tex, R, cube_coord(d)); }

// envlight.get_mip: roughness -> (fractional) mip level, and d level / d roughness
DM_HD float mip_level(const EnvAtlas& A, float rough, float& dlevel) {
    float n2 = (float)(A.n_mips - 2);
    float lo = A.min_rough_mip, hi = A.max_rough_mip;
    float level;
    if (rough < hi) {
        float c = fminf(fmaxf(rough, lo), hi);
        level = (c - lo) / (hi - lo) * n2;
        dlevel = (rough >= lo && rough <= hi) ? n2 / (hi - lo) : 0.f;
    } else {
        float c = fminf(fmaxf(rough, hi), 1.0f);
        level = (c - hi) / (1.0f - hi) + n2;
        dlevel = (rough <= 1.0f) ? 1.0f / (1.0f - hi) : 0.f;
    }
    return level;
}

struct ShadeOut {
    F3 color;        // clamp(albedo*diff + spec_albedo*spec, 0, 1)
    F3 albedo, diffuse_light, specular_light, specular_albedo;
    float metallic, roughness;
};

// Everything the backward needs is recomputed from the inputs (no saved tensors).
struct ShadeCtx {
    float s[5];          // sigmoid(features)
    F3 albedo, F0, diff, spec, spec_albedo, pre;   // pre = unclamped colour
    float metallic, roughness, fg0, fg1, dfg0_dv, dfg1_dv;
    F3 dspec_dlevel; float dlevel_drough;
};

DM_HD float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- two-stage evaluation (8-byte texel formats + FG pair table: the production configuration) --------------------
// shade_issue: activations, all addresses, ALL 8 gathers of the pixel issued back to back (raw rows kept in registers);
// shade_finish: decode, blend, compose.  A kernel puts its next-pixel input prefetch between the two, so that the
// in-order vmcnt wait in front of the first decode covers only the gathers (L2 hits) and not the younger HBM stream.
struct ShadeTaps {
    HalfRowBits s0a, s0b, s1a, s1b, da, db;     // rows y0 / y0+1 of: specular mip l0, specular mip l1, diffuse cube
    float4 fga, fgb;                             // FG x-pair rows iy0 / iy1
    float sfx0, sfy0, sfx1, sfy1, dfx, dfy, gfx, gfy, mipf, mip_sign;
};

// `mip_off` / `mip_res`: the atlas' per-mip tables; the kernels pass LDS copies, because indexing the kernel-argument copy
// with a per-lane mip level is a global load in front of the gathers that depend on it.
// (functors, not pointers: a generic pointer to LDS trips a gfx950 code-generation bug in this ROCm release)
template <int FMT, class MipOff, class MipRes, class SpecRows, class DiffRows, class FgRows>
DM_HD void shade_issue_t(const EnvAtlas& A, const MatCfg& M, int env, F3 n, F3 v, const float feat[5], ShadeCtx& c,
                         ShadeTaps& t, MipOff mip_off, MipRes mip_res, SpecRows spec_rows, DiffRows diff_rows, FgRows fg_rows) {
#pragma unroll
    for (int k = 0; k < 5; ++k) c.s[k] = sigmoidf(feat[k]);
    // (the reference's .clamp(0, 1) of the sigmoid, dreammat_material.py:749, is the identity: 1 + exp(-x) >= 1 in fp32 and
    // its reciprocal lies in [0, 1]; -ffast-math cannot prove that, so it is not written)
    c.albedo = f3(c.s[0], c.s[1], c.s[2]);
    c.metallic = c.s[3] * (M.max_metallic - M.min_metallic) + M.min_metallic;
    c.roughness = c.s[4] * (M.max_roughness - M.min_roughness) + M.min_roughness;
    float ndv = dot3(n, v);
    F3 refl = n * (2.f * ndv) - v;
    {
        int L = A.lut_res;
        float uu = sat(ndv), vv = sat(c.roughness);
        float x = uu * (float)L - 0.5f, y = vv * (float)L - 0.5f;
        float x0 = floorf(x), y0 = floorf(y);
        t.gfx = x - x0; t.gfy = y - y0;
        int iy0 = (int)y0;
        int iy1 = min(iy0 + 1, L - 1);
        iy0 = max(iy0, 0);
        t.fga = fg_rows((unsigned)(DM_MUL24(iy0, L + 1) + (int)x0 + 1) * 16u);
        t.fgb = fg_rows((unsigned)(DM_MUL24(iy1, L + 1) + (int)x0 + 1) * 16u);
    }
    {
        CubeTap d = cube_tap_addr(DM_MUL24(env, (int)A.diff_env_stride), A.diff_res, cube_coord(n));
        t.da = diff_rows(d.off0); t.db = diff_rows(d.off1);
        t.dfx = d.fx; t.dfy = d.fy;
    }
    {
        float level = mip_level(A, c.roughness, c.dlevel_drough);
        level = fminf(fmaxf(level, 0.f), (float)(A.n_mips - 1));
        int l0 = min((int)floorf(level), A.n_mips - 1);
        int l1 = min(l0 + 1, A.n_mips - 1);
        const float f = level - (float)l0;
        // The two mips go to the two fetch slots BY PARITY (slot 0 = the even level of the pair, slot 1 = the odd one), not by
        // rank: neighbouring pixels whose roughness straddles a level boundary -- pairs (1, 2) and (2, 3) -- then read the SAME
        // mip in slot 0, i.e. the same cache lines in that gather instruction (a wave whose lanes mix k levels visits ~k/2
        // mips per instruction instead of k).  t.mipf = the weight of slot 1; t.mip_sign = +-1 so that
        // d spec / d level = s(l1) - s(l0) = mip_sign (slot1 - slot0).
        const bool odd = (l0 & 1) != 0;
        t.mipf = odd ? 1.f - f : f;
        t.mip_sign = odd ? -1.f : 1.f;
        if (odd) { const int tmp = l0; l0 = l1; l1 = tmp; }
        const int envt = DM_MUL24(env, (int)A.spec_env_stride);
        CubeCoord rc = cube_coord(refl);
        CubeTap a0 = cube_tap_addr(envt + (int)mip_off(l0), mip_res(l0), rc);
        t.s0a = spec_rows(a0.off0); t.s0b = spec_rows(a0.off1);
        t.sfx0 = a0.fx; t.sfy0 = a0.fy;
        // l1 == l0 (last mip): the same rows are simply fetched twice (an L1 hit); copying the first pair instead would put
        // a wait for it in the middle of the gather sequence
        CubeTap a1 = cube_tap_addr(envt + (int)mip_off(l1), mip_res(l1), rc);
        t.s1a = spec_rows(a1.off0); t.s1b = spec_rows(a1.off1);
        t.sfx1 = a1.fx; t.sfy1 = a1.fy;
    }
}

template <int FMT>
DM_HD void shade_finish_t(const EnvAtlas& A, const MatCfg& M, const ShadeTaps& t, ShadeCtx& c) {
    const int fmt = FMT >= 0 ? FMT : A.texel_format;
    {
        const float fx = t.gfx, fy = t.gfy;
        float r0x = t.fga.x + (t.fga.z - t.fga.x) * fx, r0y = t.fga.y + (t.fga.w - t.fga.y) * fx;
        float r1x = t.fgb.x + (t.fgb.z - t.fgb.x) * fx, r1y = t.fgb.y + (t.fgb.w - t.fgb.y) * fx;
        c.fg0 = r0x + (r1x - r0x) * fy;
        c.fg1 = r0y + (r1y - r0y) * fy;
        bool inside = c.roughness > 0.f && c.roughness < 1.f;
        c.dfg0_dv = inside ? (r1x - r0x) * (float)A.lut_res : 0.f;
        c.dfg1_dv = inside ? (r1y - r0y) * (float)A.lut_res : 0.f;
    }
    c.F0 = f3(0.04f * (1.f - c.metallic) + c.metallic * c.albedo.x,
              0.04f * (1.f - c.metallic) + c.metallic * c.albedo.y,
              0.04f * (1.f - c.metallic) + c.metallic * c.albedo.z);
    c.spec_albedo = f3(c.F0.x * c.fg0 + c.fg1, c.F0.y * c.fg0 + c.fg1, c.F0.z * c.fg0 + c.fg1);
    c.diff = cube_tap_blend(fmt, t.da, t.db, t.dfx, t.dfy);
    F3 s0 = cube_tap_blend(fmt, t.s0a, t.s0b, t.sfx0, t.sfy0);
    F3 s1 = cube_tap_blend(fmt, t.s1a, t.s1b, t.sfx1, t.sfy1);
    c.spec = s0 * (1.f - t.mipf) + s1 * t.mipf;
    c.dspec_dlevel = (s1 - s0) * t.mip_sign;
    c.pre = c.albedo * c.diff + c.spec_albedo * c.spec;
}

// FMT >= 0: texel format fixed at compile time (the kernels: one code path, no per-fetch branch); FMT < 0: read from the atlas
template <int FMT>
DM_HD void shade_eval_t(const EnvAtlas& A, const MatCfg& M, int env, F3 n, F3 v, const float feat[5], ShadeCtx& c) {
    const int fmt = FMT >= 0 ? FMT : A.texel_format;
    if (fmt != kTexelF32 && A.fg_pairs) {                // production configuration: same arithmetic as the kernels' two stages
        ShadeTaps t;
        shade_issue_t<FMT>(A, M, env, n, v, feat, c, t, [&](int l) { return (int)A.mip_off[l]; },
                           [&](int l) { return A.mip_res[l]; }, PlainRows{A.spec}, PlainRows{A.diff}, PlainFgRows{A.fg_pairs});
        shade_finish_t<FMT>(A, M, t, c);
        return;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) c.s[k] = sigmoidf(feat[k]);
    c.albedo = f3(c.s[0], c.s[1], c.s[2]);          // (clamp(0, 1) of a sigmoid: the identity, see shade_issue_t)
    c.metallic = c.s[3] * (M.max_metallic - M.min_metallic) + M.min_metallic;
    c.roughness = c.s[4] * (M.max_roughness - M.min_roughness) + M.min_roughness;
    float ndv = dot3(n, v);
    F3 refl = n * (2.f * ndv) - v;
    // FG LUT, bilinear / clamp
    {
        int L = A.lut_res;
        float uu = sat(ndv), vv = sat(c.roughness);
        float x = uu * (float)L - 0.5f, y = vv * (float)L - 0.5f;
        float x0 = floorf(x), y0 = floorf(y);
        float fx = x - x0, fy = y - y0;
        int ix0 = (int)x0, iy0 = (int)y0;
        int ix1 = min(ix0 + 1, L - 1), iy1 = min(iy0 + 1, L - 1);
        ix0 = max(ix0, 0); iy0 = max(iy0, 0);
        float2 t00, t10, t01, t11;
        if (A.fg_pairs) {
            // x-pair table: entry (row, x0+1) = {lut[row][max(x0,0)], lut[row][min(x0+1,L-1)]}: the clamped pair of one
            // bilinear row as ONE aligned 16 B load (2 gathers per pixel instead of 4, same fp32 values)
            const float4 p0 = A.fg_pairs[iy0 * (L + 1) + (int)x0 + 1], p1 = A.fg_pairs[iy1 * (L + 1) + (int)x0 + 1];
            t00 = make_float2(p0.x, p0.y); t10 = make_float2(p0.z, p0.w);
            t01 = make_float2(p1.x, p1.y); t11 = make_float2(p1.z, p1.w);
        } else {
            t00 = A.fg_lut[iy0 * L + ix0]; t10 = A.fg_lut[iy0 * L + ix1];
            t01 = A.fg_lut[iy1 * L + ix0]; t11 = A.fg_lut[iy1 * L + ix1];
        }
        float r0x = t00.x + (t10.x - t00.x) * fx, r0y = t00.y + (t10.y - t00.y) * fx;
        float r1x = t01.x + (t11.x - t01.x) * fx, r1y = t01.y + (t11.y - t01.y) * fx;
        c.fg0 = r0x + (r1x - r0x) * fy;
        c.fg1 = r0y + (r1y - r0y) * fy;
        // d/d roughness of the bilinear interpolant (v = roughness inside (0,1) always: [0.1,0.95])
        bool inside = c.roughness > 0.f && c.roughness < 1.f;
        c.dfg0_dv = inside ? (r1x - r0x) * (float)L : 0.f;
        c.dfg1_dv = inside ? (r1y - r0y) * (float)L : 0.f;
    }
    c.F0 = f3(0.04f * (1.f - c.metallic) + c.metallic * c.albedo.x,
              0.04f * (1.f - c.metallic) + c.metallic * c.albedo.y,
              0.04f * (1.f - c.metallic) + c.metallic * c.albedo.z);
    c.spec_albedo = f3(c.F0.x * c.fg0 + c.fg1, c.F0.y * c.fg0 + c.fg1, c.F0.z * c.fg0 + c.fg1);
    c.diff = cube_fetch_any(fmt, A.diff, (long long)env * A.diff_env_stride, A.diff_res, cube_coord(n));
    {
        float level = mip_level(A, c.roughness, c.dlevel_drough);
        level = fminf(fmaxf(level, 0.f), (float)(A.n_mips - 1));
        int l0 = min((int)floorf(level), A.n_mips - 1);
        int l1 = min(l0 + 1, A.n_mips - 1);
        float f = level - (float)l0;
        const long long envt = (long long)env * A.spec_env_stride;
        CubeCoord rc = cube_coord(refl);
        F3 s0 = cube_fetch_any(fmt, A.spec, envt + A.mip_off[l0], A.mip_res[l0], rc);
        F3 s1 = (l1 != l0) ? cube_fetch_any(fmt, A.spec, envt + A.mip_off[l1], A.mip_res[l1], rc) : s0;
        c.spec = s0 * (1.f - f) + s1 * f;
        c.dspec_dlevel = s1 - s0;
    }
    c.pre = c.albedo * c.diff + c.spec_albedo * c.spec;
}

DM_HD void shade_eval(const EnvAtlas& A, const MatCfg& M, int env, F3 n, F3 v, const float feat[5], ShadeCtx& c) {
    shade_eval_t<-1>(A, M, env, n, v, feat, c);
}

DM_HD float lin2srgb1(float x) {
    float r = (x > 0.0031308f) ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * x;
    return sat(r);
}
DM_HD F3 lin2srgb(F3 a) { return f3(lin2srgb1(a.x), lin2srgb1(a.y), lin2srgb1(a.z)); }

// d loss / d features given d loss / d color
DM_HD void shade_backward(const MatCfg& M, const ShadeCtx& c, F3 dcol, float dfeat[5]) {
    // clamp(0,1) passes gradient where 0 <= pre <= 1 (torch.clamp semantics)
    F3 g = f3((c.pre.x >= 0.f && c.pre.x <= 1.f) ? dcol.x : 0.f,
              (c.pre.y >= 0.f && c.pre.y <= 1.f) ? dcol.y : 0.f,
              (c.pre.z >= 0.f && c.pre.z <= 1.f) ? dcol.z : 0.f);
    F3 gs = g * c.spec;                                 // d / d spec_albedo
    F3 dalb = g * c.diff + gs * (c.fg0 * c.metallic);   // via diffuse term and F0
    float dmet = c.fg0 * (gs.x * (c.albedo.x - 0.04f) + gs.y * (c.albedo.y - 0.04f) + gs.z * (c.albedo.z - 0.04f));
    float dfg0 = dot3(gs, c.F0);
    float dfg1 = gs.x + gs.y + gs.z;
    F3 gsa = g * c.spec_albedo;                         // d / d spec light
    float drough = dfg0 * c.dfg0_dv + dfg1 * c.dfg1_dv + dot3(gsa, c.dspec_dlevel) * c.dlevel_drough;
    float ds[5];
    // albedo = clamp(sigmoid, 0, 1): sigmoid is inside [0,1] => pass-through
    ds[0] = dalb.x; ds[1] = dalb.y; ds[2] = dalb.z;
    ds[3] = dmet * (M.max_metallic - M.min_metallic);
    ds[4] = drough * (M.max_roughness - M.min_roughness);
#pragma unroll
    for (int k = 0; k < 5; ++k) dfeat[k] = ds[k] * c.s[k] * (1.f - c.s[k]);
}

}  // namespace dm
