// MX-FP8 flash attention forward for 64-wide heads on the block-scaled matrix instruction of gfx950
// (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 / f16 MFMA rate) -- BASELINE.json configs[4] "fp16 UNet + fp8 MFMA attention":
// the S = 16384 self-attention of the UNet / ControlNet at 1024^2 (diffusers' Attention, reached from
// threestudio/models/guidance/dreammat_guidance.py:205-241, 261-282).  Same interface as dm_attention_fwd_bf16 / _f16: Q, K
// [B, S, Hh, 64] and V^T [B, Hh, 64, Skv] arrive as 16-bit tensors, the output is a 16-bit tensor; the 8-bit operands are an
// internal format.
//
// Format.  OCP e4m3 elements with one power-of-two scale (E8M0 byte) per 32 consecutive elements ALONG THE CONTRACTED
// DIMENSION -- the MX block the instruction dequantises in hardware.  Operand layout of the 32x32x64 instruction (measured, round
// 5: the first version assumed "lane l holds k-block l / 32" and was wrong exactly where the two blocks of a row had different
// scales): lane l holds row l % 32; with H = l / 32 its bytes 0-15 are k = 16 H .. 16 H + 15 and its bytes 16-31 are
// k = 32 + 16 H .. (two 32x32x32 operands side by side), while its SCALE byte is that of block H (k = 32 H .. 32 H + 31) of the row.
//   Q' = Q * softmax_scale * log2 e and K: blocks of 32 along the head dimension (2 per row), scale = 2^(E - 8) with E the
//        exponent of the block's largest magnitude, so that the scaled block lies in [128, 256) < 448 (no saturation);
//   V^T: blocks of 32 kv positions of one channel;
//   P:   exp2(s - shift) in [0, 256] with the constant scale 2^-8 (values below 2^-17 of the row maximum flush to zero).
// Two pre-passes (k_fp8_quant_rows for Q and K, k_fp8_quant_vt for V^T) write the 8-bit operands + scale bytes into a
// workspace: 1 byte per element instead of 2 on every re-read of K / V by the query blocks.  V^T is stored tile-major
// ([kv tile of 64][channel][64 bytes]).
//
// Kernel.  A workgroup = 4 waves x 64 query rows (two 32-row blocks per wave, one wave per SIMD), no LDS, no barrier: K / V^T tiles are 4 KB each and come straight from L2 as
// two 16-byte loads per lane and operand (the 4 waves of a workgroup and the workgroups of a (batch, head) on one XCD share
// them there), both double-buffered in registers one tile ahead.  Per 64-row kv tile and wave: 2 MFMAs form
// S^T = K.Q'^T - shift (two 32 x 32 tiles, the whole head dimension in one instruction each; the row shift enters through the C
// operand), the probabilities are ONE v_exp_f32 each, packed 4 per dword, and 3 MFMAs add V^T.P^T to the two 32 x 32 halves of
// O^T and to the row sums (a 33rd channel of ones).
//   S^T lane (q = l % 32, hi = l / 32) register r of tile t  <->  kv = 32 t + 4 hi + (r & 3) + 8 (r >> 2), i.e. the dword of
//   group g = r >> 2 holds kv 32 t + 8 g + 4 hi + {0..3}.  The B operand of V^T.P^T wants, in lane (q, H), kv 16 H + {0..15} of
//   tile 0 (bytes 0-15) and of tile 1 (bytes 16-31): groups 0, 1 of both lanes of the pair for H = 0, groups 2, 3 for H = 1.
//   v_permlane32_swap(group g, group g + 2) for g = 0, 1 hands each lane the partner's half, and the dwords (own g | partner g |
//   own g + 1 | partner g + 1) are the 16 positions in natural order -- V^T needs no permutation.
// Measured (round 5, S = 4096 x 24 x 5 / S = 16384 x 6 x 5, pre-passes included): 755 us = 683 TF/s / 2.47 ms = 835 TF/s, against
// 483 us / 1.75 ms of the bf16 kernel (attn_w128.hip) -- 0.14-0.17 of the 5 PFLOP/s MX-FP8 peak.  Three structures were
// measured and all land within 5 % of each other: one block per wave on two waves per SIMD (770 us), this one, and this one
// with the kv loop software-pipelined by hand (first product of tile j + 1 issued before the exponentials of tile j: 823 us, 512
// registers with spills outside the loop).
// What bounds it: the softmax on the vector pipe, not the matrix pipe.  Per tile a wave issues 5 MFMAs of 64 cycles (320) beside
// 64 v_exp_f32, 32 v_cvt_pk_fp8_f32 and 32 v_max3_f32 (~9.4 / 6.4 / 5.5 issue cycles each from one wave, tools/issue_probe.cpp:
// ~1000 cycles); the first version (running maximum, subtraction and row sum as VALU code: 64 more v_sub, 64 v_add, 32 v_max)
// measured 725 us at 24 x 5 x 4096^2 against 419 us of the bf16 kernel.  At D = 64 a score costs the vector pipe more than it
// costs the fp8 matrix pipe four times over, so the 2x MFMA rate cannot show.
#include <algorithm>
#include <cstdint>

#include "dm_common.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Fp8Args {
    const void* q; const void* k; const void* vt; void* out;
    long long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, vt_bs, vt_hs, vt_ds, o_bs, o_ss, o_hs;       // element strides
    int B, Hh, Sq, Skv;
    float scale_log2;
    unsigned char *q8, *qs, *k8, *ks, *v8, *vs;
};

template <typename T> struct Vec8 { typedef T type __attribute__((ext_vector_type(8))); };
template <typename T> struct Vec4 { typedef T type __attribute__((ext_vector_type(4))); };

// E8M0 byte of the block scale 2^e that brings a block with largest magnitude `amax` into [128, 256)
__device__ __forceinline__ int mx_scale_exp(float amax) {
    return amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) - 8 : 0;       // amax = f 2^E, f in [0.5, 1): amax 2^-(E-8) = f 2^8
}
__device__ __forceinline__ int e8m0(int e) { return min(max(e + 127, 0), 254); }

// 32 fp32 values of one MX block -> 8 dwords of e4m3, byte i = value order[i]
template <bool PERMUTE>
__device__ __forceinline__ void pack_block(const float (&v)[32], int e, int (&out)[8]) {
    // (the E8M0 byte may have been clamped: scale with the exponent the byte really encodes)
    const int eb = e8m0(e) - 127;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float x[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int pos = 4 * d + b;
            // PERMUTE: position 4 g + b <- element 8 g + b, position 16 + 4 g + b <- element 8 g + 4 + b (see the header)
            const int src = PERMUTE ? (pos < 16 ? 8 * (pos >> 2) + (pos & 3) : 8 * ((pos - 16) >> 2) + 4 + (pos & 3)) : pos;
            x[b] = __builtin_ldexpf(v[src], -eb);
        }
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
        out[d] = w;
    }
}

// Q / K: src [B, S, Hh, 64] through (bs, ss, hs) -> dst8 [B Hh][S][64] bytes, scales [B Hh][S][2].  One thread = one (row, half).
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_quant_rows(const T* __restrict__ src, long long bs, long long ss, long long hs, int B, int Hh,
                                                         int S, float mult, unsigned char* __restrict__ dst8, unsigned char* __restrict__ dsts) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)B * Hh * S * 2;
    if (id >= n) return;
    const int half = (int)(id & 1);
    const long long row = id >> 1;                        // (b Hh + h) S + s
    const int s = (int)(row % S);
    const long long bh = row / S;
    const int h = (int)(bh % Hh), b = (int)(bh / Hh);
    const T* p = src + b * bs + (long long)s * ss + (long long)h * hs + 32 * half;
    typedef typename Vec8<T>::type T8;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const T8 t = *reinterpret_cast<const T8*>(p + 8 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[8 * c + i] = (float)t[i] * mult; amax = fmaxf(amax, fabsf(v[8 * c + i])); }
    }
    const int e = mx_scale_exp(amax);
    int w[8];
    pack_block<false>(v, e, w);
    i32x4* o = reinterpret_cast<i32x4*>(dst8 + row * 64 + 32 * half);
    o[0] = i32x4{w[0], w[1], w[2], w[3]};
    o[1] = i32x4{w[4], w[5], w[6], w[7]};
    dsts[row * 2 + half] = (unsigned char)e8m0(e);
}

// V^T: src [B, Hh, 64, Skv] through (bs, hs, ds), kv contiguous -> dst8 [B Hh][Skv / 64][64 channels][64 bytes] (block bytes
// permuted), scales [B Hh][Skv / 64][64][2].  One thread = one (channel, block of 32 kv).
template <typename T>
__global__ __launch_bounds__(256) void k_fp8_quant_vt(const T* __restrict__ src, long long bs, long long hs, long long ds, int B, int Hh, int Skv,
                                                       unsigned char* __restrict__ dst8, unsigned char* __restrict__ dsts) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nblk = Skv / 32;
    const long long n = (long long)B * Hh * 64 * nblk;
    if (id >= n) return;
    // consecutive threads walk the kv blocks of one channel (coalesced 64-byte reads)
    const int blk = (int)(id % nblk);
    const long long r = id / nblk;
    const int d = (int)(r & 63);
    const long long bh = r >> 6;
    const int h = (int)(bh % Hh), b = (int)(bh / Hh);
    const T* p = src + b * bs + (long long)h * hs + (long long)d * ds + 32LL * blk;
    typedef typename Vec8<T>::type T8;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const T8 t = *reinterpret_cast<const T8*>(p + 8 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[8 * c + i] = (float)t[i]; amax = fmaxf(amax, fabsf(v[8 * c + i])); }
    }
    const int e = mx_scale_exp(amax);
    int w[8];
    pack_block<false>(v, e, w);
    const int tile = blk >> 1, hh = blk & 1;
    const long long slot = (bh * (Skv / 64) + tile) * 64 + d;
    i32x4* o = reinterpret_cast<i32x4*>(dst8 + slot * 64 + 32 * hh);
    o[0] = i32x4{w[0], w[1], w[2], w[3]};
    o[1] = i32x4{w[4], w[5], w[6], w[7]};
    dsts[slot * 2 + hh] = (unsigned char)e8m0(e);
}

constexpr int kNQ = 2;               // 32-row query blocks per wave
constexpr int kRowsPerWg = 4 * 32 * kNQ;      // 4 waves x kNQ x 32 query rows
constexpr int kPScaleByte = 127 - 8; // E8M0 of the constant 2^-8 that undoes the 2^8 inside the stored probabilities

struct KvFrag {
    i32x8 k[2]; int ks[2];           // K rows 32 t + l31 of the tile (this lane's two 16-byte k groups); the scale byte of its block
    i32x8 v[2]; int vs[2];           // V^T channels 32 f + l31 likewise
};

template <typename T>
__global__ __launch_bounds__(256, 1) void k_attn_fwd_fp8(Fp8Args a) {
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware (block b runs on XCD b % 8): all query blocks of one (batch, head) run on one XCD, whose L2 then serves
    // that head's 8-bit K / V^T to every one of them
    int bh, qblk;
    {
        const int nq = a.Sq / kRowsPerWg;
        const int BH = a.B * a.Hh, id = blockIdx.x;
        if ((BH & 7) == 0) {
            const int j = id >> 3;
            bh = (j / nq) * 8 + (id & 7);
            qblk = j - (j / nq) * nq;
        } else {
            bh = id / nq;
            qblk = id - bh * nq;
        }
    }
    const int n_tiles = a.Skv / 64;
    const int row0 = qblk * kRowsPerWg + wave * 32 * kNQ;     // this wave's first query row

    // ONE wave per SIMD, kNQ query blocks per wave (round 5, second version): the first version (one block per wave, two waves
    // per SIMD) spent a tile in a serial chain -- scores out of the matrix pipe, row maximum, exponentials, packing, hand-over,
    // second product -- with only the partner wave to fill the gaps and every wave fetching its own copy of K / V^T: neither
    // pipe above 40 % busy.  Two independent blocks per wave give the in-order stream a second chain to issue from and halve
    // the operand bytes per score.
    i32x8 qB[kNQ];
    int qsc[kNQ];
#pragma unroll
    for (int qb = 0; qb < kNQ; ++qb) {
        const long long qrow = (long long)bh * a.Sq + row0 + 32 * qb + l31;
        const i32x4* qp = reinterpret_cast<const i32x4*>(a.q8 + qrow * 64 + 16 * hi);
        const i32x4 q0 = qp[0], q1 = qp[2];          // bytes 16 hi .. and 32 + 16 hi .. of the row (operand layout: see the header)
        qB[qb] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        qsc[qb] = a.qs[qrow * 2 + hi];
    }

    const unsigned char* k8 = a.k8 + ((long long)bh * a.Skv + l31) * 64 + 16 * hi;
    const unsigned char* ksc = a.ks + ((long long)bh * a.Skv + l31) * 2 + hi;
    const unsigned char* v8 = a.v8 + ((long long)bh * n_tiles * 64 + l31) * 64 + 16 * hi;
    const unsigned char* vsc = a.vs + ((long long)bh * n_tiles * 64 + l31) * 2 + hi;
    auto load_tile = [&](int j, KvFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const i32x4* p = reinterpret_cast<const i32x4*>(k8 + ((long long)j * 64 + 32 * t) * 64);
            const i32x4 lo = p[0], up = p[2];
            f.k[t] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            f.ks[t] = ksc[((long long)j * 64 + 32 * t) * 2];
            const i32x4* pv = reinterpret_cast<const i32x4*>(v8 + ((long long)j * 64 + 32 * t) * 64);
            const i32x4 vlo = pv[0], vup = pv[2];
            f.v[t] = i32x8{vlo[0], vlo[1], vlo[2], vlo[3], vup[0], vup[1], vup[2], vup[3]};
            f.vs[t] = vsc[((long long)j * 64 + 32 * t) * 2];
        }
    };

    // O^T accumulators per query block: fragments 0 / 1 = channels 0-31 / 32-63; fragment 2 = a 33rd "channel" of ones in row 0
    // (lanes 0-31, register 0), whose product with P^T is the row sum of the probabilities AS ROUNDED to e4m3 -- numerator and
    // denominator see the same rounding, and the 64 v_add_f32 per tile the sum would cost on the vector pipe (5.5 cycles each,
    // beside MFMAs of 64) become one more MFMA
    f32x16 o[kNQ][3];
#pragma unroll
    for (int qb = 0; qb < kNQ; ++qb)
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][f][r] = 0.f;
    const int one_w = l31 == 0 ? 0x38383838 : 0;       // e4m3 1.0 = 0x38
    const i32x8 onesA = {one_w, one_w, one_w, one_w, one_w, one_w, one_w, one_w};
    // C operand of the first product: -shift of this lane's query row, so that the MFMA delivers s - shift and a probability is
    // ONE v_exp_f32.  shift = (row maximum so far) - 8: the stored probability exp2(s - shift) <= 2^8 < 448.  It moves (rarely
    // after the first tiles) when a row's maximum grows: O, the scores at hand and the C operand are re-based by the difference.
    f32x16 cinit[kNQ];
#pragma unroll
    for (int qb = 0; qb < kNQ; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[qb][r] = 0.f;

    auto tile_body = [&](const KvFrag& fk, bool first) __attribute__((always_inline)) {
        f32x16 s[kNQ][2];
#pragma unroll
        for (int qb = 0; qb < kNQ; ++qb)
#pragma unroll
            for (int t = 0; t < 2; ++t)  // S^T[kv][q] - shift[q] = sum_d K[kv][d] Q'[q][d] - shift[q]
                s[qb][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fk.k[t], qB[qb], cinit[qb], 0, 0, 0, fk.ks[t], 0, qsc[qb]);
        float mx[kNQ];
        bool grow_any = false;
#pragma unroll
        for (int qb = 0; qb < kNQ; ++qb) {
            float m = fmaxf(s[qb][0][0], s[qb][1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(fmaxf(m, s[qb][0][r]), s[qb][1][r]);
            mx[qb] = fmaxf(m, __shfl_xor(m, 32));
            grow_any = grow_any || mx[qb] > 8.f;
        }
        if (first || __any(grow_any)) {
#pragma unroll
            for (int qb = 0; qb < kNQ; ++qb) {
                const float delta = (first || mx[qb] > 8.f) ? mx[qb] - 8.f : 0.f;
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int ff = 0; ff < 3; ++ff)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][ff][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { cinit[qb][r] -= delta; s[qb][0][r] -= delta; s[qb][1][r] -= delta; }
            }
        }
        i32x8 pB[kNQ];
#pragma unroll
        for (int qb = 0; qb < kNQ; ++qb) {
            int pt[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(s[qb][t][4 * g + e]);
                    int w = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], 0, false);
                    pt[t][g] = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], w, true);
                }
            // B operand of V^T.P^T: lane (q, H) wants kv 16 H .. 16 H + 15 of both 32-row score tiles.  Lane (q, 0) keeps its
            // groups g = 0, 1 and receives the partner's, lane (q, 1) likewise for g = 2, 3 -- one v_permlane32_swap per dword
            // pair -- and the bytes come out in natural kv order (see the header)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)pt[t][g], (unsigned)pt[t][g + 2], false, false);
                    pB[qb][4 * t + 2 * g] = (int)sw[0];         // lanes 0-31: own (hi 0, g)     | lanes 32-63: (hi 0, g + 2) of the partner
                    pB[qb][4 * t + 2 * g + 1] = (int)sw[1];     // lanes 0-31: (hi 1, g) received | lanes 32-63: own (hi 1, g + 2)
                }
        }
#pragma unroll
        for (int qb = 0; qb < kNQ; ++qb) {
#pragma unroll
            for (int ff = 0; ff < 2; ++ff)   // O^T[d][q] += sum_kv V^T[d][kv] P[q][kv]
                o[qb][ff] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fk.v[ff], pB[qb], o[qb][ff], 0, 0, 0, fk.vs[ff], 0, kPScaleByte);
            o[qb][2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(onesA, pB[qb], o[qb][2], 0, 0, 0, 127, 0, kPScaleByte);
        }
    };

    // both operands of a tile are requested one tile ahead
    KvFrag fa, fb;
    load_tile(0, fa);
    if (n_tiles > 1) load_tile(1, fb);
    tile_body(fa, true);
    int j = 1;
    for (; j + 2 <= n_tiles; j += 2) {
        load_tile(j + 1, fa);
        tile_body(fb, false);
        if (j + 2 < n_tiles) load_tile(j + 2, fb);
        tile_body(fa, false);
    }
    if (j < n_tiles) tile_body(fb, false);

    const int bq = bh / a.Hh, hq = bh - bq * a.Hh;
    typedef typename Vec4<T>::type T4;
#pragma unroll
    for (int qb = 0; qb < kNQ; ++qb) {
        // row sum: row 0 of fragment 2 = register 0 of lanes 0-31 (column q = lane)
        const float lt = __shfl(o[qb][2][0], l31);
        const float inv = 1.f / lt;           // (numerator and denominator carry the same 2^8 / 2^-8)
        // O^T lane (q, hi), fragment f, register r = channel 32 f + 8 (r >> 2) + 4 hi + (r & 3): four consecutive channels per group
        T* op = reinterpret_cast<T*>(a.out) + bq * a.o_bs + (long long)(row0 + 32 * qb + l31) * a.o_ss + (long long)hq * a.o_hs;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (T)(o[qb][f][4 * g + e] * inv);
                *reinterpret_cast<T4*>(op + 32 * f + 8 * g + 4 * hi) = w;
            }
    }
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Fp8Ws { size_t q8, qs, k8, ks, v8, vs, total; };
Fp8Ws ws_layout(int B, int Hh, int Sq, int Skv) {
    const size_t bh = (size_t)B * Hh;
    Fp8Ws w;
    size_t off = 0;
    w.q8 = off; off += align256(bh * Sq * 64);
    w.qs = off; off += align256(bh * Sq * 2);
    w.k8 = off; off += align256(bh * Skv * 64);
    w.ks = off; off += align256(bh * Skv * 2);
    w.v8 = off; off += align256(bh * Skv * 64);
    w.vs = off; off += align256(bh * Skv * 2);
    w.total = off;
    return w;
}

template <typename T>
int launch_fp8(Fp8Args& a, hipStream_t stream) {
    DM_ENTER();
    const long long nq = (long long)a.B * a.Hh * a.Sq * 2, nk = (long long)a.B * a.Hh * a.Skv * 2, nv = (long long)a.B * a.Hh * 64 * (a.Skv / 32);
    hipLaunchKernelGGL(k_fp8_quant_rows<T>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, (const T*)a.q, a.q_bs, a.q_ss, a.q_hs, a.B,
                       a.Hh, a.Sq, a.scale_log2, a.q8, a.qs);
    hipLaunchKernelGGL(k_fp8_quant_rows<T>, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, stream, (const T*)a.k, a.k_bs, a.k_ss, a.k_hs, a.B,
                       a.Hh, a.Skv, 1.0f, a.k8, a.ks);
    hipLaunchKernelGGL(k_fp8_quant_vt<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, (const T*)a.vt, a.vt_bs, a.vt_hs, a.vt_ds, a.B,
                       a.Hh, a.Skv, a.v8, a.vs);
    DM_LAUNCH_CHECK();
    const long long blocks = (long long)a.B * a.Hh * (a.Sq / kRowsPerWg);
    hipLaunchKernelGGL(k_attn_fwd_fp8<T>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // namespace

extern "C" {

// bytes of the 8-bit operand workspace of dm_attention_fwd_fp8 for these sizes (0: the shape is not served)
size_t dm_attention_fp8_workspace_bytes(int B, int Hh, int Sq, int Skv) {
    if (B <= 0 || Hh <= 0 || Sq <= 0 || Skv <= 0 || Sq % kRowsPerWg != 0 || Skv % 64 != 0) return 0;
    return ws_layout(B, Hh, Sq, Skv).total;
}

// softmax(q k^T scale) v with the two matrix products on the MX-FP8 matrix instruction.  q, k, vt, out: the tensors and strides of
// dm_attention_fwd_bf16 (include/dreammat_hip.h) with D = 64; elem_f16 = 0: bf16 tensors, 1: IEEE half.  Sq % 256 == 0,
// Skv % 64 == 0 (self-attention of the 64-wide SD-2.1 heads; anything else: DM_ERR_UNSUPPORTED, the caller keeps the 16-bit
// kernels).  ws: dm_attention_fp8_workspace_bytes(B, Hh, Sq, Skv) bytes, 256-byte aligned, scratch.
int dm_attention_fwd_fp8(const void* q, const void* k, const void* vt, void* out, int B, int Hh, int Sq, int Skv, int D, long long q_bs,
                         long long q_ss, long long q_hs, long long k_bs, long long k_ss, long long k_hs, long long vt_bs, long long vt_hs,
                         long long vt_ds, long long o_bs, long long o_ss, long long o_hs, float scale, int elem_f16, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
    if (!q || !k || !vt || !out || !ws || B <= 0 || Hh <= 0 || Sq <= 0 || Skv <= 0) return DM_ERR_ARG;
    if (D != 64 || Sq % kRowsPerWg != 0 || Skv % 64 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)ws) & 15 || ((uintptr_t)out & 7)) return DM_ERR_ARG;
    if ((q_bs | q_ss | q_hs | k_bs | k_ss | k_hs | vt_bs | vt_hs | vt_ds) & 7) return DM_ERR_ARG;
    if ((o_bs | o_ss | o_hs) & 3) return DM_ERR_ARG;
    if ((long long)B * Hh * (Sq / kRowsPerWg) > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    const Fp8Ws w = ws_layout(B, Hh, Sq, Skv);
    if (ws_bytes < w.total) return DM_ERR_WORKSPACE;
    Fp8Args a;
    a.q = q; a.k = k; a.vt = vt; a.out = out;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = q_hs; a.k_bs = k_bs; a.k_ss = k_ss; a.k_hs = k_hs;
    a.vt_bs = vt_bs; a.vt_hs = vt_hs; a.vt_ds = vt_ds; a.o_bs = o_bs; a.o_ss = o_ss; a.o_hs = o_hs;
    a.B = B; a.Hh = Hh; a.Sq = Sq; a.Skv = Skv;
    a.scale_log2 = scale * 1.4426950408889634f;
    unsigned char* base = (unsigned char*)ws;
    a.q8 = base + w.q8; a.qs = base + w.qs; a.k8 = base + w.k8; a.ks = base + w.ks; a.v8 = base + w.v8; a.vs = base + w.vs;
    return elem_f16 ? launch_fp8<_Float16>(a, stream) : launch_fp8<__bf16>(a, stream);
}

}  // extern "C"
