// 3x3 convolution for the few-channel STEM layers of the nets (Cin < 32): ControlNetConditioningEmbedding conv_in 22->16 and
// 16->16 @512^2, 16->32 stride 2, UNet / ControlNet conv_in 4->320 @64^2 (the F.conv2d calls diffusers makes under
// threestudio/models/guidance/dreammat_guidance.py:205-292).  These are bandwidth-shaped, not GEMM-shaped: 9*Cin is 36-198
// MACs per output value, an MFMA tile would be mostly zero padding, and the im2col + GEMM lowering they used to take wrote
// and re-read a 9x copy of the activation (604 MB for 8 x 16ch @512^2: 1.5 ms of im2col per step plus the GEMMs behind it).
// Direct form: one thread = one output pixel x 16 output channels; the 3x3xCin patch comes straight from NHWC (consecutive
// lanes = consecutive pixels), the weights of the 16 channels sit in LDS and are read as broadcasts (every lane the same
// address), v_dot2c_f32_bf16 does two MACs per lane per issue, bias and the SiLU that follows every one of these layers in
// ControlNetConditioningEmbedding.forward are applied before the single rounding to bf16.
//   x [B,Hin,Win,Cin] bf16, w [Cout,9,Cin] bf16 (tap-major, the layout of dm_conv3x3_nhwc_bf16), bias [Cout] bf16 or NULL,
//   y [B,Hout,Wout,Cout] bf16; Cin even, <= 32; Cout % 16 == 0.  Algorithmic bytes: (Cin + Cout) * 2 per output pixel.
#include "dm_common.h"
#include "dm_elem.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int CO = 16;                         // output channels per thread

struct SmallConvArgs {
    const elem_t* x; const elem_t* w; const elem_t* bias; elem_t* y;
    const elem_t* res; int res_B;              // optional [res_B, Hout, Wout, Cout] added before the rounding, image b takes b % res_B
    int B, Hin, Win, Hout, Wout, Cout, stride, pad_y, pad_x, act;
    long long n_pix;                           // B*Hout*Wout
};

template <int CIN>
__global__ __launch_bounds__(256) void k_conv3x3_small(SmallConvArgs a) {
    constexpr int CP = CIN / 2;                // channel pairs
    __shared__ elem2 wl[CO * 9 * CP];         // [co][tap][pair]
    const int cog = blockIdx.y;
    for (int i = threadIdx.x; i < CO * 9 * CP; i += 256)
        wl[i] = reinterpret_cast<const elem2*>(a.w + (long long)cog * CO * 9 * CIN)[i];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_pix) return;
    const int xo = (int)(p % a.Wout);
    const long long r = p / a.Wout;
    const int yo = (int)(r % a.Hout), b = (int)(r / a.Hout);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = a.bias ? (float)a.bias[cog * CO + c] : 0.f;
    const int y0 = yo * a.stride - a.pad_y, x0 = xo * a.stride - a.pad_x;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y0 + t / 3, xx = x0 + t % 3;
        if ((unsigned)yy >= (unsigned)a.Hin || (unsigned)xx >= (unsigned)a.Win) continue;      // zero padding
        const elem2* src = reinterpret_cast<const elem2*>(a.x + (((long long)b * a.Hin + yy) * a.Win + xx) * CIN);
        elem2 v[CP];
#pragma unroll
        for (int k = 0; k < CP; ++k) v[k] = src[k];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const elem2* wr = wl + (c * 9 + t) * CP;
#pragma unroll
            for (int k = 0; k < CP; ++k) acc[c] = DM_FDOT2(v[k], wr[k], acc[c]);
        }
    }
    if (a.res) {                                // (ControlNet: conv_in(sample) + conditioning embedding, the embedding of the B views
        elem_t rv[CO];                          //  shared by the text / negative / null branches)
        const long long pr = ((long long)(b % a.res_B) * a.Hout + yo) * a.Wout + xo;
        const uint4* rp = reinterpret_cast<const uint4*>(a.res + pr * a.Cout + cog * CO);
        *reinterpret_cast<uint4*>(rv) = rp[0];
        *reinterpret_cast<uint4*>(rv + 8) = rp[1];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] += (float)rv[c];
    }
    elem_t o[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        float z = acc[c];
        if (a.act) z = z * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));     // SiLU on v_exp / v_rcp (an IEEE division was 10 instructions)
        o[c] = (elem_t)z;
    }
    uint4* dst = reinterpret_cast<uint4*>(a.y + p * a.Cout + cog * CO);
    dst[0] = *reinterpret_cast<const uint4*>(o);
    dst[1] = *reinterpret_cast<const uint4*>(o + 8);
}

template <int CIN>
int launch_small(const SmallConvArgs& a, hipStream_t stream) {
    const long long blocks = (a.n_pix + 255) / 256;
    if (blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL(k_conv3x3_small<CIN>, dim3((unsigned)blocks, a.Cout / CO), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}


// ------------------------------------------------------------------------------------------------
// The same layers on the matrix pipe (round 4).  The direct kernel above runs 22 -> 16 on 8 x 512^2 in 0.20 ms and 4 -> 128 in
// 0.52 ms where their bytes take 0.03 / 0.06: per (pixel, 16 channels) it pays nine bounds tests, nine 64-bit address products
// and one LDS read per MAC pair.  Here a workgroup owns a 16 x 16 tile of output pixels: the (15 s + 3)^2-pixel input patch goes
// to LDS ONCE (zero-filled outside the image and beyond Cin: CP = 16 / 32 / 128 channels per pixel row), the weights of 32 output
// channels at a time as [cout][tap][CP], and each wave multiplies its 64 pixels (four tile rows) by them with
// v_mfma_f32_32x32x16_bf16, nine taps = nine shifted views of the patch.  Rows are XOR-swizzled by 16-byte chunk so that the 16
// pixels of a tile row (consecutive patch rows) read 16 different bank groups.  Operand roles as in csrc/conv.hip: weights = A,
// pixels = B, so a lane holds 16 channels of ONE pixel and the epilogue (bias, residual, SiLU, one rounding) stores whole 8- or
// 16-byte channel runs.  Cin = 128 -> Cout = 4 is the data gradient of the VAE encoder's conv_in (the rendered image is the leaf).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

template <int CP, int STRIDE, int TH_ = 16, int WR_ = 32>
struct PatchCfg {
    static constexpr int TW = 16, TH = TH_, WR = WR_, PW = (TW - 1) * STRIDE + 3, PH = (TH - 1) * STRIDE + 3;
    static constexpr int RB = CP * 2, NC = CP / 8, RPL = NC >= 16 ? 1 : 16 / NC;   // row bytes, 16-byte chunks per row, rows per 256 bytes
    static constexpr int PATCH_BYTES = ((PH * PW * RB + 255) / 256) * 256, W_BYTES = WR * 9 * RB;     // W_BYTES: one block of 32 output channels (WR rows of it stored)
    __device__ __host__ static int swz(int row) { return (row / RPL) & (NC - 1); }
};

// Persistent workgroups: the weights of ALL output-channel blocks are loaded once per workgroup when they fit beside the patch
// (`w_res` blocks resident; otherwise one block at a time, reloaded per tile), then the workgroup walks tiles -- per tile one patch
// load between two barriers and no other synchronisation.  (Per-tile, per-block weight loads -- a barrier pair and an L2 round trip
// in front of 18-36 MFMAs -- were most of the first version's time.)
//
// TH = 8, WR = 4 (round 6, the Cin = 128 -> Cout <= 4 launch): a 16 x 8 tile and only the four real weight rows in LDS (lanes of the
// other 28 rows of the 32-row MFMA operand hold zeros without reading) = 55 KB per workgroup, two workgroups per CU instead of one;
// the patch arrives by unconditional buffer loads, all in flight before the first LDS store (out-of-image pixels are offsets past the
// descriptor's end: zeros, no branch -- the bounds-tested form was one serialized HBM round trip per 256 chunks, twelve per tile).
template <int CP, int STRIDE, int TH = 16, int WR = 32>
__global__ __launch_bounds__(256, 2) void k_conv3x3_patch(SmallConvArgs a, int cin, int tiles_x, int tiles_y, int n_tiles, int w_res) {
    using C = PatchCfg<CP, STRIDE, TH, WR>;
    constexpr int MB = TH / 8;                                             // 32-pixel M blocks (two tile rows each) per wave
    static_assert(TH == 8 || TH == 16, "tile height");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_cb = (a.Cout + 31) / 32;
    char* const patch = smem;
    char* const wl0 = smem + C::PATCH_BYTES;                              // w_res blocks of W_BYTES
    float* const sbias = reinterpret_cast<float*>(wl0 + w_res * C::W_BYTES);      // n_cb * 32 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // 8 channels (one 16-byte chunk) of a [.., cin] row from channel ch0 on; zeros beyond cin
    auto chunk_of = [&](const elem_t* rowp, int ch0) {
        unsigned v[4] = {0u, 0u, 0u, 0u};
        if (ch0 < cin) {
            const elem_t* src = rowp + ch0;
            if (ch0 + 8 <= cin && (cin & 7) == 0) {
                const uint4 q = *reinterpret_cast<const uint4*>(src);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ch0 + 2 * k < cin) v[k] = reinterpret_cast<const unsigned*>(src)[k];           // (Cin is even)
            }
        }
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto load_weights = [&](int cb, char* wl) {                           // block cb as [cout_local][tap][CP], zero rows past Cout
        for (int i = tid; i < WR * 9 * C::NC; i += 256) {
            const int row = i / C::NC, c = i - row * C::NC;
            const int co = cb * 32 + row / 9, tap = row - (row / 9) * 9;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (co < a.Cout) v = chunk_of(a.w + ((long long)co * 9 + tap) * cin, 8 * c);
            *reinterpret_cast<uint4*>(wl + row * C::RB + ((c ^ C::swz(row)) << 4)) = v;
        }
    };
    for (int i = tid; i < n_cb * 32; i += 256) sbias[i] = (a.bias && i < a.Cout) ? (float)a.bias[i] : 0.f;
    if (w_res >= n_cb)
        for (int cb = 0; cb < n_cb; ++cb) load_weights(cb, wl0 + cb * C::W_BYTES);
    [[maybe_unused]] const bool wide_store = a.Cout % 32 == 0 && (((uintptr_t)a.y) & 15) == 0;      // (whole 32-channel blocks: Cout = 16 has nothing to pair)
    const int trow0 = 2 * MB * wave + (l31 >> 4), tcol = l31 & 15;        // this lane's pixel of M block 0 (block 1: two tile rows down)
    [[maybe_unused]] const bool w_live = l31 < WR;                        // (WR < 32: this lane's weight row exists)
    // The patch arrives by unconditional buffer loads, all of a tile's in flight before the first LDS store (out-of-image pixels and
    // channel chunks beyond Cin are offsets past the descriptor's end: zeros, no branch; a chunk that straddles Cin -- 22 = 2.75
    // chunks -- is loaded whole and its tail dwords cleared).  The bounds-tested form it replaces was one serialized HBM round trip
    // per 256 chunks.  AHEAD (at most 12 chunks per thread): the NEXT tile's patch is requested into registers before this tile's
    // MFMAs, the loads are in flight while the workgroup multiplies, and land in LDS after the barrier that follows.
    constexpr int NCH = C::PH * C::PW * C::NC, NTRIP = (NCH + 255) / 256;
    [[maybe_unused]] constexpr bool AHEAD = NTRIP <= 12;
    [[maybe_unused]] u32x4s pv[NTRIP];
#if defined(__HIP_DEVICE_COMPILE__)
    auto request_patch = [&](int tile_) __attribute__((always_inline)) {
        int t_ = tile_;
        const int tx_ = t_ % tiles_x; t_ /= tiles_x;
        const int ty_ = t_ % tiles_y;
        const int b_ = t_ / tiles_y;
        const int iy0_ = ty_ * C::TH * STRIDE - a.pad_y, ix0_ = tx_ * C::TW * STRIDE - a.pad_x;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(unsigned)((long long)a.B * a.Hin * a.Win * cin * 2), 0x00020000);
#pragma unroll
        for (int t = 0; t < NTRIP; ++t) {
            const int i = tid + 256 * t;
            const int row = i / C::NC, c = i - row * C::NC;
            const int py = row / C::PW, px = row - py * C::PW;
            const int yy = iy0_ + py, xx = ix0_ + px;
            const bool ok = i < NCH && tile_ < n_tiles && 8 * c < cin && (unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win;
            const unsigned off = ok ? ((((unsigned)b_ * (unsigned)a.Hin + (unsigned)yy) * (unsigned)a.Win + (unsigned)xx) * (unsigned)cin + 8u * (unsigned)c) * 2u : 0xfffffff0u;
#if defined(DM_ABL_SMALL_NOLOAD)
            pv[t] = u32x4s{off, 0u, 0u, 0u};
#else
            pv[t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0);
#endif
        }
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTRIP; ++t) {
            const int i = tid + 256 * t;
            const int row = i / C::NC, c = i - row * C::NC;
            u32x4s v = pv[t];
            if constexpr (CP != 128) {                                     // (Cin even: whole dwords are inside or outside)
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (8 * c + 2 * k >= cin) v[k] = 0u;
            }
            if (i < NCH) *reinterpret_cast<u32x4s*>(patch + row * C::RB + ((c ^ C::swz(row)) << 4)) = v;
        }
    };
    if constexpr (AHEAD) request_patch((int)blockIdx.x);
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int oy0 = ty * C::TH, ox0 = tx * C::TW;
        __syncthreads();                                                   // the previous tile's patch is no longer read
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (!AHEAD) request_patch(tile);
        store_patch();
#endif
        __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (AHEAD) request_patch(tile + (int)gridDim.x);
#endif
        for (int cb = 0; cb < n_cb; ++cb) {
            const char* wl = wl0 + (w_res >= n_cb ? cb : 0) * C::W_BYTES;
            if (w_res < n_cb) {                                            // (the weights do not all fit: one block at a time)
                if (cb) __syncthreads();
                load_weights(cb, wl0);
                __syncthreads();
            }
            if constexpr (WR == 4) {
                // Four output channels: the 16 x 16 x 32 MFMA (A = 16 weight rows, 4 of them real; B = the 16 pixels of ONE tile row; a
                // lane holds the k-quarter lane >> 4 of a 32-channel step) -- half the matrix-pipe time of the 32 x 32 x 16 form, whose 32
                // weight rows did 4 rows of work.  A wave owns TH / 4 tile rows; D = [4 (lane >> 4) + r][lane & 15]: lanes 0-15 hold the
                // four channels of their pixel and store them as one 8-byte run, 128 contiguous bytes per tile row.
                static_assert(STRIDE == 1 && CP % 32 == 0, "the 4-channel form");
                constexpr int RW = TH / 4;
                const int l15 = lane & 15, kq = lane >> 4;
                const bool live4 = l15 < WR;
                f32x4 acc4[RW];
#pragma unroll
                for (int m = 0; m < RW; ++m) acc4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - 3 * dy;
                    const int wrow = (live4 ? l15 : 0) * 9 + tap;
                    int prow[RW];
#pragma unroll
                    for (int m = 0; m < RW; ++m) prow[m] = (RW * wave + m + dy) * C::PW + l15 + dx;
#pragma unroll
                    for (int kk = 0; kk < CP / 32; ++kk) {
                        const int ch = 4 * kk + kq;
                        elem8 wf = *reinterpret_cast<const elem8*>(wl + wrow * C::RB + ((ch ^ C::swz(wrow)) << 4));
                        if (!live4) wf = elem8{};
#pragma unroll
                        for (int m = 0; m < RW; ++m) {
                            const elem8 pf = *reinterpret_cast<const elem8*>(patch + prow[m] * C::RB + ((ch ^ C::swz(prow[m])) << 4));
                            acc4[m] = DM_MFMA_16x16x32(wf, pf, acc4[m]);
                        }
                    }
                }
                if (kq == 0) {
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        const int oy = oy0 + RW * wave + m, ox = ox0 + l15;
                        if (oy < a.Hout && ox < a.Wout) {
                            const long long po = (((long long)b * a.Hout + oy) * a.Wout + ox) * a.Cout;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc4[m][e] + sbias[e];
                            if (a.res) {
                                const elem4 rv = *reinterpret_cast<const elem4*>(a.res + (((long long)(b % a.res_B) * a.Hout + oy) * a.Wout + ox) * a.Cout);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                            }
                            if (a.act) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
                            }
                            f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                            const elem2 plo = __builtin_convertvector(lo, elem2), phi = __builtin_convertvector(hi2, elem2);
                            *reinterpret_cast<elem4*>(a.y + po) = elem4{plo[0], plo[1], phi[0], phi[1]};
                        }
                    }
                }
                continue;
            }
            f32x16 acc[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll CP == 128 ? 1 : 9                                            // (128 channels: eight K chunks per tap are enough to overlap; all 72 unrolled spill)
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                const int wrow = (WR < 32 ? (w_live ? l31 : 0) : l31) * 9 + tap;
                int prow[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) prow[m] = ((trow0 + 2 * m) * STRIDE + dy) * C::PW + tcol * STRIDE + dx;
#pragma unroll
                for (int kk = 0; kk < CP / 16; ++kk) {
                    const int ch = 2 * kk + hi;
                    elem8 wf = *reinterpret_cast<const elem8*>(wl + wrow * C::RB + ((ch ^ C::swz(wrow)) << 4));
                    if constexpr (WR < 32) {
                        if (!w_live) wf = elem8{};
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
#if defined(DM_ABL_SMALL_NOMMA)
                        if (CP == 128 && kk) continue;
#endif
                        const elem8 pf = *reinterpret_cast<const elem8*>(patch + prow[m] * C::RB + ((ch ^ C::swz(prow[m])) << 4));
                        acc[m] = DM_MFMA_32x32x16(wf, pf, acc[m]);
                    }
                }
            }
            // ---- epilogue: register r of a lane = channel cb*32 + (r & 3) + 8 (r >> 2) + 4 hi of its pixel
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int oy = oy0 + trow0 + 2 * m, ox = ox0 + tcol;
                const bool pix_ok = oy < a.Hout && ox < a.Wout;
                const long long po = (((long long)b * a.Hout + oy) * a.Wout + ox) * a.Cout;
                const long long pr = (((long long)(b % a.res_B) * a.Hout + oy) * a.Wout + ox) * a.Cout;
#if defined(__HIP_DEVICE_COMPILE__)
                if (wide_store) {
                    // 16-byte stores (Cout % 32 == 0): the two halves of the wave exchange one dword pair per channel octet
                    // (v_permlane32_swap, as csrc/conv.hip): a lane then holds 8 consecutive channels of its pixel
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int g = 2 * gp + q;
                            const int c0 = cb * 32 + 8 * g + 4 * hi;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[m][4 * g + e] + sbias[c0 + e];
                            if (a.res && pix_ok && c0 < a.Cout) {
                                const elem4 rv = *reinterpret_cast<const elem4*>(a.res + pr + c0);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                            }
                            if (a.act) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
                            }
                            f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                            w[q][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, elem2));
                            w[q][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, elem2));
                        }
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                        const int c8 = cb * 32 + 16 * gp + 8 * hi;
                        if (pix_ok && c8 < a.Cout) *reinterpret_cast<u32x4s*>(a.y + po + c8) = u32x4s{s0[0], s1[0], s0[1], s1[1]};
                    }
                    continue;
                }
#endif
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = 8 * g + 4 * hi, c0 = cb * 32 + cl;     // four consecutive channels
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[m][4 * g + e] + sbias[c0 + e];
                    const bool ok = pix_ok && c0 < a.Cout;                 // (Cout % 4 == 0: a run is inside or outside as a whole)
                    if (a.res && ok) {
                        const elem4 rv = *reinterpret_cast<const elem4*>(a.res + pr + c0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    if (a.act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
                    }
                    if (ok) {
                        f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                        const elem2 plo = __builtin_convertvector(lo, elem2), phi = __builtin_convertvector(hi2, elem2);
                        const elem4 o = {plo[0], plo[1], phi[0], phi[1]};
                        *reinterpret_cast<elem4*>(a.y + po + c0) = o;
                    }
                }
            }
        }
    }
}

template <int CP, int STRIDE, int TH = 16, int WR = 32>
int launch_patch(const SmallConvArgs& a, int cin, hipStream_t stream) {
    using C = PatchCfg<CP, STRIDE, TH, WR>;
    if (WR < 32 && a.Cout != WR) return DM_ERR_UNSUPPORTED;                                                 // (the 4-channel form stores whole 8-byte pixels)
    {   // 32-bit buffer offsets of the patch loads: an input beyond 4 GB (16 x 1024^2 x 128 channels) goes image group by image group
        const long long per_img = (long long)a.Hin * a.Win * cin * 2;
        if (per_img > 0xffffff00LL) return DM_ERR_UNSUPPORTED;
        if ((long long)a.B * per_img > 0xffffff00LL) {
            long long nb = 0xffffff00LL / per_img;
            if (a.res && a.res_B > 1) nb -= nb % a.res_B;                 // (image b takes residual image b % res_B: groups start on a multiple)
            if (nb <= 0) return DM_ERR_UNSUPPORTED;
            for (long long b0 = 0; b0 < a.B; b0 += nb) {
                SmallConvArgs c = a;
                c.B = (int)std::min<long long>(nb, a.B - b0);
                c.x = a.x + b0 * a.Hin * a.Win * cin;
                c.y = a.y + b0 * a.Hout * a.Wout * a.Cout;
                c.n_pix = (long long)c.B * a.Hout * a.Wout;
                const int rc = launch_patch<CP, STRIDE, TH, WR>(c, cin, stream);
                if (rc != DM_OK) return rc;
            }
            return DM_OK;
        }
    }
    const int n_cb = (a.Cout + 31) / 32;
    const size_t fixed = (size_t)C::PATCH_BYTES + (size_t)n_cb * 32 * 4;
    static_assert(C::PATCH_BYTES + C::W_BYTES + 128 <= 160 * 1024, "LDS budget");
    if (fixed + C::W_BYTES > 160 * 1024) return DM_ERR_UNSUPPORTED;
    // all blocks resident when that leaves room for two workgroups per CU (or fits at all for a single one), else one block
    int w_res = 1;
    if (fixed + (size_t)n_cb * C::W_BYTES <= 80 * 1024 || (fixed + (size_t)n_cb * C::W_BYTES <= 160 * 1024 && n_cb > 1)) w_res = n_cb;
    const size_t lds = fixed + (size_t)w_res * C::W_BYTES;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_patch<CP, STRIDE, TH, WR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    const int tiles_x = (a.Wout + C::TW - 1) / C::TW, tiles_y = (a.Hout + C::TH - 1) / C::TH;
    const long long n_tiles = (long long)a.B * tiles_x * tiles_y;
    if (n_tiles > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop; int dev = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
    const long long blocks = std::min<long long>(n_tiles, (long long)n_cu * wg_per_cu);
    DM_ENTER();
    hipLaunchKernelGGL((k_conv3x3_patch<CP, STRIDE, TH, WR>), dim3((unsigned)blocks), dim3(256), lds, stream, a, cin, tiles_x, tiles_y, (int)n_tiles, w_res);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// the patch kernel when the shape is inside its domain: stride 1 | 2, Cin even and <= 32, or Cin = 128 with Cout <= 32 (stride 1), Cout % 4 == 0,
// the input patch of a tile inside 2^31 elements.  DREAMMAT_STEM_KERNEL=direct keeps the direct kernel (A/B runs).
int try_patch(const SmallConvArgs& a, int cin, hipStream_t stream) {
    static const bool direct = getenv("DREAMMAT_STEM_KERNEL") && getenv("DREAMMAT_STEM_KERNEL")[0] == 'd';
    if (direct || (cin & 1) || a.Cout % 4 != 0 || (a.stride != 1 && a.stride != 2)) return DM_ERR_UNSUPPORTED;
    if ((((uintptr_t)a.y | (uintptr_t)a.res) & 7) || ((uintptr_t)a.bias & 1)) return DM_ERR_UNSUPPORTED;
    if (cin > 32 && ((((uintptr_t)a.x | (uintptr_t)a.w) & 15) || cin != 128 || a.Cout > 32)) return DM_ERR_UNSUPPORTED;
    if (a.stride == 1) {
        if (cin <= 16) return launch_patch<16, 1>(a, cin, stream);
        if (cin <= 32) return launch_patch<32, 1>(a, cin, stream);
        if (a.Cout == 4) {                                       // (the image gradient of the VAE encoder's conv_in)
            const int rc = launch_patch<128, 1, 8, 4>(a, cin, stream);
            if (rc != DM_ERR_UNSUPPORTED) return rc;
        }
        return launch_patch<128, 1>(a, cin, stream);
    } else {
        if (cin <= 16) return launch_patch<16, 2>(a, cin, stream);
        if (cin <= 32) return launch_patch<32, 2>(a, cin, stream);
    }
    return DM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

// act: 0 none, 1 SiLU (applied to conv + bias (+ residual) before the rounding to bf16).  Cin even and <= 32, or 128 at stride 1; Cout % 4 == 0
// (the patch kernel; the direct kernel behind it: Cin in {4, 8, 16, 22, 32}, Cout % 16 == 0).
// residual (may be NULL): [res_B, Hout, Wout, Cout] bf16 added in the same pass, image b takes residual image b % res_B.
int DM_T(dm_conv3x3_small_res_nhwc_, )(const void* x, const void* w, const void* bias, const void* residual, int res_B, void* y, int B,
                                   int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act,
                                   hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return DM_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w) & 3) || ((uintptr_t)y & 7)) return DM_ERR_UNSUPPORTED;
    if (residual && (res_B <= 0 || ((uintptr_t)residual & 7))) return DM_ERR_ARG;
    SmallConvArgs a;
    a.x = (const elem_t*)x; a.w = (const elem_t*)w; a.bias = (const elem_t*)bias; a.y = (elem_t*)y;
    a.res = (const elem_t*)residual; a.res_B = residual ? res_B : 1;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.stride = stride;
    a.pad_y = pad_y; a.pad_x = pad_x; a.act = act;
    a.n_pix = (long long)B * Hout * Wout;
    {
        const int rc = try_patch(a, Cin, stream);
        if (rc != DM_ERR_UNSUPPORTED) return rc;
    }
    if (Cout % CO != 0 || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15)) return DM_ERR_UNSUPPORTED;      // the direct kernel's domain
    switch (Cin) {
    case 4: return launch_small<4>(a, stream);
    case 8: return launch_small<8>(a, stream);
    case 16: return launch_small<16>(a, stream);
    case 22: return launch_small<22>(a, stream);
    case 32: return launch_small<32>(a, stream);
    default: return DM_ERR_UNSUPPORTED;
    }
}

int DM_T(dm_conv3x3_small_nhwc_, )(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                               int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act, hipStream_t stream) {
    return DM_T(dm_conv3x3_small_res_nhwc_, )(x, w, bias, nullptr, 0, y, B, Hin, Win, Cin, Hout, Wout, Cout, stride, pad_y, pad_x, act, stream);
}

}  // extern "C"
