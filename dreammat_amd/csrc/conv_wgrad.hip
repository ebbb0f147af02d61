// Weight gradient of the 3x3 convolutions of the TRAINABLE ControlNet copy (controlnet_train/diffusers_train_controlnet.py:
// 858-915: loss.backward() through every ResnetBlock2D / Downsample2D conv), bf16 NHWC in, fp32 out, MFMA.
//
//   dW[co][ty][tx][ci] = sum over (b, y, x) of dY[b, y, x, co] . X[b, s.y + ty - 1, s.x + tx - 1, ci]        (s = stride, pad 1)
//
// = 9 GEMMs with M = Cout, N = Cin and the PIXELS as the contraction index -- both operands arrive "transposed" (memory is
// channel-contiguous, the MFMA wants 8 consecutive k per lane).  No im2col buffer and no transposed copies: the tiles are
// staged row-major ([pixel][64 channels], 192-byte pitch) and the fragments are read with gfx950's transposing LDS read
// (ds_read_b64_tr_b16): inside a group of 16 lanes, lane i passes the address of 4 consecutive channels of pixel k0 + (i >> 2)
// and receives channel i of pixels k0 .. k0 + 3 (measured semantics: tools/tr_probe.cpp).  Every lane addresses its own
// pixel, so the halo, the stride and the tap shift are plain address arithmetic on the staged input rows.
//
// Workgroup = 4 waves = one 64 (co) x 64 (ci) tile of all 9 taps (wave: 32 x 32 x 9 taps = 144 accumulator registers) over a
// slab of 64-pixel chunks; grid.z splits the pixels, each split writes its own fp32 partial [split][Cout][9][Cin] (the caller
// adds the partials: fixed order, no atomics).
#include "dm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPitch = 192;     // bytes per staged pixel (64 channels + 64 B pad: the 4 pixels x 2 column groups of a transposing read hit 8 disjoint bank octets)
constexpr int kChunk = 64;      // output pixels per step

struct WgradArgs {
    const __bf16* x; const __bf16* dy; float* part;
    int B, H, W, Cin, Ho, Wo, Cout, stride;
    int rows_per_chunk, rows_in, cols_in;    // output rows per chunk; staged input rows / columns (with halo)
    int n_chunks, chunks_per_split;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)const_cast<char*>(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)const_cast<char*>(p1));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) void k_conv3x3_wgrad(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dy_img = smem;                          // [64 pixels][kPitch]
    char* x_img = smem + kChunk * kPitch;         // [rows_in * cols_in pixels][kPitch]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int wm = wave >> 1, wn = wave & 1;      // the wave's 32-co / 32-ci block of the tile
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64, split = blockIdx.z;

    // transposing-read addresses of this lane (as a SOURCE lane): pixel 16 ks + 8 hi + 4 r + (i >> 2), 4 channels at 4 (i & 3)
    int off_a[4][2], off_b[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int p = 16 * ks + 8 * hi + 4 * r + (i16 >> 2);
            off_a[ks][r] = p * kPitch + (wm * 32 + 16 * g1 + 4 * (i16 & 3)) * 2;
            const int yl = p / a.Wo, xl = p - yl * a.Wo;
            off_b[ks][r] = ((yl * a.stride) * a.cols_in + xl * a.stride) * kPitch + (wn * 32 + 16 * g1 + 4 * (i16 & 3)) * 2;
        }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int c_begin = split * a.chunks_per_split;
    const int c_end = min(a.n_chunks, c_begin + a.chunks_per_split);
    const int chunks_per_image = a.Ho / a.rows_per_chunk;
    const int n_x = a.rows_in * a.cols_in * 8;            // 16-byte pieces of the staged input window
    for (int c = c_begin; c < c_end; ++c) {
        const int b = c / chunks_per_image, y0 = (c - b * chunks_per_image) * a.rows_per_chunk;
        __syncthreads();                                  // the previous chunk's fragments are read
        {
            const __bf16* src = a.dy + ((long long)(b * a.Ho + y0) * a.Wo) * a.Cout + co0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 256 * k, px = idx >> 3, c8 = idx & 7;
                *reinterpret_cast<uint4*>(dy_img + px * kPitch + c8 * 16) =
                    *reinterpret_cast<const uint4*>(src + (long long)px * a.Cout + c8 * 8);
            }
        }
        for (int idx = tid; idx < n_x; idx += 256) {
            const int pix = idx >> 3, c8 = idx & 7;
            const int ry = pix / a.cols_in, rx = pix - ry * a.cols_in;
            const int yin = a.stride * y0 - 1 + ry, xin = rx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (yin >= 0 && yin < a.H && xin >= 0 && xin < a.W)
                v = *reinterpret_cast<const uint4*>(a.x + ((long long)(b * a.H + yin) * a.W + xin) * a.Cin + ci0 + c8 * 8);
            *reinterpret_cast<uint4*>(x_img + pix * kPitch + c8 * 16) = v;
        }
        __syncthreads();
        bf16x8 af[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[ks] = tr_frag(dy_img + off_a[ks][0], dy_img + off_a[ks][1]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* xt = x_img + ((t / 3) * a.cols_in + (t % 3)) * kPitch;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 bfrag = tr_frag(xt + off_b[ks][0], xt + off_b[ks][1]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfrag, acc[t], 0, 0, 0);
            }
        }
    }
    // C[m = co][n = ci]: lane = ci, register r = co row (r & 3) + 8 (r >> 2) + 4 hi
    float* out = a.part + (long long)split * a.Cout * 9 * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[((long long)co * 9 + t) * a.Cin + ci] = acc[t][r];
        }
}

}  // namespace

extern "C" {

// number of pixel splits dm_conv3x3_wgrad_nhwc_bf16 uses for a shape (= leading dimension of its partial buffer)
int dm_conv3x3_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || (Ho * Wo) % kChunk) return 0;
    const long long n_chunks = (long long)B * Ho * Wo / kChunk, tiles = (long long)(Cout / 64) * (Cin / 64);
    long long s = (1024 + tiles - 1) / tiles;             // ~4 workgroups per CU over the whole grid
    if (s > n_chunks) s = n_chunks;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    const long long per = (n_chunks + s - 1) / s;
    return (int)((n_chunks + per - 1) / per);
}

// dW of a 3x3 / pad 1 convolution with stride 1 or 2: x [B,H,W,Cin], dy [B,Ho,Wo,Cout] bf16 NHWC (Ho = (H - 1) / stride + 1),
// part [splits][Cout][3][3][Cin] fp32 with splits = dm_conv3x3_wgrad_splits(...): dW = sum over the leading dimension.
// Cin % 64 == 0, Cout % 64 == 0, Wo a power of two <= 64 and Ho * Wo % 64 == 0 (DM_ERR_UNSUPPORTED otherwise).
int dm_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* part, int B, int H, int W, int Cin, int Cout, int stride,
                               hipStream_t stream) {
    if (!x || !dy || !part || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DM_ERR_ARG;
    if (stride != 1 && stride != 2) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return DM_ERR_ARG;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (Cin % 64 || Cout % 64 || Wo > kChunk || (Wo & (Wo - 1)) || (Ho * Wo) % kChunk) return DM_ERR_UNSUPPORTED;
    WgradArgs a;
    a.x = (const __bf16*)x; a.dy = (const __bf16*)dy; a.part = part;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride;
    a.rows_per_chunk = kChunk / Wo;
    if (Ho % a.rows_per_chunk) return DM_ERR_UNSUPPORTED;
    a.rows_in = stride * (a.rows_per_chunk - 1) + 3;
    a.cols_in = stride * (Wo - 1) + 3;
    a.n_chunks = B * Ho * Wo / kChunk;
    const int splits = dm_conv3x3_wgrad_splits(B, Ho, Wo, Cin, Cout);
    a.chunks_per_split = (a.n_chunks + splits - 1) / splits;
    const int lds = (kChunk + a.rows_in * a.cols_in) * kPitch;
    if (lds > 160 * 1024) return DM_ERR_UNSUPPORTED;
    if ((long long)B * H * W * Cin >= (1LL << 31) || (long long)B * Ho * Wo * Cout >= (1LL << 31)) return DM_ERR_UNSUPPORTED;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        DM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    DM_ENTER();
    hipLaunchKernelGGL(k_conv3x3_wgrad, dim3(Cout / 64, Cin / 64, splits), dim3(256), lds, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
