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

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int CO = 16;                         // output channels per thread

struct SmallConvArgs {
    const __bf16* x; const __bf16* w; const __bf16* bias; __bf16* y;
    const __bf16* res; int res_B;              // optional [res_B, Hout, Wout, Cout] added before the rounding, image b takes b % res_B
    int B, Hin, Win, Hout, Wout, Cout, stride, pad_y, pad_x, act;
    long long n_pix;                           // B*Hout*Wout
};

template <int CIN>
__global__ __launch_bounds__(256) void k_conv3x3_small(SmallConvArgs a) {
    constexpr int CP = CIN / 2;                // channel pairs
    __shared__ bf16x2 wl[CO * 9 * CP];         // [co][tap][pair]
    const int cog = blockIdx.y;
    for (int i = threadIdx.x; i < CO * 9 * CP; i += 256)
        wl[i] = reinterpret_cast<const bf16x2*>(a.w + (long long)cog * CO * 9 * CIN)[i];
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
        const bf16x2* src = reinterpret_cast<const bf16x2*>(a.x + (((long long)b * a.Hin + yy) * a.Win + xx) * CIN);
        bf16x2 v[CP];
#pragma unroll
        for (int k = 0; k < CP; ++k) v[k] = src[k];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const bf16x2* wr = wl + (c * 9 + t) * CP;
#pragma unroll
            for (int k = 0; k < CP; ++k) acc[c] = __builtin_amdgcn_fdot2_f32_bf16(v[k], wr[k], acc[c], false);
        }
    }
    if (a.res) {                                // (ControlNet: conv_in(sample) + conditioning embedding, the embedding of the B views
        __bf16 rv[CO];                          //  shared by the text / negative / null branches)
        const long long pr = ((long long)(b % a.res_B) * a.Hout + yo) * a.Wout + xo;
        const uint4* rp = reinterpret_cast<const uint4*>(a.res + pr * a.Cout + cog * CO);
        *reinterpret_cast<uint4*>(rv) = rp[0];
        *reinterpret_cast<uint4*>(rv + 8) = rp[1];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] += (float)rv[c];
    }
    __bf16 o[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        float z = acc[c];
        if (a.act) z = z / (1.f + __expf(-z));
        o[c] = (__bf16)z;
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

}  // namespace

extern "C" {

// act: 0 none, 1 SiLU (applied to conv + bias (+ residual) before the rounding to bf16).  Cin in {4, 8, 16, 22, 32}.
// residual (may be NULL): [res_B, Hout, Wout, Cout] bf16 added in the same pass, image b takes residual image b % res_B.
int dm_conv3x3_small_res_nhwc_bf16(const void* x, const void* w, const void* bias, const void* residual, int res_B, void* y, int B,
                                   int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act,
                                   hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return DM_ERR_ARG;
    if (Cout % CO != 0 || (((uintptr_t)x | (uintptr_t)w) & 3) || ((uintptr_t)y & 15)) return DM_ERR_UNSUPPORTED;
    if (residual && (res_B <= 0 || ((uintptr_t)residual & 15))) return DM_ERR_ARG;
    SmallConvArgs a;
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const __bf16*)bias; a.y = (__bf16*)y;
    a.res = (const __bf16*)residual; a.res_B = residual ? res_B : 1;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.stride = stride;
    a.pad_y = pad_y; a.pad_x = pad_x; a.act = act;
    a.n_pix = (long long)B * Hout * Wout;
    switch (Cin) {
    case 4: return launch_small<4>(a, stream);
    case 8: return launch_small<8>(a, stream);
    case 16: return launch_small<16>(a, stream);
    case 22: return launch_small<22>(a, stream);
    case 32: return launch_small<32>(a, stream);
    default: return DM_ERR_UNSUPPORTED;
    }
}

int dm_conv3x3_small_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                               int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, int act, hipStream_t stream) {
    return dm_conv3x3_small_res_nhwc_bf16(x, w, bias, nullptr, 0, y, B, Hin, Win, Cin, Hout, Wout, Cout, stride, pad_y, pad_x, act, stream);
}

}  // extern "C"
