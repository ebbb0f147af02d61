// Implicit-GEMM 3x3 convolution (NHWC, bf16 in/out, fp32 accumulate) on MFMA for gfx950.
//
// Why it exists: the UNet / ControlNet / VAE-encoder convolutions that diffusers runs for
// threestudio/models/guidance/dreammat_guidance.py:205-292 are ~80 % of the FLOPs of an SDS step.
// MIOpen ships no gfx950 kernel database in this ROCm image (every conv shape JIT-compiles, ~25 min
// cold start) and an im2col + GEMM lowering moves 9x the activation bytes through HBM.  This kernel
// reads each activation tile straight from the NHWC tensor:
//
//   y[p, n] = bias[n] + sum_{tap, c} x[pixel(p) + tap, c] * w[n, tap, c]         (p = b*Ho*Wo + yo*Wo + xo)
//
// as one GEMM with M = pixels, N = Cout, K = 9*Cin:  A rows are gathered per tap (zero-filled outside
// the image), B = weights stored [Cout][tap][Cin] (K contiguous).  Workgroup tile 128 x BN x 32,
// 4 waves (2x2), v_mfma_f32_32x32x16_bf16, register-staged double-buffered LDS with 16 B row padding
// (conflict-free ds_read_b128), one barrier per K-step.  The same kernel computes the data gradient
// (the VAE encoder is differentiated through): dx = conv(dy, w') with w' = taps flipped, Cin<->Cout
// swapped (prepared once on the host side).  Requirements: Cin % 32 == 0, Cout % 64 == 0.
#include "dm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvArgs {
    const __bf16* x;     // [B, Hin, Win, Cin]
    const __bf16* w;     // [Cout, 9, Cin]
    const __bf16* bias;  // [Cout] or null
    __bf16* y;           // [B, Hout, Wout, Cout]
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int stride, pad_y, pad_x;
    long long M;         // B*Hout*Wout
};

constexpr int BM = 128, BK = 32;
constexpr int ROWB = BK * 2 + 16;     // LDS bytes per tile row (32 bf16 + 16 B pad)

template <int BN>
__global__ __launch_bounds__(256) void k_conv3x3(ConvArgs a) {
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int NT = BN / 64;                       // 32-wide n tiles per wave (wave tile = 64 x BN/2)
    constexpr int B_CHUNKS = BN * 4 / 256;            // 16 B chunks of the B tile per thread (BN=128: 2, 64: 1)
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread gather coordinates of its two A rows (row = tid/4 and 64 + tid/4), chunk = tid%4
    const int a_chunk = tid & 3;
    int a_b[2], a_y[2], a_x[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + (tid >> 2) + 64 * i;
        a_ok[i] = m < a.M;
        long long mm = a_ok[i] ? m : 0;
        int hw = a.Hout * a.Wout;
        a_b[i] = (int)(mm / hw);
        int rem = (int)(mm - (long long)a_b[i] * hw);
        int yo = rem / a.Wout;
        a_y[i] = yo * a.stride - a.pad_y;
        a_x[i] = (rem - yo * a.Wout) * a.stride - a.pad_x;
    }
    const int kt_per_tap = a.Cin / BK;
    const int n_steps = 9 * kt_per_tap;
    const long long Kw = 9LL * a.Cin;

    uint4 areg[2], breg[B_CHUNKS];
    auto load_step = [&](int s) {
        int tap = s / kt_per_tap;
        int c0 = (s - tap * kt_per_tap) * BK + a_chunk * 8;
        int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int yy = a_y[i] + dy, xx = a_x[i] + dx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_ok[i] && (unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win)
                v = *reinterpret_cast<const uint4*>(a.x + (((long long)a_b[i] * a.Hin + yy) * a.Win + xx) * a.Cin + c0);
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            int c = tid + 256 * i;
            int row = c >> 2, ch = c & 3;
            breg[i] = *reinterpret_cast<const uint4*>(a.w + (long long)(n0 + row) * Kw + (long long)s * BK + ch * 8);
        }
    };
    auto write_step = [&](int buf) {
        char* ab = smem + buf * (A_BYTES + B_BYTES);
        char* bb = ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint4*>(ab + ((tid >> 2) + 64 * i) * ROWB + a_chunk * 16) = areg[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            int c = tid + 256 * i;
            *reinterpret_cast<uint4*>(bb + (c >> 2) * ROWB + (c & 3) * 16) = breg[i];
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_step(0);
    write_step(0);
    __syncthreads();
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps) load_step(s + 1);
        const char* ab = smem + buf * (A_BYTES + B_BYTES);
        const char* bb = ab + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[2], bf[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(ab + (64 * wm + 32 * i + l31) * ROWB + 32 * kk + 16 * hi);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(bb + ((BN / 2) * wn + 32 * j + l31) * ROWB + 32 * kk + 16 * hi);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < n_steps) write_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row = m][col = n]: lane holds col n = l31, rows (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + (BN / 2) * wn + 32 * j + l31;
        float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < a.M) a.y[m * a.Cout + n] = (__bf16)(acc[i][j][r] + bv);
            }
    }
}

}  // namespace

extern "C" {

// x [B,Hin,Win,Cin] NHWC bf16; w [Cout,3,3,Cin] (= [Cout, 9*Cin], tap-major) bf16; bias [Cout] bf16 or NULL;
// y [B,Hout,Wout,Cout] NHWC bf16 with Hout = (Hin + pad_y + pad_y_end - 3)/stride + 1 chosen by the caller
// (pad_y / pad_x are the leading pads; trailing pads are implied by Hout/Wout and zero-filled).
int dm_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                         int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return DM_ERR_ARG;
    if (Cin % 32 != 0 || Cout % 64 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return DM_ERR_ARG;
    ConvArgs a;
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const __bf16*)bias; a.y = (__bf16*)y;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.stride = stride; a.pad_y = pad_y; a.pad_x = pad_x;
    a.M = (long long)B * Hout * Wout;
    long long mt = (a.M + BM - 1) / BM;
    if (mt > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    if (Cout % 128 == 0) {
        hipLaunchKernelGGL(k_conv3x3<128>, dim3((unsigned)mt, Cout / 128), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_conv3x3<64>, dim3((unsigned)mt, Cout / 64), dim3(256), 0, stream, a);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
