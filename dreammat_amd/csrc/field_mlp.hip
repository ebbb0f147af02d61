// The feature network of DreamMatMesh.forward (threestudio/models/geometry/dreammat_mesh.py:246-254 -> networks.py:150-187,
// VanillaMLP: bias-free Linear(2L, 64) -> ReLU -> Linear(64, n_feature_dims), fp32) as two fused kernels over the
// feature-major encodings the hash-grid kernels write ([2L, M], one coalesced row per feature).
//
// Why: through three fp32 GEMMs (hipBLASLt: a [64 x 32] weight against 2.4 M points is a skinny problem it runs at ~16 TF/s),
// a ReLU pass and a 614 MB hidden activation written and read back, forward + backward cost ~2.5 ms of the 77 ms step.  Here
// the hidden layer never leaves the registers:
//   forward : one lane = one point; x (2L values) in registers, for each hidden unit j a dot product against W1[j] (LDS
//             broadcast reads), ReLU, and the n_out multiply-adds into y -- 2L * 64 + 64 * n_out FMAs per point;
//   backward: the same recomputation, dh = (pre > 0) * W2^T dy, dx = W1^T dh in registers; the weight gradients
//             dW1 = dH^T X and dW2 = dY^T H are sums over the points, i.e. GEMMs with K = 64 points per wave batch, done on the
//             matrix pipe in fp32 (v_mfma_f32_32x32x2_f32) from LDS-staged, transposed copies of dh / h / x / dy, accumulated
//             in registers across the wave's batches and added to the gradient buffers once per wave at the end.
// fp32 throughout (the reference's field is fp32 torch; parity with the oracle 1e-5).
#include <algorithm>

#include "dm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHidden = 64;
constexpr int kMaxOut = 8;

struct MlpArgs {
    const float* x; long long x_fs;            // [IN][M]: feature stride (points contiguous)
    const float* w1; const float* w2;          // [64][IN], [OUT][64]
    float* y; long long y_fs;                  // forward: [OUT][M]
    const float* dy; long long dy_rs, dy_cs;   // backward: dy[m * rs + k * cs]
    float* dx; long long dx_fs;                // [IN][M]
    float* dw1; float* dw2;                    // ADDED into
    long long M;
    int n_out;
};

template <int IN>
__global__ __launch_bounds__(256) void k_field_mlp_fwd(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float sW1[kHidden * IN];
    __shared__ __attribute__((aligned(16))) float sW2[kHidden * kMaxOut];      // [j][k], zero beyond n_out
    for (int i = threadIdx.x; i < kHidden * IN; i += 256) sW1[i] = a.w1[i];
    for (int i = threadIdx.x; i < kHidden * kMaxOut; i += 256) {
        const int j = i / kMaxOut, k = i % kMaxOut;
        sW2[i] = k < a.n_out ? a.w2[k * kHidden + j] : 0.f;
    }
    __syncthreads();
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < a.M; m += (long long)gridDim.x * 256) {
        float x[IN];
#pragma unroll
        for (int f = 0; f < IN; ++f) x[f] = a.x[f * a.x_fs + m];
        float y[kMaxOut];
#pragma unroll
        for (int k = 0; k < kMaxOut; ++k) y[k] = 0.f;
#pragma unroll 4
        for (int j = 0; j < kHidden; ++j) {
            const float4* w = reinterpret_cast<const float4*>(sW1 + j * IN);
            float p0 = 0.f, p1 = 0.f;                                  // two chains: the adds of one dot product are dependent
#pragma unroll
            for (int q = 0; q < IN / 4; ++q) {
                const float4 v = w[q];
                p0 = fmaf(v.x, x[4 * q], p0); p1 = fmaf(v.y, x[4 * q + 1], p1);
                p0 = fmaf(v.z, x[4 * q + 2], p0); p1 = fmaf(v.w, x[4 * q + 3], p1);
            }
            const float h = fmaxf(p0 + p1, 0.f);
            const float4 u0 = reinterpret_cast<const float4*>(sW2 + j * kMaxOut)[0], u1 = reinterpret_cast<const float4*>(sW2 + j * kMaxOut)[1];
            y[0] = fmaf(u0.x, h, y[0]); y[1] = fmaf(u0.y, h, y[1]); y[2] = fmaf(u0.z, h, y[2]); y[3] = fmaf(u0.w, h, y[3]);
            y[4] = fmaf(u1.x, h, y[4]); y[5] = fmaf(u1.y, h, y[5]); y[6] = fmaf(u1.z, h, y[6]); y[7] = fmaf(u1.w, h, y[7]);
        }
#pragma unroll
        for (int k = 0; k < kMaxOut; ++k)
            if (k < a.n_out) a.y[k * a.y_fs + m] = y[k];
    }
}

// LDS per wave (floats): x [64][33] | dh chunk [64][33] | h chunk [64][33] | dy [64][9]     (odd row pitch: a lane = a point
// writes down a column, the matrix-pipe operand reads run along a row)
constexpr int kLd = 33, kLdY = 9;
constexpr int kWaveLds = 3 * 64 * kLd + 64 * kLdY;

template <int IN>
__global__ __launch_bounds__(256) void k_field_mlp_bwd(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // sW1 | sW2 ([j][k]) | 4 x per-wave stage
    float* sW1 = smem;
    float* sW2 = smem + kHidden * IN;
    float* stage = sW2 + kHidden * kMaxOut;
    for (int i = threadIdx.x; i < kHidden * IN; i += 256) sW1[i] = a.w1[i];
    for (int i = threadIdx.x; i < kHidden * kMaxOut; i += 256) {
        const int j = i / kMaxOut, k = i % kMaxOut;
        sW2[i] = k < a.n_out ? a.w2[k * kHidden + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    float* xs = stage + wave * kWaveLds;
    float* dhs = xs + 64 * kLd;
    float* hs = dhs + 64 * kLd;
    float* dys = hs + 64 * kLd;
    f32x16 acc1[2], acc2[2];                                           // dW1 / dW2 tiles of hidden units [32c, 32c + 32)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[c][r] = 0.f; acc2[c][r] = 0.f; }
    const long long n_batches = (a.M + 63) / 64;
    for (long long b = (long long)blockIdx.x * 4 + wave; b < n_batches; b += (long long)gridDim.x * 4) {
        const long long m = b * 64 + lane;
        const bool ok = m < a.M;
        float x[IN], dx[IN], dy[kMaxOut];
#pragma unroll
        for (int f = 0; f < IN; ++f) { x[f] = ok ? a.x[f * a.x_fs + m] : 0.f; dx[f] = 0.f; xs[lane * kLd + f] = x[f]; }
#pragma unroll
        for (int k = 0; k < kMaxOut; ++k) { dy[k] = (ok && k < a.n_out) ? a.dy[m * a.dy_rs + k * a.dy_cs] : 0.f; dys[lane * kLdY + k] = dy[k]; }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll 4
            for (int jj = 0; jj < 32; ++jj) {
                const int j = 32 * c + jj;
                const float4* w = reinterpret_cast<const float4*>(sW1 + j * IN);
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int q = 0; q < IN / 4; ++q) {
                    const float4 v = w[q];
                    p0 = fmaf(v.x, x[4 * q], p0); p1 = fmaf(v.y, x[4 * q + 1], p1);
                    p0 = fmaf(v.z, x[4 * q + 2], p0); p1 = fmaf(v.w, x[4 * q + 3], p1);
                }
                const float pre = p0 + p1;
                const float4 u0 = reinterpret_cast<const float4*>(sW2 + j * kMaxOut)[0], u1 = reinterpret_cast<const float4*>(sW2 + j * kMaxOut)[1];
                float g = u0.x * dy[0];
                g = fmaf(u0.y, dy[1], g); g = fmaf(u0.z, dy[2], g); g = fmaf(u0.w, dy[3], g);
                g = fmaf(u1.x, dy[4], g); g = fmaf(u1.y, dy[5], g); g = fmaf(u1.z, dy[6], g); g = fmaf(u1.w, dy[7], g);
                const float dh = pre > 0.f ? g : 0.f;                   // threshold_backward: the gradient passes where the output is > 0
                hs[lane * kLd + jj] = fmaxf(pre, 0.f);
                dhs[lane * kLd + jj] = dh;
#pragma unroll
                for (int q = 0; q < IN / 4; ++q) {
                    const float4 v = w[q];
                    dx[4 * q] = fmaf(v.x, dh, dx[4 * q]); dx[4 * q + 1] = fmaf(v.y, dh, dx[4 * q + 1]);
                    dx[4 * q + 2] = fmaf(v.z, dh, dx[4 * q + 2]); dx[4 * q + 3] = fmaf(v.w, dh, dx[4 * q + 3]);
                }
            }
            // weight gradients of this chunk on the matrix pipe: K = the 64 points of the batch, two per instruction.
            //   dW1[32c + i][f] += sum_pt dh[pt][i] x[pt][f]      A = dh^T, B = x
            //   dW2[k][32c + jl] += sum_pt dy[pt][k] h[pt][jl]    A = dy^T (rows >= n_out are zero), B = h
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 8
            for (int s = 0; s < 32; ++s) {
                const int pt = 2 * s + hi;
                const float a1 = dhs[pt * kLd + l31];
                const float b1 = l31 < IN ? xs[pt * kLd + l31] : 0.f;
                acc1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1[c], 0, 0, 0);
                const float a2 = l31 < kMaxOut ? dys[pt * kLdY + l31] : 0.f;
                const float b2 = hs[pt * kLd + l31];
                acc2[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc2[c], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();                           // (the next chunk / batch rewrites the stage)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (ok) {
#pragma unroll
            for (int f = 0; f < IN; ++f) a.dx[f * a.dx_fs + m] = dx[f];
        }
    }
    // accumulator element (row i, column j): lane = j + 32 * ((i / 4) % 2), register = i % 4 + 4 * (i / 8)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 4 * hi + 8 * (r >> 2);
            if (l31 < IN) atomicAdd(a.dw1 + (32 * c + i) * IN + l31, acc1[c][r]);              // dW1[32c + i][f = l31]
            if (i < a.n_out) atomicAdd(a.dw2 + i * kHidden + 32 * c + l31, acc2[c][r]);        // dW2[k = i][32c + l31]
        }
}

bool mlp_ok(const MlpArgs& a, int IN) { return a.M > 0 && (IN == 16 || IN == 32) && a.n_out >= 1 && a.n_out <= kMaxOut; }

}  // namespace

extern "C" {

// y[k][m] = sum_j w2[k][j] relu(sum_f w1[j][f] x[f][m]).  x [IN][M] by feature stride (points contiguous), IN = 16 | 32,
// 64 hidden units, n_out <= 8; y [n_out][M] by feature stride.  fp32.
int dm_field_mlp_fwd(const float* x, long long x_fs, long long M, const float* w1, const float* w2, int n_in, int n_out, float* y,
                     long long y_fs, hipStream_t stream) {
    MlpArgs a = {};
    a.x = x; a.x_fs = x_fs; a.w1 = w1; a.w2 = w2; a.y = y; a.y_fs = y_fs; a.M = M; a.n_out = n_out;
    if (!x || !w1 || !w2 || !y) return DM_ERR_ARG;
    if (!mlp_ok(a, n_in)) return DM_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)std::min<long long>((M + 255) / 256, 256 * 8);
    DM_ENTER();
    if (n_in == 32) hipLaunchKernelGGL(k_field_mlp_fwd<32>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_field_mlp_fwd<16>, dim3(grid), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Backward of the above: dx [IN][M] (written), dw1 [64][IN] and dw2 [n_out][64] (ADDED into; float atomics, one set per wave).
// dy[m * dy_rs + k * dy_cs].
int dm_field_mlp_bwd(const float* x, long long x_fs, long long M, const float* w1, const float* w2, int n_in, int n_out, const float* dy,
                     long long dy_rs, long long dy_cs, float* dx, long long dx_fs, float* dw1, float* dw2, hipStream_t stream) {
    MlpArgs a = {};
    a.x = x; a.x_fs = x_fs; a.w1 = w1; a.w2 = w2; a.dy = dy; a.dy_rs = dy_rs; a.dy_cs = dy_cs; a.dx = dx; a.dx_fs = dx_fs;
    a.dw1 = dw1; a.dw2 = dw2; a.M = M; a.n_out = n_out;
    if (!x || !w1 || !w2 || !dy || !dx || !dw1 || !dw2) return DM_ERR_ARG;
    if (!mlp_ok(a, n_in)) return DM_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)std::min<long long>(((M + 63) / 64 + 3) / 4, 256);
    const size_t lds = (size_t)(kHidden * n_in + kHidden * kMaxOut + 4 * kWaveLds) * sizeof(float);      // ~120 KB: one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    DM_ENTER();
    if (n_in == 32) hipLaunchKernelGGL(k_field_mlp_bwd<32>, dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(k_field_mlp_bwd<16>, dim3(grid), dim3(256), lds, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
