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
// fp32 throughout (the reference's field is fp32 torch; parity with the oracle 1e-5) -- in the forward and in the first backward
// formulation; the second one (round 6, below, the default) keeps fp32 values but multiplies them as split bf16 operands on the
// 16-bit matrix pipe.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "dm_common.h"
#include "dm_elem.h"

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

// ------------------------------------------------------------------------------------------------------------------------------
// Backward, second formulation (round 6): EVERY product on the 16-bit matrix pipe with split operands.
// The kernel above spends a 64-point batch on ~4800 dependent-latency fp32 FMAs fed by LDS broadcast reads, 128 fp32 MFMAs of 64
// cycles each and ~400 staging accesses, one wave per SIMD with nobody to switch to: 1.27 ms for 2.5 M points, 12 % of the fp32
// vector rate and 9x the 0.14 ms its 0.7 GB take.  Here a value a is carried as two bf16 numbers, a = hi + lo with hi = bf16(a),
// lo = bf16(a - hi) (16 significant bits, |a - hi - lo| <= 2^-18 |a|), and a product of two such operands is three
// v_mfma_f32_32x32x16_bf16 (hi.hi + hi.lo + lo.hi, fp32 accumulate; the dropped lo.lo term is <= 2^-18 |a| |b|): ~4e-6 relative
// per product, inside the 2e-5 gates of tests/test_hip_gpu.py::test_field_mlp_fused_vs_torch, at 16x the fp32 matrix rate / 3.
// All five products of a batch (transposed forms: the POINT is the lane of every accumulator, so loads, masks and stores are
// per point and coalesced):
//   pre^T [hid, pt] = W1 . X^T          A = W1 (static fragments), B = x of the lane's point (halves exchanged by v_permlane32_swap)
//   g^T   [hid, pt] = W2^T . dY^T       A = W2^T (static, K = n_out padded to 16), B = dy of the lane's point
//   dh^T = pre^T > 0 ? g^T : 0,  h^T = max(pre^T, 0)                     (in the accumulators)
//   dX^T  [f, pt]   = W1^T . dH^T       A = W1^T with its k (hidden) index in ACCUMULATOR ROW ORDER, B = dh^T straight from the registers
//   dW1 [hid, f] += dH^T . X,  dW2 [out, hid] += dY^T . H   contract over the points: both operands are read back TRANSPOSED
//                                       (ds_read_b64_tr_b16) from row-major LDS tiles [point][column] of one 32-point half
// 108 MFMAs of 32 cycles per 64 points; weight gradients summed in registers across a wave's batches, then across the
// workgroup's waves in LDS, then ONE set of global atomics per workgroup (the first formulation issued one set per wave:
// 2.4 M atomics at ~20 G/s = 0.12 ms of its own).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// slot e of k-half kh of a 16-wide k-step <-> row of a 32x32 accumulator block (register 8 j + e of lane-half kh is row 16 j + this)
__device__ __forceinline__ int acc_row(int kh, int e) { return 4 * kh + (e & 3) + 8 * (e >> 2); }
// (a, b) -> packed bf16 pairs hi | lo
__device__ __forceinline__ void split2(float a, float b, unsigned& ph, unsigned& pl) {
    const f32x2 v = {a, b};
    const elem2 h = __builtin_convertvector(v, elem2);
    const f32x2 r = {a - (float)h[0], b - (float)h[1]};
    const elem2 l = __builtin_convertvector(r, elem2);
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, l);
}
// (a, b) -> three packed bf16 pairs h | m | l with a = h + m + l to 24 bits (h, m are split2's hi, lo)
__device__ __forceinline__ void split3(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    const f32x2 v = {a, b};
    const elem2 h = __builtin_convertvector(v, elem2);
    const f32x2 r = {a - (float)h[0], b - (float)h[1]};
    const elem2 m = __builtin_convertvector(r, elem2);
    const f32x2 q = {r[0] - (float)m[0], r[1] - (float)m[1]};
    const elem2 l = __builtin_convertvector(q, elem2);
    ph = __builtin_bit_cast(unsigned, h);
    pm = __builtin_bit_cast(unsigned, m);
    pl = __builtin_bit_cast(unsigned, l);
}
struct Frag2 { u32x4v h, l; };       // one MFMA operand fragment (8 bf16 = 4 dwords) as hi and lo parts
__device__ __forceinline__ elem8 as_e8(const u32x4v v) { return __builtin_bit_cast(elem8, v); }
__device__ __forceinline__ void mfma3(f32x16& acc, const Frag2& a, const Frag2& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(a.h), as_e8(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(a.h), as_e8(b.l), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(a.l), as_e8(b.h), acc, 0, 0, 0);
}
// the same with 24-bit operands a = h + m + l: the six products down to 2^-24 (h.h + h.m + m.h + h.l + l.h + m.m).  Used where a SIGN
// is taken from the result: pre > 0 gates the gradient, and a 16-bit-operand pre flipped that test on ~5 of 65 k points (|pre| <
// 1e-5), each an O(1) error in dx and in the weight sums; at 24 bits the test flips as rarely as any fp32 summation order does.
__device__ __forceinline__ void mfma6(f32x16& acc, const Frag2& a, const u32x4v al, const Frag2& b, const u32x4v bl) {
    mfma3(acc, a, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(a.h), as_e8(bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(al), as_e8(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_e8(a.l), as_e8(b.l), acc, 0, 0, 0);
}
// fragment whose element e of lane-half kh is f(kh, e) (built once per workgroup: static weights)
template <class F>
__device__ __forceinline__ void make_frag(int hi, F&& f, u32x4v& h, u32x4v& m, u32x4v& l) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        unsigned ph, pm, pl;
        split3(f(hi, 2 * e2), f(hi, 2 * e2 + 1), ph, pm, pl);
        h[e2] = ph; m[e2] = pm; l[e2] = pl;
    }
}
// transposed operand fragment from a row-major bf16 tile [k rows][columns], RROW bytes per row: lane (column l31, half hi) gets
// rows 4 hi + {0..3, 8..11} of the 16-row k-step = accumulator row order (csrc/attn_bwd.hip tr_frag; tools/tr_probe.cpp)
template <int RROW>
__device__ __forceinline__ u32x4v tr_frag32(const char* p) {
    const elem4 lo = dm_ds_read_tr16_b64(p);
    const elem4 hi = dm_ds_read_tr16_b64(p + 8 * RROW);
    const elem8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4v, v);
}

constexpr int kRowH = kHidden * 2 + 16;          // tile row of 64 hidden units (bf16) + 16 B: conflict-free transposing reads
constexpr int kRowS = 32 * 2 + 16;               // tile row of 32 columns (features / outputs padded to 32)
constexpr int kTileH = 32 * kRowH, kTileS = 32 * kRowS;
constexpr int kWaveLds2 = 4 * kTileH + 4 * kTileS;       // dh hi | dh lo | h hi | h lo | x hi | x lo | dy hi | dy lo   (25.6 KB)

template <int IN>
__global__ __launch_bounds__(256) void k_field_mlp_bwd2(MlpArgs a) {
    constexpr int KS1 = IN / 16;                 // k-steps of the first product (features)
    extern __shared__ __attribute__((aligned(16))) char smem2[];       // 4 x per-wave tiles | workgroup sums of dW1, dW2 (fp32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    char* const tl = smem2 + wave * kWaveLds2;
    char* const t_dh[2] = {tl, tl + kTileH};
    char* const t_h[2] = {tl + 2 * kTileH, tl + 3 * kTileH};
    char* const t_x[2] = {tl + 4 * kTileH, tl + 4 * kTileH + kTileS};
    char* const t_dy[2] = {tl + 4 * kTileH + 2 * kTileS, tl + 4 * kTileH + 3 * kTileS};
    float* const wsum = reinterpret_cast<float*>(smem2 + 4 * kWaveLds2);          // [64 * IN] dW1 | [8 * 64] dW2
    for (int i = tid; i < kHidden * IN + kMaxOut * kHidden; i += 256) wsum[i] = 0.f;
    // the x / dy tiles' columns past IN / n_out are never written again: zero them once (rows of 32 columns are read whole)
    for (int i = lane; i < 4 * kTileS / 4; i += 64) reinterpret_cast<unsigned*>(t_x[0])[i] = 0u;

    // ---- static operand fragments (fp32 weights -> bf16 parts), once per workgroup, in LDS: fragment slot i = 1 KB, 16 bytes per lane
    // (identical for the four waves; in registers they were 96 of a wave's 512 and the loop spilled)
    //   slots 0-11: W1 [hb][ks] parts h, m, l      (A of pre^T: rows = hidden units 32 hb + l31, k = features 16 ks + 8 kh + e)
    //   slots 12-15: W2^T [hb] parts h, m           (A of g^T: k = outputs, k-half 1 zero)
    //   slots 16-23: W1^T [k4] parts h, m           (A of dX^T: rows = features, k = hidden units 16 k4 + accumulator row order)
    u32x4v* const sfr = reinterpret_cast<u32x4v*>(smem2 + 4 * kWaveLds2 + (kHidden * IN + kMaxOut * kHidden) * 4);
    auto sfrag = [&](int slot) __attribute__((always_inline)) { return sfr[slot * 64 + lane]; };
    if (wave == 0) {
        u32x4v h, m2, l;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                make_frag(hi, [&](int kh, int e) { return a.w1[(32 * hb + l31) * IN + 16 * ks + 8 * kh + e]; }, h, m2, l);
                sfr[((hb * 2 + ks) * 3 + 0) * 64 + lane] = h; sfr[((hb * 2 + ks) * 3 + 1) * 64 + lane] = m2; sfr[((hb * 2 + ks) * 3 + 2) * 64 + lane] = l;
            }
            make_frag(hi, [&](int kh, int e) { return (kh == 0 && e < a.n_out) ? a.w2[e * kHidden + 32 * hb + l31] : 0.f; }, h, m2, l);
            sfr[(12 + hb * 2) * 64 + lane] = h; sfr[(13 + hb * 2) * 64 + lane] = m2;
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            make_frag(hi, [&](int kh, int e) { return l31 < IN ? a.w1[(16 * k4 + acc_row(kh, e)) * IN + l31] : 0.f; }, h, m2, l);
            sfr[(16 + k4 * 2) * 64 + lane] = h; sfr[(17 + k4 * 2) * 64 + lane] = m2;
        }
    }
    const int tr_h = (4 * hi + ((lane & 15) >> 2)) * kRowH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int tr_s = (4 * hi + ((lane & 15) >> 2)) * kRowS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    f32x16 dw1[2], dw2[2];                 // dW1 rows [32 hb, +32) x features (lane) | dW2 outputs (rows) x hidden units [32 nb, +32) (lane)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dw1[i][r] = 0.f; dw2[i][r] = 0.f; }
    __syncthreads();

    const long long n_batches = (a.M + 63) / 64;
    for (long long b = (long long)blockIdx.x * 4 + wave; b < n_batches; b += (long long)gridDim.x * 4) {
        const long long m = b * 64 + lane;
        const bool ok = m < a.M;
        // ---- this lane's point: x, dy -> packed hi / lo pairs
        unsigned xh[IN / 2], xl[IN / 2], x3[IN / 2], dyh[4], dyl[4];       // (xh, xl) = the 16-bit pair, x3 = the third part (pre^T only)
#pragma unroll
        for (int f = 0; f < IN; f += 2) {
            const float x0 = ok ? a.x[f * a.x_fs + m] : 0.f, x1 = ok ? a.x[(f + 1) * a.x_fs + m] : 0.f;
            split3(x0, x1, xh[f / 2], xl[f / 2], x3[f / 2]);
        }
#pragma unroll
        for (int k = 0; k < kMaxOut; k += 2) {
            const float d0 = (ok && k < a.n_out) ? a.dy[m * a.dy_rs + k * a.dy_cs] : 0.f;
            const float d1 = (ok && k + 1 < a.n_out) ? a.dy[m * a.dy_rs + (k + 1) * a.dy_cs] : 0.f;
            split2(d0, d1, dyh[k / 2], dyl[k / 2]);
        }
        // B fragments of the two 32-point halves: lanes 0-31 carry k-half 0, lanes 32-63 k-half 1 of THE HALF'S points.
        // v_permlane32_swap(X = my k-half-0 dword, Y = my k-half-1 dword): [0] = fragment dword of points 0-31, [1] = of points 32-63
        Frag2 bx[2][KS1], bdy[2];
        u32x4v bx3[2][KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const auto sh = __builtin_amdgcn_permlane32_swap(xh[8 * ks + e2], xh[8 * ks + 4 + e2], false, false);
                const auto sl = __builtin_amdgcn_permlane32_swap(xl[8 * ks + e2], xl[8 * ks + 4 + e2], false, false);
                const auto s3 = __builtin_amdgcn_permlane32_swap(x3[8 * ks + e2], x3[8 * ks + 4 + e2], false, false);
                bx[0][ks].h[e2] = sh[0]; bx[1][ks].h[e2] = sh[1];
                bx[0][ks].l[e2] = sl[0]; bx[1][ks].l[e2] = sl[1];
                bx3[0][ks][e2] = s3[0]; bx3[1][ks][e2] = s3[1];
            }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {       // K = n_out <= 8: k-half 1 is zero
            const auto sh = __builtin_amdgcn_permlane32_swap(dyh[e2], 0u, false, false);
            const auto sl = __builtin_amdgcn_permlane32_swap(dyl[e2], 0u, false, false);
            bdy[0].h[e2] = sh[0]; bdy[1].h[e2] = sh[1];
            bdy[0].l[e2] = sl[0]; bdy[1].l[e2] = sl[1];
        }
        // ---- pre^T and g^T: [hidden block hb][point half pb], lane = point
        f32x16 pre[2][2], g[2][2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { pre[hb][pb][r] = 0.f; g[hb][pb][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const Frag2 aw = {sfrag((hb * 2 + ks) * 3), sfrag((hb * 2 + ks) * 3 + 1)};
                const u32x4v aw3 = sfrag((hb * 2 + ks) * 3 + 2);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) mfma6(pre[hb][pb], aw, aw3, bx[pb][ks], bx3[pb][ks]);
            }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const Frag2 aw = {sfrag(12 + hb * 2), sfrag(13 + hb * 2)};
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) mfma3(g[hb][pb], aw, bdy[pb]);
        }
        // h = relu(pre) (into pre), dh = pre > 0 ? g : 0 (into g): threshold_backward passes the gradient where the output is > 0
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = pre[hb][pb][r];
                    g[hb][pb][r] = p > 0.f ? g[hb][pb][r] : 0.f;
                    pre[hb][pb][r] = fmaxf(p, 0.f);
                }
        // ---- dX^T[f, pt] = W1^T . dH^T: registers 8 j .. 8 j + 7 of dh[hb] are the B fragment of k-step 2 hb + j
        f32x16 dxT[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dxT[pb][r] = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const Frag2 aw = {sfrag(16 + k4 * 2), sfrag(17 + k4 * 2)};
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                Frag2 bf;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned ph, pl;
                    split2(g[k4 >> 1][pb][8 * (k4 & 1) + 2 * e2], g[k4 >> 1][pb][8 * (k4 & 1) + 2 * e2 + 1], ph, pl);
                    bf.h[e2] = ph; bf.l[e2] = pl;
                }
                mfma3(dxT[pb], aw, bf);
            }
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const long long mp = b * 64 + 32 * pb + l31;
            if (mp < a.M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (f < IN) a.dx[f * a.dx_fs + mp] = dxT[pb][r];
                }
            }
        }
        // ---- weight gradients, one 32-point half at a time: tiles [point][column] -> transposed fragments
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            // dh / h of the half: this lane holds hidden units 32 hb + 8 q + 4 hi + {0..3} of point l31 in registers 4 q .. 4 q + 3
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned h0, l0, h1, l1;
                    const int off = l31 * kRowH + (32 * hb + 8 * q + 4 * hi) * 2;
                    split2(g[hb][pb][4 * q], g[hb][pb][4 * q + 1], h0, l0);
                    split2(g[hb][pb][4 * q + 2], g[hb][pb][4 * q + 3], h1, l1);
                    *reinterpret_cast<uint2*>(t_dh[0] + off) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(t_dh[1] + off) = make_uint2(l0, l1);
                    split2(pre[hb][pb][4 * q], pre[hb][pb][4 * q + 1], h0, l0);
                    split2(pre[hb][pb][4 * q + 2], pre[hb][pb][4 * q + 3], h1, l1);
                    *reinterpret_cast<uint2*>(t_h[0] + off) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(t_h[1] + off) = make_uint2(l0, l1);
                }
            if (hi == pb) {                    // x / dy rows: the half's own lanes
#pragma unroll
                for (int c = 0; c < IN / 8; ++c) {
                    *reinterpret_cast<uint4*>(t_x[0] + l31 * kRowS + 16 * c) = make_uint4(xh[4 * c], xh[4 * c + 1], xh[4 * c + 2], xh[4 * c + 3]);
                    *reinterpret_cast<uint4*>(t_x[1] + l31 * kRowS + 16 * c) = make_uint4(xl[4 * c], xl[4 * c + 1], xl[4 * c + 2], xl[4 * c + 3]);
                }
                *reinterpret_cast<uint4*>(t_dy[0] + l31 * kRowS) = make_uint4(dyh[0], dyh[1], dyh[2], dyh[3]);
                *reinterpret_cast<uint4*>(t_dy[1] + l31 * kRowS) = make_uint4(dyl[0], dyl[1], dyl[2], dyl[3]);
            }
            wave_sync();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // 32 points = two 16-wide k-steps
                Frag2 fx, fdy;
                fx.h = tr_frag32<kRowS>(t_x[0] + tr_s + ks * 16 * kRowS);
                fx.l = tr_frag32<kRowS>(t_x[1] + tr_s + ks * 16 * kRowS);
                fdy.h = tr_frag32<kRowS>(t_dy[0] + tr_s + ks * 16 * kRowS);
                fdy.l = tr_frag32<kRowS>(t_dy[1] + tr_s + ks * 16 * kRowS);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    Frag2 fdh, fh;
                    fdh.h = tr_frag32<kRowH>(t_dh[0] + tr_h + ks * 16 * kRowH + hb * 64);
                    fdh.l = tr_frag32<kRowH>(t_dh[1] + tr_h + ks * 16 * kRowH + hb * 64);
                    fh.h = tr_frag32<kRowH>(t_h[0] + tr_h + ks * 16 * kRowH + hb * 64);
                    fh.l = tr_frag32<kRowH>(t_h[1] + tr_h + ks * 16 * kRowH + hb * 64);
                    mfma3(dw1[hb], fdh, fx);           // dW1[hidden (rows), feature (lane)]
                    mfma3(dw2[hb], fdy, fh);           // dW2[output (rows), hidden (lane)]
                }
            }
            wave_sync();                       // (the next half / batch rewrites the tiles)
        }
    }
    // ---- this wave's sums -> the workgroup's (LDS) -> one set of global atomics per workgroup
    // accumulator element (row i, column l31): register r with i = (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (l31 < IN) atomicAdd(wsum + (32 * hb + i) * IN + l31, dw1[hb][r]);
            if (i < a.n_out) atomicAdd(wsum + kHidden * IN + i * kHidden + 32 * hb + l31, dw2[hb][r]);
        }
    __syncthreads();
    for (int i = tid; i < kHidden * IN; i += 256) atomicAdd(a.dw1 + i, wsum[i]);
    for (int i = tid; i < a.n_out * kHidden; i += 256) atomicAdd(a.dw2 + i, wsum[kHidden * IN + i]);
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
    // DREAMMAT_FIELD_MLP_BWD=v1: the first formulation (fp32 vector FMAs + fp32 matrix instructions), kept for A/B runs and tests
    static const bool v1 = getenv("DREAMMAT_FIELD_MLP_BWD") && !strcmp(getenv("DREAMMAT_FIELD_MLP_BWD"), "v1");
    const unsigned grid = (unsigned)std::min<long long>(((M + 63) / 64 + 3) / 4, 256);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd2<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_field_mlp_bwd2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    DM_ENTER();
    if (v1) {
        const size_t lds = (size_t)(kHidden * n_in + kHidden * kMaxOut + 4 * kWaveLds) * sizeof(float);      // ~120 KB: one workgroup per CU
        if (n_in == 32) hipLaunchKernelGGL(k_field_mlp_bwd<32>, dim3(grid), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL(k_field_mlp_bwd<16>, dim3(grid), dim3(256), lds, stream, a);
    } else {
        const size_t lds = (size_t)4 * kWaveLds2 + (size_t)(kHidden * n_in + kMaxOut * kHidden) * sizeof(float) + 24 * 1024;   // ~137 KB
        if (n_in == 32) hipLaunchKernelGGL(k_field_mlp_bwd2<32>, dim3(grid), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL(k_field_mlp_bwd2<16>, dim3(grid), dim3(256), lds, stream, a);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
