// Attention backward on MFMA for gfx950 (bf16 in / out, fp32 accumulate): the gradient of softmax(QK^T.scale).V for the
// TRAINABLE transformer blocks -- the ControlNet training loop (controlnet_train/diffusers_train_controlnet.py:858-915:
// the loss is differentiated through every attention of the ControlNet copy and of the UNet decoder it feeds).
//
// Flash-style: nothing S x S is stored.  The forward (k_attn_fwd with AttnArgs::lse set) keeps one number per query row,
// L = rowmax + log2(rowsum) of the scaled scores in the log2 domain; the backward recomputes P = exp2(S.scale.log2e - L)
// tile by tile.  Two kernels, no atomics (every output element has one owner, fixed summation order):
//   k_attn_bwd_dq   a workgroup owns 128 query rows (4 waves x 32) and walks the KV tiles:
//                     S^T = K.Q^T, dP^T = V.dO^T        (A = K / V rows from LDS, B = Q / dO rows in registers)
//                     dS^T = P^T o (dP^T - delta)        (delta = rowsum(dO o O), computed in the prologue and stored)
//                     dQ^T += K^T.dS^T                   (A = K^T from LDS, B = dS^T straight from the accumulators)
//   k_attn_bwd_dkv  a workgroup owns 128 key rows and walks the Q tiles:
//                     S = Q.K^T, dP = dO.V^T             (A = Q / dO rows from LDS, B = K / V rows in registers)
//                     dV^T += dO^T.P, dK^T += Q^T.dS     (A = dO^T / Q^T from LDS, B = P / dS from the accumulators)
// As in the forward kernels the softmax index that is NOT contracted next stays on the lane (n of the 32x32x16 MFMA), so
// the accumulator registers of the first product ARE the B operand of the second one.  The transposed A operands (K^T, Q^T,
// dO^T: 8 consecutive ROWS of one column per lane) come out of the SAME row-major LDS tiles through gfx950's transposing
// read (ds_read_b64_tr_b16; semantics measured by tools/tr_probe.cpp: in a group of 16 lanes, lane i passes the address of
// 4 consecutive columns of row k0 + (i >> 2) and receives column i of rows k0 .. k0 + 3): two reads per fragment, rows taken
// in the accumulator's order (k0 = 16 ks + 8 r + 4 hi), no transposed copy and no 2-byte scatter (the first version staged
// one: 474 TF/s executed at S = 4096; profiles/r03_attn_bwd_probe.jsonl).
// Rows past the end of a sequence are staged as zeros: a zero K row contributes nothing to dQ (K^T column is zero), a zero
// Q / dO row nothing to dK / dV (its L is +inf => P = 0); the lanes that own such rows are never stored.
#include "attn_common.h"

using namespace dm_attn;

namespace {

constexpr int kTile = 64;    // rows of the streamed operand per step
constexpr int kOwn = 128;    // rows a workgroup owns (4 waves x 32)

struct AttnBwdArgs {
    const elem_t* q; const elem_t* k; const elem_t* v; const elem_t* o; const elem_t* dout;
    const float* lse; float* delta;
    elem_t* dq; elem_t* dk; elem_t* dv;
    long long q_bs, q_ss, q_hs;   // q, o, dout, dq  [B, Sq, Hh, D] by strides, d contiguous
    long long k_bs, k_ss, k_hs;   // k, v, dk, dv    [B, Skv, Hh, D]
    int B, Hh, Sq, Skv, D;
    float scale, scale_log2;
};

template <int DP>
struct Lay {
    static constexpr int KSTEPS = DP / 16, DT = DP / 32, CPR = DP / 8;
    static constexpr int RROW = DP * 2 + 16;        // bytes per row of a row-major tile (padded: conflict-free b128 reads)
    static constexpr int RBYTES = kTile * RROW;
    static constexpr int NCH = kTile * CPR / 256;   // 16 B chunks per thread per operand
};

__device__ __forceinline__ uint4 ld16g(const elem_t* p) { return *reinterpret_cast<const uint4*>(p); }
// A-operand fragment of X^T from the row-major tile of X: lane (m = column, hi) receives rows k0 .. k0 + 3 (first read) and
// k0 + 8 .. k0 + 11 (second read) with k0 = 4 hi: the accumulator row order of one 16-wide k-step.  `p` = the lane's SOURCE
// address (row 4 hi + (i >> 2), columns 16 g1 + 4 (i & 3) of the k-step / column tile), RROW = bytes per row.
template <int RROW>
__device__ __forceinline__ elem8 tr_frag(const char* p) {
    const elem4 lo = dm_ds_read_tr16_b64(p);
    const elem4 hi = dm_ds_read_tr16_b64(p + 8 * RROW);
    return elem8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int DP>
__device__ __forceinline__ void write_rowmajor(char* dst, const uint4 (&reg)[Lay<DP>::NCH], int tid) {
    using L = Lay<DP>;
#pragma unroll
    for (int i = 0; i < L::NCH; ++i) {
        const int c = tid + 256 * i, row = c / L::CPR, col8 = c - row * L::CPR;
        *reinterpret_cast<uint4*>(dst + row * L::RROW + col8 * 16) = reg[i];
    }
}
// 64 rows x DP of a [rows, D] operand starting at row r0 (zeros past n_rows / past D)
template <int DP>
__device__ __forceinline__ void load_rows(uint4 (&reg)[Lay<DP>::NCH], const elem_t* base, long long row_stride, int r0,
                                          int n_rows, int D, int tid) {
    using L = Lay<DP>;
#pragma unroll
    for (int i = 0; i < L::NCH; ++i) {
        const int c = tid + 256 * i, row = c / L::CPR, col8 = c - row * L::CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + row < n_rows && col8 * 8 < D) v = ld16g(base + (long long)(r0 + row) * row_stride + col8 * 8);
        reg[i] = v;
    }
}
// B-operand fragments of the rows a lane owns: X[row][16 kk + 8 hi .. + 7]
template <int DP>
__device__ __forceinline__ void load_own(elem8 (&f)[Lay<DP>::KSTEPS], const elem_t* rowp, bool ok, int D, int hi) {
#pragma unroll
    for (int kk = 0; kk < Lay<DP>::KSTEPS; ++kk) {
        const int d = 16 * kk + 8 * hi;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok && d < D) v = ld16g(rowp + d);
        f[kk] = __builtin_bit_cast(elem8, v);
    }
}
// accumulator (32 rows x lane's column) -> two B-operand fragments (rows 0-15, 16-31 in accumulator order)
__device__ __forceinline__ void pack_acc(const f32x16& s, elem8& lo, elem8& hi8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2 a = {s[2 * e], s[2 * e + 1]}, b = {s[8 + 2 * e], s[8 + 2 * e + 1]};
        elem2 pa = __builtin_convertvector(a, elem2), pb = __builtin_convertvector(b, elem2);
        lo[2 * e] = pa[0]; lo[2 * e + 1] = pa[1];
        hi8[2 * e] = pb[0]; hi8[2 * e + 1] = pb[1];
    }
}
// lane's column of a [32 d-rows x 32] accumulator -> X[row][32 dt + ...] (bf16), scaled
template <int DP>
__device__ __forceinline__ void store_own(elem_t* rowp, const f32x16 (&acc)[Lay<DP>::DT], float scale, int D, int hi) {
#pragma unroll
    for (int dt = 0; dt < Lay<DP>::DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * dt + 8 * g + 4 * hi;
            if (d < D) {
                f32x2 x0 = {acc[dt][4 * g] * scale, acc[dt][4 * g + 1] * scale};
                f32x2 x1 = {acc[dt][4 * g + 2] * scale, acc[dt][4 * g + 3] * scale};
                elem2 y0 = __builtin_convertvector(x0, elem2), y1 = __builtin_convertvector(x1, elem2);
                elem4 y = {y0[0], y0[1], y1[0], y1[1]};
                *reinterpret_cast<elem4*>(rowp + d) = y;
            }
        }
}

// ------------------------------------------------------------------------------------------------ dQ (+ delta)
template <int DP>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(AttnBwdArgs a) {
    using L = Lay<DP>;
    constexpr int BUF = 2 * L::RBYTES;                  // K rows | V rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int tr_off = (4 * hi + ((lane & 15) >> 2)) * L::RROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int bh = blockIdx.y, b = bh / a.Hh, h = bh - b * a.Hh;
    const int q_row = blockIdx.x * kOwn + wave * 32 + l31;
    const bool q_ok = q_row < a.Sq;
    const long long q_off = (long long)b * a.q_bs + (long long)q_row * a.q_ss + (long long)h * a.q_hs;

    elem8 qf[L::KSTEPS], dof[L::KSTEPS];
    load_own<DP>(qf, a.q + q_off, q_ok, a.D, hi);
    load_own<DP>(dof, a.dout + q_off, q_ok, a.D, hi);
    float delta_q;
    {
        elem8 of[L::KSTEPS];
        load_own<DP>(of, a.o + q_off, q_ok, a.D, hi);
        float part = 0.f;
#pragma unroll
        for (int kk = 0; kk < L::KSTEPS; ++kk)
#pragma unroll
            for (int i = 0; i < 8; ++i) part += (float)dof[kk][i] * (float)of[kk][i];
        delta_q = part + __shfl_xor(part, 32);
    }
    const long long row_id = ((long long)b * a.Hh + h) * a.Sq + q_row;
    float lse_q = 0.f;
    if (q_ok) {
        lse_q = a.lse[row_id];
        if (hi == 0) a.delta[row_id] = delta_q;
    }

    const elem_t* kp = a.k + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const elem_t* vp = a.v + (long long)b * a.k_bs + (long long)h * a.k_hs;
    uint4 kreg[L::NCH], vreg[L::NCH];
    f32x16 dqT[L::DT];
#pragma unroll
    for (int i = 0; i < L::DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[i][r] = 0.f;

    const int n_tiles = (a.Skv + kTile - 1) / kTile;
    load_rows<DP>(kreg, kp, a.k_ss, 0, a.Skv, a.D, tid);
    load_rows<DP>(vreg, vp, a.k_ss, 0, a.Skv, a.D, tid);
    write_rowmajor<DP>(smem, kreg, tid);
    write_rowmajor<DP>(smem + L::RBYTES, vreg, tid);
    __syncthreads();

    for (int j = 0; j < n_tiles; ++j) {
        const char* kb = smem + (j & 1) * BUF;
        const char* vb = kb + L::RBYTES;
        if (j + 1 < n_tiles) {
            load_rows<DP>(kreg, kp, a.k_ss, (j + 1) * kTile, a.Skv, a.D, tid);
            load_rows<DP>(vreg, vp, a.k_ss, (j + 1) * kTile, a.Skv, a.D, tid);
        }
        elem8 dsf[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) {
                elem8 kf = *reinterpret_cast<const elem8*>(kb + (32 * t + l31) * L::RROW + 32 * kk + 16 * hi);
                s = DM_MFMA_32x32x16(kf, qf[kk], s);
            }
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) {
                elem8 vf = *reinterpret_cast<const elem8*>(vb + (32 * t + l31) * L::RROW + 32 * kk + 16 * hi);
                dp = DM_MFMA_32x32x16(vf, dof[kk], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[r] * a.scale_log2 - lse_q);
                dp[r] = p * (dp[r] - delta_q);
            }
            pack_acc(dp, dsf[2 * t], dsf[2 * t + 1]);
        }
#pragma unroll
        for (int dt = 0; dt < L::DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const elem8 ktf = tr_frag<L::RROW>(kb + tr_off + ks * 16 * L::RROW + dt * 64);
                dqT[dt] = DM_MFMA_32x32x16(ktf, dsf[ks], dqT[dt]);
            }
        if (j + 1 < n_tiles) {
            char* nb = smem + ((j + 1) & 1) * BUF;
            write_rowmajor<DP>(nb, kreg, tid);
            write_rowmajor<DP>(nb + L::RBYTES, vreg, tid);
        }
        __syncthreads();
    }
    if (q_ok) store_own<DP>(a.dq + q_off, dqT, a.scale, a.D, hi);
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int DP>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(AttnBwdArgs a) {
    using L = Lay<DP>;
    constexpr int BUF = 2 * L::RBYTES + 2 * kTile * 4;   // Q rows | dO rows | L | delta
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int tr_off = (4 * hi + ((lane & 15) >> 2)) * L::RROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int bh = blockIdx.y, b = bh / a.Hh, h = bh - b * a.Hh;
    const int kv_row = blockIdx.x * kOwn + wave * 32 + l31;
    const bool kv_ok = kv_row < a.Skv;
    const long long k_off = (long long)b * a.k_bs + (long long)kv_row * a.k_ss + (long long)h * a.k_hs;

    elem8 kf[L::KSTEPS], vf[L::KSTEPS];
    load_own<DP>(kf, a.k + k_off, kv_ok, a.D, hi);
    load_own<DP>(vf, a.v + k_off, kv_ok, a.D, hi);

    const elem_t* qp = a.q + (long long)b * a.q_bs + (long long)h * a.q_hs;
    const elem_t* dop = a.dout + (long long)b * a.q_bs + (long long)h * a.q_hs;
    const float* lsep = a.lse + ((long long)b * a.Hh + h) * a.Sq;
    const float* delp = a.delta + ((long long)b * a.Hh + h) * a.Sq;
    uint4 qreg[L::NCH], doreg[L::NCH];
    float stat = 0.f;     // threads 0-63: L of row tid; 64-127: delta of row tid - 64
    auto load_stat = [&](int q0) {
        if (tid < 2 * kTile) {
            const int r = q0 + (tid & (kTile - 1));
            const bool is_l = tid < kTile;
            stat = r < a.Sq ? (is_l ? lsep[r] : delp[r]) : (is_l ? INFINITY : 0.f);
        }
    };
    auto write_stat = [&](char* buf) {
        if (tid < 2 * kTile) reinterpret_cast<float*>(buf + 2 * L::RBYTES)[tid] = stat;
    };
    f32x16 dvT[L::DT], dkT[L::DT];
#pragma unroll
    for (int i = 0; i < L::DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvT[i][r] = 0.f; dkT[i][r] = 0.f; }

    const int n_tiles = (a.Sq + kTile - 1) / kTile;
    load_rows<DP>(qreg, qp, a.q_ss, 0, a.Sq, a.D, tid);
    load_rows<DP>(doreg, dop, a.q_ss, 0, a.Sq, a.D, tid);
    load_stat(0);
    write_rowmajor<DP>(smem, qreg, tid);
    write_rowmajor<DP>(smem + L::RBYTES, doreg, tid);
    write_stat(smem);
    __syncthreads();

    for (int j = 0; j < n_tiles; ++j) {
        const char* qb = smem + (j & 1) * BUF;
        const char* dob = qb + L::RBYTES;
        const float* lds_l = reinterpret_cast<const float*>(dob + L::RBYTES);
        const float* lds_d = lds_l + kTile;
        if (j + 1 < n_tiles) {
            load_rows<DP>(qreg, qp, a.q_ss, (j + 1) * kTile, a.Sq, a.D, tid);
            load_rows<DP>(doreg, dop, a.q_ss, (j + 1) * kTile, a.Sq, a.D, tid);
            load_stat((j + 1) * kTile);
        }
        elem8 pf[4], dsf[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) {
                elem8 qa = *reinterpret_cast<const elem8*>(qb + (32 * t + l31) * L::RROW + 32 * kk + 16 * hi);
                s = DM_MFMA_32x32x16(qa, kf[kk], s);
            }
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) {
                elem8 da = *reinterpret_cast<const elem8*>(dob + (32 * t + l31) * L::RROW + 32 * kk + 16 * hi);
                dp = DM_MFMA_32x32x16(da, vf[kk], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {       // accumulator register 4g + e = query row 32t + 8g + 4hi + e
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lds_l + 32 * t + 8 * g + 4 * hi);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(lds_d + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(s[4 * g + e] * a.scale_log2 - l4[e]);
                    s[4 * g + e] = p;
                    dp[4 * g + e] = p * (dp[4 * g + e] - d4[e]);
                }
            }
            pack_acc(s, pf[2 * t], pf[2 * t + 1]);
            pack_acc(dp, dsf[2 * t], dsf[2 * t + 1]);
        }
#pragma unroll
        for (int dt = 0; dt < L::DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const elem8 dta = tr_frag<L::RROW>(dob + tr_off + ks * 16 * L::RROW + dt * 64);
                dvT[dt] = DM_MFMA_32x32x16(dta, pf[ks], dvT[dt]);
                const elem8 qta = tr_frag<L::RROW>(qb + tr_off + ks * 16 * L::RROW + dt * 64);
                dkT[dt] = DM_MFMA_32x32x16(qta, dsf[ks], dkT[dt]);
            }
        if (j + 1 < n_tiles) {
            char* nb = smem + ((j + 1) & 1) * BUF;
            write_rowmajor<DP>(nb, qreg, tid);
            write_rowmajor<DP>(nb + L::RBYTES, doreg, tid);
            write_stat(nb);
        }
        __syncthreads();
    }
    if (kv_ok) {
        store_own<DP>(a.dv + k_off, dvT, 1.0f, a.D, hi);
        store_own<DP>(a.dk + k_off, dkT, a.scale, a.D, hi);
    }
}

template <int DP>
int launch_bwd(const AttnBwdArgs& a, hipStream_t stream) {
    using L = Lay<DP>;
    constexpr int LDS_DQ = 2 * (2 * L::RBYTES);
    constexpr int LDS_DKV = 2 * (2 * L::RBYTES + 2 * kTile * 4);
    static_assert(LDS_DKV <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        DM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_dq<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ));
        DM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd_dkv<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV));
        attr_set = true;
    }
    DM_ENTER();
    hipLaunchKernelGGL(k_attn_bwd_dq<DP>, dim3(dm_div_up(a.Sq, kOwn), a.B * a.Hh), dim3(256), LDS_DQ, stream, a);
    DM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_attn_bwd_dkv<DP>, dim3(dm_div_up(a.Skv, kOwn), a.B * a.Hh), dim3(256), LDS_DKV, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // namespace

extern "C" {

// Gradient of out = softmax(q.k^T.scale).v for one call of dm_attention_fwd_lse_bf16.
// q, out, dout, dq  [B, Sq, Hh, D]  by strides (q_bs, q_ss, q_hs), d contiguous -- the four tensors share the strides
// k, v, dk, dv      [B, Skv, Hh, D] by strides (k_bs, k_ss, k_hs)               -- v is NOT transposed here
// lse   [B, Hh, Sq] fp32 from the forward; delta [B, Hh, Sq] fp32 scratch (written by the dQ kernel, read by the dK/dV one)
// D % 8 == 0, D <= 128; pointers 16 B aligned, strides multiples of 8 elements.
int dm_attention_bwd_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                          float* delta, void* dq, void* dk, void* dv, int B, int Hh, int Sq, int Skv, int D,
                          long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss, long long k_hs,
                          float scale, hipStream_t stream) {
    if (!q || !k || !v || !out || !dout || !lse || !delta || !dq || !dk || !dv) return DM_ERR_ARG;
    if (B <= 0 || Hh <= 0 || Sq <= 0 || Skv <= 0 || D <= 0) return DM_ERR_ARG;
    if (D % 8 != 0 || D > 128) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk |
         (uintptr_t)dv) & 15)
        return DM_ERR_ARG;
    if ((q_bs | q_ss | q_hs | k_bs | k_ss | k_hs) & 7) return DM_ERR_ARG;
    if ((long long)B * Hh > 65535) return DM_ERR_UNSUPPORTED;
    AttnBwdArgs a;
    a.q = (const elem_t*)q; a.k = (const elem_t*)k; a.v = (const elem_t*)v; a.o = (const elem_t*)out;
    a.dout = (const elem_t*)dout; a.lse = lse; a.delta = delta;
    a.dq = (elem_t*)dq; a.dk = (elem_t*)dk; a.dv = (elem_t*)dv;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = q_hs; a.k_bs = k_bs; a.k_ss = k_ss; a.k_hs = k_hs;
    a.B = B; a.Hh = Hh; a.Sq = Sq; a.Skv = Skv; a.D = D;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (D <= 32) return launch_bwd<32>(a, stream);
    if (D <= 64) return launch_bwd<64>(a, stream);
    return launch_bwd<128>(a, stream);
}

}  // extern "C"
