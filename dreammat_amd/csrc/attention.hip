// Flash-style fused attention forward (QK^T . softmax . V) on MFMA for gfx950 -- bf16 in/out,
// fp32 accumulate, no S x S materialisation.  Plugged in as the attention of the UNet/ControlNet
// transformer blocks that diffusers runs for threestudio/models/guidance/dreammat_guidance.py
// :205-241 (multi_control_forward) and :261-282 (forward_unet).
//
// Design (wave64, v_mfma_f32_32x32x16_bf16):
//  * workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
//  * "swapped" products keep the softmax row on the lane:  S^T[kv,q] = K . Q^T  (A = K rows from
//    LDS, B = Q rows from registers) so lane l holds 16+16 scores of query q = l&31, and
//    O^T[d,q] = V^T . P^T (A = V^T rows from LDS, B = P straight from the score registers).
//    No cross-lane shuffles except one lane^32 exchange for the row max / final row sum.
//  * V arrives already transposed (vt[b,h,d,kv], produced for free by the projection GEMM), and
//    the kv order inside each 16-wide MFMA k-step is permuted when the tile is written to LDS so
//    that the accumulator layout of S^T *is* the B-operand layout of the P.V product.
//  * K / V^T tiles (64 kv) are register-staged (global -> VGPR -> LDS, issue-early/write-late)
//    into double-buffered LDS with 16 B row padding => conflict-free ds_read_b128; one barrier
//    per KV tile.
//  * head_dim D <= DP (template: 64/96/128/160), D % 8 == 0: SD-2.1 (64) and SD-1.5 (40/80/160)
//    shapes are both covered, the pad columns are zero-filled in registers, never in memory.
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "attn_common.h"

using namespace dm_attn;

namespace {

constexpr int kQRowsPerWave = 32;
constexpr int kWaves = 4;
constexpr int kKvTile = 64;

__device__ __forceinline__ uint4 ld16(const elem_t* p) { return *reinterpret_cast<const uint4*>(p); }

// VNAT: V arrives as [kv][d] rows (the differentiated path, dm_attention_fwd_lse_bf16): staged row-major like K and consumed
// through the transposing LDS read (ds_read_b64_tr_b16: two reads per V^T fragment, rows in accumulator order).
template <int DP, bool VNAT>
__global__ __launch_bounds__(256) void k_attn_fwd(AttnArgs a) {
    constexpr int KSTEPS = DP / 16;       // MFMA k-steps over head_dim
    constexpr int DT = DP / 32;           // 32-wide output d tiles
    constexpr int KROW = DP * 2 + 16;     // bytes per K row in LDS (padded)
    constexpr int VROW = kKvTile * 2 + 16;
    constexpr int KBYTES = kKvTile * KROW;
    constexpr int VBYTES = VNAT ? kKvTile * KROW : DP * VROW;
    constexpr int CPR = DP / 8;           // 16 B chunks per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    // source address of this lane in a transposing read: row 4 hi + (i >> 2), columns 16 g1 + 4 (i & 3)
    const int tr_off = (4 * hi + ((lane & 15) >> 2)) * KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int bh = blockIdx.y, b = bh / a.Hh, h = bh - b * a.Hh;
    const int q_row = blockIdx.x * (kWaves * kQRowsPerWave) + wave * kQRowsPerWave + l31;
    const bool q_ok = q_row < a.Sq;
    const int skv_pad8 = (a.Skv + 7) & ~7;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q][16kk + 8hi .. +7]
    elem8 qf[KSTEPS];
    {
        const elem_t* qp = a.q + (long long)b * a.q_bs + (long long)q_row * a.q_ss + (long long)h * a.q_hs;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            int d = 16 * kk + 8 * hi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q_ok && d < a.D) v = ld16(qp + d);
            qf[kk] = __builtin_bit_cast(elem8, v);
        }
    }

    const elem_t* kp = a.k + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const elem_t* vp = VNAT ? a.v_nat + (long long)b * a.k_bs + (long long)h * a.k_hs
                            : a.vt + (long long)b * a.vt_bs + (long long)h * a.vt_hs;

    uint4 kreg[DT], vreg[DT];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            int c = tid + 256 * i;
            int row = c / CPR, col8 = c - row * CPR;
            int kv = kv0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kv < a.Skv && col8 * 8 < a.D) v = ld16(kp + (long long)kv * a.k_ss + col8 * 8);
            kreg[i] = v;
            int d = c >> 3, kc = c & 7;
            uint4 w = make_uint4(0, 0, 0, 0);
            if (VNAT) {
                if (kv < a.Skv && col8 * 8 < a.D) w = ld16(vp + (long long)kv * a.k_ss + col8 * 8);
            } else if (d < a.D && kv0 + kc * 8 < skv_pad8) {
                w = ld16(vp + (long long)d * a.vt_ds + kv0 + kc * 8);
            }
            vreg[i] = w;
        }
    };
    auto write_tile = [&](int buf) {
        char* kb = smem + buf * (KBYTES + VBYTES);
        char* vb = kb + KBYTES;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            int c = tid + 256 * i;
            int row = c / CPR, col8 = c - row * CPR;
            *reinterpret_cast<uint4*>(kb + row * KROW + col8 * 16) = kreg[i];
            if (VNAT) {
                *reinterpret_cast<uint4*>(vb + row * KROW + col8 * 16) = vreg[i];
                continue;
            }
            int d = c >> 3, kc = c & 7;
            // kv permutation inside each 16-group: [0-3, 8-11, 4-7, 12-15]
            char* dst = vb + d * VROW + (kc >> 1) * 32 + (kc & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = make_uint2(vreg[i].x, vreg[i].y);
            *reinterpret_cast<uint2*>(dst + 16) = make_uint2(vreg[i].z, vreg[i].w);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int n_tiles = (a.Skv + kKvTile - 1) / kKvTile;
    load_tile(0);
    write_tile(0);
    __syncthreads();

    for (int j = 0; j < n_tiles; ++j) {
        const int buf = j & 1;
        const int kv0 = j * kKvTile;
        if (j + 1 < n_tiles) load_tile(kv0 + kKvTile);
        const char* kb = smem + buf * (KBYTES + VBYTES);
        const char* vb = kb + KBYTES;

        // ---- S^T = K . Q^T
        f32x16 s[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                elem8 kf = *reinterpret_cast<const elem8*>(kb + (32 * t + l31) * KROW + 32 * kk + 16 * hi);
                s[t] = DM_MFMA_32x32x16(kf, qf[kk], s[t]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- mask the ragged tail (cross-attention: Skv = 77)
        if (kv0 + kKvTile > a.Skv) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int kv = kv0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kv >= a.Skv) s[t][r] = -INFINITY;
                }
        }
        // ---- online softmax (log2 domain), row = lane's query
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;
        // deferred rescale: while no row of this wave outgrows its running max by more than 2^8 the
        // accumulators keep their old scale (P <= 256, harmless for the fp32 accumulation and for bf16 P,
        // whose relative precision does not depend on magnitude) and the O-wide multiply is skipped.
        // The branch is wave-uniform; the previous tile's P.V is already complete at this point.
        float m_new = m_run, alpha = 1.0f;
        const bool rescale = !__all(mx <= m_run + 8.0f);
        if (rescale) {
            m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        }
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(s[t][r] * a.scale_log2 - m_new);
                s[t][r] = p;
                psum += p;
            }
        if (rescale) {
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        l_run += psum;
        m_run = m_new;

        // ---- P -> bf16 B-operand fragments (no data movement: accumulator order == k-slot order)
        elem8 pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int t = ks >> 1, u = ks & 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 two = {s[t][8 * u + 2 * e], s[t][8 * u + 2 * e + 1]};
                elem2 pk = __builtin_convertvector(two, elem2);
                pf[ks][2 * e] = pk[0];
                pf[ks][2 * e + 1] = pk[1];
            }
        }
        // ---- O^T += V^T . P^T
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                elem8 vf;
                if (VNAT) {
                    const char* p = vb + tr_off + ks * 16 * KROW + dt * 64;
                    const elem4 lo = dm_ds_read_tr16_b64(p);
                    const elem4 up = dm_ds_read_tr16_b64(p + 8 * KROW);
                    vf = elem8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
                } else {
                    vf = *reinterpret_cast<const elem8*>(vb + (32 * dt + l31) * VROW + 32 * ks + 16 * hi);
                }
                o[dt] = DM_MFMA_32x32x16(vf, pf[ks], o[dt]);
            }

        __builtin_amdgcn_s_setprio(0);
        if (j + 1 < n_tiles) write_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, store O[q][d]
    float l_tot = l_run + __shfl_xor(l_run, 32);
    float inv = 1.0f / l_tot;
    if (a.lse && q_ok && hi == 0) a.lse[((long long)b * a.Hh + h) * a.Sq + q_row] = m_run + __builtin_amdgcn_logf(l_tot);
    if (q_ok) {
        elem_t* op = a.out + (long long)b * a.o_bs + (long long)q_row * a.o_ss + (long long)h * a.o_hs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int d = 32 * dt + 8 * g + 4 * hi;
                if (d < a.D) {
                    f32x2 x0 = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv};
                    f32x2 x1 = {o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                    elem2 y0 = __builtin_convertvector(x0, elem2), y1 = __builtin_convertvector(x1, elem2);
                    elem4 y = {y0[0], y0[1], y1[0], y1[1]};
                    *reinterpret_cast<elem4*>(op + d) = y;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Variant 3 (DP = 64 or 128): same decomposition (4 waves x 32 query rows, 64-kv tiles, 3-stage LDS-DMA ring) with
// the per-tile instruction stream cut from ~290 to ~150 non-MFMA instructions -- the kernel is bound by VALU issue
// beside the MFMAs (rocprofv3 round 1: 218 M VALU instructions against 503 M MFMA-busy cycles, SQ_WAIT_INST_ANY 46 %):
//  * DMA through a buffer descriptor (buffer_load ... lds): per-lane offsets are loop-invariant VGPRs, the tile offset
//    is an SGPR (soffset), rows past the end of K and head-dim columns >= D fall outside num_records and read as zero:
//    no address arithmetic, clamps, selects or zero page in the loop (was ~50 VALU + 12 SALU per tile);
//  * K rows land in LDS with bits 2 and 3 of the row index swapped, which makes the accumulator order of S^T the NATURAL
//    k order of the P.V product: V^T fragments are single ds_read_b128 (was two ds_read_b64 each);
//  * Q is multiplied by scale*log2(e) once in the prologue (one extra bf16 rounding of Q) and the running
//    maximum enters through the MFMA's C operand (S' = K.Q'^T - m), so the softmax numerator is a bare v_exp_f32 per
//    element: no multiply, no subtract;
//  * the running maximum is only revised when some row outgrows it by more than 2^8 (as before); the revision path
//    (rare) re-bases S', O, l and the C-operand vector.
__device__ __forceinline__ int attn_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// VLATE: the V^T fragments are read inside the P.V loop (16 VGPRs at a time) instead of ahead of the softmax (32 VGPRs
// across it): the difference between 2 and 3 resident waves per SIMD.
// KVRES (round 4): the cross-attention form.  The 77 prompt tokens of a (batch, head) are two kv tiles -- 32 KB of LDS --, and the
// plain form spends more on them than on the scores: every 128-row workgroup zero-fills its ring, DMAs both tiles, and passes two
// counted waits and barriers for 2 x 8 + 2 x 8 MFMAs per wave (46 us for 24 x 5 heads x 4096 rows: 2.7 TB/s of Q / O streaming).
// Here a workgroup loads the (at most 3) kv tiles of its (batch, head) ONCE, keeps them in the three ring stages, and walks its
// share of the query blocks with the next block's Q rows prefetched into registers: no DMA, wait or barrier inside the loop.
// Grid = (batch x heads) x `kvres_groups` workgroups (launcher: enough to fill the chip 2-3 times).
template <int DP, bool VLATE, bool KVRES = false>
__global__ __launch_bounds__(256, VLATE ? 3 : 2) void k_attn_fwd_v3(AttnArgs a, int kvres_groups) {
#if defined(__HIP_DEVICE_COMPILE__)   // __amdgpu_buffer_rsrc_t does not exist in the host pass (the stub needs no body)
    static_assert(DP == 64 || DP == 128, "power-of-two row sizes only");
    constexpr int KSTEPS = DP / 16, DT = DP / 32;
    constexpr int KROW = DP * 2, VROW = kKvTile * 2;
    constexpr int KBYTES = kKvTile * KROW, VBYTES = DP * VROW, STAGE = KBYTES + VBYTES;
    constexpr int KCH = KROW / 16, KRPI = 64 / KCH, K_INSTR = kKvTile / KRPI / 4, V_INSTR = DP / 8 / 4;
    constexpr int L = K_INSTR + V_INSTR;
    constexpr int NST = 3;                               // LDS ring depth
    constexpr int OOB = 0x40000000;                     // beyond any num_records the launcher admits
    constexpr float THR = 8.0f;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bh, qb;
    const int nq_all = (a.Sq + kWaves * kQRowsPerWave - 1) / (kWaves * kQRowsPerWave);
    {
        const int nq = KVRES ? kvres_groups : nq_all;       // KVRES: `qb` is this workgroup's FIRST query block, stride kvres_groups
        const int BH = a.B * a.Hh, id = blockIdx.x;
        if ((BH & 7) == 0) {
            const int j = id >> 3;
            bh = (j / nq) * 8 + (id & 7);
            qb = j - (j / nq) * nq;
        } else {
            bh = id / nq;
            qb = id - bh * nq;
        }
    }
    const int b = bh / a.Hh, h = bh - b * a.Hh;
    int q_row = qb * (kWaves * kQRowsPerWave) + wave * kQRowsPerWave + l31;
    bool q_ok = q_row < a.Sq;
    const int skv_pad8 = (a.Skv + 7) & ~7;
    const int n_tiles = (a.Skv + kKvTile - 1) / kKvTile;
    auto kswz = [](int r) { return KCH == 8 ? ((r >> 1) & 7) : (r & 15); };

    // ragged sequences: whatever an out-of-range DMA lane does to its LDS slot (writes zero / leaves it), the slot must
    // hold finite numbers, because masked probabilities are exact zeros and 0 x NaN would poison O
    if ((a.Skv & (kKvTile - 1)) != 0 || a.D != DP) {
        for (int o = tid * 16; o < NST * STAGE; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

    elem8 qf[KSTEPS];
    uint4 qraw[KSTEPS];
    // KVRES: whole 128-byte rows, 16 B per lane (8 lanes per row, 8 rows per instruction: 8 cache lines instead of the 32 a
    // "one row per lane pair" access touches -- tools/gather_probe.cpp: 21 vs ~70 cycles of the CU's address unit per wave
    // instruction; Q and O of a cross-attention are ALL its traffic), turned into / out of the MFMA fragment layout through a
    // wave-private 4 KB of the ring's third stage (two kv tiles are resident: the launcher's contract).
    char* const xw = smem + 2 * STAGE + wave * 4096;
    auto q_load = [&](int row, bool ok) {                // raw rows of one query block (KVRES: requested a block ahead)
        if constexpr (KVRES) {
            const int row0 = row - l31;                  // the block's first row (row = first + l31 for this lane)
#pragma unroll
            for (int i = 0; i < KSTEPS; ++i) {
                const int r = row0 + 8 * i + (lane >> 3);
                qraw[i] = make_uint4(0, 0, 0, 0);
                if (ok && r < a.Sq)
                    qraw[i] = ld16(a.q + (long long)b * a.q_bs + (long long)r * a.q_ss + (long long)h * a.q_hs + (lane & 7) * 8);
            }
        } else {
            const elem_t* qp = a.q + (long long)b * a.q_bs + (long long)row * a.q_ss + (long long)h * a.q_hs;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 16 * kk + 8 * hi;
                qraw[kk] = make_uint4(0, 0, 0, 0);
                if (ok && d < a.D) qraw[kk] = ld16(qp + d);
            }
        }
    };
    auto q_scale = [&]() {                               // Q' = Q * scale * log2 e, rounded to bf16 once
        if constexpr (KVRES) {                           // [row][chunk ^ (row & 7)] image -> fragment (row l31, chunk 2 kk + hi)
#pragma unroll
            for (int i = 0; i < KSTEPS; ++i) {
                const int r = 8 * i + (lane >> 3);
                *reinterpret_cast<uint4*>(xw + r * 128 + (((lane & 7) ^ (r & 7)) << 4)) = qraw[i];
            }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
                qraw[kk] = *reinterpret_cast<const uint4*>(xw + l31 * 128 + (((2 * kk + hi) ^ (l31 & 7)) << 4));
        }
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            qf[kk] = __builtin_bit_cast(elem8, qraw[kk]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2 two = {(float)qf[kk][e] * a.scale_log2, (float)qf[kk][e + 1] * a.scale_log2};
                elem2 pk = __builtin_convertvector(two, elem2);
                qf[kk][e] = pk[0];
                qf[kk][e + 1] = pk[1];
            }
        }
    };
    q_load(q_row, KVRES ? true : q_ok);                  // (KVRES checks every loaded row against Sq itself)
    if (!KVRES) q_scale();
    // ---- DMA descriptors: one per operand, base = this (batch, head)
    const elem_t* kp = a.k + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const elem_t* vp = a.vt + (long long)b * a.vt_bs + (long long)h * a.vt_hs;
    const int k_bytes = (int)((((long long)a.Skv - 1) * a.k_ss + a.D) * 2);
    const int v_bytes = (int)((((long long)a.D - 1) * a.vt_ds + skv_pad8) * 2);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, v_bytes, 0x00020000);
    int k_voff[K_INSTR], v_voff[V_INSTR], v_kv[V_INSTR];
#pragma unroll
    for (int i = 0; i < K_INSTR; ++i) {
        const int r = (wave * K_INSTR + i) * KRPI + lane / KCH;            // LDS row of this lane's 16 B slot
        const int col = ((lane % KCH) ^ kswz(r)) * 8;                       // source column (elements) that lands there
        k_voff[i] = col < a.D ? (int)(((long long)attn_swap23(r) * a.k_ss + col) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < V_INSTR; ++i) {
        const int d = (wave * V_INSTR + i) * 8 + (lane >> 3);
        v_kv[i] = ((lane & 7) ^ ((d >> 1) & 7)) * 8;
        v_voff[i] = d < a.D ? (int)(((long long)d * a.vt_ds + v_kv[i]) * 2) : OOB;
    }
    const int k_tile_bytes = (int)(a.k_ss * 2 * kKvTile);
    auto issue = [&](int j, int stage, auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        char* kb = smem + stage * STAGE;
        char* vb = kb + KBYTES;
        const int ks = j * k_tile_bytes, vs = j * (kKvTile * 2);
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(kb + (wave * K_INSTR + i) * 1024),
                                                     16, k_voff[i], ks, 0, 0);
#pragma unroll
        for (int i = 0; i < V_INSTR; ++i) {
            int vo = v_voff[i];
            // the kv columns of the LAST tile beyond the zero-padded row end belong to the next row (or to nobody)
            if (GUARD) vo = (j * kKvTile + v_kv[i] < skv_pad8) ? vo : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(vb + (wave * V_INSTR + i) * 1024),
                                                     16, vo, vs, 0, 0);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    f32x16 cinit;                                        // C operand of the first S^T MFMA: -m
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;                      // cinit == -m_run at all times
    elem8 pf[4];

    // S^T = K.Q^T (+ C operand) of the tile in `stage`
    auto qk = [&](int stage, f32x16 (&s)[2]) {
        const char* kb = smem + stage * STAGE;
        elem8 kf[2][KSTEPS];
        auto read_k = [&](int t) {
            const int row = 32 * t + l31;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
                kf[t][kk] = *reinterpret_cast<const elem8*>(kb + row * KROW + (((2 * kk + hi) ^ kswz(row)) << 4));
        };
        read_k(0);
        if (!VLATE) read_k(1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (VLATE && t == 0) read_k(1);
            s[t] = DM_MFMA_32x32x16(kf[t][0], qf[0], cinit);
#pragma unroll
            for (int kk = 1; kk < KSTEPS; ++kk)
                s[t] = DM_MFMA_32x32x16(kf[t][kk], qf[kk], s[t]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // softmax of tile j (scores in `s`) and O^T += V^T.P^T
    auto softmax_pv = [&](int j, int stage, bool first, f32x16 (&s)[2], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* vb = smem + stage * STAGE + KBYTES;
        elem8 vf[DT][4];
        auto read_v = [&](int dt) {
            const int row = 32 * dt + l31;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                vf[dt][ks] = *reinterpret_cast<const elem8*>(vb + row * VROW + (((2 * ks + hi) ^ sw) << 4));
        };
        if (!VLATE) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) read_v(dt);
        }
        if (MASKED) {
            if ((j + 1) * kKvTile > a.Skv) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // LDS row (r&3) + 8(r>>2) + 4hi of this half holds K row swap23(.) of the tile
                        const int kv = j * kKvTile + 32 * t + 16 * (r >> 3) + 8 * hi + 4 * ((r >> 2) & 1) + (r & 3);
                        if (kv >= a.Skv) s[t][r] = -INFINITY;
                    }
            }
        }
        // ---- row maximum of this tile (relative to the running maximum): four independent chains
        float mq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) mq[c] = fmaxf(s[0][c], s[1][c]);
#pragma unroll
        for (int r = 4; r < 16; ++r) mq[r & 3] = fmaxf(fmaxf(mq[r & 3], s[0][r]), s[1][r]);
        float mx = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float ps[4] = {0.f, 0.f, 0.f, 0.f};            // four independent partial row sums (a single chain of 32 dependent adds stalls)
        {
            if (first || !__all(mx <= THR)) {            // wave-uniform; the previous tile's P.V is complete
                const float delta = first ? mx : fmaxf(mx, 0.f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[0][r] -= delta; s[1][r] -= delta; cinit[r] = -m_run; }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[t][r]);
                    s[t][r] = p;
                    ps[r & 3] += p;
                }
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int t = ks >> 1, u = ks & 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 two = {s[t][8 * u + 2 * e], s[t][8 * u + 2 * e + 1]};
                elem2 pk = __builtin_convertvector(two, elem2);
                pf[ks][2 * e] = pk[0];
                pf[ks][2 * e + 1] = pk[1];
            }
        }
        if (VLATE) read_v(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (VLATE && dt + 1 < DT) read_v(dt + 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                o[dt] = DM_MFMA_32x32x16(vf[dt][ks], pf[ks], o[dt]);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    f32x16 sA[2];
    auto store_out = [&]() {
        float l_tot = l_run + __shfl_xor(l_run, 32);
        float inv = 1.0f / l_tot;
        if constexpr (KVRES) {                           // (D == DP == 64: the launcher's contract)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x2 x0 = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv};
                    f32x2 x1 = {o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                    elem2 y0 = __builtin_convertvector(x0, elem2), y1 = __builtin_convertvector(x1, elem2);
                    elem4 y = {y0[0], y0[1], y1[0], y1[1]};
                    // head_dim 32 dt + 8 g + 4 hi .. +3: 16 B chunk 4 dt + g (swizzled by the row), half hi
                    *reinterpret_cast<elem4*>(xw + l31 * 128 + (((4 * dt + g) ^ (l31 & 7)) << 4) + 8 * hi) = y;
                }
            const int row0 = q_row - l31;
            elem_t* op = a.out + (long long)b * a.o_bs + (long long)h * a.o_hs;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + (lane >> 3);
                const uint4 v = *reinterpret_cast<const uint4*>(xw + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
                if (row0 + r < a.Sq) *reinterpret_cast<uint4*>(op + (long long)(row0 + r) * a.o_ss + (lane & 7) * 8) = v;
            }
            return;
        }
        if (q_ok) {
            elem_t* op = a.out + (long long)b * a.o_bs + (long long)q_row * a.o_ss + (long long)h * a.o_hs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int d = 32 * dt + 8 * g + 4 * hi;
                    if (d < a.D) {
                        f32x2 x0 = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv};
                        f32x2 x1 = {o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                        elem2 y0 = __builtin_convertvector(x0, elem2), y1 = __builtin_convertvector(x1, elem2);
                        elem4 y = {y0[0], y0[1], y1[0], y1[1]};
                        *reinterpret_cast<elem4*>(op + d) = y;
                    }
                }
        }
    };
    if constexpr (KVRES) {
        // every kv tile of this (batch, head) into its own stage, once (n_tiles <= 2 by the launcher's contract: the third
        // stage is the waves' Q / O staging space)
        for (int t = 0; t < n_tiles; ++t) {
            if (t + 1 == n_tiles) issue(t, t, std::true_type{}); else issue(t, t, std::false_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int blk = qb; blk < nq_all; blk += kvres_groups) {
            q_scale();                                   // this block's rows (requested one block ago)
            const int row_now = q_row;
            const bool ok_now = q_ok;
            {   // request the next block's rows; they land while this block computes
                const int nrow = (blk + kvres_groups) * (kWaves * kQRowsPerWave) + wave * kQRowsPerWave + l31;
                q_load(nrow, blk + kvres_groups < nq_all);
                q_row = nrow; q_ok = blk + kvres_groups < nq_all && nrow < a.Sq;
            }
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
            m_run = 0.f; l_run = 0.f;
            for (int t = 0; t < n_tiles; ++t) {
                qk(t, sA);
                softmax_pv(t, t, t == 0, sA, std::true_type{});
            }
            const int keep_row = q_row; const bool keep_ok = q_ok;
            q_row = row_now; q_ok = ok_now;
            store_out();
            q_row = keep_row; q_ok = keep_ok;
        }
        return;
    }
    // prologue: tiles 0 and 1 (the last tile of the sequence is always issued through the guarded form)
    if (n_tiles == 1) issue(0, 0, std::true_type{}); else issue(0, 0, std::false_type{});
    if (n_tiles == 2) issue(1, 1, std::true_type{}); else if (n_tiles > 2) issue(1, 1, std::false_type{});
    int stage = 0, j = 0;
    {
        // main loop: tile j computes while tiles j+1, j+2 are in flight; every tile touched here is a full one
        for (; j + 3 < n_tiles; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            __builtin_amdgcn_s_barrier();
            int st2 = stage + 2; if (st2 >= 3) st2 -= 3;
            issue(j + 2, st2, std::false_type{});
            qk(stage, sA);
            softmax_pv(j, stage, j == 0, sA, std::false_type{});
            stage = stage + 1; if (stage >= 3) stage = 0;
        }
        // tail: at most three tiles; issues the (possibly ragged) last tile guarded, masks the last tile
        for (; j < n_tiles; ++j) {
            if (j + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (j + 2 < n_tiles) {
                int st2 = stage + 2; if (st2 >= 3) st2 -= 3;
                issue(j + 2, st2, std::true_type{});
            }
            qk(stage, sA);
            softmax_pv(j, stage, j == 0, sA, std::true_type{});
            stage = stage + 1; if (stage >= 3) stage = 0;
        }
    }
    store_out();
#endif
}

template <int DP, bool VLATE>
int launch_attn_v3(const AttnArgs& a, hipStream_t stream) {
    constexpr int LDS = 3 * (kKvTile * DP * 2 + DP * kKvTile * 2);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd_v3<DP, VLATE, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long nq = dm_div_up(a.Sq, kWaves * kQRowsPerWave);
    const long long n_blocks = nq * a.B * a.Hh;
    if (n_blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    // the kv-resident form (cross-attention: <= 3 kv tiles) when a (batch, head) has enough query blocks to amortise over:
    // groups per (batch, head) so that the grid covers the chip ~2.5 times (2-3 workgroups are resident per CU)
    static const bool kvres_off = getenv("DREAMMAT_ATTN_KVRES") && !strcmp(getenv("DREAMMAT_ATTN_KVRES"), "0");
    if (!kvres_off && DP == 64 && a.D == 64 && a.Skv <= 2 * kKvTile && nq >= 4 && (a.o_ss & 7) == 0 && (a.o_hs & 7) == 0 && (a.o_bs & 7) == 0 &&
        (((uintptr_t)a.out) & 15) == 0) {
        const long long bh = (long long)a.B * a.Hh;
        long long groups = std::max<long long>(1, std::min<long long>(nq / 2, (640 + bh - 1) / bh));
        static bool attr2 = false;
        if (!attr2) {
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd_v3<DP, VLATE, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e2 != hipSuccess) return (int)e2;
            attr2 = true;
        }
        hipLaunchKernelGGL((k_attn_fwd_v3<DP, VLATE, true>), dim3((unsigned)(bh * groups)), dim3(256), LDS, stream, a, (int)groups);
        hipError_t e2 = hipGetLastError();
        return e2 == hipSuccess ? DM_OK : (int)e2;
    }
    hipLaunchKernelGGL((k_attn_fwd_v3<DP, VLATE, false>), dim3((unsigned)n_blocks), dim3(256), LDS, stream, a, 0);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}

// Kernel families: "auto" (default: the dispatch below), or one family forced for A/B runs and the parity tests --
//   "w128"   one wave per SIMD, 512 query rows per workgroup (attn_w128.hip, round 4): 64-wide heads, whole 64-row kv tiles,
//            Sq a multiple of 512
//   "w64"    one wave per SIMD, 256 query rows per workgroup (attn_w64.hip): 64-wide heads, whole 64-row kv tiles
//   "v3l"    4 waves x 32 query rows, LDS-DMA ring (k_attn_fwd_v3): D <= 64 or 96 < D <= 128, any sequence lengths
//   "staged" register-staged generic kernel (k_attn_fwd): any D <= 160 (the SD-1.5 heads of 40 / 80 / 160)
// A forced family falls through to the next one when a shape is outside its domain.
enum { kAttnAuto = 0, kAttnW128 = 1, kAttnW64 = 2, kAttnV3l = 3, kAttnStaged = 4 };
#if !defined(DM_F16)
static int g_attn_mode = -1;
#endif
static int attn_mode_from_name(const char* e) {
    if (!e || !strcmp(e, "auto")) return kAttnAuto;
    if (!strcmp(e, "w128")) return kAttnW128;
    if (!strcmp(e, "w64")) return kAttnW64;
    if (!strcmp(e, "v3l")) return kAttnV3l;
    if (!strcmp(e, "staged")) return kAttnStaged;
    return -1;
}
#if defined(DM_F16)
// one selection for the process: the f16 instantiation follows what dm_attention_select / the environment chose (bf16 object)
extern "C" const char* dm_attention_selected(void);
static int attn_mode() { return attn_mode_from_name(dm_attention_selected()); }
#else
static int attn_mode() {
    if (g_attn_mode < 0) {
        g_attn_mode = attn_mode_from_name(getenv("DREAMMAT_ATTN_KERNEL"));
        if (g_attn_mode < 0) g_attn_mode = kAttnAuto;
    }
    return g_attn_mode;
}
static const char* const kAttnNames[] = {"auto", "w128", "w64", "v3l", "staged"};
#endif

// the buffer-descriptor DMA addresses one (batch, head) operand with 32-bit byte offsets
static bool attn_v3_ok(const AttnArgs& a) {
    const long long skv_pad8 = (a.Skv + 7) & ~7;
    const long long kb = (((long long)a.Skv + kKvTile) * a.k_ss + a.D) * 2, vb = (((long long)a.D - 1) * a.vt_ds + skv_pad8 + kKvTile) * 2;
    return kb < 0x40000000LL && vb < 0x40000000LL && a.k_ss >= a.D && a.vt_ds >= skv_pad8;
}

template <int DP, bool VNAT>
int launch_attn_t(const AttnArgs& a, hipStream_t stream) {
    constexpr int KROW = DP * 2 + 16, VROW = kKvTile * 2 + 16;
    constexpr int LDS = 2 * (kKvTile * KROW + (VNAT ? kKvTile * KROW : DP * VROW));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<DP, VNAT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(dm_div_up(a.Sq, kWaves * kQRowsPerWave), a.B * a.Hh);
    DM_ENTER();
    hipLaunchKernelGGL((k_attn_fwd<DP, VNAT>), grid, dim3(256), LDS, stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}
template <int DP>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
    return a.v_nat ? launch_attn_t<DP, true>(a, stream) : launch_attn_t<DP, false>(a, stream);
}

}  // namespace

extern "C" {

#if !defined(DM_F16)
// Selects the attention kernel family by name ("auto", "w128", "w64", "v3l", "staged"; NULL = DREAMMAT_ATTN_KERNEL / "auto") for
// every later dm_attention_fwd_bf16 call of the process.  Returns DM_OK or DM_ERR_ARG for an unknown name.  Meant for
// A/B measurements and for the parity tests, which run every family.
int dm_attention_select(const char* name) {
    if (!name) { g_attn_mode = -1; return DM_OK; }
    const int m = attn_mode_from_name(name);
    if (m < 0) return DM_ERR_ARG;
    g_attn_mode = m;
    return DM_OK;
}

// Name of the family the next dm_attention_fwd_bf16 call dispatches from (what dm_attention_select / the environment chose).
const char* dm_attention_selected(void) { return kAttnNames[attn_mode()]; }
#endif

// q   [B, Sq, Hh, D]  via strides (q_bs, q_ss, q_hs), d contiguous
// k   [B, Skv, Hh, D] via strides
// vt  [B, Hh, D, Skv_pad] via strides (vt_bs, vt_hs, vt_ds), kv contiguous, rows zero-padded to a
//     multiple of 8 kv (vt_ds % 8 == 0)
// out [B, Sq, Hh, D]  via strides.  All pointers 16 B aligned, all strides multiples of 8 elements
// (out: multiples of 4).  D % 8 == 0, D <= 160.  scale = softmax scale (1/sqrt(D) for diffusers).
static int attention_fwd(const void* q, const void* k, const void* vt, const void* v_nat, void* out, float* lse, int B, int Hh, int Sq, int Skv,
                         int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss,
                         long long k_hs, long long vt_bs, long long vt_hs, long long vt_ds, long long o_bs,
                         long long o_ss, long long o_hs, float scale, hipStream_t stream) {
    if (!q || !k || (!vt && !v_nat) || !out || B <= 0 || Hh <= 0 || Sq <= 0 || Skv <= 0 || D <= 0) return DM_ERR_ARG;
    if (D % 8 != 0 || D > 160) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)v_nat) & 15 || ((uintptr_t)out & 7)) return DM_ERR_ARG;
    if ((q_bs | q_ss | q_hs | k_bs | k_ss | k_hs | vt_bs | vt_hs | vt_ds) & 7) return DM_ERR_ARG;
    if ((o_bs | o_ss | o_hs) & 3) return DM_ERR_ARG;
    if ((long long)B * Hh > 65535) return DM_ERR_UNSUPPORTED;
    AttnArgs a;
    a.q = (const elem_t*)q; a.k = (const elem_t*)k; a.vt = (const elem_t*)vt; a.out = (elem_t*)out;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = q_hs; a.k_bs = k_bs; a.k_ss = k_ss; a.k_hs = k_hs;
    a.vt_bs = vt_bs; a.vt_hs = vt_hs; a.vt_ds = vt_ds; a.o_bs = o_bs; a.o_ss = o_ss; a.o_hs = o_hs;
    a.B = B; a.Hh = Hh; a.Sq = Sq; a.Skv = Skv; a.D = D;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.timeline = nullptr;
    a.lse = lse;
    a.v_nat = (const elem_t*)v_nat;
    const int mode = (lse || v_nat) ? kAttnStaged : attn_mode();     // row statistics / natural V: the generic kernel only
    // auto: the one-wave-per-SIMD kernel wherever a workgroup's 256 query rows are (nearly) filled and the sequence is long
    // enough to amortise its prologue (S >= 1024: 756 vs 738 TF/s at S = 1024, 1021 vs 980 at 4096, 981 vs 780 at batch 3;
    // v3l wins at S = 256: 375 vs 343, profiles/r03_probe_attn.json); cross-attention (77 keys) and short sequences: v3l
    // auto: 128 rows per wave (attn_w128.hip) when the halved workgroup count still covers the chip at least once
    if (mode <= kAttnW128 && attn_w128_ok(a) &&
        (mode == kAttnW128 || (a.Skv >= 1024 && (long long)a.B * a.Hh * (a.Sq / 512) >= 256)))
        return launch_attn_w128(a, stream);
    if (mode <= kAttnW64 && attn_w64_ok(a) && (mode == kAttnW64 ? a.Sq >= 128 : (a.Sq >= 1024 && a.Skv >= 1024)))
        return launch_attn_w64(a, stream);
    if (mode <= kAttnV3l && attn_v3_ok(a) && (D <= 64 || (D > 96 && D <= 128)))
        return D <= 64 ? launch_attn_v3<64, true>(a, stream) : launch_attn_v3<128, false>(a, stream);
    if (D <= 64) return launch_attn<64>(a, stream);
    if (D <= 96) return launch_attn<96>(a, stream);
    if (D <= 128) return launch_attn<128>(a, stream);
    return launch_attn<160>(a, stream);
}

int DM_T(dm_attention_fwd_, )(const void* q, const void* k, const void* vt, void* out, int B, int Hh, int Sq, int Skv,
                          int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs, long long k_ss,
                          long long k_hs, long long vt_bs, long long vt_hs, long long vt_ds, long long o_bs,
                          long long o_ss, long long o_hs, float scale, hipStream_t stream) {
    return attention_fwd(q, k, vt, nullptr, out, nullptr, B, Hh, Sq, Skv, D, q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, vt_bs, vt_hs,
                         vt_ds, o_bs, o_ss, o_hs, scale, stream);
}

// The forward of a DIFFERENTIATED attention (dm_attention_bwd_bf16, attn_bwd.hip): v is [B, Skv, Hh, D] with k's strides (NOT
// transposed: the generic kernel reads its V^T fragments with the transposing LDS read), plus
// lse [B, Hh, Sq] fp32 = rowmax + log2(rowsum) of the scaled scores in the log2 domain, from which the backward recomputes
// the probabilities.  Always the register-staged generic kernel (the only one that keeps the unscaled running maximum).
int DM_T(dm_attention_fwd_lse_, )(const void* q, const void* k, const void* v, void* out, float* lse, int B, int Hh, int Sq,
                              int Skv, int D, long long q_bs, long long q_ss, long long q_hs, long long k_bs,
                              long long k_ss, long long k_hs, long long o_bs, long long o_ss, long long o_hs, float scale,
                              hipStream_t stream) {
    if (!lse || !v) return DM_ERR_ARG;
    return attention_fwd(q, k, nullptr, v, out, lse, B, Hh, Sq, Skv, D, q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, 8, 8, 8, o_bs, o_ss,
                         o_hs, scale, stream);
}

}  // extern "C"
