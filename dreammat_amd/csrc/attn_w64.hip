// Flash-style attention forward for head_dim 64, one wave per SIMD (gfx950) -- the UNet / ControlNet self-attention of
// threestudio/models/guidance/dreammat_guidance.py:205-241, 261-282 at S = 4096 / 1024 / 256 (SD-2.1 heads of 64).
//
// Why another structure (round 3).  The round-2 kernel (4 waves x 32 query rows, two workgroups per CU) sat at 38 % of the
// MFMA peak with SQ_WAIT_INST_ANY 46 %: at D = 64 a score costs as many VALU issue slots (exp, row sum, bf16 pack) as it
// buys MFMA cycles, two waves of a SIMD fight over the same VALU port, and every K / V^T fragment read from LDS feeds ONE
// 32-row block.  Here a workgroup is 4 waves = 256 query rows, each wave owns 64 rows (two 32-row blocks), a whole SIMD and
// the whole 512-register file:
//   * every K / V^T fragment read from LDS multiplies against both query blocks (LDS reads per MFMA halved);
//   * one instruction stream per SIMD, software-pipelined by hand: while the MFMAs of S^T(j+1) = K(j+1).Q^T and of
//     O^T += V^T(j).P(j)^T run, the same wave issues the exponentials / sums / bf16 packs of tile j, two scores per MFMA slot
//     ("chunk"), pinned with sched_barrier so that the in-order wave alternates the two pipes;
//   * the fragments are re-loaded just in time: a K fragment register is refilled with tile j+2's data right after its last
//     use for tile j+1 (V^T likewise), so no LDS latency is ever exposed and no second fragment set is needed;
//   * NO row maximum in the loop.  softmax is shift-invariant, so any per-row shift m works as long as nothing overflows:
//     m is the exact maximum of the FIRST tile (prologue) and enters through the C operand of the first QK^T MFMA
//     (S' = K.Q'^T - m, Q' = Q * scale * log2 e), which leaves one bare v_exp_f32 per score.  After each tile one compare of the
//     running row sum against 2^30 decides (wave-uniform, rare) whether to re-base: shift = exponent of the row sum, applied to
//     O, l, the C operand and the already formed S'(j+1).  A row whose scores outgrow the first tile's maximum by more than
//     ~2^100 between two checks is caught by a finiteness test and the WHOLE workgroup redoes its block with the textbook
//     online softmax (exact path below; tests force it);
//   * row sums either as f32 adds (2 chains per block) or, MSUM, on the matrix pipe: v_mfma_f32_4x4x4_16b_bf16 with an
//     all-ones A operand adds the four packed probabilities of each lane into a per-lane accumulator in one issue slot;
//   * O leaves through LDS as whole 128-byte rows (16 B per lane) instead of 8-byte pieces at a row stride.
// Layouts (K rows with index bits 2/3 swapped, XOR-swizzled 16 B chunks, accumulator order == B-operand order of the
// second product) are those of attention.hip's k_attn_fwd_v3; tests/mfma_sim.py models them on the CPU.
#include <type_traits>

#include "attn_common.h"

namespace dm_attn {
namespace {

constexpr int kTile = 64;          // kv rows per tile
constexpr int kStage = 16384;      // K tile (64 x 64 bf16) + V^T tile (64 x 64 bf16)
constexpr int kKBytes = 8192;
[[maybe_unused]] constexpr int kL = 4;              // LDS-DMA instructions per wave per tile: 2 K + 2 V^T
constexpr int kRowsPerWg = 256;

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int PD, bool MSUM>
__global__ __launch_bounds__(256, 1) void k_attn_fwd_w64(AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t does not exist in the host pass)
    constexpr int NST = PD + 3;        // ring: tiles j (V^T) .. j+2+PD, and the DMA target is the stage tile j-1 left an iteration ago
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: all query blocks of one (batch, head) run on one XCD (its K / V^T are fetched from HBM once)
    int bh, qblk;
    {
        const int nq = (a.Sq + kRowsPerWg - 1) / kRowsPerWg;
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
    const int b = bh / a.Hh, h = bh - b * a.Hh;
    const int n_tiles = a.Skv / kTile;
    const int row0 = qblk * kRowsPerWg + wave * 64;          // first query row of this wave

    // ---- Q' fragments (B operand of S^T = K.Q'^T): lane holds Q'[row0 + 32 qb + l31][16 kk + 8 hi .. +7]
    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = min(row0 + 32 * qb + l31, a.Sq - 1);     // rows past the end duplicate the last one (never stored)
        const __bf16* qp = a.q + (long long)b * a.q_bs + (long long)row * a.q_ss + (long long)h * a.q_hs;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[qb][kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * hi));
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2 two = {(float)qf[qb][kk][e] * a.scale_log2, (float)qf[qb][kk][e + 1] * a.scale_log2};
                bf16x2 pk = __builtin_convertvector(two, bf16x2);
                qf[qb][kk][e] = pk[0];
                qf[qb][kk][e + 1] = pk[1];
            }
        }
    }

    // ---- LDS-DMA: one descriptor per operand (base = this batch / head), per-lane offsets loop-invariant
    const __bf16* kp = a.k + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const __bf16* vp = a.vt + (long long)b * a.vt_bs + (long long)h * a.vt_hs;
    const int k_bytes = (int)((((long long)a.Skv - 1) * a.k_ss + 64) * 2);
    const int v_bytes = (int)((63LL * a.vt_ds + a.Skv) * 2);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, v_bytes, 0x00020000);
    int k_voff[2], v_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);                       // LDS row of this lane's 16 B slot
        const int col = ((lane & 7) ^ ((r >> 1) & 7)) * 8;                     // source column that lands there
        k_voff[i] = (int)(((long long)swap23(r) * a.k_ss + col) * 2);
        const int d = r;                                                       // V^T row (head-dim index)
        v_voff[i] = (int)(((long long)d * a.vt_ds + col) * 2);
    }
    const int k_tile_bytes = (int)(a.k_ss * 2 * kTile);
    auto issue = [&](int tile, int stage) {
        const int tc = min(tile, n_tiles - 1);            // tiles past the end re-fetch the last one (keeps the vmcnt counts
        const int ks = tc * k_tile_bytes, vs = tc * (kTile * 2);   // uniform; their stage is never read)
        char* kb = smem + stage * kStage;
        char* vb = kb + kKBytes;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(kb + (wave * 2 + i) * 1024),
                                                     16, k_voff[i], ks, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(vb + (wave * 2 + i) * 1024),
                                                     16, v_voff[i], vs, 0, 0);
    };
    // fragment addresses inside a stage: K (t, kk) at t*4096 + off4[kk], V^T (dt, ks) at 8192 + dt*4096 + off4[ks]
    int off4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) off4[i] = l31 * 128 + ((((2 * i + hi) ^ ((l31 >> 1) & 7))) << 4);
    auto frag = [&](const char* base, int off) { return *reinterpret_cast<const bf16x8*>(base + off); };

    // ---- state
    f32x16 o[2][2];                  // [query block][32-row block of head_dim]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][d][r] = 0.f;
    f32x16 sA[2][2], sB[2][2];       // scores [query block][kv half], two sets alternating between tiles
    f32x16 cin[2];                   // C operand of the first QK^T MFMA of a chain: -m of the lane's row, in all 16 registers
    bf16x8 kf[8], vf[2][4], pf[2];   // K fragments (t*4+kk), V^T fragments [dt][ks], packed P of the slice in flight (two buffers)
    float lA[2] = {0.f, 0.f}, lB[2] = {0.f, 0.f};
    f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const s16x4 ones4 = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
    bool bad = false;

    // ---- prologue: tiles 0 .. PD+1 in flight, S'(0) with the exact row maximum of tile 0
#pragma unroll
    for (int t = 0; t < PD + 2; ++t) issue(t, t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * kL) : "memory");       // tiles 0 and 1 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int p = 0; p < 8; ++p) kf[p] = frag(smem, (p >> 2) * 4096 + off4[p & 3]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[qb][t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                sA[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t * 4 + kk], qf[qb][kk], sA[qb][t], 0, 0, 0);
        }
#pragma unroll
    for (int p = 0; p < 8; ++p) kf[p] = frag(smem + kStage, (p >> 2) * 4096 + off4[p & 3]);       // K(1)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[dt][ks] = frag(smem + kKBytes, dt * 4096 + off4[ks]);    // V^T(0)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float mq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) mq[c] = fmaxf(sA[qb][0][c], sA[qb][1][c]);
#pragma unroll
        for (int r = 4; r < 16; ++r) mq[r & 3] = fmaxf(fmaxf(mq[r & 3], sA[qb][0][r]), sA[qb][1][r]);
        float mx = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) { cin[qb][r] = -mx; sA[qb][0][r] -= mx; sA[qb][1][r] -= mx; }
    }

    // O^T += V^T.P^T with the accumulator pinned to the AGPR half of the register file: VALU instructions address only the 256
    // architectural VGPRs, which the two score sets (128), the C-operand vectors (32) and the fragments in flight fill; O is
    // touched by nothing but these MFMAs (and the rare re-base) until the epilogue.  Operands are at least one chunk old.
    auto pv_mfma = [](f32x16& acc, const bf16x8& va, const bf16x8& pb) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(va), "v"(pb));
    };
    // One tile: exponentials / sums / packs of the scores in `cur` (tile j), P.V of tile j, and -- unless LAST -- S'(j+1)
    // into `nxt` and the just-in-time reloads of the K fragments (tile j+2, from kb2) and V^T fragments (tile j+1, from vb1).
    // 33 "chunks": chunk c carries one MFMA, the two exponentials of scores 2c, 2c+1 of the tile (slice p = c / 4 is query
    // block p & 1, 16-wide k-step p >> 1) and the sums + pack of chunk c-1 (skewed by one: nothing waits on a v_exp result).
    auto tile_body = [&](f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], const char* kb2, const char* vb1, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
        for (int c = 0; c <= 32; ++c) {
            const int p = c >> 2, c4 = c & 3;
            if (c < 32) {
                if (c4 < 2) {
                    if (!LAST) {                                   // QK^T with fragment p = (kv half p >> 2, k-step p & 3), query block c4
                        const int tq = p >> 2, kq = p & 3, qb = c4;
                        if (kq == 0) {
                            // chain head: D = nxt, C = the loop-invariant C-operand vector.  Through the builtin hipcc ties D
                            // to C and first copies 16 registers; the instruction itself takes distinct ones.
                            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(nxt[qb][tq]) : "v"(kf[p]), "v"(qf[qb][0]), "v"(cin[qb]));
                        } else {
                            nxt[qb][tq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[p], qf[qb][kq], nxt[qb][tq], 0, 0, 0);
                        }
                        if (c4 == 1) kf[p] = frag(kb2, tq * 4096 + off4[kq]);
                    }
                } else if (p > 0) {                                // P.V of slice p-1, head_dim block c4 - 2
                    const int pp = p - 1, ks = pp >> 1, qb = pp & 1, dt = c4 - 2;
                    pv_mfma(o[qb][dt], vf[dt][ks], pf[pp & 1]);
                    if (!LAST && dt == 1 && qb == 1) {
                        vf[0][ks] = frag(vb1, off4[ks]);
                        vf[1][ks] = frag(vb1, 4096 + off4[ks]);
                    }
                }
                {
                    const int qb = p & 1, ks = p >> 1, t = ks >> 1, r = 8 * (ks & 1) + 2 * c4;
                    cur[qb][t][r] = __builtin_amdgcn_exp2f(cur[qb][t][r]);
                    cur[qb][t][r + 1] = __builtin_amdgcn_exp2f(cur[qb][t][r + 1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // exponentials first: their consumers are a whole chunk away
            if (c > 0) {
                const int cc = c - 1, pq = cc >> 2, e = cc & 3;
                const int qb = pq & 1, ks = pq >> 1, t = ks >> 1, r = 8 * (ks & 1) + 2 * e;
                const float v0 = cur[qb][t][r], v1 = cur[qb][t][r + 1];
                f32x2 two = {v0, v1};
                bf16x2 pk = __builtin_convertvector(two, bf16x2);
                pf[pq & 1][2 * e] = pk[0];
                pf[pq & 1][2 * e + 1] = pk[1];
                if (!MSUM) {
                    lA[qb] += v0;
                    lB[qb] += v1;
                    // keep the chains where they are: left alone, the SLP vectoriser gathers the adds into v_pk_add_f32
                    asm volatile("" : "+v"(lA[qb]), "+v"(lB[qb]));
                } else if (e & 1) {
                    bf16x4 four = {pf[pq & 1][2 * e - 2], pf[pq & 1][2 * e - 1], pf[pq & 1][2 * e], pf[pq & 1][2 * e + 1]};
                    lacc[qb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_bit_cast(s16x4, four), lacc[qb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // P.V of the last slice (query block 1, k-step 3)
        // (s_nop 1: the pack of chunk 31 may sit directly in front -- a VALU-written register needs two wait states before an
        // MFMA reads it, and hipcc pads nothing around an asm statement)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[1][0]) : "a"(vf[0][3]), "v"(pf[1]));
        pv_mfma(o[1][1], vf[1][3], pf[1]);
        if (!LAST) {
            vf[0][3] = frag(vb1, off4[3]);
            vf[1][3] = frag(vb1, 4096 + off4[3]);
        }
    };
    auto row_sum = [&](int qb) { return MSUM ? lacc[qb][0] : lA[qb] + lB[qb]; };
    // rare, wave-uniform: shift the rows whose sums have grown past 2^20 down to [0.5, 1)
    auto maybe_rebase = [&](f32x16 (&nxt)[2][2]) {
        const float l0 = row_sum(0), l1 = row_sum(1);
        if (__builtin_expect(__any(!(l0 <= 0x1p30f) || !(l1 <= 0x1p30f)), 0)) {     // (also true for NaN)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float lt = qb ? l1 : l0;
                const float lp = lt + __shfl_xor(lt, 32);                      // both lanes of a row must take the same shift
                const bool fin = lp < 0x1p100f;
                bad = bad || !fin;
                const int e = (fin && lp > 0x1p20f) ? __builtin_amdgcn_frexp_expf(lp) : 0;
                const float fe = (float)e;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[qb][d][r] = __builtin_ldexpf(o[qb][d][r], -e);
                        nxt[qb][d][r] -= fe;
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) cin[qb][r] -= fe;
                lA[qb] = __builtin_ldexpf(lA[qb], -e);
                lB[qb] = __builtin_ldexpf(lB[qb], -e);
#pragma unroll
                for (int r = 0; r < 4; ++r) lacc[qb][r] = __builtin_ldexpf(lacc[qb][r], -e);
            }
        }
    };

    // ---- main loop
    int j = 0, s1 = 1, s2 = 2, sd = (2 + PD) % NST;        // stages of tiles j+1, j+2, j+2+PD
    auto top = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * kL) : "memory");     // tile j+2 has landed
        __builtin_amdgcn_s_barrier();
        issue(j + 2 + PD, sd);
    };
    auto advance = [&]() {
        ++j;
        s1 = s1 + 1 == NST ? 0 : s1 + 1;
        s2 = s2 + 1 == NST ? 0 : s2 + 1;
        sd = sd + 1 == NST ? 0 : sd + 1;
    };
    // two tiles per trip (named score sets, no copies); an odd remainder and the last tile (nothing to prefetch) after the loop
    for (int trips = (n_tiles - 1) >> 1; trips > 0; --trips) {
        top();
        tile_body(sA, sB, smem + s2 * kStage, smem + s1 * kStage + kKBytes, std::false_type{});
        maybe_rebase(sB);
        advance();
        top();
        tile_body(sB, sA, smem + s2 * kStage, smem + s1 * kStage + kKBytes, std::false_type{});
        maybe_rebase(sA);
        advance();
    }
    if (((n_tiles - 1) & 1) != 0) {
        top();
        tile_body(sA, sB, smem + s2 * kStage, smem + s1 * kStage + kKBytes, std::false_type{});
        maybe_rebase(sB);
        advance();
        tile_body(sB, sB, smem, smem, std::true_type{});
    } else {
        tile_body(sA, sA, smem, smem, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the re-fetched tiles past the end must not outlive the workgroup's LDS

    float l_tot[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float lt = row_sum(qb);
        l_tot[qb] = lt + __shfl_xor(lt, 32);
        bad = bad || !(l_tot[qb] < 0x1p100f);
    }
    // ---- exact path (rare): some row overflowed the lazy shift.  The whole workgroup redoes its block with the textbook
    // online softmax, one tile at a time through stage 0.
    if (__syncthreads_or(bad)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][d][r] = 0.f;
        float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
        for (int jt = 0; jt < n_tiles; ++jt) {
            __syncthreads();
            issue(jt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 8; ++p) kf[p] = frag(smem, (p >> 2) * 4096 + off4[p & 3]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) vf[dt][ks] = frag(smem + kKBytes, dt * 4096 + off4[ks]);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 s[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t * 4 + kk], qf[qb][kk], s[t], 0, 0, 0);
                }
                float mx = s[0][0];
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
                bf16x8 px[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = ks >> 1, r = 8 * (ks & 1) + 2 * e;
                        const float p0 = __builtin_amdgcn_exp2f(s[t][r] - m_new), p1 = __builtin_amdgcn_exp2f(s[t][r + 1] - m_new);
                        l_run[qb] += p0 + p1;
                        f32x2 two = {p0, p1};
                        bf16x2 pk = __builtin_convertvector(two, bf16x2);
                        px[ks][2 * e] = pk[0];
                        px[ks][2 * e + 1] = pk[1];
                    }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        o[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][ks], px[ks], o[qb][dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) l_tot[qb] = l_run[qb] + __shfl_xor(l_run[qb], 32);
    }

    // ---- epilogue: normalise, stage this wave's 64 x 64 bf16 block through LDS, store whole 128-byte rows
    __syncthreads();                                          // every wave is done with the K / V^T stages
    char* ob = smem + wave * 8192;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float inv = 1.0f / l_tot[qb];
        const int row = 32 * qb + l31;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x2 x0 = {o[qb][dt][4 * g] * inv, o[qb][dt][4 * g + 1] * inv};
                f32x2 x1 = {o[qb][dt][4 * g + 2] * inv, o[qb][dt][4 * g + 3] * inv};
                bf16x2 y0 = __builtin_convertvector(x0, bf16x2), y1 = __builtin_convertvector(x1, bf16x2);
                bf16x4 y = {y0[0], y0[1], y1[0], y1[1]};
                // head_dim 32 dt + 8 g + 4 hi .. +3: 16 B chunk 4 dt + g (swizzled by the row), half hi
                *reinterpret_cast<bf16x4*>(ob + row * 128 + (((4 * dt + g) ^ (row & 7)) << 4) + 8 * hi) = y;
            }
    }
    {
        __bf16* op = a.out + (long long)b * a.o_bs + (long long)h * a.o_hs;
        const int chunk = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(ob + row * 128 + ((chunk ^ (row & 7)) << 4));
            if (row0 + row < a.Sq) *reinterpret_cast<uint4*>(op + (long long)(row0 + row) * a.o_ss + chunk * 8) = v;
        }
    }
#endif
}

template <int PD, bool MSUM>
int launch(const AttnArgs& a, hipStream_t stream) {
    constexpr int LDS = (PD + 3) * kStage;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd_w64<PD, MSUM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long n_blocks = (long long)dm_div_up(a.Sq, kRowsPerWg) * a.B * a.Hh;
    if (n_blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL((k_attn_fwd_w64<PD, MSUM>), dim3((unsigned)n_blocks), dim3(256), LDS, stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}

}  // namespace

// D = 64 exactly, whole kv tiles, 16-byte rows, 32-bit byte offsets inside one (batch, head) operand
bool attn_w64_ok(const AttnArgs& a) {
    if (a.D != 64 || a.Skv < kTile || (a.Skv & (kTile - 1)) != 0) return false;
    const long long kb = (((long long)a.Skv + kTile) * a.k_ss + 64) * 2, vb = (63LL * a.vt_ds + a.Skv + kTile) * 2;
    return kb < 0x40000000LL && vb < 0x40000000LL && a.k_ss >= 64 && a.vt_ds >= a.Skv && (a.o_ss & 7) == 0 && (a.o_hs & 7) == 0 &&
           (a.o_bs & 7) == 0 && (((uintptr_t)a.out) & 15) == 0;
}

int launch_attn_w64(const AttnArgs& a, int variant, hipStream_t stream) {
    return (variant & 1) ? launch<2, true>(a, stream) : launch<2, false>(a, stream);
}

}  // namespace dm_attn
