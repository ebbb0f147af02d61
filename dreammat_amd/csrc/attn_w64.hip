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
//     ("chunk"), pinned with sched_barrier so that the in-order wave alternates the two pipes.  tools/issue_probe.cpp prices
//     that chunk alone at 37.6 cycles (v_exp_f32 9.4, v_add_f32 5.5, v_cvt_pk 6.4 per instruction from one wave; 5 plain VALU
//     hide behind a 32-cycle MFMA, two exponentials and three do not), i.e. 1203 cycles per tile against 1024 of MFMA; the
//     kernel measures 1808 (s_memtime, DREAMMAT_ATTN_TIMELINE): the other ~600 are the 16 fragment reads, 4 LDS-DMA issues,
//     address / ring arithmetic and the per-tile barrier;
//   * ONE score register set: the first kv half of tile j+1 is formed in the second half of tile j, into the registers whose
//     scores have just been packed, the second half of tile j in its own first half -- 64 instead of 128 VGPRs, which is what
//     lets everything VALU-addressable fit the 256 architectural VGPRs without spills;
//   * the fragments are re-loaded just in time: a K fragment register is refilled with tile j+2's data right after its last
//     use for tile j+1 (V^T likewise), so no LDS latency is ever exposed and no second fragment set is needed;
//   * NO row maximum in the loop.  softmax is shift-invariant, so any per-row shift m works as long as nothing overflows:
//     m is the exact maximum of the FIRST tile (prologue) and enters through the C operand of the first QK^T MFMA
//     (S' = K.Q'^T - m, Q' = Q * scale * log2 e), which leaves one bare v_exp_f32 per score.  After each tile one compare of the
//     running row sum against 2^30 decides (wave-uniform, rare) whether to re-base: shift = exponent of the row sum, applied to
//     O, l, the C operand and the already formed S'(j+1).  A row whose scores outgrow the first tile's maximum by more than
//     ~2^100 between two checks is caught by a finiteness test and the WHOLE workgroup redoes its block with the textbook
//     online softmax (exact path below; tests force it);
//   * row sums as f32 adds, two chains per query block.  Measured and dropped: the sums on the matrix pipe (two
//     v_mfma_f32_16x16x32_bf16 per packed slice against a 0/1 A operand: 34 instructions fewer per tile, 2 % SLOWER; a
//     v_mfma_f32_4x4x4_16b_bf16 form 15 % slower) -- the small MFMAs serialise with the 32x32 ones they sit between;
//   * O leaves through LDS as whole 128-byte rows (16 B per lane) instead of 8-byte pieces at a row stride.
// Layouts (K rows with index bits 2/3 swapped, XOR-swizzled 16 B chunks, accumulator order == B-operand order of the
// second product) are those of attention.hip's k_attn_fwd_v3; tests/mfma_sim.py models them on the CPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "attn_common.h"

namespace dm_attn {
namespace {

constexpr int kTile = 64;          // kv rows per tile
constexpr int kStage = 16384;      // K tile (64 x 64 bf16) + V^T tile (64 x 64 bf16)
[[maybe_unused]] constexpr int kKBytes = 8192;
[[maybe_unused]] constexpr int kL = 4;              // LDS-DMA instructions per wave per tile: 2 K + 2 V^T
constexpr int kRowsPerWg = 256;

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int PD>
__global__ __launch_bounds__(256, 1) void k_attn_fwd_w64(AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t does not exist in the host pass)
    constexpr int NST = PD + 3;        // ring: tiles j (V^T) .. j+2+PD, and the DMA target is the stage tile j-1 left an iteration ago
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto stamp = [&](int i) {        // development aid: four s_memtime stamps per wave (entry, loop start, loop end, exit)
        if (a.timeline && lane == 0 && blockIdx.x < 1024) a.timeline[((long long)blockIdx.x * 4 + wave) * 4 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
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

    // ---- LDS-DMA: one descriptor per operand (base = this batch / head), per-lane offsets loop-invariant
    const elem_t* kp = a.k + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const elem_t* vp = a.vt + (long long)b * a.vt_bs + (long long)h * a.vt_hs;
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
    // piece 0, 1: this wave's two 1 KiB slices of the K tile; 2, 3: of the V^T tile
    auto issue_piece = [&](int tile, int stage, int i) {
        const int tc = min(tile, n_tiles - 1);            // tiles past the end re-fetch the last one (keeps the vmcnt counts
        char* sb = smem + stage * kStage;                 // uniform; their stage is never read)
        if (i < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(sb + (wave * 2 + i) * 1024),
                                                     16, k_voff[i], tc * k_tile_bytes, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(sb + kKBytes + (wave * 2 + i - 2) * 1024),
                                                     16, v_voff[i - 2], tc * (kTile * 2), 0, 0);
    };
    auto issue = [&](int tile, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(tile, stage, i);
    };
    // the first PD + 2 tiles are requested BEFORE the Q rows (round 4): the K / V^T stream and the row-strided, latency-bound Q
    // loads of a fresh workgroup overlap instead of queueing
#pragma unroll
    for (int t = 0; t < PD + 2; ++t) issue(t, t);
    // ---- Q' fragments (B operand of S^T = K.Q'^T): lane holds Q'[row0 + 32 qb + l31][16 kk + 8 hi .. +7]
    elem8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = min(row0 + 32 * qb + l31, a.Sq - 1);     // rows past the end duplicate the last one (never stored)
        const elem_t* qp = a.q + (long long)b * a.q_bs + (long long)row * a.q_ss + (long long)h * a.q_hs;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[qb][kk] = __builtin_bit_cast(elem8, *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * hi));
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x2 two = {(float)qf[qb][kk][e] * a.scale_log2, (float)qf[qb][kk][e + 1] * a.scale_log2};
                elem2 pk = __builtin_convertvector(two, elem2);
                qf[qb][kk][e] = pk[0];
                qf[qb][kk][e + 1] = pk[1];
            }
        }
    }

    // fragment addresses inside a stage: K (t, kk) at t*4096 + off4[kk], V^T (dt, ks) at 8192 + dt*4096 + off4[ks]
    int off4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) off4[i] = l31 * 128 + ((((2 * i + hi) ^ ((l31 >> 1) & 7))) << 4);
    auto frag = [&](const char* base, int off) { return *reinterpret_cast<const elem8*>(base + off); };
    // The loop is unrolled over the ring so that every stage is a compile-time constant: a fragment read is then ONE
    // ds_read_b128 with an immediate offset on a loop-invariant address register (8 v_add_u32 + ~20 SALU per tile before).
    // A DS offset has 16 bits; stage 4 lies past it and gets its own address registers.
    const char* fb[4];
    const char* fb4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fb[i] = smem + off4[i]; fb4[i] = smem + off4[i] + 65536; }
    auto sfrag = [&](auto stage_tag, int i, int off) {       // fragment i of `stage`, `off` bytes into the stage
        constexpr int ST = decltype(stage_tag)::value;
        if (ST * kStage + 12288 + 4096 <= 65536) return *reinterpret_cast<const elem8*>(fb[i] + ST * kStage + off);
        return *reinterpret_cast<const elem8*>(fb4[i] + (ST * kStage - 65536) + off);
    };

    // ---- state
    f32x16 o[2][2];                  // [query block][32-row block of head_dim]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][d][r] = 0.f;
    f32x16 s[2][2];                  // scores [query block][kv half] -- ONE set: half t = 0 of tile j+1 is formed while half
                                     // t = 1 of tile j is consumed, and the other way round (see tile_body)
    f32x16 cin[2];                   // C operand of the first QK^T MFMA of a chain: -m of the lane's row, in all 16 registers
    elem8 kf[8], vf[2][4], pf[2];   // K fragments (t*4+kk), V^T fragments [dt][ks], packed P of the slice in flight (two buffers)
    float lA[2] = {0.f, 0.f}, lB[2] = {0.f, 0.f};
    bool bad = false;

    // ---- prologue: tiles 0 .. PD+1 in flight; S'(0), first kv half, with the exact row maximum over those 32 keys as the shift
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * kL) : "memory");       // tiles 0 and 1 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kf[kk] = frag(smem, off4[kk]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][0][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            s[qb][0] = DM_MFMA_32x32x16(kf[kk], qf[qb][kk], s[qb][0]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = frag(smem + kStage, off4[kk]);                  // K(1), first half
        kf[4 + kk] = frag(smem, 4096 + off4[kk]);                // K(0), second half
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[dt][ks] = frag(smem + kKBytes, dt * 4096 + off4[ks]);    // V^T(0)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float mq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) mq[c] = s[qb][0][c];
#pragma unroll
        for (int r = 4; r < 16; ++r) mq[r & 3] = fmaxf(mq[r & 3], s[qb][0][r]);
        float mx = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) { cin[qb][r] = -mx; s[qb][0][r] -= mx; }
    }

    // O^T += V^T.P^T with the accumulator and the V^T fragment pinned to the AGPR half of the register file (VALU instructions
    // address only the 256 architectural VGPRs; O is touched by nothing but these MFMAs and the rare re-base until the
    // epilogue).  Operands are at least one chunk old.
    auto pv_mfma = [](f32x16& acc, const elem8& va, const elem8& pb) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "a"(va), "v"(pb));
    };
    // One tile j, 33 "chunks" of one MFMA + two exponentials (scores 2c, 2c+1 of the tile; slice p = c / 4 is query block
    // p & 1, 16-wide k-step p >> 1, kv half t = p >> 2) + the sums and the bf16 pack of chunk c-1 (skewed by one: nothing
    // waits on a v_exp result).  The MFMAs of period p = c / 4:
    //   chunks 4p, 4p+1   QK^T with one K fragment, both query blocks:  p < 4: half t = 1 of THIS tile (fragment 4+p), consumed
    //                     from period 4 on;  p >= 4: half t = 0 of tile j+1 (fragment p-4) into the registers whose slices
    //                     0..3 have just been packed.  The fragment register is then refilled for the tile after (K(j+1) second
    //                     half from ks1, K(j+2) first half from kb2): just in time, no second fragment set, no exposed LDS latency.
    //   chunks 4p+2, 4p+3 P.V of slice p-1, both head_dim blocks; after query block 1 the V^T fragments of that k-step are
    //                     refilled from tile j+1 (vb1).
    // LAST: nothing of tile j+1 exists.
    auto tile_body = [&](auto s1_tag, auto s2_tag, auto sd_tag, int dma_tile, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int dma_stage = decltype(sd_tag)::value;
#pragma unroll
        for (int c = 0; c <= 32; ++c) {
            const int p = c >> 2, c4 = c & 3;
            // the four LDS-DMA pieces of tile j+2+PD, one every eight chunks (all four at the top of the tile measured the same)
            if (!LAST && (c & 7) == 3) issue_piece(dma_tile, dma_stage, c >> 3);
            if (c < 32) {
                if (c4 < 2) {
                    if (p < 4 || !LAST) {
                        const int tq = p < 4 ? 1 : 0, kq = p & 3, qb = c4, f = p < 4 ? 4 + p : p - 4;
                        if (kq == 0) {
                            // chain head: D = s, C = the loop-invariant C-operand vector.  Through the builtin hipcc ties D
                            // to C and first copies 16 registers; the instruction itself takes distinct ones.
                            asm volatile(DM_MFMA_ASM " %0, %1, %2, %3" : "=&v"(s[qb][tq]) : "v"(kf[f]), "v"(qf[qb][0]), "v"(cin[qb]));
                        } else {
                            s[qb][tq] = DM_MFMA_32x32x16(kf[f], qf[qb][kq], s[qb][tq]);
                        }
                        if (!LAST && c4 == 1) kf[f] = p < 4 ? sfrag(s1_tag, kq, 4096) : sfrag(s2_tag, kq, 0);
                    }
                } else if (p > 0) {                                // P.V of slice p-1, head_dim block c4 - 2
                    const int pp = p - 1, ks = pp >> 1, qb = pp & 1, dt = c4 - 2;
                    pv_mfma(o[qb][dt], vf[dt][ks], pf[pp & 1]);
                    if (!LAST && dt == 1 && qb == 1) {
                        vf[0][ks] = sfrag(s1_tag, ks, kKBytes);
                        vf[1][ks] = sfrag(s1_tag, ks, kKBytes + 4096);
                    }
                }
                {
                    const int qb = p & 1, ks = p >> 1, t = ks >> 1, r = 8 * (ks & 1) + 2 * c4;
                    s[qb][t][r] = __builtin_amdgcn_exp2f(s[qb][t][r]);
                    s[qb][t][r + 1] = __builtin_amdgcn_exp2f(s[qb][t][r + 1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // exponentials first: their consumers are a whole chunk away
            if (c > 0) {
                const int cc = c - 1, pq = cc >> 2, e = cc & 3;
                const int qb = pq & 1, ks = pq >> 1, t = ks >> 1, r = 8 * (ks & 1) + 2 * e;
                const float v0 = s[qb][t][r], v1 = s[qb][t][r + 1];
                f32x2 two = {v0, v1};
                elem2 pk = __builtin_convertvector(two, elem2);
                pf[pq & 1][2 * e] = pk[0];
                pf[pq & 1][2 * e + 1] = pk[1];
                lA[qb] += v0;
                lB[qb] += v1;
                // keep the chains where they are: left alone, the SLP vectoriser gathers the adds into v_pk_add_f32
                asm volatile("" : "+v"(lA[qb]), "+v"(lB[qb]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // P.V of the last slice (query block 1, k-step 3)
        // (s_nop 1: the pack of chunk 31 may sit directly in front -- a VALU-written register needs two wait states before an
        // MFMA reads it, and hipcc pads nothing around an asm statement)
        asm volatile("s_nop 1\n\t" DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(o[1][0]) : "a"(vf[0][3]), "v"(pf[1]));
        pv_mfma(o[1][1], vf[1][3], pf[1]);
        if (!LAST) {
            vf[0][3] = sfrag(s1_tag, 3, kKBytes);
            vf[1][3] = sfrag(s1_tag, 3, kKBytes + 4096);
        }
    };
    auto row_sum = [&](int qb) { return lA[qb] + lB[qb]; };
    // rare, wave-uniform: shift the rows whose sums have grown past 2^20 down to [0.5, 1).  The scores already formed against
    // the old C operand are the first kv half of the next tile.
    auto maybe_rebase = [&]() {
        const float l0 = row_sum(0), l1 = row_sum(1);
        if (__builtin_expect(__any(!(l0 <= DM_P_REBASE_AT) || !(l1 <= DM_P_REBASE_AT)), 0)) {     // (also true for NaN)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float lt = qb ? l1 : l0;
                const float lp = lt + __shfl_xor(lt, 32);                      // both lanes of a row must take the same shift
                const bool fin = lp < DM_P_SUM_MAX;            // (f16: a numerator of this tile may have passed 65504)
                bad = bad || !fin;
                const int e = (fin && lp > DM_P_REBASE_IF) ? __builtin_amdgcn_frexp_expf(lp) : 0;
                const float fe = (float)e;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] = __builtin_ldexpf(o[qb][d][r], -e);
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[qb][0][r] -= fe; cin[qb][r] -= fe; }
                lA[qb] = __builtin_ldexpf(lA[qb], -e);
                lB[qb] = __builtin_ldexpf(lB[qb], -e);
            }
        }
    };

    // ---- main loop
    stamp(1);
    int j = 0;
    auto one = [&](auto r_tag) {                             // tile j with j % NST == R: stages of tiles j+1, j+2, j+2+PD
        constexpr int R = decltype(r_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * kL) : "memory");     // tile j+2 has landed
        __builtin_amdgcn_s_barrier();
        tile_body(std::integral_constant<int, (R + 1) % NST>{}, std::integral_constant<int, (R + 2) % NST>{},
                  std::integral_constant<int, (R + 2 + PD) % NST>{}, j + 2 + PD, std::false_type{});
        maybe_rebase();
        ++j;
    };
    static_assert(NST == 5, "the ring is unrolled by hand below");
    while (j + NST <= n_tiles - 1) {
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{});
    }
    {   // the remaining 0..4 tiles before the last one (j % NST == 0 here)
        const int rem = n_tiles - 1 - j;
        if (rem > 0) one(std::integral_constant<int, 0>{});
        if (rem > 1) one(std::integral_constant<int, 1>{});
        if (rem > 2) one(std::integral_constant<int, 2>{});
        if (rem > 3) one(std::integral_constant<int, 3>{});
    }
    tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, std::true_type{});
    stamp(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the re-fetched tiles past the end must not outlive the workgroup's LDS

    float l_tot[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float lt = row_sum(qb);
        l_tot[qb] = lt + __shfl_xor(lt, 32);
        bad = bad || !(l_tot[qb] < DM_P_SUM_MAX);
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
                f32x16 sx[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sx[t][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        sx[t] = DM_MFMA_32x32x16(kf[t * 4 + kk], qf[qb][kk], sx[t]);
                }
                float mx = sx[0][0];
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(sx[0][r], sx[1][r]));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
                elem8 px[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = ks >> 1, r = 8 * (ks & 1) + 2 * e;
                        const float p0 = __builtin_amdgcn_exp2f(sx[t][r] - m_new), p1 = __builtin_amdgcn_exp2f(sx[t][r + 1] - m_new);
                        l_run[qb] += p0 + p1;
                        f32x2 two = {p0, p1};
                        elem2 pk = __builtin_convertvector(two, elem2);
                        px[ks][2 * e] = pk[0];
                        px[ks][2 * e + 1] = pk[1];
                    }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        o[qb][dt] = DM_MFMA_32x32x16(vf[dt][ks], px[ks], o[qb][dt]);
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
                elem2 y0 = __builtin_convertvector(x0, elem2), y1 = __builtin_convertvector(x1, elem2);
                elem4 y = {y0[0], y0[1], y1[0], y1[1]};
                // head_dim 32 dt + 8 g + 4 hi .. +3: 16 B chunk 4 dt + g (swizzled by the row), half hi
                *reinterpret_cast<elem4*>(ob + row * 128 + (((4 * dt + g) ^ (row & 7)) << 4) + 8 * hi) = y;
            }
    }
    {
        elem_t* op = a.out + (long long)b * a.o_bs + (long long)h * a.o_hs;
        const int chunk = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(ob + row * 128 + ((chunk ^ (row & 7)) << 4));
            if (row0 + row < a.Sq) *reinterpret_cast<uint4*>(op + (long long)(row0 + row) * a.o_ss + chunk * 8) = v;
        }
    }
    stamp(3);
#endif
}

template <int PD>
int launch(const AttnArgs& a_in, hipStream_t stream) {
    AttnArgs a = a_in;
    constexpr int LDS = (PD + 3) * kStage;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd_w64<PD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long n_blocks = (long long)dm_div_up(a.Sq, kRowsPerWg) * a.B * a.Hh;
    if (n_blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    static const bool timeline = getenv("DREAMMAT_ATTN_TIMELINE") != nullptr;
    static unsigned long long* tl_buf = nullptr;
    constexpr int TLN = 1024 * 16;
    if (timeline) {
        if (!tl_buf && hipMalloc(&tl_buf, TLN * 8) != hipSuccess) return DM_ERR_UNSUPPORTED;
        (void)hipMemsetAsync(tl_buf, 0, TLN * 8, stream);
        a.timeline = tl_buf;
    }
    hipLaunchKernelGGL((k_attn_fwd_w64<PD>), dim3((unsigned)n_blocks), dim3(256), LDS, stream, a);
    hipError_t e = hipGetLastError();
    if (timeline && e == hipSuccess) {
        static unsigned long long host[TLN];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(host, tl_buf, sizeof(host), hipMemcpyDeviceToHost);
        const int n_tiles = a.Skv / kTile;
        for (int b : {0, 8, 300, 1000}) {
            if (b >= n_blocks) continue;
            for (int w = 0; w < 4; ++w) {
                const unsigned long long* t = host + ((size_t)b * 4 + w) * 4;
                fprintf(stderr, "[attn timeline] wg %d wave %d: prologue %llu  loop %llu (%.0f / tile, %d tiles)  epilogue %llu\n", b, w,
                        t[1] - t[0], t[2] - t[1], (double)(t[2] - t[1]) / n_tiles, n_tiles, t[3] - t[2]);
            }
        }
        unsigned long long t0 = ~0ull, t1 = 0;
        for (long long i = 0; i < std::min<long long>(n_blocks, 1024) * 4; ++i) { t0 = std::min(t0, host[i * 4]); t1 = std::max(t1, host[i * 4 + 3]); }
        fprintf(stderr, "[attn timeline] first entry -> last exit of the first %lld workgroups: %llu ticks\n", std::min<long long>(n_blocks, 1024), t1 - t0);
    }
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

int launch_attn_w64(const AttnArgs& a, hipStream_t stream) { return launch<2>(a, stream); }

}  // namespace dm_attn
