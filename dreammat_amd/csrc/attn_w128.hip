// Flash-style attention forward for head_dim 64, one wave per SIMD, 128 QUERY ROWS PER WAVE (gfx950, round 4) -- the UNet /
// ControlNet self-attention of threestudio/models/guidance/dreammat_guidance.py:205-241, 261-282 at S = 4096 (and any Sq that
// is a multiple of 512 with enough workgroups to fill the chip).  The structure is attn_w64.hip's (read its header first:
// swapped products, one skewed score set, exponentials beside the MFMAs, lazy softmax shift, LDS-DMA ring); what changes:
//   * a wave owns FOUR 32-row query blocks instead of two.  Everything a kv tile costs that does not scale with the scores --
//     16 fragment reads, 4 LDS-DMA issues (~60 cycles each), the counted waits and the barrier: ~450 of the 1658 cycles per
//     tile measured on the 64-row kernel -- is paid once per tile for twice the MFMAs, and a (batch, head) needs half the
//     workgroups (prologue + epilogue ~9 k cycles each);
//   * registers: 128 score registers + O (128) + Q' (64) leave no room for a whole tile's 16 fragments and four shift vectors.
//     K / V^T fragments are loaded JUST IN TIME into a rotation of 4 + a double buffer of 2 x 2 (a fragment multiplies four
//     query blocks in consecutive chunks and then idles for most of a tile), and ONE C-operand vector carries the shift of all
//     four rows of a lane (m = the largest of their first-tile maxima: softmax is invariant under any per-row shift, a shared
//     one scales a row's P, l and O by an exact power of two);
//   * NO re-base inside the loop.  The 64-row kernel shifts O / l / the scores down when a row sum passes 2^30; that rare
//     branch redefines every O register with VALU code in the middle of the loop, and with 512 registers in use the allocator
//     answered with ~110 register moves + spills per 5 tiles at the loop's back edge (no branch: 434 instructions per tile, no
//     spill, no move).  fp32 holds exp2(s - m) up to s - m = 127 (88 nats above the first 32 keys' maximum): rows that stay
//     below that are exact without any re-base; a row that does not is caught by the finiteness test (l in (2^-100, 2^100))
//     and the WHOLE workgroup redoes its block with the textbook online softmax (tests force it).
// Layouts are attn_w64.hip's; tests/mfma_sim.py models them on the CPU.
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

// compile-time loop: the tile body is 8 NQ + 1 chunks whose register indices must all be constants -- `#pragma unroll` gave up
// on the 65-trip loop of NQ = 4 (size threshold) and the score arrays went to scratch
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// NQ = 32-row query blocks per wave (2: 64 rows per wave, 256 per workgroup; 4: 128 / 512, round 4).  Every per-tile cost that
// does not scale with the scores -- the 16 K / V^T fragment reads, the 4 LDS-DMA issues (~60 cycles each), the waits and the
// barrier: ~450 of the 1658 cycles a 64-row tile measured in round 3 -- is paid once per tile whatever NQ is, so NQ = 4 spends
// it on twice the MFMAs; it also halves the number of workgroups (prologue + epilogue ~9 k cycles each).  The price is the
// register file: 128 score + 64 C-operand registers fill the VALU-addressable half, so O (128), Q' (64) and ALL K / V^T
// fragments (64) live in the accumulator half and every MFMA is written as asm with "a" operands.
template <int PD, int NQ>
__global__ __launch_bounds__(256, 1) void k_attn_fwd_w128(AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t does not exist in the host pass)
    constexpr int kRowsPerWave = 32 * NQ, kRowsPerWg = 4 * kRowsPerWave;
    constexpr int NS = 4 * NQ;         // 32 x 16 score slices per tile and wave: slice p = query block p % NQ, k-step p / NQ
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
    const int row0 = qblk * kRowsPerWg + wave * kRowsPerWave;          // first query row of this wave

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
    auto issue_piece = [&](int tile, int stage, int i) __attribute__((always_inline)) {
#if defined(DM_ABL_NODMA)
        if (tile > 4) return;                             // ABLATION (wrong results): K / V^T requested for the first tiles only
#endif
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
    // the first PD + 2 tiles are requested BEFORE the Q rows: the K / V^T stream and the (row-strided, latency-bound) Q loads
    // of a fresh workgroup overlap instead of queueing (prologue 12.2 k cycles of a 197 k-cycle workgroup with Q first)
#pragma unroll
    for (int t = 0; t < PD + 2; ++t) issue(t, t);
    // ---- Q' fragments (B operand of S^T = K.Q'^T): lane holds Q'[row0 + 32 qb + l31][16 kk + 8 hi .. +7]
    elem8 qf[NQ][4];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
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

    // Q' moves to the accumulator half ONCE, explicitly: every later use is an "a" operand of an asm MFMA.  (Left to the register
    // allocator, a value with one VGPR-class use -- a builtin MFMA in the prologue -- stayed in VGPRs / scratch and was copied
    // into AGPRs in front of every tile: 64 v_accvgpr_write + 1200 scratch accesses per tile at NQ = 4.)
    elem8 qa[NQ][4];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // (an empty statement whose output is tied to its input in class "a": the compiler emits the four v_accvgpr_write,
            // with their hazards, and the value is an accumulator-file value from here on)
            asm volatile("" : "=a"(qa[qb][kk]) : "0"(qf[qb][kk]));
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
    auto sfrag = [&](auto stage_tag, int i, int off) __attribute__((always_inline)) {       // fragment i of `stage`, `off` bytes into the stage
        constexpr int ST = decltype(stage_tag)::value;
        if (ST * kStage + 12288 + 4096 <= 65536) return *reinterpret_cast<const elem8*>(fb[i] + ST * kStage + off);
        return *reinterpret_cast<const elem8*>(fb4[i] + (ST * kStage - 65536) + off);
    };

    // ---- state
    f32x16 o[NQ][2];                 // [query block][32-row block of head_dim]
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][d][r] = 0.f;
    constexpr bool JIT = NQ > 2;     // (see the fragment registers below)
    f32x16 s[NQ][2];                 // scores [query block][kv half] -- ONE set: half t = 0 of tile j+1 is formed while half
                                     // t = 1 of tile j is consumed, and the other way round (see tile_body)
    // C operand of the first QK^T MFMA of a chain: -m of the lane's row, in all 16 registers.  NQ = 2: one vector per query
    // block.  NQ = 4: ONE vector for the lane's four rows (m = the largest of their four first-tile maxima) -- softmax is
    // invariant under any per-row shift, a common one only scales a row's P, l and O by an exact power of two (bf16 and fp32
    // roundings are scale-invariant), and it returns 48 VALU-addressable registers, without which the allocator spills in the
    // loop.  What a larger-than-needed shift could do is underflow a whole row; the finiteness test below therefore also
    // rejects l <= 2^-100 and sends the workgroup to the exact path.
    constexpr int NCIN = JIT ? 1 : NQ;
    f32x16 cin[NCIN];
    // NQ = 2 holds the whole tile's fragments (8 K + 8 V^T = 64 registers, each refilled right after its last use).  NQ = 4
    // has no room for that beside O (128) and Q' (64) -- the allocator spilled 1185 registers -- and does not need it: a
    // fragment multiplies four query blocks in consecutive chunks and then idles for most of a tile.  JIT = a rotation of 4 K
    // fragments (the one of "pair" g = 2 p / NQ sits in kr[g & 3] and is loaded two pairs = 16 chunks ahead) and a double buffer
    // of the two V^T fragments of a k-step (vr[ks & 1][dt], loaded a k-step ahead): 32 registers.
    elem8 kf[JIT ? 1 : 8], vf[2][JIT ? 1 : 4], pf[2];   // K fragments (t*4+kk), V^T fragments [dt][ks], packed P of the slice in flight (two buffers)
    elem8 kr[4], vr[2][2];
    float lA[NQ], lB[NQ];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) lA[qb] = lB[qb] = 0.f;
    bool bad = false;
    // every MFMA of the kernel: A / B operands from the accumulator half ("a"), so that hipcc keeps Q', K and V^T there
    auto qk_head = [](f32x16& d, const elem8& ka, const elem8& qb_, const f32x16& c) {     // chain head: D = K.Q'^T + C (distinct registers)
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %3" : "=&v"(d) : "a"(ka), "a"(qb_), "v"(c));
    };
    auto qk_acc = [](f32x16& d, const elem8& ka, const elem8& qb_) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %0" : "+v"(d) : "a"(ka), "a"(qb_));
    };
    // NQ = 4: the shared shift vector leaves ~90 VALU-addressable registers free while O + Q' fill 192 of the 256 accumulator
    // registers -- the K / V^T fragments therefore live in VGPRs there ("v"), which gives the allocator slack on BOTH sides
    // (with them in AGPRs it spilled Q' fragments and shuffled ~110 registers at the loop's back edge).
    auto qk_head_v = [](f32x16& d, const elem8& ka, const elem8& qb_, const f32x16& c) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %3" : "=&v"(d) : "v"(ka), "a"(qb_), "v"(c));
    };
    auto qk_acc_v = [](f32x16& d, const elem8& ka, const elem8& qb_) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(ka), "a"(qb_));
    };
    auto pv_mfma_v = [](f32x16& acc, const elem8& va, const elem8& pb) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(va), "v"(pb));
    };
    // a 16-pass MFMA result read (or overwritten) by a VALU instruction needs 18 wait states after the MFMA's issue; hipcc
    // pads that for its own MFMAs only.  Used outside the main loop (there every consumer is >= a chunk of MFMAs away).
    auto settle = [](f32x16& d) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(d)); };
    auto settle_a = [](f32x16& d) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(d)); };

    // ---- prologue: tiles 0 .. PD+1 in flight; S'(0), first kv half, with the exact row maximum over those 32 keys as the shift
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * kL) : "memory");       // tiles 0 and 1 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kr[kk] = frag(smem, off4[kk]);  // K(0), first half
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][0][r] = 0.f;
        asm volatile("s_nop 1" : "+v"(s[qb][0]));                  // VALU-written C -> MFMA: two wait states (nothing pads an asm MFMA)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qk_acc_v(s[qb][0], kr[kk], qa[qb][kk]);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) settle(s[qb][0]);              // MFMA result -> VALU (the row maximum below)
    if constexpr (JIT) {
        kr[0] = frag(smem, 4096 + off4[0]);                      // pairs 0, 1 of tile 0: K(0), second half, k-steps 0, 1
        kr[1] = frag(smem, 4096 + off4[1]);
        vr[0][0] = frag(smem + kKBytes, off4[0]);                // V^T(0), k-step 0
        vr[0][1] = frag(smem + kKBytes, 4096 + off4[0]);
    } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = frag(smem + kStage, off4[kk]);              // K(1), first half
            kf[4 + kk] = frag(smem, 4096 + off4[kk]);            // K(0), second half
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) vf[dt][ks] = frag(smem + kKBytes, dt * 4096 + off4[ks]);    // V^T(0)
    }
    {
        float mxq[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
            float mq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) mq[c] = s[qb][0][c];
#pragma unroll
            for (int r = 4; r < 16; ++r) mq[r & 3] = fmaxf(mq[r & 3], s[qb][0][r]);
            float mx = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
            mxq[qb] = fmaxf(mx, __shfl_xor(mx, 32));
        }
        if constexpr (JIT) {
            float mc = mxq[0];
#pragma unroll
            for (int qb = 1; qb < NQ; ++qb) mc = fmaxf(mc, mxq[qb]);
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) mxq[qb] = mc;
        }
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][0][r] -= mxq[qb];
#pragma unroll
        for (int i = 0; i < NCIN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cin[i][r] = -mxq[i];
    }

    // O^T += V^T.P^T with the accumulator and the V^T fragment pinned to the AGPR half of the register file (VALU instructions
    // address only the 256 architectural VGPRs; O is touched by nothing but these MFMAs and the rare re-base until the
    // epilogue).  Operands are at least one chunk old.
    auto pv_mfma = [](f32x16& acc, const elem8& va, const elem8& pb) {
        asm volatile(DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "a"(va), "v"(pb));
    };
    // One tile j, 8 NQ + 1 "chunks" of one MFMA + two exponentials (scores 2c, 2c+1 of the tile; slice p = c / 4 is query block
    // p % NQ, 16-wide k-step ks = p / NQ, kv half t = ks >> 1) + the sums and the bf16 pack of chunk c-1 (skewed by one: nothing
    // waits on a v_exp result).  The MFMAs of period p = c / 4:
    //   chunks 4p, 4p+1   QK^T with K fragment g = 2 p / NQ against query blocks (2 p) % NQ + {0, 1}:  g < 4: half t = 1 of THIS
    //                     tile (fragment 4+g), consumed from the second half of the periods on;  g >= 4: half t = 0 of tile j+1
    //                     (fragment g-4) into the registers whose slices have just been packed.  After its last query block the
    //                     fragment register is refilled for the tile after (K(j+1) second half from s1, K(j+2) first half from
    //                     s2): just in time, no second fragment set, no exposed LDS latency.
    //   chunks 4p+2, 4p+3 P.V of slice p-1, both head_dim blocks; after the last query block of a k-step the V^T fragments of
    //                     that k-step are refilled from tile j+1 (s1).
    // LAST: nothing of tile j+1 exists.
    // JIT (NQ = 4): the fragment of pair g+2 is requested in the first chunk of pair g, the V^T fragments of k-step ks+1 in the
    // first P.V period of k-step ks; pairs 0..3 / all V^T fragments come from THIS tile's stage (s0), pairs 4..7 and everything
    // "past the end" from the next tile's (s1).  The last tile's stage is a run-time value (`last_off` bytes, plain address
    // arithmetic: 12 reads per workgroup).
    auto tile_body = [&](auto s0_tag, auto s1_tag, auto s2_tag, auto sd_tag, int dma_tile, auto last_tag, int last_off) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int dma_stage = decltype(sd_tag)::value;
        constexpr int NC = 4 * NS;                                 // chunks per tile
        auto own = [&](int i, int off) __attribute__((always_inline)) {     // fragment i at byte `off` of this tile's own stage
            if constexpr (LAST) return frag(smem + last_off, off + off4[i]);
            else return sfrag(s0_tag, i, off);
        };
        static_for<0, NC + 1>([&](auto c_tag) __attribute__((always_inline)) {
            constexpr int c = decltype(c_tag)::value;
            constexpr int p = c >> 2, c4 = c & 3;
            // the four LDS-DMA pieces of tile j+2+PD, spread over the tile (all four at the top of the tile measured the same)
            if (!LAST && (c % (NC / 4)) == 3) issue_piece(dma_tile, dma_stage, c / (NC / 4));
            if (c < NC) {
                if (c4 < 2) {
                    constexpr int g = 2 * p / NQ;                  // K fragment (kv half, k-step) of this period, in issue order
                    if (g < 4 || !LAST) {
                        constexpr int tq = g < 4 ? 1 : 0, kq = g & 3, qb = (2 * p) % NQ + c4, f = g < 4 ? 4 + g : g - 4;
                        if constexpr (JIT) {
                            if (kq == 0) qk_head_v(s[qb][tq], kr[g & 3], qa[qb][0], cin[0]);
                            else qk_acc_v(s[qb][tq], kr[g & 3], qa[qb][kq]);
                            if (qb == 0) {                         // first chunk of pair g: request pair g + 2
                                constexpr int G2 = g + 2;
                                if (G2 < 4) kr[G2 & 3] = own(G2, 4096);                                   // K(j), second half
                                else if (G2 < 8) { if (!LAST) kr[G2 & 3] = sfrag(s1_tag, G2 - 4, 0); }    // K(j+1), first half
                                else if (!LAST) kr[G2 & 3] = sfrag(s1_tag, G2 - 8, 4096);                 // K(j+1), second half
                            }
                        } else {
                            if (kq == 0) qk_head(s[qb][tq], kf[f], qa[qb][0], cin[qb]);     // D = s, C = the loop-invariant -m vector
                            else qk_acc(s[qb][tq], kf[f], qa[qb][kq]);
                            if (!LAST && qb == NQ - 1) kf[f] = g < 4 ? sfrag(s1_tag, kq, 4096) : sfrag(s2_tag, kq, 0);
                        }
                    }
                } else if (p > 0) {                                // P.V of slice p-1, head_dim block c4 - 2
                    constexpr int pp = p - 1, ks = pp / NQ, qb = pp % NQ, dt = c4 - 2;
                    if constexpr (JIT) {
                        pv_mfma_v(o[qb][dt], vr[ks & 1][dt], pf[pp & 1]);
                        if (qb == 0) {                             // first slice of k-step ks: request k-step ks + 1
                            if (ks < 3) vr[(ks + 1) & 1][dt] = own(ks + 1, kKBytes + dt * 4096);
                            else if (!LAST) vr[0][dt] = sfrag(s1_tag, 0, kKBytes + dt * 4096);
                        }
                    } else {
                        pv_mfma(o[qb][dt], vf[dt][ks], pf[pp & 1]);
                        if (!LAST && dt == 1 && qb == NQ - 1) {
                            vf[0][ks] = sfrag(s1_tag, ks, kKBytes);
                            vf[1][ks] = sfrag(s1_tag, ks, kKBytes + 4096);
                        }
                    }
                }
                {
                    constexpr int qb = p % NQ, ks = p / NQ, t = ks >> 1, r = 8 * (ks & 1) + 2 * c4;
#if defined(DM_ABL_NOEXP)
                    asm volatile("" : "+v"(s[qb][t][r]), "+v"(s[qb][t][r + 1]));      // ABLATION (wrong results): no exponentials
#else
                    s[qb][t][r] = __builtin_amdgcn_exp2f(s[qb][t][r]);
                    s[qb][t][r + 1] = __builtin_amdgcn_exp2f(s[qb][t][r + 1]);
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // exponentials first: their consumers are a whole chunk away
            if (c > 0) {
                constexpr int cc = c - 1, pq = cc >> 2, e = cc & 3;
                constexpr int qb = pq % NQ, ks = pq / NQ, t = ks >> 1, r = 8 * (ks & 1) + 2 * e;
                const float v0 = s[qb][t][r], v1 = s[qb][t][r + 1];
                f32x2 two = {v0, v1};
                elem2 pk = __builtin_convertvector(two, elem2);
                pf[pq & 1][2 * e] = pk[0];
                pf[pq & 1][2 * e + 1] = pk[1];
#if !defined(DM_ABL_NOSUM)
                lA[qb] += v0;
                lB[qb] += v1;
#endif
                // keep the chains where they are: left alone, the SLP vectoriser gathers the adds into v_pk_add_f32
                asm volatile("" : "+v"(lA[qb]), "+v"(lB[qb]));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // P.V of the last slice (query block NQ-1, k-step 3)
        // (s_nop 1: the pack of the last chunk may sit directly in front -- a VALU-written register needs two wait states before
        // an MFMA reads it, and hipcc pads nothing around an asm statement)
        if constexpr (JIT) {
            asm volatile("s_nop 1\n\t" DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(o[NQ - 1][0]) : "v"(vr[1][0]), "v"(pf[(NS - 1) & 1]));
            pv_mfma_v(o[NQ - 1][1], vr[1][1], pf[(NS - 1) & 1]);
        } else {
            asm volatile("s_nop 1\n\t" DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(o[NQ - 1][0]) : "a"(vf[0][3]), "v"(pf[(NS - 1) & 1]));
            pv_mfma(o[NQ - 1][1], vf[1][3], pf[(NS - 1) & 1]);
            if (!LAST) {
                vf[0][3] = sfrag(s1_tag, 3, kKBytes);
                vf[1][3] = sfrag(s1_tag, 3, kKBytes + 4096);
            }
        }
    };
    auto row_sum = [&](int qb) { return lA[qb] + lB[qb]; };
    // ---- main loop
    stamp(1);
    int j = 0;
    auto one = [&](auto r_tag) __attribute__((always_inline)) {     // tile j with j % NST == R: stages of tiles j+1, j+2, j+2+PD
        constexpr int R = decltype(r_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * kL) : "memory");     // tile j+2 has landed
#if !defined(DM_ABL_NOBAR)
        __builtin_amdgcn_s_barrier();
#endif
        tile_body(std::integral_constant<int, R>{}, std::integral_constant<int, (R + 1) % NST>{}, std::integral_constant<int, (R + 2) % NST>{},
                  std::integral_constant<int, (R + 2 + PD) % NST>{}, j + 2 + PD, std::false_type{}, 0);
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
    tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{},
              0, std::true_type{}, ((n_tiles - 1) % NST) * kStage);
    stamp(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the re-fetched tiles past the end must not outlive the workgroup's LDS

    float l_tot[NQ];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        const float lt = row_sum(qb);
        l_tot[qb] = lt + __shfl_xor(lt, 32);
        bad = bad || !(l_tot[qb] < DM_P_SUM_MAX) || !(l_tot[qb] > DM_P_SUM_MIN);     // overflow, NaN, or a row underflowed by a shared shift
    }
#if defined(DM_ABL_NOEXP) || defined(DM_ABL_NOSUM) || defined(DM_ABL_NODMA)
    bad = false;                                         // (ablation builds: garbage sums must not send the workgroup down the exact path)
#endif
    // ---- exact path (rare): some row overflowed the lazy shift.  The whole workgroup redoes its block with the textbook
    // online softmax, one tile at a time through stage 0.
    if (__syncthreads_or(bad)) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][d][r] = 0.f;
        float m_run[NQ], l_run[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) { m_run[i] = -1e30f; l_run[i] = 0.f; }
        for (int jt = 0; jt < n_tiles; ++jt) {
            __syncthreads();
            issue(jt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {      // (fragments re-read per query block through the 4 + 2 JIT registers: a rare path)
                f32x16 sx[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sx[t][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) kr[kk] = frag(smem, t * 4096 + off4[kk]);
                    asm volatile("s_nop 1" : "+v"(sx[t]));          // (asm MFMAs: every hazard by hand, see settle)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) qk_acc_v(sx[t], kr[kk], qa[qb][kk]);
                    settle(sx[t]);                                  // (also keeps the next reload of kr behind these MFMAs)
                }

                float mx = sx[0][0];
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(sx[0][r], sx[1][r]));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
                settle_a(o[qb][0]); settle_a(o[qb][1]);            // the previous tile's P.V before the VALU rescales O
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
                for (int ks = 0; ks < 4; ++ks) {
                    vr[0][0] = frag(smem + kKBytes, off4[ks]);
                    vr[0][1] = frag(smem + kKBytes, 4096 + off4[ks]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)     // (s_nop 1: O and the packed P were just written by the VALU)
                        asm volatile("s_nop 1\n\t" DM_MFMA_ASM " %0, %1, %2, %0" : "+a"(o[qb][dt]) : "v"(vr[0][dt]), "v"(px[ks]));
                    settle_a(o[qb][1]);                             // (keeps the next reload of vr behind these MFMAs)
                }
            }
        }
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) { settle_a(o[qb][0]); settle_a(o[qb][1]); }
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) l_tot[qb] = l_run[qb] + __shfl_xor(l_run[qb], 32);
    }

    // ---- epilogue: normalise, stage this wave's (32 NQ) x 64 bf16 block through LDS, store whole 128-byte rows
    __syncthreads();                                          // every wave is done with the K / V^T stages
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) { settle_a(o[qb][0]); settle_a(o[qb][1]); }
    char* ob = smem + wave * (kRowsPerWave * 128);
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
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
        for (int i = 0; i < kRowsPerWave / 8; ++i) {
            const int row = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(ob + row * 128 + ((chunk ^ (row & 7)) << 4));
            if (row0 + row < a.Sq) *reinterpret_cast<uint4*>(op + (long long)(row0 + row) * a.o_ss + chunk * 8) = v;
        }
    }
    stamp(3);
#endif
}

template <int PD, int NQ>
int launch(const AttnArgs& a_in, hipStream_t stream) {
    AttnArgs a = a_in;
    constexpr int LDS = (PD + 3) * kStage;
    constexpr int kRowsPerWg = 128 * NQ;
    static_assert(4 * 32 * NQ * 128 <= LDS, "the epilogue stages the workgroup's output block in the ring");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd_w128<PD, NQ>),
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
    hipLaunchKernelGGL((k_attn_fwd_w128<PD, NQ>), dim3((unsigned)n_blocks), dim3(256), LDS, stream, a);
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
bool attn_w128_ok(const AttnArgs& a) {
    if (a.D != 64 || a.Skv < kTile || (a.Skv & (kTile - 1)) != 0) return false;
    const long long kb = (((long long)a.Skv + kTile) * a.k_ss + 64) * 2, vb = (63LL * a.vt_ds + a.Skv + kTile) * 2;
    if (a.Sq % 512 != 0) return false;                       // whole 512-row workgroups only (the 64-row kernel takes the rest)
    return kb < 0x40000000LL && vb < 0x40000000LL && a.k_ss >= 64 && a.vt_ds >= a.Skv && (a.o_ss & 7) == 0 && (a.o_hs & 7) == 0 &&
           (a.o_bs & 7) == 0 && (((uintptr_t)a.out) & 15) == 0;
}

int launch_attn_w128(const AttnArgs& a, hipStream_t stream) { return launch<2, 4>(a, stream); }

}  // namespace dm_attn
