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
#include <cstdlib>
#include <cstring>
#include <type_traits>

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
    const __bf16* rowbias;   // [B, Cout] or null: per-image channel bias (the ResnetBlock2D time embedding)
    const __bf16* res;       // [B, Hout, Wout, Cout] or null: residual added in the epilogue
    __bf16* y;           // [B, Hout, Wout, Cout]
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int stride, pad_y, pad_x;
    long long M;         // B*Hout*Wout
    // LDS-DMA kernels: an M tile is a 2-D patch of TW x (BMT/TW) output pixels of the "tall image" [B*Hout, Wout]
    int tw_log2;         // log2(TW)
    int tiles_x;         // ceil(Wout / TW)
};

constexpr int BM = 128;

// (m-tile, n-tile) of this workgroup.  Consecutive tile ids share the activation (A) tile and walk the
// Cout tiles; since the dispatcher places block b on XCD b % 8 (private L2 per XCD), each XCD gets a
// CONTIGUOUS range of tile ids so that the re-used A tile and the weight panel stay in ITS L2.
__device__ __forceinline__ bool tile_of_block(long long n_mt, int n_nt, long long& mt, int& nt) {
    const long long total = n_mt * n_nt;
    const long long lin = (long long)blockIdx.x;
    const long long per_xcd = (total + 7) / 8;
    const long long id = (lin & 7) * per_xcd + (lin >> 3);
    if ((lin >> 3) >= per_xcd || id >= total) return false;
    mt = id / n_nt;
    nt = (int)(id - mt * n_nt);
    return true;
}

template <int BN, int BK>
__global__ __launch_bounds__(256) void k_conv3x3(ConvArgs a, long long n_mt, int n_nt) {
    constexpr int ROWB = BK * 2 + 16;                  // LDS bytes per tile row (BK bf16 + 16 B pad)
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int NT = BN / 64;                        // 32-wide n tiles per wave (wave tile = 64 x BN/2)
    constexpr int CPR = BK / 8;                        // 16 B chunks per tile row
    constexpr int A_CHUNKS = BM * CPR / 256;           // per thread
    constexpr int B_CHUNKS = BN * CPR / 256;
    constexpr int RPI = 256 / CPR;                     // rows covered per chunk-iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];

    long long mt; int nt;
    if (!tile_of_block(n_mt, n_nt, mt, nt)) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;           // 2 x 2 waves
    const long long m0 = mt * BM;
    const int n0 = nt * BN;

    // ---- per-thread gather coordinates of its A rows (row = tid/CPR + RPI*i), chunk = tid%CPR
    const int a_chunk = tid % CPR;
    int a_b[A_CHUNKS], a_y[A_CHUNKS], a_x[A_CHUNKS];
    bool a_ok[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        long long m = m0 + tid / CPR + RPI * i;
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

    uint4 areg[A_CHUNKS], breg[B_CHUNKS];
    auto load_step = [&](int s) {
        int tap = s / kt_per_tap;
        int c0 = (s - tap * kt_per_tap) * BK + a_chunk * 8;
        int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            int yy = a_y[i] + dy, xx = a_x[i] + dx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_ok[i] && (unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win)
                v = *reinterpret_cast<const uint4*>(a.x + (((long long)a_b[i] * a.Hin + yy) * a.Win + xx) * a.Cin + c0);
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            int row = tid / CPR + RPI * i;
            breg[i] = *reinterpret_cast<const uint4*>(a.w + (long long)(n0 + row) * Kw + (long long)s * BK + a_chunk * 8);
        }
    };
    auto write_step = [&](int buf) {
        char* ab = smem + buf * (A_BYTES + B_BYTES);
        char* bb = ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            *reinterpret_cast<uint4*>(ab + (tid / CPR + RPI * i) * ROWB + a_chunk * 16) = areg[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            *reinterpret_cast<uint4*>(bb + (tid / CPR + RPI * i) * ROWB + a_chunk * 16) = breg[i];
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
        for (int kk = 0; kk < BK / 16; ++kk) {
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

// ------------------------------------------------------------------------------------------------
// Variant 2: LDS-DMA pipeline.  Same tiling (128 x BN x 64, 4 waves), but the tiles travel
// HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging) through a 3-stage ring, so two
// K-steps of loads are in flight behind the MFMAs instead of one: the kernel above is latency-bound
// (~16 % of MFMA peak) because a single staged tile cannot cover an L2/HBM round trip.
//  * LDS rows are unpadded 128 B (the DMA writes wave-uniform base + lane*16); bank conflicts are
//    avoided by XOR-swizzling the 16 B chunk index with (row>>1)&7 -- applied to the per-lane SOURCE
//    address on the way in and to the ds_read address on the way out (same involution both sides).
//  * out-of-image taps read from a 16-byte zero page instead of being predicated.
//  * counted s_waitcnt vmcnt(L) + raw s_barrier: one barrier per K-step, loads span the barrier.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};

template <int BMT, int BN, int NW, int WMW, int NSTAGE>
__global__ __launch_bounds__(NW * 64) void k_conv3x3_dma(ConvArgs a, long long n_mt, int n_nt) {
    // BMT x BN x 64 workgroup tile, NW waves laid out WMW (along M) x NW/WMW (along N), NSTAGE-deep LDS ring.
    //   <128, 64|128, 4, 2, 3>  wave tile 64 x 32|64, 2-5 workgroups per CU (small problems)
    //   <256, 128, 8, 4, 3>     wave tile 64 x 64, one workgroup per CU
    //   <256, 256, 8, 2, 2>     wave tile 128 x 64: 24 ds_read_b128 per 32 MFMA instead of 16 per 16, and half the
    //   <256, 320, 8, 4, 2>     wave tile 64 x 160    DMA bytes per FLOP (Cout = 320 without a ragged N tile)
    //   <512, 128, 8, 4, 2>     wave tile 128 x 64 for Cout = 128 (the 512^2 VAE layers)
    constexpr int BK = 64;
    constexpr int ROWB = 128;                          // bytes per tile row (64 bf16), unpadded
    constexpr int WNW = NW / WMW;
    constexpr int TM = BMT / WMW, TN = BN / WNW;       // wave tile
    constexpr int MT = TM / 32, NT = TN / 32;          // 32x32 accumulator fragments per wave
    constexpr int A_BYTES = BMT * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BMT / 8 / NW;              // wave-instructions (8 rows each) per wave
    constexpr int B_INSTR = BN / 8 / NW;
    constexpr int L = A_INSTR + B_INSTR;               // DMA instructions per wave per tile
    static_assert(TM % 32 == 0 && TN % 32 == 0 && BMT % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSTAGE * STAGE

    long long mt; int nt;
    {
        const long long total = n_mt * n_nt;
        const long long lin = (long long)blockIdx.x;
        const long long per_xcd = (total + 7) / 8;
        const long long id = (lin & 7) * per_xcd + (lin >> 3);
        if ((lin >> 3) >= per_xcd || id >= total) return;
        mt = id / n_nt;
        nt = (int)(id - mt * n_nt);
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WNW, wn = wave % WNW;
    // M tile = patch of TW x TH output pixels: rows Y0.. of the tall image [B*Hout, Wout], columns X0..
    // (tile row r -> pixel (Y0 + r / TW, X0 + r % TW)).  A 1-D run of BMT pixels re-reads 3 full image rows per
    // tile; the patch re-reads a one-pixel halo: (TH+2)(TW+2)/(TH*TW) = 1.2-1.3x.
    const int TWm = (1 << a.tw_log2) - 1;
    const int tile_y = (int)(mt / a.tiles_x);          // B*Hout < 2^31 (checked by the launcher): 32-bit row math
    const int Y0 = tile_y * (BMT >> a.tw_log2);
    const int X0 = (int)(mt - (long long)tile_y * a.tiles_x) << a.tw_log2;
    const int rows_total = a.B * a.Hout;
    const int n0 = nt * BN;
    const int lrow = lane >> 3, lslot = lane & 7;      // this lane's row / 16 B slot inside one DMA instruction
    const unsigned long long zero = (unsigned long long)g_zero_page;

    // A rows of this lane: wave*(BMT/NW) + 8*i + lrow.  Everything that depends on the row is computed ONCE:
    // a pointer to tap (0,0) / channel 0 of the row's receptive field (pre-swizzled chunk; it may lie outside
    // the image and is only dereferenced for taps whose bit is set in a_mask) and a 9-bit tap-validity mask.
    // The K-loop then needs one 64-bit add and one select per DMA instruction (the first version recomputed
    // ((b*Hin + y)*Win + x)*Cin per tap: ~110 VALU / 24 quarter-rate multiplies per K-step per wave, more
    // issue time than the 16 MFMAs they feed).
    const __bf16* a_ptr[A_INSTR];
    unsigned a_mask[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        int row = wave * (BMT / NW) + 8 * i + lrow;
        int Y = Y0 + (row >> a.tw_log2);
        int xo = X0 + (row & TWm);
        bool ok = Y < rows_total && xo < a.Wout;
        int Yc = ok ? Y : 0;
        int b = Yc / a.Hout;
        int yo = Yc - b * a.Hout;
        int y0 = yo * a.stride - a.pad_y;
        int x0 = (ok ? xo : 0) * a.stride - a.pad_x;
        a_ptr[i] = a.x + (((long long)b * a.Hin + y0) * a.Win + x0) * a.Cin + (lslot ^ ((row >> 1) & 7)) * 8;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int yy = y0 + t / 3, xx = x0 + t % 3;
            if (ok && (unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win) mk |= 1u << t;
        }
        a_mask[i] = mk;
    }
    // weight rows past Cout (ragged last tile, e.g. 320 = 2.5 x 128) re-read row Cout-1: finite values that
    // only reach accumulator columns the epilogue never stores
    const __bf16* b_ptr[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        int row = wave * (BN / NW) + 8 * i + lrow;
        int rc = min(n0 + row, a.Cout - 1);
        b_ptr[i] = a.w + (long long)rc * 9LL * a.Cin + (lslot ^ ((row >> 1) & 7)) * 8;
    }
    const int kt_per_tap = a.Cin / BK;
    const int n_steps = 9 * kt_per_tap;

    // issue cursor (wave-uniform => SGPRs).  K order = channel block OUTER, tap INNER: the 9 taps of one 64-channel
    // block re-read the same 128-byte line of every halo pixel back to back, so the per-CU L2 working set is
    // (TH+2)(TW+2) lines (41 KB for 16x16) whatever Cin is.  Tap-outer order swept all Cin between re-reads
    // (83 KB-830 KB per CU, x32 CUs per 4 MB L2): rocprofv3 FETCH_SIZE showed 2.1x (Cin=128) to 6.3x (Cin=320)
    // the compulsory bytes and a 66-76 % L2 hit rate (profiles/r01_pmc_conv_v0.json).
    int i_tap = 0, i_kc = 0;
    long long toff = 0, woff = 0;                      // element offsets of the step being issued
    unsigned bit = 1u;
    auto cursor_set = [&]() {
        const int dy = (i_tap * 11) >> 5, dx = i_tap - 3 * dy;                      // tap / 3, tap % 3 for tap < 9
        toff = (long long)(dy * a.Win + dx) * a.Cin + i_kc * BK;                     // from a_ptr
        woff = (long long)i_tap * a.Cin + i_kc * BK;                                 // weights are [Cout][tap][Cin]
        bit = 1u << i_tap;
    };
    auto cursor_next = [&]() { if (++i_tap == 9) { i_tap = 0; ++i_kc; } };
    // DMA piece p of the step at the cursor (p < A_INSTR: 8 activation rows, else 8 weight rows) into `stage`
    auto piece = [&](int p, int stage) {
        char* ab = smem + stage * STAGE;
        if (p < A_INSTR) {
            // select, never a branch: the DMA must execute with ALL lanes active (an inactive lane would
            // leave its LDS slot stale); out-of-image taps read the 16-byte zero page
            const void* src = (a_mask[p] & bit) ? (const void*)(a_ptr[p] + toff) : (const void*)zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ab + (wave * (BMT / NW) + 8 * p) * ROWB),
                                             16, 0, 0);
        } else {
            const int q = p - A_INSTR;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[q] + woff),
                                             (__attribute__((address_space(3))) void*)(ab + A_BYTES + (wave * (BN / NW) + 8 * q) * ROWB),
                                             16, 0, 0);
        }
    };
    auto issue = [&](int stage) {
        cursor_set();
#pragma unroll
        for (int p = 0; p < L; ++p) piece(p, stage);
        cursor_next();
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ring protocol: at the top of step s every wave waits for ITS OWN step-s DMAs (counted vmcnt leaves the
    // NSTAGE-2 younger steps in flight), then the barrier makes all waves' step-s data visible and proves
    // everybody has finished reading stage (s-1) % NSTAGE, which is the stage the next issue overwrites.
    issue(0);
    if (NSTAGE == 3 && n_steps > 1) issue(1);
    // One K-step: 4 chunks of 16 K.  Chunk kk (a) reads the fragments of chunk kk+1 into the other register set,
    // (b) issues its quarter of the NEXT stage's DMA pieces, (c) runs its MT*NT MFMAs on fragments that were read
    // one chunk earlier -- so LDS latency and the DMA issue time (60-180 cycles per 1 KB piece) sit under MFMA
    // execution instead of in front of it (the first version issued all pieces, then read, then multiplied:
    // SQ_WAIT_ANY 39 %, MFMA busy 29 %, profiles/r01_pmc_conv_v0.json).
    auto read_frags = [&](int stage, int kk, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) {
        const char* ab = smem + stage * STAGE;
        const char* bb = ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int r = TM * wm + 32 * i + l31;
            af[i] = *reinterpret_cast<const bf16x8*>(ab + r * ROWB + (((2 * kk + hi) ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int r = TN * wn + 32 * j + l31;
            bf[j] = *reinterpret_cast<const bf16x8*>(bb + r * ROWB + (((2 * kk + hi) ^ ((r >> 1) & 7)) << 4));
        }
    };
    auto mma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                // weights as the A operand, pixels as B: the accumulator fragment is D^T[n][m] -- lane = pixel, registers =
                // channels, so the epilogue can store 16 contiguous bytes (8 channels of one pixel) per lane
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    };
    auto kstep = [&](int stage, int st_next, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        // 3-deep ring: the pieces have a whole extra step to land, spread them over all 4 chunks.  2-deep ring:
        // they are needed at the next barrier, so the last chunk issues nothing (>= a quarter step of lead time)
        constexpr int NCH = (NSTAGE == 2) ? 3 : 4;
        auto pieces = [&](int kk) {
            if (ISSUE && kk < NCH) {
#pragma unroll
                for (int p = kk; p < L; p += NCH) piece(p, st_next);
            }
        };
        bf16x8 a0[MT], b0[NT], a1[MT], b1[NT];         // two fragment sets, used alternately (named, not indexed:
        if (ISSUE) cursor_set();                       // a parity-indexed array is not promoted to registers)
        read_frags(stage, 0, a0, b0);
        read_frags(stage, 1, a1, b1); pieces(0); mma(a0, b0); __builtin_amdgcn_sched_barrier(0);
        read_frags(stage, 2, a0, b0); pieces(1); mma(a1, b1); __builtin_amdgcn_sched_barrier(0);
        read_frags(stage, 3, a1, b1); pieces(2); mma(a0, b0); __builtin_amdgcn_sched_barrier(0);
        pieces(3); mma(a1, b1);
        if (ISSUE) cursor_next();
    };
    // main loop (every step issues the DMAs of step s + NSTAGE - 1) and drain (nothing left to issue) are SEPARATE
    // loops: with both bodies under one loop the register allocator gave each its own accumulator set
    int stage = 0, s = 0;
    const int n_main = n_steps - (NSTAGE - 1);
    for (; s < n_main; ++s) {
        if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int st2 = stage + NSTAGE - 1; if (st2 >= NSTAGE) st2 -= NSTAGE;
        kstep(stage, st2, std::true_type{});
        stage = stage + 1; if (stage >= NSTAGE) stage = 0;
    }
    for (; s < n_steps; ++s) {
        if (NSTAGE == 3 && s + 1 < n_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        kstep(stage, 0, std::false_type{});
        stage = stage + 1; if (stage >= NSTAGE) stage = 0;
    }

    // ---- epilogue: y = acc + bias[n] (+ rowbias[image(m), n]) (+ res[m, n]), one rounding to bf16.
    // D^T layout: lane = pixel l31 of the fragment, register r = channel (r&3) + 8*(r>>2) + 4*hi.  Group g = r>>2 is four
    // consecutive channels (8 bytes as bf16); lanes l and l+32 hold the two halves of one 8-channel run.  One
    // v_permlane32_swap per dword on the group pair (g, g+1) turns that into 16 contiguous bytes per lane: lanes 0-31 get
    // channels 8g..8g+7, lanes 32-63 channels 8g+8..8g+15 of their pixel -> TWO 16-byte stores per 32x32 fragment.  The
    // first version stored every element on its own (2 bytes per lane, 128 store instructions per wave and tile): at
    // Cin = 128 the store issue took longer than the K loop (tools/conv_fit.sh: 0.40 ms of 0.96 ms did not scale with K).
    const int img0 = Y0 / a.Hout;                      // image of the patch's first row (wave-uniform)
    const int rem0 = Y0 - img0 * a.Hout;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int d = TM * wm + 32 * i + l31;          // this lane's pixel of the tile
        const int ty = d >> a.tw_log2;
        const int Y = Y0 + ty;
        const int xo = X0 + (d & TWm);
        const bool pix_ok = Y < rows_total && xo < a.Wout;
        const long long m = (long long)Y * a.Wout + xo;           // = (b*Hout + yo)*Wout + xo
        const __bf16* rb = nullptr;
        if (a.rowbias) {
            int img = img0, t = rem0 + ty;
            while (t >= a.Hout) { t -= a.Hout; ++img; }         // a patch spans at most TH / Hout + 1 images
            rb = a.rowbias + (long long)min(img, a.B - 1) * a.Cout;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nbase = n0 + TN * wn + 32 * j;   // first channel of this fragment (wave-uniform)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {           // group pairs (0,1) and (2,3)
                // fp32 adds in the accumulator layout (register 4g+e = channel nbase + 8g + 4hi + e), one rounding to bf16
                unsigned w[2][2];                      // [group of the pair][dword] = 4 bf16 per group
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = 2 * gp + q;
                    const int n = nbase + 8 * g + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                    if (n < a.Cout) {
                        if (a.bias) {
                            const uint2 bb = *reinterpret_cast<const uint2*>(a.bias + n);
                            v[0] += __builtin_bit_cast(float, bb.x << 16); v[1] += __builtin_bit_cast(float, bb.x & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, bb.y << 16); v[3] += __builtin_bit_cast(float, bb.y & 0xffff0000u);
                        }
                        if (rb) {
                            const uint2 bb = *reinterpret_cast<const uint2*>(rb + n);
                            v[0] += __builtin_bit_cast(float, bb.x << 16); v[1] += __builtin_bit_cast(float, bb.x & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, bb.y << 16); v[3] += __builtin_bit_cast(float, bb.y & 0xffff0000u);
                        }
                        if (a.res && pix_ok) {
                            const uint2 rr = *reinterpret_cast<const uint2*>(a.res + m * a.Cout + n);
                            v[0] += __builtin_bit_cast(float, rr.x << 16); v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, rr.y << 16); v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
                        }
                    }
                    f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                    bf16x2 plo = __builtin_convertvector(lo, bf16x2), phi = __builtin_convertvector(hi2, bf16x2);
                    w[q][0] = __builtin_bit_cast(unsigned, plo);
                    w[q][1] = __builtin_bit_cast(unsigned, phi);
                }
                // v_permlane32_swap(vdst = group g, src = group g+1): lanes 32-63 of vdst <-> lanes 0-31 of src.  Afterwards
                //   result[0]: lanes 0-31 own group g (ch 8g..8g+3)        | lanes 32-63 the lower lanes' group g+1 (ch 8g+8..+11)
                //   result[1]: lanes 0-31 the upper lanes' group g (+4..+7) | lanes 32-63 own group g+1 (ch 8g+12..+15)
                const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                const int c0 = nbase + 16 * gp + 8 * hi;   // this lane now owns channels c0 .. c0+7 of its pixel
                if (pix_ok && c0 < a.Cout)
                    *reinterpret_cast<uint4*>(a.y + m * a.Cout + c0) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
}

template <int BMT, int BN, int NW, int WMW, int NSTAGE>
int launch_conv_dma(const ConvArgs& a_in, hipStream_t stream) {
    constexpr int LDS = NSTAGE * (BMT + BN) * 128;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_dma<BMT, BN, NW, WMW, NSTAGE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ConvArgs a = a_in;
    // patch width: 16 output pixels (or the next power of two >= Wout for narrower maps), height BMT / TW
    int tw_log2 = 4;
    while (tw_log2 > 0 && (1 << (tw_log2 - 1)) >= a.Wout) --tw_log2;
    a.tw_log2 = tw_log2;
    a.tiles_x = (a.Wout + (1 << tw_log2) - 1) >> tw_log2;
    const int TH = BMT >> tw_log2;
    if ((long long)a.B * a.Hout + 512 > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    long long n_mt = (((long long)a.B * a.Hout + TH - 1) / TH) * a.tiles_x;
    int n_nt = (a.Cout + BN - 1) / BN;
    long long blocks = ((n_mt * n_nt + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL((k_conv3x3_dma<BMT, BN, NW, WMW, NSTAGE>), dim3((unsigned)blocks), dim3(NW * 64), LDS, stream, a,
                       n_mt, n_nt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}

template <int BN, int BK>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * (BM + BN) * (BK * 2 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<BN, BK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    long long n_mt = (a.M + BM - 1) / BM;
    int n_nt = a.Cout / BN;
    long long total = n_mt * n_nt;
    long long blocks = ((total + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    DM_ENTER();
    hipLaunchKernelGGL((k_conv3x3<BN, BK>), dim3((unsigned)blocks), dim3(256), LDS, stream, a, n_mt, n_nt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}

}  // namespace

extern "C" {

// x [B,Hin,Win,Cin] NHWC bf16; w [Cout,3,3,Cin] (= [Cout, 9*Cin], tap-major) bf16; bias [Cout] bf16 or NULL;
// y [B,Hout,Wout,Cout] NHWC bf16 with Hout = (Hin + pad_y + pad_y_end - 3)/stride + 1 chosen by the caller
// (pad_y / pad_x are the leading pads; trailing pads are implied by Hout/Wout and zero-filled).
// rowbias [B,Cout] / residual [B,Hout,Wout,Cout] (bf16, either may be NULL) are added in the epilogue
// (LDS-DMA kernels only: Cin % 64 == 0).
int dm_conv3x3_nhwc_bf16_fused(const void* x, const void* w, const void* bias, const void* rowbias, const void* residual,
                               void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride,
                               int pad_y, int pad_x, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return DM_ERR_ARG;
    if (Cin % 32 != 0 || Cout % 64 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return DM_ERR_ARG;
    if (((uintptr_t)bias | (uintptr_t)rowbias | (uintptr_t)residual) & 7) return DM_ERR_ARG;
    ConvArgs a;
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const __bf16*)bias; a.y = (__bf16*)y;
    a.rowbias = (const __bf16*)rowbias; a.res = (const __bf16*)residual;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.stride = stride; a.pad_y = pad_y; a.pad_x = pad_x;
    a.M = (long long)B * Hout * Wout;
    // DREAMMAT_CONV_KERNEL=staged selects the register-staged variant everywhere (A/B measurements)
    static const bool use_dma = !(getenv("DREAMMAT_CONV_KERNEL") && !strcmp(getenv("DREAMMAT_CONV_KERNEL"), "staged"));
    if (use_dma && Cin % 64 == 0) {
        // Tile choice from profiles/r01_kernel_bench_conv_tiles.json (TF/s, MI355X):
        //   Cout=320 @64x64 B24: 256x128 632 | 256x320 839      960->320: 750 | 952      640 @32x32: 897 | 950
        //   Cout=256 @256x256 B8: 256x128 888 | 256x256 1023    512 @64x64: 951 | 1107
        //   1280 @16x16 B24 (M = 6144): 256x128 925 | 256x256 629 | 256x320 531   (too few workgroups for 256 CUs)
        // => the widest tile that divides Cout, as long as it still yields enough workgroups to fill the chip.
        // DREAMMAT_CONV_TILE=128|256|512|320|640 forces a variant (tests / A-B measurements).
        const char* tile_env = getenv("DREAMMAT_CONV_TILE");   // read per call: tests toggle it
        int tile = tile_env ? atoi(tile_env) : 0;
        auto n_wg = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
        if (!tile) {
            if (!(Cout >= 128 && a.M >= 2048)) tile = 128;
            else if (Cout % 256 == 0 && n_wg(256, 256) >= 200) tile = 512;
            else if ((Cout == 128 || Cout == 640) && n_wg(512, 128) >= 200) tile = 640;
            else if (Cout % 320 == 0 && Cout % 256 != 0 && n_wg(256, 320) >= 160) tile = 320;
            else tile = 256;
        }
        switch (tile) {
        case 640: {                                                          // wave tile 128 x 64, all 160 KB of LDS
            int rc = launch_conv_dma<512, 128, 8, 4, 2>(a, stream);
            if (rc <= 0 || tile_env) return rc;
            return launch_conv_dma<256, 128, 8, 4, 3>(a, stream);            // runtime refused the full-LDS variant
        }
        case 320: return launch_conv_dma<256, 320, 8, 4, 2>(a, stream);      // wave tile 64 x 160
        case 512: return launch_conv_dma<256, 256, 8, 2, 2>(a, stream);      // wave tile 128 x 64
        case 256: return launch_conv_dma<256, 128, 8, 4, 3>(a, stream);      // wave tile 64 x 64
        default:
            return (Cout % 128 == 0) ? launch_conv_dma<128, 128, 4, 2, 3>(a, stream)
                                     : launch_conv_dma<128, 64, 4, 2, 3>(a, stream);
        }
    }
    if (rowbias || residual) return DM_ERR_UNSUPPORTED;
    if (Cout % 128 == 0) return (Cin % 64 == 0) ? launch_conv<128, 64>(a, stream) : launch_conv<128, 32>(a, stream);
    return (Cin % 64 == 0) ? launch_conv<64, 64>(a, stream) : launch_conv<64, 32>(a, stream);
}

int dm_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                         int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, hipStream_t stream) {
    return dm_conv3x3_nhwc_bf16_fused(x, w, bias, nullptr, nullptr, y, B, Hin, Win, Cin, Hout, Wout, Cout, stride, pad_y,
                                      pad_x, stream);
}

}  // extern "C"
