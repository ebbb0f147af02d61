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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "dm_common.h"
#include "dm_elem.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// which of `n_ph` phases runs behind MFMA `slot` of `total` when phase p sits behind MFMA ceil(p * total / n_ph) (-1: none)
constexpr int gn_phase_of_slot(int slot, int total, int n_ph) {
    for (int p = 0; p < n_ph; ++p)
        if ((p * total + n_ph - 1) / n_ph == slot) return p;
    return -1;
}

// compile-time loop (indices as types: bodies whose register indices must all be constants)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_c(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_c<I + 1, N>(f);
    }
}

struct ConvArgs {
    const elem_t* x;     // [B, Hin, Win, Cin]
    const elem_t* w;     // [Cout, 9, Cin]
    const elem_t* bias;  // [Cout] or null
    const elem_t* rowbias;   // [B, Cout] or null: per-image channel bias (the ResnetBlock2D time embedding)
    const elem_t* res;       // [B, Hout, Wout, Cout] or null: residual added in the epilogue
    elem_t* y;           // [B, Hout, Wout, Cout]
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    int stride, pad_y, pad_x;
    long long M;         // B*Hout*Wout
    // LDS-DMA kernels: an M tile is a 2-D patch of TW x (BMT/TW) output pixels of the "tall image" [B*Hout, Wout]
    int tw_log2;         // log2(TW)
    int tiles_x;         // ceil(Wout / TW)
    int y_off;           // first row of the tall image this launch covers (0; > 0: the second launch of launch_320_balanced)
    long long mt_cap;    // at most this many M tiles from y_off on (0 = to the end)
    // 2 x 2 instantiation with EPI = 2 only: store the four Cs-channel blocks of an output pixel as the four sub-pixels of a [B, 2H', 2W', Cs] tensor
    // (0 = plain [B, Hout, Wout, Cout] rows).  1: block (py, px) of pixel (u, v) -> (2u + py, 2v + px), H' = Hout, W' = Wout (the stride-2
    // data gradient).  2: -> (2u - py, 2v - px) where that is inside H' = Hout - 1, W' = Wout - 1 (upsample + conv on its (h+1) x (w+1) grid)
    int shuf_mode, shuf_cs;
    // halo-patch kernel with the GroupNorm apply pass folded in (round 6): x is the UN-normalised tensor, gn_coef = the fp32
    // coefficient rows dm_groupnorm_nhwc_stats leaves ([B][7][Cin]: row 0 = rstd * gamma, row 1 = beta - mean * rstd * gamma), the
    // patch in LDS becomes act(x * A + S) rounded to 16 bits (what the apply kernel would have written) before any tap reads it
    const float* gn_coef;
    // batched 1-tap launch (dm_gemm_*_batched, round 6): w = [wb_count][Cout][Cin], the rows of x / y in blocks of 16 * wb_y per
    // item (a multiple of every M tile: no tile straddles two items); 0 = one weight matrix for all rows
    int wb_y, wb_count;
    int gn_act;
    unsigned long long* timeline;   // DREAMMAT_CONV_TIMELINE=1 (development): s_memtime stamps per tile, else null
    int timeline_steps;             // DREAMMAT_CONV_TIMELINE=2: stamp every K-step instead; 4: per-wave sums of body / waits / barrier
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
            elem8 af[2], bf[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const elem8*>(ab + (64 * wm + 32 * i + l31) * ROWB + 32 * kk + 16 * hi);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const elem8*>(bb + ((BN / 2) * wn + 32 * j + l31) * ROWB + 32 * kk + 16 * hi);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = DM_MFMA_32x32x16(af[i], bf[j], acc[i][j]);
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
                if (m < a.M) a.y[m * a.Cout + n] = (elem_t)(acc[i][j][r] + bv);
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
//  * out-of-image taps are buffer offsets beyond num_records (the descriptor's range check returns zeros).
//  * counted s_waitcnt vmcnt(L) + raw s_barrier: one barrier per K-step, loads span the barrier.
// TAPS = 4: a 2 x 2 window (taps (0,0) (0,1) (1,0) (1,1) from the leading pad on): the sub-pixel form of a stride-2 convolution's
// DATA GRADIENT (dm_conv2x2_nhwc_bf16 below).
// TAPS = 9: 3x3 convolution.  TAPS = 1: the same machine as a plain GEMM y[M, N] = x[M, K] w[N, K]^T (the launcher presents
// x as a [1, M/16, 16, K] image, no padding) -- the Linear / 1x1 layers of the UNet, with the same fused epilogues.
// EPI = 2 (TAPS = 4 only): the plain epilogue with the sub-pixel store of ConvArgs::shuf_mode.
// EPI = 1 (TAPS = 1 only): GEGLU epilogue.  The weight rows arrive interleaved in blocks of 32 (32 value rows, then their
// 32 gate rows), so fragment pair (2jj, 2jj+1) of a wave holds value and gate of the same 32 output channels and the
// epilogue writes value * gelu(gate) into y[M, N/2]: the 2x wide intermediate never reaches HBM.
template <int BMT, int BN, int NW, int WMW, int NSTAGE, int TAPS = 9, int EPI = 0>
__global__ __launch_bounds__(NW * 64) void k_conv3x3_dma(ConvArgs a, long long n_mt, int n_nt, int stagger_units, int ksplit, float* __restrict__ ws) {
    // BMT x BN x 64 workgroup tile, NW waves laid out WMW (along M) x NW/WMW (along N), NSTAGE-deep LDS ring.
    //   <128, 64|128, 4, 2, 3>  wave tile 64 x 32|64, 2-5 workgroups per CU (small problems)
    //   <256, 128, 8, 4, 3>     wave tile 64 x 64, one workgroup per CU
    //   <256, 256, 8, 2, 2>     wave tile 128 x 64: 24 ds_read_b128 per 32 MFMA instead of 16 per 16, and half the
    //   <256, 320, 8, 4, 2>     wave tile 64 x 160    DMA bytes per FLOP (Cout = 320 without a ragged N tile)
    //   <512, 128, 8, 4, 2>     wave tile 128 x 64 for Cout = 128 (the 512^2 VAE layers)
    [[maybe_unused]] constexpr int BK = 64;         // (the constants below are used by the device pass only)
    constexpr int ROWB = 128;                          // bytes per tile row (64 bf16), unpadded
    constexpr int WNW = NW / WMW;
    constexpr int TM = BMT / WMW, TN = BN / WNW;       // wave tile
    [[maybe_unused]] constexpr int MT = TM / 32, NT = TN / 32;     // 32x32 accumulator fragments per wave
    [[maybe_unused]] constexpr int A_BYTES = BMT * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
#if defined(DM_ASYM_DMA)
    constexpr int NWL = NW == 8 ? 4 : NW;              // EXPERIMENT: loader waves (the first wave of each SIMD pair requests for both)
#else
    constexpr int NWL = NW;
#endif
    constexpr int A_INSTR = BMT / 8 / NWL;             // wave-instructions (8 rows each) per loader wave
    constexpr int B_INSTR = BN / 8 / NWL;
    [[maybe_unused]] constexpr int L = A_INSTR + B_INSTR;    // DMA instructions per wave per tile
    static_assert(TM % 32 == 0 && TN % 32 == 0 && BMT % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSTAGE * STAGE
#if defined(__HIP_DEVICE_COMPILE__)   // __amdgpu_buffer_rsrc_t does not exist in the host pass (the stub needs no body)

    // PERSISTENT workgroups: gridDim.x (a multiple of 8) workgroups walk the tiles of their XCD's contiguous id range
    // (block b runs on XCD b % 8, private L2 each) with stride gridDim.x / 8, and the LDS-DMA ring keeps running ACROSS
    // tile boundaries: while tile t's last K-steps multiply and its epilogue stores, the first K-steps of tile t+1 are
    // already landing.  One workgroup per tile paid a cold prologue (address set-up, first DMA round trip), a store tail
    // and a relaunch per tile -- 0.25 ms of the 0.80 ms of the 8 x 128->128 @512^2 layer did not scale with K
    // (tools/conv_fit.sh), with one workgroup per CU nothing else could cover it.
    // SPLIT-K (ksplit > 1, small-M problems: 1280->1280 @8x8 has 20 output tiles for 256 CUs): a work item is (output tile,
    // K range); its fp32 partial sums go to ws[ks][M][Cout] and k_splitk_reduce adds them up with the bias / residual terms
    constexpr bool SPLIT_OK = EPI == 0 && BMT <= 256 && BN <= 128;
    const int total = (int)(n_mt * n_nt) * ksplit;     // < 2^31 (launcher)
    const int per_xcd = (total + 7) / 8;
    const int wpx = (int)(gridDim.x >> 3);             // workgroups per XCD
    const int xbeg = (int)(blockIdx.x & 7) * per_xcd;
    const int xend = min(xbeg + per_xcd, total);
    const int first = xbeg + (int)(blockIdx.x >> 3);
    if (first >= xend) return;
    // (the wave index as an SGPR value: every LDS-DMA piece needs its wave's LDS base in M0, and with `wave` derived from the
    // thread id in a VGPR each piece paid a v_add + v_readfirstlane + s_mov m0 chain with its wait states; now one s_add into m0
    // -- round 6: +2-5 % on the 320-wide tiles, +1-2 % on the Linear shapes, nothing lost elsewhere)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WNW, wn = wave % WNW;
    // M tile = patch of TW x TH output pixels: rows Y0.. of the tall image [B*Hout, Wout], columns X0..
    // (tile row r -> pixel (Y0 + r / TW, X0 + r % TW)).  A 1-D run of BMT pixels re-reads 3 full image rows per
    // tile; the patch re-reads a one-pixel halo: (TH+2)(TW+2)/(TH*TW) = 1.2-1.3x.
    const int TWm = (1 << a.tw_log2) - 1;
    const int rows_total = a.B * a.Hout;
    // 32-bit tile arithmetic (the launcher admits < 2^31 tiles): this runs inside the K loop when the issue cursor
    // crosses into the next tile, where every cycle is a cycle without MFMAs -- the first persistent version spent 13 k
    // cycles per tile here on two 64-bit divisions, 64-bit address products and the spills around them
    // (DREAMMAT_CONV_TIMELINE=2).
    auto tile_coords = [&](int id, int& Y0, int& X0, int& n0, int& ks) __attribute__((always_inline)) {
        unsigned uid = (unsigned)id;
        ks = 0;
        if (SPLIT_OK && ksplit > 1) { ks = (int)(uid % (unsigned)ksplit); uid /= (unsigned)ksplit; }
        const unsigned mt = uid / (unsigned)n_nt;
        const unsigned nt = uid - mt * (unsigned)n_nt;
        const unsigned tile_y = mt / (unsigned)a.tiles_x;
        Y0 = a.y_off + (int)(tile_y * (unsigned)(BMT >> a.tw_log2));
#if defined(DM_ABL_L2HOT)
        if (TAPS == 9) Y0 = Y0 & 63;                    // ABLATION (wrong results): every tile reads (and writes) the first 64 rows: operands L2-resident
#endif
        X0 = (int)((mt - tile_y * (unsigned)a.tiles_x) << a.tw_log2);
        n0 = (int)nt * BN;
    };
    const int kt_per_tap = a.Cin / BK;
    const int n_steps = TAPS * kt_per_tap;
    auto k_lo = [&](int ks) __attribute__((always_inline)) {     // first K-step of split ks (ks = ksplit: one past the end)
        return (SPLIT_OK && ksplit > 1) ? (int)((long long)ks * n_steps / ksplit) : (ks ? n_steps : 0);
    };

    // issue cursor (wave-uniform => SGPRs).  K order = channel block OUTER, tap INNER: the 9 taps of one 64-channel
    // block re-read the same 128-byte line of every halo pixel back to back, so the per-CU L2 working set is
    // (TH+2)(TW+2) lines (41 KB for 16x16) whatever Cin is.  Tap-outer order swept all Cin between re-reads
    // (83 KB-830 KB per CU, x32 CUs per 4 MB L2): rocprofv3 FETCH_SIZE showed 2.1x (Cin=128) to 6.3x (Cin=320)
    // the compulsory bytes and a 66-76 % L2 hit rate (profiles/r01_pmc_conv_v0.json).
    int i_tap = 0, i_kc = 0;
    int i_step = 0, i_end = 0;                         // K-step at the cursor / end of the cursor's K range
    int issue_tile = first;                            // work item the cursor is in
    int issue_on = 1;                                  // 0 once the last item's last step has been requested
    int n_ahead = 0;                                   // K-steps requested but not yet consumed

    // ---- issue side: buffer-addressed DMA (buffer_load_dwordx4 ... lds).  Both operands are described by a raw buffer
    // resource (the launcher admits tensors below 4 GB), so a lane's source is a 32-bit byte offset and an out-of-image
    // tap is the offset OOB >= num_records: the hardware returns zeros for it -- no zero page, no 64-bit pointer
    // arithmetic, half the address registers of the pointer version (which spilled once the ring ran across tiles).
    // A rows of this lane: wave*(BMT/NW) + 8*i + lrow.  Everything that depends on the row is computed ONCE per tile:
    // the byte offset of tap (0,0) / channel 0 of the row's receptive field (pre-swizzled chunk; it may lie before the
    // tensor -- it wraps mod 2^32 and is only used for taps whose bit is set in a_mask, where offset + tap offset is a
    // true in-tensor offset again) and a 9-bit tap-validity mask.  The K-loop then needs one 32-bit add and one select
    // per DMA instruction (the first version recomputed ((b*Hin + y)*Win + x)*Cin per tap: ~110 VALU / 24 quarter-rate
    // multiplies per K-step per wave, more issue time than the 16 MFMAs they feed).
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((long long)a.B * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (int)(unsigned)((long long)a.Cout * TAPS * a.Cin * 2 * (a.wb_y > 0 ? a.wb_count : 1)), 0x00020000);
    // epilogue tensors: an absent bias / rowbias / residual is a zero-sized descriptor (every load returns 0)
    const int OC = EPI == 1 ? a.Cout / 2 : a.Cout;      // output channels = row length of y / res
    const unsigned y_bytes = (unsigned)(a.M * OC * 2);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (int)y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.Cout * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.rowbias, 0, a.rowbias ? a.B * a.Cout * 2 : 0, 0x00020000);
    constexpr int NST = MT * NT * 2 / (EPI == 1 ? 2 : 1);   // 16-byte stores per wave and tile (exact: dropped ones are issued too)
    unsigned a_off[A_INSTR];
    unsigned a_mask[A_INSTR];
    unsigned b_off[B_INSTR];
    const float rcp_hout = 1.0f / (float)a.Hout;       // B*Hout < 2^22 (launcher): one correction step makes the quotient exact
    [[maybe_unused]] unsigned wb_off = 0;              // byte offset of the issue tile's weight matrix (batched 1-tap launches)
    auto setup_issue_tile = [&](int id) __attribute__((always_inline)) {
        int Y0, X0, n0, ks;
        tile_coords(id, Y0, X0, n0, ks);
        if constexpr (TAPS == 1)                            // (batched GEMM: this tile's weight matrix)
            wb_off = a.wb_y > 0 ? (unsigned)(Y0 / a.wb_y) * (unsigned)(a.Cout * a.Cin * 2) : 0u;
        i_step = k_lo(ks); i_end = k_lo(ks + 1);
        i_kc = TAPS == 1 ? i_step : i_step / TAPS;
        i_tap = TAPS == 1 ? 0 : i_step - i_kc * TAPS;
        // re-derive the lane constants from an opaque copy of the thread id: computed once before the tile loop they stay
        // live across the K loop, where there is no register for them -- the compiler spilled them and every reload in
        // here waited (in order) behind the DMAs just issued: 23 serialized scratch round trips, 11 k cycles per tile
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int wave = t_ >> 6, lrow = (t_ & 63) >> 3, lslot = t_ & 7;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int row = wave * (BMT / NWL) + 8 * i + lrow;
            const int Y = Y0 + (row >> a.tw_log2);
            const int xo = X0 + (row & TWm);
            const bool ok = Y < rows_total && xo < a.Wout;
            if constexpr (TAPS == 1) {
                // the Linear layers (dm_gemm: one image, stride 1, no padding): row m = Y * Wout + xo of x, no window to clip.
                // With K = 320 a tile is 5 K-steps and this set-up runs inside one of them: the general form below was 234 VALU
                // instructions per crossing, 1600 of a tile's 27 000 cycles (DREAMMAT_CONV_TIMELINE=2, round 5)
                a_off[i] = (((unsigned)Y * (unsigned)a.Wout + (unsigned)xo) * (unsigned)a.Cin + (unsigned)((lslot ^ ((row >> 1) & 7)) * 8)) * 2u;
                a_mask[i] = ok ? 1u : 0u;
                continue;
            }
            const int Yc = ok ? Y : 0;
            int b = (int)((float)Yc * rcp_hout);
            int yo = Yc - b * a.Hout;
            if (yo < 0) { yo += a.Hout; --b; }
            if (yo >= a.Hout) { yo -= a.Hout; ++b; }
            const int y0 = yo * a.stride - a.pad_y;
            const int x0 = (ok ? xo : 0) * a.stride - a.pad_x;
            // mod 2^32 on purpose (see above): every product wraps consistently
            a_off[i] = ((((unsigned)b * (unsigned)a.Hin + (unsigned)y0) * (unsigned)a.Win + (unsigned)x0) * (unsigned)a.Cin +
                        (unsigned)((lslot ^ ((row >> 1) & 7)) * 8)) * 2u;
            // taps (dy, dx) inside the image: 3 (TAPS == 4: 2) column bits replicated into the valid rows; bit index = tap
            constexpr int TWX = TAPS == 4 ? 2 : 3;
            unsigned xb = ((unsigned)x0 < (unsigned)a.Win ? 1u : 0u) | ((unsigned)(x0 + 1) < (unsigned)a.Win ? 2u : 0u) |
                          (TWX == 3 && (unsigned)(x0 + 2) < (unsigned)a.Win ? 4u : 0u);
            if (!ok) xb = 0;
            a_mask[i] = ((unsigned)y0 < (unsigned)a.Hin ? xb : 0u) | ((unsigned)(y0 + 1) < (unsigned)a.Hin ? xb << TWX : 0u) |
                        (TWX == 3 && (unsigned)(y0 + 2) < (unsigned)a.Hin ? xb << 6 : 0u);
        }
        // weight rows past Cout (ragged last tile, e.g. 320 = 2.5 x 128) re-read row Cout-1: finite values that
        // only reach accumulator rows the epilogue never stores
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int row = wave * (BN / NWL) + 8 * i + lrow;
            const int rc = min(n0 + row, a.Cout - 1);
            b_off[i] = ((unsigned)rc * (unsigned)TAPS * (unsigned)a.Cin + (unsigned)((lslot ^ ((row >> 1) & 7)) * 8)) * 2u;
        }
    };
    unsigned toff = 0, woff = 0;                       // byte offsets of the step being issued
    unsigned bit = 1u;
    auto cursor_set = [&]() __attribute__((always_inline)) {
        // tap -> (dy, dx): 3 x 3 window (tap / 3, tap % 3 for tap < 9) or, TAPS == 4, the 2 x 2 window of the stride-2 data gradient
        const int dy = TAPS == 4 ? (i_tap >> 1) : ((i_tap * 11) >> 5), dx = TAPS == 4 ? (i_tap & 1) : (i_tap - 3 * dy);
        toff = (unsigned)(((dy * a.Win + dx) * a.Cin + i_kc * BK) * 2);              // from a_off
        woff = (unsigned)((i_tap * a.Cin + i_kc * BK) * 2) + (TAPS == 1 ? wb_off : 0u);   // weights are [Cout][tap][Cin] (+ the item's matrix)
        bit = 1u << i_tap;
    };
    // advance to the next K-step; at the end of a tile move on to this workgroup's next tile (or stop)
    auto cursor_next = [&]() __attribute__((always_inline)) {
        ++n_ahead;
        if (TAPS == 1 || ++i_tap == TAPS) { i_tap = 0; ++i_kc; }
        if (++i_step == i_end) {
            issue_tile += wpx;
            if (issue_tile < xend) setup_issue_tile(issue_tile);
            else issue_on = 0;
        }
    };
#if defined(DM_ABL_VGPRSTAGE)
    u32x4 abl_v[L];
#pragma unroll
    for (int p = 0; p < L; ++p) abl_v[p] = u32x4{0u, 0u, 0u, 0u};
#endif
    // DMA piece p of the step at the cursor (p < A_INSTR: 8 activation rows, else 8 weight rows) into `stage`
    auto piece = [&](int p, int stage) __attribute__((always_inline)) {
#if defined(DM_ABL_NODMA)
        return;                                         // ABLATION (wrong results): no operand traffic at all
#endif
        if (NWL != NW && wave >= NWL) return;           // (asymmetric DMA: this wave requests nothing)
#if defined(DM_ABL_NOA)
        if (p < A_INSTR) return;                        // ABLATION (wrong results): weights only
#endif
#if defined(DM_ABL_NOB)
        if (p >= A_INSTR) return;                       // ABLATION (wrong results): activations only
#endif
        char* ab = smem + stage * STAGE;
#if defined(DM_ABL_VGPRSTAGE)
        // ABLATION (wrong results, right traffic): the same requests as plain buffer_load_dwordx4 into registers, written to LDS with
        // ds_write_b128 one step after they were requested (into the stage this call names) -- the classic global -> VGPR -> LDS path in
        // place of the LDS-DMA.  Round 6: 577 vs 792 TF/s on [24576, 2560, 640], 416 vs 504 on [24576, 640, 640], 486 vs 617 on
        // [6144, 1280, 1280] (the 256 x 128 tile, 217 registers, no spills): the LDS-DMA is the cheaper way to ask on this chip.
        *reinterpret_cast<u32x4*>(ab + (p < A_INSTR ? (wave * (BMT / NW) + 8 * p) * ROWB : A_BYTES + (wave * (BN / NW) + 8 * (p - A_INSTR)) * ROWB) + lane * 16) = abl_v[p];
        if (p < A_INSTR) abl_v[p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)((a_mask[p] & bit) ? a_off[p] + toff : OOB), 0, 0);
        else abl_v[p] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)b_off[p - A_INSTR], (int)woff, 0);
        return;
#endif
        if (p < A_INSTR) {
            // select, never a branch: the DMA must execute with ALL lanes active (an inactive lane would
            // leave its LDS slot stale); out-of-image taps read zeros through the descriptor's range check
            const unsigned vo = (a_mask[p] & bit) ? a_off[p] + toff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(ab + (wave * (BMT / NW) + 8 * p) * ROWB),
                                                     16, (int)vo, 0, 0, 0);
        } else {
            const int q = p - A_INSTR;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(ab + A_BYTES + (wave * (BN / NW) + 8 * q) * ROWB),
                                                     16, (int)b_off[q], (int)woff, 0, 0);
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        cursor_set();
#pragma unroll
        for (int p = 0; p < L; ++p) piece(p, stage);
        cursor_next();
    };

    f32x16 acc[MT][NT];

    // ring protocol: at the top of step s every wave waits for ITS OWN step-s DMAs (counted vmcnt leaves the
    // NSTAGE-2 younger steps in flight), then the barrier makes all waves' step-s data visible and proves
    // everybody has finished reading stage (s-1) % NSTAGE, which is the stage the next issue overwrites.
    // STAGGER: every tile costs the same, so without this all workgroups reach their epilogues together and the chip
    // writes one 32-45 MB burst per tile round at the HBM write rate (13.6 us per round at 8 x 128->128 @512^2, during
    // which the in-order vmcnt holds the next tile's DMA waits behind the stores).  Workgroup k of an XCD starts
    // (k % 8) / 8 of a tile late, so the stores of the 8 phases interleave with the other phases' K-loops.
    if (stagger_units > 0) {
        const int ph = (int)((blockIdx.x >> 3) & 7);
        for (int t = 0; t < ph * stagger_units; ++t) __builtin_amdgcn_s_sleep(127);
    }
    setup_issue_tile(first);
    issue(0);
    if (NSTAGE == 3 && issue_on) issue(1);
    // One K-step: 4 chunks of 16 K.  Chunk kk (a) reads the fragments of chunk kk+1 into the other register set,
    // (b) issues its quarter of the NEXT stage's DMA pieces, (c) runs its MT*NT MFMAs on fragments that were read
    // one chunk earlier -- so LDS latency and the DMA issue time (60-180 cycles per 1 KB piece) sit under MFMA
    // execution instead of in front of it (the first version issued all pieces, then read, then multiplied:
    // SQ_WAIT_ANY 39 %, MFMA busy 29 %, profiles/r01_pmc_conv_v0.json).
    [[maybe_unused]] bool abl_read = true;
    auto read_frags = [&](int stage, int kk, elem8 (&af)[MT], elem8 (&bf)[NT]) {
#if defined(DM_ABL_NOLDS)
        if (!abl_read) { asm volatile("" : "+v"(af[0]), "+v"(bf[0])); return; }       // ABLATION (wrong results): fragments read in a tile's first K-step only
#endif
        const char* ab = smem + stage * STAGE;
        const char* bb = ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int r = TM * wm + 32 * i + l31;
            af[i] = *reinterpret_cast<const elem8*>(ab + r * ROWB + (((2 * kk + hi) ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int r = TN * wn + 32 * j + l31;
            bf[j] = *reinterpret_cast<const elem8*>(bb + r * ROWB + (((2 * kk + hi) ^ ((r >> 1) & 7)) << 4));
        }
    };
    auto mma = [&](const elem8 (&af)[MT], const elem8 (&bf)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                // weights as the A operand, pixels as B: the accumulator fragment is D^T[n][m] -- lane = pixel, registers =
                // channels, so the epilogue can store 16 contiguous bytes (8 channels of one pixel) per lane
                acc[i][j] = DM_MFMA_32x32x16(bf[j], af[i], acc[i][j]);
    };
    // ONE body for steps that issue and steps that only drain (`issue_on` is wave-uniform and only guards the DMA pieces):
    // two instantiations in one loop made the register allocator give each its own accumulator set.
    // The MFMAs of a step's LAST chunk are issued after the NEXT step's barrier, behind that step's first fragment reads:
    // with two waves per SIMD the matrix pipe otherwise drains at every barrier for (vmcnt wait + barrier skew + LDS read
    // latency) -- MFMA busy was 46 % of the kernel's cycles with zero bank conflicts and 3 % LDS waits
    // (profiles/r02_pmc_conv_*.json).  The fragments are in registers by then (lgkmcnt(0) before the barrier), so the
    // stage they came from may be overwritten.
    constexpr bool PEND = MT * NT < 10;                // (the 64 x 160 wave tile has no registers left to carry a set across)
    elem8 a0[MT], b0[NT], a1[MT], b1[NT];             // two fragment sets, used alternately (named, not indexed:
                                                       // a parity-indexed array is not promoted to registers)
    auto kstep = [&](int stage, int st_next, bool pending) __attribute__((always_inline)) {
        // 3-deep ring: the pieces have a whole extra step to land, spread them over all 4 chunks.  2-deep ring:
        // they are needed at the next barrier, so the last chunk issues nothing (>= a quarter step of lead time)
        constexpr int NCH = (NSTAGE == 2) ? 3 : 4;
        const bool on = __builtin_amdgcn_readfirstlane(issue_on) != 0;
        auto pieces = [&](int kk) __attribute__((always_inline)) {
            if (on && kk < NCH) {
#pragma unroll
                for (int p = kk; p < L; p += NCH) piece(p, st_next);
            }
        };
        if (on) cursor_set();
        read_frags(stage, 0, a0, b0);
        if (PEND && pending) mma(a1, b1);              // chunk 3 of the previous step of this tile
        __builtin_amdgcn_sched_barrier(0);
        read_frags(stage, 1, a1, b1); pieces(0); mma(a0, b0); __builtin_amdgcn_sched_barrier(0);
        read_frags(stage, 2, a0, b0); pieces(1); mma(a1, b1); __builtin_amdgcn_sched_barrier(0);
        read_frags(stage, 3, a1, b1); pieces(2); mma(a0, b0); __builtin_amdgcn_sched_barrier(0);
        pieces(3);
        if (!PEND) mma(a1, b1);
        if (on) cursor_next();
    };
    int stage = 0;
    int tl_i = 0;
    unsigned long long tl_sum[5] = {0, 0, 0, 0, 0}, tl_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&]() __attribute__((always_inline)) {
        if (a.timeline && a.timeline_steps != 4 && tid == 0 && tl_i < 64) a.timeline[(long long)blockIdx.x * 64 + tl_i++] = __builtin_amdgcn_s_memtime();
    };
    for (int ct = first; ct < xend; ct += wpx) {
        int Y0, X0, n0, ks;
        tile_coords(ct, Y0, X0, n0, ks);
        const int s_lo = k_lo(ks), s_hi = k_lo(ks + 1);
        stamp();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int s = s_lo; s < s_hi; ++s) {
            // Wait for the DMAs of the step being consumed; vmcnt counts in issue order, so what may stay in flight is
            // whatever was issued AFTER them: (NSTAGE == 3) the next step's L DMAs, and -- during the first NSTAGE-1 steps
            // of every tile but the first -- the NST epilogue stores of the previous tile, which were issued between this
            // step's DMAs and the ones that followed.  Counting them lets the stores drain under the new tile's first
            // K-steps instead of being waited for at its first barrier.
            const bool two = NSTAGE == 3 && n_ahead >= 2;
            // DREAMMAT_CONV_TIMELINE=4: s_memtime differences summed in SGPRs (no stores inside the loop, every wave):
            // MFMA body | fragment-read wait | DMA wait | barrier.  Each column carries one ~130-tick s_memtime round trip.
            const bool tl4 = a.timeline_steps == 4;
            unsigned long long t_a = 0, t_b = 0, t_c = 0;
            if (tl4) { t_a = __builtin_amdgcn_s_memtime(); tl_sum[0] += t_a - tl_prev; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (tl4) { t_b = __builtin_amdgcn_s_memtime(); tl_sum[1] += t_b - t_a; }
#if defined(DM_ABL_NOWAIT)
            if (false) {}                               // ABLATION (wrong results): DMA issued, never waited for
            else
#endif
            if (s - s_lo < NSTAGE - 1 && ct != first && !(SPLIT_OK && ksplit > 1)) {
                if (two && (NWL == NW || wave < NWL)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L + NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            } else {
                if (two && (NWL == NW || wave < NWL)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (tl4) { t_c = __builtin_amdgcn_s_memtime(); tl_sum[2] += t_c - t_b; }
#if !defined(DM_ABL_NOBAR)
            __builtin_amdgcn_s_barrier();               // (ABLATION DM_ABL_NOBAR, wrong results: no workgroup barrier in the K loop)
#endif
            if (tl4) { tl_prev = __builtin_amdgcn_s_memtime(); tl_sum[3] += tl_prev - t_c; ++tl_sum[4]; }
            if (a.timeline_steps == 2) stamp();
            --n_ahead;
            int st2 = stage + NSTAGE - 1; if (st2 >= NSTAGE) st2 -= NSTAGE;
            kstep(stage, st2, s > s_lo);
            stage = stage + 1; if (stage >= NSTAGE) stage = 0;
            abl_read = false;
        }
        abl_read = true;
        if (PEND) mma(a1, b1);                         // the last step's last chunk
        stamp();

        // ---- epilogue: y = acc + bias[n] (+ rowbias[image(m), n]) (+ res[m, n]), one rounding to bf16.
        // D^T layout: lane = pixel l31 of the fragment, register r = channel (r&3) + 8*(r>>2) + 4*hi.  Group g = r>>2 is
        // four consecutive channels (8 bytes as bf16); lanes l and l+32 hold the two halves of one 8-channel run.  One
        // v_permlane32_swap per dword on the group pair (g, g+1) turns that into 16 contiguous bytes per lane: lanes 0-31
        // get channels 8g..8g+7, lanes 32-63 channels 8g+8..8g+15 of their pixel -> TWO 16-byte stores per 32x32
        // fragment.  (The first version stored every element on its own: 128 store instructions per wave and tile.)
        // Every access is a buffer instruction whose out-of-range cases (no bias / rowbias / residual tensor, ragged
        // pixel or channel) are the offset OOB or a zero-sized descriptor -- loads return 0, stores are dropped -- so the
        // epilogue is straight-line code: the compiler batches the loads of a fragment ahead of their use (the branchy
        // version waited vmcnt(0) after each of its 8-byte loads: 12 exposed round trips per fragment, 15 us per tile,
        // the 0.25 ms of the 512^2 layers that did not scale with K in tools/conv_fit.sh), and each wave issues EXACTLY
        // NST stores per tile, which is what lets the next tile's first K-steps count them in s_waitcnt (see above).
        const int img0 = TAPS == 1 ? 0 : Y0 / a.Hout;  // image of the patch's first row (wave-uniform)
        const int rem0 = Y0 - img0 * a.Hout;
        unsigned poff[MT], rboff[MT];                  // byte offset of this lane's pixel in y / res, of its image's rowbias row
        // (EPI == 2, sub-pixel store) byte offset of sub-pixel (0, 0) of this lane's pixel in the 2x finer tensor, and which of its
        // four sub-pixels exist there (bit 2 py + px)
        [[maybe_unused]] unsigned s_base[MT], s_ok[MT];
        int te_ = tid;                                 // (opaque copy: see setup_issue_tile)
        asm volatile("" : "+v"(te_));
        const int l31 = te_ & 31, hi = (te_ >> 5) & 1, wm = (te_ >> 6) / WNW, wn = (te_ >> 6) % WNW;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int d = TM * wm + 32 * i + l31;      // this lane's pixel of the tile
            const int ty = d >> a.tw_log2;
            const int Y = Y0 + ty;
            const int xo = X0 + (d & TWm);
            const bool pix_ok = Y < rows_total && xo < a.Wout;
            poff[i] = pix_ok ? ((unsigned)Y * (unsigned)a.Wout + (unsigned)xo) * (unsigned)OC * 2u : OOB;   // m = (b*Hout + yo)*Wout + xo
            if constexpr (TAPS == 1) { rboff[i] = 0; continue; }                              // (dm_gemm: no per-image bias row)
            int img = img0, t = rem0 + ty;
            while (t >= a.Hout) { t -= a.Hout; ++img; }                                       // a patch spans at most TH / Hout + 1 images
            rboff[i] = (unsigned)(min(img, a.B - 1) * a.Cout * 2);
            if constexpr (EPI == 2) {
                const bool up = a.shuf_mode == 2;
                const int Wd = up ? 2 * (a.Wout - 1) : 2 * a.Wout;
                const int row = 2 * Y - (up ? 2 * img : 0), col = 2 * xo;          // mode 2: image b starts at tall row 2 b (Hout - 1)
                s_base[i] = ((unsigned)row * (unsigned)Wd + (unsigned)col) * (unsigned)a.shuf_cs * 2u;      // (mod 2^32 when a sub-pixel is missing)
                unsigned okb = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int py = q >> 1, px = q & 1;
                    const bool ok = !up || (t >= py && t - py < a.Hout - 1 && xo >= px && xo - px < a.Wout - 1);
                    okb |= (pix_ok && ok ? 1u : 0u) << q;
                }
                s_ok[i] = okb;
            }
        }
        auto unpack_add = [](float (&v)[4], const u32x2 p) __attribute__((always_inline)) {
            v[0] += dm_elem_lo(p[0]); v[1] += dm_elem_hi(p[0]);
            v[2] += dm_elem_lo(p[1]); v[3] += dm_elem_hi(p[1]);
        };
        const bool has_res = a.res != nullptr, has_rb = a.rowbias != nullptr;
        // Loads are issued ahead of the stores (the compiler keeps program order between a buffer load and a buffer store
        // it cannot prove disjoint): the rowbias row and the residual of fragment f + 1 are requested before the stores of
        // fragment f, so only the first fragment of a tile waits for a full round trip.
        auto frag_load = [&](int i, int j, u32x2 (&r)[8]) __attribute__((always_inline)) {
            const int nbase = n0 + TN * wn + 32 * j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + 8 * g + 4 * hi;
                const bool n_ok = n < a.Cout;
                if (has_rb) r[g] = __builtin_amdgcn_raw_buffer_load_b64(rbrs, (int)(n_ok ? rboff[i] + (unsigned)n * 2u : OOB), 0, 0);
                if (has_res)
                    r[4 + g] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(n_ok && poff[i] != OOB ? poff[i] + (unsigned)n * 2u : OOB), 0, 0);
            }
        };
        if (SPLIT_OK && ksplit > 1) {
            // fp32 partial sums of this K range: ws[ks][m][n], 16 bytes (4 consecutive channels) per lane and group
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(ws + (long long)ks * a.M * a.Cout), 0, (int)(unsigned)(a.M * a.Cout * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + TN * wn + 32 * j + 8 * g + 4 * hi;
                        // (element copies first: __builtin_bit_cast applied directly to an ext-vector element reads element 0)
                        const float f0 = acc[i][j][4 * g], f1 = acc[i][j][4 * g + 1], f2 = acc[i][j][4 * g + 2], f3 = acc[i][j][4 * g + 3];
                        const u32x4 out = {__builtin_bit_cast(unsigned, f0), __builtin_bit_cast(unsigned, f1),
                                           __builtin_bit_cast(unsigned, f2), __builtin_bit_cast(unsigned, f3)};
                        __builtin_amdgcn_raw_buffer_store_b128(out, prs, (int)(poff[i] != OOB && n < a.Cout ? poff[i] * 2u + (unsigned)n * 4u : OOB), 0, 0);
                    }
        } else if constexpr (EPI == 1) {
            // GEGLU: fragments (2jj, 2jj+1) = value / gate of output channels ob .. ob+31 (rows interleaved by the host).
            // The gate function is the exact (erf) GELU evaluated as value * gate * Phi(gate) with
            //   Phi(g) = g < 0 ? h : 1 - h,   h = erfc(|g| / sqrt 2) / 2 = poly5(t) 2^(-z^2),  t = 1 / (1 + p |z|),  z = g sqrt(log2(e) / 2)
            // (Abramowitz-Stegun 7.1.26, coefficients halved; |error| of Phi < 3e-7, the negative tail without the cancellation
            // of 1 + erf), two elements per instruction on the packed fp32 pipe: 17 VALU instructions per PAIR of outputs
            // where 0.5 g (1 + erff(g / sqrt 2)) took ~44 per ELEMENT, divergent branches included.  At K = 320 this epilogue was
            // 70 % of a tile (2834 VALU instructions per wave against 5 K-steps of 32 MFMAs; round 5).
            static_assert(EPI != 1 || NT % 2 == 0, "GEGLU needs value/gate fragment pairs");
            const f32x2 kKz = {0.84932180028801905f, 0.84932180028801905f};          // sqrt(log2(e) / 2)
            const f32x2 kKp = {0.27273748f, 0.27273748f};                              // 0.3275911 / sqrt(log2 e)
            const f32x2 kA5 = {0.5307027145f, 0.5307027145f}, kA4 = {-0.7265760135f, -0.7265760135f},
                        kA3 = {0.7107068705f, 0.7107068705f}, kA2 = {-0.142248368f, -0.142248368f}, kA1 = {0.127414796f, 0.127414796f};
            const f32x2 kOne = {1.f, 1.f};
            auto widen = [](const unsigned w) __attribute__((always_inline)) { return f32x2{dm_elem_lo(w), dm_elem_hi(w)}; };
#pragma unroll
            for (int jj = 0; jj < NT / 2; ++jj) {
                const int nb = n0 + TN * wn + 64 * jj;     // first (interleaved) weight row of the value fragment
                const int ob = (n0 + TN * wn) / 2 + 32 * jj;
                f32x2 bv[4][2], bg[4][2];                  // bias of this lane's 16 value / gate channels, widened once per fragment pair
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + 8 * g + 4 * hi;
                    const u32x2 pv = __builtin_amdgcn_raw_buffer_load_b64(brs, (int)(n < a.Cout ? (unsigned)n * 2u : OOB), 0, 0);
                    const u32x2 pg = __builtin_amdgcn_raw_buffer_load_b64(brs, (int)(n + 32 < a.Cout ? (unsigned)(n + 32) * 2u : OOB), 0, 0);
                    bv[g][0] = widen(pv[0]); bv[g][1] = widen(pv[1]);
                    bg[g][0] = widen(pg[0]); bg[g][1] = widen(pg[1]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int g = 2 * gp + q;
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                // (element copies first: indexing an ext-vector element inside an initialiser list is fine, bit_cast is not)
                                const float v0 = acc[i][2 * jj][4 * g + 2 * h2], v1 = acc[i][2 * jj][4 * g + 2 * h2 + 1];
                                const float g0 = acc[i][2 * jj + 1][4 * g + 2 * h2], g1 = acc[i][2 * jj + 1][4 * g + 2 * h2 + 1];
                                // (rounds 2-4 rounded value and gate to 16 bits here, as the unfused Linear -> gelu -> mul sequence does
                                // between its kernels: 6 of the 23 instructions per pair, for a result FARTHER from the fp32 one)
                                const f32x2 vr = f32x2{v0, v1} + bv[g][h2];
                                const f32x2 gr = f32x2{g0, g1} + bg[g][h2];
                                const f32x2 z = gr * kKz;
                                const f32x2 zz = z * z;
                                const f32x2 az = {__builtin_fabsf(z[0]), __builtin_fabsf(z[1])};
                                const f32x2 d = __builtin_elementwise_fma(az, kKp, kOne);
                                const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                                const f32x2 e2 = {__builtin_amdgcn_exp2f(-zz[0]), __builtin_amdgcn_exp2f(-zz[1])};
                                f32x2 pl = __builtin_elementwise_fma(t, kA5, kA4);
                                pl = __builtin_elementwise_fma(pl, t, kA3);
                                pl = __builtin_elementwise_fma(pl, t, kA2);
                                pl = __builtin_elementwise_fma(pl, t, kA1);
                                const f32x2 hh = pl * t * e2;              // erfc(|g| / sqrt 2) / 2
                                const f32x2 uu = kOne - hh;
                                const f32x2 phi = {gr[0] < 0.f ? hh[0] : uu[0], gr[1] < 0.f ? hh[1] : uu[1]};
                                const f32x2 o = vr * gr * phi;
                                w[q][h2] = __builtin_bit_cast(unsigned, __builtin_convertvector(o, elem2));
                            }
                        }
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                        const int c0 = ob + 16 * gp + 8 * hi;
                        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
#if defined(DM_ABL_NOSTORE)
                        if (a.M < 0)      // ABLATION (wrong results): the output is never stored
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(out, yrs, (int)(poff[i] != OOB && c0 < OC ? poff[i] + (unsigned)c0 * 2u : OOB), 0, 0);
                    }
            }
        } else {
        u32x2 pre[2][8];                               // [parity of the fragment counter][rowbias g 0..3 | residual g 0..3]
#pragma unroll
        for (int g = 0; g < 8; ++g) pre[0][g] = pre[1][g] = u32x2{0u, 0u};
        // Fragment order.  PIXEL-MAJOR when a wave has at most two channel fragments: the stores that together fill one
        // 128-byte line of a pixel row (64 channels = 2 fragments x 2 group pairs) are then issued back to back instead of MT
        // fragments apart, so the write path sees whole lines (WRITE_SIZE was 1.33x the tensor on 8 x 128->128 @512^2:
        // lines written back partially and again).  Channel-major (bias registers per j) for the 5-fragment wave tile.
        constexpr bool PIX_MAJOR = NT <= 2;
        constexpr int NBP = PIX_MAJOR ? NT : 1;
        u32x2 bp[NBP][4];                              // bias of this lane's 16 channels of fragment j (register 4g+e = channel nbase + 8g + 4hi + e)
        auto load_bias = [&](int j, u32x2 (&bq)[4]) __attribute__((always_inline)) {
            const int nbase = n0 + TN * wn + 32 * j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + 8 * g + 4 * hi;
                bq[g] = __builtin_amdgcn_raw_buffer_load_b64(brs, (int)(n < a.Cout ? (unsigned)n * 2u : OOB), 0, 0);
            }
        };
        if (PIX_MAJOR) {
#pragma unroll
            for (int j = 0; j < NT; ++j) load_bias(j, bp[j % NBP]);
        }
        frag_load(0, 0, pre[0]);
#pragma unroll
        for (int f = 0; f < MT * NT; ++f) {
            constexpr int kLast = MT * NT - 1;
            const int i = PIX_MAJOR ? f / NT : f % MT, j = PIX_MAJOR ? f % NT : f / MT;
            const int nbase = n0 + TN * wn + 32 * j;   // first channel of this fragment (wave-uniform)
            if (!PIX_MAJOR && i == 0) load_bias(j, bp[0]);
            if (f < kLast) frag_load(PIX_MAJOR ? (f + 1) / NT : (f + 1) % MT, PIX_MAJOR ? (f + 1) % NT : (f + 1) / MT, pre[(f + 1) & 1]);
            {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {       // group pairs (0,1) and (2,3)
                    unsigned w[2][2];                  // [group of the pair][dword] = 4 bf16 per group
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int g = 2 * gp + q;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                        unpack_add(v, bp[PIX_MAJOR ? j : 0][g]);
                        unpack_add(v, pre[f & 1][g]);
                        unpack_add(v, pre[f & 1][4 + g]);
                        f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                        elem2 plo = __builtin_convertvector(lo, elem2), phi = __builtin_convertvector(hi2, elem2);
                        w[q][0] = __builtin_bit_cast(unsigned, plo);
                        w[q][1] = __builtin_bit_cast(unsigned, phi);
                    }
                    // v_permlane32_swap(vdst = group g, src = group g+1): lanes 32-63 of vdst <-> lanes 0-31 of src.  Afterwards
                    //   result[0]: lanes 0-31 own group g (ch 8g..8g+3)        | lanes 32-63 the lower lanes' group g+1 (ch 8g+8..+11)
                    //   result[1]: lanes 0-31 the upper lanes' group g (+4..+7) | lanes 32-63 own group g+1 (ch 8g+12..+15)
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    const int c0 = nbase + 16 * gp + 8 * hi;   // this lane now owns channels c0 .. c0+7 of its pixel
                    const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
                    unsigned off = poff[i] != OOB && c0 < OC ? poff[i] + (unsigned)c0 * 2u : OOB;
                    if constexpr (EPI == 2) {
                        // sub-pixel store.  Cs % 16 == 0: the 16 channels of this store pair lie inside ONE block, which block is
                        // wave-uniform (scalar arithmetic); block 2 py + px of pixel (u, v) -> sub-pixel (2u +- py, 2v +- px)
                        const int cs = a.shuf_cs, cb = nbase + 16 * gp;
                        const int blk = (cb >= cs) + (cb >= 2 * cs) + (cb >= 3 * cs), py = blk >> 1, px = blk & 1;
                        const bool up = a.shuf_mode == 2;
                        const int Wd = up ? 2 * (a.Wout - 1) : 2 * a.Wout;
                        const int delta = ((up ? -1 : 1) * (py * Wd + px) * cs + (cb - blk * cs)) * 2;
                        off = ((s_ok[i] >> blk) & 1u) && c0 < OC ? s_base[i] + (unsigned)delta + 16u * (unsigned)hi : OOB;
                    }
#if defined(DM_ABL_NOSTORE)
                    if (a.M < 0)      // ABLATION (wrong results): the output is never stored
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(out, yrs, (int)off, 0, 0);
                }
            }
        }
        }
        stamp();
        if (a.timeline_steps == 4) tl_prev = __builtin_amdgcn_s_memtime();     // (the epilogue is not part of the next body)
    }   // tile loop
    if (a.timeline && a.timeline_steps == 4 && lane == 0)      // every wave's sums: slots 8 + 5 * wave ..
        for (int i = 0; i < 5; ++i) a.timeline[(long long)blockIdx.x * 64 + 8 + 5 * wave + i] = tl_sum[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// Variant 3 (round 6): HALO PATCH.  The ablation builds of round 6 (profiles/r06_conv_ablation.md) say what the K loop of the
// kernel above loses its time to: not waiting for its operands (+0-1.5 % with the waits removed) but REQUESTING them -- every
// `buffer_load_dwordx4 ... lds` holds its SIMD for the better part of 100 cycles, 8-10 of them per wave and 64-wide K-step beside
// 32 MFMAs, +17-33 % with none of them, and roughly linear in their number (weights only: +12-24 %, activations only: +6-7 %).
// A 3 x 3 convolution requests every activation row NINE times (once per tap).  Here a workgroup owns a TW x TH patch of output
// pixels of ONE image and keeps the (TH + 2) x (TW + 2)-pixel input patch of a 64-channel block in LDS: it is requested ONCE per
// channel block (its pieces spread over the nine K-steps of the block BEFORE, into the other of two buffers) and read at nine
// shifted offsets; only the weights still arrive per K-step.  Requests per MFMA: 512 x 128 tile 1.25 -> 384 x 128 patch 0.47;
// 256 x 256 tile 1.0 -> 256 x 256 patch 0.57.
//  * patch pixel p = prow * (TW + 2) + pcol lives at p * 128 B, its 16-byte chunks XOR-swizzled by (pcol >> 1) & 7: the 16 lanes
//    of a ds_read_b128 group are 16 consecutive columns (whatever the tap's shift), pcol & 1 picks the bank half (the row
//    pitch of 18 pixels is even) and (pcol >> 1) & 7 the 16-byte group -> conflict-free for all nine taps;
//  * K order: channel block outer, then tap COLUMN dx, then tap row dy: a lane's MT x 4 fragment addresses are recomputed when
//    dx changes (three times per block), dy is an immediate offset.  (The kernel above runs dy-major: same sums, fp32 additions
//    in another order -- the two kernels agree to the last bf16 rounding, not bit for bit.)
//  * out-of-image patch pixels (the zero padding, and rows / columns past a ragged last band) are buffer offsets beyond
//    num_records, as above; a patch never straddles two images (bands are per image, the last one ragged).
//  * stride 1, pad 1, 3 x 3 only; no split-K (large-M layers only); epilogue = the plain one of the kernel above.
//  * GN = true (round 6): the GroupNorm [+ SiLU] that diffusers runs in front of the convolution (ResnetBlock2D: conv(silu(norm(x))))
//    is applied to the patch IN LDS: a wave transforms the 1 KB piece it requested one step earlier (it knows the piece has landed:
//    its own vmcnt), y = act(x * A[image, channel] + S[image, channel]) rounded to 16 bits -- the value the apply pass would have
//    stored -- and out-of-image pixels stay zero (the convolution pads the NORMALISED tensor).  The 64 + 64 coefficients of a
//    channel block arrive by two 256-byte LDS-DMA requests of one wave per block.  The apply pass (a read and a write of the whole
//    activation per GroupNorm) disappears; the statistics pass stays.
template <int TH, int BN, int WMW, int NB, bool GN = false>
__global__ __launch_bounds__(512) void k_conv3x3_halo(ConvArgs a, int tiles_x, int tiles_y, int n_nt) {
    constexpr int NW = 8, TW = 16, BMT = TW * TH, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
    constexpr int PA = (NPIX + 7) / 8;                      // 1 KB pieces (8 pixels x 128 B) of a patch
    constexpr int A_BYTES = PA * 1024, B_BYTES = BN * 128;
    [[maybe_unused]] constexpr int WNW = NW / WMW, TM = BMT / WMW, TN = BN / WNW, MT = TM / 32, NT = TN / 32;
    [[maybe_unused]] constexpr int B_INSTR = BN / 8 / NW;   // weight pieces per wave and K-step
    static_assert(TM % 32 == 0 && TN % 32 == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert((PA + NW - 1) / NW <= 9, "a wave requests at most one patch piece per tap");
    static_assert(NB == 2 || NB == 3, "weight ring depth");
    static_assert(2 * A_BYTES + NB * B_BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // patch 0 | patch 1 | NB weight stages | (GN) 2 x (A[64] | S[64]) fp32
#if defined(__HIP_DEVICE_COMPILE__)
    const int total = (int)(a.B * tiles_y * tiles_x) * n_nt;
    const int per_xcd = (total + 7) / 8;
    const int wpx = (int)(gridDim.x >> 3);
    const int xbeg = (int)(blockIdx.x & 7) * per_xcd;
    const int xend = min(xbeg + per_xcd, total);
    const int first = xbeg + (int)(blockIdx.x >> 3);
    if (first >= xend) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WNW, wn = wave % WNW;
    const int H = a.Hout, W = a.Wout, Cin = a.Cin;
    const int n_kc = Cin / 64;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(unsigned)((long long)a.B * H * W * Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)(unsigned)((long long)a.Cout * 9 * Cin * 2), 0x00020000);
    const unsigned y_bytes = (unsigned)(a.M * a.Cout * 2);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? (int)y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.Cout * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.rowbias, 0, a.rowbias ? a.B * a.Cout * 2 : 0, 0x00020000);
    constexpr int NST = MT * NT * 2;                        // 16-byte stores per wave and tile

    // work item id -> image, first row / column of the patch's interior, first output channel (wave-uniform)
    auto coords = [&](int id, int& b, int& Y0, int& X0, int& n0) __attribute__((always_inline)) {
        const unsigned uid = (unsigned)id;
        const unsigned mt = uid / (unsigned)n_nt;
        n0 = (int)(uid - mt * (unsigned)n_nt) * BN;
        const unsigned row = mt / (unsigned)tiles_x;
        X0 = (int)(mt - row * (unsigned)tiles_x) * TW;
        const unsigned bb = row / (unsigned)tiles_y;
        Y0 = (int)(row - bb * (unsigned)tiles_y) * TH;
        b = (int)bb;
    };
    [[maybe_unused]] int sidx_abl = 0;                      // (ablation builds: 0 during the prologue, then 1)
    // Where this lane's slot of this wave's patch piece j (piece index wave + 8 j) lies in the patch does not depend on the tile: one
    // packed word per piece, formed once -- patch row | column << 8 | 16-byte channel chunk of the slot << 16 | (inside the patch) << 20.
    // (Recomputed per step it was ~25 vector instructions per piece, twice with the GroupNorm transform: a division by 18 each.)
    int pinfo[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int i = wave + NW * j;
        const int p = 8 * min(i, PA - 1) + (lane >> 3);
        const int prow = p / PW, pcol = p - prow * PW;
        pinfo[j] = prow | (pcol << 8) | (((lane & 7) ^ ((pcol >> 1) & 7)) << 16) | ((i < PA && p < NPIX) ? 1 << 20 : 0);
    }
    // patch piece j of this wave, channel block kc of the patch at (image b, rows from Y0, columns from X0):
    // this lane's source offset (straight-line code: it is scheduled behind an MFMA, see the K-step) ...
    auto a_offset = [&](int b, int Y0, int X0, int kc, int j) __attribute__((always_inline)) {
        const int pi = pinfo[j];
        const int y = Y0 + (pi & 0xff) - 1, x = X0 + ((pi >> 8) & 0xff) - 1;
        const bool ok = (pi >> 20) && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        // out-of-image = zeros through the descriptor's range check (all lanes execute the request: an inactive lane would leave its
        // LDS slot stale)
        const unsigned off = ((((unsigned)b * (unsigned)H + (unsigned)y) * (unsigned)W + (unsigned)x) * (unsigned)Cin + (unsigned)(kc * 64) +
                              (unsigned)(((pi >> 16) & 7) * 8)) * 2u;
        return ok ? off : OOB;
    };
    // ... and the request itself (waves whose piece index is past the patch request nothing)
    auto issue_a_at = [&](unsigned off, int j, char* buf) __attribute__((always_inline)) {
        const int i = wave + NW * j;
        if (i >= PA) return;
#if defined(DM_ABL_NOA) || defined(DM_ABL_NODMA)
        if (sidx_abl > 0) return;                           // ABLATION (wrong results)
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(buf + i * 1024), 16, (int)off, 0, 0, 0);
    };
    auto issue_a = [&](int b, int Y0, int X0, int kc, int j, char* buf) __attribute__((always_inline)) {
        issue_a_at(a_offset(b, Y0, X0, kc, j), j, buf);
    };
    // weight piece q of this wave for (first output channel n0, channel block kc, tap) into stage `st`
    auto issue_b = [&](int n0, int kc, int tap, int q, char* st) __attribute__((always_inline)) {
#if defined(DM_ABL_NOB) || defined(DM_ABL_NODMA)
        if (sidx_abl > 0) return;                           // ABLATION (wrong results)
#endif
        const int row = wave * (BN / NW) + 8 * q + (lane >> 3);
        const int rc = min(n0 + row, a.Cout - 1);           // rows past Cout re-read the last one (never stored)
        const unsigned off = ((unsigned)rc * 9u * (unsigned)Cin + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(st + (wave * (BN / NW) + 8 * q) * 128), 16,
                                                 (int)off, (tap * Cin + kc * 64) * 2, 0, 0);
    };
    char* const bst0 = smem + 2 * A_BYTES;              // patch buffer k at smem + k * A_BYTES
    [[maybe_unused]] float* const cbuf0 = reinterpret_cast<float*>(smem + 2 * A_BYTES + NB * B_BYTES);     // (GN) coefficient buffer k at cbuf0 + 128 k
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.gn_coef, 0, GN ? (int)(unsigned)((long long)a.B * 7 * Cin * 4) : 0, 0x00020000);
    // (GN) the A and S rows of channel block kc of image b -> coefficient buffer `cb` (one wave: two 256-byte requests)
    [[maybe_unused]] auto issue_coef = [&](int b, int kc, float* cb) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(crs, (__attribute__((address_space(3))) void*)cb, 4, lane * 4, ((b * 7) * Cin + kc * 64) * 4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(crs, (__attribute__((address_space(3))) void*)(cb + 64), 4, lane * 4, ((b * 7 + 1) * Cin + kc * 64) * 4, 0, 0);
    };
    // (GN) patch piece j of this wave, landed in `buf` (patch of image rows from Y0, columns from X0): x -> act(x A + S) in place.
    // STRAIGHT-LINE code in SIX PHASES (`live` false: the same instructions on a 1 KB dummy area, nothing kept): the ~90 vector
    // instructions have to sit BETWEEN the MFMAs of a chunk -- as one block they held the matrix pipe idle for their whole length,
    // every step (conv + GroupNorm 713 us where the plain patch kernel takes 605), and neither the scheduler on its own nor
    // sched_group_barrier patterns would interleave them.  Phase 0: addresses + the five LDS reads; 1-4: a channel pair each ... 5: the write.
    [[maybe_unused]] char* const dummy = smem + 2 * A_BYTES + NB * B_BYTES + 1024;
    [[maybe_unused]] u32x4 gs_raw = {0u, 0u, 0u, 0u}, gs_o = {0u, 0u, 0u, 0u};      // (the transform's state between its phases)
    [[maybe_unused]] f32x4v gs_A0 = {0.f, 0.f, 0.f, 0.f}, gs_A1 = gs_A0, gs_S0 = gs_A0, gs_S1 = gs_A0;
    [[maybe_unused]] int gs_q = 0;                                                   // LDS byte offset of this lane's slot
    [[maybe_unused]] bool gs_ok = false;
    [[maybe_unused]] float gs_z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // phases: 0 = the lane's pixel / slot, 1 = the five LDS reads, 2 .. 9 = one channel each (z = x A + S, SiLU), 10 = pack + select, 11 = write
    constexpr int kGnPhases = 12;
    [[maybe_unused]] bool gs_live = false;
    [[maybe_unused]] auto gn_phase = [&](auto ph_tag, bool live, int Y0, int X0, int j, char* buf, const float* cb) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        if constexpr (PH == 0) {
            const int i_raw = wave + NW * j;
            gs_live = live && i_raw < PA;
            const int pi = pinfo[j];
            const int y = Y0 + (pi & 0xff) - 1, x = X0 + ((pi >> 8) & 0xff) - 1;
            gs_ok = gs_live && (pi >> 20) && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            gs_q = (int)((gs_live ? buf + i_raw * 1024 : dummy) - smem) + (t_ & 63) * 16;
        } else if constexpr (PH == 1) {
            const int cidx = (pinfo[j] >> 16) & 7;              // which 8 channels of the block this slot holds
            gs_raw = *reinterpret_cast<const u32x4*>(smem + gs_q);
            gs_A0 = *reinterpret_cast<const f32x4v*>(cb + cidx * 8); gs_A1 = *reinterpret_cast<const f32x4v*>(cb + cidx * 8 + 4);
            gs_S0 = *reinterpret_cast<const f32x4v*>(cb + 64 + cidx * 8); gs_S1 = *reinterpret_cast<const f32x4v*>(cb + 64 + cidx * 8 + 4);
        } else if constexpr (PH <= 9) {
            constexpr int e = PH - 2;                           // channel e of the slot
            const unsigned wraw = gs_raw[e / 2];                // (element copy first: __builtin_bit_cast applied directly to an ext-vector element reads element 0)
            const float xv = (e & 1) ? dm_elem_hi(wraw) : dm_elem_lo(wraw);
            float z = xv * (e < 4 ? gs_A0[e & 3] : gs_A1[e & 3]) + (e < 4 ? gs_S0[e & 3] : gs_S1[e & 3]);
            // SiLU on the hardware's 2^x and 1/x, as csrc/groupnorm.hip (a select, not a branch: the phase stays in its MFMA's block)
            const float sg = z * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
            gs_z[e] = a.gn_act ? sg : z;
        } else if constexpr (PH == 10) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 two = {gs_z[e], gs_z[e + 1]};
                const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(two, elem2));
                const unsigned old_w = gs_raw[e / 2];
                gs_o[e / 2] = gs_ok ? pk : old_w;               // out-of-image pixels stay zero (the convolution pads the normalised tensor)
            }
        } else {
            *reinterpret_cast<u32x4*>(smem + gs_q) = gs_o;
        }
    };
    [[maybe_unused]] auto gn_transform = [&](bool live, int Y0, int X0, int j, char* buf, const float* cb) __attribute__((always_inline)) {
        static_for_c<0, kGnPhases>([&](auto ph) __attribute__((always_inline)) { gn_phase(ph, live, Y0, X0, j, buf, cb); });
    };

    f32x16 acc[MT][NT];
    elem8 a0[MT], b0[NT], a1[MT], b1[NT];
    // this lane's rows of the tile: r = TM wm + 32 i + l31 -> (ty, tx) = (r >> 4, r & 15)
    int a4[MT][4];                                          // LDS byte offsets of the fragment reads at (dy = 0, current dx, current patch buffer)
    auto set_a4 = [&](int dx, int buf_off) __attribute__((always_inline)) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));                        // (opaque copy: keeps the lane constants out of the K loop's live set)
        const int l31_ = t_ & 31, hi_ = (t_ >> 5) & 1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = TM * wm + 32 * i + l31_;
            const int ty = r >> 4, c = (r & 15) + dx;
            const int base = buf_off + (ty * PW + c) * 128;
            const int sw = (c >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a4[i][kk] = base + (((2 * kk + hi_) ^ sw) << 4);
        }
    };
    auto read_frags = [&](int dy, int kk, const char* bb, elem8 (&af)[MT], elem8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const elem8*>(smem + a4[i][kk] + dy * PW * 128);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = TN * wn + 32 * j + l31;
            bf[j] = *reinterpret_cast<const elem8*>(bb + r * 128 + (((2 * kk + hi) ^ ((r >> 1) & 7)) << 4));
        }
    };
    auto mma = [&](const elem8 (&af)[MT], const elem8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = DM_MFMA_32x32x16(bf[j], af[i], acc[i][j]);
    };

    // ---- prologue: the first block's patch and the first step's weights
    {
        int b_, Y_, X_, n_;
        coords(first, b_, Y_, X_, n_);
#pragma unroll
        for (int j = 0; j < 9; ++j) issue_a(b_, Y_, X_, 0, j, smem);
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q) issue_b(n_, 0, 0, q, bst0);
        if constexpr (GN) {
            if (wave == NW - 1) issue_coef(b_, 0, cbuf0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // the coefficients (one wave's request) are visible to all
#pragma unroll
            for (int j = 0; j < 9; ++j) gn_transform(true, Y_, X_, j, smem, cbuf0);
        }
    }
    sidx_abl = 1;
    int sidx = 0;                                           // K-steps taken so far (weight stage = sidx % NB)
    int blk = 0;                                            // channel blocks taken so far (patch buffer = blk & 1)
    bool after_epilogue = false;
    for (int ct = first; ct < xend; ct += wpx) {
        int cb, Y0, X0, n0;
        coords(ct, cb, Y0, X0, n0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // the MFMAs of a step's last chunk run at the top of the NEXT step (behind its first fragment reads, with the patch piece's
        // address arithmetic woven between them); a tile's first step multiplies zero fragments there: 6-8 idle MFMAs per tile, no branch
#pragma unroll
        for (int i = 0; i < MT; ++i) a1[i] = elem8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NT; ++j) b1[j] = elem8{0, 0, 0, 0, 0, 0, 0, 0};
        for (int kc = 0; kc < n_kc; ++kc, ++blk) {
            // the block after this one (its patch is requested during this block's nine steps), if any
            const bool last_kc = kc + 1 == n_kc;
            const int nid = last_kc ? ct + wpx : ct, nkc = last_kc ? 0 : kc + 1;
            const bool has_next = nid < xend;
            int nb_ = cb, nY0 = Y0, nX0 = X0, nn0 = n0;              // (the next block's patch: another tile's only after this tile's last block)
            if (last_kc && has_next) coords(nid, nb_, nY0, nX0, nn0);
            const int cur_off = (blk & 1) * A_BYTES;
            char* const nbuf = smem + ((blk + 1) & 1) * A_BYTES;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                set_a4(dx, cur_off);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    constexpr int kDummy = 0; (void)kDummy;
                    const int seq = 3 * dx + dy;                    // position of this step (tap 3 dy + dx) in the block
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // every request this step needs is older than the previous tile's NST stores (issued after them: nothing)
                    if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    after_epilogue = false;
#if !defined(DM_ABL_NOBAR)
                    __builtin_amdgcn_s_barrier();
#endif
                    const char* bb = bst0 + (sidx % NB) * B_BYTES;
                    char* const nst = bst0 + ((sidx + 1) % NB) * B_BYTES;
                    // the step after this one: (this block, next tap in dx-major order) or (next block, first tap)
                    const bool step_next = seq < 8 || has_next;
                    const int s_n0 = seq < 8 ? n0 : nn0, s_kc = seq < 8 ? kc : nkc;
                    const int s_tap = seq < 8 ? (seq + 1 < 9 ? 3 * ((seq + 1) % 3) + (seq + 1) / 3 : 0) : 0;
                    read_frags(dy, 0, bb, a0, b0);
                    // chunk 3 of the previous step, and behind its first MFMA the address arithmetic of this step's patch piece (~35
                    // vector instructions that otherwise stand between the barrier and the first MFMA, every step); the request itself
                    // right behind the second -- early in the step: it is this wave's only request that comes from HBM (every patch
                    // pixel is requested once), and the next step's vmcnt(0), which is there for the weights requested behind it, also
                    // waits for it: requested in the last chunk it had ~300 cycles of its ~1000 and every step stalled (-21 % measured)
                    unsigned a_off_now = OOB;
                    static_for_c<0, MT * NT>([&](auto k_tag) __attribute__((always_inline)) {
                        constexpr int k = decltype(k_tag)::value;
                        acc[k / NT][k % NT] = DM_MFMA_32x32x16(b1[k % NT], a1[k / NT], acc[k / NT][k % NT]);
                        if constexpr (k == 0) a_off_now = a_offset(nb_, nY0, nX0, nkc, seq);
                        if constexpr (k == 1) {
                            if (has_next) issue_a_at(a_off_now, seq, nbuf);
                            if constexpr (GN) {
                                if (has_next && seq == 0 && wave == NW - 1) issue_coef(nb_, nkc, cbuf0 + 128 * ((blk + 1) & 1));
                            }
                        }
                        if constexpr (k <= 1) __builtin_amdgcn_sched_barrier(0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    // chunks 1-3: fragment reads of the next chunk, the weight requests, the chunk's MFMAs -- and (GN) behind each MFMA one
                    // phase of the transform of the patch piece this wave requested a step ago (it has landed: this step's vmcnt(0); its
                    // coefficients were requested at the block's first step and are visible since this step's barrier): ~8 vector
                    // instructions per MFMA shadow.  As one block in front of a chunk's MFMAs the ~100 instructions held the matrix pipe
                    // idle for their whole length, every step (conv + GroupNorm 713 us where the plain patch kernel took 605)
                    [[maybe_unused]] const bool gn_live = has_next && seq >= 1;
                    [[maybe_unused]] const int gn_j = seq >= 1 ? seq - 1 : 0;
                    [[maybe_unused]] const float* gn_cb = cbuf0 + 128 * ((blk + 1) & 1);
                    auto chunk_mma = [&](auto c_tag, const elem8 (&af)[MT], const elem8 (&bf)[NT]) __attribute__((always_inline)) {
                        constexpr int C = decltype(c_tag)::value;       // 0, 1, 2 = chunks 1, 2, 3 of the step
                        if constexpr (GN) {
                            static_for_c<0, MT * NT>([&](auto k_tag) __attribute__((always_inline)) {
                                constexpr int k = decltype(k_tag)::value;
                                acc[k / NT][k % NT] = DM_MFMA_32x32x16(bf[k % NT], af[k / NT], acc[k / NT][k % NT]);
                                // phase schedule: the twelve phases spread evenly over the 3 MT NT MFMAs of chunks 1-3 (phase p behind MFMA
                                // ceil(p * total / 12): every MFMA at 2 x 2 fragments, every second one at 4 x 2)
                                constexpr int total = 3 * MT * NT, slot = C * MT * NT + k;
                                constexpr int ph = gn_phase_of_slot(slot, total, kGnPhases);
                                if constexpr (ph >= 0) gn_phase(std::integral_constant<int, ph>{}, gn_live, nY0, nX0, gn_j, nbuf, gn_cb);
                                __builtin_amdgcn_sched_barrier(0);
                            });
                        } else {
                            mma(af, bf);
                        }
                    };
                    static_assert(!GN || 3 * MT * NT >= kGnPhases, "enough MFMAs in chunks 1-3 for the transform's phases");
                    read_frags(dy, 1, bb, a1, b1);
                    if (step_next) {
#pragma unroll
                        for (int q = 0; q < (B_INSTR + 1) / 2; ++q) issue_b(s_n0, s_kc, s_tap, q, nst);
                    }
                    chunk_mma(std::integral_constant<int, 0>{}, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(dy, 2, bb, a0, b0);
                    if (step_next) {
#pragma unroll
                        for (int q = (B_INSTR + 1) / 2; q < B_INSTR; ++q) issue_b(s_n0, s_kc, s_tap, q, nst);
                    }
                    chunk_mma(std::integral_constant<int, 1>{}, a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(dy, 3, bb, a1, b1);
                    chunk_mma(std::integral_constant<int, 2>{}, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    ++sidx;
                }
            }
        }
        mma(a1, b1);                                        // the last step's last chunk

        // ---- epilogue (the plain one of k_conv3x3_dma): y = acc + bias[n] (+ rowbias[image, n]) (+ res[m, n]), one rounding
        unsigned poff[MT];
        int te_ = tid;
        asm volatile("" : "+v"(te_));
        const int l31e = te_ & 31, hie = (te_ >> 5) & 1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int d = TM * wm + 32 * i + l31e;
            const int y = Y0 + (d >> 4), x = X0 + (d & 15);
            poff[i] = (y < H && x < W) ? (((unsigned)cb * (unsigned)H + (unsigned)y) * (unsigned)W + (unsigned)x) * (unsigned)a.Cout * 2u : OOB;
        }
        const unsigned rboff = (unsigned)(cb * a.Cout * 2);
        auto unpack_add = [](float (&v)[4], const u32x2 p) __attribute__((always_inline)) {
            v[0] += dm_elem_lo(p[0]); v[1] += dm_elem_hi(p[0]);
            v[2] += dm_elem_lo(p[1]); v[3] += dm_elem_hi(p[1]);
        };
        const bool has_res = a.res != nullptr, has_rb = a.rowbias != nullptr;
        auto frag_load = [&](int i, int j, u32x2 (&r)[8]) __attribute__((always_inline)) {
            const int nbase = n0 + TN * wn + 32 * j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + 8 * g + 4 * hie;
                const bool n_ok = n < a.Cout;
                if (has_rb) r[g] = __builtin_amdgcn_raw_buffer_load_b64(rbrs, (int)(n_ok ? rboff + (unsigned)n * 2u : OOB), 0, 0);
                if (has_res)
                    r[4 + g] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(n_ok && poff[i] != OOB ? poff[i] + (unsigned)n * 2u : OOB), 0, 0);
            }
        };
        u32x2 pre[2][8];
#pragma unroll
        for (int g = 0; g < 8; ++g) pre[0][g] = pre[1][g] = u32x2{0u, 0u};
        static_assert(NT <= 2, "pixel-major epilogue");
        u32x2 bp[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + TN * wn + 32 * j + 8 * g + 4 * hie;
                bp[j][g] = __builtin_amdgcn_raw_buffer_load_b64(brs, (int)(n < a.Cout ? (unsigned)n * 2u : OOB), 0, 0);
            }
        frag_load(0, 0, pre[0]);
#pragma unroll
        for (int f = 0; f < MT * NT; ++f) {
            constexpr int kLast = MT * NT - 1;
            const int i = f / NT, j = f % NT;
            const int nbase = n0 + TN * wn + 32 * j;
            if (f < kLast) frag_load((f + 1) / NT, (f + 1) % NT, pre[(f + 1) & 1]);
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = 2 * gp + q;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                    unpack_add(v, bp[j][g]);
                    unpack_add(v, pre[f & 1][g]);
                    unpack_add(v, pre[f & 1][4 + g]);
                    f32x2 lo = {v[0], v[1]}, hi2 = {v[2], v[3]};
                    elem2 plo = __builtin_convertvector(lo, elem2), phi = __builtin_convertvector(hi2, elem2);
                    w[q][0] = __builtin_bit_cast(unsigned, plo);
                    w[q][1] = __builtin_bit_cast(unsigned, phi);
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                const int c0 = nbase + 16 * gp + 8 * hie;
                const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
                const unsigned off = poff[i] != OOB && c0 < a.Cout ? poff[i] + (unsigned)c0 * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(out, yrs, (int)off, 0, 0);
            }
        }
        after_epilogue = true;
    }
#endif
}

int cu_count();
template <int TH, int BN, int WMW, int NB, bool GN = false>
int launch_conv_halo(const ConvArgs& a, hipStream_t stream) {
    constexpr int PA = ((16 + 2) * (TH + 2) + 7) / 8;
    static_assert(PA <= 64, "no patch piece is requested in a block's last step (GN transforms a piece one step after its request)");
    constexpr int LDS = 2 * PA * 1024 + NB * BN * 128 + (GN ? 2048 : 0);        // (GN: coefficient buffers + the transform's dummy piece)
    static_assert(LDS <= 160 * 1024, "LDS budget");
    if (GN && !a.gn_coef) return DM_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_halo<TH, BN, WMW, NB, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (a.stride != 1 || a.pad_y != 1 || a.pad_x != 1 || a.Hin != a.Hout || a.Win != a.Wout || a.Cin % 64 != 0) return DM_ERR_UNSUPPORTED;
    if ((long long)a.B * a.Hin * a.Win * a.Cin * 2 > 0xffffff00LL || (long long)a.Cout * 9 * a.Cin * 2 > 0xffffff00LL || a.M * a.Cout * 2 > 0xffffff00LL)
        return DM_ERR_UNSUPPORTED;                                                    // 32-bit buffer offsets
    const int tiles_x = (a.Wout + 15) / 16, tiles_y = (a.Hout + TH - 1) / TH, n_nt = (a.Cout + BN - 1) / BN;
    const long long total = (long long)a.B * tiles_y * tiles_x * n_nt;
    if (total > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    const int n_cu = cu_count();
    if (n_cu <= 0) return DM_ERR_UNSUPPORTED;
    const long long per_xcd = (total + 7) / 8;
    const long long wpx = std::max<long long>(1, std::min<long long>(per_xcd, n_cu / 8));       // one 512-thread workgroup per CU
    DM_ENTER();
    hipLaunchKernelGGL((k_conv3x3_halo<TH, BN, WMW, NB, GN>), dim3((unsigned)(8 * wpx)), dim3(512), LDS, stream, a, tiles_x, tiles_y, n_nt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DM_OK : (int)e;
}

// split-K reduction: y[m][n] = sum_ks ws[ks][m][n] + bias[n] + rowbias[image(m)][n] + res[m][n], 8 channels per thread
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, int ksplit, long long M, int Cout, long long hw,
                                                        const elem_t* __restrict__ bias, const elem_t* __restrict__ rowbias,
                                                        const elem_t* __restrict__ res, elem_t* __restrict__ y) {
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (idx >= M * Cout) return;
    const long long m = idx / Cout;
    const int n = (int)(idx - m * Cout);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int ks = 0; ks < ksplit; ++ks) {
        const float4* p = reinterpret_cast<const float4*>(ws + (long long)ks * M * Cout + idx);
        const float4 lo = p[0], hi = p[1];
        v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
    auto add8 = [&](const elem_t* src) {
        const uint4 u = *reinterpret_cast<const uint4*>(src);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += dm_elem_lo(w[e]);
            v[2 * e + 1] += dm_elem_hi(w[e]);
        }
    };
    if (bias) add8(bias + n);
    if (rowbias) add8(rowbias + (m / hw) * Cout + n);
    if (res) add8(res + idx);
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2 pr = {v[2 * e], v[2 * e + 1]};
        o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, elem2));
    }
    *reinterpret_cast<uint4*>(y + idx) = make_uint4(o[0], o[1], o[2], o[3]);
}

// grow-only fp32 scratch for the split-K partials.  Owned by the library, used in stream order: launches that need it on
// DIFFERENT streams at the same time would share it (the nets of this repo run on one stream).  First use allocates, so a
// stream capture must be preceded by an eager run of the same shapes (torch's capture warm-up does that).  A buffer that has
// been handed out is NEVER freed: a captured hipGraph (guidance `hip_graph`) keeps replaying kernels that hold its address,
// so growing means allocating a larger one (at least 2x) next to it -- the dead ones add up to less than the live one.
static float* splitk_workspace(size_t floats, hipStream_t stream) {
    static float* buf = nullptr;
    static size_t cap = 0;
    if (floats > cap) {
        // hipMalloc is not capturable: growing under an active stream capture would invalidate the capture (or, worse, succeed
        // on another runtime version and leave the graph with a buffer a later eager call may outgrow).  Refuse instead -- the
        // caller reports DM_ERR_UNSUPPORTED and the message of hipops says to run the shape eagerly once first.
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;
        (void)hipGetLastError();
        size_t got = std::max(floats, 2 * cap);
        float* nb = nullptr;
        if (hipMalloc(&nb, got * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            got = floats;
            if (hipMalloc(&nb, got * sizeof(float)) != hipSuccess) return nullptr;
        }
        buf = nb;
        cap = got;
    }
    return buf;
}

int cu_count() {
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}

template <int BMT, int BN, int NW, int WMW, int NSTAGE, int TAPS = 9, int EPI = 0>
int launch_conv_dma(const ConvArgs& a_in, hipStream_t stream) {
    constexpr int LDS = NSTAGE * (BMT + BN) * 128;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_dma<BMT, BN, NW, WMW, NSTAGE, TAPS, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ConvArgs a = a_in;
    // the 1-tap instantiation is the Linear kernel of dm_gemm (its row set-up addresses x by the output row alone)
    if (TAPS == 1 && (a.stride != 1 || a.pad_y != 0 || a.pad_x != 0 || a.Hin != a.Hout || a.Win != a.Wout || a.rowbias)) return DM_ERR_UNSUPPORTED;
    // patch width: 16 output pixels (or the next power of two >= Wout for narrower maps), height BMT / TW
    int tw_log2 = 4;
    while (tw_log2 > 0 && (1 << (tw_log2 - 1)) >= a.Wout) --tw_log2;
    a.tw_log2 = tw_log2;
    a.tiles_x = (a.Wout + (1 << tw_log2) - 1) >> tw_log2;
    const int TH = BMT >> tw_log2;
    if ((long long)a.B * a.Hout + 512 >= (1 << 22)) return DM_ERR_UNSUPPORTED;      // float-reciprocal row -> image split
    if ((long long)a.B * a.Hin * a.Win * a.Cin * 2 > 0xffffff00LL || (long long)a.Cout * TAPS * a.Cin * 2 > 0xffffff00LL)
        return DM_ERR_UNSUPPORTED;                                                    // 32-bit buffer offsets
    if (a.M * a.Cout * 2 > 0xffffff00LL) return DM_ERR_UNSUPPORTED;
    long long n_mt = (((long long)a.B * a.Hout - a.y_off + TH - 1) / TH) * a.tiles_x;
    if (a.mt_cap > 0) n_mt = std::min(n_mt, a.mt_cap);
    int n_nt = (a.Cout + BN - 1) / BN;
    // persistent grid: as many workgroups as the chip holds at once (LDS- and thread-limited), never more than tiles
    const int n_cu = cu_count();
    if (n_cu <= 0) return DM_ERR_UNSUPPORTED;
    const int wg_per_cu = std::max(1, std::min((160 * 1024) / LDS, 2048 / (NW * 64)));
    // split-K when the output tiles alone would leave most of the chip idle: enough K ranges to fill it, each at least 6
    // K-steps long (DREAMMAT_CONV_SPLITK = 0 disables, = S forces S)
    constexpr bool SPLIT_OK = EPI == 0 && BMT <= 256 && BN <= 128;
    int ksplit = 1;
    if (SPLIT_OK) {
        const long long slots = (long long)n_cu * wg_per_cu, tiles = n_mt * n_nt;
        const int n_steps = TAPS * (a.Cin / 64);
        const char* env = getenv("DREAMMAT_CONV_SPLITK");
        if (env && atoi(env) > 0) ksplit = std::min(atoi(env), n_steps);
        else if (!env && tiles * 2 <= slots) ksplit = (int)std::max<long long>(1, std::min<long long>({slots / tiles, (long long)n_steps / 6, 16LL}));
        if (ksplit > 1 && ((long long)ksplit * a.M * a.Cout * 4 > 0xffffff00LL || a.Cout % 8)) ksplit = 1;
    }
    float* ws = nullptr;
    if (ksplit > 1 && !(ws = splitk_workspace((size_t)ksplit * a.M * a.Cout, stream))) return DM_ERR_UNSUPPORTED;
    const long long total = n_mt * n_nt * ksplit, per_xcd = (total + 7) / 8;
    long long wpx = std::min<long long>(per_xcd, (long long)n_cu * wg_per_cu / 8);
    long long blocks = 8 * std::max<long long>(wpx, 1);
    if (blocks > 0x7fffffffLL || total > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
    // stagger unit = 1/8 of a tile's K loop in s_sleep(127) periods (8128 cycles): per K-step each SIMD runs
    // (NW/4) waves x (BMT/WMW/32)(BN/(NW/WMW)/32) x 4 MFMAs of 32 cycles, at ~2/3 of the matrix pipe
    int stagger = 0;
    // OFF by default: measured on 8 x 128->128 @512^2 it removes the post-epilogue drain (tile 85 k -> 75 k cycles) but the
    // late starters finish 7/8 of a tile later, a wash at 16 tiles per workgroup (0.79 vs 0.81 ms); it needs a dynamic
    // tile queue to pay.  DREAMMAT_CONV_STAGGER=1 enables it for experiments.
    static const bool use_stagger = getenv("DREAMMAT_CONV_STAGGER") && !strcmp(getenv("DREAMMAT_CONV_STAGGER"), "1");
    if (use_stagger && per_xcd >= 3 * wpx) {
        const long long step_cycles = (long long)(NW / 4) * (BMT / WMW / 32) * (BN / (NW / WMW) / 32) * 4 * 32 * 3 / 2;
        stagger = (int)std::max<long long>(1, (long long)TAPS * (a.Cin / 64) * step_cycles / 8 / 8128);
    }
    DM_ENTER();
    static const bool timeline = getenv("DREAMMAT_CONV_TIMELINE") != nullptr;      // development aid, see tools/conv_fit.sh
    static unsigned long long* tl_buf = nullptr;
    a.timeline = nullptr;
    if (timeline) {
        if (!tl_buf && hipMalloc(&tl_buf, 4096 * 64 * 8) != hipSuccess) return DM_ERR_UNSUPPORTED;
        if (blocks <= 4096) { (void)hipMemsetAsync(tl_buf, 0, 4096 * 64 * 8, stream); a.timeline = tl_buf; }
        a.timeline_steps = atoi(getenv("DREAMMAT_CONV_TIMELINE")) >= 2 ? atoi(getenv("DREAMMAT_CONV_TIMELINE")) : 0;
    }
    hipLaunchKernelGGL((k_conv3x3_dma<BMT, BN, NW, WMW, NSTAGE, TAPS, EPI>), dim3((unsigned)blocks), dim3(NW * 64), LDS, stream, a,
                       n_mt, n_nt, stagger, ksplit, ws);
    hipError_t e = hipGetLastError();
    if (ksplit > 1 && e == hipSuccess) {
        const long long n8 = a.M * a.Cout / 8;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, ws, ksplit, a.M, a.Cout,
                           (long long)a.Hout * a.Wout, a.bias, a.rowbias, a.res, a.y);
        e = hipGetLastError();
    }
    if (a.timeline && e == hipSuccess) {
        // per-tile phases of workgroups 0 and 8 (s_memtime ticks): K loop | epilogue | gap to the next tile's first stamp
        static unsigned long long host[4096 * 64];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(host, tl_buf, sizeof(host), hipMemcpyDeviceToHost);
        for (int b : {0, 8}) {
            fprintf(stderr, "[conv timeline] wg %d:", b);
            const unsigned long long* t = host + (size_t)b * 64;
            int n = 0;
            while (n < 64 && t[n]) ++n;
            if (a.timeline_steps == 4) {
                fprintf(stderr, " per K-step, s_memtime ticks\n");
                for (int w = 0; w < NW; ++w) {
                    const unsigned long long* u = t + 8 + 5 * w;
                    const double ns = (double)std::max<unsigned long long>(1, u[4]);
                    fprintf(stderr, "    wave %d over %llu steps: body %.0f | LDS wait %.0f | DMA wait %.0f | barrier %.0f\n", w, u[4],
                            u[0] / ns, u[1] / ns, u[2] / ns, u[3] / ns);
                }
                continue;
            } else if (a.timeline_steps) {
                for (int i = 1; i < n; ++i) fprintf(stderr, " %llu", t[i] - t[i - 1]);
            } else {
                for (int i = 0; i + 2 < n; i += 3) fprintf(stderr, " K=%llu E=%llu", t[i + 1] - t[i], t[i + 2] - t[i + 1]);
            }
            fprintf(stderr, "  total=%llu\n", n ? t[n - 1] - t[0] : 0ULL);
        }
    }
    return e == hipSuccess ? DM_OK : (int)e;
}

// Cout = 320 with 256-row tiles is ONE N tile and one workgroup per CU: the SD-2.1 layers at 64 x 64 x 24 images are 384 M tiles,
// 1.5 rounds of 256 CUs -- half the chip idles through the second round (0.206 ms where 1.5 balanced rounds are 0.155).  When the
// last round would be at most 60 % full, the rows of that round go to a second launch with 128-row tiles (twice as many, half
// as long): 1 + ~0.55 rounds.  (128-row tiles for everything is only 3 % faster than 256-row tiles: the narrow wave tile re-reads
// its weights twice as often.)
template <int TAPS>
int launch_320_balanced(const ConvArgs& a_in, hipStream_t stream) {
    ConvArgs a = a_in;
    a.y_off = 0; a.mt_cap = 0;
    const int n_cu = cu_count();
    int tw_log2 = 4;
    while (tw_log2 > 0 && (1 << (tw_log2 - 1)) >= a.Wout) --tw_log2;
    const int tiles_x = (a.Wout + (1 << tw_log2) - 1) >> tw_log2, TH = 256 >> tw_log2;
    const long long rows = (long long)a.B * a.Hout, bands = (rows + TH - 1) / TH, n_mt = bands * tiles_x;
    static const bool off = getenv("DREAMMAT_CONV_BALANCE") && !strcmp(getenv("DREAMMAT_CONV_BALANCE"), "0");
    if (!off && n_cu > 0 && a.Cout <= 320 && n_mt > n_cu && n_mt % n_cu != 0 && (n_mt % n_cu) * 5 <= (long long)n_cu * 3) {
        const long long full_bands = (n_mt / n_cu) * n_cu / tiles_x;          // whole row bands inside the full rounds
        if (full_bands > 0 && full_bands < bands) {
            a.mt_cap = full_bands * tiles_x;
            int rc = launch_conv_dma<256, 320, 8, 4, 2, TAPS, 0>(a, stream);
            if (rc != DM_OK) return rc;
            a.mt_cap = 0;
            a.y_off = (int)(full_bands * TH);
            return launch_conv_dma<128, 320, 8, 4, 2, TAPS, 0>(a, stream);
        }
    }
    return launch_conv_dma<256, 320, 8, 4, 2, TAPS, 0>(a, stream);
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

// Tile variant of a 3 x 3 LDS-DMA launch (DREAMMAT_CONV_TILE=128|256|512|320|640 forces one: tests / A-B measurements).
int conv_tile_choice(long long M, int Cout) {
    const char* tile_env = getenv("DREAMMAT_CONV_TILE");   // read per call: tests toggle it
    int tile = tile_env ? atoi(tile_env) : 0;
    auto n_wg = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
    if (!tile) {
        if (!(Cout >= 128 && M >= 2048)) tile = 128;
        else if (Cout % 256 == 0 && n_wg(256, 256) >= 200) tile = 512;
        else if ((Cout == 128 || Cout == 640) && n_wg(512, 128) >= 200) tile = 640;
        else if (Cout % 320 == 0 && Cout % 256 != 0 && n_wg(256, 320) >= 160) tile = 320;
        // Cout = 320 at batch 3-6 (1 view per rank): 256-row tiles leave the chip half empty (144 tiles), 128x64 tiles fit two
        // workgroups per CU and all 480 run at once: 47 -> 41 us, 93 -> 79 us (tools/conv_b3.sh)
        else if (Cout % 128 != 0 && n_wg(256, 128) < 400) tile = 128;
        else tile = 256;
    }
    return tile;
}

// Which halo-patch variant (k_conv3x3_halo) serves this launch: 24 = 384 x 128 patches (the Cout = 128 layers that took the
// 512 x 128 tile), 16 = 256 x 256 patches (where the 256 x 256 tile ran), 128 = 256 x 128 patches with three weight stages (where the
// 256 x 128 tile ran on whole channel tiles and whole 16 x 16 patches), 0 = none (the per-tap kernels).  Stride 1, pad 1,
// Cin % 64 == 0, enough patches to fill the chip.  DREAMMAT_CONV_HALO=0: off (A/B runs; a forced tile variant also turns it off:
// the tile-variant tests compare the per-tap kernels bit for bit); =24 | 16 forces a variant whatever the size (tests: ragged
// bands / columns, tiny images).
int conv_halo_choice(const ConvArgs& a) {
    if (a.stride != 1 || a.pad_y != 1 || a.pad_x != 1 || a.Hin != a.Hout || a.Win != a.Wout || a.Cin % 64 != 0 || a.Cout % 64 != 0) return 0;
    if (getenv("DREAMMAT_CONV_TILE")) return 0;
    const char* halo_env = getenv("DREAMMAT_CONV_HALO");
    if (halo_env && halo_env[0] == '0') return 0;
    if (halo_env && !strcmp(halo_env, "24")) return 24;
    if (halo_env && !strcmp(halo_env, "16")) return 16;
    if (halo_env && !strcmp(halo_env, "128")) return 128;
    const int tile = conv_tile_choice(a.M, a.Cout);
    if (tile == 640 && a.Cout == 128) return 24;      // (the Cout = 640 layers at 32 x 32 lose a third to the ragged second band and re-request the patch per channel tile: 800 vs 1100 TF/s)
    if (tile == 512) return 16;
    // 256 x 128 patches, three weight stages, where the 256 x 128 tile ran with whole 128-wide channel tiles (the 1280-channel layers
    // at 16 x 16: one patch per image): 24 x 1280->1280 909 -> 1097 TF/s, 2560->1280 826 -> 1012; not the 640-channel layers at 32 x 32
    // (-2 %), not ragged channel tiles (320: -15 %)
    // -- and only with enough (patch, channel tile) items to fill the chip: there is no split-K here, and 3 images (one view per rank
    // of an 8-GPU job) are 30 items for 256 CUs: 84 us against 42 us on the per-tap kernel with its K ranges
    if (tile == 256 && a.Cout % 128 == 0 && a.Hout % 16 == 0 && a.Wout % 16 == 0 &&
        (long long)a.B * (a.Hout / 16) * (a.Wout / 16) * (a.Cout / 128) * 4 >= 3LL * cu_count()) return 128;
    return 0;
}

}  // namespace

extern "C" {

#if !defined(DM_F16)      // (dtype-independent: exported once)
// 1 when dm_conv3x3_gn_nhwc_*_fused serves this stride-1 pad-1 problem (the halo-patch kernel takes it), else 0: the caller then runs
// the GroupNorm apply pass and the plain convolution.
int dm_conv3x3_gn_ok(int B, int H, int W, int Cin, int Cout) {
    ConvArgs a = {};
    a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = Cin; a.Cout = Cout; a.stride = 1; a.pad_y = a.pad_x = 1;
    a.M = (long long)B * H * W;
    return B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && conv_halo_choice(a) != 0;
}
#endif

// conv3x3(act(GroupNorm(x))) + bias (+ rowbias) (+ residual) with the GroupNorm APPLY pass folded into the convolution (round 6):
// x [B,H,W,Cin] is the un-normalised activation, gn_coef the fp32 workspace dm_groupnorm_nhwc_stats left for it ([B][7][Cin]; rows
// 0 and 1 are read), gn_act 0 | 1 (SiLU).  Stride 1, pad 1.  Same results as dm_groupnorm_nhwc_fwd + dm_conv3x3_nhwc_*_fused up to the
// summation order of the taps.  DM_ERR_UNSUPPORTED when dm_conv3x3_gn_ok says 0.
int DM_T(dm_conv3x3_gn_nhwc_, _fused)(const void* x, const float* gn_coef, int gn_act, const void* w, const void* bias, const void* rowbias,
                                  const void* residual, void* y, int B, int H, int W, int Cin, int Cout, hipStream_t stream) {
    if (!x || !gn_coef || !w || !y || B <= 0 || H <= 0 || W <= 0) return DM_ERR_ARG;
    if (Cin % 64 != 0 || Cout % 64 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)gn_coef) & 15) return DM_ERR_ARG;
    if (((uintptr_t)bias | (uintptr_t)rowbias | (uintptr_t)residual) & 7) return DM_ERR_ARG;
    ConvArgs a = {};
    a.x = (const elem_t*)x; a.w = (const elem_t*)w; a.bias = (const elem_t*)bias; a.y = (elem_t*)y;
    a.rowbias = (const elem_t*)rowbias; a.res = (const elem_t*)residual; a.timeline = nullptr; a.timeline_steps = 0;
    a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = Cin; a.Cout = Cout;
    a.stride = 1; a.pad_y = 1; a.pad_x = 1;
    a.M = (long long)B * H * W;
    a.gn_coef = gn_coef; a.gn_act = gn_act;
    switch (conv_halo_choice(a)) {
    case 24: return launch_conv_halo<24, 128, 4, 2, true>(a, stream);
    case 16: return launch_conv_halo<16, 256, 2, 2, true>(a, stream);
    case 128: return launch_conv_halo<16, 128, 4, 3, true>(a, stream);
    default: return DM_ERR_UNSUPPORTED;
    }
}

// x [B,Hin,Win,Cin] NHWC bf16; w [Cout,3,3,Cin] (= [Cout, 9*Cin], tap-major) bf16; bias [Cout] bf16 or NULL;
// y [B,Hout,Wout,Cout] NHWC bf16 with Hout = (Hin + pad_y + pad_y_end - 3)/stride + 1 chosen by the caller
// (pad_y / pad_x are the leading pads; trailing pads are implied by Hout/Wout and zero-filled).
// rowbias [B,Cout] / residual [B,Hout,Wout,Cout] (bf16, either may be NULL) are added in the epilogue
// (LDS-DMA kernels only: Cin % 64 == 0).
int DM_T(dm_conv3x3_nhwc_, _fused)(const void* x, const void* w, const void* bias, const void* rowbias, const void* residual,
                               void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int stride,
                               int pad_y, int pad_x, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return DM_ERR_ARG;
    if (Cin % 32 != 0 || Cout % 64 != 0) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return DM_ERR_ARG;
    if (((uintptr_t)bias | (uintptr_t)rowbias | (uintptr_t)residual) & 7) return DM_ERR_ARG;
    ConvArgs a = {};
    a.x = (const elem_t*)x; a.w = (const elem_t*)w; a.bias = (const elem_t*)bias; a.y = (elem_t*)y;
    a.rowbias = (const elem_t*)rowbias; a.res = (const elem_t*)residual; a.timeline = nullptr; a.timeline_steps = 0;
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
        auto n_wg = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
        const int tile = conv_tile_choice(a.M, Cout);
        // halo-patch kernel (round 6) where conv_halo_choice says so
        switch (conv_halo_choice(a)) {
        case 24: { int rc = launch_conv_halo<24, 128, 4, 2>(a, stream); if (rc != DM_ERR_UNSUPPORTED) return rc; break; }
        case 16: { int rc = launch_conv_halo<16, 256, 2, 2>(a, stream); if (rc != DM_ERR_UNSUPPORTED) return rc; break; }
        case 128: { int rc = launch_conv_halo<16, 128, 4, 3>(a, stream); if (rc != DM_ERR_UNSUPPORTED) return rc; break; }
        default: break;
        }
        switch (tile) {
        case 640: {                                                          // wave tile 128 x 64, all 160 KB of LDS
            int rc = launch_conv_dma<512, 128, 8, 4, 2>(a, stream);
            if (rc <= 0 || tile_env) return rc;
            return launch_conv_dma<256, 128, 8, 4, 3>(a, stream);            // runtime refused the full-LDS variant
        }
        case 320: return tile_env ? launch_conv_dma<256, 320, 8, 4, 2>(a, stream) : launch_320_balanced<9>(a, stream);   // wave tile 64 x 160
        case 1320: return launch_conv_dma<128, 320, 8, 4, 2>(a, stream);     // wave tile 32 x 160
        // (round 4: ONE wave per SIMD -- <256, 256, 4, 2, 2> / <512, 128, 4, 4, 2>, wave tile 128 x 128, the 256 accumulators in
        // the AGPR half -- is SLOWER twice over.  With the K-step schedule below as it is: 12-20 % (937 -> 816, 785 -> 670 TF/s,
        // profiles/r04_experiments/conv_one_wave_per_simd.txt).  With every fragment read and DMA piece woven behind an MFMA
        // (sched_group_barrier pipeline, branch-free pieces, accumulator reads pinned to their use so nothing spills: the patch
        // and its numbers are profiles/r04_experiments/conv_one_wave_weave.*): still 6-7 % (971 -> 902, 808 -> 755 TF/s).  Its
        // timeline says why: 3330 cycles of body per K-step for 2048 cycles of MFMA + 460 of DMA wait -- a wave is held ~60
        // cycles per LDS-DMA instruction (four waves in lockstep ask the CU's one address unit at once; 80 KB per K-step is
        // 1280 of its cycles whoever asks), and with one wave per SIMD nobody multiplies meanwhile.  The same weave applied to the
        // two-wave kernels is 0-10 % slower than their burst order (conv_two_wave_weave_ab.txt).  What bounds this kernel is
        // the address unit's 64 B/clk against 9 taps re-requesting the same activation rows; the lever left is a halo patch in
        // LDS read at 9 shifted offsets (one DMA of the patch per 64-channel block), a different kernel.)
        case 512: return launch_conv_dma<256, 256, 8, 2, 2>(a, stream);      // wave tile 128 x 64
        case 256: return launch_conv_dma<256, 128, 8, 4, 3>(a, stream);      // wave tile 64 x 64
        default:
            if (Cout % 128 != 0) return launch_conv_dma<128, 64, 4, 2, 3>(a, stream);
            // long K and enough tiles: the 2-stage ring (64 KB, two workgroups per CU) wins (24 x 1280->1280 @8x8 82 -> 70 us,
            // 3 x 640->640 @32x32 48 -> 43 us); short K or a handful of tiles: the 3-stage ring (3 x 1280 @8x8 26 vs 30 us)
            return (Cin >= 320 && n_wg(128, 128) >= 48) ? launch_conv_dma<128, 128, 4, 2, 2>(a, stream)
                                                         : launch_conv_dma<128, 128, 4, 2, 3>(a, stream);
        }
    }
    if (rowbias || residual) return DM_ERR_UNSUPPORTED;
    if (Cout % 128 == 0) return (Cin % 64 == 0) ? launch_conv<128, 64>(a, stream) : launch_conv<128, 32>(a, stream);
    return (Cin % 64 == 0) ? launch_conv<64, 64>(a, stream) : launch_conv<64, 32>(a, stream);
}

int DM_T(dm_conv3x3_nhwc_, )(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin,
                         int Hout, int Wout, int Cout, int stride, int pad_y, int pad_x, hipStream_t stream) {
    return DM_T(dm_conv3x3_nhwc_, _fused)(x, w, bias, nullptr, nullptr, y, B, Hin, Win, Cin, Hout, Wout, Cout, stride, pad_y,
                                      pad_x, stream);
}

// 2 x 2 "convolution": y[b, yo, xo, n] = sum_{dy, dx in {0, 1}} sum_c x[b, yo - pad_y + dy, xo - pad_x + dx, c] w[n][2 dy + dx][c]
// (zero outside the image), x [B,Hin,Win,Cin], w [Cout, 4, Cin], y [B,Hout,Wout,Cout] NHWC bf16; Cin % 64 == 0, Cout % 256 == 0.
// What it is for: the data gradient of a stride-2 3x3 convolution, sub-pixel form.  dx[2u + py, 2v + px] of
// `y = conv3x3(F.pad(x, (0,1,0,1)), stride 2)` (AutoencoderKL's Downsample2D, differentiated at dreammat_guidance.py:284-292) only
// involves g[u - 1 .. u, v - 1 .. v]: 4, 2, 2 and 1 of the nine taps for the four parities.  With the four parities as four blocks of
// output channels (Cout = 4 Cin_x; the caller interleaves the result back to full resolution) this is ONE 2 x 2 convolution at
// the gradient's resolution with pad 1: 16 tap-blocks per gradient pixel, where the zero-inserted form (a 3x3 convolution over a
// 4x larger tensor that is 3/4 zeros: dm_conv3x3_nhwc_bf16 on g_up) runs 36 and needs the zero tensor built first.
// The same machine runs nearest-2x upsampling + 3x3 convolution (diffusers Upsample2D in the UNet's up blocks) without the 4x
// larger tensor: output pixel (2u + py, 2v + px) sees the source through a 2 x 2 window whose weights are SUMS of the 3x3 taps
// that fall on the same source pixel; evaluated with pad 1 on a (h + 1) x (w + 1) grid, parity (py, px) of output (u, v) is
// channel block 2 py + px at grid position (u + py, v + px) (hipops.subpixel_upsample_weights / conv3x3_upsampled_nhwc).
// bias [Cout] bf16 or NULL.
static int conv2x2_launch(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                          int Cout, int pad_y, int pad_x, int shuf_mode, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return DM_ERR_ARG;
    if (Cin % 64 != 0 || Cout % 256 != 0) return DM_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) || ((uintptr_t)bias & 7)) return DM_ERR_ARG;
    if (shuf_mode < 0 || shuf_mode > 2 || (shuf_mode && (Cout / 4) % 16 != 0) || (shuf_mode == 2 && (Hout < 2 || Wout < 2))) return DM_ERR_ARG;
    ConvArgs a = {};
    a.x = (const elem_t*)x; a.w = (const elem_t*)w; a.bias = (const elem_t*)bias; a.y = (elem_t*)y;
    a.rowbias = nullptr; a.res = nullptr; a.timeline = nullptr; a.timeline_steps = 0;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.stride = 1; a.pad_y = pad_y; a.pad_x = pad_x;
    a.M = (long long)B * Hout * Wout;
    a.shuf_mode = shuf_mode; a.shuf_cs = Cout / 4;
    if (shuf_mode) return launch_conv_dma<256, 256, 8, 2, 2, 4, 2>(a, stream);
    return launch_conv_dma<256, 256, 8, 2, 2, 4, 0>(a, stream);
}

int DM_T(dm_conv2x2_nhwc_, )(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                         int Cout, int pad_y, int pad_x, hipStream_t stream) {
    return conv2x2_launch(x, w, bias, y, B, Hin, Win, Cin, Hout, Wout, Cout, pad_y, pad_x, 0, stream);
}

// The same products with the four Cout / 4-channel blocks of an output pixel stored as the four SUB-PIXELS of the 2x finer tensor the
// two callers want (ABI v12; they used to interleave the blocks with a copy of the whole tensor: 0.48 ms of the step for the three
// stride-2 data gradients of the VAE encoder).  mode 1: y is [B, 2 Hout, 2 Wout, Cout / 4], block 2 py + px of pixel (u, v) goes to
// (2u + py, 2v + px) -- the data gradient of a stride-2 convolution.  mode 2: y is [B, 2 (Hout - 1), 2 (Wout - 1), Cout / 4], block
// 2 py + px of grid position (u, v) goes to (2u - py, 2v - px) where that lies inside -- nearest-2x upsampling + 3x3 convolution.
int DM_T(dm_conv2x2_subpixel_nhwc_, )(const void* x, const void* w, const void* bias, void* y, int B, int Hin, int Win, int Cin, int Hout,
                                  int Wout, int Cout, int pad_y, int pad_x, int mode, hipStream_t stream) {
    if (mode != 1 && mode != 2) return DM_ERR_ARG;
    return conv2x2_launch(x, w, bias, y, B, Hin, Win, Cin, Hout, Wout, Cout, pad_y, pad_x, mode, stream);
}

// y[M, N] = x[M, K] w[N, K]^T + bias[N] (+ residual[M, N]); all bf16 row-major, fp32 accumulate, one rounding.
// geglu != 0: w / bias rows are interleaved in blocks of 32 (32 value rows, their 32 gate rows, ...) and
// y[M, N/2] = value * gelu(gate), value and gate kept in fp32 up to the product (ONE rounding, round 5; the unfused Linear -> GEGLU pair
// rounds both to 16 bits first) and gelu = the exact erf form evaluated as gate * Phi(gate) with Abramowitz-Stegun 7.1.26 for erfc
// (|error of Phi| < 3e-7; see the EPI = 1 epilogue above).
// Runs the 1-tap instantiation of the LDS-DMA convolution kernel above (M % 16 == 0, K % 64 == 0, N % 64 == 0; geglu: N % 128).
static int gemm_dispatch(const void* x, const void* w, const void* bias, const void* residual, void* y, long long M, int K, int N, int geglu,
                         int batch, hipStream_t stream) {
    // batch > 0: w = [batch][N][K], x / y rows in `batch` blocks of M / batch (dm_gemm_*_batched)
    ConvArgs a = {};
    a.x = (const elem_t*)x; a.w = (const elem_t*)w; a.bias = (const elem_t*)bias; a.y = (elem_t*)y;
    a.rowbias = nullptr; a.res = (const elem_t*)residual; a.timeline = nullptr; a.timeline_steps = 0;
    a.B = 1; a.Hin = a.Hout = (int)(M / 16); a.Win = a.Wout = 16; a.Cin = K; a.Cout = N;
    a.stride = 1; a.pad_y = 0; a.pad_x = 0;
    a.M = M;
    if (batch > 0) { a.wb_y = (int)(M / batch / 16); a.wb_count = batch; }
    auto n_wg = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    const char* tile_env = getenv("DREAMMAT_GEMM_TILE");     // 128 | 256 | 320 | 512 forces a variant (tests / A-B measurements)
    int tile = tile_env ? atoi(tile_env) : 0;
    if (!tile) {
        if (!(N >= 128 && M >= 2048)) tile = 128;
        else if (N % 256 == 0 && n_wg(256, 256) >= 200) tile = 512;
        else if (!geglu && N == 320 && n_wg(256, 320) >= 160) tile = 320;   // no ragged N tile, x read once (K = 320: 64 -> 59 us, K = 1280: 163 -> 152)
        else tile = 256;
    }
    if (batch > 0 && tile == 320) tile = 256;               // (the balanced two-launch form does not carry the batch offset)
    if (geglu) {
        if (tile == 320) tile = 256;
        switch (tile) {
        case 512: return launch_conv_dma<256, 256, 8, 2, 2, 1, 1>(a, stream);
        case 256: return launch_conv_dma<256, 128, 8, 4, 3, 1, 1>(a, stream);
        default: return launch_conv_dma<128, 128, 4, 2, 2, 1, 1>(a, stream);
        }
    }
    switch (tile) {
    case 512: return launch_conv_dma<256, 256, 8, 2, 2, 1, 0>(a, stream);
    case 320: return tile_env ? launch_conv_dma<256, 320, 8, 4, 2, 1, 0>(a, stream) : launch_320_balanced<1>(a, stream);
    case 1320: return launch_conv_dma<128, 320, 8, 4, 2, 1, 0>(a, stream);
    case 256: return launch_conv_dma<256, 128, 8, 4, 3, 1, 0>(a, stream);
    default:
        // 2-stage ring (64 KB): TWO workgroups per CU, one's epilogue under the other's K loop -- 25 % faster than the 3-stage
        // single-workgroup form on every small-M Linear shape (tools/gemm_fit.sh)
        return (N % 128 == 0) ? launch_conv_dma<128, 128, 4, 2, 2, 1, 0>(a, stream)
                              : launch_conv_dma<128, 64, 4, 2, 3, 1, 0>(a, stream);
    }
}

int DM_T(dm_gemm_, _fused)(const void* x, const void* w, const void* bias, const void* residual, void* y, long long M, int K,
                       int N, int geglu, hipStream_t stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return DM_ERR_ARG;
    if (M % 16 != 0 || K % 64 != 0 || N % 64 != 0 || (geglu && (N % 128 != 0 || residual))) return DM_ERR_UNSUPPORTED;
    if (M / 16 >= (1 << 22)) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return DM_ERR_ARG;
    if (((uintptr_t)bias | (uintptr_t)residual) & 7) return DM_ERR_ARG;
    return gemm_dispatch(x, w, bias, residual, y, M, K, N, geglu, 0, stream);
}

// `batch` independent products in one launch: y[i] = x[i] w[i]^T, x [batch, M, K], w [batch, N, K], y [batch, M, N] (all contiguous,
// 16-bit; no bias / residual).  M % 256 == 0 (no tile of any variant straddles two items), K % 64 == N % 64 == 0.  The per-image
// products of the VAE mid-block attention's GEMM form, several images per launch (hipops._WideHeadAttention).
int DM_T(dm_gemm_, _batched)(const void* x, const void* w, void* y, int batch, long long M, int K, int N, hipStream_t stream) {
    if (!x || !w || !y || batch <= 0 || M <= 0 || K <= 0 || N <= 0) return DM_ERR_ARG;
    if (M % 256 != 0 || K % 64 != 0 || N % 64 != 0) return DM_ERR_UNSUPPORTED;
    if ((long long)batch * M / 16 >= (1 << 22) || (long long)batch * N * K * 2 > 0xffffff00LL) return DM_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return DM_ERR_ARG;
    return gemm_dispatch(x, w, nullptr, nullptr, y, (long long)batch * M, K, N, 0, batch, stream);
}

}  // extern "C"
