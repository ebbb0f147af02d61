// Fused G-buffer -> split-sum PBR shade kernel (forward + backward) for gfx950.
// One thread per covered pixel; all views of the step in ONE launch.  HBM-bound by design:
// algorithmic traffic = n(12) + v(12) + features(20) in, colour(12) out = 56 B / covered pixel
// forward, 76 B backward; LUT / cubemap taps (1-texel face borders so a bilinear footprint never branches)
// are served by L2/MALL.  What bounds the kernel is the number of scattered cache lines per wave the texture
// address unit visits: 16 gather instructions per pixel with RGBA-fp32 texels and plain LUT taps, 8 with 8-byte
// texels (one 16 B load per bilinear row) and the FG x-pair table -- see shade_core.h.
// Reference: threestudio/models/materials/dreammat_material.py:679-711, 746-762.
#include <cstdlib>
#include <type_traits>

#include "shade_core.h"

using namespace dm;

struct Strided {            // element (row i, channel c) at p[i*rs + c*cs]
    const float* p;
    long long rs, cs;
};
struct StridedOut {
    float* p;
    long long rs, cs;
};

struct ShadeArgs {
    EnvAtlas atlas;
    MatCfg mat;
    Strided nrm, view, feat;
    const int* pix_idx;      // [N] global pixel index (b*HW + y*W + x)
    const int* env_of_view;  // [B]
    const int* n_dev;        // device count of rows
    int HW;
    int n_views;             // entries of env_of_view
    int offsets32;           // inputs and outputs are SoA (unit row stride): the fast loop applies
    StridedOut color;        // [N,3]
    // optional debug outputs (null => skipped); rows of 3/3/3/3/1/1 floats, dense [N,C]
    float* albedo; float* spec_light; float* diff_light; float* spec_color; float* diff_color;
    float* metallic; float* roughness;
    // backward
    Strided dcolor;
    StridedOut dfeat;
};

// Both kernels are persistent grid-stride loops with a software prefetch of the 11 input floats (+ pixel index) of later
// pixels (see shade_pixel_loop).  A one-thread-per-pixel launch has all waves streaming inputs, then all computing, then
// all gathering at the same time (measured: runtime ~ sum of the three phases); the prefetch overlaps them.
constexpr int kMaxViewsLds = 256;

struct ShadeIn {
    F3 n, v;
    float f[5];
    F3 dc;
    int pix;
};

template <bool BWD>
__device__ __forceinline__ void shade_load(const ShadeArgs& a, long long i, ShadeIn& in) {
    in.n = f3(a.nrm.p[i * a.nrm.rs], a.nrm.p[i * a.nrm.rs + a.nrm.cs], a.nrm.p[i * a.nrm.rs + 2 * a.nrm.cs]);
    in.v = f3(a.view.p[i * a.view.rs], a.view.p[i * a.view.rs + a.view.cs], a.view.p[i * a.view.rs + 2 * a.view.cs]);
#pragma unroll
    for (int k = 0; k < 5; ++k) in.f[k] = a.feat.p[i * a.feat.rs + k * a.feat.cs];
    if (BWD)
        in.dc = f3(a.dcolor.p[i * a.dcolor.rs], a.dcolor.p[i * a.dcolor.rs + a.dcolor.cs],
                   a.dcolor.p[i * a.dcolor.rs + 2 * a.dcolor.cs]);
    in.pix = a.pix_idx[i];
}
// Pixel loop of both kernels.  8-byte texel formats (production): two-stage form with a prefetch distance of TWO pixels,
// the prefetch issued BETWEEN the gathers of the current pixel and their first use.  vmcnt retires in order, so with the
// round-1 order (prefetch first, gathers second) the wait in front of the first texel decode also waited for the OLDER HBM
// input stream of the next pixel, every iteration: the streaming latency was never hidden (rocprofv3 on the real
// G-buffer: 28 us with an atlas of ONE texel per face, i.e. without any gather divergence at all, against 31 us with the real
// atlas).  Now the only loads older than the gathers were issued a whole iteration earlier.
// `out`: the kernel's per-pixel result rows (forward: colour, 3 channels; backward: d loss / d features, 5 channels); `body`
// receives a store functor (channel, value) for them -- 64-bit pointer arithmetic on the general path, buffer stores with one
// 32-bit lane offset per pixel on the fast path (three v_mad_u64_u32 per store cost more issue slots than the shading of a texel)
template <int FMT, bool BWD, class Body>
__device__ __forceinline__ void shade_pixel_loop(const ShadeArgs& a, const StridedOut& out, Body&& body) {
    const long long N = *a.n_dev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (FMT == kTexelF32 || !a.atlas.fg_pairs || a.n_views > kMaxViewsLds || !a.offsets32 || N < 64) {   // generic: monolithic,
                                                                                        // one-pixel prefetch, any row strides
        if (i >= N) return;
        ShadeIn cur, nxt;
        shade_load<BWD>(a, i, cur);
        for (; i < N; i += stride) {
            const bool more = i + stride < N;
            if (more) shade_load<BWD>(a, i + stride, nxt);
            int env = a.env_of_view[cur.pix / a.HW];
            ShadeCtx c;
            shade_eval_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c);
            const long long ii = i;
            body(i, cur.dc, c, [&](int k, float v) { out.p[ii * out.rs + k * out.cs] = v; });
            if (more) cur = nxt;
        }
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t does not exist in the host pass)
    // per-mip tables and the view -> environment table in LDS: a per-lane index into the kernel-argument copies (or into
    // env_of_view) is a global load whose latency sits in front of every gather of the pixel
    constexpr int NJ = BWD ? 4 : 3;             // 16-byte wave loads per 64-row batch: each covers 4 channels x 16 row quads
    constexpr int NOUT = BWD ? 5 : 3;           // result channels
    __shared__ int s_mip_off[kMaxMips], s_mip_res[kMaxMips], s_env[kMaxViewsLds];
    __shared__ unsigned long long s_chan[16], s_outp[8];      // byte address of every input / output channel row
    __shared__ __attribute__((aligned(16))) unsigned s_x[4][16 * 64];    // one 4 KB transposition buffer per wave (wave-private)
    if (threadIdx.x < kMaxMips) {
        s_mip_off[threadIdx.x] = (int)a.atlas.mip_off[threadIdx.x];
        s_mip_res[threadIdx.x] = a.atlas.mip_res[threadIdx.x];
    }
    if ((int)threadIdx.x < a.n_views) s_env[threadIdx.x] = a.env_of_view[threadIdx.x];
    if (threadIdx.x < 16) {                                   // channel order: nrm 0-2, view 3-5, features 6-10, pixel index 11, d colour 12-14
        const int c = threadIdx.x;
        const void* p = a.pix_idx;
        if (c < 3) p = a.nrm.p + c * a.nrm.cs;
        else if (c < 6) p = a.view.p + (c - 3) * a.view.cs;
        else if (c < 11) p = a.feat.p + (c - 6) * a.feat.cs;
        else if (BWD && c >= 12 && c < 15) p = a.dcolor.p + (c - 12) * a.dcolor.cs;
        s_chan[c] = (unsigned long long)p;
    }
    if (threadIdx.x >= 32 && threadIdx.x < 32 + NOUT) s_outp[threadIdx.x - 32] = (unsigned long long)(out.p + (threadIdx.x - 32) * out.cs);
    __syncthreads();
    // (no thread leaves before this barrier: in a small launch the threads that fill the upper table entries may have no row)
    const float inv_hw = 1.0f / (float)a.HW;
    const int hw_shift = (a.HW & (a.HW - 1)) == 0 ? __builtin_ctz(a.HW) : -1;     // 512^2, 1024^2, ...: a shift
    auto env_of = [&](int pix) {
        int view;
        if (hw_shift >= 0) {
            view = pix >> hw_shift;
        } else {
            view = (int)((float)pix * inv_hw);           // pix < 2^24 is exact in fp32; one step of correction covers the rounding
            view -= (view * a.HW > pix) ? 1 : 0;
            view += ((view + 1) * a.HW <= pix) ? 1 : 0;
        }
        return s_env[view];                              // (n_views <= kMaxViewsLds on this path)
    };
    constexpr int kAll = 0x7ffffffc;
    const __amdgpu_buffer_rsrc_t r_spec = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.spec, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_diff = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.diff, 0, kAll, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_fg = __builtin_amdgcn_make_buffer_rsrc((void*)a.atlas.fg_pairs, 0, kAll, 0x00020000);
    auto rows = [](__amdgpu_buffer_rsrc_t r) {
        return [r](unsigned off) { return __builtin_bit_cast(HalfRowBits, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
    };
    auto fg_rows = [r_fg](unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_fg, (int)off, 0, 0)); };
    // THE KERNEL IS BOUND BY THE NUMBER OF VECTOR-MEMORY INSTRUCTIONS (round 4, tools/gather_probe.cpp): the texture-address
    // unit takes 4 lane addresses per clock whatever the access width -- a streaming wave64 load costs 15.6 (4 B per lane),
    // 19.2 (8 B), 20.7 (16 B) cycles of its CU; a scattered 16 B gather 2.1 cycles per distinct line, 16-21 when the lanes
    // share lines.  Rounds 1-3 issued 13 (backward: 16) dword loads + 8 gathers + 3 (5) dword stores per 64 rows = 24 (29)
    // instructions ~ 480 cycles, measured 486.  Now the SoA input rows are fetched SIXTEEN BYTES PER LANE: lane l of load j
    // reads rows 4 (l & 15) .. + 3 of channel 4 j + (l >> 4), i.e. 3 (4) loads per 64 rows, and a wave-private LDS buffer
    // turns the [channel][row] image into one row per lane (4 ds_write_b128 lane-linear, 12-16 ds_read_b32: the LDS pipe is
    // idle in this kernel); the results leave the same way, one (two) 16-byte stores.  12 (14) vector-memory instructions.
    //
    // WORK DISTRIBUTION (round 4).  Unit = one 64-row batch (one wave, one step).  Rounds 1-3 ran a grid-stride loop over
    // 256 x wg_per_cu workgroups, two pixels per trip: at the bench size (1.2 M rows, 4.4 rows per thread) every thread ran
    // THREE two-pixel trips -- 6 steps of gathers and arithmetic for 4.4 rows of work.  Now:
    //   * the host sizes the grid so that every wave has the SAME whole number of steps (shade_grid below): the fewest waves
    //     that finish in ceil(batches / resident waves) steps;
    //   * a wave that has no batch left leaves (wave-uniform), it does not shade clamped rows;
    //   * XCD x (workgroup b runs on XCD b % 8) owns the x-th eighth of the batches, its waves take them round-robin: the
    //     rows are view-major (either G-buffer order), so an XCD's L2 holds the atlas of one or two environments instead of
    //     all of them (rounds 1-3: every XCD saw every 8th 256-row chunk of every view, 13 MB of atlas through a 4 MB L2:
    //     TCC hit rate 75 %, 1.76x the algorithmic bytes fetched).
    // Two raw register sets: the inputs of step k + 2 are loaded INTO the set whose rows have just issued their gathers
    // (everything the second stage needs lives in ShadeCtx / ShadeTaps by then), so no set is ever copied -- a copy of a
    // freshly requested register is a wait for the whole stream in front of it.  The loop body is branch-free up to the
    // stores: a step whose wave has no batch k + 2 re-reads its own batch (L1 hits) instead of skipping the prefetch, because
    // a conditional load makes the compiler's vmcnt bookkeeping assume the load-free path and wait for "all but the newest
    // 7" at the first texel use -- which on the taken path includes the prefetch just issued.
    const unsigned n32 = (unsigned)N;
    const unsigned NB = (n32 + 63u) >> 6;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned nwx = ((gridDim.x - xcd + 7u) >> 3) * (blockDim.x >> 6);                  // waves on this XCD
    const unsigned b_lo = (unsigned)(((unsigned long long)NB * xcd) >> 3), b_hi = (unsigned)(((unsigned long long)NB * (xcd + 1)) >> 3);
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wx = (blockIdx.x >> 3) * (blockDim.x >> 6) + wave;
    const int cnt = (b_lo + wx < b_hi) ? (int)((b_hi - b_lo - wx + nwx - 1) / nwx) : 0;      // batches of this wave (wave-uniform)
    if (cnt <= 0) return;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned rb0 = (b_lo + wx) << 6, rstep = nwx << 6;                                 // first row of step k: rb0 + k rstep
    const unsigned qlast = (n32 - 1u) & ~3u;                                                 // last row quad that holds a valid row
    typedef unsigned U4 __attribute__((ext_vector_type(4), may_alias, aligned(16)));
    typedef unsigned U1 __attribute__((may_alias));
    struct Raw { U4 r[NJ]; };
    U1* const xw = reinterpret_cast<U1*>(&s_x[wave][0]);
    auto load_raw = [&](unsigned rb, Raw& R) {          // rb = first row of a batch (wave-uniform)
        const unsigned qr = min(rb + 4u * (lane & 15u), qlast);                              // rows past the end: a valid quad instead
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // (an integer turned pointer is a GENERIC pointer: flat_load, both counters; say "global" explicitly)
            const unsigned long long p = s_chan[4 * j + (lane >> 4)] + (unsigned long long)qr * 4u;
            // (measured: non-temporal loads / stores for these once-streamed rows change nothing forward and cost 5-10 % backward
            // in the step -- 26.3 / 30.9 us vs 26.2 / 34.1; not used)
            R.r[j] = *reinterpret_cast<const __attribute__((address_space(1))) U4*>(p);
        }
    };
    // rows past the end (the one ragged batch of a launch) take the LAST VALID row's inputs: every gather address stays inside
    // the atlas; their results are not stored
    auto unpack = [&](const Raw& R, unsigned rb, ShadeIn& in) {      // [channel][row] in registers of 16 B -> one row per lane, through LDS
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<U4*>(xw + (j * 64 + lane) * 4) = R.r[j];
        const unsigned col = min(lane, n32 - 1u - rb);
        auto rd = [&](int c) { return __builtin_bit_cast(float, (unsigned)xw[c * 64 + col]); };
        in.n = f3(rd(0), rd(1), rd(2));
        in.v = f3(rd(3), rd(4), rd(5));
#pragma unroll
        for (int c = 0; c < 5; ++c) in.f[c] = rd(6 + c);
        in.pix = (int)xw[11 * 64 + col];
        if (BWD) in.dc = f3(rd(12), rd(13), rd(14));
    };
    Raw A, B;
    load_raw(rb0, A);
    load_raw(cnt > 1 ? rb0 + rstep : rb0, B);
    auto step = [&](Raw& raw, int k, auto prefetch) {
        const unsigned rb = rb0 + (unsigned)k * rstep, idx = rb + lane;
        ShadeIn cur;
        unpack(raw, rb, cur);
        const int env = env_of(cur.pix);
        ShadeCtx c;
        ShadeTaps t;
        shade_issue_t<FMT>(a.atlas, a.mat, env, cur.n, cur.v, cur.f, c, t, [](int l) { return s_mip_off[l]; },
                           [](int l) { return s_mip_res[l]; }, rows(r_spec), rows(r_diff), fg_rows);
        if constexpr (decltype(prefetch)::value) {
            __builtin_amdgcn_sched_barrier(0);
            load_raw(k + 2 < cnt ? rb + 2u * rstep : rb, raw);
            __builtin_amdgcn_sched_barrier(0);
        }
        shade_finish_t<FMT>(a.atlas, a.mat, t, c);
        float ov[NOUT];
        body(idx < n32 ? (long long)idx : -1, cur.dc, c, [&](int ch, float v) { ov[ch] = v; });
        if (rb + 64u <= n32) {                          // whole batch inside: [row][channel] -> [channel][row quads], 16 B stores
#pragma unroll
            for (int ch = 0; ch < NOUT; ++ch) xw[ch * 64 + lane] = __builtin_bit_cast(unsigned, ov[ch]);
#pragma unroll
            for (int o = 0; o * 64 < NOUT * 16; ++o) {
                const unsigned L = o * 64 + lane;
                if (L < NOUT * 16) {
                    const U4 v = *reinterpret_cast<const U4*>(xw + L * 4);
                    const unsigned long long p = s_outp[L >> 4] + (unsigned long long)(rb + 4u * (L & 15u)) * 4u;
                    *reinterpret_cast<__attribute__((address_space(1))) U4*>(p) = v;
                }
            }
        } else if (idx < n32) {                         // the one ragged batch of a launch
#pragma unroll
            for (int ch = 0; ch < NOUT; ++ch) out.p[(long long)idx + ch * out.cs] = ov[ch];
        }
    };
    int k = 0;
    for (; k + 1 < cnt; k += 2) {
        step(A, k, std::true_type{});
        step(B, k + 1, std::true_type{});
    }
    if (k < cnt) step(A, k, std::false_type{});
#endif
}

template <int FMT, bool DBG>
__global__ __launch_bounds__(256, 4) void k_shade_fwd(ShadeArgs a) {
    shade_pixel_loop<FMT, false>(a, a.color, [&](long long i, F3, const ShadeCtx& c, auto&& st) {
        st(0, sat(c.pre.x));
        st(1, sat(c.pre.y));
        st(2, sat(c.pre.z));
        if (DBG && i >= 0) {
            F3 sl = lin2srgb(c.spec), dl = lin2srgb(c.diff), sc = lin2srgb(c.spec_albedo), dc = lin2srgb(c.albedo);
            a.albedo[3 * i] = c.albedo.x; a.albedo[3 * i + 1] = c.albedo.y; a.albedo[3 * i + 2] = c.albedo.z;
            a.spec_light[3 * i] = sl.x; a.spec_light[3 * i + 1] = sl.y; a.spec_light[3 * i + 2] = sl.z;
            a.diff_light[3 * i] = dl.x; a.diff_light[3 * i + 1] = dl.y; a.diff_light[3 * i + 2] = dl.z;
            a.spec_color[3 * i] = sc.x; a.spec_color[3 * i + 1] = sc.y; a.spec_color[3 * i + 2] = sc.z;
            a.diff_color[3 * i] = dc.x; a.diff_color[3 * i + 1] = dc.y; a.diff_color[3 * i + 2] = dc.z;
            a.metallic[i] = c.metallic;
            a.roughness[i] = c.roughness;
        }
    });
}

template <int FMT>
__global__ __launch_bounds__(256, 3) void k_shade_bwd(ShadeArgs a) {
    shade_pixel_loop<FMT, true>(a, a.dfeat, [&](long long i, F3 dc, const ShadeCtx& c, auto&& st) {
        float df[5];
        shade_backward(a.mat, c, dc, df);
#pragma unroll
        for (int k = 0; k < 5; ++k) st(k, df[k]);
    });
}

// the fast path of shade_pixel_loop: every row tensor SoA (unit row stride; any channel pitch), fewer than 2^30 rows
static inline int shade_offsets32(const ShadeArgs& a, long long n_max, bool bwd) {
    // the fast path moves 16 bytes (four consecutive rows of one channel) per lane: every channel must start on a 16-byte
    // boundary (base aligned, pitch a multiple of 4 floats); anything else takes the generic loop
    auto soa = [&](const void* p, long long rs, long long cs) { return rs == 1 && cs >= 0 && cs % 4 == 0 && ((uintptr_t)p & 15) == 0; };
    return soa(a.nrm.p, a.nrm.rs, a.nrm.cs) && soa(a.view.p, a.view.rs, a.view.cs) && soa(a.feat.p, a.feat.rs, a.feat.cs) &&
           (bwd ? soa(a.dcolor.p, a.dcolor.rs, a.dcolor.cs) && soa(a.dfeat.p, a.dfeat.rs, a.dfeat.cs) : soa(a.color.p, a.color.rs, a.color.cs)) &&
           n_max < 0x3fffffffLL;
}

// Persistent launch, sized so that every wave runs the same whole number of steps (see WORK DISTRIBUTION in
// shade_pixel_loop): S = ceil(batches / resident waves), then the FEWEST waves that still finish in S steps -- per XCD, because
// each XCD owns an eighth of the batches.  Always a multiple of 8 workgroups (one per XCD) unless the launch is tiny.
static inline int shade_grid(long long n_max, int wg_per_cu) {
    if (const char* e = getenv("DREAMMAT_SHADE_WGPCU")) { if (atoi(e) > 0) wg_per_cu = atoi(e); }     // development knob
    const long long nb = (n_max + 63) / 64;                  // 64-row batches
    const long long nbx = (nb + 7) / 8;                      // per XCD (ceil)
    const long long waves_x = 32LL * wg_per_cu * 4;          // resident waves of one XCD: 32 CUs x workgroups x 4 waves
    const long long steps = std::max<long long>(1, (nbx + waves_x - 1) / waves_x);
    const long long need_waves_x = (nbx + steps - 1) / steps;
    const long long wgs_x = std::max<long long>(1, (need_waves_x + 3) / 4);
    return (int)(8 * wgs_x);
}

// Material smoothness regulariser (dreammat_material.py:110-123) fused: forward partial sums and
// analytic gradient wrt both feature sets.  loss = mean(kd_luma*kd_b)*0.25 + mean(ks_0*ks_1)*0.1
__global__ __launch_bounds__(256) void k_matreg(Strided feat, Strided featj, const int* n_dev, float gscale_in,
                                                float* __restrict__ loss_accum /*[1]*/, StridedOut dfeat,
                                                StridedOut dfeatj, int want_grad) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int N = *n_dev;
    float part = 0.f;
    if (i < N) {
        float s[5], sj[5], d[5], sg[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            s[k] = sigmoidf(feat.p[i * feat.rs + k * feat.cs]);
            sj[k] = sigmoidf(featj.p[i * featj.rs + k * featj.cs]);
            float df = s[k] - sj[k];
            d[k] = fabsf(df);
            sg[k] = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
        }
        float luma = (d[0] + d[1] + d[2]) / 3.f;
        float invN = 1.0f / (float)N;
        part = (luma * d[2] * 0.25f + d[3] * d[4] * 0.1f) * invN;
        if (want_grad) {
            float g = gscale_in * invN;
            float gd[5];
            gd[0] = g * 0.25f * d[2] / 3.f;
            gd[1] = g * 0.25f * d[2] / 3.f;
            gd[2] = g * 0.25f * (d[2] / 3.f + luma);
            gd[3] = g * 0.1f * d[4];
            gd[4] = g * 0.1f * d[3];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float gs = gd[k] * sg[k];
                dfeat.p[i * dfeat.rs + k * dfeat.cs] = gs * s[k] * (1.f - s[k]);
                dfeatj.p[i * dfeatj.rs + k * dfeatj.cs] = -gs * sj[k] * (1.f - sj[k]);
            }
        }
    }
    if (!want_grad) {
        for (int ofs = 32; ofs > 0; ofs >>= 1) part += __shfl_xor(part, ofs);
        __shared__ float wsum[4];
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_accum, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    }
}

extern "C" {

// Mirror of EnvAtlas for the C ABI (plain pointers and sizes).
struct dm_env_atlas {
    const float* spec; const float* diff; const float* fg_lut;
    long long spec_env_stride, diff_env_stride;
    long long mip_off[8];
    int mip_res[8];
    int n_mips, diff_res, lut_res;
    float min_rough_mip, max_rough_mip;
    int texel_format;
    const float* fg_pairs;
};
struct dm_mat_cfg { float min_metallic, max_metallic, min_roughness, max_roughness; };

static bool conv_atlas(const dm_env_atlas* in, EnvAtlas& A) {
    if (!in || !in->spec || !in->diff || !in->fg_lut || in->n_mips < 2 || in->n_mips > kMaxMips) return false;
    A.spec = (const float4*)in->spec; A.diff = (const float4*)in->diff; A.fg_lut = (const float2*)in->fg_lut;
    A.spec_env_stride = in->spec_env_stride; A.diff_env_stride = in->diff_env_stride;
    for (int i = 0; i < kMaxMips; ++i) { A.mip_off[i] = in->mip_off[i]; A.mip_res[i] = in->mip_res[i]; }
    A.n_mips = in->n_mips; A.diff_res = in->diff_res; A.lut_res = in->lut_res;
    A.min_rough_mip = in->min_rough_mip; A.max_rough_mip = in->max_rough_mip;
    if (in->texel_format < 0 || in->texel_format > 2) return false;
    A.texel_format = in->texel_format;
    A.fg_pairs = (const float4*)in->fg_pairs;
    return true;
}

// Forward.  Row tensors are addressed as p[i*row_stride + c*col_stride] (elements), so both the
// reference's [N,C] layout and the internal SoA [C,N] layout are accepted without copies.
int dm_shade_fwd(const dm_env_atlas* atlas, const dm_mat_cfg* mat, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, int n_views, float* color, long long color_rs,
                 long long color_cs, float* dbg_albedo, float* dbg_spec_light, float* dbg_diff_light,
                 float* dbg_spec_color, float* dbg_diff_color, float* dbg_metallic, float* dbg_roughness,
                 hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !color || n_max <= 0 || HW <= 0 || n_views <= 0)
        return DM_ERR_ARG;
    int ndbg = (dbg_albedo != 0) + (dbg_spec_light != 0) + (dbg_diff_light != 0) + (dbg_spec_color != 0) +
               (dbg_diff_color != 0) + (dbg_metallic != 0) + (dbg_roughness != 0);
    if (ndbg != 0 && ndbg != 7) return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW; a.n_views = n_views;
    a.color = {color, color_rs, color_cs};
    a.albedo = dbg_albedo; a.spec_light = dbg_spec_light; a.diff_light = dbg_diff_light;
    a.spec_color = dbg_spec_color; a.diff_color = dbg_diff_color; a.metallic = dbg_metallic;
    a.roughness = dbg_roughness;
    DM_ENTER();
    const dim3 grid(shade_grid(n_max, 2));                 // 2 workgroups per CU measured best (tools/r4_shade_probe.py: 1 / 2 / 3 / 4 ->
                                                           // 25.1 / 20.2 / 20.4 / 21.3 us on smooth, 25.7 / 25.3 / 25.6 / 26.9 on noisy features)
    a.offsets32 = shade_offsets32(a, n_max, false);
    const bool dbg = ndbg != 0;
    switch (a.atlas.texel_format) {
        case kTexelRgb18e8:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelRgb18e8, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelRgb18e8, false>), grid, dim3(256), 0, stream, a);
            break;
        case kTexelF16:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelF16, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelF16, false>), grid, dim3(256), 0, stream, a);
            break;
        default:
            if (dbg) hipLaunchKernelGGL((k_shade_fwd<kTexelF32, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_shade_fwd<kTexelF32, false>), grid, dim3(256), 0, stream, a);
            break;
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_shade_bwd(const dm_env_atlas* atlas, const dm_mat_cfg* mat, const float* nrm, long long nrm_rs,
                 long long nrm_cs, const float* view, long long view_rs, long long view_cs, const float* feat,
                 long long feat_rs, long long feat_cs, const int32_t* pix_idx, const int32_t* env_of_view,
                 const int32_t* n_dev, long long n_max, int HW, int n_views, const float* dcolor, long long dcolor_rs,
                 long long dcolor_cs, float* dfeat, long long dfeat_rs, long long dfeat_cs, hipStream_t stream) {
    ShadeArgs a = {};
    if (!conv_atlas(atlas, a.atlas) || !mat || !nrm || !view || !feat || !pix_idx || !env_of_view || !n_dev ||
        !dcolor || !dfeat || n_max <= 0 || HW <= 0 || n_views <= 0)
        return DM_ERR_ARG;
    a.mat = {mat->min_metallic, mat->max_metallic, mat->min_roughness, mat->max_roughness};
    a.nrm = {nrm, nrm_rs, nrm_cs}; a.view = {view, view_rs, view_cs}; a.feat = {feat, feat_rs, feat_cs};
    a.pix_idx = pix_idx; a.env_of_view = env_of_view; a.n_dev = n_dev; a.HW = HW; a.n_views = n_views;
    a.dcolor = {dcolor, dcolor_rs, dcolor_cs};
    a.dfeat = {dfeat, dfeat_rs, dfeat_cs};
    DM_ENTER();
    a.offsets32 = shade_offsets32(a, n_max, true);
    const dim3 grid(shade_grid(n_max, 3));                 // (backward: 32.8 / 25.2 / 24.5 / 26.0 and 32.1 / 27.7 / 29.2 / 30.7 us; in the
                                                           // step itself 3 beat 2: 30.6 vs 33.5 us)
    switch (a.atlas.texel_format) {
        case kTexelRgb18e8: hipLaunchKernelGGL(k_shade_bwd<kTexelRgb18e8>, grid, dim3(256), 0, stream, a); break;
        case kTexelF16: hipLaunchKernelGGL(k_shade_bwd<kTexelF16>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(k_shade_bwd<kTexelF32>, grid, dim3(256), 0, stream, a); break;
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// loss_out (device float) must be zeroed by the caller before the forward call.
int dm_matreg_fwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float* loss_out, hipStream_t stream) {
    if (!feat || !featj || !n_dev || !loss_out || n_max <= 0) return DM_ERR_ARG;
    Strided f = {feat, f_rs, f_cs}, j = {featj, j_rs, j_cs};
    StridedOut z = {nullptr, 0, 0};
    DM_ENTER();
    hipLaunchKernelGGL(k_matreg, dim3(dm_div_up(n_max, 256)), dim3(256), 0, stream, f, j, n_dev, 0.f, loss_out, z, z, 0);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_matreg_bwd(const float* feat, long long f_rs, long long f_cs, const float* featj, long long j_rs,
                  long long j_cs, const int32_t* n_dev, long long n_max, float grad_scale, float* dfeat,
                  long long df_rs, long long df_cs, float* dfeatj, long long dj_rs, long long dj_cs,
                  hipStream_t stream) {
    if (!feat || !featj || !n_dev || !dfeat || !dfeatj || n_max <= 0) return DM_ERR_ARG;
    Strided f = {feat, f_rs, f_cs}, j = {featj, j_rs, j_cs};
    StridedOut df = {dfeat, df_rs, df_cs}, dj = {dfeatj, dj_rs, dj_cs};
    DM_ENTER();
    hipLaunchKernelGGL(k_matreg, dim3(dm_div_up(n_max, 256)), dim3(256), 0, stream, f, j, n_dev, grad_scale, nullptr,
                       df, dj, 1);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
