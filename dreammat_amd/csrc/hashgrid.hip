// Multiresolution hash-grid encoding (forward gather / backward scatter-add) for gfx950.
// Replaces tiny-cuda-nn's `tcnn.Encoding(3, {HashGrid,16,2,2^19,16,1.4473})` used by
// threestudio/models/networks.py:55-64 <- threestudio/models/geometry/dreammat_mesh.py:128-130,250,
// including contract_to_unisphere (geometry/base.py:20-32, bounded branch).
// Layout: table [total_entries][2] fp32 (level-major, tcnn's flat parameter order);
//         output enc addressed as out[m*rs + (2*level+f)*cs] -- the internal path uses the
//         feature-major form (rs=1, cs=M) so both the gather results and the GEMM that consumes
//         them are fully coalesced.
// One thread per (point, level); blockIdx.y = level so a workgroup hammers one level's table
// region (L2 locality).  Backward uses hardware fp32 atomics (-munsafe-fp-atomics).
#include <algorithm>

#include "dm_common.h"

namespace {

constexpr int kMaxLevels = 32;

struct GridLevels {
    float scale[kMaxLevels];
    unsigned res[kMaxLevels];
    unsigned size[kMaxLevels];     // entries in this level
    unsigned offset[kMaxLevels];   // entry offset
    int n_levels;
};

__device__ __forceinline__ unsigned grid_index(unsigned x, unsigned y, unsigned z, unsigned res, unsigned size) {
    // tcnn grid_index<3>: dense while the running stride stays <= size, else coherent prime hash
    unsigned stride = 1, index = 0;
    bool overflow = false;
    if (stride <= size) { index += x * stride; unsigned long long s = (unsigned long long)stride * res; if (s > 0xffffffffull) overflow = true; stride = (unsigned)s; }
    if (!overflow && stride <= size) { index += y * stride; unsigned long long s = (unsigned long long)stride * res; if (s > 0xffffffffull) overflow = true; stride = (unsigned)s; }
    if (!overflow && stride <= size) { index += z * stride; unsigned long long s = (unsigned long long)stride * res; if (s > 0xffffffffull) overflow = true; stride = (unsigned)s; }
    if (overflow || size < stride) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % size;
}

struct HashArgs {
    const float* x; long long x_rs, x_cs;      // points [M,3] world space
    const float2* table;
    float* out; long long out_rs, out_cs;      // enc [M, 2L]
    const float* dout; long long dout_rs, dout_cs;
    float* dtable;
    const int* m_dev;                          // device row count (may be null => m_max)
    long long m_max;
    float inv_2r, radius;                      // contract_to_unisphere: (x + r) / (2r)
    GridLevels lv;
};

template <bool BWD>
__global__ __launch_bounds__(256) void k_hashgrid(HashArgs a) {
    long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long M = a.m_dev ? (long long)*a.m_dev : a.m_max;
    if (m >= M) return;
    const int l = blockIdx.y;
    const float scale = a.lv.scale[l];
    const unsigned res = a.lv.res[l], size = a.lv.size[l], off = a.lv.offset[l];
    float pos[3], w[3];
    unsigned cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float xn = (a.x[m * a.x_rs + d * a.x_cs] + a.radius) * a.inv_2r;
        float p = xn * scale + 0.5f;
        float fl = floorf(p);
        w[d] = p - fl;
        cell[d] = (unsigned)(int)fl;
        pos[d] = p;
    }
    (void)pos;
    if (!BWD) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1), cz = cell[2] + ((c >> 2) & 1);
            float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]) *
                       (((c >> 2) & 1) ? w[2] : 1.f - w[2]);
            float2 t = a.table[off + grid_index(cx, cy, cz, res, size)];
            acc.x += t.x * wt;
            acc.y += t.y * wt;
        }
        a.out[m * a.out_rs + (2 * l) * a.out_cs] = acc.x;
        a.out[m * a.out_rs + (2 * l + 1) * a.out_cs] = acc.y;
    } else {
        float g0 = a.dout[m * a.dout_rs + (2 * l) * a.dout_cs];
        float g1 = a.dout[m * a.dout_rs + (2 * l + 1) * a.dout_cs];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1), cz = cell[2] + ((c >> 2) & 1);
            float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]) *
                       (((c >> 2) & 1) ? w[2] : 1.f - w[2]);
            float* dst = a.dtable + 2 * (size_t)(off + grid_index(cx, cy, cz, res, size));
            atomicAdd(dst, g0 * wt);
            atomicAdd(dst + 1, g1 * wt);
        }
    }
}

// ---- uv-space field (dreammat_mesh.py:128-135, n_input_dims = 2): tcnn's grid over 2-D points.  grid_index<2>: dense
// (x + y * res) while the running stride stays <= size, else (x * 1) ^ (y * 2654435761).  Not on the measured path (no
// shipped config uses it): one thread per (point, level), one atomic pair per corner in the backward.
__device__ __forceinline__ unsigned grid_index2(unsigned x, unsigned y, unsigned res, unsigned size) {
    unsigned stride = 1, index = 0;
    bool overflow = false;
    if (stride <= size) { index += x * stride; unsigned long long s = (unsigned long long)stride * res; if (s > 0xffffffffull) overflow = true; stride = (unsigned)s; }
    if (!overflow && stride <= size) { index += y * stride; unsigned long long s = (unsigned long long)stride * res; if (s > 0xffffffffull) overflow = true; stride = (unsigned)s; }
    if (overflow || size < stride) index = (x * 1u) ^ (y * 2654435761u);
    return index % size;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_hashgrid2d(HashArgs a) {
    long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long M = a.m_dev ? (long long)*a.m_dev : a.m_max;
    if (m >= M) return;
    const int l = blockIdx.y;
    const float scale = a.lv.scale[l];
    const unsigned res = a.lv.res[l], size = a.lv.size[l], off = a.lv.offset[l];
    float w[2];
    unsigned cell[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        float xn = (a.x[m * a.x_rs + d * a.x_cs] + a.radius) * a.inv_2r;
        float p = xn * scale + 0.5f;
        float fl = floorf(p);
        w[d] = p - fl;
        cell[d] = (unsigned)(int)fl;
    }
    float g0 = 0.f, g1 = 0.f;
    if (BWD) { g0 = a.dout[m * a.dout_rs + (2 * l) * a.dout_cs]; g1 = a.dout[m * a.dout_rs + (2 * l + 1) * a.dout_cs]; }
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1);
        const float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]);
        const size_t e = off + grid_index2(cx, cy, res, size);
        if (!BWD) {
            const float2 t = a.table[e];
            acc.x += t.x * wt; acc.y += t.y * wt;
        } else {
            atomicAdd(a.dtable + 2 * e, g0 * wt);
            atomicAdd(a.dtable + 2 * e + 1, g1 * wt);
        }
    }
    if (!BWD) {
        a.out[m * a.out_rs + (2 * l) * a.out_cs] = acc.x;
        a.out[m * a.out_rs + (2 * l + 1) * a.out_cs] = acc.y;
    }
}

// Backward, second formulation.  Device-scope fp32 atomics retire at only ~20 G/s on MI355X (measured,
// profiles/r01_hashgrid_bwd_per_level_v0.json: every level costs ~1.7-6.7 ms regardless of table size),
// so the kernel is organised around ISSUING FEWER OF THEM:
//  * lane pair (2i, 2i+1) = the two features of ONE point, so the pair's atomics hit one 8-byte
//    slot of one cache line (half the line requests per wave-instruction);
//  * DEDUP (coarse / mid levels): consecutive lanes are consecutive covered pixels of an image row, whose
//    samples fall into the same grid cell for long runs; a segmented wave scan sums each run and only
//    the run's last lane issues the atomic (level 0: ~30x fewer atomics).
template <bool DEDUP>
__global__ __launch_bounds__(256) void k_hashgrid_bwd2(HashArgs a, int level_base) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = a.m_dev ? (long long)*a.m_dev : a.m_max;
    const long long m = gid >> 1;
    const int f = (int)(gid & 1);
    const int lane = threadIdx.x & 63;
    const bool valid = m < M;
    const int l = level_base + blockIdx.y;
    const float scale = a.lv.scale[l];
    const unsigned res = a.lv.res[l], size = a.lv.size[l], off = a.lv.offset[l];
    float w[3];
    unsigned cell[3];
    const long long mc = valid ? m : 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float xn = (a.x[mc * a.x_rs + d * a.x_cs] + a.radius) * a.inv_2r;
        float p = xn * scale + 0.5f;
        float fl = floorf(p);
        w[d] = p - fl;
        cell[d] = (unsigned)(int)fl;
    }
    const float g = valid ? a.dout[mc * a.dout_rs + (2 * l + f) * a.dout_cs] : 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1), cz = cell[2] + ((c >> 2) & 1);
        float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]) *
                   (((c >> 2) & 1) ? w[2] : 1.f - w[2]);
        unsigned idx = valid ? grid_index(cx, cy, cz, res, size) : 0xffffffffu;
        float v = g * wt;
        if (DEDUP) {
            // run start of this lane's key among same-feature lanes (stride 2)
            unsigned prev = (unsigned)__shfl_up((int)idx, 2);
            int start = (lane < 2 || prev != idx) ? lane : 0;
#pragma unroll
            for (int ofs = 2; ofs < 64; ofs <<= 1) {
                int s2 = __shfl_up(start, ofs);
                if (lane >= ofs) start = max(start, s2);
            }
#pragma unroll
            for (int ofs = 2; ofs < 64; ofs <<= 1) {
                float v2 = __shfl_up(v, ofs);
                if (lane - ofs >= start) v += v2;
            }
            unsigned nxt = (unsigned)__shfl_down((int)idx, 2);
            bool tail = (lane >= 62) || (nxt != idx);
            if (tail && valid) atomicAdd(a.dtable + 2 * (size_t)(off + idx) + f, v);
        } else {
            if (valid) atomicAdd(a.dtable + 2 * (size_t)(off + idx) + f, v);
        }
    }
}

// Backward, coarse levels (cells several pixels wide): each workgroup accumulates its 1024 consecutive points --
// a strip a few image rows tall -- in an LDS hash table keyed by the level-local entry index and only then
// issues ONE global atomic pair per distinct entry.  The run-combining variant above merges along an image row
// only; the strip also shares entries between rows.  Open addressing, linear probing; a point that cannot find a
// slot within LH_PROBES falls back to the direct global atomic, so a full table costs time, never correctness.
// The values are 64-bit FIXED POINT (round 6, as the binned route below since round 3): ds_add_f32 is a slow path on this chip --
// 99 G tuple pairs/s chip-wide against 855 G/s for ds_add_u64 (tools/lds_atomic_probe.cpp) -- and this kernel issued 48 M of them
// per step.  Scale = 2^44 / the workgroup's own largest |dL/denc| (a first pass over its 1024 points; the table is per workgroup):
// 8192 corner updates of magnitude <= max cannot wrap, the sums resolve 2^-44 of the largest gradient and no longer depend on
// the order of the updates inside the workgroup.
constexpr int LH_SLOTS = 2048;                 // 8 KB keys + 32 KB values => 4 workgroups per CU
constexpr int LH_THREADS = 256;
constexpr int LH_PTS = 4;                      // points per thread (1024 consecutive points per workgroup)
constexpr int LH_PROBES = 8;
constexpr unsigned LH_EMPTY = 0xffffffffu;
constexpr unsigned LH_INVALID = 0xfffffffeu;   // key of lanes past the end of the point list (never inserted)

// round-to-nearest double -> int64 for |d| < 2^51 (two plain fp64 / integer ops)
__device__ __forceinline__ long long lh_fixed(double d) {
    const double magic = 6755399441055744.0;               // 1.5 * 2^52
    return __double_as_longlong(d + magic) - __double_as_longlong(magic);
}

__global__ __launch_bounds__(LH_THREADS) void k_hashgrid_bwd_lds(HashArgs a, int level_base) {
    __shared__ unsigned keys[LH_SLOTS];
    __shared__ unsigned long long vals[2 * LH_SLOTS];
    __shared__ unsigned wg_max;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LH_SLOTS; i += LH_THREADS) { keys[i] = LH_EMPTY; vals[2 * i] = 0ull; vals[2 * i + 1] = 0ull; }
    if (tid == 0) wg_max = 0u;
    __syncthreads();
    const long long M = a.m_dev ? (long long)*a.m_dev : a.m_max;
    const int l = level_base + blockIdx.y;
    const float scale = a.lv.scale[l];
    const unsigned res = a.lv.res[l], size = a.lv.size[l], off = a.lv.offset[l];
    {   // the workgroup's largest |gradient| of this level (NaN / Inf are left out: their sums are garbage either way)
        float gm = 0.f;
#pragma unroll
        for (int q = 0; q < LH_PTS; ++q) {
            const long long m = ((long long)blockIdx.x * LH_PTS + q) * LH_THREADS + tid;
            if (m < M) gm = fmaxf(gm, fmaxf(fabsf(a.dout[m * a.dout_rs + (2 * l) * a.dout_cs]), fabsf(a.dout[m * a.dout_rs + (2 * l + 1) * a.dout_cs])));
        }
        if (!(gm <= 3.0e38f)) gm = 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o));
        if (lane == 0) atomicMax(&wg_max, __float_as_uint(gm));          // (non-negative floats order like their bits)
    }
    __syncthreads();
    const float gmax = __uint_as_float(wg_max);
    const double two44 = 17592186044416.0;
    const double S = gmax > 0.f ? two44 / (double)gmax : 0.0, S_inv = (double)gmax / two44;
    // cells >= ~7 pixels wide: whole runs of lanes share a corner, and 64 LDS atomics on one address serialise --
    // combine each run with a segmented wave scan first (as k_hashgrid_bwd2<true>) and insert once per run
    const bool premerge = res <= 64;
#pragma unroll 1
    for (int q = 0; q < LH_PTS; ++q) {
        const long long m = ((long long)blockIdx.x * LH_PTS + q) * LH_THREADS + tid;
        const bool valid = m < M;
        const long long mc = valid ? m : 0;
        float w[3];
        unsigned cell[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float xn = (a.x[mc * a.x_rs + d * a.x_cs] + a.radius) * a.inv_2r;
            float p = xn * scale + 0.5f;
            float fl = floorf(p);
            w[d] = p - fl;
            cell[d] = (unsigned)(int)fl;
        }
        const float g0 = valid ? a.dout[mc * a.dout_rs + (2 * l) * a.dout_cs] : 0.f;
        const float g1 = valid ? a.dout[mc * a.dout_rs + (2 * l + 1) * a.dout_cs] : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1), cz = cell[2] + ((c >> 2) & 1);
            float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]) *
                       (((c >> 2) & 1) ? w[2] : 1.f - w[2]);
            const unsigned idx = valid ? grid_index(cx, cy, cz, res, size) : LH_INVALID;
            float v0 = g0 * wt, v1 = g1 * wt;
            bool emit = valid;
            if (premerge) {                            // wave-uniform branch: every lane takes part in the shuffles
                unsigned prev = (unsigned)__shfl_up((int)idx, 1);
                int start = (lane == 0 || prev != idx) ? lane : 0;
#pragma unroll
                for (int ofs = 1; ofs < 64; ofs <<= 1) {
                    int s2 = __shfl_up(start, ofs);
                    if (lane >= ofs) start = max(start, s2);
                }
#pragma unroll
                for (int ofs = 1; ofs < 64; ofs <<= 1) {
                    float t0 = __shfl_up(v0, ofs), t1 = __shfl_up(v1, ofs);
                    if (lane - ofs >= start) { v0 += t0; v1 += t1; }
                }
                unsigned nxt = (unsigned)__shfl_down((int)idx, 1);
                emit = valid && (lane == 63 || nxt != idx);
            }
            if (emit) {
                unsigned h = (idx * 2654435761u) >> 21;                  // 11 bits = LH_SLOTS
                bool placed = false;
#pragma unroll 1
                for (int pr = 0; pr < LH_PROBES; ++pr) {
                    unsigned old = atomicCAS(&keys[h], LH_EMPTY, idx);
                    if (old == LH_EMPTY || old == idx) {
                        atomicAdd(&vals[2 * h], (unsigned long long)lh_fixed((double)v0 * S));
                        atomicAdd(&vals[2 * h + 1], (unsigned long long)lh_fixed((double)v1 * S));
                        placed = true;
                        break;
                    }
                    h = (h + 1) & (LH_SLOTS - 1);
                }
                if (!placed) {
                    float* dst = a.dtable + 2 * (size_t)(off + idx);
                    atomicAdd(dst, v0);
                    atomicAdd(dst + 1, v1);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < LH_SLOTS; i += LH_THREADS) {
        const unsigned k = keys[i];
        if (k != LH_EMPTY) {
            float* dst = a.dtable + 2 * (size_t)(off + k);
            atomicAdd(dst, (float)((double)(long long)vals[2 * i] * S_inv));
            atomicAdd(dst + 1, (float)((double)(long long)vals[2 * i + 1] * S_inv));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, third formulation -- the hashed levels.  fp32 global atomics retire at ~20.7 G/s on MI355X whatever the
// table size or the sharing pattern (tools/atomic_probe.cpp: shared table, one private table per XCD, lane pairs on one
// 8-byte slot: all 20.5-20.8 G/s), and on a hashed level nothing can be merged locally: a level's 18 M corner updates
// (8 x 2.3 M points) hit its 2^19 entries pseudo-randomly, ~35 updates per entry but never from neighbouring lanes.
// So the updates are first ROUTED, then accumulated where atomics are cheap:
//   pass 1  k_hg_bin    every point emits its 8 (entry, g0*w, g1*w) tuples per level into one of size/16384 bins (a bin =
//                       16384 consecutive table entries); a workgroup ranks its tuples per bin with LDS counters and
//                       reserves the bin space with ONE global atomic per (workgroup, bin) -- 128x fewer than per tuple;
//   pass 2  k_hg_acc    one workgroup per (level, bin, split) sums its share of the bin's tuples into a 128 KB LDS table
//                       and writes the table to a partial slab.  The table is 64-bit FIXED POINT (value * 2^shift / max|dL/denc|, shift <= 44: hb_pow2_scale,
//                       of the level, found by pass 1): ds_add_u64 retires 8.6x the tuples per second of two ds_add_f32
//                       (855 vs 99 G tuples/s, tools/lds_atomic_probe.cpp -- LDS float atomics are a slow path on this
//                       chip), resolves ~1e-13 of the level's largest gradient, cannot wrap, and makes the sums order-independent;
//   pass 3  k_hg_sum    dtable += sum over the splits' slabs (plain read-modify-write, no atomics).
// A tuple that finds its bin full (capacity = 1.3 x the uniform share) falls back to the global atomic: time, not
// correctness.  Bytes: 8 B per tuple written and read once = 0.13 GB per level and pass at 1 M points.
constexpr int HB_LOG2 = 13, HB_ENTRIES = 1 << HB_LOG2;     // table entries per bin: 2 x 8192 x 8 B fixed point = 128 KB of LDS
constexpr int HB_MAX_BINS = 128;                           // log2_hashmap_size <= 20
constexpr int HB_SPLITS = 4;                               // at most: BinArgs::splits (1: pass 2 adds into dtable itself, no slabs, no pass 3)
constexpr int HB_ACC_THREADS = 1024;

struct BinArgs {
    HashArgs h;
    int levels[kMaxLevels];        // binned levels (indices into h.lv)
    int n_binned;
    uint2* tuples;                 // [n_binned][bins][cap], 8 bytes each: entry-in-bin (14 bits) | v0 (25) | v1 (25)
    int bins;                      // bins of the largest binned level (row pitch of `tuples`)
    unsigned* counts;              // [n_binned][HB_MAX_BINS]   (zeroed by the launcher)
    unsigned* gmax_bits;           // [n_binned] max |dL/denc| of the level as float bits (zeroed by the launcher)
    long long* partial;            // [n_binned][splits][max_size * 2] fixed point (splits > 1)
    int splits;                    // workgroups per (level, bin) in pass 2
    long long cap;                 // tuples per bin
    long long max_size;            // largest level size among the binned levels
};

// 8-byte tuple: bits [0,14) entry within the bin, [14,39) v0, [39,64) v1, the values as fp32 with the low 7 mantissa bits
// rounded away (sign, 8 exponent bits, 16 mantissa bits: relative error <= 2^-17 per contribution, far inside the 1e-3
// gradient tolerance) -- half the bytes of the (index, fp32, fp32, pad) tuple of the first version, on both passes.
__device__ __forceinline__ uint2 hb_pack(unsigned e, float v0, float v1) {
    const unsigned long long a = (__float_as_uint(v0) + 0x40u) >> 7, b = (__float_as_uint(v1) + 0x40u) >> 7;   // round to nearest
    const unsigned long long w = (unsigned long long)e | (a << HB_LOG2) | (b << (HB_LOG2 + 25));
    return make_uint2((unsigned)w, (unsigned)(w >> 32));
}
// round-to-nearest double -> int64 for |d| < 2^51 with two plain fp64 / integer ops (no 64-bit convert instruction needed)
__device__ __forceinline__ long long hb_fixed(double d) {
    const double magic = 6755399441055744.0;               // 1.5 * 2^52
    return __double_as_longlong(d + magic) - __double_as_longlong(magic);
}
// Fixed-point scale of a level = 2^shift / max|g| with shift = min(44, 62 - ceil(log2(cap))): an entry receives at most `cap`
// tuples (one bin's capacity, over all splits), each |q| <= 2^shift, so the int64 sums of k_hg_acc + k_hg_sum cannot wrap
// however many points fall into one cell (flat / degenerate geometry); 2^-43 of the level's largest gradient at the bench
// size is far below the fp32 result's own rounding.
__device__ __forceinline__ double hb_pow2_scale(long long cap) {
    const int bits = cap > 1 ? 64 - __clzll(cap - 1) : 0;
    const int shift = min(44, 62 - bits);
    return __longlong_as_double((long long)(1023 + shift) << 52);
}
__device__ __forceinline__ void hb_unpack(unsigned lo, unsigned hi, unsigned& e, float& v0, float& v1) {
    const unsigned long long w = (unsigned long long)lo | ((unsigned long long)hi << 32);
    e = (unsigned)w & (HB_ENTRIES - 1);
    v0 = __uint_as_float((unsigned)((w >> HB_LOG2) & 0x1ffffffu) << 7);
    v1 = __uint_as_float((unsigned)(w >> (HB_LOG2 + 25)) << 7);
}

__global__ __launch_bounds__(256) void k_hg_bin(BinArgs a) {
    __shared__ unsigned lcnt[HB_MAX_BINS], lbase[HB_MAX_BINS], lmax;
    const int tid = threadIdx.x;
    const int li = blockIdx.y, l = a.levels[li];
    if (tid < HB_MAX_BINS) lcnt[tid] = 0;
    if (tid == HB_MAX_BINS) lmax = 0;
    __syncthreads();
    const long long M = a.h.m_dev ? (long long)*a.h.m_dev : a.h.m_max;
    const long long m = (long long)blockIdx.x * blockDim.x + tid;
    const bool valid = m < M;
    const float scale = a.h.lv.scale[l];
    const unsigned res = a.h.lv.res[l], size = a.h.lv.size[l], off = a.h.lv.offset[l];
    unsigned idx[8], rank[8];
    float v0[8], v1[8];
    float gm = 0.f;
    if (valid) {
        float w[3];
        unsigned cell[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float xn = (a.h.x[m * a.h.x_rs + d * a.h.x_cs] + a.h.radius) * a.h.inv_2r;
            float p = xn * scale + 0.5f;
            float fl = floorf(p);
            w[d] = p - fl;
            cell[d] = (unsigned)(int)fl;
        }
        const float g0 = a.h.dout[m * a.h.dout_rs + (2 * l) * a.h.dout_cs];
        const float g1 = a.h.dout[m * a.h.dout_rs + (2 * l + 1) * a.h.dout_cs];
        gm = fmaxf(fabsf(g0), fabsf(g1));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned cx = cell[0] + (c & 1), cy = cell[1] + ((c >> 1) & 1), cz = cell[2] + ((c >> 2) & 1);
            float wt = ((c & 1) ? w[0] : 1.f - w[0]) * (((c >> 1) & 1) ? w[1] : 1.f - w[1]) *
                       (((c >> 2) & 1) ? w[2] : 1.f - w[2]);
            idx[c] = grid_index(cx, cy, cz, res, size);
            v0[c] = g0 * wt;
            v1[c] = g1 * wt;
#if defined(DM_ABL_HG_NORANK)
            rank[c] = (unsigned)(c * 4 + (tid & 3));        // ABLATION (wrong results): no LDS ranking atomics
#else
            rank[c] = atomicAdd(&lcnt[idx[c] >> HB_LOG2], 1u);
#endif
        }
    }
    // the level's largest |gradient| (scale of pass 2's fixed point): wave max, one global atomicMax per wave; NaN / Inf
    // gradients are not representable in fixed point and are left out of the max (their tuples then saturate to 0 / garbage,
    // like any sum that contains them)
    if (!(gm <= 3.0e38f)) gm = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o));
    if ((tid & 63) == 0) atomicMax(&lmax, __float_as_uint(gm));          // (non-negative floats order like their bits)
    __syncthreads();
    if (tid < HB_MAX_BINS) {
        const unsigned n = lcnt[tid];
#if defined(DM_ABL_HG_NOGATOMIC)
        lbase[tid] = (unsigned)((blockIdx.x * 37u) % 100000u);      // ABLATION (wrong results): no global reservation
#else
        lbase[tid] = n ? atomicAdd(&a.counts[li * HB_MAX_BINS + tid], n) : 0u;
#endif
    }
    // one same-address global atomic per WORKGROUP at most, and only while this workgroup's max still beats the published
    // one (a plain read first): 15 k waves per level hammering one address cost 4 ms
    if (tid == HB_MAX_BINS && lmax > __atomic_load_n(&a.gmax_bits[li], __ATOMIC_RELAXED)) atomicMax(&a.gmax_bits[li], lmax);
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned bin = idx[c] >> HB_LOG2;
            const long long pos = (long long)lbase[bin] + rank[c];
            if (pos < a.cap) {
#if defined(DM_ABL_HG_NOSTORE)
                if (v0[c] == 123.456f)                      // ABLATION (wrong results): tuples never stored
#endif
                a.tuples[((long long)li * a.bins + bin) * a.cap + pos] = hb_pack(idx[c] & (HB_ENTRIES - 1), v0[c], v1[c]);
            } else {                                       // bin full: the slow, always-correct route
                float* dst = a.h.dtable + 2 * (size_t)(off + idx[c]);
                atomicAdd(dst, v0[c]);
                atomicAdd(dst + 1, v1[c]);
            }
        }
    }
}

__global__ __launch_bounds__(HB_ACC_THREADS) void k_hg_acc(BinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];      // [HB_ENTRIES][2] fixed point
    const int tid = threadIdx.x;
    const int li = blockIdx.y, l = a.levels[li];
    const int bin = blockIdx.x / a.splits, split = blockIdx.x - bin * a.splits;
    const unsigned size = a.h.lv.size[l];
    if ((long long)bin * HB_ENTRIES >= (long long)size) return;      // this level has fewer bins
    for (int i = tid; i < HB_ENTRIES; i += HB_ACC_THREADS) reinterpret_cast<uint4*>(tab)[i] = make_uint4(0u, 0u, 0u, 0u);
    const float gmax = __uint_as_float(a.gmax_bits[li]);
    [[maybe_unused]] const double S = gmax > 0.f ? hb_pow2_scale(a.cap) / (double)gmax : 0.0;   // |q| <= 2^shift, cap terms at most: no wrap
    __syncthreads();
    const long long n = min((long long)a.counts[li * HB_MAX_BINS + bin], a.cap);
    [[maybe_unused]] const long long lo = n * split / a.splits, hi = n * (split + 1) / a.splits;      // (host pass: unused)
    [[maybe_unused]] const uint2* src = a.tuples + ((long long)li * a.bins + bin) * a.cap;
    // eight independent 16-byte loads in flight per thread.  They are BUFFER loads whose descriptor ends at this split's
    // last tuple: lanes past the end read zeros (index 0, value 0: adds nothing) without a branch -- a conditional
    // `i < hi ? src[i] : 0` compiles to a branch and an s_waitcnt vmcnt(0) per load (the waitcnt pass assumes the load-free
    // path), i.e. eight SERIALIZED HBM round trips per trip: 2.6 ms for 1.4 GB.
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(unsigned)(hi * 8), 0x00020000);
    constexpr int U = 16;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    for (long long i0 = lo; i0 < hi; i0 += (long long)U * HB_ACC_THREADS) {
        u32x2 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            t[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(unsigned)((i0 + (long long)u * HB_ACC_THREADS + tid) * 8), 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned e;
            float v0, v1;
            hb_unpack(t[u][0], t[u][1], e, v0, v1);         // an all-zero (out-of-range) tuple adds 0 to entry 0
            atomicAdd(&tab[2 * e], (unsigned long long)hb_fixed((double)v0 * S));
            atomicAdd(&tab[2 * e + 1], (unsigned long long)hb_fixed((double)v1 * S));
        }
    }
#endif
    __syncthreads();
    const int n_ent = (int)min((long long)HB_ENTRIES, (long long)size - (long long)bin * HB_ENTRIES);
    if (a.splits == 1) {
        // one workgroup per (level, bin): it owns the bin's table entries -- convert and add into dtable here (no slab, no pass 3)
        const double inv = (double)gmax / hb_pow2_scale(a.cap);
        float2* dt = reinterpret_cast<float2*>(a.h.dtable) + a.h.lv.offset[l] + (long long)bin * HB_ENTRIES;
        for (int i = tid; i < n_ent; i += HB_ACC_THREADS) {
            const long long q0 = (long long)tab[2 * i], q1 = (long long)tab[2 * i + 1];
            if ((q0 | q1) != 0) {
                float2 cur = dt[i];
                cur.x += (float)((double)q0 * inv); cur.y += (float)((double)q1 * inv);
                dt[i] = cur;
            }
        }
        return;
    }
    long long* dst = a.partial + ((long long)li * a.splits + split) * a.max_size * 2 + (long long)bin * HB_ENTRIES * 2;
    for (int i = tid; i < n_ent; i += HB_ACC_THREADS) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<uint4*>(tab)[i];
}

__global__ __launch_bounds__(256) void k_hg_sum(BinArgs a) {
    const int li = blockIdx.y, l = a.levels[li];
    const unsigned size = a.h.lv.size[l], off = a.h.lv.offset[l];
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= size) return;
    long long q0 = 0, q1 = 0;
#pragma unroll
    for (int s = 0; s < a.splits; ++s) {
        const longlong2 p = reinterpret_cast<const longlong2*>(a.partial + ((long long)li * a.splits + s) * a.max_size * 2)[e];
        q0 += p.x; q1 += p.y;
    }
    const double inv = (double)__uint_as_float(a.gmax_bits[li]) / hb_pow2_scale(a.cap);
    const float2 acc = make_float2((float)((double)q0 * inv), (float)((double)q1 * inv));
    float2* dst = reinterpret_cast<float2*>(a.h.dtable) + off + e;
    float2 cur = *dst;
    cur.x += acc.x; cur.y += acc.y;
    *dst = cur;
}

bool fill_levels(GridLevels& lv, int n_levels, const float* scale, const uint32_t* res, const uint32_t* size,
                 const uint32_t* offset) {
    if (n_levels <= 0 || n_levels > kMaxLevels || !scale || !res || !size || !offset) return false;
    lv.n_levels = n_levels;
    for (int i = 0; i < n_levels; ++i) {
        if (size[i] == 0) return false;
        lv.scale[i] = scale[i]; lv.res[i] = res[i]; lv.size[i] = size[i]; lv.offset[i] = offset[i];
    }
    return true;
}

}  // namespace

extern "C" {

int dm_hashgrid_fwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                    const float* table, int n_levels, const float* lv_scale, const uint32_t* lv_res,
                    const uint32_t* lv_size, const uint32_t* lv_offset, float radius, float* enc, long long enc_rs,
                    long long enc_cs, hipStream_t stream) {
    HashArgs a = {};
    if (!x || !table || !enc || m_max <= 0 || !(radius > 0.f) || !fill_levels(a.lv, n_levels, lv_scale, lv_res, lv_size, lv_offset))
        return DM_ERR_ARG;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs; a.table = (const float2*)table; a.out = enc; a.out_rs = enc_rs;
    a.out_cs = enc_cs; a.m_dev = m_dev; a.m_max = m_max; a.radius = radius; a.inv_2r = 1.0f / (2.0f * radius);
    dim3 grid(dm_div_up(m_max, 256), n_levels);
    DM_ENTER();
    hipLaunchKernelGGL(k_hashgrid<false>, grid, dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// dtable must be zero-initialised (or hold the running gradient) by the caller: this ADDS into it.
// The 2-D grid of the uv-space field (n_input_dims = 2): x [M,2] by strides, level sizes from res^2 (see csrc comment).
int dm_hashgrid2d_fwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max, const float* table,
                      int n_levels, const float* lv_scale, const uint32_t* lv_res, const uint32_t* lv_size, const uint32_t* lv_offset,
                      float radius, float* enc, long long enc_rs, long long enc_cs, hipStream_t stream) {
    HashArgs a = {};
    if (!x || !table || !enc || m_max <= 0 || !(radius > 0.f) || !fill_levels(a.lv, n_levels, lv_scale, lv_res, lv_size, lv_offset))
        return DM_ERR_ARG;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs; a.table = (const float2*)table; a.out = enc; a.out_rs = enc_rs;
    a.out_cs = enc_cs; a.m_dev = m_dev; a.m_max = m_max; a.radius = radius; a.inv_2r = 1.0f / (2.0f * radius);
    DM_ENTER();
    hipLaunchKernelGGL(k_hashgrid2d<false>, dim3(dm_div_up(m_max, 256), n_levels), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_hashgrid2d_bwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max, const float* denc,
                      long long denc_rs, long long denc_cs, int n_levels, const float* lv_scale, const uint32_t* lv_res,
                      const uint32_t* lv_size, const uint32_t* lv_offset, float radius, float* dtable, hipStream_t stream) {
    HashArgs a = {};
    if (!x || !denc || !dtable || m_max <= 0 || !(radius > 0.f) || !fill_levels(a.lv, n_levels, lv_scale, lv_res, lv_size, lv_offset))
        return DM_ERR_ARG;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs; a.dout = denc; a.dout_rs = denc_rs; a.dout_cs = denc_cs;
    a.dtable = dtable; a.m_dev = m_dev; a.m_max = m_max; a.radius = radius; a.inv_2r = 1.0f / (2.0f * radius);
    DM_ENTER();
    hipLaunchKernelGGL(k_hashgrid2d<true>, dim3(dm_div_up(m_max, 256), n_levels), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

int dm_hashgrid_bwd(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                    const float* denc, long long denc_rs, long long denc_cs, int n_levels, const float* lv_scale,
                    const uint32_t* lv_res, const uint32_t* lv_size, const uint32_t* lv_offset, float radius,
                    float* dtable, hipStream_t stream) {
    HashArgs a = {};
    if (!x || !denc || !dtable || m_max <= 0 || !(radius > 0.f) || !fill_levels(a.lv, n_levels, lv_scale, lv_res, lv_size, lv_offset))
        return DM_ERR_ARG;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs; a.dout = denc; a.dout_rs = denc_rs; a.dout_cs = denc_cs;
    a.dtable = dtable; a.m_dev = m_dev; a.m_max = m_max; a.radius = radius; a.inv_2r = 1.0f / (2.0f * radius);
    // coarsest levels: LDS hash accumulation per 1024-point strip; mid levels (cells ~1-3 pixels): run combining
    // along the image row; fine levels: one atomic pair per corner (nothing to share)
    int n_lds = 0;
    while (n_lds < n_levels && lv_res[n_lds] <= 110) ++n_lds;
    int n_dedup = n_lds;
    while (n_dedup < n_levels && lv_res[n_dedup] <= 512) ++n_dedup;
    DM_ENTER();
    if (n_lds > 0)
        hipLaunchKernelGGL(k_hashgrid_bwd_lds, dim3(dm_div_up(m_max, LH_THREADS * LH_PTS), n_lds), dim3(LH_THREADS), 0,
                           stream, a, 0);
    if (n_dedup > n_lds)
        hipLaunchKernelGGL(k_hashgrid_bwd2<true>, dim3(dm_div_up(2 * m_max, 256), n_dedup - n_lds), dim3(256), 0, stream,
                           a, n_lds);
    if (n_levels > n_dedup)
        hipLaunchKernelGGL(k_hashgrid_bwd2<false>, dim3(dm_div_up(2 * m_max, 256), n_levels - n_dedup), dim3(256), 0,
                           stream, a, n_dedup);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Workspace of dm_hashgrid_bwd_binned for at most m_max points (0 if no level of this grid takes the binned route).
static bool hg_level_hashed(uint32_t res, uint32_t size) { return (unsigned long long)res * res * res > size; }
static long long hg_bin_cap(long long m_max, uint32_t size) {
    const long long bins = (size + HB_ENTRIES - 1) / HB_ENTRIES;
    return (8 * m_max / bins) * 13 / 10 + 4096;
}
size_t dm_hashgrid_bwd_workspace_bytes(long long m_max, int n_levels, const uint32_t* lv_res, const uint32_t* lv_size) {
    if (m_max <= 0 || n_levels <= 0 || n_levels > kMaxLevels || !lv_res || !lv_size) return 0;
    long long n_binned = 0, cap = 0, max_size = 0;
    for (int i = 0; i < n_levels; ++i)
        if (hg_level_hashed(lv_res[i], lv_size[i]) && lv_size[i] <= (uint32_t)HB_MAX_BINS * HB_ENTRIES) {
            ++n_binned;
            cap = std::max(cap, hg_bin_cap(m_max, lv_size[i]));
            max_size = std::max<long long>(max_size, lv_size[i]);
        }
    if (!n_binned) return 0;
    const long long bins = (max_size + HB_ENTRIES - 1) / HB_ENTRIES;
    return (size_t)(n_binned * bins * cap * 8 + n_binned * (HB_MAX_BINS + 1) * 4 + 256 + n_binned * HB_SPLITS * max_size * 16 + 256);
}

// dm_hashgrid_bwd with the hashed levels routed through bins (see k_hg_bin); the dense levels run the kernels of
// dm_hashgrid_bwd.  `workspace` = dm_hashgrid_bwd_workspace_bytes(m_max, ...) bytes of device memory, 256 B aligned.
int dm_hashgrid_bwd_binned(const float* x, long long x_rs, long long x_cs, const int32_t* m_dev, long long m_max,
                           const float* denc, long long denc_rs, long long denc_cs, int n_levels, const float* lv_scale,
                           const uint32_t* lv_res, const uint32_t* lv_size, const uint32_t* lv_offset, float radius,
                           float* dtable, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    BinArgs b = {};
    HashArgs& a = b.h;
    if (!x || !denc || !dtable || m_max <= 0 || !(radius > 0.f) || !fill_levels(a.lv, n_levels, lv_scale, lv_res, lv_size, lv_offset))
        return DM_ERR_ARG;
    const size_t need = dm_hashgrid_bwd_workspace_bytes(m_max, n_levels, lv_res, lv_size);
    if (need && (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))) return DM_ERR_WORKSPACE;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs; a.dout = denc; a.dout_rs = denc_rs; a.dout_cs = denc_cs;
    a.dtable = dtable; a.m_dev = m_dev; a.m_max = m_max; a.radius = radius; a.inv_2r = 1.0f / (2.0f * radius);
    // dense levels: contiguous prefix (resolution grows with the level), same kernels as dm_hashgrid_bwd
    int n_dense = 0;
    while (n_dense < n_levels && !(hg_level_hashed(lv_res[n_dense], lv_size[n_dense]) &&
                                   lv_size[n_dense] <= (uint32_t)HB_MAX_BINS * HB_ENTRIES)) ++n_dense;
    long long cap = 0, max_size = 0;
    for (int i = n_dense; i < n_levels; ++i) {
        if (!(hg_level_hashed(lv_res[i], lv_size[i]) && lv_size[i] <= (uint32_t)HB_MAX_BINS * HB_ENTRIES)) return DM_ERR_UNSUPPORTED;
        b.levels[b.n_binned++] = i;
        cap = std::max(cap, hg_bin_cap(m_max, lv_size[i]));
        max_size = std::max<long long>(max_size, lv_size[i]);
    }
    DM_ENTER();
    if (n_dense > 0) {
        int n_lds = 0;
        while (n_lds < n_dense && lv_res[n_lds] <= 110) ++n_lds;
        if (n_lds > 0)
            hipLaunchKernelGGL(k_hashgrid_bwd_lds, dim3(dm_div_up(m_max, LH_THREADS * LH_PTS), n_lds), dim3(LH_THREADS), 0, stream, a, 0);
        if (n_dense > n_lds)
            hipLaunchKernelGGL(k_hashgrid_bwd2<true>, dim3(dm_div_up(2 * m_max, 256), n_dense - n_lds), dim3(256), 0, stream, a, n_lds);
    }
    if (b.n_binned > 0) {
        char* ws = (char*)workspace;
        b.bins = (int)((max_size + HB_ENTRIES - 1) / HB_ENTRIES);
        b.tuples = (uint2*)ws;
        ws += (size_t)b.n_binned * b.bins * cap * 8;
        b.counts = (unsigned*)ws;
        b.gmax_bits = b.counts + (size_t)b.n_binned * HB_MAX_BINS;
        ws += ((size_t)b.n_binned * (HB_MAX_BINS + 1) * 4 + 255) / 256 * 256;
        b.partial = (long long*)ws;
        b.cap = cap; b.max_size = max_size;
        if (cap * 8 > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;       // k_hg_acc addresses a bin with 32-bit byte offsets
        hipError_t e = hipMemsetAsync(b.counts, 0, (size_t)b.n_binned * (HB_MAX_BINS + 1) * 4, stream);
        if (e != hipSuccess) return (int)e;
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hg_acc), hipFuncAttributeMaxDynamicSharedMemorySize, HB_ENTRIES * 16);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const int max_bins = (int)((max_size + HB_ENTRIES - 1) / HB_ENTRIES);
        hipLaunchKernelGGL(k_hg_bin, dim3(dm_div_up(m_max, 256), b.n_binned), dim3(256), 0, stream, b);
        // workgroups per (level, bin): one when the (level, bin) pairs alone fill the chip twice over (the bench: 11 levels x 64 bins
        // = 704 workgroups of 1024 threads: pass 2 then adds into dtable itself -- no 128 KB slab per workgroup written and read back,
        // no pass 3), else up to HB_SPLITS
        static const int env_splits = getenv("DREAMMAT_HASHGRID_SPLITS") ? atoi(getenv("DREAMMAT_HASHGRID_SPLITS")) : 0;
        const long long pairs = (long long)max_bins * b.n_binned;
        b.splits = env_splits >= 1 && env_splits <= HB_SPLITS ? env_splits : (int)std::max<long long>(1, std::min<long long>(HB_SPLITS, 512 / std::max<long long>(1, pairs)));
        hipLaunchKernelGGL(k_hg_acc, dim3(max_bins * b.splits, b.n_binned), dim3(HB_ACC_THREADS), HB_ENTRIES * 16, stream, b);
        if (b.splits > 1) hipLaunchKernelGGL(k_hg_sum, dim3(dm_div_up(max_size, 256), b.n_binned), dim3(256), 0, stream, b);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // extern "C"
