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
constexpr int LH_SLOTS = 2048;                 // 8 KB keys + 16 KB values => 6 workgroups per CU
constexpr int LH_THREADS = 256;
constexpr int LH_PTS = 4;                      // points per thread (1024 consecutive points per workgroup)
constexpr int LH_PROBES = 8;
constexpr unsigned LH_EMPTY = 0xffffffffu;
constexpr unsigned LH_INVALID = 0xfffffffeu;   // key of lanes past the end of the point list (never inserted)

__global__ __launch_bounds__(LH_THREADS) void k_hashgrid_bwd_lds(HashArgs a, int level_base) {
    __shared__ unsigned keys[LH_SLOTS];
    __shared__ float vals[2 * LH_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LH_SLOTS; i += LH_THREADS) { keys[i] = LH_EMPTY; vals[2 * i] = 0.f; vals[2 * i + 1] = 0.f; }
    __syncthreads();
    const long long M = a.m_dev ? (long long)*a.m_dev : a.m_max;
    const int l = level_base + blockIdx.y;
    const float scale = a.lv.scale[l];
    const unsigned res = a.lv.res[l], size = a.lv.size[l], off = a.lv.offset[l];
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
                        atomicAdd(&vals[2 * h], v0);
                        atomicAdd(&vals[2 * h + 1], v1);
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
            atomicAdd(dst, vals[2 * i]);
            atomicAdd(dst + 1, vals[2 * i + 1]);
        }
    }
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

}  // extern "C"
