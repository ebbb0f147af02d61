// Micro-benchmark behind the round-4 atlas layout decision: what does a scattered 16-byte gather cost on gfx950 as a function of
//   * its ALIGNMENT (16-byte aligned vs 8-byte aligned at an odd 8-byte slot: the bilinear row {texel x0, texel x0+1} of an
//     8-byte-texel atlas is misaligned for every odd x0),
//   * its WIDTH (one dwordx4 vs two dwordx2 vs four dword),
//   * the COHERENCE of a wave's addresses (all lanes random; quads of 4 lanes share a 64-byte block; 16-lane groups share a
//     128-byte line; the whole wave inside 1 KB).
// Table = 2.6 MB (one environment's specular atlas: L2-resident), every thread issues `G` independent gathers per trip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_gather_probe tools/gather_probe.cpp && tools/_gather_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE: 0 dwordx4 aligned, 1 dwordx4 at +8 (misaligned), 2 two dwordx2 (at +8), 3 four dword (at +8), 4 dwordx2 only (8 B per gather),
//       5 one dword (streaming case only)
// COH : 4 = streaming (lane-linear addresses); 0 random per lane, 1 quad-coherent (4 lanes in one 64 B block), 2 16-lane groups in one 128 B line, 3 wave within 1 KB
template <int MODE, int COH>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ tab, unsigned n_slots16, int trips, unsigned* __restrict__ out) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, 0x7ffffffc, 0x00020000);
    for (int t = 0; t < trips; ++t) {
        unsigned offs[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            unsigned key, sub;
            if (COH == 0) { key = tid * 131u + t * 8u + g; sub = 0; }
            else if (COH == 1) { key = (tid >> 2) * 131u + t * 8u + g; sub = (lane & 3u); }           // 4 x 16 B = one 64 B block
            else if (COH == 2) { key = (tid >> 4) * 131u + t * 8u + g; sub = (lane & 7u); }           // 16 lanes over 8 slots of a 128 B line
            else if (COH == 4) {                                                                        // streaming: lane-linear
                const unsigned w = MODE == 5 ? 4u : (MODE == 4 ? 8u : 16u);                             // bytes per lane
                offs[g] = (unsigned)((((unsigned long long)(tid >> 6) * 64u * (unsigned)trips * 8u + (unsigned)(t * 8 + g) * 64u + lane) * w) % (n_slots16 * 16u - 1024u));
                continue;
            }
            else { key = (tid >> 6) * 131u + t * 8u + g; sub = lane & 63u; }                           // 64 slots = 1 KB
            unsigned slot = hash32(key) % (n_slots16 - 64u);
            if (COH == 1) slot &= ~3u;
            if (COH == 2) slot &= ~7u;
            if (COH == 3) slot &= ~63u;
            offs[g] = (slot + sub) * 16u + ((MODE >= 1 && MODE <= 3) ? 8u : 0u);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE <= 1) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)offs[g], 0, 0);
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            } else if (MODE == 2) {
                auto a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)offs[g], 0, 0);
                auto b = __builtin_amdgcn_raw_buffer_load_b64(r, (int)offs[g] + 8, 0, 0);
                acc ^= a[0] ^ a[1] ^ b[0] ^ b[1];
            } else if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc ^= __builtin_amdgcn_raw_buffer_load_b32(r, (int)offs[g] + 4 * k, 0, 0);
            } else if (MODE == 5) {
                acc ^= __builtin_amdgcn_raw_buffer_load_b32(r, (int)offs[g], 0, 0);
            } else {
                auto a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)offs[g], 0, 0);
                acc ^= a[0] ^ a[1];
            }
        }
    }
    if (acc == 0x12345u) out[tid] = acc;          // never true in practice: keeps the loads alive
}

template <int MODE, int COH>
static void run(const char* name, const char* tab, unsigned n_slots16, unsigned* out) {
    const int blocks = 256 * 8, trips = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gather<MODE, COH>), dim3(blocks), dim3(256), 0, 0, tab, n_slots16, trips, out);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gather<MODE, COH>), dim3(blocks), dim3(256), 0, 0, tab, n_slots16, trips, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gathers = (double)blocks * 256 * trips * 8;             // lane-gathers of 16 B (8 B for mode 4)
    const double us = ms * 1e3 / reps;
    printf("{\"probe\": \"gather\", \"case\": \"%s\", \"us\": %.2f, \"lane_gathers_per_ns\": %.2f, \"cycles_per_wave_gather_per_CU_at_2p1GHz\": %.1f}\n",
           name, us, gathers / (us * 1e3), us * 1e-6 * 2.1e9 / (gathers / 64.0 / 256.0));
}

int main() {
    const size_t bytes = 2600 * 1024;
    const unsigned n_slots16 = (unsigned)(bytes / 16);
    char* tab; unsigned* out;
    CK(hipMalloc(&tab, bytes + 4096));
    CK(hipMemset(tab, 1, bytes + 4096));
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0, 0>("x4 aligned, random", tab, n_slots16, out);
    run<1, 0>("x4 +8 misaligned, random", tab, n_slots16, out);
    run<2, 0>("2 x x2 (+8), random", tab, n_slots16, out);
    run<3, 0>("4 x dword (+8), random", tab, n_slots16, out);
    run<4, 0>("x2 only (8 B), random", tab, n_slots16, out);
    run<0, 1>("x4 aligned, quad-coherent", tab, n_slots16, out);
    run<1, 1>("x4 +8 misaligned, quad-coherent", tab, n_slots16, out);
    run<0, 2>("x4 aligned, 16-lane line-coherent", tab, n_slots16, out);
    run<1, 2>("x4 +8 misaligned, 16-lane line-coherent", tab, n_slots16, out);
    run<0, 3>("x4 aligned, wave within 1 KB", tab, n_slots16, out);
    run<1, 3>("x4 +8 misaligned, wave within 1 KB", tab, n_slots16, out);
    // streaming (lane-linear) loads of three widths: is a wave-instruction's cost its bytes or its 64 addresses?
    run<0, 4>("x4 streaming (1 KB per wave-instruction)", tab, n_slots16, out);
    run<4, 4>("x2 streaming (512 B per wave-instruction)", tab, n_slots16, out);
    run<5, 4>("dword streaming (256 B per wave-instruction)", tab, n_slots16, out);
    return 0;
}
