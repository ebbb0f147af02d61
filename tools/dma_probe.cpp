// What does one CU sustain when its waves stream 1 KB pieces from L2 into LDS -- the operand path of the LDS-DMA conv / GEMM kernels
// (csrc/conv.hip) -- as `buffer_load_dwordx4 ... lds` (no VGPR staging) and as `buffer_load_dwordx4` into VGPRs followed by
// ds_write_b128?  Round 5: the ping-pong schedule experiment left the K-step time unchanged (64 pieces in ~4600 cycles whoever
// issues them and wherever), which points at the path's THROUGHPUT, not at the schedule.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_dma_probe tools/dma_probe.cpp && tools/_dma_probe
// One workgroup per CU (256 workgroups x NW waves); every wave issues `trips` x 8 pieces with a vmcnt(8)-style wait per trip; the
// source is a per-workgroup 64 KB window of a 16 MB table (L2 / MALL resident after the warm-up).  Patterns:
//   lin   : lane l reads bytes [16 l, 16 l + 16) of a 1 KB run
//   rows  : lane l reads chunk l & 7 of row l >> 3, rows `pitch` bytes apart (the conv kernels' activation / weight pieces:
//           8 rows x 128 B; pitch = 2 Cin)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA.  MODE 1: VGPR load + ds_write_b128.  MODE 2: VGPR load only (address-unit cost of the load itself).
template <int MODE>
__global__ __launch_bounds__(1024) void k_dma(const char* __restrict__ tab, unsigned tab_bytes, int trips, int pitch, unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)tab_bytes, 0x00020000);
    const unsigned win = (blockIdx.x * 65536u) % (tab_bytes - 4u * 65536u);
    const unsigned lane_off = pitch ? (unsigned)((lane >> 3) * pitch + (lane & 7) * 16) : (unsigned)(lane * 16);
    const unsigned piece_bytes = pitch ? 8u * (unsigned)pitch : 1024u;
    char* my = smem + wave * 8 * 1024;                 // 8 slots of 1 KB per wave
    unsigned acc = 0;
    for (int t = 0; t < trips; ++t) {
        u32x4 v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const unsigned off = win + ((unsigned)((t * 8 + p) * nw + wave) * piece_bytes) % 49152u + lane_off;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + p * 1024), 16, (int)off, 0, 0, 0);
                // 4 to 8 pieces in flight per wave: a slot is rewritten 8 pieces after it was requested, 4 of them waited for
                if (p == 3 || p == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                v[p] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (MODE == 1) *reinterpret_cast<u32x4*>(my + p * 1024 + lane * 16) = v[p];
                else acc ^= v[p][0] ^ v[p][3];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) acc ^= *reinterpret_cast<unsigned*>(smem + (tid * 16) % (nw * 8192));
    if (acc == 0x12345u) out[blockIdx.x * 1024 + tid] = acc;
#endif
}

template <int MODE>
static void run(const char* name, int nw, int pitch, const char* tab, unsigned tab_bytes, unsigned* out) {
    const int blocks = 256, trips = 64;
    const size_t lds = (size_t)nw * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_dma<MODE>), dim3(blocks), dim3(nw * 64), lds, 0, tab, tab_bytes, trips, pitch, out);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_dma<MODE>), dim3(blocks), dim3(nw * 64), lds, 0, tab, tab_bytes, trips, pitch, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double pieces_per_cu = (double)nw * trips * 8;
    printf("{\"probe\": \"dma\", \"case\": \"%s\", \"waves_per_cu\": %d, \"pitch\": %d, \"us\": %.2f, \"ns_per_piece_per_cu\": %.2f, \"GBps_per_cu\": %.1f, \"TBps_chip\": %.2f}\n",
           name, nw, pitch, us, us * 1e3 / pieces_per_cu, pieces_per_cu * 1024 / (us * 1e3), pieces_per_cu * 1024 * 256 / (us * 1e6));
}

int main() {
    const unsigned bytes = 16u << 20;
    char* tab; unsigned* out;
    CK(hipMalloc(&tab, bytes));
    CK(hipMemset(tab, 1, bytes));
    CK(hipMalloc(&out, 256 * 1024 * 4));
    for (int nw : {4, 8, 16}) {
        run<0>("lds-dma, lane-linear", nw, 0, tab, bytes, out);
        run<0>("lds-dma, 8 rows x 128 B, pitch 256", nw, 256, tab, bytes, out);
        run<0>("lds-dma, 8 rows x 128 B, pitch 640", nw, 640, tab, bytes, out);
        run<1>("vgpr load + ds_write_b128, lane-linear", nw, 0, tab, bytes, out);
        run<1>("vgpr load + ds_write_b128, 8 rows x 128 B, pitch 256", nw, 256, tab, bytes, out);
        run<2>("vgpr load only, lane-linear", nw, 0, tab, bytes, out);
        run<2>("vgpr load only, 8 rows x 128 B, pitch 256", nw, 256, tab, bytes, out);
    }
    return 0;
}
