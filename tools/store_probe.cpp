// How fast does a CU retire the conv / Linear epilogue's store pattern?  The epilogue writes an output tile as 16-byte pieces, one
// per lane: lanes (r, hi) of an instruction write bytes [32 g + 16 hi, +16) of row r -- 32 contiguous bytes in each of 32 rows, the
// 128-byte line of a row completed by four consecutive instructions.  Round 5's timelines say a K = 320 Linear tile spends more
// time draining its stores than multiplying.  Patterns, same bytes, one workgroup of 8 waves per CU, persistent over row blocks:
//   piece32 : the epilogue's (32 B per row and instruction, 4 instructions per line)
//   piece64 : 64 B per row and instruction (4 lanes per row, 16 rows per instruction)
//   line128 : 128 B per row and instruction (8 lanes per row, 8 rows per instruction)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_store_probe tools/store_probe.cpp && tools/_store_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows x pitch bytes output; each wave owns 32-row blocks, writes `cols` bytes per row (cols = 128 * k)
template <int LPR>   // lanes per row: 2, 4, 8
__global__ __launch_bounds__(512) void k_store(char* __restrict__ y, long long rows, int pitch, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nblk = rows / 32;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (long long b = (long long)blockIdx.x * 8 + wave; b < nblk; b += (long long)gridDim.x * 8) {
        char* base = y + b * 32 * (long long)pitch;
        // per 128 bytes of columns: 32 rows x 128 B = 4 KB = 4 instructions of 1 KB whatever the shape
        for (int c = 0; c < cols; c += 128) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int row, off;
                if (LPR == 2) { row = lane & 31; off = 32 * t + 16 * (lane >> 5); }                 // 32 rows x 32 B
                else if (LPR == 4) { row = (lane >> 2) + 16 * (t & 1); off = 64 * (t >> 1) + 16 * (lane & 3); }   // 16 rows x 64 B
                else { row = (lane >> 3) + 8 * t; off = 16 * (lane & 7); }                          // 8 rows x 128 B
                *reinterpret_cast<u32x4*>(base + (long long)row * pitch + c + off) = v;
            }
        }
    }
}

template <int LPR>
static void run(const char* name, char* y, long long rows, int pitch, int cols) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_store<LPR>), dim3(256), dim3(512), 0, 0, y, rows, pitch, cols);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_store<LPR>), dim3(256), dim3(512), 0, 0, y, rows, pitch, cols);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)rows * cols;
    printf("{\"probe\": \"store\", \"pattern\": \"%s\", \"rows\": %lld, \"pitch\": %d, \"cols\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", name, rows, pitch, cols, us, bytes / (us * 1e3));
}

int main() {
    const long long rows = 98304 * 4;
    char* y;
    CK(hipMalloc(&y, rows * 2560LL));
    for (int pitch : {640, 1280, 2560}) {
        const int cols = pitch;      // whole rows: the Linear output [M, N] bf16 with N = 320 / 640 / 1280
        run<2>("piece32", y, rows, pitch, cols);
        run<4>("piece64", y, rows, pitch, cols);
        run<8>("line128", y, rows, pitch, cols);
    }
    // a 128-channel N tile of a wider output (conv: Cout = 128 of 128; the 256 x 128 tile of N = 640)
    run<2>("piece32 (256 B of 1280)", y, rows, 1280, 256);
    run<8>("line128 (256 B of 1280)", y, rows, 1280, 256);
    return 0;
}
