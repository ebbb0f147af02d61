// Single-wave issue-cost microbenchmark (gfx950): how many cycles does one wave, alone on its SIMD, need per instruction
// of the attention inner loop -- bare and beside v_mfma_f32_32x32x16_bf16?  s_memtime around unrolled blocks, 4 waves per
// workgroup (one per SIMD), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.cpp -o tools/_issue_probe && tools/_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int iters) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(float)(threadIdx.x & 3); B[i] = (__bf16)1.0f; }
    unsigned pk[4] = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 32 independent v_exp_f32 (16 registers round-robin: each result is consumed 16 instructions later)
            REP32(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[0]), "+v"(v[1]));
                  asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[2]), "+v"(v[3]));)
        } else if (MODE == 1) {   // v_add_f32
            REP32(asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
                  asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));)
        } else if (MODE == 2) {   // v_cvt_pk_bf16_f32
            REP32(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %3, %2, %1" : "=v"(pk[0]), "+v"(v[0]), "+v"(v[1]), "=v"(pk[1]));
                  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %3, %2, %1" : "=v"(pk[2]), "+v"(v[2]), "+v"(v[3]), "=v"(pk[3]));)
        } else if (MODE == 3) {   // bare MFMAs, four independent accumulators
            REP32(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc[0]), "+v"(acc[1]) : "v"(A), "v"(B));
                  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B));)
        } else if (MODE == 4) {   // the attention chunk: 1 MFMA + 2 exp + 1 cvt_pk + 2 add   (x4 per asm, 32 asm = 128 chunks)
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %10, %8, %9\n\tv_add_f32 %11, %11, %8\n\tv_add_f32 %12, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_cvt_pk_bf16_f32 %10, %6, %7\n\tv_add_f32 %11, %11, %6\n\tv_add_f32 %12, %12, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %10, %8, %9\n\tv_add_f32 %11, %11, %8\n\tv_add_f32 %12, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_cvt_pk_bf16_f32 %10, %6, %7\n\tv_add_f32 %11, %11, %6\n\tv_add_f32 %12, %12, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(pk[0]), "v"(v[4]), "v"(v[5]));)
        } else if (MODE == 5) {   // the same without the exponentials: 1 MFMA + 1 cvt_pk + 2 add + 2 v_mul (plain VALU in their place)
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mul_f32 %6, %6, %6\n\tv_mul_f32 %7, %7, %7\n\tv_cvt_pk_bf16_f32 %10, %8, %9\n\tv_add_f32 %11, %11, %8\n\tv_add_f32 %12, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_mul_f32 %8, %8, %8\n\tv_mul_f32 %9, %9, %9\n\tv_cvt_pk_bf16_f32 %10, %6, %7\n\tv_add_f32 %11, %11, %6\n\tv_add_f32 %12, %12, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mul_f32 %6, %6, %6\n\tv_mul_f32 %7, %7, %7\n\tv_cvt_pk_bf16_f32 %10, %8, %9\n\tv_add_f32 %11, %11, %8\n\tv_add_f32 %12, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_mul_f32 %8, %8, %8\n\tv_mul_f32 %9, %9, %9\n\tv_cvt_pk_bf16_f32 %10, %6, %7\n\tv_add_f32 %11, %11, %6\n\tv_add_f32 %12, %12, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(pk[0]), "v"(v[4]), "v"(v[5]));)
        } else if (MODE == 6) {   // 1 MFMA + 2 exp only
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));)
        } else if (MODE == 7) {   // 1 MFMA + 4 exp
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));)
        } else if (MODE == 8) {   // 1 MFMA + 6 plain VALU
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_add_f32 %6, %6, %7\n\tv_add_f32 %7, %7, %8\n\tv_add_f32 %8, %8, %9\n\tv_add_f32 %9, %9, %6\n\tv_add_f32 %10, %10, %6\n\tv_add_f32 %11, %11, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_add_f32 %6, %6, %7\n\tv_add_f32 %7, %7, %8\n\tv_add_f32 %8, %8, %9\n\tv_add_f32 %9, %9, %6\n\tv_add_f32 %10, %10, %6\n\tv_add_f32 %11, %11, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_add_f32 %6, %6, %7\n\tv_add_f32 %7, %7, %8\n\tv_add_f32 %8, %8, %9\n\tv_add_f32 %9, %9, %6\n\tv_add_f32 %10, %10, %6\n\tv_add_f32 %11, %11, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_add_f32 %6, %6, %7\n\tv_add_f32 %7, %7, %8\n\tv_add_f32 %8, %8, %9\n\tv_add_f32 %9, %9, %6\n\tv_add_f32 %10, %10, %6\n\tv_add_f32 %11, %11, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int a = 0; a < 4; ++a) s += acc[a][threadIdx.x & 15];
    s += (float)(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]);
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, double instr_per_iter, int waves_per_simd) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
    const int iters = 20;
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads > 1024 ? 1024 : threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads > 1024 ? 1024 : threads), 0, 0, d, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), d, 256 * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += x;
    const double ticks = sum / h.size() / iters;
    printf("{\"probe\": \"%s\", \"ticks_per_block\": %.1f, \"ticks_per_unit\": %.2f, \"kernel_us\": %.1f}\n", name, ticks, ticks / instr_per_iter, ms * 1e3);
    hipFree(d); hipFree(sink);
}

int main() {
    run<0>("v_exp_f32 x128 (per instr)", 128, 1);
    run<1>("v_add_f32 x128 (per instr)", 128, 1);
    run<2>("v_cvt_pk_bf16_f32 x128 (per instr)", 128, 1);
    run<3>("mfma 32x32x16 bf16 x128 (per mfma)", 128, 1);
    run<4>("chunk: mfma + 2 exp + cvt + 2 add, x128 (per chunk)", 128, 1);
    run<5>("chunk: mfma + 2 mul + cvt + 2 add, x128 (per chunk)", 128, 1);
    run<6>("chunk: mfma + 2 exp, x128 (per chunk)", 128, 1);
    run<7>("chunk: mfma + 4 exp, x128 (per chunk)", 128, 1);
    run<8>("chunk: mfma + 6 add, x128 (per chunk)", 128, 1);
    return 0;
}
