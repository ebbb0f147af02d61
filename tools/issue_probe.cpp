// Single-wave issue-cost microbenchmark (gfx950): how many cycles does one wave, alone on its SIMD, need per instruction
// of the attention inner loop -- bare and beside v_mfma_f32_32x32x16_bf16?  s_memtime around unrolled blocks, 4 waves per
// workgroup (one per SIMD), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.cpp -o tools/_issue_probe && tools/_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
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
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 d2[4];
    for (int i = 0; i < 4; ++i) d2[i] = f32x2{threadIdx.x * 0.5f, (float)i};
    const unsigned ones = 0x3f803f80u, sel = 0x07060302u;
    __shared__ __attribute__((aligned(16))) char lds[16384];
    if (MODE == 15) { for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.f; __syncthreads(); }
    const unsigned lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024 * 0;
    bf16x8 F0 = A, F1 = B;
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
        } else if (MODE == 9) {   // v_pk_add_f32 (two sums per instruction)
            REP32(asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %2, %2, %1" : "+v"(d2[0]), "+v"(d2[1]), "+v"(d2[2]));
                  asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %2, %2, %1" : "+v"(d2[3]), "+v"(d2[1]), "+v"(d2[2]));)
        } else if (MODE == 10) {  // v_dot2_f32_bf16 against {1, 1}: the sum of a packed pair in one instruction
            REP32(asm volatile("v_dot2_f32_bf16 %0, %2, %3, %0\n\tv_dot2_f32_bf16 %1, %2, %3, %1" : "+v"(v[0]), "+v"(v[1]) : "v"(pk[0]), "v"(ones));
                  asm volatile("v_dot2_f32_bf16 %0, %2, %3, %0\n\tv_dot2_f32_bf16 %1, %2, %3, %1" : "+v"(v[2]), "+v"(v[3]) : "v"(pk[1]), "v"(ones));)
        } else if (MODE == 11) {  // chunk with ONE v_pk_add_f32 for the two row-sum adds
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_pk_add_f32 %9, %9, %13\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_pk_add_f32 %9, %9, %12\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_pk_add_f32 %9, %9, %13\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_pk_add_f32 %9, %9, %12"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(pk[0]), "v"(d2[0]), "v"(v[2]), "v"(v[3]), "v"(d2[1]), "v"(d2[2]));)
        } else if (MODE == 12) {  // chunk with ONE v_dot2_f32_bf16 on the packed pair
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_dot2_f32_bf16 %9, %8, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_dot2_f32_bf16 %9, %8, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_dot2_f32_bf16 %9, %8, %12, %9\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_dot2_f32_bf16 %9, %8, %12, %9"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(pk[0]), "v"(v[4]), "v"(v[2]), "v"(v[3]), "v"(ones));)
        } else if (MODE == 13) {  // v_perm_b32 (bf16 pack by truncation)
            REP32(asm volatile("v_perm_b32 %0, %1, %2, %4\n\tv_perm_b32 %3, %2, %1, %4" : "=v"(pk[0]), "+v"(v[0]), "+v"(v[1]), "=v"(pk[1]) : "v"(sel));
                  asm volatile("v_perm_b32 %0, %1, %2, %4\n\tv_perm_b32 %3, %2, %1, %4" : "=v"(pk[2]), "+v"(v[2]), "+v"(v[3]), "=v"(pk[3]) : "v"(sel));)
        } else if (MODE == 15) {  // the pk_add chunk x4 + 2 ds_read_b128 (the kernel's ratio: 16 fragment reads per 32 chunks)
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tds_read_b128 %14, %16\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_pk_add_f32 %9, %9, %13\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_pk_add_f32 %9, %9, %12\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tds_read_b128 %15, %16 offset:4096\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\tv_pk_add_f32 %9, %9, %13\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_pk_add_f32 %9, %9, %12\n\ts_waitcnt lgkmcnt(0)"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(pk[0]), "v"(d2[0]), "v"(v[2]), "v"(v[3]), "v"(d2[1]), "v"(d2[2]), "v"(F0), "v"(F1), "v"(lds_addr) : "memory");)
        } else if (MODE == 14) {  // chunk: mfma + 2 exp + cvt only (no sums)
            REP32(asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\t"
                "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_cvt_pk_bf16_f32 %8, %10, %11\n\t"
                "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_cvt_pk_bf16_f32 %8, %6, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B), "v"(v[0]), "v"(v[1]), "v"(pk[0]), "v"(v[4]), "v"(v[2]), "v"(v[3]));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int a = 0; a < 4; ++a) s += acc[a][threadIdx.x & 15];
    s += (float)(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]);
    for (int i = 0; i < 4; ++i) s += d2[i][0] + d2[i][1];
    if (s == 12345.678f) sink[0] = s;
}


// The conv kernel's chunk without DMA and barriers: 6 ds_read_b128 (next fragments) + 8 independent MFMAs (current fragments),
// 1 or 2 waves per SIMD.  Cycles per chunk against 8 x 32 = 256 (x2 with a partner wave on the SIMD).
template <int ORDER>
__global__ __launch_bounds__(512, 1) void k_conv_chunk(unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)(size_t)lds + (wave & 3) * 8192 + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 fa[6], fb[6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { fa[i][j] = (__bf16)1.0f; fb[i][j] = (__bf16)1.0f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (ORDER == 0) {           // reads, then the MFMA burst (the kernel's order)
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:4096\n\tds_read_b128 %2, %6 offset:8192\n\t"
                             "ds_read_b128 %3, %6 offset:12288\n\tds_read_b128 %4, %6 offset:32768\n\tds_read_b128 %5, %6 offset:36864"
                             : "=v"(fb[0]), "=v"(fb[1]), "=v"(fb[2]), "=v"(fb[3]), "=v"(fb[4]), "=v"(fb[5]) : "v"(addr) : "memory");
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, %10, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %9, %10, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %8, %11, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %9, %11, %3\n\t"
                             "v_mfma_f32_32x32x16_bf16 %4, %8, %12, %4\n\tv_mfma_f32_32x32x16_bf16 %5, %9, %12, %5\n\t"
                             "v_mfma_f32_32x32x16_bf16 %6, %8, %13, %6\n\tv_mfma_f32_32x32x16_bf16 %7, %9, %13, %7"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                             : "v"(fa[4]), "v"(fa[5]), "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {                    // one read behind each of the first six MFMAs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %14, %16, %0\n\tds_read_b128 %8, %20\n\t"
                             "v_mfma_f32_32x32x16_bf16 %1, %15, %16, %1\n\tds_read_b128 %9, %20 offset:4096\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %14, %17, %2\n\tds_read_b128 %10, %20 offset:8192\n\t"
                             "v_mfma_f32_32x32x16_bf16 %3, %15, %17, %3\n\tds_read_b128 %11, %20 offset:12288\n\t"
                             "v_mfma_f32_32x32x16_bf16 %4, %14, %18, %4\n\tds_read_b128 %12, %20 offset:32768\n\t"
                             "v_mfma_f32_32x32x16_bf16 %5, %15, %18, %5\n\tds_read_b128 %13, %20 offset:36864\n\t"
                             "v_mfma_f32_32x32x16_bf16 %6, %14, %19, %6\n\tv_mfma_f32_32x32x16_bf16 %7, %15, %19, %7\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                               "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3]), "=&v"(fb[4]), "=&v"(fb[5])
                             : "v"(fa[4]), "v"(fa[5]), "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(addr) : "memory");
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) { bf16x8 t = fa[i]; fa[i] = fb[i]; fb[i] = t; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0.f;
    for (int a = 0; a < 8; ++a) s += acc[a][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

template <int ORDER>
static void run_conv_chunk(const char* name, int waves) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
    hipMemset(d, 0, 256 * 8 * 8);
    const int iters = 50;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_conv_chunk<ORDER>, dim3(256), dim3(64 * waves), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) lo += h[b * 8 + w]; for (int w = 4; w < waves; ++w) hi += h[b * 8 + w]; }
    printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_chunk_waves0_3\": %.1f, \"ticks_per_chunk_waves4_7\": %.1f}\n", name, waves / 4,
           lo / (256 * 4) / iters / 4, waves > 4 ? hi / (256 * (waves - 4)) / iters / 4 : 0.0);
    hipFree(d); hipFree(sink);
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

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "conv")) {
        run_conv_chunk<0>("conv chunk: 6 ds_read_b128 then 8 MFMA", 4);
        run_conv_chunk<0>("conv chunk: 6 ds_read_b128 then 8 MFMA", 8);
        run_conv_chunk<1>("conv chunk: reads behind the MFMAs", 4);
        run_conv_chunk<1>("conv chunk: reads behind the MFMAs", 8);
        return 0;
    }
    run<0>("v_exp_f32 x128 (per instr)", 128, 1);
    run<1>("v_add_f32 x128 (per instr)", 128, 1);
    run<2>("v_cvt_pk_bf16_f32 x128 (per instr)", 128, 1);
    run<3>("mfma 32x32x16 bf16 x128 (per mfma)", 128, 1);
    run<4>("chunk: mfma + 2 exp + cvt + 2 add, x128 (per chunk)", 128, 1);
    run<5>("chunk: mfma + 2 mul + cvt + 2 add, x128 (per chunk)", 128, 1);
    run<6>("chunk: mfma + 2 exp, x128 (per chunk)", 128, 1);
    run<7>("chunk: mfma + 4 exp, x128 (per chunk)", 128, 1);
    run<8>("chunk: mfma + 6 add, x128 (per chunk)", 128, 1);
    run<9>("v_pk_add_f32 x128 (per instr)", 128, 1);
    run<10>("v_dot2_f32_bf16 x128 (per instr)", 128, 1);
    run<11>("chunk: mfma + 2 exp + cvt + 1 pk_add, x128 (per chunk)", 128, 1);
    run<12>("chunk: mfma + 2 exp + cvt + 1 dot2, x128 (per chunk)", 128, 1);
    run<13>("v_perm_b32 x128 (per instr)", 128, 1);
    run<14>("chunk: mfma + 2 exp + cvt, x128 (per chunk)", 128, 1);
    run<15>("chunk (pk_add) + ds_read_b128 every 2nd chunk, x128 (per chunk)", 128, 1);
    return 0;
}
