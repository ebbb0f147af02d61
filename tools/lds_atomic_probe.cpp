// Micro-benchmark: LDS atomic-add throughput on MI355X for k_hg_acc's access pattern (1024 threads, random entries of a
// 16384-entry x 2 table in 128 KB of LDS): ds_add_f32 vs ds_add_u32 vs ds_add_u64 (fixed point) vs plain (racy) ds read/add/write.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_probe.cpp -o tools/_lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("hip error %d line %d\n", (int)r_, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(const unsigned* idx, int per_thread, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tf = reinterpret_cast<float*>(smem);
    unsigned* tu = reinterpret_cast<unsigned*>(smem);
    unsigned long long* tq = reinterpret_cast<unsigned long long*>(smem);
    for (int i = threadIdx.x; i < 32768; i += 1024) tu[i] = 0;
    __syncthreads();
    const unsigned* p = idx + ((size_t)blockIdx.x * 1024 + threadIdx.x) * per_thread;
    for (int k = 0; k < per_thread; ++k) {
        const unsigned e = p[k] & 16383u;
        if (MODE == 0) { atomicAdd(&tf[2 * e], 1.0f); atomicAdd(&tf[2 * e + 1], 0.5f); }
        if (MODE == 1) { atomicAdd(&tu[2 * e], 3u); atomicAdd(&tu[2 * e + 1], 5u); }
        if (MODE == 2) { atomicAdd(&tq[e], 0x0000000500000003ull); }                      // both values in one 64-bit add
        if (MODE == 3) { tf[2 * e] += 1.0f; tf[2 * e + 1] += 0.5f; }                      // racy: the LDS traffic without atomicity
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 32768; i += 1024) s += tf[i];
    if (s == 123.456f) out[0] = s;
}

int main() {
    const int blocks = 1024, per_thread = 64;
    const size_t n = (size_t)blocks * 1024 * per_thread;
    std::vector<unsigned> h(n);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    unsigned* didx; float* out;
    CK(hipMalloc(&didx, n * 4)); CK(hipMemcpy(didx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(M) do { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); \
    for (int it = 0; it < 3; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds<M>, dim3(blocks), dim3(1024), 131072, 0, didx, per_thread, out); \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
      if (it == 2) printf("{\"mode\":%d,\"ms\":%.3f,\"G_tuples_per_s\":%.1f}\n", M, ms, (double)n / ms / 1e6); } } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3);
    return 0;
}
