// Micro-benchmark: fp32 global atomic-add throughput on MI355X for the hash-grid backward's access pattern (random entries of a
// 4 MB level table, two adjacent floats per request pair) -- shared table vs one private table per XCD (selected by the XCC the
// workgroup really runs on), and 1 vs 2 atomics per lane.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("hip error %d line %d\n", (int)r_, __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); return id & 15u; }

template <int MODE>   // 0: shared table, 1: private table per XCC, 2: shared, lane pairs hit the two floats of one entry
__global__ __launch_bounds__(256) void k_atomic(float* table, unsigned entries, const unsigned* idx, long long n, int reps) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* base = table + (MODE == 1 ? (size_t)xcc_id() * entries * 2 : 0);
    for (int r = 0; r < reps; ++r) {
        unsigned e = idx[(i + (long long)r * 7919) % n] % entries;
        if (MODE == 2) { atomicAdd(base + 2 * (size_t)(e & ~1u) + 2 * 0 + (threadIdx.x & 1), 1.0f); }
        else { atomicAdd(base + 2 * (size_t)e, 1.0f); atomicAdd(base + 2 * (size_t)e + 1, 0.5f); }
    }
}

int main() {
    const unsigned entries = 1u << 19;
    const long long n = 1 << 22;
    std::vector<unsigned> h(n);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    unsigned* didx; float* table;
    CK(hipMalloc(&didx, n * 4)); CK(hipMemcpy(didx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&table, (size_t)entries * 2 * 4 * 8)); CK(hipMemset(table, 0, (size_t)entries * 2 * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 8;
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(n / 256), dim3(256), 0, 0, table, entries, didx, n, reps);
            if (mode == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(n / 256), dim3(256), 0, 0, table, entries, didx, n, reps);
            if (mode == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(n / 256), dim3(256), 0, 0, table, entries, didx, n, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double atoms = (double)n * reps * (mode == 2 ? 1 : 2);
            if (it == 2) printf("{\"mode\":%d,\"ms\":%.3f,\"G_atomics_per_s\":%.1f}\n", mode, ms, atoms / ms / 1e6);
        }
    }
    // sanity of the private-table mode: the 8 copies must add up to what the shared table would hold
    std::vector<float> t((size_t)entries * 2 * 8);
    CK(hipMemcpy(t.data(), table, t.size() * 4, hipMemcpyDeviceToHost));
    double tot = 0; for (float v : t) tot += v;
    printf("{\"total_mass\":%.1f}\n", tot);
    return 0;
}
