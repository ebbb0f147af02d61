// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds lds[e] = e (16-bit); lane l passes the byte address 8*l (its own
// 4 consecutive elements 4l..4l+3, what a plain ds_read_b64 would return) and prints what the transposing read returns.
// usage: tools/_tr_probe   -> one line per lane: lane: e0 e1 e2 e3   (element indices; source lane = e / 4, source slot = e % 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_probe(uint16_t* out, int lane_stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * lane_stride_bytes;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int stride : {8, 32}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lane stride %d bytes (lane l owns elements %d l .. + 3)\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) printf("%2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
