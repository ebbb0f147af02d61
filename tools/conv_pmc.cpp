// Stand-alone driver for hardware-counter runs of the implicit-GEMM conv kernel:
//   hipcc --offload-arch=gfx950 -O2 tools/conv_pmc.cpp -o tools/_conv_pmc -Ldreammat_amd -ldreammat_hip -Wl,-rpath,$PWD/dreammat_amd
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <out> --output-format csv -- tools/_conv_pmc 8 512 512 128 128 5
// (`rocprofv3 --pmc` segfaults under python + torch in this image -- profiles/r01_pmc_attempt_segfault.log -- so the
// counters are collected on the C ABI directly; no torch, no python in the process.)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/dreammat_hip.h"

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("hip error %d at %s:%d\n", (int)r, __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    if (argc < 7) { printf("usage: %s B H W Cin Cout iters\n", argv[0]); return 1; }
    const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    const int iters = atoi(argv[6]);
    const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)B * H * W * Cout;
    std::vector<unsigned short> hx(nx), hw(nw);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(0x3c00u + ((s >> 16) & 0x1ffu) + ((s >> 31) << 15)); };  // bf16 ~ +-[0.0078, 0.03]
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    void *dx, *dw, *dy;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 2));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rc = dm_conv3x3_nhwc_bf16(dx, dw, nullptr, dy, B, H, W, Cin, H, W, Cout, 1, 1, 1, nullptr);   // warm-up
    if (rc) { printf("dm_conv3x3_nhwc_bf16 rc=%d\n", rc); return 3; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) dm_conv3x3_nhwc_bf16(dx, dw, nullptr, dy, B, H, W, Cin, H, W, Cout, 1, 1, 1, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 2.0 * B * H * W * Cout * 9.0 * Cin;
    printf("{\"B\":%d,\"H\":%d,\"W\":%d,\"Cin\":%d,\"Cout\":%d,\"ms\":%.4f,\"TFLOPs\":%.1f,\"alg_in_MB\":%.1f,\"alg_out_MB\":%.1f}\n", B, H, W, Cin,
           Cout, ms / iters, flops / (ms / iters * 1e-3) / 1e12, (nx + nw) * 2 / 1e6, ny * 2 / 1e6);
    return 0;
}
