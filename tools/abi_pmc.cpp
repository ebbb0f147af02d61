// Stand-alone C-ABI driver for hardware-counter / kernel-trace runs (no python, no torch in the process):
//   hipcc --offload-arch=gfx950 -O2 tools/abi_pmc.cpp -o tools/_abi_pmc -Ldreammat_amd -ldreammat_hip -Wl,-rpath,'$ORIGIN/../dreammat_amd'
//   tools/_abi_pmc conv  B H W Cin Cout iters        3x3 conv, stride 1, pad 1, NHWC bf16
//   tools/_abi_pmc attn  B heads Sq Skv D iters      attention forward, bf16
//   tools/_abi_pmc shade N n_env iters               split-sum shade forward + backward over N covered pixels
//   tools/_abi_pmc shadef FILE iters                 the same on a dumped G-buffer + packed atlas (tools/r2_probe.py writes
//                                                    /tmp/shade_case_*.bin: the step's REAL coherent G-buffer)
//   tools/_abi_pmc mc N n_lon n_lat grid|bvh iters   Monte-Carlo shading forward over N surface points of a displaced sphere
//   (attn takes an optional 8th argument: the kernel variant name for dm_attention_select)
// Prints one JSON line with the HIP-event time per launch.  See tools/pmc_abi.sh for the rocprofv3 passes.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/dreammat_hip.h"

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)r_, __FILE__, __LINE__); return 2; } } while (0)
#define DM(e) do { int r_ = (e); if (r_) { printf("dm error %d at %s:%d\n", r_, __FILE__, __LINE__); return 3; } } while (0)

static unsigned g_seed = 12345u;
static unsigned rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed; }
static float frand() { return (rnd() >> 8) * (1.0f / 16777216.0f); }                       // [0,1)
static unsigned short bf16_small() { unsigned r = rnd(); return (unsigned short)(0x3c00u + ((r >> 16) & 0x1ffu) + ((r >> 31) << 15)); }
static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x8000u) >> 16); }

template <class T> static int upload(void** d, const std::vector<T>& h) {
    CK(hipMalloc(d, h.size() * sizeof(T)));
    CK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

template <class F> static int timed(int iters, float* ms_per, F&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    DM(f());                                             // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per = ms / iters;
    return 0;
}

static int run_conv(int B, int H, int W, int Cin, int Cout, int iters) {
    const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)B * H * W * Cout;
    std::vector<unsigned short> hx(nx), hw(nw);
    for (auto& v : hx) v = bf16_small();
    for (auto& v : hw) v = bf16_small();
    void *dx, *dw, *dy;
    if (upload(&dx, hx) || upload(&dw, hw)) return 2;
    CK(hipMalloc(&dy, ny * 2));
    float ms;
    int rc = timed(iters, &ms, [&] { return dm_conv3x3_nhwc_bf16(dx, dw, nullptr, dy, B, H, W, Cin, H, W, Cout, 1, 1, 1, nullptr); });
    if (rc) return rc;
    const double flops = 2.0 * B * H * W * Cout * 9.0 * Cin;
    printf("{\"op\":\"conv\",\"B\":%d,\"H\":%d,\"W\":%d,\"Cin\":%d,\"Cout\":%d,\"ms\":%.4f,\"TFLOPs\":%.1f,\"alg_in_MB\":%.1f,\"alg_out_MB\":%.1f}\n",
           B, H, W, Cin, Cout, ms, flops / (ms * 1e-3) / 1e12, (nx + nw) * 2 / 1e6, ny * 2 / 1e6);
    return 0;
}

static int run_gemm(long long M, int K, int N, int geglu, int iters) {
    const size_t nx = (size_t)M * K, nw = (size_t)N * K, ny = (size_t)M * (geglu ? N / 2 : N);
    std::vector<unsigned short> hx(nx), hw(nw), hb(N);
    for (auto& v : hx) v = bf16_small();
    for (auto& v : hw) v = bf16_small();
    for (auto& v : hb) v = bf16_small();
    void *dx, *dw, *db, *dy, *dr;
    if (upload(&dx, hx) || upload(&dw, hw) || upload(&db, hb)) return 2;
    CK(hipMalloc(&dy, ny * 2));
    CK(hipMalloc(&dr, ny * 2));
    CK(hipMemset(dr, 0, ny * 2));
    float ms;
    int rc = timed(iters, &ms, [&] { return dm_gemm_bf16_fused(dx, dw, db, geglu ? nullptr : dr, dy, M, K, N, geglu, nullptr); });
    if (rc) return rc;
    const double flops = 2.0 * M * K * (double)N;
    printf("{\"op\":\"gemm\",\"M\":%lld,\"K\":%d,\"N\":%d,\"geglu\":%d,\"ms\":%.4f,\"TFLOPs\":%.1f,\"alg_MB\":%.1f,\"GBps\":%.0f}\n", M, K, N,
           geglu, ms, flops / (ms * 1e-3) / 1e12, (nx + nw + ny * (geglu ? 1 : 2)) * 2 / 1e6,
           (nx + nw + ny * (geglu ? 1 : 2)) * 2 / 1e9 / (ms * 1e-3));
    return 0;
}

static int run_attn(int B, int Hh, int Sq, int Skv, int D, int iters) {
    const int C = Hh * D, Sp = (Skv + 7) / 8 * 8;
    std::vector<unsigned short> hq((size_t)B * Sq * C), hk((size_t)B * Skv * C), hv((size_t)B * C * Sp);
    for (auto& v : hq) v = bf16_of(frand() * 2.f - 1.f);
    for (auto& v : hk) v = bf16_of(frand() * 2.f - 1.f);
    for (auto& v : hv) v = bf16_of(frand() * 2.f - 1.f);
    void *dq, *dk, *dv, *dout;
    if (upload(&dq, hq) || upload(&dk, hk) || upload(&dv, hv)) return 2;
    CK(hipMalloc(&dout, hq.size() * 2));
    float ms;
    int rc = timed(iters, &ms, [&] {
        return dm_attention_fwd_bf16(dq, dk, dv, dout, B, Hh, Sq, Skv, D, (long long)Sq * C, C, D, (long long)Skv * C, C, D,
                                     (long long)C * Sp, (long long)D * Sp, Sp, (long long)Sq * C, C, D, 1.0f / sqrtf((float)D), nullptr);
    });
    if (rc) return rc;
    const double flops = 4.0 * B * Sq * (double)Skv * C;
    printf("{\"op\":\"attn\",\"B\":%d,\"heads\":%d,\"Sq\":%d,\"Skv\":%d,\"D\":%d,\"ms\":%.4f,\"TFLOPs\":%.1f,\"alg_MB\":%.1f}\n", B, Hh, Sq, Skv, D,
           ms, flops / (ms * 1e-3) / 1e12, (2.0 * hq.size() + hk.size() + hv.size()) * 2 / 1e6);
    return 0;
}

static int run_shade(long long N, int n_env, int iters) {
    N &= ~3LL;                               // whole 16-byte groups per SoA channel (the kernels' fast path; see run_shade_file)
    const long long Np = N;
    // atlas: envlight geometry (max_res 128 -> min_res 16, diffuse 16), every face with a 1-texel border, RGBA fp32
    dm_env_atlas at;
    memset(&at, 0, sizeof(at));
    const int res[4] = {128, 64, 32, 16};
    long long off = 0;
    for (int m = 0; m < 4; ++m) { at.mip_off[m] = off; at.mip_res[m] = res[m]; off += 6LL * (res[m] + 2) * (res[m] + 2); }
    at.n_mips = 4; at.spec_env_stride = off; at.diff_res = 16; at.diff_env_stride = 6LL * 18 * 18; at.lut_res = 256;
    at.min_rough_mip = 0.08f; at.max_rough_mip = 0.5f;
    std::vector<float> hspec((size_t)n_env * off * 4), hdiff((size_t)n_env * at.diff_env_stride * 4), hlut(256 * 256 * 2);
    for (auto& v : hspec) v = frand() * 2.f;
    for (auto& v : hdiff) v = frand();
    for (auto& v : hlut) v = frand();
    void *dspec, *ddiff, *dlut;
    if (upload(&dspec, hspec) || upload(&ddiff, hdiff) || upload(&dlut, hlut)) return 2;
    at.spec = (const float*)dspec; at.diff = (const float*)ddiff; at.fg_lut = (const float*)dlut;
    dm_mat_cfg mc = {0.0f, 0.9f, 0.08f, 0.9f};
    const int views = 8, HW = 512 * 512;
    std::vector<float> hn(3 * N), hv(3 * N), hf(5 * N), hg(3 * N);
    std::vector<int> hp(N), henv(views);
    for (long long i = 0; i < N; ++i) {
        float n[3] = {frand() - .5f, frand() - .5f, frand() + .2f}, v[3] = {frand() - .5f, frand() - .5f, frand() + .2f};
        float ln = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), lv = 1.f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int c = 0; c < 3; ++c) { hn[c * N + i] = n[c] * ln; hv[c * N + i] = v[c] * lv; hg[c * N + i] = frand() - .5f; }
        for (int c = 0; c < 5; ++c) hf[c * N + i] = frand() * 4.f - 2.f;
        hp[i] = (int)((double)i * views * HW / N);
    }
    for (int v = 0; v < views; ++v) henv[v] = v % n_env;
    std::vector<int> hcount(1, (int)N);
    void *dn, *dv, *df, *dg, *dp, *denv, *dcount, *dcol, *ddf;
    if (upload(&dn, hn) || upload(&dv, hv) || upload(&df, hf) || upload(&dg, hg) || upload(&dp, hp) || upload(&denv, henv) || upload(&dcount, hcount)) return 2;
    CK(hipMalloc(&dcol, 3 * Np * 4)); CK(hipMalloc(&ddf, 5 * Np * 4));
    float ms_f, ms_b;
    int rc = timed(iters, &ms_f, [&] {
        return dm_shade_fwd(&at, &mc, (float*)dn, 1, Np, (float*)dv, 1, Np, (float*)df, 1, Np, (int*)dp, (int*)denv, (int*)dcount, N, HW, views,
                            (float*)dcol, 1, Np, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    });
    if (rc) return rc;
    rc = timed(iters, &ms_b, [&] {
        return dm_shade_bwd(&at, &mc, (float*)dn, 1, Np, (float*)dv, 1, Np, (float*)df, 1, Np, (int*)dp, (int*)denv, (int*)dcount, N, HW, views,
                            (float*)dg, 1, Np, (float*)ddf, 1, Np, nullptr);
    });
    if (rc) return rc;
    printf("{\"op\":\"shade\",\"N\":%lld,\"n_env\":%d,\"fwd_ms\":%.4f,\"bwd_ms\":%.4f,\"fwd_GBps\":%.0f,\"bwd_GBps\":%.0f,\"alg_fwd_MB\":%.1f,\"alg_bwd_MB\":%.1f}\n",
           N, n_env, ms_f, ms_b, 56.0 * N / (ms_f * 1e-3) / 1e9, 76.0 * N / (ms_b * 1e-3) / 1e9, 56.0 * N / 1e6, 76.0 * N / 1e6);
    return 0;
}

// Monte-Carlo shading forward (row f-1) on the bench mesh's shape: displaced UV sphere (n_lon x n_lat), N surface points at
// triangle centroids, 200 + 128 directions, 5 random lat-long environments; tracer = "grid" | "bvh".
static int run_mc(long long N, int n_lon, int n_lat, const char* tracer, int iters) {
    const float radius = 0.8f, amp = 0.15f, pi = 3.14159265358979f;
    const int nr = n_lat - 1;
    std::vector<float> v((size_t)(nr * n_lon + 2) * 3);
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < n_lon; ++j) {
            const float t = pi * (i + 1) / n_lat, p = 2 * pi * j / n_lon, r = radius * (1 + amp * sinf(5 * t) * sinf(7 * p));
            float* o = &v[(size_t)(i * n_lon + j) * 3];
            o[0] = r * sinf(t) * cosf(p); o[1] = r * sinf(t) * sinf(p); o[2] = r * cosf(t);
        }
    const int i_n = nr * n_lon, i_s = i_n + 1;
    v[3 * i_n + 2] = radius; v[3 * i_s + 2] = -radius;
    std::vector<int> f;
    for (int j = 0; j < n_lon; ++j) { f.push_back(i_n); f.push_back(j); f.push_back((j + 1) % n_lon); }
    for (int i = 0; i + 1 < nr; ++i)
        for (int j = 0; j < n_lon; ++j) {
            const int jn = (j + 1) % n_lon, a = i * n_lon + j, b = i * n_lon + jn, c = (i + 1) * n_lon + j, d = (i + 1) * n_lon + jn;
            f.push_back(a); f.push_back(c); f.push_back(b); f.push_back(b); f.push_back(c); f.push_back(d);
        }
    for (int j = 0; j < n_lon; ++j) { f.push_back((nr - 1) * n_lon + j); f.push_back(i_s); f.push_back((nr - 1) * n_lon + (j + 1) % n_lon); }
    const int n_tri = (int)f.size() / 3, n_vert = (int)v.size() / 3;
    std::vector<int> nodes((size_t)2 * n_tri * 8), order(n_tri), nodes4((size_t)2 * n_tri * 32);
    std::vector<float> tris((size_t)n_tri * 12);
    int n_nodes = 0, n_nodes4 = 0;
    DM(dm_bvh_build(v.data(), n_vert, f.data(), n_tri, nodes.data(), tris.data(), order.data(), &n_nodes));
    DM(dm_bvh_collapse4(nodes.data(), n_nodes, nodes4.data(), &n_nodes4));
    nodes.resize((size_t)n_nodes * 8); nodes4.resize((size_t)n_nodes4 * 32);
    dm_grid g;
    uint32_t* blob = nullptr;
    int64_t words = 0;
    DM(dm_grid_build(tris.data(), n_tri, 0, &g, &blob, &words));
    void *dnodes, *dnodes4, *dtris, *dblob;
    if (upload(&dnodes, nodes) || upload(&dnodes4, nodes4) || upload(&dtris, tris)) return 2;
    CK(hipMalloc(&dblob, (size_t)words * 4));
    CK(hipMemcpy(dblob, blob, (size_t)words * 4, hipMemcpyHostToDevice));
    dm_host_free(blob);
    auto pad4 = [](long long w) { return (w + 3) / 4 * 4; };
    const long long n_blocks = (long long)((g.dim[0] + 1) / 2) * ((g.dim[1] + 1) / 2) * ((g.dim[2] + 1) / 2);
    const long long o_sb = pad4(g.n_words), o_off = o_sb + pad4((g.n_words + 63) / 64), o_dist = o_off + pad4((g.n_words + 1) / 2),
                    o_occ = o_dist + pad4((n_blocks + 7) / 8), o_tri = o_occ + pad4(g.n_occ + 1);
    const uint32_t* b32 = (const uint32_t*)dblob;
    g.bits = b32; g.sbase = b32 + o_sb; g.off16 = (const uint16_t*)(b32 + o_off); g.dist4 = (const uint8_t*)(b32 + o_dist);
    g.occ_start = b32 + o_occ; g.cell_tris = (const float*)(b32 + o_tri);
    // points: centroids of random triangles pushed 1e-4 along the outward normal, view directions around it
    std::vector<float> hp(3 * N), hn(3 * N), hv(3 * N), hf(5 * N), hrd(N), hrs(N);
    std::vector<int> hpix(N), henv(8);
    for (long long i = 0; i < N; ++i) {
        const int t = (int)(rnd() % (unsigned)n_tri);
        const float* a = &v[3 * (size_t)f[3 * t]]; const float* b = &v[3 * (size_t)f[3 * t + 1]]; const float* c = &v[3 * (size_t)f[3 * t + 2]];
        float cen[3], e1[3], e2[3], n[3];
        for (int k = 0; k < 3; ++k) { cen[k] = (a[k] + b[k] + c[k]) / 3.f; e1[k] = b[k] - a[k]; e2[k] = c[k] - a[k]; }
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        float ln = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if ((n[0] * cen[0] + n[1] * cen[1] + n[2] * cen[2]) < 0) ln = -ln;
        float vv[3], lv = 0;
        for (int k = 0; k < 3; ++k) { n[k] *= ln; vv[k] = n[k] + 0.6f * (frand() * 2 - 1); lv += vv[k] * vv[k]; }
        lv = 1.f / sqrtf(lv);
        for (int k = 0; k < 3; ++k) { hp[3 * i + k] = cen[k] + 1e-4f * n[k]; hn[3 * i + k] = n[k]; hv[3 * i + k] = vv[k] * lv; }
        for (int k = 0; k < 5; ++k) hf[5 * i + k] = frand() * 4.f - 2.f;
        hrd[i] = frand(); hrs[i] = frand();
        hpix[i] = (int)(i % 8) * 512 * 512;
    }
    for (int k = 0; k < 8; ++k) henv[k] = k % 5;
    const int lh = getenv("MC_LIGHT_H") ? atoi(getenv("MC_LIGHT_H")) : 512, lw = 2 * lh, nd = 200, ns = 128;   // (MC_LIGHT_H: cache-footprint experiments)
    std::vector<float> hl((size_t)5 * lh * lw * 3), sd(2 * nd), ss(2 * ns);
    for (auto& x : hl) x = frand();
    auto fib = [&](std::vector<float>& o, int n) {
        for (int k = 0; k < n; ++k) {
            const double m = n + k, z = 2.0 * m / (2 * n) - 1.0, az = fmod(2 * 3.14159265358979 * m * ((sqrt(5.0) - 1) / 2), 2 * 3.14159265358979);
            o[2 * k] = (float)(az * 0.5 / 3.14159265358979); o[2 * k + 1] = (float)(1 - 2 * asin(z) / 3.14159265358979);
        }
    };
    fib(sd, nd); fib(ss, ns);
    std::vector<int> hcount(1, (int)N);
    void *dp, *dn, *dv, *df, *drd, *drs, *dpix, *denv, *dcount, *dl, *dsd, *dss, *dbits, *dcol;
    if (upload(&dp, hp) || upload(&dn, hn) || upload(&dv, hv) || upload(&df, hf) || upload(&drd, hrd) || upload(&drs, hrs) || upload(&dpix, hpix) ||
        upload(&denv, henv) || upload(&dcount, hcount) || upload(&dl, hl) || upload(&dsd, sd) || upload(&dss, ss)) return 2;
    const int hw = dm_mc_hit_words(nd, ns);
    CK(hipMalloc(&dbits, (size_t)N * hw * 4)); CK(hipMalloc(&dcol, (size_t)3 * N * 4));
    dm_mc_scene sc;
    memset(&sc, 0, sizeof(sc));
    sc.bvh_nodes = dnodes; sc.bvh_tris = (const float*)dtris; sc.lights = (const float*)dl; sc.n_env = 5; sc.light_h = lh; sc.light_w = lw;
    sc.samples_diffuse = (const float*)dsd; sc.samples_specular = (const float*)dss; sc.n_diffuse = nd; sc.n_specular = ns;
    sc.geometry_ggx_smith = 0; sc.bvh_nodes4 = dnodes4; sc.grid = strcmp(tracer, "bvh") ? &g : nullptr;
    dm_mat_cfg mc = {0.0f, 0.9f, 0.01f, 0.9f};
    float ms;
    int rc = timed(iters, &ms, [&] {
        return dm_mc_shade_fwd(&sc, &mc, (float*)dp, 3, 1, (float*)dn, 3, 1, (float*)dv, 3, 1, (float*)df, 5, 1, (int*)dpix, (int*)denv, (int*)dcount, N,
                               512 * 512, (float*)drd, (float*)drs, (uint32_t*)dbits, (float*)dcol, 1, N, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr);
    });
    if (rc) return rc;
    printf("{\"op\":\"mc\",\"tracer\":\"%s\",\"N\":%lld,\"tris\":%d,\"grid_dim\":[%d,%d,%d],\"rays\":%lld,\"fwd_ms\":%.4f,\"Grays_per_s\":%.3f}\n", tracer, N, n_tri,
           g.dim[0], g.dim[1], g.dim[2], N * (nd + ns), ms, N * (nd + ns) / (ms * 1e-3) / 1e9);
    return 0;
}

// File layout (little endian, written by tools/r2_probe.py): int64 header[16] = {magic 0x444d5348, N, views, HW, n_mips, diff_res,
// lut_res, texel_format, spec_env_stride, diff_env_stride, spec_bytes, diff_bytes, has_pairs, 0, 0, 0}; int64 mip_off[8];
// int32 mip_res[8]; then nrm[3][N] view[3][N] feat[5][N] dcol[3][N] (f32), pix[N] env_of_view[views] (i32), spec, diff,
// fg_lut[lut_res^2*2] f32, fg_pairs[lut_res*(lut_res+1)*4] f32 (if has_pairs).
static int run_shade_file(const char* path, int iters) {
    FILE* fh = fopen(path, "rb");
    if (!fh) { printf("cannot open %s\n", path); return 1; }
    long long hd[16], mip_off[8];
    int mip_res[8];
    if (fread(hd, 8, 16, fh) != 16 || hd[0] != 0x444d5348LL || fread(mip_off, 8, 8, fh) != 8 || fread(mip_res, 4, 8, fh) != 8) return 1;
    const long long N = hd[1];
    const int views = (int)hd[2], HW = (int)hd[3];
    auto rd = [&](size_t bytes, void** dptr) -> int {
        std::vector<char> h(bytes);
        if (fread(h.data(), 1, bytes, fh) != bytes) { printf("short read\n"); return 1; }
        return upload(dptr, h);
    };
    // SoA row tensors [C][N] of the file -> device tensors with the channel pitch rounded up to 4 floats: the shade kernels' fast
    // path streams 16 bytes per lane and (since round 5) requires every channel to start on a 16-byte boundary, which is what
    // dreammat_amd/hipops.py allocates
    const long long Np = (N + 3) / 4 * 4;
    auto rd_soa = [&](int C, void** dptr) -> int {
        std::vector<float> h((size_t)C * N), hp((size_t)C * Np, 0.f);
        if (fread(h.data(), 4, (size_t)C * N, fh) != (size_t)C * N) { printf("short read\n"); return 1; }
        for (int c = 0; c < C; ++c) memcpy(hp.data() + (size_t)c * Np, h.data() + (size_t)c * N, (size_t)N * 4);
        return upload(dptr, hp);
    };
    void *dn, *dv, *df, *dg, *dp, *denv, *dspec, *ddiff, *dlut, *dpairs = nullptr, *dcount, *dcol, *ddf;
    if (rd_soa(3, &dn) || rd_soa(3, &dv) || rd_soa(5, &df) || rd_soa(3, &dg) || rd(4 * N, &dp) || rd(4 * views, &denv) ||
        rd(hd[10], &dspec) || rd(hd[11], &ddiff) || rd((size_t)hd[6] * hd[6] * 8, &dlut))
        return 2;
    if (hd[12] && rd((size_t)hd[6] * (hd[6] + 1) * 16, &dpairs)) return 2;
    fclose(fh);
    dm_env_atlas at;
    memset(&at, 0, sizeof(at));
    at.spec = (const float*)dspec; at.diff = (const float*)ddiff; at.fg_lut = (const float*)dlut; at.fg_pairs = (const float*)dpairs;
    at.spec_env_stride = hd[8]; at.diff_env_stride = hd[9];
    for (int i = 0; i < 8; ++i) { at.mip_off[i] = mip_off[i]; at.mip_res[i] = mip_res[i]; }
    at.n_mips = (int)hd[4]; at.diff_res = (int)hd[5]; at.lut_res = (int)hd[6]; at.texel_format = (int)hd[7];
    at.min_rough_mip = 0.08f; at.max_rough_mip = 0.5f;
    dm_mat_cfg mc = {0.0f, 0.9f, 0.1f, 0.95f};
    std::vector<int> hcount(1, (int)N);
    if (upload(&dcount, hcount)) return 2;
    CK(hipMalloc(&dcol, 3 * Np * 4)); CK(hipMalloc(&ddf, 5 * Np * 4));
    float ms_f, ms_b;
    int rc = timed(iters, &ms_f, [&] {
        return dm_shade_fwd(&at, &mc, (float*)dn, 1, Np, (float*)dv, 1, Np, (float*)df, 1, Np, (int*)dp, (int*)denv, (int*)dcount, N, HW, views,
                            (float*)dcol, 1, Np, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    });
    if (rc) return rc;
    rc = timed(iters, &ms_b, [&] {
        return dm_shade_bwd(&at, &mc, (float*)dn, 1, Np, (float*)dv, 1, Np, (float*)df, 1, Np, (int*)dp, (int*)denv, (int*)dcount, N, HW, views,
                            (float*)dg, 1, Np, (float*)ddf, 1, Np, nullptr);
    });
    if (rc) return rc;
    printf("{\"op\":\"shadef\",\"N\":%lld,\"texel_format\":%d,\"fg_pairs\":%d,\"fwd_ms\":%.4f,\"bwd_ms\":%.4f,\"fwd_GBps\":%.0f,\"bwd_GBps\":%.0f,\"alg_fwd_MB\":%.1f,\"alg_bwd_MB\":%.1f}\n",
           N, at.texel_format, dpairs ? 1 : 0, ms_f, ms_b, 56.0 * N / (ms_f * 1e-3) / 1e9, 76.0 * N / (ms_b * 1e-3) / 1e9, 56.0 * N / 1e6, 76.0 * N / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 7 && !strcmp(argv[1], "gemm")) return run_gemm(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    if (argc >= 8 && !strcmp(argv[1], "conv")) return run_conv(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
    if (argc >= 9 && !strcmp(argv[1], "attn") && dm_attention_select(argv[8])) { printf("unknown attention variant %s\n", argv[8]); return 1; }
    if (argc >= 4 && !strcmp(argv[1], "shadef")) return run_shade_file(argv[2], atoi(argv[3]));
    if (argc >= 8 && !strcmp(argv[1], "attn")) return run_attn(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
    if (argc >= 7 && !strcmp(argv[1], "mc")) return run_mc(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), argv[5], atoi(argv[6]));
    if (argc >= 5 && !strcmp(argv[1], "shade")) return run_shade(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]));
    printf("usage: %s conv B H W Cin Cout iters | attn B heads Sq Skv D iters | shade N n_env iters | mc N n_lon n_lat grid|bvh iters\n", argv[0]);
    return 1;
}
