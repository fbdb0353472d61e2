// Standalone GEMM laboratory (not part of the product): includes the production kernel source and times
// ablated / alternative schedules with HIP events.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I bsi_amd/csrc tools/experiments/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bsi_amd/csrc/gemm_bf16.hip"
#include "gemm_variants.inc"
#include "../../bsi_amd/csrc/bsi_ops.hip"  // bsi_set_error
int g_bsi_cu_reserve = 0;  // compute_cus() of common.h (defined in prof.hip in the library)

template <int EPI, int ABL>
float time_pp(LabParams p, int iters) {
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    const size_t lds = 2 * 512 * ROW_BYTES;
    auto kern = gemm_bf16_pp_kernel<EPI, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, 0, p);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, 0, p);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int EPI, int ABL, int RING = 4>
float time_pring(LabParams p, int iters) {
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    p.gm = 4;
    const size_t lds = RING == 4 ? 4 * 512 * 64 + 32768 : RING * 512 * 64;  // RING 5: no epilogue scratch (ABL & 4 only)
    auto kern = lab_pring_kernel<EPI, ABL, false, RING>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int EPI, int ABL>
float time_w1(LabParams p, int iters) {
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    p.gm = 4;
    const size_t lds = 2 * 512 * 128;
    auto kern = gemm_bf16_w1_kernel<EPI, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

// L2 -> LDS fill-rate microbenchmark: 8 waves per CU issue nothing but global_load_lds_dwordx4 over an L2-resident region.
// seg = bytes each group of lanes reads contiguously (64: lanes 0..3 form a row segment as in the K=32 GEMM stage,
// 128: lanes 0..7 as in a K=64 stage, 1024: the whole wave reads 1 KB contiguously).
// mode 0: global_load_lds; mode 1: buffer_load ... lds (offen, what the convolution uses); mode 2: buffer, every 4th row out of range
__global__ __launch_bounds__(512) void dma_bw_kernel(const char* src, unsigned region, int iters, int seg, int ld, int mode) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lps = seg / 16;                       // lanes per contiguous segment
    const unsigned lane_off = (unsigned)(lane / lps) * ld + (lane % lps) * 16;
    unsigned base = (blockIdx.x * 8 + wave) * 4096u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = (base + q * 1024u * 37u + lane_off) % region;
            if (mode == 0) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(src + (off & ~15u)), LDS_PTR(lds + ((it & 3) * 32 + q * 8 + wave) * 1024), 16, 0, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x80000000u, 0x00020000);
                const unsigned vo = (mode == 2 && ((lane / lps) & 3) == 3) ? 0xFFFFFF00u : (off & ~15u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + ((it & 3) * 32 + q * 8 + wave) * 1024), 16, vo, 0, 0, 0);
            }
        }
        base += 8 * 4096u * 61u;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static void dma_bench() {
    const unsigned region = 2u << 20;  // 2 MB: resident in every XCD's L2
    char* d;
    hipMalloc(&d, region + (1 << 20));
    hipMemset(d, 1, region + (1 << 20));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_bw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 4000;
    struct Case { int seg, ld, mode; } cases[] = {{64, 2048, 0}, {128, 2048, 0}, {256, 2048, 0}, {1024, 1024, 0}, {64, 256, 0}, {64, 512, 0},
                                                   {128, 256, 0}, {64, 2048, 1}, {64, 256, 1}, {64, 256, 2}, {128, 2048, 1}};
    for (auto cs : cases) {
        const int seg = cs.seg, ld = cs.ld, mode = cs.mode;  // segment bytes, row pitch of the source matrix
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(dma_bw_kernel, dim3(256), dim3(512), 131072, 0, d, region, 100, seg, ld, mode);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(dma_bw_kernel, dim3(256), dim3(512), 131072, 0, d, region, iters, seg, ld, mode);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double bytes = 256.0 * 8 * 4 * 1024.0 * iters;
        printf("LDS-DMA fill, %4d-B segments, pitch %4d, %s: %7.2f TB/s aggregate = %6.1f GB/s per CU (%.1f B/clk at 2.1 GHz)\n", seg, ld, mode == 0 ? "global" : mode == 1 ? "buffer" : "buffer 1/4 OOB", bytes / ms / 1e9,
               bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
    }
    hipFree(d);
}

template <int EPI, int ABL>
float time_k64r(LabParams p, int iters, int gm = 4, int grid = 256, int groups = 0, int group_delay = 0, int ng = 0) {
    p.groups = groups; p.group_delay = group_delay; p.ng = ng;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    p.gm = gm;
    const size_t lds = 5 * 256 * 128;
    auto kern = lab_k64r_kernel<EPI, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    if (getenv("LAB_DMA")) { dma_bench(); return 0; }
    const int M = getenv("LAB_M") ? atoi(getenv("LAB_M")) : 65536;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 3072, 1024}, {"fc2", 1024, 4096}, {"fc1", 4096, 1024}, {"out", 1024, 1024}};
    for (auto sh : shapes) {
        size_t na = (size_t)M * sh.K, nw = (size_t)sh.N * sh.K, no = (size_t)M * sh.N;
        std::vector<unsigned short> ha(na), hw(nw);
        srand(1);
        for (auto& v : ha) v = (unsigned short)((rand() & 0xFFFF) & 0xBFFF) | 0x3000;  // random finite bf16 ~[0.0, 2]
        for (auto& v : ha) if (rand() & 1) v |= 0x8000;
        for (auto& v : hw) { v = (unsigned short)(0x3800 | (rand() & 0x3FF)); if (rand() & 1) v |= 0x8000; }
        void *dA, *dW, *dO; float* dB;
        hipMalloc(&dA, na * 2); hipMalloc(&dW, nw * 2); hipMalloc(&dO, no * 4); hipMalloc(&dB, sh.N * 4);
        hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
        hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemset(dB, 0, sh.N * 4);
        LabParams p{};
        p.A = (const __bf16*)dA; p.W = (const __bf16*)dW; p.bias = dB; p.out = dO;
        p.M = M; p.N = sh.N; p.K = sh.K; p.lda = sh.K; p.ldw = sh.K; p.ldo = sh.N; p.tokens = 256;
        const double fl = 2.0 * M * sh.N * sh.K;
        auto rep = [&](const char* what, float ms) { printf("%s %-28s %8.3f ms %8.0f TF\n", sh.name, what, ms, fl / ms / 1e9); };
        constexpr int E = BSI_EPI_BIAS_BF16;
        if (getenv("LAB_CLK")) {
            // shader clock under load for two operand distributions: the narrow-exponent lab data above, and N(0,1)
            // activations x U(-1/sqrt(K), 1/sqrt(K)) weights (what the model feeds the kernel)
            unsigned long long* dC; hipMalloc(&dC, 64);
            LabParams q = p; q.out2 = dC;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    auto f2b = [](float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
                    for (auto& v : ha) {
                        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
                        v = f2b(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
                    }
                    for (auto& v : hw) v = f2b((2.f * rand() / (float)RAND_MAX - 1.f) / sqrtf((float)sh.K));
                    hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
                    hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
                }
                for (int rep_i = 0; rep_i < 2; ++rep_i) {
                    float ms = time_pring<E, 64>(q, 50);
                    unsigned long long hc[2];
                    hipMemcpy(hc, dC, 16, hipMemcpyDeviceToHost);
                    printf("%s %-14s %8.3f ms %8.0f TF   shader clock %.0f MHz\n", sh.name, pass ? "randn data" : "lab data", ms,
                           fl / ms / 1e9, 100.0 * hc[0] / hc[1]);
                }
            }
            continue;
        }
        if (getenv("LAB_EARLYBAR")) {
            // round 3: the barrier that ends an MFMA phase taken 8 / 16 MFMAs early
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            for (int r = 0; r < 3; ++r) {
                rep("production", time_k64r<E, 0>(p, 20));
                rep("barrier 8 MFMAs early", time_k64r<E, 524288>(p, 20));
                rep("barrier 16 MFMAs early", time_k64r<E, 1048576>(p, 20));
                rep("priority to the load phase", time_k64r<E, 2097152>(p, 20));
                if (sh.N == 4096) {
                    rep("gelu production", time_k64r<G, 0>(p, 20));
                    rep("gelu barrier 8 MFMAs early", time_k64r<G, 524288>(p, 20));
                }
            }
            continue;
        }
        if (getenv("LAB_SEG")) {
            // round 3: where a wave's time goes inside its phases (s_memtime stamps, one wave of each group per workgroup)
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            unsigned long long* dC; hipMalloc(&dC, 18 * 256 * 8);
            LabParams q = p; q.out2 = dC;
            auto show = [&](const char* what, float ms_plain, float ms) {
                unsigned long long hc[18 * 256];
                hipMemcpy(hc, dC, sizeof hc, hipMemcpyDeviceToHost);
                const char* names[9] = {"DMA issue + bookkeeping", "vmcnt wait", "lgkmcnt wait", "barrier before MFMAs", "32 MFMAs", "(epilogue,) barrier after", "fragment-read issue",
                                        "epilogue issue (per tile)", "epilogue end -> next tile's accumulators written (per tile)"};
                const double tiles = ((M + 255) / 256) * ((sh.N + 255) / 256) / 256.0;
                const double phases = tiles * (sh.K / 64) * 2;
                for (int grp = 0; grp < 2; ++grp) {
                    double sum[9] = {0};
                    for (int b = 0; b < 256; ++b) for (int i = 0; i < 9; ++i) sum[i] += (double)hc[18 * b + 9 * grp + i];
                    printf("%s %s (%.3f ms stamped, %.3f plain) group %c, cycles per phase pair:", sh.name, what, ms, ms_plain, grp ? 'B' : 'A');
                    double tot = 0;
                    for (int i : {0, 6, 1, 2, 3, 4, 5}) { printf("  %s %.0f", names[i], sum[i] / 256.0 / phases); tot += sum[i] / 256.0 / phases; }
                    printf("  | total %.0f | %s %.0f, %s %.0f\n", tot, names[7], sum[7] / 256.0 / tiles, names[8], sum[8] / 256.0 / (tiles - 1));
                }
            };
            for (int r = 0; r < 2; ++r) {
                const float a = time_k64r<E, 4194304>(q, 20), b = time_k64r<E, 4194304 | 16777216>(q, 20);
                show("plain epilogue", a, b);
                if (sh.N == 4096) {
                    const float c = time_k64r<G, 4194304>(q, 20), d = time_k64r<G, 4194304 | 16777216>(q, 20);
                    show("gelu epilogue", c, d);
                }
            }
            hipFree(dC);
            continue;
        }
        if (getenv("LAB_SNAKE")) {
            // round 3: boustrophedon MFMA order (tools/experiments/mfma_power.hip: +2 % on a register-only stream)
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            for (int r = 0; r < 4; ++r) {
                rep("production", time_k64r<E, 0>(p, 20));
                rep("boustrophedon MFMA order", time_k64r<E, 4194304>(p, 20));
                rep("boustrophedon + reads before DMA issue", time_k64r<E, 4194304 | 8388608>(p, 20));
                if (sh.N == 4096) {
                    rep("gelu production", time_k64r<G, 0>(p, 20));
                    rep("gelu boustrophedon MFMA order", time_k64r<G, 4194304>(p, 20));
                }
            }
            continue;
        }
        if (getenv("LAB_DMAONLY")) {
            // round 3: how long does the operand DMA stream take by itself (no MFMAs, fragments read once, no epilogue), on 256 and on 64 CUs?
            for (int ncu : {256, 64}) {
                LabParams q = p;
                q.M = (M / 256) * ncu;
                for (int r = 0; r < 2; ++r) {
                    const float full = time_k64r<E, 0>(q, 20, 4, ncu), dma = time_k64r<E, 2 | 4 | 8>(q, 20, 4, ncu), none = time_k64r<E, 1 | 2 | 4 | 8>(q, 20, 4, ncu),
                                mf = time_k64r<E, 1 | 4>(q, 20, 4, ncu);
                    const double bytes = (double)((q.M + 255) / 256) * ((sh.N + 255) / 256) * (512.0 * sh.K * 2.0);
                    printf("%s %3d CUs: full %.1f us | DMA stream alone %.1f us = %.1f GB/s per CU | barriers alone %.1f us | MFMA + reads, no DMA, no epilogue %.1f us\n", sh.name, ncu,
                           1e3 * full, 1e3 * dma, bytes / (dma * 1e-3) / 1e9 / ncu, 1e3 * none, 1e3 * mf);
                }
            }
            continue;
        }
        if (getenv("LAB_WDUP")) {
            // round 4: what NOT sharing the W half-stage between the two wave groups costs the steady state (ABL 33554432: every W
            // instruction issued twice, bit-identical results) -- the price of offsetting the groups by an epilogue
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            for (int r = 0; r < 3; ++r) {
                rep("production issue", time_k64r<E, 0>(p, 20));
                rep("W half-stages issued twice", time_k64r<E, 33554432>(p, 20));
                rep("no epilogue (upper bound of hiding it)", time_k64r<E, 4>(p, 20));
                if (sh.N == 4096) {
                    rep("gelu production issue", time_k64r<G, 0>(p, 20));
                    rep("gelu W half-stages issued twice", time_k64r<G, 33554432>(p, 20));
                    rep("gelu no epilogue", time_k64r<G, 4>(p, 20));
                }
            }
            continue;
        }
        if (getenv("LAB_SPLIT")) {
            // round 3: A half-stages' DMA issue split between the load phase and the MFMA phase (ABL 262144)
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            for (int r = 0; r < 3; ++r) {
                rep("production issue", time_k64r<E, 0>(p, 20));
                rep("split A half-stages", time_k64r<E, 262144>(p, 20));
                if (sh.N == 4096) {
                    rep("gelu production issue", time_k64r<G, 0>(p, 20));
                    rep("gelu split A half-stages", time_k64r<G, 262144>(p, 20));
                }
            }
            continue;
        }
        if (getenv("LAB_NS")) {
            // round 3: n-stationary tile order (an XCD keeps ng weight panels across rounds) x band height x skewed phase groups
            // that share those panels, static priority, and the shader clock of the production schedule on 256 / 64 CUs
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            const int tiles_per_wg = ((M + 255) / 256) * ((sh.N + 255) / 256) / 256;
            unsigned long long* dC; hipMalloc(&dC, 2 * 256 * 8);
            for (int r = 0; r < 2; ++r) {
                const float t1 = time_k64r<E, 0>(p, 20);
                rep("band order gm 4 (production)", t1);
                const double period64 = (double)t1 * 1e-3 / tiles_per_wg * 1.7e9 / 64.0;  // tile period, units of 64 clocks at ~1.7 GHz
                rep("static priority for waves 4-7", time_k64r<E, 131072>(p, 20));
                rep("ns ng 4 gm 8", time_k64r<E, 0>(p, 20, 8, 256, 0, 0, 4));
                rep("ns ng 4 gm 8, wb stores", time_k64r<E, 32>(p, 20, 8, 256, 0, 0, 4));
                rep("ns ng 4 gm 8, A nt", time_k64r<E, 2048>(p, 20, 8, 256, 0, 0, 4));
                rep("ns ng 2 gm 16", time_k64r<E, 0>(p, 20, 16, 256, 0, 0, 2));
                rep("ns ng 8 gm 4", time_k64r<E, 0>(p, 20, 4, 256, 0, 0, 8));
                for (int ngr : {2, 4})
                    for (double frac : {0.0, 1.0}) {
                        const int d = (int)(frac * period64 / ngr);
                        char nm[96];
                        snprintf(nm, sizeof nm, "ns ng 4 gm %d, %d groups, delay %.1f", 8 / ngr, ngr, frac);
                        rep(nm, time_k64r<E, 0>(p, 20, 8 / ngr, 256, ngr, d, 4));
                    }
                if (sh.N == 4096) {
                    rep("gelu band order (production)", time_k64r<G, 0>(p, 20));
                    rep("gelu static priority", time_k64r<G, 131072>(p, 20));
                    rep("gelu ns ng 4 gm 8", time_k64r<G, 0>(p, 20, 8, 256, 0, 0, 4));
                    rep("gelu ns ng 4 gm 2, 4 groups, delay 1.0", time_k64r<G, 0>(p, 20, 2, 256, 4, (int)(period64 / 4), 4));
                    rep("gelu ns ng 4 gm 4, 2 groups, delay 1.0", time_k64r<G, 0>(p, 20, 4, 256, 2, (int)(period64 / 2), 4));
                }
            }
            for (int ncu : {256, 64}) {  // shader clock of the production schedule (stamps in a laboratory instance only)
                LabParams q = p;
                q.M = 256 * ncu; q.out2 = dC;
                for (int abl : {0, 1}) {
                    const float ms = abl ? time_k64r<E, 64 | 1 | 4>(q, 50, 4, ncu) : time_k64r<E, 64>(q, 50, 4, ncu);
                    unsigned long long hc[512];
                    hipMemcpy(hc, dC, 2 * ncu * 8, hipMemcpyDeviceToHost);
                    double mn = 1e9, mx = 0, sum = 0;
                    for (int i = 0; i < ncu; ++i) { const double c = 100.0 * hc[2 * i] / hc[2 * i + 1]; mn = c < mn ? c : mn; mx = c > mx ? c : mx; sum += c; }
                    printf("%s %3d CUs %-22s %8.3f ms %8.0f TF-equivalent of 256 CUs, shader clock %.0f MHz (min %.0f, max %.0f)\n", sh.name, ncu,
                           abl ? "MFMA + fragment reads" : "full kernel", ms, 2.0 * q.M * sh.N * sh.K / ms / 1e9 * 256 / ncu, sum / ncu, mn, mx);
                }
            }
            hipFree(dC);
            continue;
        }
        if (getenv("LAB_GROUPS")) {
            // phase groups per XCD (k64r): 1 / 2 / 4 groups, start offset = frac x (tile period / groups); tile period from the plain run
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            const float t1 = time_k64r<E, 0>(p, 20);
            rep("k64r plain", t1);
            const int tiles_per_wg = ((M + 255) / 256) * ((sh.N + 255) / 256) / 256;
            const double period64 = (double)t1 * 1e-3 / tiles_per_wg * 1.7e9 / 64.0;  // tile period in 64-clock units at ~1.7 GHz
            for (int ng : {2, 4})
                for (double frac : {0.0, 0.5, 1.0, 1.5}) {
                    const int d = (int)(frac * period64 / ng);
                    char nm[96];
                    snprintf(nm, sizeof nm, "k64r %d groups, delay %.1f (%d x 64 clk)", ng, frac, d);
                    rep(nm, time_k64r<E, 0>(p, 20, 4, 256, ng, d));
                }
            rep("k64r plain (again)", time_k64r<E, 0>(p, 20));
            if (sh.N == 4096) {
                rep("k64r gelu plain", time_k64r<G, 0>(p, 20));
                rep("k64r gelu 2 groups, delay 1.0", time_k64r<G, 0>(p, 20, 4, 256, 2, (int)(period64 / 2)));
                rep("k64r gelu 4 groups, delay 1.0", time_k64r<G, 0>(p, 20, 4, 256, 4, (int)(period64 / 4)));
            }
            continue;
        }
        if (getenv("LAB_REGULAR")) {
            // never-ending issue stream (ABL 65536) against the production stream that stops after the last tile
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            for (int r = 0; r < 3; ++r) {
                rep("k64r stream ends", time_k64r<E, 0>(p, 20));
                rep("k64r never-ending stream", time_k64r<E, 65536>(p, 20));
                if (sh.N == 4096) {
                    rep("k64r gelu stream ends", time_k64r<G, 0>(p, 20));
                    rep("k64r gelu never-ending stream", time_k64r<G, 65536>(p, 20));
                }
            }
            continue;
        }
        if (getenv("LAB_STORES")) {
            // what the epilogue's stores cost with and without the operand stream beside them (256 CUs), each row twice
            for (int r = 0; r < 2; ++r) {
                rep("k64r full", time_k64r<E, 0>(p, 20));
                rep("k64r without global stores", time_k64r<E, 128>(p, 20));
                rep("k64r no operand DMA, with stores", time_k64r<E, 1>(p, 20));
                rep("k64r no operand DMA, without stores", time_k64r<E, 1 | 128>(p, 20));
                rep("k64r no operand DMA, no epilogue", time_k64r<E, 1 | 4>(p, 20));
            }
            continue;
        }
        if (getenv("LAB_FEWCU")) {
            // Is the epilogue's store tail a per-CU limit or the chip-wide write bandwidth of 256 synchronised epilogues?  The same
            // number of tiles per workgroup (M = 256 rows per workgroup) on 256 / 64 / 16 / 8 CUs: time per tile with the stores,
            // without them, without the epilogue.
            for (int ncu : {256, 64, 16, 8}) {
                LabParams q = p;
                q.M = 256 * ncu;
                const int tiles_per_wg = (sh.N + 255) / 256;
                const float f = time_k64r<E, 0>(q, 20, 4, ncu), ns = time_k64r<E, 128>(q, 20, 4, ncu), ne = time_k64r<E, 4>(q, 20, 4, ncu);
                printf("%s %3d CUs, %d tiles each: per tile %.2f us full, %.2f us without the global stores, %.2f us without epilogue\n", sh.name, ncu,
                       tiles_per_wg, 1e3 * f / tiles_per_wg, 1e3 * ns / tiles_per_wg, 1e3 * ne / tiles_per_wg);
            }
            continue;
        }
        if (getenv("LAB_GM")) {  // band height x cache policy of the operand streams (variant 12)
            for (int gm : {1, 2, 4, 8, 16, 32}) {
                char nm[64];
                snprintf(nm, sizeof nm, "v12 gm %2d default", gm); rep(nm, time_k64r<E, 0>(p, 20, gm));
                snprintf(nm, sizeof nm, "v12 gm %2d A nt", gm); rep(nm, time_k64r<E, 2048>(p, 20, gm));
                snprintf(nm, sizeof nm, "v12 gm %2d W nt", gm); rep(nm, time_k64r<E, 4096>(p, 20, gm));
            }
            continue;
        }
        if (getenv("LAB_V12")) {
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            rep("v6 full (wb stores)", time_pring<E, 32>(p, 20));
            rep("v6 no epilogue", time_pring<E, 4>(p, 20));
            rep("v6 no epi, no glds", time_pring<E, 5>(p, 20));
            rep("v12 full", time_k64r<E, 0>(p, 20));
            rep("v12 gelu full", time_k64r<G, 0>(p, 20));
            rep("v12 A loads nt", time_k64r<E, 2048>(p, 20));
            rep("v12 W loads nt", time_k64r<E, 4096>(p, 20));
            rep("v12 A and W loads nt", time_k64r<E, 2048 | 4096>(p, 20));
            rep("v12 A loads sc1", time_k64r<E, 8192>(p, 20));
            rep("v12 W loads sc1", time_k64r<E, 16384>(p, 20));
            rep("v12 full (again)", time_k64r<E, 0>(p, 20));
            rep("v12 epilogue w/o global stores", time_k64r<E, 128>(p, 20));
            rep("v12 gelu w/o global stores", time_k64r<G, 128>(p, 20));
            rep("v12 ordinary (wb) stores", time_k64r<E, 32>(p, 20));
            rep("v12 no epilogue", time_k64r<E, 4>(p, 20));
            rep("v12 no epi, no glds", time_k64r<E, 5>(p, 20));
            continue;
        }
        if (getenv("LAB_W1")) {
            rep("v6 full", time_pring<E, 0>(p, 20));
            rep("v6 epilogue w/o global stores", time_pring<E, 128>(p, 20));
            rep("v6 stores into 256 rows (nt)", time_pring<E, 256>(p, 20));
            rep("v6 stores into 256 rows (wb)", time_pring<E, 256 | 32>(p, 20));
            rep("v6 quarter of the stores (nt)", time_pring<E, 512>(p, 20));
            rep("v6 normal stores", time_pring<E, 32>(p, 20));
            rep("v6 no epilogue", time_pring<E, 4>(p, 20));
            rep("v6 no epi, ring 5", time_pring<E, 4, 5>(p, 20));
            rep("v6 no epi, ring 3", time_pring<E, 4, 3>(p, 20));
            rep("v6 no epi, ring 2", time_pring<E, 4, 2>(p, 20));
            rep("v6 no epi, no glds", time_pring<E, 5>(p, 20));
            rep("v6 no epi, no frag reads", time_pring<E, 6>(p, 20));
            rep("v6 no epi, no glds, no frag reads", time_pring<E, 7>(p, 20));
            rep("w1 full", time_w1<E, 0>(p, 20));
            rep("w1 no epilogue", time_w1<E, 4>(p, 20));
            rep("w1 no gload/lstore, no epi", time_w1<E, 5>(p, 20));
            rep("w1 no frag reads, no epi", time_w1<E, 6>(p, 20));
            rep("w1 mfma + barrier only", time_w1<E, 7>(p, 20));
            rep("w1 mfma only", time_w1<E, 15>(p, 20));
            continue;
        }
        if (getenv("LAB_V6")) {
            constexpr int G = BSI_EPI_BIAS_GELU_BF16;
            rep("v6 full", time_pring<E, 0>(p, 20));
            rep("v6 no epilogue", time_pring<E, 4>(p, 20));
            rep("v6 no glds", time_pring<E, 1>(p, 20));
            rep("v6 no glds, no epi", time_pring<E, 5>(p, 20));
            rep("v6 gelu full", time_pring<G, 0>(p, 20));
            rep("v6 full nt stores", time_pring<E, 32>(p, 20));
            rep("v6 gelu nt stores", time_pring<G, 32>(p, 20));
            continue;
        }
        if (getenv("LAB_PMC")) {
            rep("full", time_pp<E, 0>(p, 3));
            rep("no epilogue", time_pp<E, 4>(p, 3));
            rep("no glds, no epi", time_pp<E, 5>(p, 3));
            continue;
        }
        rep("full", time_pp<E, 0>(p, 20));
        rep("full + L2 prefetch", time_pp<E, 16>(p, 20));
        rep("no epi + L2 prefetch", time_pp<E, 20>(p, 20));
        for (int st : {8, 16, 32, 64, 128}) {
            LabParams q = p; q.stagger = st;
            char nm[64]; snprintf(nm, sizeof nm, "full stagger %d", st);
            rep(nm, time_pp<E, 0>(q, 20));
        }
        rep("no epilogue", time_pp<E, 4>(p, 20));
        rep("no glds", time_pp<E, 1>(p, 20));
        rep("no glds, no epi", time_pp<E, 5>(p, 20));
        rep("no ds_read, no epi", time_pp<E, 6>(p, 20));
        rep("no glds/ds_read/epi", time_pp<E, 7>(p, 20));
        rep("mfma only (no barriers)", time_pp<E, 15>(p, 20));
        rep("no barriers, no epi", time_pp<E, 12>(p, 20));
        hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dB);
    }
    return 0;
}
