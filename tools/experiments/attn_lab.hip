// Attention-forward laboratory (not part of the product): times ablated builds of the production kernel at the DiT shape
// (B x 16 heads x 256 tokens x 64) to see which resource bounds it.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I bsi_amd/csrc tools/experiments/attn_lab.hip -o build/attn_lab && build/attn_lab
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../bsi_amd/csrc/attention.hip"
#include "../../bsi_amd/csrc/attention_persist.hip"
#include "../../bsi_amd/csrc/bsi_ops.hip"  // bsi_set_error

template <int ABL>
float time_attn(const __bf16* qkv, __bf16* out, int B, int iters) {
    constexpr int T = 256, H = 16, DH = 64;
    const size_t lds = 2 * (size_t)T * DH * 2;
    auto kern = attention_fwd_kernel<DH, 64, false, true, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float sl = 1.4426950408889634f / 8.0f;
    dim3 grid(1, B * H);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, qkv, 3 * H * DH, T, H, out, H * DH, sl, nullptr, DropCfg{});
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, qkv, 3 * H * DH, T, H, out, H * DH, sl, nullptr, DropCfg{});
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const int B = getenv("LAB_B") ? atoi(getenv("LAB_B")) : 256;
    const size_t n = (size_t)B * 256 * 3072;
    std::vector<unsigned short> h(n);
    srand(1);
    for (auto& v : h) { v = (unsigned short)(0x3f00 | (rand() & 0xff)); if (rand() & 1) v |= 0x8000; }  // ~ +-[0.5, 1)
    void *dq, *dout;
    (void)hipMalloc(&dq, n * 2); (void)hipMalloc(&dout, (size_t)B * 256 * 1024 * 2);
    (void)hipMemcpy(dq, h.data(), n * 2, hipMemcpyHostToDevice);
    const __bf16* q = (const __bf16*)dq;
    __bf16* o = (__bf16*)dout;
    const double fl = 4.0 * B * 16 * 256.0 * 256 * 64;
    auto rep = [&](const char* what, float ms) { printf("%-44s %8.1f us %7.0f TF\n", what, ms * 1e3, fl / ms / 1e9); };
    rep("full", time_attn<0>(q, o, B, 30));
    rep("no exp2", time_attn<1>(q, o, B, 30));
    rep("no global stores", time_attn<2>(q, o, B, 30));
    rep("no P.V", time_attn<4>(q, o, B, 30));
    rep("no K.Q^T", time_attn<8>(q, o, B, 30));
    rep("no K.Q^T, no P.V", time_attn<12>(q, o, B, 30));
    rep("no K/V loads", time_attn<16>(q, o, B, 30));
    rep("no K/V, no Q loads", time_attn<48>(q, o, B, 30));
    rep("no loads, no stores", time_attn<50>(q, o, B, 30));
    rep("no loads, no stores, no exp2", time_attn<51>(q, o, B, 30));
    rep("no loads/stores/MFMA (softmax only)", time_attn<62>(q, o, B, 30));
    rep("loads + stores only (no MFMA, no exp2)", time_attn<13>(q, o, B, 30));
    rep("full (again)", time_attn<0>(q, o, B, 30));
    return 0;
}
