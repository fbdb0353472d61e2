// Which physical compute units does bit i of a hipExtStreamCreateWithCUMask mask enable?  For a list of masks: launch a kernel of
// many small workgroups on a stream created with that mask, every workgroup records (XCC_ID, HW_ID), and the host prints the set of
// (xcc, se, sh, cu) the workgroups ran on.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay a little so that the workgroups spread over every CU the mask allows
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

static void run(const char* name, const std::vector<uint32_t>& mask, unsigned* dbuf, int nwg, bool verbose) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        printf("%s: stream creation failed\n", name);
        return;
    }
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, s, dbuf);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nwg);
    hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;  // xcc -> set of (se, sh, cu)
    std::map<unsigned, int> wg_per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        wg_per_xcc[xcc]++;
    }
    int total = 0;
    printf("%-28s:", name);
    for (auto& kv : per_xcc) {
        printf(" xcc%u:%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("  = %d CUs;  workgroups per xcc:", total);
    for (auto& kv : wg_per_xcc) printf(" %d", kv.second);
    printf("\n");
    if (verbose)
        for (auto& kv : per_xcc) {
            printf("    xcc%u:", kv.first);
            for (unsigned v : kv.second) printf(" se%u.sh%u.cu%u", v >> 8, (v >> 4) & 1, v & 0xf);
            printf("\n");
        }
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int all = prop.multiProcessorCount, words = (all + 31) / 32;
    printf("device: %s, %d CUs\n", prop.name, all);
    unsigned* dbuf;
    const int nwg = 8192;
    hipMalloc(&dbuf, 2 * nwg * 4);
    auto range = [&](int lo, int hi) {
        std::vector<uint32_t> m(words, 0u);
        for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32);
        return m;
    };
    char name[64];
    for (int b : {0, 1, 2, 7, 8, 9, 31, 32, 33, 64, 128, 255}) {
        snprintf(name, sizeof name, "bit %d", b);
        run(name, range(b, b + 1), dbuf, nwg, true);
    }
    for (int n : {8, 16, 32, 64, 224, 232, 240, 256}) {
        snprintf(name, sizeof name, "bits [0, %d)", n);
        run(name, range(0, n), dbuf, nwg, n <= 16);
    }
    for (int n : {16, 24, 32}) {
        snprintf(name, sizeof name, "bits [%d, 256)", all - n);
        run(name, range(all - n, all), dbuf, nwg, true);
    }
    {   // every 8th bit
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < all; i += 8) m[i / 32] |= 1u << (i % 32);
        run("bits 0, 8, 16, ...", m, dbuf, nwg, true);
    }
    {   // 4 bits of every 32
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < all; ++i)
            if (i % 32 >= 28) m[i / 32] |= 1u << (i % 32);
        run("bits 28..31 of every 32", m, dbuf, nwg, true);
    }
    return 0;
}
