// Sustained MFMA rate and shader clock of the whole chip under the two bf16 MFMA shapes, register operands only (no LDS, no
// memory): is the forward GEMM's clock ceiling a property of the 16x16x32 instruction stream?
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_power.hip -o build/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>  // 0: 16x16x32, wave tile 128x64 (8 A x 4 B fragments, 32 accumulators of 4 registers)
                      // 1: 32x32x16, wave tile 128x64 (4 A x 2 B fragments, 8 accumulators of 16 registers), two k16 steps
__global__ __launch_bounds__(512, 1) void mfma_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                      long long* __restrict__ clk) {
    const int t = threadIdx.x;
    bf16x8 a[2][8], b[2][4];
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 8; ++i) a[s][i] = src[(s * 12 + i) * 512 + t];
        for (int i = 0; i < 4; ++i) b[s][i] = src[(s * 12 + 8 + i) * 512 + t];
    }
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    if constexpr (SHAPE == 0) {
        f32x4 acc[8][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
        float r = 0.f;
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        out[blockIdx.x * 512 + t] = r;
    } else {
        f32x16 acc[4][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h)  // two k16 halves of the k32 step: the same flops per iteration as SHAPE 0
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i * 2 + h], b[s][j * 2 + h], acc[i][j], 0, 0, 0);
        }
        float r = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 16; ++k) r += acc[i][j][k];
        out[blockIdx.x * 512 + t] = r;
    }
    if (t == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

// 16x16x32 only.  ORDER 0: row-major over the 8 x 4 fragment grid (A changes every fourth instruction, B every instruction);
// 1: boustrophedon (one operand changes per instruction); 2: diagonal (both operands change every instruction).
// NREAD: ds_read_b128 fragment reads per k32 step (32 MFMAs), refilling the operand registers from an LDS image of random data:
// 12 = the k64r kernel's 128 x 64 wave tile, 8 = what a 128 x 128 wave tile would need per 32 MFMAs, 0 = none.
// NDMA: global_load_lds_dwordx4 per wave and k32 step into a separate 32 KB of LDS (4 = the k64r kernel's stream: 32 KB per
// workgroup and k32 step), walking `region` bytes of `stream` per workgroup (small = L2 hits, large = HBM).
// NGL (round 4): the 4 W fragments of a k32 step come STRAIGHT FROM MEMORY into registers (global_load_dwordx4, a step ahead, rows of a
// [256][1024] bf16 weight tile shared by 16 workgroups: L2 hits), not through LDS -- with NREAD = 8 and NDMA = 2 that is the k64r
// kernel's mix with the W panel taken off the LDS path (fragment reads 12 -> 8, LDS-DMA 32 -> 16 KB per k32 step).
template <int ORDER, int NREAD, int NDMA, int NGL = 0>
__global__ __launch_bounds__(512, 1) void mfma_lds_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                          long long* __restrict__ clk, const char* __restrict__ stream = nullptr,
                                                          unsigned region = 0) {
    __shared__ bf16x8 lds[24 * 512 / 2];  // 96 KB
    __shared__ char landing[NDMA ? 32768 : 16];
    const char* my = stream + (size_t)blockIdx.x * region + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    unsigned walk = 0;
    const int t = threadIdx.x;
    for (int i = t; i < 24 * 512 / 2; i += 512) lds[i] = src[i];
    __syncthreads();
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[i * 512 + t];
    for (int i = 0; i < 4; ++i) b[i] = src[(8 + i) * 512 + t];
    f32x4 acc[8][4] = {};
    // W tile of this workgroup: 16 distinct tiles of 512 KB behind `stream` + 64 MB, rows of 2 KB; wave (wm, wn): columns 64 wn ..; a
    // fragment = 16 rows x 64 B (lane: row lane & 15, 16-B chunk lane >> 4)
    const char* wt = stream + (64u << 20) + (size_t)(blockIdx.x & 15) * (512u << 10) + (size_t)(((t >> 6) & 3) * 64 + (t & 15)) * 2048 + ((t >> 4) & 3) * 16;
    bf16x8 bn[4];
    unsigned kk = 0;
    if (NGL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(wt + j * 16 * 2048);
    }
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters * 2; ++it) {
        if (NGL) {  // this step's fragments arrived during the last one; the next step's go out now
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = bn[j];
            kk = (kk + 64) & 2047;
#pragma unroll
            for (int j = 0; j < NGL; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(wt + j * 16 * 2048 + kk);
        }
        const int base = ((it * 12) & 127) * 32 + (t & 63);  // walks the image; lane-linear 16-byte reads (conflict free)
        if (NDMA) {
#pragma unroll
            for (int q = 0; q < NDMA; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(my + walk + q * 8192),
                                                 (__attribute__((address_space(3))) void*)(landing + (t >> 6) * 1024 + (q & 3) * 8192), 16, 0, 0);
            walk += NDMA * 8192;
            if (walk >= region) walk = 0;
        }
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            int i, j;
            if (ORDER == 0) { i = n >> 2; j = n & 3; }
            else if (ORDER == 1) { i = n >> 2; j = (i & 1) ? 3 - (n & 3) : (n & 3); }
            else { i = n & 7; j = (n + (n >> 3)) & 3; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (NREAD && n >= 32 - NREAD) {  // the reads sit behind the last use of the register they refill
                const int f = n - (32 - NREAD);
                if (f < 8) a[(ORDER == 2) ? f : f] = lds[(base + f * 64) % (24 * 512 / 2)];
                else if (!NGL) b[f - 8] = lds[(base + f * 64) % (24 * 512 / 2)];
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + t] = r;
    if (t == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

// One wave per SIMD (256-thread workgroup, up to 512 registers per wave): a 128 x 128 wave tile = 8 x 8 fragment grid, 64 accumulators
// of 4 registers, 16 fragment reads and 2 * NDMA4 LDS-DMA instructions per wave per k32 step (64 MFMAs) -- the instruction mix of a
// 256 x 256 tile on four waves (review item 1c), free running.
template <int NREAD, int NDMA>
__global__ __launch_bounds__(256, 1) void mfma_w4_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters,
                                                         long long* __restrict__ clk, const char* __restrict__ stream, unsigned region) {
    __shared__ bf16x8 lds[24 * 512 / 2];  // 96 KB
    __shared__ char landing[NDMA ? 32768 : 16];
    const int t = threadIdx.x;
    for (int i = t; i < 24 * 512 / 2; i += 256) lds[i] = src[i];
    __syncthreads();
    const char* my = stream + (size_t)blockIdx.x * region + (t & 63) * 16 + (t >> 6) * 1024;
    unsigned walk = 0;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) a[i] = src[i * 512 + t];
    for (int i = 0; i < 8; ++i) b[i] = src[(8 + i) * 512 + t];
    f32x4 acc[8][8] = {};
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {  // 64 MFMAs per iteration, as one iteration of mfma_kernel<0>
        const int base = ((it * 16) & 127) * 32 + (t & 63);
        if (NDMA) {
#pragma unroll
            for (int q = 0; q < NDMA; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(my + walk + q * 4096),
                                                 (__attribute__((address_space(3))) void*)(landing + (t >> 6) * 1024 + (q & 7) * 4096), 16, 0, 0);
            walk += NDMA * 4096;
            if (walk >= region) walk = 0;
        }
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            const int i = n >> 3, j = (i & 1) ? 7 - (n & 7) : (n & 7);
            // accumulators pinned to AGPRs: with the builtin hipcc moves them between the two halves of the register file inside the
            // loop (6 v_accvgpr moves per MFMA at this size)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
            if (NREAD && n >= 64 - NREAD) {
                const int f = n - (64 - NREAD);
                if (f < 8) a[f] = lds[(base + f * 64) % (24 * 512 / 2)];
                else b[f - 8] = lds[(base + f * 64) % (24 * 512 / 2)];
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + t] = r;
    if (t == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<unsigned short> h(24 * 512 * 8);
    unsigned x = 12345u;
    for (auto& v : h) {  // bf16 values of magnitude ~1 with random mantissas and signs (what activations / weights toggle)
        x = x * 1664525u + 1013904223u;
        v = zero ? 0 : (unsigned short)(((x >> 16) & 0x80ffu) | (((x >> 9) & 3u) + 0x7eu) << 7);
    }
    bf16x8* src;
    float* out;
    long long* clk;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, grid * 512 * 4);
    hipMalloc(&clk, grid * 16);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"16x16x32", "32x32x16", "16x16x32 boustrophedon", "16x16x32 diagonal", "16x16x32 + 12 LDS reads / 32",
                           "16x16x32 + 8 LDS reads / 32", "16x16x32 + 4 LDS reads / 32", "+ 12 LDS reads + 4 DMA / 32 (L2)",
                           "+ 12 LDS reads + 4 DMA / 32 (HBM)", "+ 8 LDS reads + 4 DMA / 32 (L2)", "+ 0 LDS reads + 4 DMA / 32 (L2)",
                           "1 wave/SIMD 128x128: MFMAs only", "1 wave/SIMD: + 16 reads / 64", "1 wave/SIMD: + 16 reads + 8 DMA / 64",
                           "W from memory: 8 LDS reads + 2 DMA + 4 global loads / 32", "W from memory, boustrophedon", "12 LDS reads + 4 DMA, boustrophedon"};
    char* stream;
    hipMalloc(&stream, (size_t)grid * (4u << 20) + (80u << 20));
    hipMemset(stream, 0x3f, (size_t)grid * (4u << 20) + (80u << 20));
    const int nshape = argc > 4 ? atoi(argv[4]) : 2;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < nshape; ++shape) {
            hipEventRecord(e0);
            const dim3 g(grid), b(512);
            if (shape == 0) hipLaunchKernelGGL(mfma_kernel<0>, g, b, 0, 0, src, out, iters, clk);
            else if (shape == 1) hipLaunchKernelGGL(mfma_kernel<1>, g, b, 0, 0, src, out, iters, clk);
            else if (shape == 2) hipLaunchKernelGGL((mfma_lds_kernel<1, 0, 0>), g, b, 0, 0, src, out, iters, clk, nullptr, 0u);
            else if (shape == 3) hipLaunchKernelGGL((mfma_lds_kernel<2, 0, 0>), g, b, 0, 0, src, out, iters, clk, nullptr, 0u);
            else if (shape == 4) hipLaunchKernelGGL((mfma_lds_kernel<0, 12, 0>), g, b, 0, 0, src, out, iters, clk, nullptr, 0u);
            else if (shape == 5) hipLaunchKernelGGL((mfma_lds_kernel<0, 8, 0>), g, b, 0, 0, src, out, iters, clk, nullptr, 0u);
            else if (shape == 6) hipLaunchKernelGGL((mfma_lds_kernel<0, 4, 0>), g, b, 0, 0, src, out, iters, clk, nullptr, 0u);
            else if (shape == 7) hipLaunchKernelGGL((mfma_lds_kernel<0, 12, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 8) hipLaunchKernelGGL((mfma_lds_kernel<0, 12, 4>), g, b, 0, 0, src, out, iters, clk, stream, 4u << 20);
            else if (shape == 9) hipLaunchKernelGGL((mfma_lds_kernel<0, 8, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 10) hipLaunchKernelGGL((mfma_lds_kernel<0, 0, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 11) hipLaunchKernelGGL((mfma_w4_kernel<0, 0>), g, dim3(256), 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 12) hipLaunchKernelGGL((mfma_w4_kernel<16, 0>), g, dim3(256), 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 13) hipLaunchKernelGGL((mfma_w4_kernel<16, 8>), g, dim3(256), 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 14) hipLaunchKernelGGL((mfma_lds_kernel<0, 8, 2, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            else if (shape == 15) hipLaunchKernelGGL((mfma_lds_kernel<1, 8, 2, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            else hipLaunchKernelGGL((mfma_lds_kernel<1, 12, 4>), g, b, 0, 0, src, out, iters, clk, stream, 65536u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(grid * 2);
            hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            for (int i = 0; i < grid; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
            const double flops = (double)grid * ((shape >= 11 && shape <= 13) ? 4 : 8) * iters * 64.0 * 16384.0;  // 64 MFMA-equivalents of 16x16x32 per wave and iteration
            printf("%-32s grid %d data %s: %.2f ms  %.0f TFLOP/s  shader clock %.0f MHz  (%.3f TFLOP/s per MHz)\n",
                   names[shape], grid, zero ? "zeros" : "random", ms, flops / ms * 1e-9,
                   cyc / wall * 100.0, flops / ms * 1e-9 / (cyc / wall * 100.0));
        }
    return 0;
}
