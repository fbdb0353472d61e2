// Do ds_read_b128 fragment reads of the 64-B-row swizzled image stay bank-conflict free when the 16 rows of a fragment start at an
// arbitrary row (the pixel-slab convolution reads rows base + rho + dy*W + dx)?  Times 8 reads per trip for several row shifts and
// two swizzles.   hipcc -O3 --offload-arch=gfx950 lds_shift_conflicts.hip -o /tmp/lds_shift && /tmp/lds_shift
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(int shift, int mode, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, rho = lane & 15, qd = lane >> 4;
    for (int i = threadIdx.x; i < 40960 / 4; i += 256) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const int row = 64 + rho + shift;
    // mode = a | b << 4 | neg << 8 | add << 9: key = +-((row >> a) (^ or +) (row >> b)) & 3   (b = 15: single term)
    const int a = mode & 15, b = (mode >> 4) & 15, neg = (mode >> 8) & 1, add = (mode >> 9) & 1;
    int key = row >> a;
    if (b != 15) key = add ? key + (row >> b) : key ^ (row >> b);
    key = (neg ? -key : key) & 3;
    const int addr = row * 64 + ((qd ^ key) << 4);
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
        // volatile asm: the addresses are loop invariant and the compiler would hoist plain loads out of the loop
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(lds + addr + j * 1024)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8); hipMalloc(&s, 1024);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 500;
    for (int neg = 0; neg < 2; ++neg)
        for (int add = 0; add < 2; ++add)
            for (int a = 0; a < 4; ++a)
                for (int b : {15, 1, 2, 3, 4}) {
                    if (b != 15 && b <= a) continue;
                    if (b == 15 && add) continue;
                    const int mode = a | b << 4 | neg << 8 | add << 9;
                    double worst = 0, best = 1e9;
                    for (int shift = 0; shift < 16; ++shift) {
                        hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, shift, mode, 10, d, s);
                        hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, shift, mode, iters, d, s);
                        unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
                        const double cyc = (double)c / (iters * 8 * 4);
                        worst = cyc > worst ? cyc : worst; best = cyc < best ? cyc : best;
                    }
                    printf("key = %s((row >> %d)%s) & 3: best %.1f worst %.1f cycles per read over shifts 0..15\n", neg ? "-" : "", a,
                           b == 15 ? "" : (add ? (b == 1 ? " + (row >> 1)" : b == 2 ? " + (row >> 2)" : b == 3 ? " + (row >> 3)" : " + (row >> 4)")
                                              : (b == 1 ? " ^ (row >> 1)" : b == 2 ? " ^ (row >> 2)" : b == 3 ? " ^ (row >> 3)" : " ^ (row >> 4)")), best, worst);
                }
    return 0;
}
