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
    int key;
    if (mode == 0) key = (-(row >> 2)) & 3;              // the kernels' swizzle: chunk ^= (-(row >> 2)) & 3
    else if (mode == 1) key = (row >> 2) & 3;            // plain (row >> 2) & 3
    else key = ((row >> 2) ^ (row >> 4)) & 3;            // folded
    const int addr = row * 64 + ((qd ^ key) << 4);
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(lds + addr + j * 1024);  // 8 reads in flight
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
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int shift : {0, 1, 2, 3, 4, 5, 8, 15, 16, 31, 32, 33, -1, -33}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, shift, mode, 10, d, s);
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, shift, mode, iters, d, s);
            unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
            printf("swizzle %d shift %4d: %6.1f cycles per ds_read_b128 (4 waves, one per SIMD; 8 = conflict free)\n", mode, shift, (double)c / (iters * 8 * 4));
        }
    return 0;
}
