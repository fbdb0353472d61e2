// Does buffer_load_dwordx4 ... lds write ZEROS to LDS for lanes whose offset is out of range?  (conv padding taps rely on it)
// build: hipcc -O3 --offload-arch=gfx950 buf_lds_oob.hip -o /tmp/buf_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* a, unsigned n, float* out, const unsigned* offs) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 256; i += 64) ((float*)lds)[i] = -7.0f;  // garbage that must be overwritten
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n, 0x00020000);
    unsigned vo = offs[threadIdx.x];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, vo, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((float*)lds)[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    std::vector<unsigned> offs(64);
    for (int l = 0; l < 64; ++l) offs[l] = (l % 3 == 1) ? 0xFFFFFF00u : (unsigned)(l * 64);  // every third lane out of range
    float *da, *dout; unsigned* doffs;
    hipMalloc(&da, 16384); hipMalloc(&dout, 1024); hipMalloc(&doffs, 256);
    hipMemcpy(da, h.data(), 16384, hipMemcpyHostToDevice);
    hipMemcpy(doffs, offs.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, (const char*)da, 0x80000000u, dout, doffs);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const float want = (l % 3 == 1) ? 0.0f : (float)(l * 16 + e);
            if (o[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, e, o[l * 4 + e], want); ++bad; }
        }
    printf(bad ? "FAIL %d\n" : "OK: out-of-range lanes wrote zeros to LDS\n", bad);
    return bad != 0;
}
