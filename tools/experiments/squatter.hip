// Laboratory code (not part of libbsi_hip.so): a kernel that HOLDS `ncu` compute units for `us` microseconds the way a
// communication kernel does while a bucket is in flight -- one workgroup per CU (it asks for all 160 KB of LDS, so nothing
// else fits beside it: the pessimistic case; an RCCL workgroup is lighter), spinning on the 100 MHz real-time counter.
// tools/experiments/squat_ab.py launches it on a high-priority side stream next to DPTrainer.train_step to show what a
// static one-workgroup-per-CU tile partition costs when some CUs are taken, and what bsi_set_cu_reserve buys back.
//   hipcc -O2 -shared -fPIC --offload-arch=gfx950 tools/experiments/squatter.hip -o tools/experiments/build/libsquatter.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void squat_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = (char)threadIdx.x;  // touch the allocation
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && lds[(threadIdx.x * 7) & 255] == 123 && ticks == 0) *sink = 1;  // keeps the LDS write alive
}

extern "C" int squat_launch(int ncu, int us, int lds_bytes, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(squat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return -1;
        attr = true;
    }
    hipLaunchKernelGGL(squat_kernel, dim3(ncu), dim3(256), lds_bytes, reinterpret_cast<hipStream_t>(stream),
                       (unsigned long long)us * 100ull, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
