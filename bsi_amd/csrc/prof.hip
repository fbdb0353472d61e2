// Optional in-library kernel timing with HIP events (used by bench.py to measure the dominant kernel's
// average launch duration live, on the stream the kernels are launched on).  Disabled by default: when the
// class mask is 0 the hooks are a single predictable branch.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"
#include "prof.h"

namespace {
struct Pair {
    hipEvent_t a, b;
};
unsigned g_mask = 0;
std::vector<Pair> g_rec[BSI_PROF_NCLASS];
std::vector<hipEvent_t> g_pool;
std::mutex g_mu;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

thread_local int g_bsi_cu_reserve = 0;  // see compute_cus(), common.h
thread_local int g_bsi_cu_masked = 0;
thread_local int g_bsi_ln_stream_cus = [] {
    const char* e = getenv("BSI_LN_STREAM_CUS");  // experiments: the persistent LayerNorm pass on the whole chip (256) or a part of it
    return e ? atoi(e) : 0;
}();

extern "C" int bsi_set_cu_reserve(int cus) {
    BSI_CHECK_ARG(cus >= 0 && cus % 8 == 0 && cus <= BSI_MAX_CU_RESERVE,
                  "bsi_set_cu_reserve: %d is not 0 or a multiple of 8 up to %d (workgroups are dealt to the 8 XCDs round robin)", cus,
                  BSI_MAX_CU_RESERVE);
    g_bsi_cu_reserve = cus;
    return BSI_OK;
}

extern "C" int bsi_compute_cus(void) { return compute_cus(); }

extern "C" int bsi_set_ln_stream_cus(int cus) {
    BSI_CHECK_ARG(cus >= 0 && cus <= 1024, "bsi_set_ln_stream_cus: %d is not in [0, 1024]", cus);
    g_bsi_ln_stream_cus = cus;
    return BSI_OK;
}

// ---- tile queue control blocks (common.h) --------------------------------------------------------------------------------------
thread_local int g_bsi_tile_queue = [] {
    const char* e = getenv("BSI_TILE_QUEUE");
    return e ? atoi(e) : 0;
}();

extern "C" int bsi_set_tile_queue(int on) {
    BSI_CHECK_ARG(on == 0 || on == 1, "bsi_set_tile_queue: %d is not 0 or 1", on);
    g_bsi_tile_queue = on;
    return BSI_OK;
}

namespace {
constexpr int TQ_STREAMS = 64, TQ_DEVICES = 16;
struct TqPool {
    unsigned* base = nullptr;     // TQ_STREAMS zeroed blocks of BSI_TQ_WORDS words
    hipStream_t owner[TQ_STREAMS] = {};
    bool used[TQ_STREAMS] = {};
    bool failed = false;
};
TqPool g_tq[TQ_DEVICES];
std::mutex g_tq_mu;
}  // namespace

#ifdef BSI_LAB
unsigned* g_lab_static_block = nullptr;  // gemm_bf16.hip (laboratory build): the stamp block of the static schedule
extern "C" void* bsi_lab_tile_queue_block(void* stream) {
    return g_bsi_tile_queue ? bsi_tile_queue_block(reinterpret_cast<hipStream_t>(stream)) : g_lab_static_block;
}
#endif

unsigned* bsi_tile_queue_block(hipStream_t s) {
    if (!g_bsi_tile_queue) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TQ_DEVICES) return nullptr;
    // A launch that is being CAPTURED takes the static schedule, always: a graph would bake this stream's control block into its
    // kernel node and share its ticket counters with whatever stream the graph is replayed on, concurrently with eager launches.
    // (The pool is also allocated and zeroed synchronously at the first launch that wants it, which cannot happen inside a capture.)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lk(g_tq_mu);
    TqPool& pl = g_tq[dev];
    if (!pl.base) {
        if (pl.failed) return nullptr;
        void* mem = nullptr;
        const size_t bytes = (size_t)TQ_STREAMS * BSI_TQ_WORDS * sizeof(unsigned);
        if (hipMalloc(&mem, bytes) != hipSuccess || hipMemset(mem, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            pl.failed = true;
            return nullptr;
        }
        pl.base = reinterpret_cast<unsigned*>(mem);
    }
    int free_slot = -1;
    for (int i = 0; i < TQ_STREAMS; ++i) {
        if (pl.used[i] && pl.owner[i] == s) return pl.base + (size_t)i * BSI_TQ_WORDS;
        if (!pl.used[i] && free_slot < 0) free_slot = i;
    }
    if (free_slot < 0) return nullptr;  // more streams than blocks: static schedule
    pl.used[free_slot] = true;
    pl.owner[free_slot] = s;
    return pl.base + (size_t)free_slot * BSI_TQ_WORDS;
}

void bsi_prof_begin(int cls, hipStream_t s) {
    if (!(g_mask & (1u << cls))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pair p{get_event(), get_event()};
    (void)hipEventRecord(p.a, s);
    g_rec[cls].push_back(p);
}

void bsi_prof_end(int cls, hipStream_t s) {
    if (!(g_mask & (1u << cls))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rec[cls].empty()) (void)hipEventRecord(g_rec[cls].back().b, s);
}

extern "C" int bsi_prof_enable(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
    return BSI_OK;
}

extern "C" int bsi_prof_read(int cls, int* count, double* total_ms) {
    BSI_CHECK_ARG(cls >= 0 && cls < BSI_PROF_NCLASS && count && total_ms, "bsi_prof_read: bad args");
    std::lock_guard<std::mutex> lk(g_mu);
    double tot = 0.0;
    int n = 0;
    for (Pair& p : g_rec[cls]) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            tot += ms;
            ++n;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_rec[cls].clear();
    *count = n;
    *total_ms = tot;
    return BSI_OK;
}

// Shader-clock probe: one wave spins for ~`us` microseconds of the constant 100 MHz counter and reports
// {shader cycles, 100 MHz ticks}; launched right after a kernel it shows the DVFS state that kernel left the chip in
// (power management reacts over milliseconds).  Diagnostic only.
namespace {
__global__ void clock_probe_kernel(unsigned long long* out, unsigned ticks) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        r = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) {
        out[0] = __builtin_readcyclecounter() - c0;
        out[1] = r - r0;
    }
}
}  // namespace

extern "C" int bsi_clock_probe(unsigned long long* out /*device, 2 words*/, int us, bsi_stream_t stream) {
    BSI_CHECK_ARG(out && us > 0 && us <= 100000, "bsi_clock_probe: bad args");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, (unsigned)us * 100u);
    BSI_CHECK_LAUNCH("bsi_clock_probe");
    return BSI_OK;
}

// Box yardstick for the benchmark line (bench.py `summary.mfma_probe_*`): what THIS GPU sustains on a register-only stream of
// v_mfma_f32_16x16x32_bf16 with random operands -- the instruction mix of the engine's GEMMs without LDS, memory or barriers
// (tools/experiments/mfma_power.hip shape 0: 2.03 PFLOP/s at 1.9-1.95 GHz on the boxes of round 3).  The chip is power limited under
// this load, and boxes of the pool differ by a few per cent: headline / probe is comparable across boxes, the headline alone is not.
namespace {
__global__ __launch_bounds__(512, 1) void mfma_probe_kernel(int iters, unsigned long long* __restrict__ clk, float* __restrict__ sink) {
    const unsigned t = threadIdx.x + 512u * blockIdx.x;
    bf16x8 a[2][8], b[2][4];
    auto rnd = [&](unsigned k) {  // a bf16x8 of hashed values in [-1, 1)
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned h = (t * 0x9E3779B1u) ^ ((k * 8u + e) * 0x85EBCA77u);
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
            v[e] = (__bf16)((float)(int)h * (1.0f / 2147483648.0f));
        }
        return v;
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[s][i] = rnd(s * 12 + i);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[s][i] = rnd(s * 12 + 8 + i);
    }
    f32x4 acc[8][4] = {};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    if (r == 12345.678f) sink[0] = r;  // keeps the arithmetic alive
    if ((threadIdx.x & 63) == 0) {  // per WAVE: the two waves of a SIMD take turns rather than interleave, so one wave's span is not the kernel's
        unsigned long long* w = clk + 4 * (size_t)(blockIdx.x * 8 + (threadIdx.x >> 6));
        w[0] = c1 - c0;
        w[1] = r1 - r0;
        w[2] = r0;
        w[3] = r1;
    }
}
}  // namespace

extern "C" size_t bsi_mfma_probe_workspace_bytes(void) { return (size_t)(4 * 8 * 1024 + 2) * sizeof(unsigned long long); }

extern "C" int bsi_mfma_probe(int iters, void* workspace, int* workgroups, double* flop_per_workgroup, bsi_stream_t stream) {
    BSI_CHECK_ARG(iters > 0 && iters <= 2000000 && workspace && workgroups && flop_per_workgroup, "bsi_mfma_probe: bad args");
    const int grid = device_cus() < 1024 ? device_cus() : 1024;
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), iters, clk,
                       reinterpret_cast<float*>(clk + 4 * 8 * 1024));
    BSI_CHECK_LAUNCH("bsi_mfma_probe");
    *workgroups = grid;
    *flop_per_workgroup = 8.0 * iters * 64.0 * (2.0 * 16 * 16 * 32);  // 8 waves x iters x 64 MFMAs x 2*16*16*32
    return BSI_OK;
}
