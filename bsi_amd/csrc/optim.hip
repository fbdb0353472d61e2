// Train-step tail on flat fp32 buffers: global gradient norm, then ONE fused pass doing
// clip_grad_norm_(max_norm) + AdamW (decoupled weight decay, bias correction) + EMA lerp
// (config/train.yaml:40 `gradient_clip_val`, config/task/optimizer/adamw.yaml, bsi/tasks/ema_pytorch.py:316-434).
// HBM-bound: norm pass 4 B/param, update pass reads p,g,m,v,ema and writes p,m,v,ema = 36 B/param.
#include "common.h"

namespace {

__global__ void sqnorm_partial_kernel(const float* __restrict__ g, size_t n4, size_t n, float* __restrict__ partial) {
    __shared__ float sm[4];
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        a = __fmaf_rn(v[0], v[0], a); a = __fmaf_rn(v[1], v[1], a); a = __fmaf_rn(v[2], v[2], a); a = __fmaf_rn(v[3], v[3], a);
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) a = __fmaf_rn(g[i], g[i], a);
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ void sqnorm_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
    __shared__ double sm[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) a += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

struct AdamArgs {
    float max_norm, grad_scale, lr, beta1, beta2, eps, wd, bc1, sqrt_bc2, ema_w;
};

__global__ void clip_adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                      float* __restrict__ v, float* __restrict__ ema, size_t n,
                                      const float* __restrict__ sqnorm, AdamArgs a) {
    // clip coefficient: max_norm / (total_norm + 1e-6), clamped to 1 (torch.nn.utils.clip_grad_norm_)
    float coef = a.grad_scale;
    if (a.max_norm > 0.f) {
        const float total = sqrtf(sqnorm[0]) * a.grad_scale;
        coef *= fminf(a.max_norm / (total + 1e-6f), 1.0f);
    }
    const float step = a.lr / a.bc1;
    const float decay = 1.0f - a.lr * a.wd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * decay;
        const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;
        const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
        const float denom = sqrtf(vi) / a.sqrt_bc2 + a.eps;
        pi -= step * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema && a.ema_w >= 0.f) {
            const float e = ema[i];
            ema[i] = (a.ema_w >= 1.0f) ? pi : e + a.ema_w * (pi - e);  // lerp_(p, 1 - decay); copy while warming up
        }
    }
}


// ---- segment forms (data-parallel step, bsi_amd/dp.py): the flat buffers are cut into SEGMENTS -- the (bucket, rank) slices of the
// gradient exchange -- and every segment into chunks of BSI_SQNORM_CHUNK elements.  The squared norm is one fp32 partial per chunk
// (fixed order inside the chunk) summed in chunk order by sqnorm_final_kernel: whoever computes a chunk -- every rank on the
// all-reduced gradient, or only the rank that owns the slice after a reduce-scatter -- gets the same bits, and adding the ranks'
// partial arrays (zeros outside a rank's own chunks) is exact.  The update takes the same table: parameters / moments / EMA at
// p_off, the gradient at g_off (its own compact buffer in the sharded step).
constexpr int CHUNK = BSI_SQNORM_CHUNK;
static_assert(CHUNK == 256 * 16 * 4, "one chunk = 256 threads x 16 float4");

__device__ __forceinline__ int find_segment(const bsi_seg* __restrict__ segs, int nseg, size_t c) {
    int lo = 0, hi = nseg - 1;  // last segment whose my_chunk <= c (uniform over the workgroup: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].my_chunk <= c) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sqnorm_segments_kernel(const float* __restrict__ g, const bsi_seg* __restrict__ segs, int nseg,
                                                              size_t nchunks, float* __restrict__ partial) {
    __shared__ float sm[4];
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int si = find_segment(segs, nseg, c);
        const bsi_seg sg = segs[si];
        const size_t k = c - sg.my_chunk, base = k * CHUNK;
        const size_t left = sg.len - base;             // elements of this chunk (a multiple of 4: bsi_seg contract)
        const f32x4* src = reinterpret_cast<const f32x4*>(g + sg.g_off + base);
        float a = 0.f;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const size_t i = (size_t)j * 256 + threadIdx.x;
            if (i * 4 < left) {
                const f32x4 v = src[i];
                a = __fmaf_rn(v[0], v[0], a); a = __fmaf_rn(v[1], v[1], a); a = __fmaf_rn(v[2], v[2], a); a = __fmaf_rn(v[3], v[3], a);
            }
        }
        a = wave_sum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) partial[sg.out_chunk + k] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

__global__ __launch_bounds__(256) void clip_adamw_ema_segments_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                      float* __restrict__ m, float* __restrict__ v,
                                                                      float* __restrict__ ema, const bsi_seg* __restrict__ segs,
                                                                      int nseg, size_t nchunks, const float* __restrict__ sqnorm,
                                                                      AdamArgs a) {
    float coef = a.grad_scale;
    if (a.max_norm > 0.f) {
        const float total = sqrtf(sqnorm[0]) * a.grad_scale;
        coef *= fminf(a.max_norm / (total + 1e-6f), 1.0f);
    }
    const float step = a.lr / a.bc1;
    const float decay = 1.0f - a.lr * a.wd;
    const bool do_ema = ema && a.ema_w >= 0.f;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int si = find_segment(segs, nseg, c);
        const bsi_seg sg = segs[si];
        const size_t base = (c - sg.my_chunk) * CHUNK;
        const size_t left = sg.len - base;
        const size_t po = sg.p_off + base, go = sg.g_off + base;
#pragma unroll 2
        for (int j = 0; j < 16; ++j) {
            const size_t i = ((size_t)j * 256 + threadIdx.x) * 4;
            if (i >= left) break;
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + go + i);
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + po + i);
            f32x4 mv = *reinterpret_cast<const f32x4*>(m + po + i);
            f32x4 vv = *reinterpret_cast<const f32x4*>(v + po + i);
            f32x4 ev = do_ema ? *reinterpret_cast<const f32x4*>(ema + po + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // the arithmetic of clip_adamw_ema_kernel, element for element
                const float gi = gv[e] * coef;
                float pi = pv[e] * decay;
                const float mi = a.beta1 * mv[e] + (1.0f - a.beta1) * gi;
                const float vi = a.beta2 * vv[e] + (1.0f - a.beta2) * gi * gi;
                const float denom = sqrtf(vi) / a.sqrt_bc2 + a.eps;
                pi -= step * (mi / denom);
                pv[e] = pi; mv[e] = mi; vv[e] = vi;
                if (do_ema) ev[e] = (a.ema_w >= 1.0f) ? pi : ev[e] + a.ema_w * (pi - ev[e]);
            }
            *reinterpret_cast<f32x4*>(p + po + i) = pv;
            *reinterpret_cast<f32x4*>(m + po + i) = mv;
            *reinterpret_cast<f32x4*>(v + po + i) = vv;
            if (do_ema) *reinterpret_cast<f32x4*>(ema + po + i) = ev;
        }
    }
}

}  // namespace

extern "C" size_t bsi_sqnorm_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int bsi_grad_sqnorm(const float* g, size_t n, float* out_sq, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(g && out_sq && workspace && n > 0, "bsi_grad_sqnorm: bad args");
    BSI_CHECK_ARG((reinterpret_cast<uintptr_t>(g) & 15) == 0, "bsi_grad_sqnorm: buffer must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t n4 = n / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(grid), dim3(256), 0, s, g, n4, n, part);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, part, grid, out_sq);
    BSI_CHECK_LAUNCH("bsi_grad_sqnorm");
    return BSI_OK;
}

extern "C" int bsi_clip_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, const float* sqnorm,
                                  float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int step, float ema_weight, bsi_stream_t stream) {
    BSI_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bsi_clip_adamw_ema: bad args");
    BSI_CHECK_ARG(max_norm <= 0.f || sqnorm, "bsi_clip_adamw_ema: clipping needs the squared norm");
    AdamArgs a;
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, step));
    a.ema_w = ema_weight;
    size_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(clip_adamw_ema_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                       ema, n, sqnorm, a);
    BSI_CHECK_LAUNCH("bsi_clip_adamw_ema");
    return BSI_OK;
}

extern "C" int bsi_sqnorm_segments(const float* g, const bsi_seg* segs, int nseg, size_t nchunks, float* partials, bsi_stream_t stream) {
    BSI_CHECK_ARG(g && segs && partials && nseg > 0 && nchunks > 0, "bsi_sqnorm_segments: bad args");
    BSI_CHECK_ARG((reinterpret_cast<uintptr_t>(g) & 15) == 0, "bsi_sqnorm_segments: buffer must be 16-byte aligned");
    const size_t grid = nchunks < 4096 ? nchunks : 4096;
    hipLaunchKernelGGL(sqnorm_segments_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, segs, nseg,
                       nchunks, partials);
    BSI_CHECK_LAUNCH("bsi_sqnorm_segments");
    return BSI_OK;
}

extern "C" int bsi_sqnorm_finish(const float* partials, size_t nchunks, float* out_sq, bsi_stream_t stream) {
    BSI_CHECK_ARG(partials && out_sq && nchunks > 0 && nchunks < (size_t)1 << 30, "bsi_sqnorm_finish: bad args");
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partials, (int)nchunks, out_sq);
    BSI_CHECK_LAUNCH("bsi_sqnorm_finish");
    return BSI_OK;
}

extern "C" int bsi_clip_adamw_ema_segments(float* p, const float* g, float* m, float* v, float* ema, const bsi_seg* segs, int nseg,
                                           size_t nchunks, const float* sqnorm, float max_norm, float grad_scale, float lr, float beta1,
                                           float beta2, float eps, float weight_decay, int step, float ema_weight, bsi_stream_t stream) {
    BSI_CHECK_ARG(p && g && m && v && segs && nseg > 0 && nchunks > 0 && step >= 1, "bsi_clip_adamw_ema_segments: bad args");
    BSI_CHECK_ARG(max_norm <= 0.f || sqnorm, "bsi_clip_adamw_ema_segments: clipping needs the squared norm");
    BSI_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(ema)) & 15) == 0,
                  "bsi_clip_adamw_ema_segments: buffers must be 16-byte aligned");
    AdamArgs a;
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, step));
    a.ema_w = ema_weight;
    const size_t grid = nchunks < 8192 ? nchunks : 8192;
    hipLaunchKernelGGL(clip_adamw_ema_segments_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                       ema, segs, nseg, nchunks, sqnorm, a);
    BSI_CHECK_LAUNCH("bsi_clip_adamw_ema_segments");
    return BSI_OK;
}


// ---- batched fp32 copy (round 5): n jobs (src, dst, len) in one launch.  The VDM-UNet backward hands the stacked FiLM gradients and the
// shared conv2 / skip bias gradients to their parameters' places in the flat gradient buffer with it: torch._foreach_copy_ issues one
// hipMemcpyAsync per pair on ROCm (167 per step, 3 % of the UNet train step's GPU time and ~1 ms of host time).
namespace {
constexpr int CPY_TILE = 2048;  // elements per workgroup
__global__ __launch_bounds__(256) void copy_batch_kernel(const bsi_copy_desc* __restrict__ descs, int n) {
    int lo = 0, hi = n - 1;  // the job of this tile: the last descriptor whose tile0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const bsi_copy_desc d = descs[lo];
    const size_t e0 = (size_t)(blockIdx.x - d.tile0) * CPY_TILE;
    const size_t left = d.len - e0;
    const int cnt = left < (size_t)CPY_TILE ? (int)left : CPY_TILE;
    const float* s = d.src + e0;
    float* t = d.dst + e0;
    if ((((uintptr_t)s | (uintptr_t)t) & 15) == 0) {
        const int c4 = cnt >> 2;
        for (int i = threadIdx.x; i < c4; i += 256) reinterpret_cast<float4*>(t)[i] = reinterpret_cast<const float4*>(s)[i];
        for (int i = (c4 << 2) + threadIdx.x; i < cnt; i += 256) t[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) t[i] = s[i];
    }
}
}  // namespace

extern "C" int bsi_copy_batch_tiles(size_t len) { return (int)((len + CPY_TILE - 1) / CPY_TILE); }

extern "C" int bsi_copy_batch_f32(const bsi_copy_desc* descs, int n, int tiles, bsi_stream_t stream) {
    if (n <= 0 || tiles <= 0) return BSI_OK;
    if (descs == nullptr) return BSI_EINVAL;
    hipLaunchKernelGGL(copy_batch_kernel, dim3(tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs, n);
    BSI_CHECK_LAUNCH("bsi_copy_batch_f32");
    return BSI_OK;
}
