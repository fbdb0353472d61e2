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
