// Elementwise / per-row kernels of the two other algorithm wrappers of the reference that share the denoisers with BSI
// (SURVEY §8(f) rank 4): Variational Diffusion Models (bsi/vdm.py) and Bayesian Flow Networks (bsi/bfn.py).
// Everything here is HBM- or latency-bound fp32 arithmetic that follows the reference's op order (separate roundings
// where torch issues separate ops, fma where it calls addcmul); the denoiser evaluation in between is the DiT / UNet
// engine.  Row r of an [n_samples, batch] quantity is r = s * B + b; `x` is broadcast over samples (row r reads x[r % B]).
#include <math.h>

#include "common.h"

namespace {

constexpr int TPB = 256;

inline dim3 row_grid(int rows, int D4) {
    int gx = (D4 + TPB - 1) / TPB;
    if (gx > 64) gx = 64;
    return dim3(gx, rows);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// t = (perm / (1 + total) + offset) mod 1   (vdm.py:385-397, bfn.py:313-325: the low-discrepancy time grid)
__global__ void tgrid_kernel(const int64_t* perm, const float* offset, int total, float* t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float g = (float)perm[i] / (float)(1 + total);
    const float s = __fadd_rn(g, offset[0]);
    float r = s - floorf(s);
    if (r >= 1.0f) r = 0.0f;
    t[i] = r;
}

// out[r] = addcmul(a[r] * x[r % B], b[r], eps[r])   (vdm.py:343-347, bfn.py:302-309)
__global__ void affine_noise_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ eps, int B, int D4, float* __restrict__ out) {
    const int r = blockIdx.y;
    const float ar = a[r], br = b[r];
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + (size_t)(r % B) * D4;
    const f32x4* er = reinterpret_cast<const f32x4*>(eps) + (size_t)r * D4;
    f32x4* o = reinterpret_cast<f32x4*>(out) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        const f32x4 xv = xr[i], ev = er[i];
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __fmaf_rn(br, ev[k], __fmul_rn(ar, xv[k]));
        o[i] = v;
    }
}

// out[r] = a[ia] * z[r] + b[ib] * xh[r] + c[ic] * eps[r] with per-call scalar indices (one sampler step for all rows):
//   VDM ancestral step z_s = c_z z_t + c_x x_hat + std eps   (vdm.py:350-379)
__global__ void axpbypcz_kernel(const float* __restrict__ z, const float* __restrict__ xh, const float* __restrict__ eps,
                                const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int idx,
                                size_t n4, float* __restrict__ out) {
    const float av = a[idx], bv = b[idx], cv = c[idx];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i], xv = reinterpret_cast<const f32x4*>(xh)[i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[i];
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mean = __fadd_rn(__fmul_rn(av, zv[k]), __fmul_rn(bv, xv[k]));  // a*z + b*x (two products, one add)
            v[k] = __fmaf_rn(cv, ev[k], mean);                                         // addcmul(mean, std, eps)
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// out = clamp(x, lo, hi)  (bfn.py:291 .clip);  backward: g where lo <= raw <= hi else 0 (torch.clamp's subgradient)
__global__ void clip_kernel(const float* __restrict__ x, float lo, float hi, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(x[i], lo), hi);
}
__global__ void clip_bwd_kernel(const float* __restrict__ g, const float* __restrict__ raw, float lo, float hi, size_t n,
                                float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float r = raw[i];
        out[i] = (r >= lo && r <= hi) ? g[i] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// VDM (bsi/vdm.py): gamma(t) = lerp(gamma_0, gamma_1, t), sigma2 = sigmoid(gamma), alpha = sqrt(sigmoid(-gamma)),
// snr = exp(-gamma) (138-150).  Per time t[i]:
//   alpha[i], sigma[i] = sqrt(sigma2), snr[i], and the _predict_x coefficients (324-329)
//   x_hat = (z - sigma f) / alpha  ->  c_skip = 1/alpha, c_out = -sigma/alpha.
__device__ __forceinline__ float vdm_gamma(float g0, float g1, float t) {
    // torch.lerp(start, end, w): start + w*(end-start) for w < 0.5, end - (end-start)*(1-w) otherwise
    const float d = __fsub_rn(g1, g0);
    return t < 0.5f ? __fmaf_rn(t, d, g0) : __fsub_rn(g1, __fmul_rn(d, __fsub_rn(1.0f, t)));
}
__global__ void vdm_coeffs_kernel(const float* t, int n, float g0, float g1, float* alpha, float* sigma, float* snr, float* c_skip,
                                  float* c_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = vdm_gamma(g0, g1, t[i]);
    const float a = sqrtf(sigmoid_f(-g)), s = sqrtf(sigmoid_f(g));
    if (alpha) alpha[i] = a;
    if (sigma) sigma[i] = s;
    if (snr) snr[i] = expf(-g);
    if (c_skip) c_skip[i] = 1.0f / a;
    if (c_out) c_out[i] = -(s / a);
}

// Ancestral-sampling coefficients over a schedule t[0..k] (vdm.py:350-379), step i goes from t = t[i] to s = t[i+1]:
//   r = -expm1(softplus(-g_t) - softplus(g_t) - softplus(-g_s) + softplus(g_s))           (sigma2_{t|s} / sigma2_t)
//   c_z[i] = exp(0.5 (softplus(g_s) - softplus(g_t)) + softplus(-g_t) - softplus(-g_s)),  c_x[i] = alpha(s) * r,
//   std[i] = sqrt(sigma2(s) * r)
__global__ void vdm_step_coeffs_kernel(const float* t, int k, float g0, float g1, float* c_z, float* c_x, float* std_, float* dsnr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const float gt = vdm_gamma(g0, g1, t[i]), gs = vdm_gamma(g0, g1, t[i + 1]);
    if (dsnr) dsnr[i] = expf(-gs) - expf(-gt);  // snr(s_i) - snr(t_i), the finite diffusion loss weight (vdm.py:231)
    const float spt = softplus_f(gt), smt = softplus_f(-gt), sps = softplus_f(gs), sms = softplus_f(-gs);
    const float r = -expm1f(((smt - spt) - sms) + sps);
    c_z[i] = expf((0.5f * (sps - spt) + smt) - sms);
    c_x[i] = sqrtf(sigmoid_f(-gs)) * r;
    std_[i] = sqrtf(sigmoid_f(gs) * r);
}

// Sum of `acc` over the workgroup, waves added in index order through LDS (reproducible; one workgroup owns one output row, so
// the row needs neither atomics nor a zero fill).  Returned to thread 0.
__device__ __forceinline__ float block_sum_ordered(float acc) {
    __shared__ float wsum[16];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    return t;
}

// VDM reconstruction term with a discretisation (vdm.py:152-195): x_hat = z_0 / alpha_0, the Normal(x_hat, std) density is
// evaluated at the k bin centres and normalised (log_softmax over bins); out[r] = -sum_D log p[bin(x)].
// One wave per (row, 64-element chunk); every lane owns one element and loops over the bins.
__global__ void vdm_recon_nll_kernel(const float* __restrict__ x, const float* __restrict__ x_hat, float std_, const float* __restrict__ bounds,
                                     float lo_edge, float dx, int k, int B, int D, float* __restrict__ out) {
    extern __shared__ float centers[];
    for (int j = threadIdx.x; j < k; j += blockDim.x) centers[j] = (bounds[j + 1] + bounds[j]) / 2.0f;
    __syncthreads();
    const int r = blockIdx.y;
    const float* xr = x + (size_t)(r % B) * D;
    const float* hr = x_hat + (size_t)r * D;
    const float log_norm = -logf(std_) - 0.9189385332046727f;  // Normal.log_prob: -((v-mu)^2)/(2 var) - log(std) - log(sqrt(2 pi))
    const float var2 = 2.0f * (std_ * std_);
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
        const float mu = hr[i], xv = xr[i];
        int idx = (int)((xv - lo_edge) / dx);
        idx = idx < 0 ? 0 : (idx > k - 1 ? k - 1 : idx);
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) {
            const float dlt = centers[j] - mu;
            mx = fmaxf(mx, -(dlt * dlt) / var2 + log_norm);
        }
        float se = 0.f;
        for (int j = 0; j < k; ++j) {
            const float dlt = centers[j] - mu;
            se += expf((-(dlt * dlt) / var2 + log_norm) - mx);
        }
        const float dl = centers[idx] - mu;
        acc += ((-(dl * dl) / var2 + log_norm) - mx) - logf(se);
    }
    const float tot = block_sum_ordered(acc);
    if (threadIdx.x == 0) out[r] = -tot;
}

// out[r] = 0.5 * sum_D (var1 + (1 - var1) x^2 - log(var1) - 1)   (prior_loss, vdm.py:127-136), var1 = sigma2(t = 1)
__global__ void vdm_prior_kernel(const float* __restrict__ x, float var1, int D, float* __restrict__ out) {
    const int r = blockIdx.y;
    const float lv = logf(var1), omv = 1.0f - var1;
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
        const float v = x[(size_t)r * D + i];
        acc += ((var1 + omv * (v * v)) - lv) - 1.0f;
    }
    const float tot = block_sum_ordered(acc);
    if (threadIdx.x == 0) out[r] = 0.5f * tot;
}

// ---------------------------------------------------------------------------------------------------------------------
// BFN (bsi/bfn.py), gamma(t) = 1 - sigma_1^(2 t).  Per time t[i]:
//   flow distribution (302-309):  fa = gamma, fb = sqrt(gamma (1 - gamma))
//   _predict_x (282-292) with g = 1 - sigma_1^(2 max(t, t_min)):  x_hat = clip(mu/g - sqrt((1-g)/g) eps_hat), 0 where t < t_min
//                                 ->  c_skip = 1/g, c_out = -sqrt((1-g)/g)  (both 0 where t < t_min)
//   loss weight (197-198):        w = sigma_1^(-2 t)
__global__ void bfn_coeffs_kernel(const float* t, int n, float sigma_1, float t_min, float* fa, float* fb, float* c_skip, float* c_out,
                                  float* w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float tv = t[i];
    const float gam = 1.0f - powf(sigma_1, 2.0f * tv);
    if (fa) fa[i] = gam;
    if (fb) fb[i] = sqrtf(gam * (1.0f - gam));
    const float g = 1.0f - powf(sigma_1, 2.0f * fmaxf(tv, t_min));
    const bool off = tv < t_min;
    if (c_skip) c_skip[i] = off ? 0.0f : 1.0f / g;
    if (c_out) c_out[i] = off ? 0.0f : -sqrtf((1.0f - g) / g);
    if (w) w[i] = powf(sigma_1, -2.0f * tv);
}

// Sampler schedule (bfn.py:215-226): alpha[i] = sigma_1^(-2 t[i+1]) (1 - sigma_1^(2 (t[i+1] - t[i]))), rho[0] = 1,
// rho[i+1] = rho[i] + alpha[i]  (sequential, one thread: k is 50..1000)
// wdisc[i] = sigma_1^((-2/k)(i+1)): the discrete-time loss weight of step i (bfn.py:176-181)
__global__ void bfn_schedule_kernel(const float* t, int k, float sigma_1, float* alpha, float* rho, float* wdisc) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float r = 1.0f;
    rho[0] = r;
    for (int i = 0; i < k; ++i) {
        const float a = powf(sigma_1, -2.0f * t[i + 1]) * (1.0f - powf(sigma_1, 2.0f * (t[i + 1] - t[i])));
        alpha[i] = a;
        r = r + a;
        rho[i + 1] = r;
        if (wdisc) wdisc[i] = powf(sigma_1, (-2.0f / (float)k) * (float)(i + 1));
    }
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int bsi_tgrid(const int64_t* perm, const float* offset, int total, float* t, bsi_stream_t stream) {
    BSI_CHECK_ARG(perm && offset && t && total > 0, "bsi_tgrid: bad args");
    hipLaunchKernelGGL(tgrid_kernel, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, S(stream), perm, offset, total, t);
    BSI_CHECK_LAUNCH("bsi_tgrid");
    return BSI_OK;
}

extern "C" int bsi_affine_noise(const float* x, const float* a, const float* b, const float* eps, int rows, int B, int D, float* out,
                                bsi_stream_t stream) {
    BSI_CHECK_ARG(x && a && b && eps && out && rows > 0 && B > 0 && D > 0 && D % 4 == 0, "bsi_affine_noise: bad args");
    hipLaunchKernelGGL(affine_noise_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), x, a, b, eps, B, D / 4, out);
    BSI_CHECK_LAUNCH("bsi_affine_noise");
    return BSI_OK;
}

extern "C" int bsi_axpbypcz(const float* z, const float* xh, const float* eps, const float* a, const float* b, const float* c, int idx,
                            size_t n, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(z && xh && eps && a && b && c && out && n > 0 && n % 4 == 0 && idx >= 0, "bsi_axpbypcz: bad args");
    size_t g = (n / 4 + TPB - 1) / TPB;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(axpbypcz_kernel, dim3((int)g), dim3(TPB), 0, S(stream), z, xh, eps, a, b, c, idx, n / 4, out);
    BSI_CHECK_LAUNCH("bsi_axpbypcz");
    return BSI_OK;
}

extern "C" int bsi_clip(const float* x, float lo, float hi, size_t n, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && out && n > 0 && lo <= hi, "bsi_clip: bad args");
    size_t g = (n + TPB - 1) / TPB;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(clip_kernel, dim3((int)g), dim3(TPB), 0, S(stream), x, lo, hi, n, out);
    BSI_CHECK_LAUNCH("bsi_clip");
    return BSI_OK;
}

extern "C" int bsi_clip_bwd(const float* g, const float* raw, float lo, float hi, size_t n, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(g && raw && out && n > 0, "bsi_clip_bwd: bad args");
    size_t gr = (n + TPB - 1) / TPB;
    if (gr > 8192) gr = 8192;
    hipLaunchKernelGGL(clip_bwd_kernel, dim3((int)gr), dim3(TPB), 0, S(stream), g, raw, lo, hi, n, out);
    BSI_CHECK_LAUNCH("bsi_clip_bwd");
    return BSI_OK;
}

extern "C" int bsi_vdm_coeffs(const float* t, int n, float gamma_0, float gamma_1, float* alpha, float* sigma, float* snr, float* c_skip,
                              float* c_out, bsi_stream_t stream) {
    BSI_CHECK_ARG(t && n > 0, "bsi_vdm_coeffs: bad args");
    hipLaunchKernelGGL(vdm_coeffs_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, S(stream), t, n, gamma_0, gamma_1, alpha, sigma, snr,
                       c_skip, c_out);
    BSI_CHECK_LAUNCH("bsi_vdm_coeffs");
    return BSI_OK;
}

extern "C" int bsi_vdm_step_coeffs(const float* t, int k, float gamma_0, float gamma_1, float* c_z, float* c_x, float* std_,
                                   float* dsnr, bsi_stream_t stream) {
    BSI_CHECK_ARG(t && c_z && c_x && std_ && k > 0, "bsi_vdm_step_coeffs: bad args");
    hipLaunchKernelGGL(vdm_step_coeffs_kernel, dim3((k + TPB - 1) / TPB), dim3(TPB), 0, S(stream), t, k, gamma_0, gamma_1, c_z, c_x,
                       std_, dsnr);
    BSI_CHECK_LAUNCH("bsi_vdm_step_coeffs");
    return BSI_OK;
}

extern "C" int bsi_vdm_recon_nll(const float* x, const float* x_hat, float std_, const float* bounds, float lo_edge, float dx, int k,
                                 int rows, int B, int D, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && x_hat && bounds && out && std_ > 0.f && k > 0 && k <= 4096 && rows > 0 && B > 0 && D > 0,
                  "bsi_vdm_recon_nll: bad args");
    const int gx = 1;  // one workgroup per row: its sum is ordered (no atomics); rows = samples x batch give the parallelism
    hipLaunchKernelGGL(vdm_recon_nll_kernel, dim3(gx, rows), dim3(TPB), (size_t)k * sizeof(float), S(stream), x, x_hat, std_, bounds,
                       lo_edge, dx, k, B, D, out);
    BSI_CHECK_LAUNCH("bsi_vdm_recon_nll");
    return BSI_OK;
}

extern "C" int bsi_vdm_prior(const float* x, float var_1, int rows, int D, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && out && var_1 > 0.f && rows > 0 && D > 0, "bsi_vdm_prior: bad args");
    const int gx = 1;
    hipLaunchKernelGGL(vdm_prior_kernel, dim3(gx, rows), dim3(TPB), 0, S(stream), x, var_1, D, out);
    BSI_CHECK_LAUNCH("bsi_vdm_prior");
    return BSI_OK;
}

extern "C" int bsi_bfn_coeffs(const float* t, int n, float sigma_1, float t_min, float* fa, float* fb, float* c_skip, float* c_out,
                              float* w, bsi_stream_t stream) {
    BSI_CHECK_ARG(t && n > 0 && sigma_1 > 0.f && sigma_1 < 1.f, "bsi_bfn_coeffs: bad args");
    hipLaunchKernelGGL(bfn_coeffs_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, S(stream), t, n, sigma_1, t_min, fa, fb, c_skip,
                       c_out, w);
    BSI_CHECK_LAUNCH("bsi_bfn_coeffs");
    return BSI_OK;
}

extern "C" int bsi_bfn_schedule(const float* t, int k, float sigma_1, float* alpha, float* rho, float* wdisc, bsi_stream_t stream) {
    BSI_CHECK_ARG(t && alpha && rho && k > 0 && sigma_1 > 0.f && sigma_1 < 1.f, "bsi_bfn_schedule: bad args");
    hipLaunchKernelGGL(bfn_schedule_kernel, dim3(1), dim3(64), 0, S(stream), t, k, sigma_1, alpha, rho, wdisc);
    BSI_CHECK_LAUNCH("bsi_bfn_schedule");
    return BSI_OK;
}
