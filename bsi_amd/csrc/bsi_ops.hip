// Elementwise / row-reduction kernels of the BSI algorithm wrapper (bsi/bsi.py of the reference).
// All fp32, HBM-bound: one pass over the data, float4-coalesced where the layout allows it.
// The arithmetic mirrors the reference's op sequence (separate roundings where torch issues
// separate ops, fused multiply-add exactly where torch.addcmul fuses on CPU) so that fp32
// parity with the oracle is at the 1-ulp level; compile with -ffp-contract=off.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void bsi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bsi_last_error(void) { return g_err; }
extern "C" int bsi_version(void) { return 105; }  // 100 + the round that last added entry points (round 5: 122 of them)

namespace {

constexpr int TPB = 256;
inline int blocks_for(size_t n, int per_block = TPB) {
    size_t b = (n + per_block - 1) / per_block;
    return (int)(b > 0 ? b : 1);
}

__device__ __forceinline__ float icdf_f(float t, float delta, float ln_low) {
    // bsi.py:84  torch.exp(delta * q + ln_low): mul and add are separate torch ops
    return expf(__fadd_rn(__fmul_rn(delta, t), ln_low));
}
__device__ __forceinline__ float rsqrt_rn(float x) { return 1.0f / sqrtf(x); }  // torch.rsqrt on CPU: 1/sqrt

__global__ void edm_coeffs_kernel(const float* t, int n, float lambda_0, float delta, float ln_low, float* lam_o,
                                  float* cs_o, float* co_o, float* ci_o) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lam = icdf_f(t[i], delta, ln_low);
    float alpha = __fsub_rn(lam, lambda_0);
    float kappa = __fadd_rn(1.0f, __fmul_rn(alpha, alpha / lam));  // bsi.py:399
    if (lam_o) lam_o[i] = lam;
    if (cs_o) cs_o[i] = alpha / kappa;
    if (co_o) co_o[i] = rsqrt_rn(kappa);
    if (ci_o) ci_o[i] = sqrtf(lam / kappa);
}

__global__ void lambda_to_t_kernel(const float* lam, int n, float delta, float ln_low, float* t, float* rpdf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float l = lam[i];
    if (t) t[i] = __fsub_rn(logf(l), ln_low) / delta;  // bsi.py:81
    if (rpdf) rpdf[i] = __fmul_rn(l, delta);            // bsi.py:78
}

__global__ void schedule_kernel(const float* t, int k1, float delta, float ln_low, float* lam, float* alpha) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k1) return;
    float l = icdf_f(t[i], delta, ln_low);
    lam[i] = l;
    if (alpha && i + 1 < k1) alpha[i] = __fsub_rn(icdf_f(t[i + 1], delta, ln_low), l);  // lambda_.diff()
}

__global__ void lambda_grid_kernel(const int64_t* perm, const float* offset, int total, float delta, float ln_low,
                                   float* lam) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // bsi.py:434-439: randperm(total) / (1 + total)  [int64 / int -> fp32 true division], + offset, remainder 1
    float g = (float)perm[i] / (float)(1 + total);
    float s = __fadd_rn(g, offset[0]);
    float t = s - floorf(s);  // torch.remainder(s, 1) for s >= 0
    if (t >= 1.0f) t = 0.0f;
    lam[i] = icdf_f(t, delta, ln_low);
}

// rows x D elementwise kernels: grid.y = row, grid.x strides over D/4 float4 groups -----------------
__global__ void q_sample_kernel(const float* __restrict__ x, const float* __restrict__ lam,
                                const float* __restrict__ eps, float lambda_0, int B, int D4, float* __restrict__ mu) {
    const int r = blockIdx.y;
    const float l = lam[r];
    const float a = __fsub_rn(l, lambda_0) / l;  // (lam - lam0)/lam
    const float s = rsqrt_rn(l);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + (size_t)(r % B) * D4;
    const f32x4* er = reinterpret_cast<const f32x4*>(eps) + (size_t)r * D4;
    f32x4* mr = reinterpret_cast<f32x4*>(mu) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 xv = xr[i], ev = er[i], o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = __fmaf_rn(s, ev[c], __fmul_rn(a, xv[c]));  // addcmul(a*x, s, eps)
        mr[i] = o;
    }
}

__global__ void scale_rows_kernel(const float* __restrict__ mu, const float* __restrict__ c, int c_stride, int D4,
                                  float* __restrict__ out) {
    const int r = blockIdx.y;
    const float s = c[(size_t)r * c_stride];
    const f32x4* mr = reinterpret_cast<const f32x4*>(mu) + (size_t)r * D4;
    f32x4* o = reinterpret_cast<f32x4*>(out) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 v = mr[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __fmul_rn(s, v[k]);
        o[i] = v;
    }
}

// bsi.py:325: mu_0 = rsqrt(lam[0]) * eps0
__global__ void sample_init_kernel(const float* __restrict__ eps0, const float* __restrict__ lam, int D4,
                                   float* __restrict__ mu) {
    const int r = blockIdx.y;
    const float s = rsqrt_rn(lam[0]);
    const f32x4* er = reinterpret_cast<const f32x4*>(eps0) + (size_t)r * D4;
    f32x4* o = reinterpret_cast<f32x4*>(mu) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 v = er[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __fmul_rn(s, v[k]);
        o[i] = v;
    }
}

__global__ void predict_combine_kernel(const float* __restrict__ mu, const float* __restrict__ f,
                                       const float* __restrict__ cs, const float* __restrict__ co, int c_stride, int D4,
                                       float* __restrict__ xh) {
    const int r = blockIdx.y;
    const float a = cs[(size_t)r * c_stride], b = co[(size_t)r * c_stride];
    const f32x4* mr = reinterpret_cast<const f32x4*>(mu) + (size_t)r * D4;
    const f32x4* fr = reinterpret_cast<const f32x4*>(f) + (size_t)r * D4;
    f32x4* o = reinterpret_cast<f32x4*>(xh) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 m = mr[i], fv = fr[i], v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __fmaf_rn(b, fv[k], __fmul_rn(a, m[k]));  // addcmul(c_skip*mu, c_out, f)
        o[i] = v;
    }
}

__global__ void predict_combine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ cs,
                                           const float* __restrict__ co, int c_stride, int D4, float* __restrict__ gf,
                                           float* __restrict__ gmu) {
    const int r = blockIdx.y;
    const float a = cs[(size_t)r * c_stride], b = co[(size_t)r * c_stride];
    const f32x4* gr = reinterpret_cast<const f32x4*>(g) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 gv = gr[i];
        if (gf) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = b * gv[k];
            (reinterpret_cast<f32x4*>(gf) + (size_t)r * D4)[i] = v;
        }
        if (gmu) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = a * gv[k];
            (reinterpret_cast<f32x4*>(gmu) + (size_t)r * D4)[i] = v;
        }
    }
}

// Counter-based Gaussian noise for the measurement step y = x_hat + eps / sqrt(alpha) (bsi.py:331-333) generated IN the kernel:
// Philox4x32-10 keyed by a 64-bit seed that lives in device memory (drawn once per chain from the caller's torch.Generator, so
// no host synchronisation), counter = (index of the group of four elements, noise stream), Box-Muller on 24-bit uniforms.
// This is an opt-in (BSI.sample(device_noise=True)): the default path draws eps with the caller's generator exactly as the
// reference does (bsi.py:332-334), because a different generator is a different random stream.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, unsigned long long group, unsigned stream_id) {
    unsigned x[4];
    philox4x32_10((unsigned)group, (unsigned)(group >> 32), stream_id, 0x42534931u, (unsigned)seed, (unsigned)(seed >> 32), x);
    f32x4 n;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = (float)((x[2 * h] >> 8) + 1u) * 5.9604644775390625e-08f;   // (0, 1]
        const float u2 = (float)(x[2 * h + 1] >> 8) * 5.9604644775390625e-08f;      // [0, 1): v_sin / v_cos take revolutions
        const float r = sqrtf(-2.0f * __logf(u1));
        n[2 * h] = r * __builtin_amdgcn_cosf(u2);
        n[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
    }
    return n;
}

__global__ void philox_normal_kernel(const unsigned long long* __restrict__ seed, unsigned stream_id, size_t n4,
                                     float* __restrict__ out) {
    const unsigned long long sd = seed[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = philox_normal4(sd, i, stream_id);
}

// Known-answer access to the generator: raw Philox4x32-10 blocks for caller-given (counter, key) pairs, and the integer stream
// (before Box-Muller) in the library's counter / key layout.  Both run the device function the fused step uses.
__global__ void philox_blocks_kernel(const unsigned* __restrict__ ctr_key, size_t nblocks, unsigned* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned* c = ctr_key + 6 * i;
        unsigned x[4];
        philox4x32_10(c[0], c[1], c[2], c[3], c[4], c[5], x);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[4 * i + k] = x[k];
    }
}
__global__ void philox_uint32_kernel(const unsigned long long* __restrict__ seed, unsigned stream_id, size_t n4,
                                     unsigned* __restrict__ out) {
    const unsigned long long sd = seed[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x[4];
        philox4x32_10((unsigned)i, (unsigned)(i >> 32), stream_id, 0x42534931u, (unsigned)sd, (unsigned)(sd >> 32), x);
        reinterpret_cast<u32x4*>(out)[i] = u32x4{x[0], x[1], x[2], x[3]};
    }
}

// bsi.py:331-335: the fused measure/refine step.  Algorithmic traffic: read mu, f, eps + write mu' = 16 B/elt (12 with PHILOX).
template <bool PHILOX>
__global__ void refine_step_kernel(const float* __restrict__ mu, const float* __restrict__ f,
                                   const float* __restrict__ eps, const unsigned long long* __restrict__ seed,
                                   const float* __restrict__ lam,
                                   const float* __restrict__ alpha, const float* __restrict__ cs,
                                   const float* __restrict__ co, int step, int f_is_xhat, size_t n4,
                                   float* __restrict__ xh_out, float* __restrict__ y_out, float* __restrict__ mu_next) {
    const float a = alpha[step], l0 = lam[step], l1 = lam[step + 1];
    const float ra = rsqrt_rn(a);
    const float c_skip = f_is_xhat ? 0.f : cs[step], c_out = f_is_xhat ? 1.f : co[step];
    unsigned long long sd = 0;
    if constexpr (PHILOX) sd = seed[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 m = reinterpret_cast<const f32x4*>(mu)[i];
        const f32x4 fv = reinterpret_cast<const f32x4*>(f)[i];
        f32x4 e;
        if constexpr (PHILOX) e = philox_normal4(sd, i, (unsigned)step);
        else e = reinterpret_cast<const f32x4*>(eps)[i];
        f32x4 xh, y, mn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xh[k] = f_is_xhat ? fv[k] : __fmaf_rn(c_out, fv[k], __fmul_rn(c_skip, m[k]));
            y[k] = __fadd_rn(xh[k], __fmul_rn(ra, e[k]));                              // x_hat + rsqrt(alpha)*eps
            mn[k] = __fadd_rn(__fmul_rn(a, y[k]), __fmul_rn(l0, m[k])) / l1;          // (alpha*y + lam_i*mu)/lam_{i+1}
        }
        if (xh_out) reinterpret_cast<f32x4*>(xh_out)[i] = xh;
        if (y_out) reinterpret_cast<f32x4*>(y_out)[i] = y;
        reinterpret_cast<f32x4*>(mu_next)[i] = mn;
    }
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += sm[i];
    __syncthreads();
    return t;
}

// out[r] = w[r]*scale * reduce_D (x[r%B]-x_hat[r])^2     one workgroup per row
__global__ void sqerr_rows_kernel(const float* __restrict__ x, const float* __restrict__ xh, const float* __restrict__ w,
                                  float scale, int mean, int B, int D, float* __restrict__ out) {
    __shared__ float sm[8];
    const int r = blockIdx.x;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)(r % B) * D);
    const f32x4* hr = reinterpret_cast<const f32x4*>(xh + (size_t)r * D);
    float acc = 0.f;
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) {
        f32x4 a = xr[i], b = hr[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = __fsub_rn(a[k], b[k]);
            acc = __fmaf_rn(d, d, acc);
        }
    }
    float tot = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        float red = mean ? tot / (float)D : tot;
        float wr = w ? w[r] : 1.0f;
        // train_loss (bsi.py:310): rpdf * mean ; inf (289): (0.5*rpdf)*sum ; finite (274): ((0.5k)*alpha)*sum
        out[r] = __fmul_rn(__fmul_rn(scale, wr), red);
    }
}

__global__ void sqerr_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ xh,
                                      const float* __restrict__ w, const float* __restrict__ g, float scale, int mean,
                                      int B, int D4, float* __restrict__ gx) {
    const int r = blockIdx.y;
    float coef = -2.0f * g[r] * (w ? w[r] : 1.0f) * scale;
    if (mean) coef /= (float)(D4 * 4);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + (size_t)(r % B) * D4;
    const f32x4* hr = reinterpret_cast<const f32x4*>(xh) + (size_t)r * D4;
    f32x4* o = reinterpret_cast<f32x4*>(gx) + (size_t)r * D4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D4; i += gridDim.x * blockDim.x) {
        f32x4 a = xr[i], b = hr[i], v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = coef * (a[k] - b[k]);
        o[i] = v;
    }
}

// bsi.py:230-247. One workgroup per row.
__global__ void recon_nll_kernel(const float* __restrict__ x, const float* __restrict__ xh, float alpha_R,
                                 const float* __restrict__ bounds, float lo_edge, float dx, int k, int B, int D,
                                 float* __restrict__ out) {
    __shared__ float sm[8];
    const int r = blockIdx.x;
    const float* xr = x + (size_t)(r % B) * D;
    const float* hr = xh + (size_t)r * D;
    const float sigma = rsqrt_rn(alpha_R);
    const float inv_sigma = 1.0f / sigma;           // Normal.cdf: scale.reciprocal()
    const float rsqrt2 = 0.7071067811865476f;
    float acc = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float xv = xr[i], m = hr[i];
        float logp;
        if (k > 0) {
            float q = __fsub_rn(xv, lo_edge) / dx;  // bucketize: trunc toward zero then clamp
            long long idx = (long long)q;
            idx = idx < 0 ? 0 : (idx > k - 1 ? k - 1 : idx);
            const float bl = bounds[idx], br = bounds[idx + 1];
            // 0.5 * (1 + erf((v - loc) * scale.reciprocal() / sqrt(2)))
            float cl = 0.5f * (1.0f + erff(__fmul_rn(__fsub_rn(bl, m), inv_sigma) / 1.4142135623730951f));
            float cr = 0.5f * (1.0f + erff(__fmul_rn(__fsub_rn(br, m), inv_sigma) / 1.4142135623730951f));
            if (idx == 0) cl = 0.0f;
            if (idx == k - 1) cr = 1.0f;
            float pr = fmaxf(__fsub_rn(cr, cl), 1e-20f);
            logp = logf(pr);
        } else {
            // Normal.log_prob: -((x-loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))
            float d = __fsub_rn(xv, m);
            float var = __fmul_rn(sigma, sigma);
            logp = -(d * d) / (2.0f * var) - logf(sigma) - 0.9189385332046727f;
        }
        (void)rsqrt2;
        acc -= logp;
    }
    float tot = block_sum(acc, sm);
    if (threadIdx.x == 0) out[r] = tot;
}

__global__ void to_uint8_kernel(const float* __restrict__ x, float lo, float range, size_t n, uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float u = __fmul_rn(__fsub_rn(x[i], lo) / range, 255.0f);
        u = fminf(fmaxf(u, 0.0f), 255.0f);
        out[i] = (uint8_t)u;  // truncation, as .to(torch.uint8)
    }
}

inline dim3 row_grid(int rows, int D4) {
    int gx = (D4 + TPB - 1) / TPB;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    return dim3(gx, rows);
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int bsi_edm_coeffs(const bsi_params* p, const float* t, int n, float* lam, float* c_skip, float* c_out,
                              float* c_in, bsi_stream_t stream) {
    BSI_CHECK_ARG(p && t && n > 0, "bsi_edm_coeffs: bad args");
    hipLaunchKernelGGL(edm_coeffs_kernel, dim3(blocks_for(n)), dim3(TPB), 0, S(stream), t, n, p->lambda_0, p->delta,
                       p->ln_low, lam, c_skip, c_out, c_in);
    BSI_CHECK_LAUNCH("bsi_edm_coeffs");
    return BSI_OK;
}

extern "C" int bsi_lambda_to_t(const bsi_params* p, const float* lam, int n, float* t, float* rpdf,
                               bsi_stream_t stream) {
    BSI_CHECK_ARG(p && lam && n > 0, "bsi_lambda_to_t: bad args");
    hipLaunchKernelGGL(lambda_to_t_kernel, dim3(blocks_for(n)), dim3(TPB), 0, S(stream), lam, n, p->delta, p->ln_low, t,
                       rpdf);
    BSI_CHECK_LAUNCH("bsi_lambda_to_t");
    return BSI_OK;
}

extern "C" int bsi_schedule(const bsi_params* p, const float* t, int k1, float* lam, float* alpha,
                            bsi_stream_t stream) {
    BSI_CHECK_ARG(p && t && lam && k1 > 0, "bsi_schedule: bad args");
    hipLaunchKernelGGL(schedule_kernel, dim3(blocks_for(k1)), dim3(TPB), 0, S(stream), t, k1, p->delta, p->ln_low, lam,
                       alpha);
    BSI_CHECK_LAUNCH("bsi_schedule");
    return BSI_OK;
}

extern "C" int bsi_lambda_grid(const bsi_params* p, const int64_t* perm, const float* offset, int total, float* lam,
                               bsi_stream_t stream) {
    BSI_CHECK_ARG(p && perm && offset && lam && total > 0, "bsi_lambda_grid: bad args");
    hipLaunchKernelGGL(lambda_grid_kernel, dim3(blocks_for(total)), dim3(TPB), 0, S(stream), perm, offset, total,
                       p->delta, p->ln_low, lam);
    BSI_CHECK_LAUNCH("bsi_lambda_grid");
    return BSI_OK;
}

extern "C" int bsi_q_sample(const bsi_params* p, const float* x, const float* lam, const float* eps, int rows, int B,
                            int D, float* mu, bsi_stream_t stream) {
    BSI_CHECK_ARG(p && x && lam && eps && mu, "bsi_q_sample: null pointer");
    BSI_CHECK_ARG(rows > 0 && B > 0 && D > 0 && D % 4 == 0, "bsi_q_sample: rows=%d B=%d D=%d (D %% 4 must be 0)", rows, B, D);
    hipLaunchKernelGGL(q_sample_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), x, lam, eps, p->lambda_0, B,
                       D / 4, mu);
    BSI_CHECK_LAUNCH("bsi_q_sample");
    return BSI_OK;
}

extern "C" int bsi_scale_rows(const float* mu, const float* c, int c_stride, int rows, int D, float* out,
                              bsi_stream_t stream) {
    BSI_CHECK_ARG(mu && c && out && rows > 0 && D > 0 && D % 4 == 0, "bsi_scale_rows: bad args");
    hipLaunchKernelGGL(scale_rows_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), mu, c, c_stride, D / 4, out);
    BSI_CHECK_LAUNCH("bsi_scale_rows");
    return BSI_OK;
}

extern "C" int bsi_sample_init(const float* eps0, const float* lam, int rows, int D, float* mu, bsi_stream_t stream) {
    BSI_CHECK_ARG(eps0 && lam && mu && rows > 0 && D > 0 && D % 4 == 0, "bsi_sample_init: bad args");
    hipLaunchKernelGGL(sample_init_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), eps0, lam, D / 4, mu);
    BSI_CHECK_LAUNCH("bsi_sample_init");
    return BSI_OK;
}

extern "C" int bsi_predict_combine(const float* mu, const float* f, const float* c_skip, const float* c_out,
                                   int c_stride, int rows, int D, float* x_hat, bsi_stream_t stream) {
    BSI_CHECK_ARG(mu && f && c_skip && c_out && x_hat && rows > 0 && D > 0 && D % 4 == 0, "bsi_predict_combine: bad args");
    hipLaunchKernelGGL(predict_combine_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), mu, f, c_skip, c_out,
                       c_stride, D / 4, x_hat);
    BSI_CHECK_LAUNCH("bsi_predict_combine");
    return BSI_OK;
}

extern "C" int bsi_predict_combine_bwd(const float* g, const float* c_skip, const float* c_out, int c_stride, int rows,
                                       int D, float* g_f, float* g_mu, bsi_stream_t stream) {
    BSI_CHECK_ARG(g && c_skip && c_out && rows > 0 && D > 0 && D % 4 == 0, "bsi_predict_combine_bwd: bad args");
    hipLaunchKernelGGL(predict_combine_bwd_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), g, c_skip, c_out,
                       c_stride, D / 4, g_f, g_mu);
    BSI_CHECK_LAUNCH("bsi_predict_combine_bwd");
    return BSI_OK;
}

extern "C" int bsi_refine_step(const float* mu, const float* f, const float* eps, const float* lam, const float* alpha,
                               const float* c_skip, const float* c_out, int i, int f_is_xhat, int rows, int D,
                               float* x_hat_out, float* y_out, float* mu_next, bsi_stream_t stream) {
    BSI_CHECK_ARG(mu && f && eps && lam && alpha && mu_next, "bsi_refine_step: null pointer");
    BSI_CHECK_ARG(f_is_xhat || (c_skip && c_out), "bsi_refine_step: coefficients missing");
    BSI_CHECK_ARG(rows > 0 && D > 0 && ((size_t)rows * D) % 4 == 0 && i >= 0, "bsi_refine_step: bad sizes");
    const size_t n4 = (size_t)rows * D / 4;
    int grid = blocks_for(n4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(refine_step_kernel<false>, dim3(grid), dim3(TPB), 0, S(stream), mu, f, eps, nullptr, lam, alpha, c_skip,
                       c_out, i, f_is_xhat, n4, x_hat_out, y_out, mu_next);
    BSI_CHECK_LAUNCH("bsi_refine_step");
    return BSI_OK;
}

extern "C" int bsi_refine_step_philox(const float* mu, const float* f, const unsigned long long* seed, const float* lam,
                                      const float* alpha, const float* c_skip, const float* c_out, int i, int f_is_xhat, int rows,
                                      int D, float* x_hat_out, float* y_out, float* mu_next, bsi_stream_t stream) {
    BSI_CHECK_ARG(mu && f && seed && lam && alpha && mu_next, "bsi_refine_step_philox: null pointer");
    BSI_CHECK_ARG(f_is_xhat || (c_skip && c_out), "bsi_refine_step_philox: coefficients missing");
    BSI_CHECK_ARG(rows > 0 && D > 0 && ((size_t)rows * D) % 4 == 0 && i >= 0, "bsi_refine_step_philox: bad sizes");
    const size_t n4 = (size_t)rows * D / 4;
    int grid = blocks_for(n4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(refine_step_kernel<true>, dim3(grid), dim3(TPB), 0, S(stream), mu, f, nullptr, seed, lam, alpha, c_skip,
                       c_out, i, f_is_xhat, n4, x_hat_out, y_out, mu_next);
    BSI_CHECK_LAUNCH("bsi_refine_step_philox");
    return BSI_OK;
}

extern "C" int bsi_philox_normal(const unsigned long long* seed, unsigned stream_id, size_t n, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(seed && out && n > 0 && n % 4 == 0, "bsi_philox_normal: bad args (n must be a multiple of 4)");
    int grid = blocks_for(n / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid), dim3(TPB), 0, S(stream), seed, stream_id, n / 4, out);
    BSI_CHECK_LAUNCH("bsi_philox_normal");
    return BSI_OK;
}

extern "C" int bsi_philox4x32_10(const unsigned* ctr_key, size_t nblocks, unsigned* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(ctr_key && out && nblocks > 0, "bsi_philox4x32_10: bad args");
    int grid = blocks_for(nblocks);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(philox_blocks_kernel, dim3(grid), dim3(TPB), 0, S(stream), ctr_key, nblocks, out);
    BSI_CHECK_LAUNCH("bsi_philox4x32_10");
    return BSI_OK;
}

extern "C" int bsi_philox_uint32(const unsigned long long* seed, unsigned stream_id, size_t n, unsigned* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(seed && out && n > 0 && n % 4 == 0, "bsi_philox_uint32: bad args (n must be a multiple of 4)");
    int grid = blocks_for(n / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(philox_uint32_kernel, dim3(grid), dim3(TPB), 0, S(stream), seed, stream_id, n / 4, out);
    BSI_CHECK_LAUNCH("bsi_philox_uint32");
    return BSI_OK;
}

extern "C" int bsi_sqerr_rows(const float* x, const float* x_hat, const float* w, float scale, int mean, int rows,
                              int B, int D, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && x_hat && out && rows > 0 && B > 0 && D > 0 && D % 4 == 0, "bsi_sqerr_rows: bad args");
    hipLaunchKernelGGL(sqerr_rows_kernel, dim3(rows), dim3(TPB), 0, S(stream), x, x_hat, w, scale, mean, B, D, out);
    BSI_CHECK_LAUNCH("bsi_sqerr_rows");
    return BSI_OK;
}

extern "C" int bsi_sqerr_rows_bwd(const float* x, const float* x_hat, const float* w, const float* g, float scale,
                                  int mean, int rows, int B, int D, float* g_xhat, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && x_hat && g && g_xhat && rows > 0 && B > 0 && D > 0 && D % 4 == 0, "bsi_sqerr_rows_bwd: bad args");
    hipLaunchKernelGGL(sqerr_rows_bwd_kernel, row_grid(rows, D / 4), dim3(TPB), 0, S(stream), x, x_hat, w, g, scale,
                       mean, B, D / 4, g_xhat);
    BSI_CHECK_LAUNCH("bsi_sqerr_rows_bwd");
    return BSI_OK;
}

extern "C" int bsi_recon_nll(const float* x, const float* x_hat, float alpha_R, const float* bounds, float lo_edge,
                             float dx, int k, int rows, int B, int D, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && x_hat && out && rows > 0 && B > 0 && D > 0, "bsi_recon_nll: bad args");
    BSI_CHECK_ARG(k == 0 || bounds, "bsi_recon_nll: discretised likelihood needs bin boundaries");
    hipLaunchKernelGGL(recon_nll_kernel, dim3(rows), dim3(TPB), 0, S(stream), x, x_hat, alpha_R, bounds, lo_edge, dx, k,
                       B, D, out);
    BSI_CHECK_LAUNCH("bsi_recon_nll");
    return BSI_OK;
}

extern "C" int bsi_to_uint8(const float* x, float lo, float hi, size_t n, uint8_t* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && out && n > 0 && hi > lo, "bsi_to_uint8: bad args");
    int grid = blocks_for(n);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(grid), dim3(TPB), 0, S(stream), x, lo, hi - lo, n, out);
    BSI_CHECK_LAUNCH("bsi_to_uint8");
    return BSI_OK;
}
