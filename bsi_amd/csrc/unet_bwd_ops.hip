// Memory-bound kernels of the VDM-UNet TRAINING path (autograd of bsi/nn/residual_block.py:21-24,40-48,61-64 and
// bsi/models/vdm_unet.py:72,100 of the reference): FiLM + SiLU + Dropout forward/backward, GroupNorm(+SiLU) backward,
// backward of the fp32 1x1 decode convolution.
#include <math.h>

#include "common.h"
#include "dit_ops.h"
#include "unet_ops.h"

int bsi_reduce_slabs_launch(const float* slabs, size_t slab_stride, int splits, size_t n, int accumulate, float* out, hipStream_t s);
int bsi_reduce_slabs2_launch(const float* slabsA, size_t strideA, size_t nA, float* outA, const float* slabsB, size_t strideB, size_t nB,
                             float* outB, int splits, int accumulate, hipStream_t s);

namespace {

__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    return s * (1.0f + z * (1.0f - s));
}

__device__ __forceinline__ void unpack8(const u32x4 w, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}

// y = Dropout(SiLU(h1 * (scale + 1) + shift))  (FeatureModulation: addcmul(shift, scale + 1, y), residual_block.py:21-24,
// then ActFn and nn.Dropout, :44-46).  h1, y bf16 [M, N]; film row of pixel m = (m / HW) % film_rows.
__global__ void film_silu_drop_kernel(const __bf16* __restrict__ h1, unsigned M, int N, int HW, const float* __restrict__ film,
                                      int film_rows, int film_stride, DropCfg dc, __bf16* __restrict__ y) {
    const unsigned n8 = N / 8;
    const unsigned total = M * n8;  // launcher guarantees M * N < 2^32
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned m = i / n8;
        const unsigned c = (i - m * n8) * 8;
        const float* fr = film + (size_t)((m / (unsigned)HW) % (unsigned)film_rows) * film_stride;
        float v[8], sc[8], sh[8];
        const unsigned rh = drop_row(dc, m);
        unpack8(*reinterpret_cast<const u32x4*>(h1 + (size_t)i * 8), v);
        *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(fr + c);
        *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(fr + c + 4);
        *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(fr + N + c);
        *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(fr + N + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float u = silu_f(__fmaf_rn(sc[e] + 1.0f, v[e], sh[e]));
            if (dc.thr) u = drop_keep_rc(dc, rh, c + e) ? u * dc.scale : 0.0f;  // element (row = pixel, column = channel)
            v[e] = u;
        }
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4*>(y + (size_t)i * 8) = w;
    }
}

// Backward of the above: dU = dy * mask/(1-p) * silu'(u), u = h1*(scale+1)+shift;
//   dh1 = dU * (scale + 1)  (bf16),  dscale[b, n] += sum_p dU * h1,  dshift[b, n] += sum_p dU.
// One workgroup (256 threads) per slab of FB_ROWS pixels of one image; thread = (8-channel chunk, pixel sub-row).
constexpr int FB_ROWS = 64;
__global__ __launch_bounds__(256) void film_silu_bwd_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ h1, int N,
                                                            int HW, const float* __restrict__ film, int film_rows,
                                                            int film_stride, DropCfg dc, __bf16* __restrict__ dh1,
                                                            float* __restrict__ dfilm, int dfilm_stride, size_t part_stride) {
    // part_stride > 0: the slab STORES its sums into plane (slab index inside the image) of dfilm, planes part_stride floats apart,
    // summed later in fixed order (bsi_sum_cast_rows_bf16): reproducible, no zero fill; 0: fp32 atomics (the C-ABI contract)
    __shared__ float red[2][256 * 8];
    const int n8 = N / 8;            // chunks per pixel (8, 16)
    const int rows_par = 256 / n8;   // pixel rows handled in parallel
    const int ch = threadIdx.x % n8, sub = threadIdx.x / n8;
    const int c = ch * 8;
    const size_t m0 = (size_t)blockIdx.x * FB_ROWS;
    const int b = (int)(m0 / HW);
    const float* fr = film + (size_t)(b % film_rows) * film_stride;
    float sc1[8], sh[8], gs[8], gh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc1[e] = fr[c + e] + 1.0f; sh[e] = fr[N + c + e]; gs[e] = 0.f; gh[e] = 0.f; }
    for (int r = sub; r < FB_ROWS; r += rows_par) {
        const size_t m = m0 + r;
        float g[8], h[8];
        const unsigned rh = drop_row(dc, (unsigned)m);
        unpack8(*reinterpret_cast<const u32x4*>(dy + m * N + c), g);
        unpack8(*reinterpret_cast<const u32x4*>(h1 + m * N + c), h);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float du = g[e] * silu_grad_f(__fmaf_rn(sc1[e], h[e], sh[e]));
            if (dc.thr) du = drop_keep_rc(dc, rh, c + e) ? du * dc.scale : 0.0f;
            gs[e] = __fmaf_rn(du, h[e], gs[e]);
            gh[e] += du;
            g[e] = du * sc1[e];
        }
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(g[2 * e], g[2 * e + 1]);
        *reinterpret_cast<u32x4*>(dh1 + m * N + c) = w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = gs[e]; red[1][threadIdx.x * 8 + e] = gh[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * N; i += 256) {
        const int which = i / N, n = i % N;
        float a = 0.f;
        for (int s = 0; s < rows_par; ++s) a += red[which][(s * n8 + n / 8) * 8 + (n & 7)];
        float* dst = dfilm + (size_t)b * dfilm_stride + which * N + n;
        if (part_stride) dst[(size_t)((m0 % HW) / FB_ROWS) * part_stride] = a;
        else atomicAdd(dst, a);
    }
}

// Backward of GroupNorm(32, affine)(+ SiLU) over cat(x1, x2) of one image (see groupnorm_kernel):
//   z = n*gamma + beta, n = (x - mean)*rstd;  dz = da * silu'(z) (or da);  dgamma += sum dz*n, dbeta += sum dz,
//   dn = dz*gamma;  dx = rstd * (dn - mean_g(dn) - n * mean_g(dn * n));   out = dx (+ add) (+ add_b on the x1 part).
// Grid (image, 32-channel slice) like the forward kernel; three passes (statistics; group sums and affine gradients; write).
constexpr int GN_CS = 32, GN_TPB = 256;
__global__ __launch_bounds__(GN_TPB) void groupnorm_bwd_kernel(const __bf16* __restrict__ da, const float* __restrict__ x1, int C1,
                                                             const float* __restrict__ x2, int C2, int HW,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int silu, const float* __restrict__ add,
                                                             const float* __restrict__ add_b, float* __restrict__ out1,
                                                             float* __restrict__ out2, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, __bf16* __restrict__ out1_bf, const float* __restrict__ stats,
                                                             float* __restrict__ partials) {
    __shared__ float red_s[512], red_q[512];
    __shared__ float red_g[1024], red_b[1024];  // [pixel row][channel of the slice]
    __shared__ float mean_s[16], rstd_s[16], m1_s[16], m2_s[16];
    const int C = C1 + C2, cpg = C / 32;
    constexpr int CH4 = GN_CS / 4;
    const int b = blockIdx.x, t = threadIdx.x, cs0 = blockIdx.y * GN_CS;
    const int ch = t % CH4, prow = t / CH4, PPI = GN_TPB / CH4;
    const int c0 = cs0 + ch * 4, cl = ch * 4;
    const bool second = c0 >= C1;
    const float* src = second ? x2 + (size_t)b * HW * C2 + (c0 - C1) : x1 + (size_t)b * HW * C1 + c0;
    const int sstride = second ? C2 : C1;
    const int NS = CH4 * 2;
    // ---- pass 0: statistics (taken from the forward pass when it saved them: one pass over x less)
    if (stats) {
        if (t < GN_CS / cpg) {
            const float* st = stats + ((size_t)b * 32 + cs0 / cpg + t) * 2;
            mean_s[t] = st[0];
            rstd_s[t] = st[1];
        }
        __syncthreads();
    } else {
        {
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
            for (int p = prow; p < HW; p += PPI) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * sstride);
                s0 += v[0] + v[1];
                q0 += v[0] * v[0] + v[1] * v[1];
                s1 += v[2] + v[3];
                q1 += v[2] * v[2] + v[3] * v[3];
            }
            red_s[prow * NS + 2 * ch] = s0; red_q[prow * NS + 2 * ch] = q0;
            red_s[prow * NS + 2 * ch + 1] = s1; red_q[prow * NS + 2 * ch + 1] = q1;
        }
        __syncthreads();
        if (t < GN_CS / cpg) {
            const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
            float ts = 0.f, tq = 0.f;
            for (int r = 0; r < PPI; ++r)
                for (int k = k0; k < k1; ++k) { ts += red_s[r * NS + k]; tq += red_q[r * NS + k]; }
            const float n = (float)HW * cpg;
            const float mean = ts / n;
            mean_s[t] = mean;
            rstd_s[t] = 1.0f / sqrtf(fmaxf(tq / n - mean * mean, 0.f) + eps);
        }
        __syncthreads();
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = mean_s[(cl + k) / cpg]; rstd[k] = rstd_s[(cl + k) / cpg]; }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    const __bf16* dap = da + (size_t)b * HW * C + c0;
    auto dz_of = [&](const f32x4 v, const u32x2 gw, float* nrm, float* dz) {
        const float g[4] = {__uint_as_float(gw[0] << 16), __uint_as_float(gw[0] & 0xffff0000u), __uint_as_float(gw[1] << 16),
                            __uint_as_float(gw[1] & 0xffff0000u)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nrm[k] = (v[k] - mean[k]) * rstd[k];
            dz[k] = silu ? g[k] * silu_grad_f(__fmaf_rn(nrm[k], ga[k], be[k])) : g[k];
        }
    };
    // ---- pass 1: group sums of dn and dn*n, per-channel dgamma / dbeta
    {
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f, gg[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
        // 4 pixel rows per trip, all 8 loads in flight before the (exp + rcp) arithmetic of the first: the loop is latency bound
        for (int p0 = prow; p0 < HW; p0 += 4 * PPI) {
            f32x4 vv[4];
            u32x2 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * PPI;
                vv[u] = p < HW ? *reinterpret_cast<const f32x4*>(src + (size_t)p * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
                gv[u] = p < HW ? *reinterpret_cast<const u32x2*>(dap + (size_t)p * C) : u32x2{0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u * PPI >= HW) continue;
                float nrm[4], dz[4];
                dz_of(vv[u], gv[u], nrm, dz);
#pragma unroll
                for (int k = 0; k < 4; ++k) { gg[k] = __fmaf_rn(dz[k], nrm[k], gg[k]); gb[k] += dz[k]; }
                const float d0 = dz[0] * ga[0], d1 = dz[1] * ga[1], d2 = dz[2] * ga[2], d3 = dz[3] * ga[3];
                s0 += d0 + d1; q0 += d0 * nrm[0] + d1 * nrm[1];
                s1 += d2 + d3; q1 += d2 * nrm[2] + d3 * nrm[3];
            }
        }
        red_s[prow * NS + 2 * ch] = s0; red_q[prow * NS + 2 * ch] = q0;
        red_s[prow * NS + 2 * ch + 1] = s1; red_q[prow * NS + 2 * ch + 1] = q1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { red_g[prow * GN_CS + cl + k] = gg[k]; red_b[prow * GN_CS + cl + k] = gb[k]; }
    }
    __syncthreads();
    if (t < GN_CS / cpg) {
        const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < PPI; ++r)
            for (int k = k0; k < k1; ++k) { ts += red_s[r * NS + k]; tq += red_q[r * NS + k]; }
        const float n = (float)HW * cpg;
        m1_s[t] = ts / n;
        m2_s[t] = tq / n;
    }
    if (t >= 64 && t < 64 + GN_CS) {
        const int c = t - 64;
        float a = 0.f, bb = 0.f;
        for (int r = 0; r < PPI; ++r) { a += red_g[r * GN_CS + c]; bb += red_b[r * GN_CS + c]; }
        if (partials) {  // [B][2][C] per-image sums, added over the images in fixed order by the launcher: reproducible
            partials[((size_t)b * 2 + 0) * C + cs0 + c] = a;
            partials[((size_t)b * 2 + 1) * C + cs0 + c] = bb;
        } else {
            atomicAdd(dgamma + cs0 + c, a);
            atomicAdd(dbeta + cs0 + c, bb);
        }
    }
    __syncthreads();
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { m1[k] = m1_s[(cl + k) / cpg]; m2[k] = m2_s[(cl + k) / cpg]; }
    // ---- pass 2: dx
    float* dst = second ? out2 + (size_t)b * HW * C2 + (c0 - C1) : out1 + (size_t)b * HW * C1 + c0;
    const float* ab = (!second && add_b) ? add_b + (size_t)b * HW * C1 + c0 : nullptr;
    const float* aa = add ? add + (size_t)b * HW * C + c0 : nullptr;
    for (int p0 = prow; p0 < HW; p0 += 4 * PPI) {
        f32x4 vv[4], a4[4], b4[4];
        u32x2 gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * PPI;
            const bool ok = p < HW;
            vv[u] = ok ? *reinterpret_cast<const f32x4*>(src + (size_t)p * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
            gv[u] = ok ? *reinterpret_cast<const u32x2*>(dap + (size_t)p * C) : u32x2{0u, 0u};
            a4[u] = (ok && aa) ? *reinterpret_cast<const f32x4*>(aa + (size_t)p * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            b4[u] = (ok && ab) ? *reinterpret_cast<const f32x4*>(ab + (size_t)p * C1) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * PPI;
            if (p >= HW) continue;
            float nrm[4], dz[4];
            dz_of(vv[u], gv[u], nrm, dz);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rstd[k] * (dz[k] * ga[k] - m1[k] - nrm[k] * m2[k]) + a4[u][k] + b4[u][k];
            *reinterpret_cast<f32x4*>(dst + (size_t)p * sstride) = o;
            if (out1_bf && !second) {  // bf16 copy of the x1 gradient: the next block's convolutions take it as their dY operand
                u32x2 w2;
                w2[0] = pack_bf16x2(o[0], o[1]);
                w2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(out1_bf + ((size_t)b * HW + p) * C1 + c0) = w2;
            }
        }
    }
}

// The same with the workgroup's slice RESIDENT in registers (round 5): HW = ROWS x 64 pixel rows (the UNet's 32 x 32 maps: ROWS = 16).
// Deterministic like the streaming kernel (the partial sums are cut by 64 instead of 32 row classes: other last bits).
constexpr int GNR_TPB = 512;  // 64 pixel rows in parallel: 16 resident rows per thread at 32 x 32 pixels (with 256 threads and 32 rows hipcc
                               // schedules itself to 490 registers, or spills at 256)
template <int ROWS, int CT, int CSRC>  // channels of cat(x1, x2) (= row pitch of dA / add), channels of x1 and of x2 (= their row pitch)
__global__ __launch_bounds__(GNR_TPB) void groupnorm_bwd_res_kernel(const __bf16* __restrict__ da, const float* __restrict__ x1, int C1,
                                                             const float* __restrict__ x2, int C2, int HW,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int silu, const float* __restrict__ add,
                                                             const float* __restrict__ add_b, float* __restrict__ out1,
                                                             float* __restrict__ out2, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, __bf16* __restrict__ out1_bf, const float* __restrict__ stats,
                                                             float* __restrict__ partials) {
    __shared__ float red_s[GNR_TPB * 2], red_q[GNR_TPB * 2];
    __shared__ float red_g[GNR_TPB * 4], red_b[GNR_TPB * 4];  // [pixel row][channel of the slice]
    __shared__ float mean_s[16], rstd_s[16], m1_s[16], m2_s[16];
    constexpr int C = CT, cpg = C / 32;  // (C1 = CSRC, C2 = CT - CSRC: compile-time row pitches keep the 2 x ROWS row addresses out of registers)
    constexpr int CH4 = GN_CS / 4;
    const int b = blockIdx.x, t = threadIdx.x, cs0 = blockIdx.y * GN_CS;
    const int ch = t % CH4, prow = t / CH4, PPI = GNR_TPB / CH4;
    const int c0 = cs0 + ch * 4, cl = ch * 4;
    // wave-uniform bases + ONE 32-bit lane offset per stream (a 32-channel slice never straddles x1 | x2): per-lane 64-bit row addresses
    // (16 rows x 6 streams) were what pushed the allocation past 256 registers
    const bool second = cs0 >= C1;
    const float* src = second ? x2 + (size_t)b * HW * C2 + (cs0 - C1) : x1 + (size_t)b * HW * C1 + cs0;
    const unsigned lo_s = (unsigned)(prow * CSRC + cl), lo_c = (unsigned)(prow * CT + cl);  // element offsets at row pitch CSRC / CT
    constexpr int sstride = CSRC;
    const int NS = CH4 * 2;
    const __bf16* dap = da + (size_t)b * HW * C + cs0;
    // ---- the slice stays in registers between the passes: ROWS pixel rows of 4 channels per thread (x fp32 + dA bf16 = 6 registers a
    //      row), every load issued before the first use.  The streaming kernel read x and dA twice, four rows at a time.
    f32x4 vv[ROWS];
    u32x2 gv[ROWS];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        vv[u] = *reinterpret_cast<const f32x4*>(src + u * PPI * sstride + lo_s);
        gv[u] = *reinterpret_cast<const u32x2*>(dap + u * PPI * C + lo_c);
    }
    // ---- pass 0: statistics (taken from the forward pass when it saved them: one pass over x less)
    if (stats) {
        if (t < GN_CS / cpg) {
            const float* st = stats + ((size_t)b * 32 + cs0 / cpg + t) * 2;
            mean_s[t] = st[0];
            rstd_s[t] = st[1];
        }
        __syncthreads();
    } else {
        {
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const f32x4 v = vv[u];
                s0 += v[0] + v[1];
                q0 += v[0] * v[0] + v[1] * v[1];
                s1 += v[2] + v[3];
                q1 += v[2] * v[2] + v[3] * v[3];
            }
            red_s[prow * NS + 2 * ch] = s0; red_q[prow * NS + 2 * ch] = q0;
            red_s[prow * NS + 2 * ch + 1] = s1; red_q[prow * NS + 2 * ch + 1] = q1;
        }
        __syncthreads();
        if (t < GN_CS / cpg) {
            const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
            float ts = 0.f, tq = 0.f;
            for (int r = 0; r < PPI; ++r)
                for (int k = k0; k < k1; ++k) { ts += red_s[r * NS + k]; tq += red_q[r * NS + k]; }
            const float n = (float)HW * cpg;
            const float mean = ts / n;
            mean_s[t] = mean;
            rstd_s[t] = 1.0f / sqrtf(fmaxf(tq / n - mean * mean, 0.f) + eps);
        }
        __syncthreads();
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = mean_s[(cl + k) / cpg]; rstd[k] = rstd_s[(cl + k) / cpg]; }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    auto dz_of = [&](const f32x4 v, const u32x2 gw, float* nrm, float* dz) {
        const float g[4] = {__uint_as_float(gw[0] << 16), __uint_as_float(gw[0] & 0xffff0000u), __uint_as_float(gw[1] << 16),
                            __uint_as_float(gw[1] & 0xffff0000u)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nrm[k] = (v[k] - mean[k]) * rstd[k];
            dz[k] = silu ? g[k] * silu_grad_f(__fmaf_rn(nrm[k], ga[k], be[k])) : g[k];
        }
    };
    // ---- pass 1: group sums of dn and dn*n, per-channel dgamma / dbeta
    {
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f, gg[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            float nrm[4], dz[4];
            dz_of(vv[u], gv[u], nrm, dz);
#pragma unroll
            for (int k = 0; k < 4; ++k) { gg[k] = __fmaf_rn(dz[k], nrm[k], gg[k]); gb[k] += dz[k]; }
            const float d0 = dz[0] * ga[0], d1 = dz[1] * ga[1], d2 = dz[2] * ga[2], d3 = dz[3] * ga[3];
            s0 += d0 + d1; q0 += d0 * nrm[0] + d1 * nrm[1];
            s1 += d2 + d3; q1 += d2 * nrm[2] + d3 * nrm[3];
            if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two rows' arithmetic at a time: interleaving all of them costs > 256 registers
        }
        red_s[prow * NS + 2 * ch] = s0; red_q[prow * NS + 2 * ch] = q0;
        red_s[prow * NS + 2 * ch + 1] = s1; red_q[prow * NS + 2 * ch + 1] = q1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { red_g[prow * GN_CS + cl + k] = gg[k]; red_b[prow * GN_CS + cl + k] = gb[k]; }
    }
    __syncthreads();
    if (t < GN_CS / cpg) {
        const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < PPI; ++r)
            for (int k = k0; k < k1; ++k) { ts += red_s[r * NS + k]; tq += red_q[r * NS + k]; }
        const float n = (float)HW * cpg;
        m1_s[t] = ts / n;
        m2_s[t] = tq / n;
    }
    if (t >= 64 && t < 64 + GN_CS) {
        const int c = t - 64;
        float a = 0.f, bb = 0.f;
        for (int r = 0; r < PPI; ++r) { a += red_g[r * GN_CS + c]; bb += red_b[r * GN_CS + c]; }
        if (partials) {  // [B][2][C] per-image sums, added over the images in fixed order by the launcher: reproducible
            partials[((size_t)b * 2 + 0) * C + cs0 + c] = a;
            partials[((size_t)b * 2 + 1) * C + cs0 + c] = bb;
        } else {
            atomicAdd(dgamma + cs0 + c, a);
            atomicAdd(dbeta + cs0 + c, bb);
        }
    }
    __syncthreads();
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { m1[k] = m1_s[(cl + k) / cpg]; m2[k] = m2_s[(cl + k) / cpg]; }
    // ---- pass 2: dx
    float* dst = second ? out2 + (size_t)b * HW * C2 + (cs0 - C1) : out1 + (size_t)b * HW * C1 + cs0;
    const float* ab = (!second && add_b) ? add_b + (size_t)b * HW * C1 + cs0 : nullptr;
    const float* aa = add ? add + (size_t)b * HW * C + cs0 : nullptr;
    __bf16* obf = (out1_bf && !second) ? out1_bf + (size_t)b * HW * CSRC + cs0 : nullptr;
#pragma unroll
    for (int u0 = 0; u0 < ROWS; u0 += 4) {
        f32x4 a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a4[u] = aa ? *reinterpret_cast<const f32x4*>(aa + (u0 + u) * PPI * C + lo_c) : f32x4{0.f, 0.f, 0.f, 0.f};
            b4[u] = ab ? *reinterpret_cast<const f32x4*>(ab + (u0 + u) * PPI * CSRC + lo_s) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float nrm[4], dz[4];
            // opaque to the optimiser: it would otherwise keep pass 1's n and dz of all ROWS rows alive across the barrier (256 registers)
            asm volatile("" : "+v"(vv[u0 + u]), "+v"(gv[u0 + u]));
            dz_of(vv[u0 + u], gv[u0 + u], nrm, dz);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rstd[k] * (dz[k] * ga[k] - m1[k] - nrm[k] * m2[k]) + a4[u][k] + b4[u][k];
            *reinterpret_cast<f32x4*>(dst + (u0 + u) * PPI * sstride + lo_s) = o;
            if (obf) {  // bf16 copy of the x1 gradient: the next block's convolutions take it as their dY operand
                u32x2 w2;
                w2[0] = pack_bf16x2(o[0], o[1]);
                w2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(obf + (u0 + u) * PPI * CSRC + lo_s) = w2;
            }
            if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}


// Backward of unet_decode_kernel: dY[m, o] = c_out[b] * g_xhat[b, o, pix];  dh[m, :] = sum_o dY[m, o] * w[o, :];
// dw[o, :] += sum_m dY[m, o] * h[m, :];  db[o] += sum_m dY[m, o].  One workgroup per slab of 256 pixels.
constexpr int DB_PIX = 256;
__global__ __launch_bounds__(256) void unet_decode_bwd_kernel(const float* __restrict__ g_xhat, const float* __restrict__ c_out,
                                                              int coef_stride, const float* __restrict__ h, int M, int C,
                                                              int HW, const float* __restrict__ w, int Cout,
                                                              float* __restrict__ dh, float* __restrict__ dw,
                                                              float* __restrict__ db, float* __restrict__ parts) {
    // parts != null: the block STORES its sums as slab blockIdx.x = [Cout * C weight sums][4 bias sums]; the launcher adds the slabs
    // in index order (reproducible); null: fp32 atomics into dw / db (the C-ABI contract)
    __shared__ float dys[DB_PIX][4];
    __shared__ float red[256][4];
    __shared__ float dbw[4][4];
    const int m0 = blockIdx.x * DB_PIX;
    {
        const int m = m0 + threadIdx.x;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
            const int b = m / HW, pix = m % HW;
            const float co = c_out ? c_out[(size_t)b * coef_stride] : 1.0f;
            for (int o = 0; o < Cout; ++o) v[o] = co * g_xhat[((size_t)b * Cout + o) * HW + pix];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            dys[threadIdx.x][o] = v[o];
            const float s = wave_sum(v[o]);
            if ((threadIdx.x & 63) == 0) {
                if (parts) dbw[threadIdx.x >> 6][o] = s;
                else if (o < Cout) atomicAdd(db + o, s);
            }
        }
    }
    __syncthreads();
    float* slab = parts ? parts + (size_t)blockIdx.x * ((size_t)Cout * C + 4) : nullptr;
    if (parts && threadIdx.x < 4) slab[(size_t)Cout * C + threadIdx.x] = (dbw[0][threadIdx.x] + dbw[1][threadIdx.x]) + (dbw[2][threadIdx.x] + dbw[3][threadIdx.x]);
    const int c = threadIdx.x % C, sub = threadIdx.x / C, S = 256 / C;
    float wv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o) wv[o] = o < Cout ? w[(size_t)o * C + c] : 0.f;
    for (int p = sub; p < DB_PIX && m0 + p < M; p += S) {
        const float hv = h[(size_t)(m0 + p) * C + c];
        float d = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o] = __fmaf_rn(dys[p][o], hv, acc[o]);
            d = __fmaf_rn(dys[p][o], wv[o], d);
        }
        dh[(size_t)(m0 + p) * C + c] = d;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) red[threadIdx.x][o] = acc[o];
    __syncthreads();
    if (sub == 0) {
        for (int o = 0; o < Cout; ++o) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += red[s * C + c][o];
            if (parts) slab[(size_t)o * C + c] = a;
            else atomicAdd(dw + (size_t)o * C + c, a);
        }
    }
}

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)

int bsi_film_silu_drop(const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride, DropCfg dc,
                       void* y, bsi_stream_t stream) {
    BSI_CHECK_ARG(h1 && film && y && M > 0 && N > 0 && N % 8 == 0 && HW > 0 && film_rows > 0 && film_stride % 4 == 0 &&
                      (size_t)M * N < (1ull << 32),
                  "bsi_film_silu: bad args");
    size_t g = ((size_t)M * (N / 8) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(film_silu_drop_kernel, dim3((int)g), dim3(256), 0, S_(stream), reinterpret_cast<const __bf16*>(h1), (unsigned)M,
                       N, HW, film, film_rows, film_stride, dc, reinterpret_cast<__bf16*>(y));
    BSI_CHECK_LAUNCH("bsi_film_silu");
    return BSI_OK;
}

int bsi_film_silu_bwd_drop(const void* dy, const void* h1, int M, int N, int HW, const float* film, int film_rows,
                           int film_stride, DropCfg dc, void* dh1, float* dfilm, int dfilm_stride, bsi_stream_t stream,
                           size_t part_stride) {
    BSI_CHECK_ARG(dy && h1 && film && dh1 && dfilm && M > 0, "bsi_film_silu_bwd: bad args");
    BSI_CHECK_ARG((N == 64 || N == 128) && HW % FB_ROWS == 0 && M % HW == 0 && film_rows > 0,
                  "bsi_film_silu_bwd: N=%d (64 or 128), HW=%d (multiple of %d)", N, HW, FB_ROWS);
    hipLaunchKernelGGL(film_silu_bwd_kernel, dim3(M / FB_ROWS), dim3(256), 0, S_(stream), reinterpret_cast<const __bf16*>(dy),
                       reinterpret_cast<const __bf16*>(h1), N, HW, film, film_rows, film_stride, dc,
                       reinterpret_cast<__bf16*>(dh1), dfilm, dfilm_stride, part_stride);
    BSI_CHECK_LAUNCH("bsi_film_silu_bwd");
    return BSI_OK;
}

extern "C" int bsi_film_silu(const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride,
                             float dropout_p, unsigned long long seed, unsigned site, void* y, bsi_stream_t stream) {
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_film_silu: dropout probability %g outside [0, 1)", (double)dropout_p);
    return bsi_film_silu_drop(h1, M, N, HW, film, film_rows, film_stride, make_drop(dropout_p, seed, site), y, stream);
}

extern "C" int bsi_film_silu_bwd(const void* dy, const void* h1, int M, int N, int HW, const float* film, int film_rows,
                                 int film_stride, float dropout_p, unsigned long long seed, unsigned site, void* dh1,
                                 float* dfilm, int dfilm_stride, bsi_stream_t stream) {
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_film_silu_bwd: dropout probability %g outside [0, 1)", (double)dropout_p);
    return bsi_film_silu_bwd_drop(dy, h1, M, N, HW, film, film_rows, film_stride, make_drop(dropout_p, seed, site), dh1, dfilm,
                                  dfilm_stride, stream, 0);
}

static int groupnorm_bwd_impl(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                              const float* beta, float eps, int silu, const float* add, const float* add_b, float* out1, float* out2,
                              float* dgamma, float* dbeta, void* out1_bf16, const float* stats, bsi_stream_t stream,
                              float* partials = nullptr) {
    BSI_CHECK_ARG(da && x1 && gamma && beta && out1 && dgamma && dbeta && B > 0 && HW > 0, "bsi_groupnorm_bwd_nhwc: bad args");
    const int C = C1 + C2;
    BSI_CHECK_ARG((C == 128 || C == 256 || C == 64) && C1 % 32 == 0 && C2 % 32 == 0 && (C2 == 0 || (x2 && out2)),
                  "bsi_groupnorm_bwd_nhwc: C1+C2=%d unsupported (64, 128 or 256 channels, 32 groups)", C);
    // per launch at 128 images (tools/experiments/gn_bwd_ab.sh): 128 channels 76.6 against 80.6 us streaming, cat(x, skip) with 256 channels 125.1
    // against 150.8.  BSI_GN_BWD_STREAM=1: the streaming kernel everywhere (the A/B partner; other last bits).
    static const bool streaming = getenv("BSI_GN_BWD_STREAM") != nullptr;
    if (HW == 1024 && C1 == 128 && C2 == 0 && !streaming)
        hipLaunchKernelGGL((groupnorm_bwd_res_kernel<16, 128, 128>), dim3(B, C / GN_CS), dim3(GNR_TPB), 0, S_(stream), reinterpret_cast<const __bf16*>(da),
                           x1, C1, x2, C2, HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, reinterpret_cast<__bf16*>(out1_bf16),
                           stats, partials);
    else if (HW == 1024 && C1 == 128 && C2 == 128 && !streaming)
        hipLaunchKernelGGL((groupnorm_bwd_res_kernel<16, 256, 128>), dim3(B, C / GN_CS), dim3(GNR_TPB), 0, S_(stream), reinterpret_cast<const __bf16*>(da),
                           x1, C1, x2, C2, HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, reinterpret_cast<__bf16*>(out1_bf16),
                           stats, partials);
    else
        hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(B, C / GN_CS), dim3(GN_TPB), 0, S_(stream), reinterpret_cast<const __bf16*>(da), x1, C1, x2, C2,
                           HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, reinterpret_cast<__bf16*>(out1_bf16), stats, partials);
    BSI_CHECK_LAUNCH("bsi_groupnorm_bwd_nhwc");
    if (partials) {  // dgamma / dbeta are WRITTEN: per-image rows summed in image order
        return bsi_reduce_slabs2_launch(partials, (size_t)2 * C, (size_t)C, dgamma, partials + C, (size_t)2 * C, (size_t)C, dbeta, B, 0, S_(stream));
    }
    return BSI_OK;
}

// engine-internal (unet_ops.h): the reproducible form of bsi_groupnorm_bwd_cast_nhwc -- `partials`: B * 2 * (C1 + C2) floats of scratch
int bsi_groupnorm_bwd_cast_det(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                               const float* beta, float eps, int silu, const float* add, const float* add_b, float* out1, float* out2,
                               float* dgamma, float* dbeta, void* out1_bf16, const float* stats, float* partials, bsi_stream_t stream) {
    BSI_CHECK_ARG(out1_bf16 && partials, "bsi_groupnorm_bwd_cast_det: bf16 output or scratch missing");
    return groupnorm_bwd_impl(da, x1, C1, x2, C2, B, HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, out1_bf16, stats, stream,
                              partials);
}

extern "C" int bsi_groupnorm_bwd_nhwc(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW,
                                      const float* gamma, const float* beta, float eps, int silu, const float* add,
                                      const float* add_b, float* out1, float* out2, float* dgamma, float* dbeta,
                                      bsi_stream_t stream) {
    return groupnorm_bwd_impl(da, x1, C1, x2, C2, B, HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, nullptr, nullptr, stream);
}

extern "C" int bsi_groupnorm_bwd_cast_nhwc(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW,
                                           const float* gamma, const float* beta, float eps, int silu, const float* add,
                                           const float* add_b, float* out1, float* out2, float* dgamma, float* dbeta,
                                           void* out1_bf16, const float* stats, bsi_stream_t stream) {
    BSI_CHECK_ARG(out1_bf16, "bsi_groupnorm_bwd_cast_nhwc: bf16 output missing");
    return groupnorm_bwd_impl(da, x1, C1, x2, C2, B, HW, gamma, beta, eps, silu, add, add_b, out1, out2, dgamma, dbeta, out1_bf16, stats, stream);
}

extern "C" int bsi_unet_decode_bwd(const float* g_xhat, const float* c_out, int coef_stride, const float* h, int B, int HW, int C,
                                   const float* w, int Cout, float* dh, float* dw, float* db, bsi_stream_t stream) {
    BSI_CHECK_ARG(g_xhat && h && w && dh && dw && db && B > 0 && HW > 0, "bsi_unet_decode_bwd: bad args");
    BSI_CHECK_ARG(Cout >= 1 && Cout <= 4 && C >= 1 && C <= 256 && 256 % C == 0, "bsi_unet_decode_bwd: Cout=%d (<= 4), C=%d (divides 256)",
                  Cout, C);
    const int M = B * HW;
    hipLaunchKernelGGL(unet_decode_bwd_kernel, dim3((M + DB_PIX - 1) / DB_PIX), dim3(256), 0, S_(stream), g_xhat, c_out, coef_stride,
                       h, M, C, HW, w, Cout, dh, dw, db, static_cast<float*>(nullptr));
    BSI_CHECK_LAUNCH("bsi_unet_decode_bwd");
    return BSI_OK;
}

// engine-internal (unet_ops.h): reproducible form; dw / db are WRITTEN.  parts: bsi_unet_decode_bwd_parts_floats(B*HW, C, Cout) floats.
size_t bsi_unet_decode_bwd_parts_floats(int M, int C, int Cout) { return (size_t)((M + DB_PIX - 1) / DB_PIX) * ((size_t)Cout * C + 4) + 4; }
int bsi_unet_decode_bwd_det(const float* g_xhat, const float* c_out, int coef_stride, const float* h, int B, int HW, int C, const float* w,
                            int Cout, float* dh, float* dw, float* db, float* parts, bsi_stream_t stream) {
    BSI_CHECK_ARG(g_xhat && h && w && dh && dw && db && parts && B > 0 && HW > 0, "bsi_unet_decode_bwd: bad args");
    BSI_CHECK_ARG(Cout >= 1 && Cout <= 4 && C >= 4 && C <= 256 && 256 % C == 0, "bsi_unet_decode_bwd: Cout=%d (<= 4), C=%d (divides 256)",
                  Cout, C);
    const int M = B * HW, nb = (M + DB_PIX - 1) / DB_PIX;
    hipLaunchKernelGGL(unet_decode_bwd_kernel, dim3(nb), dim3(256), 0, S_(stream), g_xhat, c_out, coef_stride, h, M, C, HW, w, Cout, dh, dw, db,
                       parts);
    BSI_CHECK_LAUNCH("bsi_unet_decode_bwd");
    const size_t stride = (size_t)Cout * C + 4;
    int rc = bsi_reduce_slabs_launch(parts, stride, nb, (size_t)Cout * C, 0, dw, S_(stream));
    float* tmp = parts + (size_t)nb * stride;  // 4 padded bias sums, the first Cout are copied out
    if (rc == BSI_OK) rc = bsi_reduce_slabs_launch(parts + (size_t)Cout * C, stride, nb, 4, 0, tmp, S_(stream));
    if (rc == BSI_OK && hipMemcpyAsync(db, tmp, (size_t)Cout * sizeof(float), hipMemcpyDeviceToDevice, S_(stream)) != hipSuccess) {
        bsi_set_error("bsi_unet_decode_bwd: bias gradient copy failed");
        rc = BSI_ELAUNCH;
    }
    return rc;
}
