// Memory-bound kernels of the VDM-UNet path (bsi/models/vdm_unet.py, bsi/nn/residual_block.py of the reference):
// GroupNorm(32) (+ SiLU) producer for the convolutions' A operand, the 1x1 decode convolution fused with the
// preconditioning epilogue, NCHW <-> NHWC helpers.
#include <math.h>

#include "common.h"
#include "unet_ops.h"
#include "dit_ops.h"

namespace {

// GroupNorm over one image (H*W pixels x C channels, NHWC fp32; optionally the channel-concatenation of two tensors,
// simplified_unet.py:45-46 `cat((x, x_skip), dim=-3)`), 32 groups, affine, optional SiLU  ->  bf16 NHWC.
// Groups are independent, so the grid is (image, 32-channel slice): one workgroup of 256 threads normalises the 32/cpg
// groups of its slice (B*C/32 workgroups fill the chip also at small batch).  Thread t owns the float4 channel chunk
// (t % 8) of pixels t / 8 + k*32: a wave-instruction covers 8 pixels x 128 B (whole cache lines).
// Two passes over the image (statistics, then normalise); `raw` optionally receives the un-normalised bf16 copy
// (A operand of the 1x1 skip convolution, residual_block.py:40).
constexpr int GN_CS = 32, GN_TPB = 256;  // channels per slice, threads per workgroup
__global__ __launch_bounds__(GN_TPB) void groupnorm_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2,
                                                         int C2, int HW, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int silu,
                                                         __bf16* __restrict__ out, __bf16* __restrict__ raw) {
    __shared__ float red_s[512], red_q[512];  // [pixel row][2-channel sub-chunk]: PPI * CH4 * 2 = 512
    __shared__ float mean_s[16], rstd_s[16];
    const int C = C1 + C2;
    constexpr int CH4 = GN_CS / 4;                // float4 chunks per pixel in this slice
    const int cpg = C / 32;                       // channels per group (2, 4 or 8)
    const int b = blockIdx.x, t = threadIdx.x, cs0 = blockIdx.y * GN_CS;
    const int ch = t % CH4, prow = t / CH4, PPI = GN_TPB / CH4;  // pixel rows handled in parallel
    const int c0 = cs0 + ch * 4;
    const bool second = c0 >= C1;
    const float* src = second ? x2 + (size_t)b * HW * C2 + (c0 - C1) : x1 + (size_t)b * HW * C1 + c0;
    const int sstride = second ? C2 : C1;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;  // channel pairs (c0, c0+1) and (c0+2, c0+3)
    for (int p = prow; p < HW; p += PPI) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * sstride);
        s0 += v[0] + v[1];
        q0 += v[0] * v[0] + v[1] * v[1];
        s1 += v[2] + v[3];
        q1 += v[2] * v[2] + v[3] * v[3];
    }
    const int NS = CH4 * 2;  // sub-chunks per pixel
    red_s[prow * NS + 2 * ch] = s0; red_q[prow * NS + 2 * ch] = q0;
    red_s[prow * NS + 2 * ch + 1] = s1; red_q[prow * NS + 2 * ch + 1] = q1;
    __syncthreads();
    if (t < GN_CS / cpg) {  // local group t: sub-chunks [t*cpg/2, (t+1)*cpg/2)
        const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < PPI; ++r)
            for (int k = k0; k < k1; ++k) { ts += red_s[r * NS + k]; tq += red_q[r * NS + k]; }
        const float n = (float)HW * cpg;
        const float mean = ts / n;
        const float var = fmaxf(tq / n - mean * mean, 0.f);
        mean_s[t] = mean;
        rstd_s[t] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = mean_s[(ch * 4 + k) / cpg]; rstd[k] = rstd_s[(ch * 4 + k) / cpg]; }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    for (int p = prow; p < HW; p += PPI) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * sstride);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y[k] = __fmaf_rn((v[k] - mean[k]) * rstd[k], ga[k], be[k]);
            if (silu) y[k] = y[k] / (1.0f + __expf(-y[k]));
        }
        const size_t o = ((size_t)b * HW + p) * C + c0;
        u32x2 w;
        w[0] = pack_bf16x2(y[0], y[1]);
        w[1] = pack_bf16x2(y[2], y[3]);
        *reinterpret_cast<u32x2*>(out + o) = w;
        if (raw) {
            u32x2 r;
            r[0] = pack_bf16x2(v[0], v[1]);
            r[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(raw + o) = r;
        }
    }
}

// Single-pass variant for images of at most 1024 pixels (every UNet level of the reference, 32 x 32): the
// workgroup's 1024 x CS fp32 slice (128 / 256 KB) lives in REGISTERS (8 / 16 float4 per thread, all loads issued up front), so HBM
// sees one read and one write per element: (4 + 2) B/element instead of (8 + 2).  The variance is taken about the mean
// from the registers (second LDS reduction), which is also the better-conditioned formula.
constexpr int GN_RTPB = 1024;  // threads per workgroup
template <int CS, int GN_RP>     // channels per slice, pixel rows per thread: CS/4 * GN_RP * ... = 1024 pixels x CS channels
__global__ __launch_bounds__(GN_RTPB) void groupnorm_reg_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2,
                                                             int C2, int HW, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int silu,
                                                             __bf16* __restrict__ out, __bf16* __restrict__ raw,
                                                             float* __restrict__ stats) {
    __shared__ float red_a[GN_RTPB / 64 * CS / 2], red_b[GN_RTPB / 64 * CS / 2];  // [wave][2-channel sub-chunk]
    __shared__ float mean_s[CS / 2], rstd_s[CS / 2];  // up to CS / 2 groups per slice (2 channels per group at C = 64)
    const int C = C1 + C2;
    constexpr int CH4 = CS / 4, PPI = GN_RTPB / CH4, NS = CH4 * 2, NWV = GN_RTPB / 64;
    const int cpg = C / 32;
    const int b = blockIdx.x, t = threadIdx.x, cs0 = blockIdx.y * CS;
    const int ch = t % CH4, prow = t / CH4, wv = t >> 6;
    const int c0 = cs0 + ch * 4;
    const bool second = c0 >= C1;
    const float* src = second ? x2 + (size_t)b * HW * C2 + (c0 - C1) : x1 + (size_t)b * HW * C1 + c0;
    const int sstride = second ? C2 : C1;
    f32x4 v[GN_RP];
#pragma unroll
    for (int k = 0; k < GN_RP; ++k) {
        const int p = prow + k * PPI;
        v[k] = p < HW ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)p * sstride)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ngl = CS / cpg;  // groups in this slice
    const float n = (float)HW * cpg;
    // mean
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < GN_RP; ++k) { s0 += v[k][0] + v[k][1]; s1 += v[k][2] + v[k][3]; }
    // a wave holds 64 / CH4 pixel rows x CH4 chunks: fold the rows (upper lane bits), lanes 0..CH4-1 publish the wave's sums
#pragma unroll
    for (int m = CH4; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
    if ((t & 63) < CH4) { red_a[wv * NS + 2 * ch] = s0; red_a[wv * NS + 2 * ch + 1] = s1; }
    __syncthreads();
    if (t < ngl) {
        const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
        float ts = 0.f;
        for (int r = 0; r < NWV; ++r)
            for (int k = k0; k < k1; ++k) ts += red_a[r * NS + k];
        mean_s[t] = ts / n;
    }
    __syncthreads();
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mean[k] = mean_s[(ch * 4 + k) / cpg];
    // variance about the mean (padding rows beyond HW hold zeros and are skipped)
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int k = 0; k < GN_RP; ++k) {
        if (prow + k * PPI < HW) {
            const float d0 = v[k][0] - mean[0], d1 = v[k][1] - mean[1], d2 = v[k][2] - mean[2], d3 = v[k][3] - mean[3];
            q0 += d0 * d0 + d1 * d1;
            q1 += d2 * d2 + d3 * d3;
        }
    }
#pragma unroll
    for (int m = CH4; m < 64; m <<= 1) { q0 += __shfl_xor(q0, m); q1 += __shfl_xor(q1, m); }
    if ((t & 63) < CH4) { red_b[wv * NS + 2 * ch] = q0; red_b[wv * NS + 2 * ch + 1] = q1; }
    __syncthreads();
    if (t < ngl) {
        const int k0 = t * cpg / 2, k1 = (t + 1) * cpg / 2;
        float tq = 0.f;
        for (int r = 0; r < NWV; ++r)
            for (int k = k0; k < k1; ++k) tq += red_b[r * NS + k];
        rstd_s[t] = 1.0f / sqrtf(tq / n + eps);
        if (stats) {  // (mean, rstd) of group cs0/cpg + t of image b: the backward kernel starts from them (training tape)
            float* st = stats + ((size_t)b * 32 + cs0 / cpg + t) * 2;
            st[0] = mean_s[t];
            st[1] = rstd_s[t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) rstd[k] = rstd_s[(ch * 4 + k) / cpg];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
#pragma unroll
    for (int k = 0; k < GN_RP; ++k) {
        const int p = prow + k * PPI;
        if (p >= HW) continue;
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = __fmaf_rn((v[k][e] - mean[e]) * rstd[e], ga[e], be[e]);
            if (silu) y[e] = y[e] / (1.0f + __expf(-y[e]));
        }
        const size_t o = ((size_t)b * HW + p) * C + c0;
        u32x2 w;
        w[0] = pack_bf16x2(y[0], y[1]);
        w[1] = pack_bf16x2(y[2], y[3]);
        *reinterpret_cast<u32x2*>(out + o) = w;
        if (raw) {
            u32x2 r;
            r[0] = pack_bf16x2(v[k][0], v[k][1]);
            r[1] = pack_bf16x2(v[k][2], v[k][3]);
            *reinterpret_cast<u32x2*>(raw + o) = r;
        }
    }
}

// GroupNorm as one streaming pass (bsi_groupnorm_apply_nhwc): the statistics arrive as (mean, M2) partials of 128-pixel x
// 4-channel blocks from the epilogue of the convolution that produced x (conv_igemm.hip, store_f32_rows), so nothing has to be
// held between a reduction and the normalisation.  Workgroup = 128 pixels of one image; its first wave merges the image's
// partials per group (fixed order, Chan et al.'s pairwise update: deterministic), then every thread owns ONE 4-channel column
// quad (its mean / rstd / gamma / beta stay in registers) and walks the pixels with 8 independent 16-B loads in flight.
constexpr int GA_TPB = 256, GA_PIX = 128;
template <int C>  // channels of cat(x1, x2): 128 or 256
__global__ __launch_bounds__(GA_TPB) void groupnorm_apply_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ part1,
                                                                 const float* __restrict__ x2, const float* __restrict__ part2, int HW,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 int silu, __bf16* __restrict__ out, __bf16* __restrict__ raw,
                                                                 float* __restrict__ stats) {
    __shared__ float mean_s[32], rstd_s[32];
    constexpr int CPG = C / 32, UPG = CPG / 4, Q = C / 4, PPI = GA_TPB / Q;  // units per group, quads per pixel, pixels per trip
    const int b = blockIdx.x, chunk = blockIdx.y, t = threadIdx.x;
    const int nblk = HW / 128, C2 = C - C1;
    if (t < 32) {
        const int c0 = t * CPG;
        const bool second = c0 >= C1;
        const int Cs = second ? C2 : C1;
        const float* part = (second ? part2 : part1) + ((size_t)b * nblk * (Cs / 4) + (second ? c0 - C1 : c0) / 4) * 2;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int k = 0; k < nblk; ++k)
#pragma unroll
            for (int u = 0; u < UPG; ++u) {
                const f32x2 pm = *reinterpret_cast<const f32x2*>(part + ((size_t)k * (Cs / 4) + u) * 2);
                const float nb = 512.f, nn = n + nb, delta = pm[0] - mean;
                mean += delta * (nb / nn);
                m2 += pm[1] + delta * delta * (n * nb / nn);
                n = nn;
            }
        const float rstd = 1.0f / sqrtf(m2 / n + eps);
        mean_s[t] = mean;
        rstd_s[t] = rstd;
        if (stats && chunk == 0) {
            float* st = stats + ((size_t)b * 32 + t) * 2;
            st[0] = mean;
            st[1] = rstd;
        }
    }
    __syncthreads();
    const int q = t % Q, prow = t / Q;
    const int c0 = q * 4;
    const bool second = c0 >= C1;
    const int sstride = second ? C2 : C1;
    const size_t pix0 = (size_t)b * HW + (size_t)chunk * GA_PIX;
    const float* src = (second ? x2 + pix0 * C2 + (c0 - C1) : x1 + pix0 * C1 + c0);
    const float mean = mean_s[c0 / CPG], rstd = rstd_s[c0 / CPG];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    constexpr int TRIPS = GA_PIX / PPI, U = 8;
#pragma unroll 1
    for (int k0 = 0; k0 < TRIPS; k0 += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(prow + (k0 + u) * PPI) * sstride));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = __fmaf_rn((v[u][e] - mean) * rstd, ga[e], be[e]);
                if (silu) y[e] = y[e] / (1.0f + __expf(-y[e]));
            }
            const size_t o = (pix0 + prow + (k0 + u) * PPI) * C + c0;
            u32x2 w;
            w[0] = pack_bf16x2(y[0], y[1]);
            w[1] = pack_bf16x2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(out + o) = w;
            if (raw) {
                u32x2 r;
                r[0] = pack_bf16x2(v[u][0], v[u][1]);
                r[1] = pack_bf16x2(v[u][2], v[u][3]);
                *reinterpret_cast<u32x2*>(raw + o) = r;
            }
        }
    }
}

// decode: Conv2d(C -> Cout, 1x1) in fp32 on the NHWC fp32 feature map, written NCHW, fused with
// x_hat = c_skip*mu + c_out*f (vdm_unet.py:72,100; bsi.py:382-386).  One thread per pixel.
__global__ void unet_decode_kernel(const float* __restrict__ h, int M, int C, int HW, const float* __restrict__ w,
                                   const float* __restrict__ bias, int Cout, const float* __restrict__ mu,
                                   const float* __restrict__ c_skip, const float* __restrict__ c_out, int coef_stride,
                                   float* __restrict__ out) {
    extern __shared__ float wsm[];  // [Cout][C]
    for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) wsm[i] = w[i];
    __syncthreads();
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int b = m / HW, pix = m % HW;
    const float* hr = h + (size_t)m * C;
    for (int o = 0; o < Cout; ++o) {
        float a = 0.f;
        for (int c = 0; c < C; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hr + c);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(wsm + o * C + c);
            a = __fmaf_rn(v[0], ww[0], a); a = __fmaf_rn(v[1], ww[1], a); a = __fmaf_rn(v[2], ww[2], a); a = __fmaf_rn(v[3], ww[3], a);
        }
        a += bias[o];
        const size_t gi = ((size_t)b * Cout + o) * HW + pix;
        if (c_skip) a = __fmaf_rn(c_out[(size_t)b * coef_stride], a, __fmul_rn(c_skip[(size_t)b * coef_stride], mu[gi]));
        out[gi] = a;
    }
}

// weight re-arrangement for the implicit GEMM: fp32 [Cout][Cin][kh][kw] -> bf16 [Cout][ld] with K index (tap, channel),
// Cin zero-padded to cin_pad, written at column offset col0 (so a 1x1 skip weight can be appended behind a 3x3 one).
__global__ void conv_weight_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int cin_pad, int ld, int col0,
                                        __bf16* __restrict__ out) {
    const size_t total = (size_t)Cout * taps * cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin_pad);
        const int tap = (int)((i / cin_pad) % taps);
        const int o = (int)(i / ((size_t)cin_pad * taps));
        const float v = c < Cin ? w[((size_t)o * Cin + c) * taps + tap] : 0.0f;
        out[(size_t)o * ld + col0 + tap * cin_pad + c] = (__bf16)v;
    }
}

// fp32 Conv2d weight [Cout][Cin][taps] -> bf16 [Cin][ld] with out[ci][(taps-1-tap)*Cout + co] = w[co][ci][tap]: the weights
// of the input-gradient convolution (180-degree rotated taps, channels swapped), in the forward kernel's packed layout.
__global__ void conv_weight_pack_t_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int ld,
                                          __bf16* __restrict__ out) {
    const size_t total = (size_t)Cin * taps * Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % Cout);
        const int tap = (int)((i / Cout) % taps);
        const int c = (int)(i / ((size_t)Cout * taps));
        out[(size_t)c * ld + (taps - 1 - tap) * Cout + o] = (__bf16)w[((size_t)o * Cin + c) * taps + tap];
    }
}

// Both re-arrangements for MANY convolutions in one launch (a train step re-packs every weight of the UNet after the optimizer
// update: 167 + 166 launches of ~4 us otherwise).  blockIdx.x = descriptor, blockIdx.y strides its elements.
__global__ void conv_weight_pack_batch_kernel(const bsi_conv_pack_desc* __restrict__ descs, int transposed) {
    // one thread per (output channel, input channel) pair: its `taps` source floats are contiguous; the pair index is ordered so
    // that neighbouring threads WRITE neighbouring bf16 (input channel fastest for the forward layout, output channel fastest for
    // the transposed one)
    const bsi_conv_pack_desc d = descs[blockIdx.x];
    const float* __restrict__ w = d.w;
    __bf16* __restrict__ out = reinterpret_cast<__bf16*>(d.out);
    const int stride = gridDim.y * blockDim.x;
    if (!transposed) {
        const int total = d.Cout * d.cin_pad;
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int c = i % d.cin_pad, o = i / d.cin_pad;
            const float* src = w + ((size_t)o * d.Cin + c) * d.taps;
            __bf16* dst = out + (size_t)o * d.ld + d.col0 + c;
            for (int tap = 0; tap < d.taps; ++tap) dst[tap * d.cin_pad] = (__bf16)(c < d.Cin ? src[tap] : 0.0f);
        }
    } else {
        const int total = d.Cin * d.Cout;
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int o = i % d.Cout, c = i / d.Cout;
            const float* src = w + ((size_t)o * d.Cin + c) * d.taps;
            __bf16* dst = out + (size_t)c * d.ld + o;
            for (int tap = 0; tap < d.taps; ++tap) dst[(d.taps - 1 - tap) * d.Cout] = (__bf16)src[tap];
        }
    }
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int bsi_conv_weight_pack_t(const float* w, int Cout, int Cin, int taps, int ld, void* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && taps > 0 && ld >= taps * Cout, "bsi_conv_weight_pack_t: bad args");
    const size_t total = (size_t)Cin * taps * Cout;
    size_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(conv_weight_pack_t_kernel, dim3((int)g), dim3(256), 0, S(stream), w, Cout, Cin, taps, ld,
                       reinterpret_cast<__bf16*>(out));
    BSI_CHECK_LAUNCH("bsi_conv_weight_pack_t");
    return BSI_OK;
}

extern "C" int bsi_conv_weight_pack_batch(const bsi_conv_pack_desc* descs, int n, int transposed, bsi_stream_t stream) {
    BSI_CHECK_ARG(descs && n > 0, "bsi_conv_weight_pack_batch: bad args");
    hipLaunchKernelGGL(conv_weight_pack_batch_kernel, dim3(n, 16), dim3(256), 0, S(stream), descs, transposed);
    BSI_CHECK_LAUNCH("bsi_conv_weight_pack_batch");
    return BSI_OK;
}

static int groupnorm_impl(const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma, const float* beta, float eps,
                          int silu, void* out_bf16, void* raw_bf16, float* stats, bsi_stream_t stream) {
    BSI_CHECK_ARG(x1 && gamma && beta && out_bf16 && B > 0 && HW > 0, "bsi_groupnorm_nhwc: bad args");
    const int C = C1 + C2;
    BSI_CHECK_ARG((C == 128 || C == 256 || C == 64) && C1 % 32 == 0 && C2 % 32 == 0 && (C2 == 0 || x2),
                  "bsi_groupnorm_nhwc: C1+C2=%d unsupported (64, 128 or 256 channels, 32 groups)", C);
    BSI_CHECK_ARG(!stats || HW <= 1024, "bsi_groupnorm_stats_nhwc: statistics output needs H*W <= 1024");
    __bf16* o = reinterpret_cast<__bf16*>(out_bf16);
    __bf16* r = reinterpret_cast<__bf16*>(raw_bf16);
    if (HW <= 1024 && C1 % 64 == 0 && C2 % 64 == 0)  // 64-channel slices: whole 128-B lines of bf16 output (measured 60 / 126 us vs 67 / 154)
        hipLaunchKernelGGL((groupnorm_reg_kernel<64, 16>), dim3(B, C / 64), dim3(GN_RTPB), 0, S(stream), x1, C1, x2, C2, HW, gamma, beta, eps,
                           silu, o, r, stats);
    else if (HW <= 1024)
        hipLaunchKernelGGL((groupnorm_reg_kernel<32, 8>), dim3(B, C / 32), dim3(GN_RTPB), 0, S(stream), x1, C1, x2, C2, HW, gamma, beta, eps,
                           silu, o, r, stats);
    else
        hipLaunchKernelGGL(groupnorm_kernel, dim3(B, C / GN_CS), dim3(GN_TPB), 0, S(stream), x1, C1, x2, C2, HW, gamma, beta, eps, silu, o, r);
    BSI_CHECK_LAUNCH("bsi_groupnorm_nhwc");
    return BSI_OK;
}

extern "C" int bsi_groupnorm_nhwc(const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                                  const float* beta, float eps, int silu, void* out_bf16, void* raw_bf16,
                                  bsi_stream_t stream) {
    return groupnorm_impl(x1, C1, x2, C2, B, HW, gamma, beta, eps, silu, out_bf16, raw_bf16, nullptr, stream);
}

extern "C" int bsi_groupnorm_stats_nhwc(const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                                        const float* beta, float eps, int silu, void* out_bf16, void* raw_bf16, float* stats,
                                        bsi_stream_t stream) {
    BSI_CHECK_ARG(stats, "bsi_groupnorm_stats_nhwc: statistics pointer missing");
    return groupnorm_impl(x1, C1, x2, C2, B, HW, gamma, beta, eps, silu, out_bf16, raw_bf16, stats, stream);
}

extern "C" int bsi_groupnorm_apply_nhwc(const float* x1, int C1, const float* part1, const float* x2, int C2, const float* part2, int B,
                                        int HW, const float* gamma, const float* beta, float eps, int silu, void* out_bf16,
                                        void* raw_bf16, float* stats, bsi_stream_t stream) {
    BSI_CHECK_ARG(x1 && part1 && gamma && beta && out_bf16 && B > 0 && HW > 0, "bsi_groupnorm_apply_nhwc: bad args");
    const int C = C1 + C2;
    BSI_CHECK_ARG((C == 128 || C == 256) && C1 % (C / 32) == 0 && C1 % 4 == 0 && C2 % 4 == 0 && (C2 == 0 || (x2 && part2)),
                  "bsi_groupnorm_apply_nhwc: C1=%d C2=%d unsupported (128 or 256 channels in all, groups within one tensor)", C1, C2);
    BSI_CHECK_ARG(HW % 128 == 0, "bsi_groupnorm_apply_nhwc: H*W=%d must be a multiple of 128 (the partials' block)", HW);
    __bf16* o = reinterpret_cast<__bf16*>(out_bf16);
    __bf16* r = reinterpret_cast<__bf16*>(raw_bf16);
    const dim3 grid(B, HW / GA_PIX);
    if (C == 128)
        hipLaunchKernelGGL(groupnorm_apply_kernel<128>, grid, dim3(GA_TPB), 0, S(stream), x1, C1, part1, x2, part2, HW, gamma, beta, eps, silu, o, r, stats);
    else
        hipLaunchKernelGGL(groupnorm_apply_kernel<256>, grid, dim3(GA_TPB), 0, S(stream), x1, C1, part1, x2, part2, HW, gamma, beta, eps, silu, o, r, stats);
    BSI_CHECK_LAUNCH("bsi_groupnorm_apply_nhwc");
    return BSI_OK;
}

// See unet_ops.h (bsi_groupnorm_apply_split).  Same streaming structure as groupnorm_apply_kernel<128>: workgroup = 128 pixels of one
// image, a thread owns one 4-channel quad; the statistics of BOTH groupings (4- and 8-channel groups) come from the same partials.
struct GnTargetDev { __bf16* out; __bf16* raw; const float* gamma; const float* beta; int ld, col0, silu; };
template <int CPG1, int CPG2>  // channels per group of the two targets (CPG2 = 0: one target)
__global__ __launch_bounds__(GA_TPB) void groupnorm_apply_split_kernel(const float* __restrict__ x, const float* __restrict__ part, int HW, float eps,
                                                                       GnTargetDev t1, GnTargetDev t2) {
    constexpr int C = 128, Q = C / 4, PPI = GA_TPB / Q;  // 32 quads per pixel, 8 pixels per trip
    __shared__ float mean1[32], rstd1[32], mean2[32], rstd2[32];
    const int b = blockIdx.x, chunk = blockIdx.y, t = threadIdx.x;
    const int nblk = HW / 128;
    auto merge = [&](int g, int cpg, float* mean_s, float* rstd_s) {  // group g = channels [g cpg, (g + 1) cpg): cpg / 4 units, fixed order
        const float* pp = part + ((size_t)b * nblk * (C / 4) + (size_t)g * (cpg / 4)) * 2;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int k = 0; k < nblk; ++k)
            for (int u = 0; u < cpg / 4; ++u) {
                const f32x2 pm = *reinterpret_cast<const f32x2*>(pp + ((size_t)k * (C / 4) + u) * 2);
                const float nb = 512.f, nn = n + nb, delta = pm[0] - mean;
                mean += delta * (nb / nn);
                m2 += pm[1] + delta * delta * (n * nb / nn);
                n = nn;
            }
        mean_s[g] = mean;
        rstd_s[g] = 1.0f / sqrtf(m2 / n + eps);
    };
    if (t < C / CPG1) merge(t, CPG1, mean1, rstd1);
    if constexpr (CPG2 > 0) {
        if (t >= 64 && t < 64 + C / CPG2) merge(t - 64, CPG2, mean2, rstd2);
    }
    __syncthreads();
    const int q = t % Q, prow = t / Q, c0 = q * 4;
    const size_t pix0 = (size_t)b * HW + (size_t)chunk * GA_PIX;
    const float* src = x + pix0 * C + c0;
    const float m1 = mean1[c0 / CPG1], r1 = rstd1[c0 / CPG1];
    const f32x4 ga1 = *reinterpret_cast<const f32x4*>(t1.gamma + c0), be1 = *reinterpret_cast<const f32x4*>(t1.beta + c0);
    float m2v = 0.f, r2 = 0.f;
    f32x4 ga2 = f32x4{0.f, 0.f, 0.f, 0.f}, be2 = ga2;
    if constexpr (CPG2 > 0) {
        m2v = mean2[c0 / CPG2]; r2 = rstd2[c0 / CPG2];
        ga2 = *reinterpret_cast<const f32x4*>(t2.gamma + c0); be2 = *reinterpret_cast<const f32x4*>(t2.beta + c0);
    }
    auto emit = [&](const GnTargetDev& tg, const f32x4& v, float mean, float rstd, const f32x4& ga, const f32x4& be, size_t pix) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = __fmaf_rn((v[e] - mean) * rstd, ga[e], be[e]);
            if (tg.silu) y[e] = y[e] / (1.0f + __expf(-y[e]));
        }
        const size_t o = pix * tg.ld + tg.col0 + c0;
        u32x2 w;
        w[0] = pack_bf16x2(y[0], y[1]);
        w[1] = pack_bf16x2(y[2], y[3]);
        *reinterpret_cast<u32x2*>(tg.out + o) = w;
        if (tg.raw) {
            u32x2 r;
            r[0] = pack_bf16x2(v[0], v[1]);
            r[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(tg.raw + o) = r;
        }
    };
    constexpr int TRIPS = GA_PIX / PPI, U = 8;
#pragma unroll 1
    for (int k0 = 0; k0 < TRIPS; k0 += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(prow + (k0 + u) * PPI) * C));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t pix = pix0 + prow + (k0 + u) * PPI;
            emit(t1, v[u], m1, r1, ga1, be1, pix);
            if constexpr (CPG2 > 0) emit(t2, v[u], m2v, r2, ga2, be2, pix);
        }
    }
}

int bsi_groupnorm_apply_split(const float* x, const float* part, int B, int HW, float eps, GnTarget first, GnTarget second, bsi_stream_t stream) {
    BSI_CHECK_ARG(x && part && first.out && first.gamma && first.beta && B > 0 && HW > 0 && HW % 128 == 0, "bsi_groupnorm_apply_split: bad args");
    BSI_CHECK_ARG((first.cpg == 4 || first.cpg == 8) && (!second.out || second.cpg == 8) && first.ld % 4 == 0 && first.col0 % 4 == 0 &&
                      (!second.out || (second.gamma && second.beta && second.ld % 4 == 0 && second.col0 % 4 == 0)),
                  "bsi_groupnorm_apply_split: unsupported grouping / layout");
    auto dev = [](const GnTarget& g) {
        return GnTargetDev{reinterpret_cast<__bf16*>(g.out), reinterpret_cast<__bf16*>(g.raw), g.gamma, g.beta, g.ld, g.col0, g.silu};
    };
    const dim3 grid(B, HW / GA_PIX);
    const GnTargetDev a = dev(first), b = dev(second);
    if (second.out && first.cpg == 4) hipLaunchKernelGGL((groupnorm_apply_split_kernel<4, 8>), grid, dim3(GA_TPB), 0, S(stream), x, part, HW, eps, a, b);
    else if (second.out) hipLaunchKernelGGL((groupnorm_apply_split_kernel<8, 8>), grid, dim3(GA_TPB), 0, S(stream), x, part, HW, eps, a, b);
    else if (first.cpg == 4) hipLaunchKernelGGL((groupnorm_apply_split_kernel<4, 0>), grid, dim3(GA_TPB), 0, S(stream), x, part, HW, eps, a, b);
    else hipLaunchKernelGGL((groupnorm_apply_split_kernel<8, 0>), grid, dim3(GA_TPB), 0, S(stream), x, part, HW, eps, a, b);
    BSI_CHECK_LAUNCH("bsi_groupnorm_apply_split");
    return BSI_OK;
}

extern "C" int bsi_unet_decode(const float* h, int B, int HW, int C, const float* w, const float* bias, int Cout,
                               const float* mu, const float* c_skip, const float* c_out, int coef_stride, float* out,
                               bsi_stream_t stream) {
    BSI_CHECK_ARG(h && w && bias && out && B > 0 && HW > 0 && C % 4 == 0 && Cout > 0 && Cout * C * 4 <= 64 * 1024,
                  "bsi_unet_decode: bad args");
    BSI_CHECK_ARG((c_skip == nullptr) == (c_out == nullptr) && (!c_skip || mu), "bsi_unet_decode: coefficients incomplete");
    const int M = B * HW;
    hipLaunchKernelGGL(unet_decode_kernel, dim3((M + 255) / 256), dim3(256), (size_t)Cout * C * sizeof(float), S(stream), h, M, C,
                       HW, w, bias, Cout, mu, c_skip, c_out, coef_stride, out);
    BSI_CHECK_LAUNCH("bsi_unet_decode");
    return BSI_OK;
}

extern "C" int bsi_conv_weight_pack(const float* w, int Cout, int Cin, int taps, int cin_pad, int ld, int col0, void* out,
                                    bsi_stream_t stream) {
    BSI_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && (taps == 1 || taps == 9) && cin_pad >= Cin && ld >= col0 + taps * cin_pad,
                  "bsi_conv_weight_pack: bad args");
    const size_t total = (size_t)Cout * taps * cin_pad;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3((int)g), dim3(256), 0, S(stream), w, Cout, Cin, taps, cin_pad, ld, col0,
                       reinterpret_cast<__bf16*>(out));
    BSI_CHECK_LAUNCH("bsi_conv_weight_pack");
    return BSI_OK;
}
