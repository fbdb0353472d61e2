// Memory-bound backward kernels of the DiT block (autograd of bsi/models/dit.py:50-55,87-103 of the reference):
// gated-residual backward, LayerNorm+modulate backward, final LayerNorm+decoder backward, small helpers.
// Layout/roles as in dit_ops.hip: one wave per token row, lanes own float4 column groups; the per-sample
// reductions (gradients of the adaLN chunks) are accumulated over a 32-row slab in registers/LDS and then added
// with one fp32 atomic per element and slab (8 adds per element for 256 tokens).
#include <math.h>

#include "common.h"
#include "dit_ops.h"

namespace {

constexpr int TPB = 256;   // 4 waves
constexpr int RPB = 32;    // rows per block (8 per wave); tokens per sample must be a multiple of it

__device__ __forceinline__ f32x4 bf16x4_to_f32(u32x2 w) {
    return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16),
                 __uint_as_float(w[1] & 0xffff0000u)};
}
__device__ __forceinline__ u32x2 f32x4_to_bf16(f32x4 v) {
    u32x2 w;
    w[0] = pack_bf16x2(v[0], v[1]);
    w[1] = pack_bf16x2(v[2], v[3]);
    return w;
}

// x2 = x1 + gate*delta (dit.py:93-102).  Given dX = dL/dx2:
//   d_delta = gate * dX (bf16 out),  d_gate[b] += sum_tokens dX * delta,  x <- x - gate*delta (recovers x1),
//   dL/dx1 = dX (unchanged).
template <int VPL>
__global__ void gate_bwd_kernel(const float* __restrict__ dX, const __bf16* __restrict__ delta, float* __restrict__ x,
                                const float* __restrict__ gate, int gate_stride, float* __restrict__ dgate,
                                int dgate_stride, int M, int d, int tokens, __bf16* __restrict__ ddelta) {
    __shared__ float red[3][1024 * 2];  // [wave 1..3][up to 2048 columns]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * RPB;
    const int b = row0 / tokens;
    const int d4 = d >> 2;
    f32x4 g[VPL], acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        g[i] = (c < d4) ? reinterpret_cast<const f32x4*>(gate + (size_t)b * gate_stride)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int r = 0; r < RPB / 4; ++r) {
        const int row = row0 + wave * (RPB / 4) + r;
        if (row >= M) break;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                const f32x4 dx = reinterpret_cast<const f32x4*>(dX + (size_t)row * d)[c];
                const f32x4 dl = bf16x4_to_f32(reinterpret_cast<const u32x2*>(delta + (size_t)row * d)[c]);
                f32x4 xv = reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c];
                f32x4 dd;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dd[k] = g[i][k] * dx[k];
                    acc[i][k] = __fmaf_rn(dx[k], dl[k], acc[i][k]);
                    xv[k] = __fmaf_rn(-g[i][k], dl[k], xv[k]);
                }
                reinterpret_cast<u32x2*>(ddelta + (size_t)row * d)[c] = f32x4_to_bf16(dd);
                reinterpret_cast<f32x4*>(x + (size_t)row * d)[c] = xv;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) reinterpret_cast<f32x4*>(red[wave - 1])[c] = acc[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                f32x4 t = acc[i];
                for (int w = 0; w < 3; ++w) {
                    const f32x4 o = reinterpret_cast<const f32x4*>(red[w])[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] += o[k];
                }
                float* dst = dgate + (size_t)b * dgate_stride + c * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(dst + k, t[k]);
            }
        }
    }
}

// xn = LN(x) * (1 + scale) + shift (dit.py:50-55).  Given dxn (bf16) and the residual-path gradient dX (fp32, in/out):
//   n = (x - mean) * rstd;  d_shift[b] += sum_t dxn;  d_scale[b] += sum_t dxn * n;  dn = dxn * (1 + scale)
//   dX += rstd * (dn - mean(dn) - n * mean(dn * n))
template <int VPL>
__global__ void ln_mod_bwd_kernel(const __bf16* __restrict__ dxn, const float* __restrict__ x,
                                  const float* __restrict__ scale, int mod_stride, float* __restrict__ dshift,
                                  float* __restrict__ dscale, int dmod_stride, float* __restrict__ dX, int M, int d,
                                  int tokens, float eps, DropCfg dc) {
    __shared__ float red[3][2][1024 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * RPB;
    const int b = row0 / tokens;
    const int d4 = d >> 2;
    f32x4 sc1[VPL], ash[VPL], asc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        const f32x4 s = (c < d4) ? reinterpret_cast<const f32x4*>(scale + (size_t)b * mod_stride)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) sc1[i][k] = s[k] + 1.0f;
        ash[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int r = 0; r < RPB / 4; ++r) {
        const int row = row0 + wave * (RPB / 4) + r;
        if (row >= M) break;
        f32x4 v[VPL], dv[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            v[i] = (c < d4) ? reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            dv[i] = (c < d4) ? bf16x4_to_f32(reinterpret_cast<const u32x2*>(dxn + (size_t)row * d)[c])
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            if (dc.thr && c < d4) {  // gradient through the forward's dropout mask
                const unsigned rh = drop_row(dc, (unsigned)row);
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[i][k] = drop_keep_rc(dc, rh, (unsigned)c * 4 + k) ? dv[i][k] * dc.scale : 0.0f;
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dd = v[i][k] - mean;
                    q = __fmaf_rn(dd, dd, q);
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float n = (v[i][k] - mean) * rstd;
                    const float g = dv[i][k];
                    ash[i][k] += g;
                    asc[i][k] = __fmaf_rn(g, n, asc[i][k]);
                    const float dn = g * sc1[i][k];
                    v[i][k] = n;
                    dv[i][k] = dn;
                    s1 += dn;
                    s2 = __fmaf_rn(dn, n, s2);
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                f32x4 o = reinterpret_cast<const f32x4*>(dX + (size_t)row * d)[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += rstd * (dv[i][k] - m1 - v[i][k] * m2);
                reinterpret_cast<f32x4*>(dX + (size_t)row * d)[c] = o;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                reinterpret_cast<f32x4*>(red[wave - 1][0])[c] = ash[i];
                reinterpret_cast<f32x4*>(red[wave - 1][1])[c] = asc[i];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                f32x4 t0 = ash[i], t1 = asc[i];
                for (int w = 0; w < 3; ++w) {
                    const f32x4 o0 = reinterpret_cast<const f32x4*>(red[w][0])[c];
                    const f32x4 o1 = reinterpret_cast<const f32x4*>(red[w][1])[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { t0[k] += o0[k]; t1[k] += o1[k]; }
                }
                float* d0 = dshift + (size_t)b * dmod_stride + c * 4;
                float* d1 = dscale + (size_t)b * dmod_stride + c * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) { atomicAdd(d0 + k, t0[k]); atomicAdd(d1 + k, t1[k]); }
            }
        }
    }
}

// Fused residual-path backward of one LayerNorm-modulate and the gated residual add below it (dit.py:50-55,93-102), for the
// training tape that keeps the LayerNorm INPUT x (fp32) and its (mean, rstd) per row.  One pass over a row:
//   LN part   (dxn != null):  n = (x - mean) * rstd;  g = dropout(dxn);  dshift[b] += sum_t g;  dscale[b] += sum_t g*n;
//                             dn = g * (1 + scale);   dX += rstd * (dn - mean(dn) - n * mean(dn * n))
//   gate part (delta != null, x_below = x - gate*delta is not needed: the tape holds every LayerNorm input):
//                             ddelta = bf16(gate * dX);  dgate[b] += sum_t dX * delta
// Bytes per element: dxn 2 + x 4 + dX 4 + delta 2 read, dX 4 + ddelta 2 written = 18 (the separate kernels moved 30 and
// rewound x in place).  One wave per row, 16 rows per wave, 64 rows per workgroup; all loads of a row are issued before the
// first use; the two row reductions run on the DPP / lane-swap path; the per-sample column sums are kept in registers over
// the wave's rows, reduced through LDS and added with one fp32 atomic per element per 64-row slab.
constexpr int FRW = 16;         // rows per wave
constexpr int FRB = 4 * FRW;    // rows per workgroup

// HAS_BIAS (round 4, gate part only): the column sums of the bf16 ddelta rows this workgroup writes -- the bias gradient of the Linear
// whose output gradient ddelta is (out-projection / fc2: autograd's grad_output.sum(0)) -- go to row blockIdx.x of `dbias_rows`
// ([M / 64][d] fp32, summed in row order by bsi_colsum_rows_f32).  The weight-gradient GEMM then runs without its fused bias
// gradient, which costs it 10-15 % (tools/tn_bench.py); here the pass is HBM bound and the sums ride along.
template <int VPL, bool HAS_LN, bool HAS_GATE, bool HAS_BIAS = false>
__global__ __launch_bounds__(TPB) void ln_gate_bwd_kernel(
    const __bf16* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ stats,
    const float* __restrict__ scale, int mod_stride, float* __restrict__ dshift, float* __restrict__ dscale, int dmod_stride,
    float* dX, const __bf16* __restrict__ delta, const float* __restrict__ gate, int gate_stride, float* __restrict__ dgate,
    int dgate_stride, __bf16* __restrict__ ddelta, int M, int d, int tokens, DropCfg dc, size_t part_stride,
    float* __restrict__ dbias_rows = nullptr) {
    static_assert(!HAS_BIAS || HAS_GATE, "the bias sums are those of ddelta");
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3 waves][nacc][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * FRB;
    const int b = row0 / tokens;
    // part_stride > 0: this 64-row slab STORES its sums into plane (slab index inside the image) of the modulation gradients --
    // planes are part_stride floats apart and summed in fixed order by bsi_sum_cast_rows_bf16 (reproducible, no zero fill);
    // part_stride == 0: the sums are added to dshift / dscale / dgate with fp32 atomics (the C-ABI entry's accumulate contract)
    const size_t poff = part_stride ? (size_t)((row0 - b * tokens) / FRB) * part_stride : 0;
    const int d4 = d >> 2;
    const float inv_d = 1.0f / (float)d;
    f32x4 sc1[VPL], gt[VPL], ash[VPL], asc[VPL], agt[VPL], abs_[HAS_BIAS ? VPL : 1];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        sc1[i] = z; gt[i] = z; ash[i] = z; asc[i] = z; agt[i] = z;
        if constexpr (HAS_BIAS) abs_[i] = z;
        if (c < d4) {
            if constexpr (HAS_LN) {
                const f32x4 s = reinterpret_cast<const f32x4*>(scale + (size_t)b * mod_stride)[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) sc1[i][k] = s[k] + 1.0f;
            }
            if constexpr (HAS_GATE) gt[i] = reinterpret_cast<const f32x4*>(gate + (size_t)b * gate_stride)[c];
        }
    }
    // The NEXT row's loads go out before this row's arithmetic (round 5): at 190-210 registers two waves share a SIMD, and with one row
    // per wave in flight the memory pipe idled while both reduced their rows.  The bf16 operands wait packed (24 registers more).
    f32x4 nx[VPL], ndx[VPL];
    u32x2 ng[VPL], ndl[VPL];
    float nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](int row) {
        if constexpr (HAS_LN) {
            nmean = stats[2 * (size_t)row];
            nrstd = stats[2 * (size_t)row + 1];
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            nx[i] = z; ndx[i] = z; ng[i] = u32x2{0u, 0u}; ndl[i] = u32x2{0u, 0u};
            if (c < d4) {
                ndx[i] = reinterpret_cast<const f32x4*>(dX + (size_t)row * d)[c];
                if constexpr (HAS_LN) {
                    nx[i] = reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c];
                    ng[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(dxn + (size_t)row * d) + c);
                }
                if constexpr (HAS_GATE) ndl[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(delta + (size_t)row * d) + c);
            }
        }
    };
    if (row0 + wave * FRW < M) fetch(row0 + wave * FRW);
    for (int r = 0; r < FRW; ++r) {
        const int row = row0 + wave * FRW + r;
        if (row >= M) break;
        f32x4 xv[VPL], gv[VPL], dxv[VPL], dlv[VPL];
        const float mean = nmean, rstd = nrstd;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            xv[i] = nx[i];
            dxv[i] = ndx[i];
            gv[i] = bf16x4_to_f32(ng[i]);
            dlv[i] = bf16x4_to_f32(ndl[i]);
        }
        if (r + 1 < FRW && row + 1 < M) fetch(row + 1);
        if constexpr (HAS_LN) {
            float s1 = 0.f, s2 = 0.f;
            const unsigned rh = dc.thr ? drop_row(dc, (unsigned)row) : 0u;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < d4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float g = gv[i][k];
                        if (dc.thr) g = drop_keep_rc(dc, rh, (unsigned)c * 4 + k) ? g * dc.scale : 0.0f;  // forward's mask
                        const float n = (xv[i][k] - mean) * rstd;
                        ash[i][k] += g;
                        asc[i][k] = __fmaf_rn(g, n, asc[i][k]);
                        const float dn = g * sc1[i][k];
                        xv[i][k] = n;
                        gv[i][k] = dn;
                        s1 += dn;
                        s2 = __fmaf_rn(dn, n, s2);
                    }
                }
            }
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            const float m1 = rows_sum(s1) * inv_d, m2 = rows_sum(s2) * inv_d;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) dxv[i][k] += rstd * (gv[i][k] - m1 - xv[i][k] * m2);
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                if constexpr (HAS_LN) reinterpret_cast<f32x4*>(dX + (size_t)row * d)[c] = dxv[i];
                if constexpr (HAS_GATE) {
                    f32x4 dd;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dd[k] = gt[i][k] * dxv[i][k];
                        agt[i][k] = __fmaf_rn(dxv[i][k], dlv[i][k], agt[i][k]);
                    }
                    const u32x2 ddb = f32x4_to_bf16(dd);
                    __builtin_nontemporal_store(ddb, reinterpret_cast<u32x2*>(ddelta + (size_t)row * d) + c);
                    if constexpr (HAS_BIAS) {  // the sum of the values the weight-gradient GEMM will read (bf16), as its fused form took it
                        const f32x4 r = bf16x4_to_f32(ddb);
#pragma unroll
                        for (int k = 0; k < 4; ++k) abs_[i][k] += r[k];
                    }
                }
            }
        }
    }
    // per-sample column sums: waves 1..3 -> LDS -> wave 0 -> one atomic per element
    constexpr int NACC = (HAS_LN ? 2 : 0) + (HAS_GATE ? 1 : 0);
    if (wave > 0) {
        float* my = red + (size_t)(wave - 1) * NACC * d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                int a = 0;
                if constexpr (HAS_LN) {
                    reinterpret_cast<f32x4*>(my)[c] = ash[i];
                    reinterpret_cast<f32x4*>(my + d)[c] = asc[i];
                    a = 2;
                }
                if constexpr (HAS_GATE) reinterpret_cast<f32x4*>(my + (size_t)a * d)[c] = agt[i];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                f32x4 t0 = ash[i], t1 = asc[i], t2 = agt[i];
                for (int w = 0; w < 3; ++w) {
                    const float* o = red + (size_t)w * NACC * d;
                    int a = 0;
                    if constexpr (HAS_LN) {
                        const f32x4 o0 = reinterpret_cast<const f32x4*>(o)[c];
                        const f32x4 o1 = reinterpret_cast<const f32x4*>(o + d)[c];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { t0[k] += o0[k]; t1[k] += o1[k]; }
                        a = 2;
                    }
                    if constexpr (HAS_GATE) {
                        const f32x4 o2 = reinterpret_cast<const f32x4*>(o + (size_t)a * d)[c];
#pragma unroll
                        for (int k = 0; k < 4; ++k) t2[k] += o2[k];
                    }
                }
                if constexpr (HAS_LN) {
                    float* d0 = dshift + poff + (size_t)b * dmod_stride + c * 4;
                    float* d1 = dscale + poff + (size_t)b * dmod_stride + c * 4;
                    if (part_stride) {
                        *reinterpret_cast<f32x4*>(d0) = t0;
                        *reinterpret_cast<f32x4*>(d1) = t1;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { atomicAdd(d0 + k, t0[k]); atomicAdd(d1 + k, t1[k]); }
                    }
                }
                if constexpr (HAS_GATE) {
                    float* d2 = dgate + poff + (size_t)b * dgate_stride + c * 4;
                    if (part_stride) {
                        *reinterpret_cast<f32x4*>(d2) = t2;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) atomicAdd(d2 + k, t2[k]);
                    }
                }
            }
        }
    }
    if constexpr (HAS_BIAS) {  // second round through the same LDS (it stays at 3 x NACC x d floats: four workgroups per CU)
        __syncthreads();
        if (wave > 0) {
            float* my = red + (size_t)(wave - 1) * NACC * d;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < d4) reinterpret_cast<f32x4*>(my)[c] = abs_[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < d4) {
                    f32x4 t = abs_[i];
                    for (int w = 0; w < 3; ++w) {
                        const f32x4 o = reinterpret_cast<const f32x4*>(red + (size_t)w * NACC * d)[c];
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[k] += o[k];
                    }
                    reinterpret_cast<f32x4*>(dbias_rows + (size_t)blockIdx.x * d)[c] = t;
                }
            }
        }
    }
}

// Backward of dit.py:163-172,181 + bsi.py:382-386: given g_xhat [B,C,H,W] (gradient of x_hat = c_skip*mu + c_out*f):
//   dY[token, o] = c_out[b] * g_xhat[b, ch, hh, ww]  (patchify order), y = LN_affine(x) (fp32)
//   dy = sum_o dY[o] * Wdec[o,:],  dlnw += dy * n, dlnb += dy,  dbdec[o] += dY[o],
//   dX = LN_bwd(dy * lnw)     (dX is WRITTEN, it starts the residual gradient)
// The decoder weight gradient dWdec = dY^T y is a token-contraction and goes to the TN GEMM: this kernel writes its
// operands yb = bf16(y) [M, d] and dYb = bf16(dY) [M, Pp] (Pp = P rounded up to 8, zero padded).
// dlnw / dlnb / dbdec: per-wave register accumulators over the wave's rows, one LDS reduction per workgroup, one
// global atomic per element per workgroup.
template <int VPL, bool WLDS>
__global__ void dit_final_bwd_kernel(const float* __restrict__ x, int Mtok, int d, int P, int Pp,
                                     const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                     const float* __restrict__ dec_w, int C, int H, int W, int ps,
                                     const float* __restrict__ g_xhat, const float* __restrict__ c_out, int coef_stride,
                                     float* __restrict__ dX, __bf16* __restrict__ yb, __bf16* __restrict__ dYb,
                                     float* __restrict__ parts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [2][d] + [64] reduction, then (WLDS) [P][d] weights
    float* gsm = sm;
    float* wsm = sm + 2 * (size_t)d + 64;
    const int d4 = d >> 2;
    for (int i = threadIdx.x; i < 2 * d4; i += blockDim.x) reinterpret_cast<f32x4*>(gsm)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WLDS)
        for (int i = threadIdx.x; i < P * d4; i += blockDim.x)
            reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(dec_w)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int nw = W / ps, tokens = (H / ps) * nw, HW = H * W;
    float db_acc = 0.f;  // lane o accumulates d_dec_b[o]
    f32x4 glw[VPL], glb[VPL], lw[VPL], lb[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        glw[i] = glb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        lw[i] = (c < d4) ? reinterpret_cast<const f32x4*>(ln_w)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        lb[i] = (c < d4) ? reinterpret_cast<const f32x4*>(ln_b)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < Mtok; row += gridDim.x * wpb) {
        const int b_idx = row / tokens, tok = row % tokens;
        const int th = tok / nw, tw = tok % nw;
        float dy_o = 0.f;  // lane o holds dY[o]
        if (lane < P) {
            const int intra = lane / C, ch = lane % C;
            const int hh = th * ps + intra / ps, ww = tw * ps + intra % ps;
            dy_o = g_xhat[((size_t)b_idx * C + ch) * HW + (size_t)hh * W + ww];
            if (c_out) dy_o *= c_out[(size_t)b_idx * coef_stride];
            db_acc += dy_o;
        }
        if (lane < Pp) dYb[(size_t)row * Pp + lane] = (__bf16)dy_o;
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            v[i] = (c < d4) ? reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dd = v[i][k] - mean;
                    q = __fmaf_rn(dd, dd, q);
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
        f32x4 dy[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            dy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = (v[i][k] - mean) * rstd;  // n
            if (c < d4) {
                u32x2 pk;
                pk[0] = pack_bf16x2(__fmaf_rn(v[i][0], lw[i][0], lb[i][0]), __fmaf_rn(v[i][1], lw[i][1], lb[i][1]));
                pk[1] = pack_bf16x2(__fmaf_rn(v[i][2], lw[i][2], lb[i][2]), __fmaf_rn(v[i][3], lw[i][3], lb[i][3]));
                reinterpret_cast<u32x2*>(yb + (size_t)row * d)[c] = pk;
            }
        }
        for (int o = 0; o < P; ++o) {
            const float g = __shfl(dy_o, o, 64);
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < d4) {
                    const f32x4 wv = WLDS ? reinterpret_cast<const f32x4*>(wsm + (size_t)o * d)[c]
                                          : reinterpret_cast<const f32x4*>(dec_w + (size_t)o * d)[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) dy[i][k] = __fmaf_rn(g, wv[k], dy[i][k]);
                }
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                glw[i][k] = __fmaf_rn(dy[i][k], v[i][k], glw[i][k]);
                glb[i][k] += dy[i][k];
                const float dn = dy[i][k] * lw[i][k];
                dy[i][k] = dn;
                s1 += dn;
                s2 = __fmaf_rn(dn, v[i][k], s2);
            }
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = rstd * (dy[i][k] - m1 - v[i][k] * m2);
                reinterpret_cast<f32x4*>(dX + (size_t)row * d)[c] = o;
            }
        }
    }
    // waves add their sums to the block's in wave order (one barrier each: once per block, and the order is fixed), the block
    // stores them as slab blockIdx.x of `parts` = [gridDim.x][2 d + 64]; bsi_dit_final_bwd_launch sums the slabs in index order.
    // No atomics: the three gradients are bit reproducible.
    float* dbs = gsm + 2 * (size_t)d;  // [64] decoder bias sums: behind the two LayerNorm rows (the launcher sizes the region)
    if (threadIdx.x < 64) dbs[threadIdx.x] = 0.f;
    __syncthreads();
    for (int w = 0; w < wpb; ++w) {
        if ((int)(threadIdx.x >> 6) == w) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < d4) {
                    f32x4 a = reinterpret_cast<f32x4*>(gsm)[c], b2 = reinterpret_cast<f32x4*>(gsm + d)[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[k] += glw[i][k]; b2[k] += glb[i][k]; }
                    reinterpret_cast<f32x4*>(gsm)[c] = a;
                    reinterpret_cast<f32x4*>(gsm + d)[c] = b2;
                }
            }
            if (lane < P) dbs[lane] += db_acc;
        }
        __syncthreads();
    }
    float* slab = parts + (size_t)blockIdx.x * (2 * (size_t)d + 64);
    for (int i = threadIdx.x; i < 2 * d + 64; i += blockDim.x) slab[i] = gsm[i];
}

// out_bf16 = in_f32 elementwise (optionally * silu'(pre) with pre fp32): used for the adaLN MLP backward
__global__ void silu_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ pre, size_t n, __bf16* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = ds[i];
        if (pre) {
            const float p = pre[i];
            const float sg = 1.0f / (1.0f + __expf(-p));
            g *= sg * (1.0f + p * (1.0f - sg));
        }
        out[i] = (__bf16)g;
    }
}

// out_bf16[r][c] = bf16(sum_p parts[p * part_stride + r * row_stride + c]), planes added in index order (fixed -> reproducible):
// the modulation gradients of one DiT block from the per-slab planes ln_gate_bwd_kernel stored
__global__ void sum_cast_rows_kernel(const float* __restrict__ parts, int nparts, size_t part_stride, int row_stride, int rows, int cols,
                                     __bf16* __restrict__ out, int ld_out) {
    const int c4 = cols >> 2;
    const size_t total = (size_t)rows * c4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), c = (int)(i % c4) * 4;
        const float* src = parts + (size_t)r * row_stride + c;
        f32x4 a = *reinterpret_cast<const f32x4*>(src);
        for (int p = 1; p < nparts; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * part_stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += v[k];
        }
        *reinterpret_cast<u32x2*>(out + (size_t)r * ld_out + c) = f32x4_to_bf16(a);
    }
}

// fp32 [rows, cols] -> bf16 [cols, ld_out] transposed (weight shadows W^T for the input-gradient GEMMs)
__global__ void cast_transpose_kernel(const float* __restrict__ in, int rows, int cols, __bf16* __restrict__ out, int ld) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        tile[ty + k][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.0f;
    }
    __syncthreads();
    for (int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;  // out[c][r]
        if (c < cols && r < ld) out[(size_t)c * ld + r] = (__bf16)tile[tx][ty + k];
    }
}

// Every bf16 shadow of a model's fp32 weight matrices in ONE launch (round 4; before: one 8-us launch per matrix and layout, 265 per
// DiT-L optimizer step): a workgroup takes a 64 x 64 tile of one matrix, reads it once (coalesced 256-B rows) and writes the row-major
// shadow (columns cols .. ld-1 zero) and / or the transposed one (through LDS, 128-B rows on both sides).
constexpr int CB_T = 64;
__global__ __launch_bounds__(256) void cast_batch_kernel(const bsi_cast_desc* __restrict__ descs, int n) {
    __shared__ float tile[CB_T][CB_T + 1];
    int lo = 0, hi = n - 1;  // last descriptor whose first tile is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const bsi_cast_desc d = descs[lo];
    const int t = blockIdx.x - d.tile0;
    const int ldmax = d.dst && d.ld > d.cols ? d.ld : d.cols;  // the row-major shadow may be padded (zero columns)
    const int tcols = (ldmax + CB_T - 1) / CB_T;
    const int r0 = (t / tcols) * CB_T, c0 = (t % tcols) * CB_T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 quads of columns x 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < d.rows) {
            if (c + 3 < d.cols && (d.cols & 3) == 0) v = *reinterpret_cast<const f32x4*>(d.src + (size_t)r * d.cols + c);
            else
                for (int e = 0; e < 4; ++e) v[e] = c + e < d.cols ? d.src[(size_t)r * d.cols + c + e] : 0.f;
            if (d.dst) {
                __bf16* o = reinterpret_cast<__bf16*>(d.dst) + (size_t)r * d.ld + c;
                if (c + 3 < d.ld && (d.ld & 3) == 0) {
                    u32x2 w;
                    w[0] = pack_bf16x2(v[0], v[1]);
                    w[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(o) = w;
                } else {
                    for (int e = 0; e < 4 && c + e < d.ld; ++e) o[e] = (__bf16)v[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ty + 16 * k][4 * tx + e] = v[e];
    }
    if (!d.dst_t) return;  // wave-uniform
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;  // out_t[c][r .. r + 3]
        if (c >= d.cols || r >= d.rows) continue;
        __bf16* o = reinterpret_cast<__bf16*>(d.dst_t) + (size_t)c * d.ld_t + r;
        if (r + 3 < d.rows && (d.ld_t & 3) == 0) {
            u32x2 w;
            w[0] = pack_bf16x2(tile[4 * tx][ty + 16 * k], tile[4 * tx + 1][ty + 16 * k]);
            w[1] = pack_bf16x2(tile[4 * tx + 2][ty + 16 * k], tile[4 * tx + 3][ty + 16 * k]);
            *reinterpret_cast<u32x2*>(o) = w;
        } else {
            for (int e = 0; e < 4 && r + e < d.rows; ++e) o[e] = (__bf16)tile[4 * tx + e][ty + 16 * k];
        }
    }
}

}  // namespace

#define S(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int bsi_cast_batch_tiles(int rows, int cols, int ld) {  // tiles of one descriptor (host helper for tile0)
    const int ldmax = ld > cols ? ld : cols;
    return ((rows + CB_T - 1) / CB_T) * ((ldmax + CB_T - 1) / CB_T);
}

extern "C" int bsi_cast_batch_bf16(const bsi_cast_desc* descs, int n, int tiles, bsi_stream_t stream) {
    BSI_CHECK_ARG(descs && n > 0 && tiles > 0, "bsi_cast_batch_bf16: descriptor table (device), n=%d, tiles=%d", n, tiles);
    hipLaunchKernelGGL(cast_batch_kernel, dim3(tiles), dim3(256), 0, S(stream), descs, n);
    BSI_CHECK_LAUNCH("bsi_cast_batch_bf16");
    return BSI_OK;
}

extern "C" int bsi_gate_bwd(const float* dX, const void* delta, float* x, const float* gate, int gate_stride, float* dgate,
                            int dgate_stride, int M, int d, int tokens, void* ddelta, bsi_stream_t stream) {
    BSI_CHECK_ARG(dX && delta && x && gate && dgate && ddelta, "bsi_gate_bwd: null pointer");
    BSI_CHECK_ARG(M > 0 && d % 4 == 0 && d <= 2048 && tokens % RPB == 0 && M % tokens == 0,
                  "bsi_gate_bwd: M=%d d=%d tokens=%d (tokens must be a multiple of %d)", M, d, tokens, RPB);
    dim3 grid(M / RPB);
    const __bf16* dl = reinterpret_cast<const __bf16*>(delta);
    __bf16* dd = reinterpret_cast<__bf16*>(ddelta);
    if (d <= 256) hipLaunchKernelGGL(gate_bwd_kernel<1>, grid, dim3(TPB), 0, S(stream), dX, dl, x, gate, gate_stride, dgate, dgate_stride, M, d, tokens, dd);
    else if (d <= 1024) hipLaunchKernelGGL(gate_bwd_kernel<4>, grid, dim3(TPB), 0, S(stream), dX, dl, x, gate, gate_stride, dgate, dgate_stride, M, d, tokens, dd);
    else hipLaunchKernelGGL(gate_bwd_kernel<8>, grid, dim3(TPB), 0, S(stream), dX, dl, x, gate, gate_stride, dgate, dgate_stride, M, d, tokens, dd);
    BSI_CHECK_LAUNCH("bsi_gate_bwd");
    return BSI_OK;
}

int bsi_ln_mod_bwd_drop(const void* dxn, const float* x, const float* scale, int mod_stride, float* dshift,
                        float* dscale, int dmod_stride, float* dX, int M, int d, int tokens, float eps, DropCfg dc,
                        bsi_stream_t stream) {
    BSI_CHECK_ARG(dxn && x && scale && dshift && dscale && dX, "bsi_ln_mod_bwd: null pointer");
    BSI_CHECK_ARG(M > 0 && d % 4 == 0 && d <= 2048 && tokens % RPB == 0 && M % tokens == 0,
                  "bsi_ln_mod_bwd: M=%d d=%d tokens=%d", M, d, tokens);
    dim3 grid(M / RPB);
    const __bf16* g = reinterpret_cast<const __bf16*>(dxn);
    if (d <= 256) hipLaunchKernelGGL(ln_mod_bwd_kernel<1>, grid, dim3(TPB), 0, S(stream), g, x, scale, mod_stride, dshift, dscale, dmod_stride, dX, M, d, tokens, eps, dc);
    else if (d <= 1024) hipLaunchKernelGGL(ln_mod_bwd_kernel<4>, grid, dim3(TPB), 0, S(stream), g, x, scale, mod_stride, dshift, dscale, dmod_stride, dX, M, d, tokens, eps, dc);
    else hipLaunchKernelGGL(ln_mod_bwd_kernel<8>, grid, dim3(TPB), 0, S(stream), g, x, scale, mod_stride, dshift, dscale, dmod_stride, dX, M, d, tokens, eps, dc);
    BSI_CHECK_LAUNCH("bsi_ln_mod_bwd");
    return BSI_OK;
}

extern "C" int bsi_ln_mod_bwd(const void* dxn, const float* x, const float* scale, int mod_stride, float* dshift,
                              float* dscale, int dmod_stride, float* dX, int M, int d, int tokens, float eps,
                              bsi_stream_t stream) {
    return bsi_ln_mod_bwd_drop(dxn, x, scale, mod_stride, dshift, dscale, dmod_stride, dX, M, d, tokens, eps, DropCfg{}, stream);
}

// Fused LayerNorm-modulate backward + gated-residual backward (see ln_gate_bwd_kernel).  Either part may be absent:
// dxn == NULL: gate part only (top of the network); delta == NULL: LayerNorm part only (bottom of the network).
int bsi_ln_gate_bwd_drop(const void* dxn, const float* x, const float* stats, const float* scale, int mod_stride, float* dshift,
                         float* dscale, int dmod_stride, float* dX, const void* delta, const float* gate, int gate_stride,
                         float* dgate, int dgate_stride, void* ddelta, int M, int d, int tokens, DropCfg dc, bsi_stream_t stream,
                         size_t part_stride, float* dbias_rows) {
    const bool ln = dxn != nullptr, gt = delta != nullptr;
    BSI_CHECK_ARG(part_stride % 4 == 0 && (part_stride == 0 || (dmod_stride % 4 == 0 && dgate_stride % 4 == 0)),
                  "bsi_ln_gate_bwd: partial planes need 16-B aligned strides");
    BSI_CHECK_ARG(dX && (ln || gt), "bsi_ln_gate_bwd: nothing to do");
    BSI_CHECK_ARG(!ln || (x && stats && scale && dshift && dscale), "bsi_ln_gate_bwd: LayerNorm part needs x, stats, scale, dshift, dscale");
    BSI_CHECK_ARG(!gt || (gate && dgate && ddelta), "bsi_ln_gate_bwd: gate part needs gate, dgate, ddelta");
    BSI_CHECK_ARG(M > 0 && d % 4 == 0 && d <= 1024 && tokens % FRB == 0 && M % tokens == 0,
                  "bsi_ln_gate_bwd: M=%d d=%d (<= 1024) tokens=%d (tokens must be a multiple of %d)", M, d, tokens, FRB);
    dim3 grid(M / FRB);
    const __bf16* g = reinterpret_cast<const __bf16*>(dxn);
    const __bf16* dl = reinterpret_cast<const __bf16*>(delta);
    __bf16* dd = reinterpret_cast<__bf16*>(ddelta);
    const size_t lds = (size_t)3 * ((ln ? 2 : 0) + (gt ? 1 : 0)) * d * sizeof(float);
    BSI_CHECK_ARG(!dbias_rows || (gt && d > 256), "bsi_ln_gate_bwd: bias-gradient rows need the gate part (d > 256 instance)");
#define LGB(V, L, G)                                                                                                         \
    do {                                                                                                                     \
        if (dbias_rows && G)                                                                                                 \
            hipLaunchKernelGGL((ln_gate_bwd_kernel<V, L, G, G>), grid, dim3(TPB), lds, S(stream), g, x, stats, scale, mod_stride, dshift, \
                               dscale, dmod_stride, dX, dl, gate, gate_stride, dgate, dgate_stride, dd, M, d, tokens, dc, part_stride, dbias_rows); \
        else                                                                                                                 \
            hipLaunchKernelGGL((ln_gate_bwd_kernel<V, L, G>), grid, dim3(TPB), lds, S(stream), g, x, stats, scale, mod_stride, dshift, \
                               dscale, dmod_stride, dX, dl, gate, gate_stride, dgate, dgate_stride, dd, M, d, tokens, dc, part_stride); \
    } while (0)
#define LGB_V(L, G)                   \
    do {                              \
        if (d <= 256) LGB(1, L, G);   \
        else LGB(4, L, G);            \
    } while (0)
    if (ln && gt) LGB_V(true, true);
    else if (ln) LGB_V(true, false);
    else LGB_V(false, true);
#undef LGB_V
#undef LGB
    BSI_CHECK_LAUNCH("bsi_ln_gate_bwd");
    return BSI_OK;
}

extern "C" int bsi_ln_gate_bwd(const void* dxn, const float* x, const float* stats, const float* scale, int mod_stride,
                               float* dshift, float* dscale, int dmod_stride, float* dX, const void* delta, const float* gate,
                               int gate_stride, float* dgate, int dgate_stride, void* ddelta, int M, int d, int tokens,
                               bsi_stream_t stream) {
    return bsi_ln_gate_bwd_drop(dxn, x, stats, scale, mod_stride, dshift, dscale, dmod_stride, dX, delta, gate, gate_stride, dgate,
                                dgate_stride, ddelta, M, d, tokens, DropCfg{}, stream, 0);
}

int bsi_reduce_slabs_launch(const float* slabs, size_t slab_stride, int splits, size_t n, int accumulate, float* out, hipStream_t s);

static int final_bwd_grid(int Mtok) {
    int grid = (Mtok + 4 * 16 - 1) / (4 * 16);
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    return grid;
}

int bsi_sum_cast_rows_bf16(const float* parts, int nparts, size_t part_stride, int row_stride, int rows, int cols, void* out, int ld_out,
                           bsi_stream_t stream) {
    BSI_CHECK_ARG(parts && out && nparts > 0 && rows > 0 && cols > 0 && cols % 4 == 0 && row_stride % 4 == 0 && part_stride % 4 == 0 &&
                      ld_out % 4 == 0,
                  "bsi_sum_cast_rows_bf16: bad args");
    const size_t total = (size_t)rows * (cols / 4);
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(sum_cast_rows_kernel, dim3((int)g), dim3(256), 0, S(stream), parts, nparts, part_stride, row_stride, rows, cols,
                       reinterpret_cast<__bf16*>(out), ld_out);
    BSI_CHECK_LAUNCH("bsi_sum_cast_rows_bf16");
    return BSI_OK;
}

int bsi_dit_final_bwd_launch(const float* x, int Mtok, int d, int P, const float* ln_w, const float* ln_b,
                             const float* dec_w, int C, int H, int W, int ps, const float* g_xhat, const float* c_out,
                             int coef_stride, float* dX, void* yb, void* dYb, float* d_dec_b, float* d_ln_w,
                             float* d_ln_b, float* parts, hipStream_t s) {
    // parts: bsi_dit_final_bwd_parts_floats(Mtok, d) floats of scratch; d_dec_b / d_ln_w / d_ln_b are WRITTEN (not accumulated)
    if (!parts) {
        bsi_set_error("bsi_dit_final_bwd: scratch for the per-block sums missing");
        return BSI_EINVAL;
    }
    if (P > 64) {
        bsi_set_error("bsi_dit_final_bwd: decoder P=%d unsupported (needs patch*patch*C <= 64)", P);
        return BSI_EINVAL;
    }
    const int Pp = (P + 7) / 8 * 8;
    const bool wlds = (size_t)(P + 2) * d * sizeof(float) + 256 <= 128 * 1024;
    const size_t lds = (size_t)(wlds ? P + 2 : 2) * d * sizeof(float) + 64 * sizeof(float);
    const int grid = final_bwd_grid(Mtok);
#define LAUNCH_FB(V)                                                                                                   \
    do {                                                                                                               \
        auto kern = wlds ? dit_final_bwd_kernel<V, true> : dit_final_bwd_kernel<V, false>;                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, s, x, Mtok, d, P, Pp, ln_w, ln_b, dec_w, C, H, W, ps, g_xhat, \
                           c_out, coef_stride, dX, reinterpret_cast<__bf16*>(yb), reinterpret_cast<__bf16*>(dYb), parts);  \
    } while (0)
    if (d <= 256) LAUNCH_FB(1);
    else if (d <= 1024) LAUNCH_FB(4);
    else LAUNCH_FB(8);
#undef LAUNCH_FB
    BSI_CHECK_LAUNCH("bsi_dit_final_bwd");
    // slabs [grid][2 d + 64] -> the three gradients, summed in slab order
    const size_t stride = 2 * (size_t)d + 64;
    int rc = bsi_reduce_slabs_launch(parts, stride, grid, (size_t)d, 0, d_ln_w, s);
    if (rc == BSI_OK) rc = bsi_reduce_slabs_launch(parts + d, stride, grid, (size_t)d, 0, d_ln_b, s);
    if (rc == BSI_OK && P % 4 == 0) {
        rc = bsi_reduce_slabs_launch(parts + 2 * (size_t)d, stride, grid, (size_t)P, 0, d_dec_b, s);
    } else if (rc == BSI_OK) {  // odd decoder widths: sum the padded 64 columns behind the slabs, copy the first P
        float* tmp = parts + (size_t)grid * stride;
        rc = bsi_reduce_slabs_launch(parts + 2 * (size_t)d, stride, grid, 64, 0, tmp, s);
        if (rc == BSI_OK && hipMemcpyAsync(d_dec_b, tmp, (size_t)P * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            bsi_set_error("bsi_dit_final_bwd: bias gradient copy failed");
            rc = BSI_ELAUNCH;
        }
    }
    return rc;
}

size_t bsi_dit_final_bwd_parts_floats(int Mtok, int d) { return (size_t)final_bwd_grid(Mtok) * (2 * (size_t)d + 64) + 64; }

extern "C" int bsi_silu_bwd_bf16(const float* ds, const float* pre, size_t n, void* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(ds && out && n > 0, "bsi_silu_bwd_bf16: bad args");
    size_t g = (n + TPB - 1) / TPB;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((int)g), dim3(TPB), 0, S(stream), ds, pre, n, reinterpret_cast<__bf16*>(out));
    BSI_CHECK_LAUNCH("bsi_silu_bwd_bf16");
    return BSI_OK;
}

extern "C" int bsi_cast_transpose_bf16(const float* in, int rows, int cols, void* out, int ld_out, bsi_stream_t stream) {
    BSI_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_out >= rows, "bsi_cast_transpose_bf16: bad args");
    dim3 grid((cols + 31) / 32, (ld_out + 31) / 32);
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(TPB), 0, S(stream), in, rows, cols, reinterpret_cast<__bf16*>(out), ld_out);
    BSI_CHECK_LAUNCH("bsi_cast_transpose_bf16");
    return BSI_OK;
}
