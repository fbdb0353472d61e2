// Backward of the non-causal softmax attention (autograd of F.scaled_dot_product_attention at
// bsi/models/dit.py:43-44): dQ, dK, dV from Q, K, V, O, dO and the saved log-sum-exp.
//
// One workgroup (8 waves) per (batch, head); all of Q, K, V, dO (tokens <= 256, dh = 64: 4 x 32 KB) sit in LDS
// in ONE image per tile that serves both row reads (ds_read_b128) and hardware-transposed reads
// (ds_read_b64_tr_b16): 16-B chunk index ^= ((row >> 1) & 3) << 1.
//   pass 1: a wave owns 32 queries.  S^T = K.Q^T and dP^T = V.dO^T put the query on the lane, so
//           P = exp(S*scale - lse), dS = scale * P * (dP - delta) are lane-local, and dS^T (packed to bf16) is
//           directly the B operand of dQ^T = K^T . dS^T.
//   pass 2: a wave owns 32 keys.  S = Q.K^T and dP = dO.V^T put the key on the lane; P and dS are the B operands
//           of dV^T = dO^T . P and dK^T = Q^T . dS.  No cross-wave reduction, no atomics, deterministic.
// S is recomputed in both passes (7 matrix products instead of 5); attention is 4 % of the model's FLOPs.
#include "common.h"
#include "dit_ops.h"

namespace {

constexpr int DH = 64, RB = 128;

__device__ __forceinline__ int sw(int r) { return ((r >> 1) & 3) << 1; }
__device__ __forceinline__ const char* row_chunk(const char* tile, int r, int c) { return tile + r * RB + ((c ^ sw(r)) << 4); }

#define TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

__global__ __launch_bounds__(512) void attention_bwd_kernel(const __bf16* __restrict__ qkv, int ld_qkv,
                                                            const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                            int ld_o, const float* __restrict__ lse, int T, int heads,
                                                            __bf16* __restrict__ dqkv, int ld_dqkv, float scale, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Ql = lds;
    char* Kl = Ql + T * RB;
    char* Vl = Kl + T * RB;
    char* Dl = Vl + T * RB;                              // dO
    float* lse_s = reinterpret_cast<float*>(Dl + T * RB);  // [T] lse * log2(e)
    float* dlt_s = lse_s + T;                            // [T] delta = rowsum(dO * O)
    unsigned* rowh_s = reinterpret_cast<unsigned*>(dlt_s + T);  // [T] per-query dropout row hash

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int g = lane >> 4, c16 = lane & 15, qp = c16 >> 2, pp = c16 & 3;
    const float L2E = 1.4426950408889634f;
    const float sl2 = scale * L2E;

    const __bf16* Qg = qkv + (size_t)b * T * ld_qkv + h * DH;
    const __bf16* Kg = Qg + heads * DH;
    const __bf16* Vg = Qg + 2 * heads * DH;
    const __bf16* Og = o + (size_t)b * T * ld_o + h * DH;
    const __bf16* Dg = dout + (size_t)b * T * ld_o + h * DH;

    // ---- stage the four tiles (8 chunks of 16 B per row)
    for (int idx = tid; idx < T * 8; idx += 512) {
        const int r = idx >> 3, c = idx & 7;
        const int off = r * RB + ((c ^ sw(r)) << 4);
        *reinterpret_cast<u32x4*>(Ql + off) = *reinterpret_cast<const u32x4*>(Qg + (size_t)r * ld_qkv + c * 8);
        *reinterpret_cast<u32x4*>(Kl + off) = *reinterpret_cast<const u32x4*>(Kg + (size_t)r * ld_qkv + c * 8);
        *reinterpret_cast<u32x4*>(Vl + off) = *reinterpret_cast<const u32x4*>(Vg + (size_t)r * ld_qkv + c * 8);
        *reinterpret_cast<u32x4*>(Dl + off) = *reinterpret_cast<const u32x4*>(Dg + (size_t)r * ld_o + c * 8);
    }
    // delta[q] = sum_d dO[q,d] * O[q,d]: two threads per row, 32 elements each
    for (int idx = tid; idx < T * 2; idx += 512) {
        const int r = idx >> 1, hlf = idx & 1;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 dv = *reinterpret_cast<const u32x4*>(Dg + (size_t)r * ld_o + hlf * 32 + c * 8);
            const u32x4 ov = *reinterpret_cast<const u32x4*>(Og + (size_t)r * ld_o + hlf * 32 + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a = __fmaf_rn(__uint_as_float(dv[e] << 16), __uint_as_float(ov[e] << 16), a);
                a = __fmaf_rn(__uint_as_float(dv[e] & 0xffff0000u), __uint_as_float(ov[e] & 0xffff0000u), a);
            }
        }
        a += __shfl_xor(a, 1, 64);
        if (hlf == 0) {
            dlt_s[r] = a;
            lse_s[r] = lse[(size_t)bh * T + r] * L2E;
            rowh_s[r] = drop_row(dc, (unsigned)bh * T + r);
        }
    }
    __syncthreads();

    union Frag { bf16x8 v; s16x4 h[2]; };
    const int r0 = wave * 32;  // this wave's 32 queries (pass 1) / 32 keys (pass 2)
    if (r0 < T) {
        // =========================== pass 1: dQ for queries r0 .. r0+31 ===========================
        bf16x8 qf[2][2], dof[2][2];
        float lq[2], dq_delta[2];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            const int q = r0 + 16 * jq + c16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[jq][ks] = *reinterpret_cast<const bf16x8*>(row_chunk(Ql, q, 4 * ks + g));
                dof[jq][ks] = *reinterpret_cast<const bf16x8*>(row_chunk(Dl, q, 4 * ks + g));
            }
            lq[jq] = lse_s[q];
            dq_delta[jq] = dlt_s[q];
        }
        f32x4 dq[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dq[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kc = 0; kc < T; kc += 64) {
            f32x4 s[4][2], dp[4][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt][0] = s[kt][1] = dp[kt][0] = dp[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int row = kc + 16 * kt + c16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(row_chunk(Kl, row, 4 * ks + g));
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(row_chunk(Vl, row, 4 * ks + g));
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq) {
                        s[kt][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[jq][ks], s[kt][jq], 0, 0, 0);
                        dp[kt][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[jq][ks], dp[kt][jq], 0, 0, 0);
                    }
                }
            }
            // dS^T = scale * P^T * (dP^T - delta[q]), P^T = exp2(S^T*scale*log2e - lse*log2e)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int jq = 0; jq < 2; ++jq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[kt][jq][r], sl2, -lq[jq]));
                        float dpv = dp[kt][jq][r];
                        if (dc.thr) dpv = drop_keep_rc(dc, rowh_s[r0 + 16 * jq + c16], kc + 16 * kt + 4 * g + r) ? dpv * dc.scale : 0.0f;
                        s[kt][jq][r] = scale * pv * (dpv - dq_delta[jq]);
                    }
            // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 dsf[2];
#pragma unroll
                for (int jq = 0; jq < 2; ++jq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dsf[jq][r] = (__bf16)s[2 * kb][jq][r];
                        dsf[jq][4 + r] = (__bf16)s[2 * kb + 1][jq][r];
                    }
                const int rowA = kc + 32 * kb + 4 * g + qp, rowB = rowA + 16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag kt_;
                    kt_.h[0] = TR(row_chunk(Kl, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    kt_.h[1] = TR(row_chunk(Kl, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_.v, dsf[0], dq[dt][0], 0, 0, 0);
                    dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_.v, dsf[1], dq[dt][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            __bf16* drow = dqkv + ((size_t)b * T + r0 + 16 * jq + c16) * ld_dqkv + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 w;
                w[0] = pack_bf16x2(dq[dt][jq][0], dq[dt][jq][1]);
                w[1] = pack_bf16x2(dq[dt][jq][2], dq[dt][jq][3]);
                *reinterpret_cast<u32x2*>(drow + 16 * dt + 4 * g) = w;
            }
        }

        // =========================== pass 2: dK, dV for keys r0 .. r0+31 ===========================
        bf16x8 kf[2][2], vf[2][2];
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
            const int k = r0 + 16 * jk + c16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[jk][ks] = *reinterpret_cast<const bf16x8*>(row_chunk(Kl, k, 4 * ks + g));
                vf[jk][ks] = *reinterpret_cast<const bf16x8*>(row_chunk(Vl, k, 4 * ks + g));
            }
        }
        f32x4 dk[4][2], dv[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt][0] = dk[dt][1] = dv[dt][0] = dv[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int qc = 0; qc < T; qc += 64) {
            f32x4 s[4][2], dp[4][2];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt][0] = s[qt][1] = dp[qt][0] = dp[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int row = qc + 16 * qt + c16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(row_chunk(Ql, row, 4 * ks + g));
                    const bf16x8 da = *reinterpret_cast<const bf16x8*>(row_chunk(Dl, row, 4 * ks + g));
#pragma unroll
                    for (int jk = 0; jk < 2; ++jk) {
                        s[qt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[jk][ks], s[qt][jk], 0, 0, 0);
                        dp[qt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[jk][ks], dp[qt][jk], 0, 0, 0);
                    }
                }
            }
            // rows of these tiles are queries qc + 16qt + 4g + r
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const f32x4 lr = *reinterpret_cast<const f32x4*>(lse_s + qc + 16 * qt + 4 * g);
                const f32x4 dr = *reinterpret_cast<const f32x4*>(dlt_s + qc + 16 * qt + 4 * g);
#pragma unroll
                for (int jk = 0; jk < 2; ++jk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[qt][jk][r], sl2, -lr[r]));
                        float dpv = dp[qt][jk][r], pd = pv;
                        if (dc.thr) {
                            const bool keep = drop_keep_rc(dc, rowh_s[qc + 16 * qt + 4 * g + r], r0 + 16 * jk + c16);
                            dpv = keep ? dpv * dc.scale : 0.0f;
                            pd = keep ? pv * dc.scale : 0.0f;
                        }
                        s[qt][jk][r] = pd;                          // dropped P (for dV)
                        dp[qt][jk][r] = scale * pv * (dpv - dr[r]);  // dS
                    }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                bf16x8 pf[2], dsf[2];
#pragma unroll
                for (int jk = 0; jk < 2; ++jk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pf[jk][r] = (__bf16)s[2 * qb][jk][r];
                        pf[jk][4 + r] = (__bf16)s[2 * qb + 1][jk][r];
                        dsf[jk][r] = (__bf16)dp[2 * qb][jk][r];
                        dsf[jk][4 + r] = (__bf16)dp[2 * qb + 1][jk][r];
                    }
                const int rowA = qc + 32 * qb + 4 * g + qp, rowB = rowA + 16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag dot, qt_;
                    dot.h[0] = TR(row_chunk(Dl, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    dot.h[1] = TR(row_chunk(Dl, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    qt_.h[0] = TR(row_chunk(Ql, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    qt_.h[1] = TR(row_chunk(Ql, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
#pragma unroll
                    for (int jk = 0; jk < 2; ++jk) {
                        dv[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot.v, pf[jk], dv[dt][jk], 0, 0, 0);
                        dk[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_.v, dsf[jk], dk[dt][jk], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
            __bf16* drow = dqkv + ((size_t)b * T + r0 + 16 * jk + c16) * ld_dqkv + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 wk_, wv_;
                wk_[0] = pack_bf16x2(dk[dt][jk][0], dk[dt][jk][1]);
                wk_[1] = pack_bf16x2(dk[dt][jk][2], dk[dt][jk][3]);
                wv_[0] = pack_bf16x2(dv[dt][jk][0], dv[dt][jk][1]);
                wv_[1] = pack_bf16x2(dv[dt][jk][2], dv[dt][jk][3]);
                *reinterpret_cast<u32x2*>(drow + heads * DH + 16 * dt + 4 * g) = wk_;
                *reinterpret_cast<u32x2*>(drow + 2 * heads * DH + 16 * dt + 4 * g) = wv_;
            }
        }
    }
}
#undef TR

}  // namespace

int bsi_attention_bwd_drop(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                           int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, DropCfg dc, bsi_stream_t stream) {
    BSI_CHECK_ARG(qkv && out && dout && lse && dqkv && B > 0 && heads > 0, "bsi_attention_bwd: bad args");
    BSI_CHECK_ARG(dh == 64, "bsi_attention_bwd: head dim %d unsupported (64)", dh);
    BSI_CHECK_ARG(tokens > 0 && tokens % 64 == 0 && tokens <= 256, "bsi_attention_bwd: tokens=%d must be 64..256, multiple of 64", tokens);
    BSI_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_dqkv % 4 == 0, "bsi_attention_bwd: bad leading dimensions");
    const size_t lds = (size_t)4 * tokens * RB + 3 * tokens * sizeof(float);
    set_max_lds(reinterpret_cast<const void*>(attention_bwd_kernel), 4 * 256 * RB + 3 * 256 * 4);
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * heads), dim3(512), lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const __bf16*>(qkv), ld_qkv, reinterpret_cast<const __bf16*>(out),
                       reinterpret_cast<const __bf16*>(dout), ld_o, lse, tokens, heads, reinterpret_cast<__bf16*>(dqkv),
                       ld_dqkv, 1.0f / sqrtf((float)dh), dc);
    BSI_CHECK_LAUNCH("bsi_attention_bwd");
    return BSI_OK;
}

extern "C" int bsi_attention_bwd(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o,
                                 const float* lse, int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv,
                                 bsi_stream_t stream) {
    return bsi_attention_bwd_drop(qkv, ld_qkv, out, dout, ld_o, lse, B, tokens, heads, dh, dqkv, ld_dqkv, DropCfg{}, stream);
}
