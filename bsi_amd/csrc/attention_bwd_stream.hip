// Backward of the non-causal softmax attention for LONG sequences / wide heads (autograd of
// F.scaled_dot_product_attention at bsi/nn/attention.py:18,38 of the reference: the VDM-UNet centre attention,
// 1024 positions, one head of 128 channels): dQ, dK, dV from Q, K, V, O, dO and the saved log-sum-exp.
//
// Same two passes as attention_bwd.hip (which keeps all of Q, K, V, dO of a 256-token head in LDS), but the "other side"
// of each pass is STREAMED through LDS in 64-row chunks, so the sequence length is unbounded:
//   pass 1: a wave owns 16*JR queries (fragments in registers), K and V chunks stream:  S^T = K.Q^T, dP^T = V.dO^T,
//           dS^T = scale * P^T * (dP^T - delta), dQ^T += K^T . dS^T;
//   pass 2: a wave owns 16*JR keys, Q and dO chunks stream:  S = Q.K^T, dP = dO.V^T, dV^T += dO^T . P, dK^T += Q^T . dS.
// No atomics, deterministic.  One LDS image per tile serves row reads (ds_read_b128) and transposed reads
// (ds_read_b64_tr_b16): 16-B chunk index ^= swz(row).
#include "common.h"
#include "dit_ops.h"

namespace {

template <int DH>
__device__ __forceinline__ int swz(int r) {
    if constexpr (DH == 64) return ((r >> 1) & 3) << 1;
    else return (r & 7) << 1;
}
template <int DH>
__device__ __forceinline__ const char* rc(const char* tile, int r, int c) {
    return tile + r * (DH * 2) + ((c ^ swz<DH>(r)) << 4);
}

#define TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

template <int DH>
__global__ __launch_bounds__(512) void attention_bwd_stream_kernel(const __bf16* __restrict__ qkv, int ld_qkv,
                                                                   const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                                   int ld_o, const float* __restrict__ lse, int T, int heads,
                                                                   __bf16* __restrict__ dqkv, int ld_dqkv, float scale) {
    constexpr int RB = DH * 2, CPR = DH / 8, KS = DH / 32, DT = DH / 16;
    constexpr int JR = DH == 64 ? 2 : 1;      // 16-row tiles owned by a wave
    constexpr int LPT = 2 * 64 * CPR / 512;   // 16-B loads per thread per streamed chunk (two tiles)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* TA = lds;
    char* TB = TA + 64 * RB;
    float* lse_s = reinterpret_cast<float*>(TB + 64 * RB);  // [T] lse * log2(e)
    float* dlt_s = lse_s + T;                               // [T] delta = rowsum(dO * O)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int g = lane >> 4, c16 = lane & 15, qp = c16 >> 2, pp = c16 & 3;
    const float L2E = 1.4426950408889634f;
    const float sl2 = scale * L2E;

    const __bf16* Qg = qkv + (size_t)b * T * ld_qkv + h * DH;
    const __bf16* Kg = Qg + heads * DH;
    const __bf16* Vg = Qg + 2 * heads * DH;
    const __bf16* Og = o + (size_t)b * T * ld_o + h * DH;
    const __bf16* Dg = dout + (size_t)b * T * ld_o + h * DH;

    for (int idx = tid; idx < T * 2; idx += 512) {
        const int r = idx >> 1, hlf = idx & 1;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CPR / 2; ++c) {
            const u32x4 dv = *reinterpret_cast<const u32x4*>(Dg + (size_t)r * ld_o + hlf * (DH / 2) + c * 8);
            const u32x4 ov = *reinterpret_cast<const u32x4*>(Og + (size_t)r * ld_o + hlf * (DH / 2) + c * 8);
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
        }
    }

    u32x4 pre[LPT];
    auto prefetch = [&](const __bf16* A, int lda, const __bf16* Bp, int ldb, int row0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + i * 512;
            const int tile = idx / (64 * CPR), rem = idx % (64 * CPR);
            const int r = rem / CPR, c = rem % CPR;
            const __bf16* src = tile == 0 ? A + (size_t)(row0 + r) * lda + c * 8 : Bp + (size_t)(row0 + r) * ldb + c * 8;
            pre[i] = *reinterpret_cast<const u32x4*>(src);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + i * 512;
            const int tile = idx / (64 * CPR), rem = idx % (64 * CPR);
            const int r = rem / CPR, c = rem % CPR;
            *reinterpret_cast<u32x4*>((tile == 0 ? TA : TB) + r * RB + ((c ^ swz<DH>(r)) << 4)) = pre[i];
        }
    };

    union Frag { bf16x8 v; s16x4 h[2]; };
    const int r0 = (blockIdx.x * 8 + wave) * 16 * JR;  // this wave's rows (queries in pass 1, keys in pass 2)
    const bool active = r0 < T;
    const int r0c = active ? r0 : 0;

    // =========================== pass 1: dQ ===========================
    {
        bf16x8 qf[JR][KS], dof[JR][KS];
        float lq[JR], dq_delta[JR];
#pragma unroll
        for (int jq = 0; jq < JR; ++jq) {
            const int q = r0c + 16 * jq + c16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qf[jq][ks] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)q * ld_qkv + 32 * ks + 8 * g);
                dof[jq][ks] = *reinterpret_cast<const bf16x8*>(Dg + (size_t)q * ld_o + 32 * ks + 8 * g);
            }
        }
        f32x4 dq[DT][JR];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int jq = 0; jq < JR; ++jq) dq[dt][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
        prefetch(Kg, ld_qkv, Vg, ld_qkv, 0);
        for (int kc = 0; kc < T; kc += 64) {
            __syncthreads();
            commit();
            __syncthreads();
            if (kc == 0) {
#pragma unroll
                for (int jq = 0; jq < JR; ++jq) { lq[jq] = lse_s[r0c + 16 * jq + c16]; dq_delta[jq] = dlt_s[r0c + 16 * jq + c16]; }
            }
            if (kc + 64 < T) prefetch(Kg, ld_qkv, Vg, ld_qkv, kc + 64);
            f32x4 s[4][JR], dp[4][JR];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int jq = 0; jq < JR; ++jq) s[kt][jq] = dp[kt][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int row = 16 * kt + c16;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rc<DH>(TA, row, 4 * ks + g));
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(rc<DH>(TB, row, 4 * ks + g));
#pragma unroll
                    for (int jq = 0; jq < JR; ++jq) {
                        s[kt][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[jq][ks], s[kt][jq], 0, 0, 0);
                        dp[kt][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[jq][ks], dp[kt][jq], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int jq = 0; jq < JR; ++jq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[kt][jq][r], sl2, -lq[jq]));
                        s[kt][jq][r] = scale * pv * (dp[kt][jq][r] - dq_delta[jq]);
                    }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 dsf[JR];
#pragma unroll
                for (int jq = 0; jq < JR; ++jq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dsf[jq][r] = (__bf16)s[2 * kb][jq][r];
                        dsf[jq][4 + r] = (__bf16)s[2 * kb + 1][jq][r];
                    }
                const int rowA = 32 * kb + 4 * g + qp, rowB = rowA + 16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    Frag kt_;
                    kt_.h[0] = TR(rc<DH>(TA, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    kt_.h[1] = TR(rc<DH>(TA, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
#pragma unroll
                    for (int jq = 0; jq < JR; ++jq)
                        dq[dt][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_.v, dsf[jq], dq[dt][jq], 0, 0, 0);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int jq = 0; jq < JR; ++jq) {
                __bf16* drow = dqkv + ((size_t)b * T + r0 + 16 * jq + c16) * ld_dqkv + h * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    u32x2 w;
                    w[0] = pack_bf16x2(dq[dt][jq][0], dq[dt][jq][1]);
                    w[1] = pack_bf16x2(dq[dt][jq][2], dq[dt][jq][3]);
                    *reinterpret_cast<u32x2*>(drow + 16 * dt + 4 * g) = w;
                }
            }
        }
    }

    // =========================== pass 2: dK, dV ===========================
    {
        bf16x8 kf[JR][KS], vf[JR][KS];
#pragma unroll
        for (int jk = 0; jk < JR; ++jk) {
            const int k = r0c + 16 * jk + c16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[jk][ks] = *reinterpret_cast<const bf16x8*>(Kg + (size_t)k * ld_qkv + 32 * ks + 8 * g);
                vf[jk][ks] = *reinterpret_cast<const bf16x8*>(Vg + (size_t)k * ld_qkv + 32 * ks + 8 * g);
            }
        }
        f32x4 dk[DT][JR], dv[DT][JR];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int jk = 0; jk < JR; ++jk) dk[dt][jk] = dv[dt][jk] = f32x4{0.f, 0.f, 0.f, 0.f};
        prefetch(Qg, ld_qkv, Dg, ld_o, 0);
        for (int qc = 0; qc < T; qc += 64) {
            __syncthreads();
            commit();
            __syncthreads();
            if (qc + 64 < T) prefetch(Qg, ld_qkv, Dg, ld_o, qc + 64);
            f32x4 s[4][JR], dp[4][JR];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
                for (int jk = 0; jk < JR; ++jk) s[qt][jk] = dp[qt][jk] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int row = 16 * qt + c16;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(rc<DH>(TA, row, 4 * ks + g));
                    const bf16x8 da = *reinterpret_cast<const bf16x8*>(rc<DH>(TB, row, 4 * ks + g));
#pragma unroll
                    for (int jk = 0; jk < JR; ++jk) {
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
                for (int jk = 0; jk < JR; ++jk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[qt][jk][r], sl2, -lr[r]));
                        s[qt][jk][r] = pv;                                        // P (for dV)
                        dp[qt][jk][r] = scale * pv * (dp[qt][jk][r] - dr[r]);     // dS
                    }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                bf16x8 pf[JR], dsf[JR];
#pragma unroll
                for (int jk = 0; jk < JR; ++jk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pf[jk][r] = (__bf16)s[2 * qb][jk][r];
                        pf[jk][4 + r] = (__bf16)s[2 * qb + 1][jk][r];
                        dsf[jk][r] = (__bf16)dp[2 * qb][jk][r];
                        dsf[jk][4 + r] = (__bf16)dp[2 * qb + 1][jk][r];
                    }
                const int rowA = 32 * qb + 4 * g + qp, rowB = rowA + 16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    Frag dot, qt_;
                    dot.h[0] = TR(rc<DH>(TB, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    dot.h[1] = TR(rc<DH>(TB, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    qt_.h[0] = TR(rc<DH>(TA, rowA, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
                    qt_.h[1] = TR(rc<DH>(TA, rowB, 2 * dt + (pp >> 1)) + 8 * (pp & 1));
#pragma unroll
                    for (int jk = 0; jk < JR; ++jk) {
                        dv[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot.v, pf[jk], dv[dt][jk], 0, 0, 0);
                        dk[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_.v, dsf[jk], dk[dt][jk], 0, 0, 0);
                    }
                }
            }
        }
        if (active) {
#pragma unroll
            for (int jk = 0; jk < JR; ++jk) {
                __bf16* drow = dqkv + ((size_t)b * T + r0 + 16 * jk + c16) * ld_dqkv + h * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
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
}
#undef TR

template <int DH>
int launch_stream(const __bf16* qkv, int ld_qkv, const __bf16* o, const __bf16* dout, int ld_o, const float* lse, int B, int T,
                  int heads, __bf16* dqkv, int ld_dqkv, hipStream_t s) {
    constexpr int JR = DH == 64 ? 2 : 1;
    const size_t lds = (size_t)2 * 64 * DH * 2 + (size_t)2 * T * sizeof(float);
    auto kern = attention_bwd_stream_kernel<DH>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((T + 128 * JR - 1) / (128 * JR), B * heads);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, qkv, ld_qkv, o, dout, ld_o, lse, T, heads, dqkv, ld_dqkv,
                       1.0f / sqrtf((float)DH));
    BSI_CHECK_LAUNCH("bsi_attention_bwd_long");
    return BSI_OK;
}

}  // namespace

extern "C" int bsi_attention_bwd_long(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                                      int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, bsi_stream_t stream) {
    BSI_CHECK_ARG(qkv && out && dout && lse && dqkv && B > 0 && heads > 0, "bsi_attention_bwd_long: bad args");
    BSI_CHECK_ARG(dh == 64 || dh == 128, "bsi_attention_bwd_long: head dim %d unsupported (64 or 128)", dh);
    BSI_CHECK_ARG(tokens > 0 && tokens % 64 == 0 && tokens <= 8192, "bsi_attention_bwd_long: tokens=%d must be a multiple of 64, <= 8192",
                  tokens);
    BSI_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_dqkv % 4 == 0, "bsi_attention_bwd_long: bad leading dimensions");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const __bf16* q = reinterpret_cast<const __bf16*>(qkv);
    const __bf16* o = reinterpret_cast<const __bf16*>(out);
    const __bf16* d = reinterpret_cast<const __bf16*>(dout);
    __bf16* dq = reinterpret_cast<__bf16*>(dqkv);
    if (dh == 64) return launch_stream<64>(q, ld_qkv, o, d, ld_o, lse, B, tokens, heads, dq, ld_dqkv, s);
    return launch_stream<128>(q, ld_qkv, o, d, ld_o, lse, B, tokens, heads, dq, ld_dqkv, s);
}
