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
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent form for the DiT geometry (256 tokens, head dim 64).  The resident kernel above stages its four tiles synchronously
// (128 KB per workgroup through registers, one 131-KB workgroup per CU: 46 % of its wave time in waits).  Here one workgroup
// per CU walks the (batch, head) pairs and the staging of the NEXT tiles runs under the arithmetic of the current pass:
//   pass 1 (dQ) reads the K / V tiles from LDS and this wave's own Q / dO rows from registers: while it runs, the pair's Q and
//           dO tiles are fetched by LDS-DMA (no register round trip) into the image pass 2 reads;
//   pass 2 (dK, dV) reads the Q / dO tiles from LDS and this wave's own K / V rows from registers (read from LDS at the end of
//           pass 1): while it runs, the NEXT pair's K and V tiles are fetched by LDS-DMA, and the next pair's own Q / dO / O rows
//           and log-sum-exp values by asynchronous loads into registers.
// Two barriers per pair; every wait is a counted vmcnt that leaves the pass's output stores in flight.  The issue stream never
// ends (past the last pair it re-fetches that pair), so every count is an immediate.  delta = rowsum(dO * O) is formed in
// registers by the wave that owns the rows.  Same two 8-wave passes, tile layout and arithmetic as the resident kernel (two
// waves per SIMD at <= 256 registers: the 16-wave variant of round 2 lost instruction-level parallelism).
constexpr int PT = 256;
constexpr int QL = 0, DL = PT * RB, KL = 2 * PT * RB, VL = 3 * PT * RB, STATS = 4 * PT * RB;  // Q and dO first: 16-bit ds offsets

__device__ __forceinline__ unsigned lds_u32(const char* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
template <int OFF>
__device__ __forceinline__ s16x4 tr_rd(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ float group_sum(float v) {  // sum over the four 16-lane groups (same c16)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// DROP: 0 = no dropout; 1 = the mask is evaluated here from the counter hash (drop_keep_rc); 2 = the mask arrives as the 64-bit
// words the persistent forward kernel wrote (attention_persist.hip: word (16-query block qb, 16-key block kt, r), bit 16 g + c =
// keep(query 16 qb + c, key 16 kt + 4 g + r); 8 KB per pair, fetched by LDS-DMA beside the K / V tiles, double buffered).  Pass 1
// has exactly the forward's lane <-> (query, key) map, so a word is the select mask of v_cndmask as it stands (scalar register
// pair); pass 2 (a lane = one key, four queries) reads the 16-bit group of its key and tests four bits.  The survivors' scale
// multiplies dP inside the fused multiply-add of dS and the dV tile once at the end.
constexpr int MASKW = STATS + 3 * PT * 4, MASKW_BYTES = 8192;

template <int DROP>
__global__ __launch_bounds__(512) void attention_bwd_p_kernel(const __bf16* __restrict__ qkv, int ld_qkv, const __bf16* __restrict__ o,
                                                              const __bf16* __restrict__ dout, int ld_o, const float* __restrict__ lse,
                                                              int pairs, int heads, __bf16* __restrict__ dqkv, int ld_dqkv, float scale,
                                                              DropCfg dc, const char* __restrict__ maskw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lse_s = reinterpret_cast<float*>(lds + STATS);
    float* dlt_s = lse_s + PT;
    unsigned* rowh_s = reinterpret_cast<unsigned*>(dlt_s + PT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15, qp = c16 >> 2, pp = c16 & 3;
    const float L2E = 1.4426950408889634f;
    const float sl2 = scale * L2E;
    const int r0 = wave * 32;

    // ---- LDS-DMA plan: a tile is 32 instructions of 1 KB (8 rows x 8 chunks, lane-linear in LDS, the chunk swizzle applied to the
    //      SOURCE address); 8 per wave and pass: waves 0-3 fetch the first tile of the pass (Q / K), waves 4-7 the second (dO / V)
    const int second = wave >> 2;
    const int drow = 8 * ((wave & 3) * 8) + (lane >> 3);                    // + 8 t
    const unsigned dchunk = (unsigned)(((lane & 7) ^ sw(drow)) << 4);        // sw(drow + 8 t) = sw(drow)
    const int ld1 = second ? ld_o : ld_qkv;                                  // pass 1 fetches Q | dO, pass 2 fetches K | V
    const unsigned off1 = (unsigned)drow * (unsigned)(ld1 * 2) + dchunk, str1 = 8u * (unsigned)(ld1 * 2);
    const unsigned off2 = (unsigned)drow * (unsigned)(ld_qkv * 2) + dchunk, str2 = 8u * (unsigned)(ld_qkv * 2);
    char* dst1 = lds + (second ? DL : QL) + (wave & 3) * 8 * 1024;
    char* dst2 = lds + (second ? VL : KL) + (wave & 3) * 8 * 1024;
    auto q_base = [&](int pr) { return reinterpret_cast<const char*>(qkv) + ((size_t)(pr / heads) * PT * ld_qkv + (size_t)(pr % heads) * DH) * 2; };
    auto o_off = [&](int pr) { return ((size_t)(pr / heads) * PT * ld_o + (size_t)(pr % heads) * DH) * 2; };
    auto issue_qd = [&](int pr) {  // Q and dO tiles of pair pr
        const char* src = second ? reinterpret_cast<const char*>(dout) + o_off(pr) : q_base(pr);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + off1 + t * str1), LDS_PTR(dst1 + t * 1024), 16, 0, 0);
    };
    auto issue_kv = [&](int pr) {  // K and V tiles of pair pr
        const char* src = q_base(pr) + (size_t)(second ? 2 : 1) * heads * DH * 2;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + off2 + t * str2), LDS_PTR(dst2 + t * 1024), 16, 0, 0);
    };
    auto issue_mask = [&](int pr, int mbuf) {  // DROP == 2: the pair's 8 KB of mask words, 1 KB per wave (always in front of group A: no count changes)
        if constexpr (DROP == 2)
            __builtin_amdgcn_global_load_lds(GLB_PTR(maskw + (size_t)pr * MASKW_BYTES + wave * 1024 + lane * 16),
                                             LDS_PTR(lds + MASKW + mbuf * MASKW_BYTES + wave * 1024), 16, 0, 0);
    };
    // ---- this wave's own rows of the NEXT pair, as MFMA fragments straight from memory (inline asm: the compiler must not wait for them)
    u32x4 qn[2][2], dn[2][2], on[2][2];
    float lsn[2];
    const unsigned fq = (unsigned)(r0 + c16) * (unsigned)(ld_qkv * 2) + (unsigned)g * 16u, fq16 = 16u * (unsigned)(ld_qkv * 2);
    const unsigned fo = (unsigned)(r0 + c16) * (unsigned)(ld_o * 2) + (unsigned)g * 16u, fo16 = 16u * (unsigned)(ld_o * 2);
    auto issue_group_a = [&](int pr) {  // 10 loads: Q and dO fragments, log-sum-exp
        const char* qb = q_base(pr) + fq;
        const char* db = reinterpret_cast<const char*>(dout) + o_off(pr) + fo;
        const float* lp = lse + (size_t)pr * PT + r0 + c16;
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qn[jq][0]) : "v"(qb + jq * fq16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(qn[jq][1]) : "v"(qb + jq * fq16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dn[jq][0]) : "v"(db + jq * fo16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(dn[jq][1]) : "v"(db + jq * fo16) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(lsn[jq]) : "v"(lp + 16 * jq) : "memory");
        }
    };
    auto issue_group_b = [&](int pr) {  // 4 loads: O fragments (for delta)
        const char* ob = reinterpret_cast<const char*>(o) + o_off(pr) + fo;
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(on[jq][0]) : "v"(ob + jq * fo16) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(on[jq][1]) : "v"(ob + jq * fo16) : "memory");
        }
    };
#define WAIT_A(N_)                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N_ ") ; data of %0 %1 %2 %3 %4 %5 %6 %7 %8 %9"                                                   \
                 : "+v"(qn[0][0]), "+v"(qn[0][1]), "+v"(qn[1][0]), "+v"(qn[1][1]), "+v"(dn[0][0]), "+v"(dn[0][1]), "+v"(dn[1][0]),   \
                   "+v"(dn[1][1]), "+v"(lsn[0]), "+v"(lsn[1])::"memory")
#define WAIT_B(N_)                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N_ ") ; data of %0 %1 %2 %3" : "+v"(on[0][0]), "+v"(on[0][1]), "+v"(on[1][0]), "+v"(on[1][1])::"memory")
#define PBARRIER()                                   \
    do {                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);           \
        __builtin_amdgcn_s_barrier();                \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

    int pr = blockIdx.x;
    if (pr >= pairs) return;
    int mbuf = 0;
    issue_kv(pr);
    issue_mask(pr, 0);
    issue_group_a(pr);
    issue_group_b(pr);
    WAIT_A(0);
    WAIT_B(0);
    PBARRIER();

    // ---- per-lane LDS addresses of the transposed reads: row 4g + qp (+ 16, + 32 kb, + 64 chunk as immediates), 32-B block dt
    const int swh = (2 * g + (qp >> 1)) & 3;  // ((row >> 1) & 3) of rows 4g + qp + 16 n
    unsigned tq[4], tk[4];                    // Q tile (the dO tile is DL further: immediate), K tile
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        tq[dt] = lds_u32(lds) + (4 * g + qp) * RB + ((dt ^ swh) << 5) + (pp >> 1) * 16 + 8 * (pp & 1);
        tk[dt] = tq[dt] + KL;
    }
    union Frag { bf16x8 v; s16x4 h[2]; u32x4 u; };
    const char* Ql = lds + QL;
    const char* Dl = lds + DL;
    const char* Kl = lds + KL;
    const char* Vl = lds + VL;

    while (true) {
        const int nxt = pr + gridDim.x < pairs ? pr + gridDim.x : pr;  // past the end: re-fetch this pair (never read)
        const int b = pr / heads, h = pr % heads;
        // =========================== pass 1: dQ for queries r0 .. r0+31 ===========================
        issue_qd(pr);
        bf16x8 qf[2][2], dof[2][2];
        float lq[2], dq_delta[2];
        unsigned rh[2] = {0u, 0u};
        WAIT_B(24);  // younger than the O fragments: the previous pass's 16 dK / dV stores and the 8 DMA instructions just issued
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            float a = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag fq_, fd_;
                fq_.u = qn[jq][ks];
                fd_.u = dn[jq][ks];
                qf[jq][ks] = fq_.v;
                dof[jq][ks] = fd_.v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned dv = dn[jq][ks][e], ov = on[jq][ks][e];
                    a = __fmaf_rn(__uint_as_float(dv << 16), __uint_as_float(ov << 16), a);
                    a = __fmaf_rn(__uint_as_float(dv & 0xffff0000u), __uint_as_float(ov & 0xffff0000u), a);
                }
            }
            dq_delta[jq] = group_sum(a);
            lq[jq] = lsn[jq] * L2E;
            const int q = r0 + 16 * jq + c16;
            if constexpr (DROP == 1) rh[jq] = drop_row(dc, (unsigned)pr * PT + q);
            if (g == 0) {
                dlt_s[q] = dq_delta[jq];
                lse_s[q] = lq[jq];
                rowh_s[q] = rh[jq];
            }
        }
        f32x4 dq[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dq[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
        for (int kc = 0; kc < PT; kc += 64) {
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
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int jq = 0; jq < 2; ++jq) {
                    u32x4 mwa = u32x4{0u, 0u, 0u, 0u}, mwb = mwa;
                    if constexpr (DROP == 2) {  // the four words (r = 0..3) of (query block 2 wave + jq, key block kc / 16 + kt): wave-uniform address
                        const char* mp = lds + MASKW + mbuf * MASKW_BYTES + (((2 * wave + jq) * 16 + (kc >> 4) + kt) << 5);
                        mwa = *reinterpret_cast<const u32x4*>(mp);
                        mwb = *reinterpret_cast<const u32x4*>(mp + 16);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[kt][jq][r], sl2, -lq[jq]));
                        float dpv = dp[kt][jq][r];
                        if constexpr (DROP == 1) dpv = drop_keep_rc(dc, rh[jq], kc + 16 * kt + 4 * g + r) ? dpv * dc.scale : 0.0f;
                        if constexpr (DROP == 2) {
                            const unsigned lo = r < 2 ? mwa[2 * r] : mwb[2 * r - 4], hi = r < 2 ? mwa[2 * r + 1] : mwb[2 * r - 3];
                            const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(hi) << 32) |
                                                         (unsigned)__builtin_amdgcn_readfirstlane(lo);
                            dpv = __builtin_amdgcn_inverse_ballot_w64(w) ? dpv : 0.0f;
                            s[kt][jq][r] = scale * pv * __fmaf_rn(dpv, dc.scale, -dq_delta[jq]);
                        } else {
                            s[kt][jq][r] = scale * pv * (dpv - dq_delta[jq]);
                        }
                    }
                    if constexpr (DROP == 2) {  // one (key block, query block) at a time: otherwise every block's words (64 scalar registers) are live at once
                        asm volatile("" : "+v"(s[kt][jq][0]), "+v"(s[kt][jq][1]), "+v"(s[kt][jq][2]), "+v"(s[kt][jq][3]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]: the eight transposed reads of a 32-key block as inline asm (behind the
            // builtin hipcc drains vmcnt -- the Q / dO tiles in flight), then one wait that names them
            const unsigned kco = (unsigned)kc * RB;
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
                Frag kt_[4];
                if (kb == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) { kt_[dt].h[0] = tr_rd<0>(tk[dt] + kco); kt_[dt].h[1] = tr_rd<16 * RB>(tk[dt] + kco); }
                } else {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) { kt_[dt].h[0] = tr_rd<32 * RB>(tk[dt] + kco); kt_[dt].h[1] = tr_rd<48 * RB>(tk[dt] + kco); }
                }
                asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                             : "+v"(kt_[0].h[0]), "+v"(kt_[0].h[1]), "+v"(kt_[1].h[0]), "+v"(kt_[1].h[1]), "+v"(kt_[2].h[0]), "+v"(kt_[2].h[1]),
                               "+v"(kt_[3].h[0]), "+v"(kt_[3].h[1]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt].v, dsf[0], dq[dt][0], 0, 0, 0);
                    dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt].v, dsf[1], dq[dt][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {  // 8 stores
            __bf16* drow_ = dqkv + ((size_t)b * PT + r0 + 16 * jq + c16) * ld_dqkv + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 w;
                w[0] = pack_bf16x2(dq[dt][jq][0], dq[dt][jq][1]);
                w[1] = pack_bf16x2(dq[dt][jq][2], dq[dt][jq][3]);
                *reinterpret_cast<u32x2*>(drow_ + 16 * dt + 4 * g) = w;
            }
        }
        // this wave's own K / V rows for pass 2, while the tiles are still there
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
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this wave's share of the Q / dO tiles has landed (the 8 dQ stores may fly)
        PBARRIER();

        // =========================== pass 2: dK, dV for keys r0 .. r0+31 ===========================
        issue_kv(nxt);
        issue_mask(nxt, mbuf ^ 1);
        issue_group_a(nxt);
        f32x4 dk[4][2], dv[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt][0] = dk[dt][1] = dv[dt][0] = dv[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 32 queries per trip (the resident kernel takes 64): the prefetched fragments of the next pair live through this pass, and
        // with 64-query trips the allocation went past the 256 registers of a two-waves-per-SIMD kernel (12 spilled)
#pragma unroll 1
        for (int qc = 0; qc < PT; qc += 32) {
            f32x4 s[2][2], dp[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
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
            const unsigned qco = (unsigned)qc * RB;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const f32x4 lr = *reinterpret_cast<const f32x4*>(lse_s + qc + 16 * qt + 4 * g);
                const f32x4 dr = *reinterpret_cast<const f32x4*>(dlt_s + qc + 16 * qt + 4 * g);
                u32x4 rhq = u32x4{0u, 0u, 0u, 0u};
                if constexpr (DROP == 1) rhq = *reinterpret_cast<const u32x4*>(rowh_s + qc + 16 * qt + 4 * g);
#pragma unroll
                for (int jk = 0; jk < 2; ++jk) {
                    unsigned mbits = 0u;  // bit r: keep(query qc + 16 qt + 4 g + r, this lane's key r0 + 16 jk + c16)
                    if constexpr (DROP == 2) {
                        // word (query block qc / 16 + qt, key block 2 wave + jk, r = key & 3), its 16-bit group (key >> 2) & 3, bits 4 g ..
                        const char* mp = lds + MASKW + mbuf * MASKW_BYTES + ((((qc >> 4) + qt) * 16 + 2 * wave + jk) << 5) + (c16 & 3) * 8 + (c16 >> 2) * 2;
                        mbits = (unsigned)*reinterpret_cast<const unsigned short*>(mp) >> (4 * g);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[qt][jk][r], sl2, -lr[r]));
                        float dpv = dp[qt][jk][r], pd = pv;
                        if constexpr (DROP == 1) {
                            const bool keep = drop_keep_rc(dc, rhq[r], r0 + 16 * jk + c16);
                            dpv = keep ? dpv * dc.scale : 0.0f;
                            pd = keep ? pv * dc.scale : 0.0f;
                        }
                        if constexpr (DROP == 2) {
                            const bool keep = (mbits >> r) & 1u;
                            dpv = keep ? dpv : 0.0f;
                            pd = keep ? pv : 0.0f;       // 1 / (1 - p) multiplies the dV tile at the end
                            s[qt][jk][r] = pd;
                            dp[qt][jk][r] = scale * pv * __fmaf_rn(dpv, dc.scale, -dr[r]);
                        } else {
                            s[qt][jk][r] = pd;                          // dropped P (for dV)
                            dp[qt][jk][r] = scale * pv * (dpv - dr[r]);  // dS
                        }
                    }
                }
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int jk = 0; jk < 2; ++jk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[jk][r] = (__bf16)s[0][jk][r];
                    pf[jk][4 + r] = (__bf16)s[1][jk][r];
                    dsf[jk][r] = (__bf16)dp[0][jk][r];
                    dsf[jk][4 + r] = (__bf16)dp[1][jk][r];
                }
            // transposed Q / dO fragments as inline asm (behind the builtin hipcc drains vmcnt, i.e. the next pair's tiles in flight),
            // two 32-B blocks of the head dimension at a time: 8 reads, one wait that names them, 8 MFMAs
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                Frag dot[2], qt_[2];
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    const int dt = 2 * half + d2;
                    dot[d2].h[0] = tr_rd<DL>(tq[dt] + qco); dot[d2].h[1] = tr_rd<DL + 16 * RB>(tq[dt] + qco);
                    qt_[d2].h[0] = tr_rd<0>(tq[dt] + qco);  qt_[d2].h[1] = tr_rd<16 * RB>(tq[dt] + qco);
                }
                asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                             : "+v"(dot[0].h[0]), "+v"(dot[0].h[1]), "+v"(dot[1].h[0]), "+v"(dot[1].h[1]), "+v"(qt_[0].h[0]),
                               "+v"(qt_[0].h[1]), "+v"(qt_[1].h[0]), "+v"(qt_[1].h[1]));
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    const int dt = 2 * half + d2;
#pragma unroll
                    for (int jk = 0; jk < 2; ++jk) {
                        dv[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[d2].v, pf[jk], dv[dt][jk], 0, 0, 0);
                        dk[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_[d2].v, dsf[jk], dk[dt][jk], 0, 0, 0);
                    }
                }
            }
        }
        issue_group_b(nxt);  // late: their registers are needed only once s / dp are dead
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {  // 16 stores
            __bf16* drow_ = dqkv + ((size_t)b * PT + r0 + 16 * jk + c16) * ld_dqkv + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 wk_, wv_;
                wk_[0] = pack_bf16x2(dk[dt][jk][0], dk[dt][jk][1]);
                wk_[1] = pack_bf16x2(dk[dt][jk][2], dk[dt][jk][3]);
                const float vs = DROP == 2 ? dc.scale : 1.0f;
                wv_[0] = pack_bf16x2(dv[dt][jk][0] * vs, dv[dt][jk][1] * vs);
                wv_[1] = pack_bf16x2(dv[dt][jk][2] * vs, dv[dt][jk][3] * vs);
                *reinterpret_cast<u32x2*>(drow_ + heads * DH + 16 * dt + 4 * g) = wk_;
                *reinterpret_cast<u32x2*>(drow_ + 2 * heads * DH + 16 * dt + 4 * g) = wv_;
            }
        }
        mbuf ^= 1;
        WAIT_A(20);  // the next K / V tiles (8) and fragment group A (10) have landed; O fragments (4) and the 16 stores may fly
        PBARRIER();
        if (pr + (int)gridDim.x >= pairs) break;
        pr += gridDim.x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus fetches of the last pair land before the LDS is released
#undef WAIT_A
#undef WAIT_B
#undef PBARRIER
}

}  // namespace

static bool single_sweep(int tokens, int dh, DropCfg dc, const void* maskw) {
    static const bool resident_only = getenv("BSI_ATTN_BWD_RESIDENT") != nullptr;  // A/B partner of the persistent kernels
    static const bool two_pass = getenv("BSI_ATTN_BWD_TWO_PASS") != nullptr;        // A/B partner of the single-sweep kernel (round 4)
    return tokens == PT && dh == DH && !resident_only && !two_pass &&
           (!dc.thr || (maskw && bsi_attention_uses_mask_words(tokens, dh)));  // the hash form of the mask stays with the two-pass kernel
}
bool bsi_attention_bwd_emits_bias(int tokens, int dh, DropCfg dc, const void* maskw) { return single_sweep(tokens, dh, dc, maskw); }

int bsi_attention_bwd_drop(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                           int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, DropCfg dc, bsi_stream_t stream,
                           const void* maskw, float* bias_rows) {
    BSI_CHECK_ARG(qkv && out && dout && lse && dqkv && B > 0 && heads > 0, "bsi_attention_bwd: bad args");
    BSI_CHECK_ARG(dh == 64, "bsi_attention_bwd: head dim %d unsupported (64)", dh);
    BSI_CHECK_ARG(tokens > 0 && tokens % 64 == 0 && tokens <= 256, "bsi_attention_bwd: tokens=%d must be 64..256, multiple of 64", tokens);
    BSI_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_dqkv % 4 == 0, "bsi_attention_bwd: bad leading dimensions");
    static const bool resident_only = getenv("BSI_ATTN_BWD_RESIDENT") != nullptr;  // A/B partner of the persistent kernel
    BSI_CHECK_ARG(!bias_rows || single_sweep(tokens, dh, dc, maskw), "bsi_attention_bwd: bias rows are an output of the single-sweep kernel only");
    if (single_sweep(tokens, dh, dc, maskw))
        return bsi_attention_bwd_exchange(qkv, ld_qkv, out, dout, ld_o, lse, B, heads, dqkv, ld_dqkv, dc, maskw, reinterpret_cast<hipStream_t>(stream),
                                          bias_rows);
    if (tokens == PT && !resident_only) {
        const int pairs = B * heads, ncu = compute_cus();
        const int grid = pairs < ncu ? pairs : ncu;
        constexpr int plds = STATS + 3 * PT * 4;
        const float sc = 1.0f / sqrtf((float)dh);
        const bool words = dc.thr && maskw && bsi_attention_uses_mask_words(tokens, dh);
        auto go = [&](auto kern, int lds_bytes) {
            set_max_lds(reinterpret_cast<const void*>(kern), lds_bytes);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, reinterpret_cast<hipStream_t>(stream),
                               reinterpret_cast<const __bf16*>(qkv), ld_qkv, reinterpret_cast<const __bf16*>(out),
                               reinterpret_cast<const __bf16*>(dout), ld_o, lse, pairs, heads, reinterpret_cast<__bf16*>(dqkv), ld_dqkv, sc, dc,
                               reinterpret_cast<const char*>(maskw));
        };
        if (words) go(attention_bwd_p_kernel<2>, plds + 2 * MASKW_BYTES);
        else if (dc.thr) go(attention_bwd_p_kernel<1>, plds);
        else go(attention_bwd_p_kernel<0>, plds);
        BSI_CHECK_LAUNCH("bsi_attention_bwd(persistent)");
        return BSI_OK;
    }
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
