// Non-causal softmax attention for gfx950 (replaces F.scaled_dot_product_attention at
// bsi/models/dit.py:43-44 and bsi/nn/attention.py:18,38 of the reference).
//
// One workgroup = 8 waves = up to 256 queries of one (batch, head); each wave owns 32 queries.
// K and V of a KC-key chunk live in LDS (row-major, XOR-swizzled); scores are computed TRANSPOSED
// (S^T = K . Q^T) so that a lane holds one query column: the softmax row reduction is lane-local
// plus two shuffles, and the S^T accumulators are, after a bf16 pack, directly the B operand of
// O^T = V^T . P^T (no LDS round trip for P).  V^T fragments come from the row-major V tile through
// ds_read_b64_tr_b16.  Online softmax over chunks makes the kernel independent of the sequence length
// (DiT: 256 tokens = one chunk; UNet centre attention: 1024 positions, dh 128).
#include <cstdlib>

#include "common.h"
#include "dit_ops.h"

namespace {

template <int DH>
__device__ __forceinline__ int k_chunk_swz(int row, int chunk) {  // 16-B chunk index within a K row
    if constexpr (DH == 64) return chunk ^ ((row >> 1) & 7);
    else return chunk ^ (row & 15);
}
template <int DH>
__device__ __forceinline__ int v_block_swz(int row, int blk) {  // 32-B block index within a V row
    if constexpr (DH == 64) return blk ^ ((row >> 1) & 3);
    else return blk ^ (row & 7);
}

// max / sum over the four 16-lane groups of a wave (lanes c, c+16, c+32, c+48) without LDS: the gfx950 lane-group swaps
// (v_permlane16_swap: odd groups of a <-> even groups of b; v_permlane32_swap: upper half of a <-> lower half of b) applied
// to two copies of the value leave "mine" in one result and "the partner group's" in the other.
__device__ __forceinline__ float lane_group_max(float v) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float lane_group_sum(float v) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// ALL: the whole K and V of the (batch, head) are staged ONCE (tokens * DH * 4 bytes of LDS: 64 KB for the DiT's 256 tokens x 64),
// the chunk loop then runs without barriers or global loads; two such workgroups share a CU, so one's staging round trip is
// covered by the other's chunk loop.  Without ALL (long sequences: the UNet's 1024 positions x 128) every chunk is staged.
// ABL (laboratory, tools/experiments/attn_lab.hip): 1 = no exp2 (plain copy of the argument), 2 = no global stores, 4 = no P.V
// products / V reads, 8 = no K.Q^T products / K reads, 16 = no K/V global loads (LDS holds garbage), 32 = no Q loads.
template <int DH, int KC, bool DROP, bool ALL, int ABL = 0>
__global__ __launch_bounds__(512) void attention_fwd_kernel(const __bf16* __restrict__ qkv, int ld_qkv, int tokens,
                                                            int heads, __bf16* __restrict__ out, int ld_out,
                                                            float scale_log2e, float* __restrict__ lse, DropCfg dc) {
    constexpr int RB = DH * 2;        // bytes per K/V row
    constexpr int KS = DH / 32;       // k-steps of the QK^T contraction
    constexpr int KT = KC / 16;       // 16-key tiles per chunk
    constexpr int DT = DH / 16;       // 16-wide d tiles of the output
    constexpr int CPR = DH / 8;       // 16-B chunks per row
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Kl = lds;
    char* Vl = lds + KC * RB;  // (ALL: set below to lds + tokens * RB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int q0 = blockIdx.x * 256 + wave * 32;
    const bool active = q0 < tokens;
    const int g = lane >> 4, c16 = lane & 15;

    const __bf16* base = qkv + (size_t)b * tokens * ld_qkv + h * DH;
    const __bf16* Qg = base;
    const __bf16* Kg = base + heads * DH;
    const __bf16* Vg = base + 2 * heads * DH;

    // Q fragments (B operand): lane holds Q[q0 + 16jq + c16][32ks + 8g .. +7]
    bf16x8 qf[2][KS];
    if (active) {
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (ABL & 32) { for (int e = 0; e < 8; ++e) qf[jq][ks][e] = (__bf16)(0.01f * (float)(lane + e)); }
                else qf[jq][ks] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)(q0 + 16 * jq + c16) * ld_qkv + 32 * ks + 8 * g);
            }
    }

    f32x4 o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) o[dt][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};
    unsigned rowh[2] = {0u, 0u};
    if constexpr (DROP) {
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) rowh[jq] = drop_row(dc, (unsigned)bh * tokens + q0 + 16 * jq + c16);
    }

    if constexpr (ALL) {
        Vl = lds + tokens * RB;
        for (int idx = tid; idx < ((ABL & 16) ? 0 : tokens * CPR); idx += 512) {
            const int r = idx / CPR, c = idx % CPR;
            const u32x4 kv = *reinterpret_cast<const u32x4*>(Kg + (size_t)r * ld_qkv + c * 8);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(Vg + (size_t)r * ld_qkv + c * 8);
            *reinterpret_cast<u32x4*>(Kl + r * RB + k_chunk_swz<DH>(r, c) * 16) = kv;
            *reinterpret_cast<u32x4*>(Vl + r * RB + (v_block_swz<DH>(r, c >> 1) * 2 + (c & 1)) * 16) = vv;
        }
        __syncthreads();
    }
    for (int kc0 = 0; kc0 < tokens; kc0 += KC) {
        if constexpr (!ALL) {
            __syncthreads();  // previous chunk fully consumed
            // ---- stage K and V chunk: KC rows x CPR chunks each, 512 threads ---------------------------
            for (int idx = tid; idx < KC * CPR; idx += 512) {
                const int r = idx / CPR, c = idx % CPR;
                const u32x4 kv = *reinterpret_cast<const u32x4*>(Kg + (size_t)(kc0 + r) * ld_qkv + c * 8);
                const u32x4 vv = *reinterpret_cast<const u32x4*>(Vg + (size_t)(kc0 + r) * ld_qkv + c * 8);
                *reinterpret_cast<u32x4*>(Kl + r * RB + k_chunk_swz<DH>(r, c) * 16) = kv;
                *reinterpret_cast<u32x4*>(Vl + r * RB + (v_block_swz<DH>(r, c >> 1) * 2 + (c & 1)) * 16) = vv;
            }
            __syncthreads();
        }
        const char* Kc = ALL ? Kl + kc0 * RB : Kl;   // this chunk's rows (the swizzle keys depend on row & 15 only)
        const char* Vc = ALL ? Vl + kc0 * RB : Vl;
        if (!active) continue;

        // ---- S^T = K . Q^T : rows = keys, cols = queries --------------------------------------------
        f32x4 s[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = 16 * kt + c16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (ABL & 8) {
                    s[kt][0][ks] += (float)qf[0][ks][kt & 7];
                    s[kt][1][ks + 2] += (float)qf[1][ks][kt & 7];
                    continue;
                }
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kc + row * RB + k_chunk_swz<DH>(row, 4 * ks + g) * 16);
                s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
                s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
            }
        }

        // ---- online softmax (per query column) -------------------------------------------------------
        float alpha[2];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][jq][r]);
            mx = lane_group_max(mx);
            const float m_new = fmaxf(m_run[jq], mx);
            alpha[jq] = __builtin_amdgcn_exp2f((m_run[jq] - m_new) * scale_log2e);
            m_run[jq] = m_new;
            const float mb = m_new * scale_log2e;
            f32x2 ps2 = {0.f, 0.f};  // two partial sums on the packed fp32 pipe
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 arg = __builtin_elementwise_fma(f32x2{s[kt][jq][r], s[kt][jq][r + 1]}, f32x2{scale_log2e, scale_log2e},
                                                                f32x2{-mb, -mb});
                    f32x2 pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                    if constexpr (ABL & 1) pv = arg * arg;
                    ps2 += pv;  // the normaliser uses the undropped probabilities
                    if constexpr (DROP) {  // dropout on the attention weights (dit.py:43-44 dropout_p), training only:
                        // element (row = (b, h, query), column = key)
                        pv[0] = drop_keep_rc(dc, rowh[jq], kc0 + 16 * kt + 4 * g + r) ? pv[0] * dc.scale : 0.0f;
                        pv[1] = drop_keep_rc(dc, rowh[jq], kc0 + 16 * kt + 4 * g + r + 1) ? pv[1] * dc.scale : 0.0f;
                    }
                    s[kt][jq][r] = pv[0];
                    s[kt][jq][r + 1] = pv[1];
                }
            l_run[jq] = l_run[jq] * alpha[jq] + (ps2[0] + ps2[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][jq][r] *= alpha[jq];
        }

        // ---- O^T += V^T . P^T  over 32-key blocks ---------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < KC / 32; ++kb) {
            bf16x8 pf[2];
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[jq][r] = (__bf16)s[2 * kb][jq][r];
                    pf[jq][4 + r] = (__bf16)s[2 * kb + 1][jq][r];
                }
            }
            // transposed V reads: in-group lane i = 4q'+p supplies &V[key0 + q'][16dt + 4p]
            const int qp = c16 >> 2, pp = c16 & 3;
            const int rowA = 32 * kb + 4 * g + qp;
            const int rowB = rowA + 16;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if constexpr (ABL & 4) {
                    o[dt][0][kb & 3] += (float)pf[0][dt];
                    o[dt][1][kb & 3] += (float)pf[1][dt + 4];
                    continue;
                }
                const int blk = dt >> 1;                      // 32-B block holding columns 16dt..16dt+15
                const int inblk = (dt & 1) * 32 + pp * 8;     // byte offset inside the 64-B block pair... see below
                // a 32-B block holds 16 bf16 columns: block index = dt (16 cols * 2 B = 32 B)
                (void)blk; (void)inblk;
                const int offA = rowA * RB + v_block_swz<DH>(rowA, dt) * 32 + pp * 8;
                const int offB = rowB * RB + v_block_swz<DH>(rowB, dt) * 32 + pp * 8;
                const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Vc + offA));
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Vc + offB));
                union { bf16x8 v; s16x4 h[2]; } vf;
                vf.h[0] = va;
                vf.h[1] = vb;
                o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[0], o[dt][0], 0, 0, 0);
                o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[1], o[dt][1], 0, 0, 0);
            }
        }
    }

    if constexpr (ALL && DH == 64) {
        // Output through LDS: the wave's 32 x 64 tile (4 KB) is written into its slice of the K region (free once every wave has
        // left the chunk loop) with the 16-B chunk index XOR-ed by row & 7, read back by rows and stored as whole 128-B lines
        // (8 lanes x 16 B per row, 8 rows per instruction): 4 store instructions per wave instead of 8 that each touch 16 rows
        // with 32 B (partial-line writes: measured as the kernel's bound, TA busy 84 %).
        __syncthreads();
        char* scr = lds + wave * 4096;
        if (active) {
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                float l = l_run[jq];
                l = lane_group_sum(l);
                const float inv = 1.0f / l;
                if (lse && g == 0) lse[(size_t)bh * tokens + q0 + 16 * jq + c16] = (m_run[jq] * scale_log2e + __log2f(l)) * 0.6931471805599453f;
                const int row = 16 * jq + c16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    u32x2 w;
                    w[0] = pack_bf16x2(o[dt][jq][0] * inv, o[dt][jq][1] * inv);
                    w[1] = pack_bf16x2(o[dt][jq][2] * inv, o[dt][jq][3] * inv);
                    *reinterpret_cast<u32x2*>(scr + row * 128 + (((2 * dt + (g >> 1)) ^ (row & 7)) << 4) + (g & 1) * 8) = w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row * 128 + ((ch ^ (row & 7)) << 4));
                if constexpr (ABL & 2) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) out[lane] = (__bf16)1.0f; continue; }
                *reinterpret_cast<u32x4*>(out + ((size_t)b * tokens + q0 + row) * ld_out + h * DH + ch * 8) = v;
            }
        }
        return;
    }
    if (!active) return;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        float l = l_run[jq];
        l = lane_group_sum(l);
        const float inv = 1.0f / l;
        // log-sum-exp of the scaled scores (natural log), saved for the backward pass
        if (lse && g == 0) lse[(size_t)bh * tokens + q0 + 16 * jq + c16] = (m_run[jq] * scale_log2e + __log2f(l)) * 0.6931471805599453f;
        __bf16* orow = out + ((size_t)b * tokens + q0 + 16 * jq + c16) * ld_out + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2 w;
            w[0] = pack_bf16x2(o[dt][jq][0] * inv, o[dt][jq][1] * inv);
            w[1] = pack_bf16x2(o[dt][jq][2] * inv, o[dt][jq][3] * inv);
            *reinterpret_cast<u32x2*>(orow + 16 * dt + 4 * g) = w;
        }
    }
}

template <int DH, int KC, bool ALL = false>
int launch_attn(const __bf16* qkv, int ld_qkv, int B, int tokens, int heads, __bf16* out, int ld_out, float* lse,
                DropCfg dc, hipStream_t s) {
    const size_t lds = 2 * (size_t)(ALL ? tokens : KC) * DH * 2;
    auto kern = attention_fwd_kernel<DH, KC, false, ALL>;
    auto kern_d = attention_fwd_kernel<DH, KC, true, ALL>;
    const int max_lds = ALL ? 2 * 256 * DH * 2 : (int)lds;
    set_max_lds(reinterpret_cast<const void*>(kern), max_lds);
    set_max_lds(reinterpret_cast<const void*>(kern_d), max_lds);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    dim3 grid((tokens + 255) / 256, B * heads);
    if (dc.thr) hipLaunchKernelGGL(kern_d, grid, dim3(512), lds, s, qkv, ld_qkv, tokens, heads, out, ld_out, scale_log2e, lse, dc);
    else hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, qkv, ld_qkv, tokens, heads, out, ld_out, scale_log2e, lse, dc);
    BSI_CHECK_LAUNCH("bsi_attention_fwd");
    return BSI_OK;
}

}  // namespace

int bsi_attention_fwd_persistent(const void* qkv, int ld_qkv, int B, int heads, void* out, int ld_out, float* lse, DropCfg dc,
                                 void* maskw, bool mask_ready, hipStream_t s);

// 256 tokens x head dim 64 with dropout on: the persistent kernels exchange the dropout mask as 64-bit words (8 KB per
// (batch, head) pair, attention_persist.hip) -- the forward writes them when `maskw` is given, the backward reads them
bool bsi_attention_uses_mask_words(int tokens, int dh) {
    static const bool off = getenv("BSI_ATTN_CHUNKED") != nullptr || getenv("BSI_ATTN_NO_PERSIST") != nullptr ||
                            getenv("BSI_ATTN_BWD_RESIDENT") != nullptr || getenv("BSI_ATTN_NO_MASKW") != nullptr;
    return tokens == 256 && dh == 64 && !off;
}

static int attention_fwd_impl(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out,
                              float* lse, DropCfg dc, bsi_stream_t stream, void* maskw = nullptr, bool mask_ready = false) {
    BSI_CHECK_ARG(qkv && out && B > 0 && heads > 0, "bsi_attention_fwd: bad args");
    BSI_CHECK_ARG(dh == 64 || dh == 128, "bsi_attention_fwd: head dim %d unsupported (64 or 128)", dh);
    BSI_CHECK_ARG(tokens > 0 && tokens % 64 == 0, "bsi_attention_fwd: tokens=%d must be a multiple of 64", tokens);
    BSI_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= 3 * heads * dh && ld_out % 8 == 0 && ld_out >= heads * dh,
                  "bsi_attention_fwd: bad leading dimensions");
    const __bf16* q = reinterpret_cast<const __bf16*>(qkv);
    __bf16* o = reinterpret_cast<__bf16*>(out);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dh == 64) {
        // 64-key chunks with online softmax: 117 VGPRs, so two workgroups share a CU.  Up to 256 tokens the whole K and V of the
        // (batch, head) are staged once (64 KB) and the chunk loop has no barrier; longer sequences stage chunk by chunk.
        static const bool chunked = getenv("BSI_ATTN_CHUNKED") != nullptr;  // A/B switches for experiments
        static const bool no_persist = getenv("BSI_ATTN_NO_PERSIST") != nullptr;
        // the DiT geometry proper (256 tokens): persistent kernel, next pair's traffic under this pair's arithmetic
        // (with dropout it needs the mask-word buffer of the training engine; without one the chunked kernel evaluates the hash itself)
        if (tokens == 256 && !no_persist && !chunked && (size_t)tokens * ld_qkv * 2 < (1ull << 32) && (!dc.thr || maskw))
            return bsi_attention_fwd_persistent(qkv, ld_qkv, B, heads, out, ld_out, lse, dc, maskw, mask_ready, s);
        if (tokens <= 256 && !chunked) return launch_attn<64, 64, true>(q, ld_qkv, B, tokens, heads, o, ld_out, lse, dc, s);
        return launch_attn<64, 64>(q, ld_qkv, B, tokens, heads, o, ld_out, lse, dc, s);
    }
    if (tokens % 128 == 0) return launch_attn<128, 128>(q, ld_qkv, B, tokens, heads, o, ld_out, lse, dc, s);
    return launch_attn<128, 64>(q, ld_qkv, B, tokens, heads, o, ld_out, lse, dc, s);
}

extern "C" int bsi_attention_fwd(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out,
                                 int ld_out, bsi_stream_t stream) {
    return attention_fwd_impl(qkv, ld_qkv, B, tokens, heads, dh, out, ld_out, nullptr, DropCfg{}, stream);
}

extern "C" int bsi_attention_fwd_lse(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out,
                                     int ld_out, float* lse, bsi_stream_t stream) {
    BSI_CHECK_ARG(lse, "bsi_attention_fwd_lse: null lse");
    return attention_fwd_impl(qkv, ld_qkv, B, tokens, heads, dh, out, ld_out, lse, DropCfg{}, stream);
}

int bsi_attention_fwd_train(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out,
                            float* lse, DropCfg dc, bsi_stream_t stream, void* maskw, bool mask_ready) {
    return attention_fwd_impl(qkv, ld_qkv, B, tokens, heads, dh, out, ld_out, lse, dc, stream,
                              bsi_attention_uses_mask_words(tokens, dh) ? maskw : nullptr, mask_ready);
}

// Training attention with the counter-based dropout of the DiT blocks as entry points of their own (the engine calls the two
// functions above): words = NULL, or the 8 KB per (image, head) of lane-mask words (256 tokens, head dim 64) that the forward writes and
// the backward reads -- what bsi_dit_train_forward / bsi_dit_backward keep on their tape.
extern "C" int bsi_attention_fwd_dropout(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out, float* lse,
                                         float p, unsigned long long seed, unsigned site, void* words, bsi_stream_t stream) {
    BSI_CHECK_ARG(p >= 0.f && p < 1.f, "bsi_attention_fwd_dropout: dropout probability %g outside [0, 1)", (double)p);
    BSI_CHECK_ARG(lse, "bsi_attention_fwd_dropout: the log-sum-exp output is required (the backward reads it)");
    return bsi_attention_fwd_train(qkv, ld_qkv, B, tokens, heads, dh, out, ld_out, lse, make_drop(p, seed, site), stream, words, false);
}
extern "C" int bsi_attention_bwd_dropout(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse, int B,
                                         int tokens, int heads, int dh, void* dqkv, int ld_dqkv, float p, unsigned long long seed, unsigned site,
                                         const void* words, bsi_stream_t stream) {
    BSI_CHECK_ARG(p >= 0.f && p < 1.f, "bsi_attention_bwd_dropout: dropout probability %g outside [0, 1)", (double)p);
    return bsi_attention_bwd_drop(qkv, ld_qkv, out, dout, ld_o, lse, B, tokens, heads, dh, dqkv, ld_dqkv, make_drop(p, seed, site), stream, words);
}
