// Persistent, software-pipelined softmax attention for the DiT geometry (256 tokens, head dim 64; replaces
// F.scaled_dot_product_attention at bsi/models/dit.py:43-44).  The chunked kernel of attention.hip runs its loads, its
// products and its stores one after the other inside short-lived workgroups: measured (tools/experiments/attn_lab.hip,
// 256 images x 16 heads) 116 us for its memory traffic alone and 80 us for everything else, 130-150 us together.  Here one
// workgroup per CU walks the (batch, head) pairs and the traffic of pair i+1 runs under the arithmetic of pair i:
//   * 16 waves x 16 queries (four waves per SIMD: one wave's softmax VALU runs beside the others' MFMAs), <= 128 VGPRs;
//   * K and V of the NEXT pair go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write) into the
//     other half of a 2 x 64 KB ring, its Q fragments into registers by plain loads, all issued at the START of the current
//     pair; ONE barrier per pair, behind a counted vmcnt that leaves only the previous pair's output stores in flight;
//   * the LDS images are the ones the chunk loop of attention.hip reads (K rows XOR-swizzled by 16-B chunk for ds_read_b128,
//     V rows by 32-B block for ds_read_b64_tr_b16); an LDS-DMA writes lane-linearly, so the swizzle is applied to the SOURCE
//     address of each lane;
//   * scores transposed (S^T = K.Q^T), softmax over two 128-key chunks (one online rescale per pair), P^T packed to bf16 = B
//     operand of O^T = V^T.P^T; every LDS address = a per-lane register + an immediate (no vector-ALU address arithmetic);
//   * the 16 x 64 output tile of a wave leaves through 2 KB of LDS as whole 128-B lines (2 store instructions per wave);
//   * training dropout (dit.py:43-44, `dropout_p` of SDPA) arrives as MASK WORDS made by attn_dropmask_kernel below: per
//     (pair, 16-query block qb, 16-key block kt, r) one 64-bit word whose bit 16 g + c is keep(query 16 qb + c, key 16 kt + 4 g + r)
//     -- exactly this kernel's (and the backward's first pass's) lane <-> (query, key) map, so a word IS the select mask of
//     v_cndmask (a scalar register pair): one vector instruction per probability instead of the ~10 the counter hash took inside
//     the softmax (round 3: 240 -> 373 us with dropout).  A wave's 512 B of words for the NEXT pair are fetched by LDS-DMA into
//     its output scratch once the epilogue has read it back.  The survivors' scale 1 / (1 - p) multiplies the output row once.
#include <type_traits>

#include "common.h"
#include "dit_ops.h"

namespace {

constexpr int PT = 256, PDH = 64, PRB = 128;
constexpr int P_BUF = 2 * PT * PRB;          // K + V of one pair: 64 KB
constexpr int P_SCR = 2 * P_BUF;             // output scratch: 16 waves x 2 KB

__device__ __forceinline__ float pgroup_max(float v) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

template <bool DROP, bool LSE>
__global__ __launch_bounds__(1024) void attention_fwd_p_kernel(const __bf16* __restrict__ qkv, int ld_qkv, int pairs, int heads,
                                                               __bf16* __restrict__ out, int ld_out, float scale_log2e,
                                                               float* __restrict__ lse, DropCfg dc, const char* __restrict__ maskw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int q0 = wave * 16;

    // ---- LDS-DMA plan: 64 instructions of 1 KB per pair (K rows 0..255, then V rows 0..255), 4 per wave
    const int tile = wave >> 3;                      // waves 0-7 fetch K, waves 8-15 fetch V
    const int dma_dst = tile * (PT * PRB) + ((4 * wave) & 31) * 1024;   // + t * 1024 + buf * P_BUF
    auto pair_base = [&](int pr) { return reinterpret_cast<const char*>(qkv) + ((size_t)(pr / heads) * PT * ld_qkv + (size_t)(pr % heads) * PDH) * 2; };
    // The per-lane source offsets are RECOMPUTED at every issue (a dozen vector instructions per pair) from a lane id the compiler
    // cannot see through: as loop invariants they were four 64-bit address pairs live through the whole pair, and the kernel sits at
    // its 128-register cap -- the training instances spilled exactly those, and every reload (scratch load + s_waitcnt vmcnt(0) right
    // behind an LDS-DMA issue) waited for that DMA to land.
    auto issue_kv = [&](const char* base, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 8 * ((4 * wave + t) & 31) + (ln >> 3), p = ln & 7;
            const int c = tile == 0 ? (p ^ ((row >> 1) & 7)) : ((((p >> 1) ^ ((row >> 1) & 3)) << 1) | (p & 1));
            const unsigned so = (unsigned)row * (unsigned)(ld_qkv * 2) + (unsigned)c * 16u + (unsigned)((tile + 1) * heads * PDH * 2);
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + so), LDS_PTR(lds + buf * P_BUF + dma_dst + t * 1024), 16, 0, 0);
        }
    };
    // Q fragments of the NEXT pair: plain loads the compiler must not wait for (its own waits would drain the output stores too)
    const unsigned qoff = (unsigned)(q0 + c16) * (unsigned)(ld_qkv * 2) + (unsigned)g * 16u;
    u32x4 qn0, qn1;
    auto issue_q = [&](const char* base) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qn0) : "v"(base + qoff) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(qn1) : "v"(base + qoff) : "memory");
    };

    char* scr = lds + P_SCR + wave * 2048;
    auto issue_mask = [&](int p_) {  // this wave's 64 mask words (512 B: lanes 0..31) of pair p_ into the head of its scratch
        if constexpr (DROP) {
            int ln = lane;
            asm volatile("" : "+v"(ln));  // recomputed per pair, like the K / V offsets (as a loop invariant this address was spilled)
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((p_ * 16 + wave) * 512);  // scalar: pairs * 8 KB < 4 GB (launcher)
            if (ln < 32) __builtin_amdgcn_global_load_lds(GLB_PTR(maskw + so + ln * 16), LDS_PTR(scr), 16, 0, 0);
        }
    };
    int pr = blockIdx.x;
    if (pr >= pairs) return;
    {
        const char* base = pair_base(pr);
        issue_mask(pr);
        issue_kv(base, 0);
        issue_q(base);
    }
    asm volatile("s_waitcnt vmcnt(0) ; data of %0 %1" : "+v"(qn0), "+v"(qn1)::"memory");
    int buf = 0;
    while (true) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // every wave's share of this pair's K/V has landed; the previous pair is finished
        __builtin_amdgcn_sched_barrier(0);
        union { u32x4 u; bf16x8 v; } qa, qb;
        qa.u = qn0;
        qb.u = qn1;
        const bf16x8 qf0 = qa.v, qf1 = qb.v;
        const int next = pr + gridDim.x;
        const bool has_next = next < pairs;
        if (has_next) {
            const char* base = pair_base(next);
            issue_kv(base, buf ^ 1);
            issue_q(base);
        }
        const int b = pr / heads, h = pr % heads;
        const char* Kl = lds + buf * P_BUF;
        const char* Vl = Kl + PT * PRB;

        // Two chunks of 128 keys (32 scores per lane: the online rescale runs once per pair), fully unrolled: every LDS address
        // is a per-lane register that does not depend on the chunk or the pair (the swizzle keys depend on the lane only) plus
        // an immediate, so the loop body has no address arithmetic on the vector ALU.
        const char* kp0 = Kl + c16 * PRB + ((g ^ ((c16 >> 1) & 7)) << 4);            // + 2048 kt + 16384 chunk
        const char* kp1 = Kl + c16 * PRB + (((4 + g) ^ ((c16 >> 1) & 7)) << 4);
        const int qp = c16 >> 2, pp = c16 & 3;
        const int vkey = (2 * g + (qp >> 1)) & 3;                                     // ((row >> 1) & 3) of rows 4g + qp (+16, +32, ...)
        const unsigned vbase = lds_addr(Vl) + (4 * g + qp) * PRB + pp * 8;
        const unsigned va0 = vbase + ((0 ^ vkey) << 5), va1 = vbase + ((1 ^ vkey) << 5), va2 = vbase + ((2 ^ vkey) << 5),
                       va3 = vbase + ((3 ^ vkey) << 5);                               // + 4096 kb + 2048 (second 16 keys) + 16384 chunk
        f32x4 o[4];
        float m_run = 0.f, l_run = 0.f;
        // (DROP: the mask words of this pair were issued at the end of the previous pair, in front of the 6 loads above)
        if constexpr (DROP) {
            if (has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // last pair: nothing younger was issued behind the words
        }
        auto chunk = [&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            // ---- S^T = K . Q^T : rows = keys, cols = queries
            f32x4 s[8];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kp0 + ch * 16384 + kt * 2048);
                const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kp1 + ch * 16384 + kt * 2048);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf1, s[kt], 0, 0, 0);
            }
            // ---- softmax of the query column this lane holds (32 of the chunk's 128 keys per lane)
            // v_max3_f32 by hand: fmaxf() on MFMA outputs makes hipcc canonicalise every operand first (one extra v_max per value).
            // hipcc pads nothing for an asm statement: the wait states between an MFMA and a vector instruction that reads its result
            // (8-pass MFMA: 11) are ours to insert.  Left to the scheduler the (non-volatile) v_max3 statements sat 3-5 instructions
            // behind the MFMAs that feed them in the training instance -- a stale maximum, exp2 overflow, NaN on real activations
            // while random inputs passed.  All MFMAs of the chunk first, 16 wait states, then the chain.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            float mx = max3(s[0][0], s[0][1], s[0][2]);
            mx = max3(mx, s[0][3], s[1][0]);
#pragma unroll
            for (int kt = 1; kt < 8; ++kt) {
                if (kt > 1) mx = max3(mx, s[kt][0], s[kt - 1][3]);
                mx = max3(mx, s[kt][1], s[kt][2]);
            }
            mx = fmaxf(mx, s[7][3]);
            mx = pgroup_max(mx);
            float m_new = mx;
            if constexpr (ch > 0) {
                m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
            }
            m_run = m_new;
            const float mb = m_new * scale_log2e;
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                u32x4 mwa = u32x4{0u, 0u, 0u, 0u}, mwb = mwa;
                if constexpr (DROP) {  // the four words (r = 0..3) of key block (ch, kt): a wave-uniform address, broadcast
                    mwa = *reinterpret_cast<const u32x4*>(scr + (ch * 32 + kt * 4) * 8);
                    mwb = *reinterpret_cast<const u32x4*>(scr + (ch * 32 + kt * 4) * 8 + 16);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __builtin_amdgcn_exp2f(__fmaf_rn(s[kt][r], scale_log2e, -mb));
                    if (r & 1) ps1 += pv; else ps0 += pv;  // the normaliser uses the undropped probabilities
                    if constexpr (DROP) {
                        const unsigned lo = r < 2 ? mwa[2 * r] : mwb[2 * r - 4], hi = r < 2 ? mwa[2 * r + 1] : mwb[2 * r - 3];
                        const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(hi) << 32) |
                                                     (unsigned)__builtin_amdgcn_readfirstlane(lo);
                        pv = __builtin_amdgcn_inverse_ballot_w64(w) ? pv : 0.0f;  // 1 / (1 - p) multiplies the output row (epilogue)
                    }
                    s[kt][r] = pv;
                }
                // one key block at a time: left alone, the probabilities sink towards their use in the PV section and every block's
                // words (64 scalar registers) stay live until then.  The empty statement consumes the four results here.
                if constexpr (DROP) {
                    asm volatile("" : "+v"(s[kt][0]), "+v"(s[kt][1]), "+v"(s[kt][2]), "+v"(s[kt][3]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            l_run += ps0 + ps1;
            // the mask words are ordinary LDS loads; the transposed reads below are inline asm behind hand-counted lgkmcnt waits: keep
            // the compiler from moving a load of the next chunk's words in between them
            if constexpr (DROP) __builtin_amdgcn_sched_barrier(0);
            // ---- O^T += V^T . P^T over 32-key blocks.  The transposed reads go out as inline asm: behind the builtin hipcc waits
            //      vmcnt(0) -- for the NEXT pair's LDS-DMA, which it cannot tell apart from this buffer -- and the pipeline is gone.
            s16x4 va[2][4], vb[2][4];
            auto reads = [&](auto KB, int slot) {
                constexpr int off = ch * 16384 + decltype(KB)::value * 4096;
                va[slot][0] = tr_read<off>(va0); vb[slot][0] = tr_read<off + 2048>(va0);
                va[slot][1] = tr_read<off>(va1); vb[slot][1] = tr_read<off + 2048>(va1);
                va[slot][2] = tr_read<off>(va2); vb[slot][2] = tr_read<off + 2048>(va2);
                va[slot][3] = tr_read<off>(va3); vb[slot][3] = tr_read<off + 2048>(va3);
            };
            auto pv_block = [&](int kb, int slot, bool more_in_flight) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (__bf16)s[2 * kb][r];
                    pf[4 + r] = (__bf16)s[2 * kb + 1][r];
                }
                // in-order LDS returns: this block's 8 reads are complete once at most the 8 younger ones are outstanding
                if (more_in_flight) asm volatile("s_waitcnt lgkmcnt(8) ; data of %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(va[slot][0]), "+v"(va[slot][1]), "+v"(va[slot][2]), "+v"(va[slot][3]), "+v"(vb[slot][0]), "+v"(vb[slot][1]), "+v"(vb[slot][2]), "+v"(vb[slot][3]));
                else asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7" : "+v"(va[slot][0]), "+v"(va[slot][1]), "+v"(va[slot][2]), "+v"(va[slot][3]), "+v"(vb[slot][0]), "+v"(vb[slot][1]), "+v"(vb[slot][2]), "+v"(vb[slot][3]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    union { bf16x8 v; s16x4 hh[2]; } vf;
                    vf.hh[0] = va[slot][dt];
                    vf.hh[1] = vb[slot][dt];
                    if (ch == 0 && kb == 0) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf, o[dt], 0, 0, 0);
                }
            };
            reads(std::integral_constant<int, 0>{}, 0);
            reads(std::integral_constant<int, 1>{}, 1);
            pv_block(0, 0, true);
            reads(std::integral_constant<int, 2>{}, 0);
            pv_block(1, 1, true);
            reads(std::integral_constant<int, 3>{}, 1);
            pv_block(2, 0, true);
            pv_block(3, 1, false);
            if constexpr (DROP) __builtin_amdgcn_sched_barrier(0);
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        // ---- epilogue: normalise, 16 x 64 tile through this wave's 2 KB of LDS, whole 128-B lines to HBM
        {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            float l = l_run;
            u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(t[0]) + __uint_as_float(t[1]);
            t = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(t[0]) + __uint_as_float(t[1]);
            const float inv = DROP ? dc.scale / l : 1.0f / l;
            if constexpr (LSE) {  // log-sum-exp of the scaled scores (natural log), saved for the backward pass
                if (g == 0) lse[(size_t)pr * PT + q0 + c16] = (m_run * scale_log2e + __log2f(l)) * 0.6931471805599453f;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 w;
                w[0] = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                w[1] = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                // rows c and c + 8 of a 16-lane group meet in the same 16-B chunk (same c & 7): they take opposite 8-B halves of it
                // (2-way bank conflict on every ds_write_b64 otherwise: 1.05 M LDS cycles per launch in the round-3 counters)
                *reinterpret_cast<u32x2*>(scr + c16 * 128 + (((2 * dt + (g >> 1)) ^ (c16 & 7)) << 4) + (((g & 1) ^ (c16 >> 3)) << 3)) = w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 8 * i + rr;
                u32x4 v = *reinterpret_cast<const u32x4*>(scr + row * 128 + ((ch ^ (row & 7)) << 4));
                if (i == 1) v = u32x4{v[2], v[3], v[0], v[1]};  // rows 8..15 hold their chunk's halves swapped (see the writes)
                *reinterpret_cast<u32x4*>(out + ((size_t)b * PT + q0 + row) * ld_out + h * PDH + ch * 8) = v;
            }
        }
        if (!has_next) break;
        pr = next;
        buf ^= 1;
        issue_mask(pr);  // the scratch has been read back (the stores above hold its data): the next pair's mask words may land in it
        // the next pair's K/V (4 LDS-DMA) and Q fragments (2 loads) are older than this pair's output stores: leave the stores (and
        // the mask-word DMA just issued) in flight
        constexpr int INFLIGHT = 2 + (LSE ? 1 : 0) + (DROP ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(%2) ; data of %0 %1" : "+v"(qn0), "+v"(qn1) : "n"(INFLIGHT) : "memory");
    }
}

// Dropout-mask words of one attention call (layout: header; generator: drop_mask_block, common.h): one wave per block of 64 words.
__global__ __launch_bounds__(256) void attn_dropmask_kernel(DropCfg dc, int blocks16, unsigned long long* __restrict__ maskw) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);  // = pair * 16 + qb
    if (wid >= blocks16) return;
    maskw[(size_t)wid * 64 + lane] = drop_mask_block(dc, (unsigned)wid, lane);
}

template <bool DROP, bool LSE>
int launch_p(const __bf16* qkv, int ld_qkv, int B, int heads, __bf16* out, int ld_out, float* lse, DropCfg dc, const void* maskw, hipStream_t s) {
    auto kern = attention_fwd_p_kernel<DROP, LSE>;
    constexpr int lds = P_SCR + 16 * 2048;  // 160 KB
    set_max_lds(reinterpret_cast<const void*>(kern), lds);
    const int pairs = B * heads, ncu = compute_cus();
    const int grid = pairs < ncu ? pairs : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, s, qkv, ld_qkv, pairs, heads, out, ld_out,
                       1.4426950408889634f / sqrtf((float)PDH), lse, dc, reinterpret_cast<const char*>(maskw));
    BSI_CHECK_LAUNCH("bsi_attention_fwd(persistent)");
    return BSI_OK;
}

}  // namespace

// tokens == 256, dh == 64 (checked by the caller in attention.hip)
// The attention layer's dropout mask in the form the persistent kernels consume (test hook and stand-alone use): `pairs` (batch, head)
// pairs of 256 x 256 weights -> [pairs][16][64] 64-bit words (layout: header); bit-for-bit the mask bsi_dropout_mask exports for the
// same (p, seed, site) with rows = pairs * 256, cols = 256.
extern "C" int bsi_attention_dropout_words(float p, unsigned long long seed, unsigned site, int pairs, void* words, bsi_stream_t stream) {
    BSI_CHECK_ARG(words && pairs > 0 && p > 0.f && p < 1.f, "bsi_attention_dropout_words: bad args");
    const int blocks16 = pairs * 16;
    hipLaunchKernelGGL(attn_dropmask_kernel, dim3((blocks16 + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), make_drop(p, seed, site),
                       blocks16, reinterpret_cast<unsigned long long*>(words));
    BSI_CHECK_LAUNCH("bsi_attention_dropout_words");
    return BSI_OK;
}

// maskw (dropout only): [B * heads][16][64] 64-bit words, FILLED HERE (attn_dropmask_kernel) unless mask_ready, consumed by this launch and by
// bsi_attention_bwd_drop.  Dropout without a word buffer is not this kernel's business (the caller takes the chunked kernel).
int bsi_attention_fwd_persistent(const void* qkv, int ld_qkv, int B, int heads, void* out, int ld_out, float* lse, DropCfg dc,
                                 void* maskw, bool mask_ready, hipStream_t s) {
    const __bf16* q = reinterpret_cast<const __bf16*>(qkv);
    __bf16* o = reinterpret_cast<__bf16*>(out);
    if (dc.thr) {
        BSI_CHECK_ARG(maskw != nullptr, "bsi_attention_fwd(persistent): dropout needs the mask-word buffer");
        BSI_CHECK_ARG((size_t)B * heads * 8192 < (1ull << 31), "bsi_attention_fwd(persistent): %d pairs exceed the 32-bit mask-word offsets", B * heads);
        if (!mask_ready) {
            const int blocks16 = B * heads * 16;
            hipLaunchKernelGGL(attn_dropmask_kernel, dim3((blocks16 + 3) / 4), dim3(256), 0, s, dc, blocks16, reinterpret_cast<unsigned long long*>(maskw));
            BSI_CHECK_LAUNCH("bsi_attention_fwd(dropout mask words)");
        }
        return lse ? launch_p<true, true>(q, ld_qkv, B, heads, o, ld_out, lse, dc, maskw, s)
                   : launch_p<true, false>(q, ld_qkv, B, heads, o, ld_out, lse, dc, maskw, s);
    }
    return lse ? launch_p<false, true>(q, ld_qkv, B, heads, o, ld_out, lse, dc, nullptr, s)
               : launch_p<false, false>(q, ld_qkv, B, heads, o, ld_out, lse, dc, nullptr, s);
}
