// Backward of the non-causal softmax attention for the DiT geometry (256 tokens, head dim 64), ONE sweep per (batch, head) pair
// (autograd of F.scaled_dot_product_attention at bsi/models/dit.py:43-44; round 4).
//
// The two-pass kernel (attention_bwd.hip) forms S, P and dS twice -- once with the query on the lane for dQ, once with the key on
// the lane for dK / dV -- because each wave can only sum over what it owns: 7 matrix products instead of 5 and, what costs more,
// the exponentials, the dropout selects and the bf16 packing of all 65536 score elements twice (the kernel is vector-ALU bound:
// 27-30 issue cycles per score element against 448 MFMAs per wave and pair).  Here a wave owns 32 KEYS and walks the queries
// 32 at a time (the second pass of that kernel, unchanged arithmetic): P^T and dS^T are the B operands of
// dV^T = dO^T . P and dK^T = Q^T . dS as they leave the lane, and the packed dS block ALSO goes to an LDS exchange buffer
// X[key][query] (8 waves x 32 keys x 32 queries of bf16 = 16 KB per trip, row pitch 96 B: the 8 rows an LDS cycle of a
// transposed read touches start in 8 disjoint 8-bank windows).  Behind the trip's barrier wave w forms ONE 16 x 16 tile of
// dQ^T = K^T . dS^T for those 32 queries -- head-dim block w & 3, query block w >> 2 -- over all 256 keys: eight MFMAs whose A
// operands (its K^T rows, 32 registers) were read once per pair and whose B operands are transposed reads of X.  No score element
// is formed twice, nothing is summed across waves, no atomics: deterministic.
//
// Staging: Q and dO tiles (32 KB each) are refilled ROLLING -- a trip's 32 rows are dead behind its barrier, so the NEXT pair's rows
// are fetched into them by LDS-DMA right there (one 1-KB instruction per wave and trip) -- and so are the trip's 1 KB of dropout
// words; the K tile is only read in the pair's prologue (own K rows + the K^T fragments) and refilled during trip 0; a wave's own V
// rows come straight from memory into registers (no V tile: its 32 KB hold the two X buffers).  delta = rowsum(dO * O) and the
// log-sum-exp of the next pair are formed four queries per wave and trip from 8-byte loads.  One barrier per trip (lock step) or one
// per half trip (SKEW, the default since round 5: see the kernel's comment); every wait is a counted vmcnt (the issue sequence per
// trip is fixed; past the last pair the stream re-fetches that pair).
#include <cstdlib>

#include "common.h"
#include "dit_ops.h"

namespace {

constexpr int DH = 64, RB = 128, PT = 256;
constexpr int XP = 96, XSZ = PT * XP;                 // exchange buffer: row pitch (64 B of data + 32 B of pad), bytes per buffer (24 KB)
constexpr int QL = 0, DL = PT * RB, KL = 2 * PT * RB; // Q and dO first: 16-bit ds offsets
constexpr int XL = 3 * PT * RB;                       // X[2]
constexpr int ST = XL + 2 * XSZ;                      // lse_s[2][PT], dlt_s[2][PT]
constexpr int MK = ST + 4 * PT * 4;                   // dropout words of the current pair (8 KB, refilled rolling)
constexpr int MK_BYTES = 8192;
constexpr int LDS_BYTES = MK + MK_BYTES;              // 159744
// the bias partials of a pair ([8 waves][192] floats: dQ 64 | dK 64 | dV 64) live in the 32-byte pads of X[0]'s rows (8 floats per
// row, 24 rows per wave): nothing else reads or writes those bytes, so they need no ordering against the exchange traffic
constexpr int BPW = 24 * XP;  // bytes between two waves' partials; float i of wave w: XL + w * BPW + (i >> 3) * XP + 64 + (i & 7) * 4
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
#ifndef ATTN_ABL
#define ATTN_ABL 0  // timing ablations of tools/experiments/attn_bwd_ablate.sh (wrong results): 1 softmax, 2 dV / dK, 4 dQ, 8 barrier, 16 S / dP, 32 next-pair fetches, 64 X write, 1024 dK / dV stores
#endif
constexpr int ABL = ATTN_ABL;
#ifdef ATTN_NO_BIAS
#define BIAS_ON false
#else
#define BIAS_ON (bias_rows != nullptr)
#endif

__device__ __forceinline__ int sw(int r) { return ((r >> 1) & 3) << 1; }
__device__ __forceinline__ const char* row_chunk(const char* tile, int r, int c) { return tile + r * RB + ((c ^ sw(r)) << 4); }
__device__ __forceinline__ unsigned lds_u32(const char* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
template <int OFF>
__device__ __forceinline__ s16x4 tr_rd(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

// DROP: 0 = no dropout, 2 = the mask arrives as the 64-bit words of attention_persist.hip (word (16-query block qb, 16-key block
// kt, r), bit 16 g + c = keep(query 16 qb + c, key 16 kt + 4 g + r)); a lane = one key, four queries reads the 16-bit group of its
// key and tests four bits.  (The hash form, DROP = 1, stays with the two-pass kernel.)
// SKEW (round 5): the two wave groups (waves 0-3 = A, 4-7 = B; a SIMD hosts one wave of each) run HALF A TRIP apart -- the persistent
// GEMM's ping-pong.  A trip is two slots with a barrier behind each: H1 = S / dP MFMAs, softmax, dS block to X[t & 1]; H2 = dV / dK
// MFMAs, a dQ^T tile, the fetches.  B executes one barrier more in front of its first H1 of every pair (and A one behind its pair
// epilogue), so B's H1(t) runs next to A's H2(t) and B's H2(t) next to A's H1(t + 1): vector and matrix phases of a SIMD's two waves
// alternate instead of coinciding.  X[t & 1] is whole when B's H1(t) ends, i.e. when A's H2(t) ends too: A forms the dQ^T tile of trip
// t - 1 in H2(t) (that of trip 7 in front of its pair epilogue), B that of trip t -- X stays two deep (A writes X[t + 1] in the slot
// after its read of X[t - 1], next to B's read of X[t]).  Rows and words of a trip are dead one slot later than without the skew, so
// H2(t) refills the slot of trip t - 1 (trip 0: slot 7, for THIS pair's trip 7).  The per-wave issue order is the unskewed one, the
// counted waits carry over (A's extra dQ store in the epilogue only makes two of them stricter).
template <int DROP, bool SKEW>
__global__ __launch_bounds__(512) void attention_bwd_x_kernel(const __bf16* __restrict__ qkv, int ld_qkv, const __bf16* __restrict__ o,
                                                              const __bf16* __restrict__ dout, int ld_o, const float* __restrict__ lse,
                                                              int pairs, int heads, __bf16* __restrict__ dqkv, int ld_dqkv, float scale,
                                                              DropCfg dc, const char* __restrict__ maskw, float* __restrict__ bias_rows) {
    // bias_rows (or null): [B][3 * heads * 64] fp32, row b = the column sums of image b's dqkv rows (bf16 values as stored) -- the
    // per-image slabs of the qkv bias gradient, added over the images by bsi_colsum_rows_f32 (the weight-gradient GEMM then runs
    // without its bias rider: 884 -> 788 us).  Per pair: a DPP row sum per accumulator register, 192 floats per wave through LDS.
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lse_s = reinterpret_cast<float*>(lds + ST);
    float* dlt_s = lse_s + 2 * PT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15, qp = c16 >> 2, pp = c16 & 3;
    const float L2E = 1.4426950408889634f;
    const float sl2 = scale * L2E;
    const int r0 = wave * 32;            // this wave's keys
    const int dtw = wave & 3, jqw = wave >> 2;  // its dQ^T tile of a trip: head-dim block, query block

    // ---- LDS-DMA plan (a 1-KB instruction = 8 rows x 8 chunks, lane-linear in LDS, the chunk swizzle applied to the SOURCE address).
    //      Every address is a wave-uniform 64-bit base (scalar unit) plus a 32-bit per-lane offset that is a loop invariant: one register
    //      each, instead of the ~50 registers of hoisted 64-bit addresses (or ~100 integer instructions per trip to redo them) that
    //      per-lane pointer arithmetic cost.
    const int second = wave >> 2, w4 = wave & 3;
    const int grp = second;  // SKEW: wave group (B = 1 runs half a trip late)
    const int drow8 = lane >> 3;
    const unsigned dchunk = (unsigned)(((lane & 7) ^ sw(drow8)) << 4);  // sw(8 k + drow8) = sw(drow8)
    const unsigned ldb1 = (unsigned)((second ? ld_o : ld_qkv) * 2), ldbq = (unsigned)(ld_qkv * 2), ldbo = (unsigned)(ld_o * 2);
    const unsigned qd_off = (unsigned)drow8 * ldb1 + dchunk;    // Q / dO rows of a DMA instruction
    const unsigned k_off = (unsigned)drow8 * ldbq + dchunk;     // K rows
    const unsigned st_off = (unsigned)g * ldbo + (unsigned)c16 * 8u;  // statistics: row g of four, 8 bytes of the head dimension
    const unsigned v_off = (unsigned)c16 * ldbq + (unsigned)g * 16u;   // V fragment rows
    const unsigned out_off = (unsigned)c16 * (unsigned)(ld_dqkv * 2) + (unsigned)g * 8u;  // dQ / dK / dV stores: row c16, 4 columns at 4 g
    auto q_base = [&](int pr) { return reinterpret_cast<const char*>(qkv) + ((size_t)(pr / heads) * PT * ld_qkv + (size_t)(pr % heads) * DH) * 2; };
    auto o_off = [&](int pr) { return ((size_t)(pr / heads) * PT * ld_o + (size_t)(pr % heads) * DH) * 2; };
    auto issue_qd = [&](int pr, int t) {  // rows 32 t .. 32 t + 31 of Q (waves 0-3) and dO (waves 4-7): 8 rows per wave
        const int row0 = 32 * t + 8 * w4;
        const char* src = (second ? reinterpret_cast<const char*>(dout) + o_off(pr) : q_base(pr)) + (size_t)row0 * ldb1;
        if constexpr (SKEW) asm volatile("" : "+s"(src));  // (pair, slot) vary per trip there: without the fence the sum becomes a per-lane 64-bit induction variable
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + qd_off), LDS_PTR(lds + (second ? DL : QL) + row0 * RB), 16, 0, 0);
    };
    auto issue_k = [&](int pr) {  // K tile: rows 32 wave .. + 31 (4 instructions)
        const char* src = q_base(pr) + (size_t)heads * DH * 2 + (size_t)(32 * wave) * ldbq;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const char* s4 = src + (size_t)(8 * t4) * ldbq;
            if constexpr (SKEW) asm volatile("" : "+s"(s4));  // (without the fence: four per-lane 64-bit addresses per pair, spilled in the skewed instances)
            __builtin_amdgcn_global_load_lds(GLB_PTR(s4 + k_off), LDS_PTR(lds + KL + (32 * wave + 8 * t4) * RB), 16, 0, 0);
        }
    };
    auto issue_mask = [&](int pr, int t) {  // DROP == 2: the trip's 1 KB of words, 128 B per wave (8 lanes)
        if constexpr (DROP == 2) {
            const char* mb = maskw + (size_t)pr * MK_BYTES + t * 1024 + wave * 128;  // wave-uniform base + a 32-bit lane offset, as everywhere
            if constexpr (SKEW) asm volatile("" : "+s"(mb));
            if (lane < 8) __builtin_amdgcn_global_load_lds(GLB_PTR(mb + (unsigned)lane * 16u), LDS_PTR(lds + MK + t * 1024 + wave * 128), 16, 0, 0);
        }
    };
    // ---- this wave's own V rows, as MFMA fragments straight from memory (inline asm: the compiler must not wait for them).  They are
    //      fetched INTO the working registers behind their last use of the pair (trip 7): a second set for the whole pair does not fit
    u32x4 vn[2][2];
    auto issue_v = [&](int pr) {  // 4 loads
        const char* vb = q_base(pr) + (size_t)2 * heads * DH * 2 + (size_t)r0 * ldbq;
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
            const char* vr = vb + (size_t)(16 * jk) * ldbq;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(vn[jk][0]) : "v"(v_off), "s"(vr) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(vn[jk][1]) : "v"(v_off), "s"(vr) : "memory");
        }
    };
    // ---- delta and log-sum-exp of four queries per wave and trip: query 32 wave + 4 t + g, 16 lanes x 4 head-dim elements
    //      Two register sets, used by the even and the odd trips: a set is consumed TWO trips after its loads went out (the trip loop is
    //      unrolled by two for that), so the wait for it never meets a load younger than a whole trip.
    struct StatSet { u32x2 sd, so; float sl; int sq; };
    StatSet S0, S1;
    S0.sq = S1.sq = 0;
    auto issue_stats = [&](StatSet& S, int pr, int t) {  // 3 loads
        const int q0 = 32 * wave + 4 * t;
        S.sq = q0 + g;
        const size_t ro = o_off(pr) + (size_t)q0 * ldbo;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(S.sd) : "v"(st_off), "s"(reinterpret_cast<const char*>(dout) + ro) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(S.so) : "v"(st_off), "s"(reinterpret_cast<const char*>(o) + ro) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(S.sl) : "v"(g * 4), "s"(lse + (size_t)pr * PT + q0) : "memory");
    };
    auto consume_stats = [&](StatSet& S, int buf) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            a = __fmaf_rn(__uint_as_float(S.sd[e] << 16), __uint_as_float(S.so[e] << 16), a);
            a = __fmaf_rn(__uint_as_float(S.sd[e] & 0xffff0000u), __uint_as_float(S.so[e] & 0xffff0000u), a);
        }
        a = row16_sum(a);
        if (c16 == 0) {
            dlt_s[buf * PT + S.sq] = a;
            lse_s[buf * PT + S.sq] = S.sl * L2E;
        }
    };
    // vector-memory instructions a wave issues per trip, in order: dQ store | [4 V loads (trip 7)] | [4 K-tile DMA (trip 0)] | 3 statistics
    // loads | Q / dO rows | [dropout words]
    constexpr int NM = DROP == 2 ? 1 : 0;
#define WAIT_S(S_, N_) asm volatile("s_waitcnt vmcnt(%3) ; data of %0 %1 %2" : "+v"(S_.sd), "+v"(S_.so), "+v"(S_.sl) : "n"(N_) : "memory")
#define WAIT_V(N_) \
    asm volatile("s_waitcnt vmcnt(%4) ; data of %0 %1 %2 %3" : "+v"(vn[0][0]), "+v"(vn[0][1]), "+v"(vn[1][0]), "+v"(vn[1][1]) : "n"(N_) : "memory")
#define PBARRIER()                                         \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

    int pr = blockIdx.x;
    if (pr >= pairs) return;
    int pb = 0;  // statistics buffer of the current pair
    // ---- the first pair, synchronously
    issue_k(pr);
    issue_v(pr);
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        issue_qd(pr, t);
        issue_mask(pr, t);
        issue_stats(S0, pr, t);
        WAIT_S(S0, 0);
        consume_stats(S0, 0);
    }
    WAIT_V(0);
    PBARRIER();

    // ---- per-lane LDS addresses of the transposed reads: row 4g + qp (+ 16, + 32 kb as immediates), 32-B block dt
    const int swh = (2 * g + (qp >> 1)) & 3;  // ((row >> 1) & 3) of rows 4g + qp + 16 n
    unsigned tq[4];                           // Q tile (the dO tile is DL further: immediate)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tq[dt] = lds_u32(lds) + (4 * g + qp) * RB + ((dt ^ swh) << 5) + (pp >> 1) * 16 + 8 * (pp & 1);
    // X[key][32 queries] at a pitch of 96 B: the 8 consecutive key rows an LDS cycle of a transposed read touches start in 8 disjoint
    // 8-bank windows.  (A 64-byte pitch with the row halves swapped every four rows has no bank conflict at all -- the padded pitch
    // costs 25 M conflict cycles per launch in the two-address stores -- and measured 4 % SLOWER, 860 against 828 us: not kept.)
    const unsigned xr0 = lds_u32(lds) + XL + (4 * g + qp) * XP + jqw * 32 + pp * 8;  // as B operand: key rows 4 g + qp, this wave's query block
    const int xw0 = XL + (r0 + c16) * XP + g * 8;                                      // as written: this lane's key, queries 4 g .. (an LDS offset: one register)
    union Frag { bf16x8 v; s16x4 h[2]; u32x4 u; };
    const char* Ql = lds + QL;
    const char* Dl = lds + DL;
    const char* Kl = lds + KL;

    // the eight waves' bias partials of pair (b_, h_) in wave order -> bias_rows; a dQ column has two contributors (the waves of its head-dim
    // block).  192 threads: waves 0-2 behind the pair's last barrier; SKEW: waves 4-6 behind B's first barrier of the next pair.
    auto bias_reduce = [&](int to, int b_, int h_) {
        asm volatile("" : "+v"(to));  // LDS addresses formed here, not hoisted out of the pair loop (two were spilled)
        const int part = to >> 6, d = to & 63;
        float a;
        const char* bq = lds + XL + (to >> 3) * XP + 64 + (to & 7) * 4;  // float `to` of wave 0's partials
        if (part == 0) {
            a = *reinterpret_cast<const float*>(bq + (d >> 4) * BPW) + *reinterpret_cast<const float*>(bq + (d >> 4) * BPW + 4 * BPW);
        } else {
            a = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) a += *reinterpret_cast<const float*>(bq + w8 * BPW);
        }
        float* br = bias_rows;
        asm volatile("" : "+s"(br));  // a scalar base at the point of use (hoisted as a vector pair it was spilled: a scratch reload here drains vmcnt)
        br[(size_t)b_ * (3 * heads * DH) + part * heads * DH + h_ * DH + d] = a;  // (one more store on three waves: the counted waits only over-wait)
    };
    int pprev = -1;  // SKEW: the pair whose bias partials group B still has to reduce

    while (true) {
        const int nxt = pr + gridDim.x < pairs ? pr + gridDim.x : pr;  // past the end: re-fetch this pair (never read)
        const int b = pr / heads, h = pr % heads;
        // ---- prologue: this wave's K rows (B operands of S) and the K^T fragments of its dQ^T tile, from the K tile; V rows from registers
        bf16x8 kf[2][2];
        Frag ktf[8];
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
            const int k = r0 + 16 * jk + c16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[jk][ks] = *reinterpret_cast<const bf16x8*>(row_chunk(Kl, k, 4 * ks + g));
            }
        }
        // ---- S = Q . K^T, dP = dO . V^T for 32 queries x this wave's 32 keys (key on the lane, four queries per register quad)
        f32x4 s[2][2], dp[2][2];
        auto m1 = [&](const int t) {
            const int qc = 32 * t;
            if (t == 0) WAIT_V(20 + NM);  // fetched in trip 7 of the previous pair, in front of its 3 + 1 + NM fetches and the 16 stores
            bf16x8 vf[2][2];
#pragma unroll
            for (int jk = 0; jk < 2; ++jk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    Frag f;
                    f.u = vn[jk][ks];
                    vf[jk][ks] = f.v;
                }
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
                        if constexpr (ABL & 16) {
                            s[qt][jk][0] += __builtin_bit_cast(f32x4, qa)[0] + __builtin_bit_cast(f32x4, kf[jk][ks])[0];
                            dp[qt][jk][0] += __builtin_bit_cast(f32x4, da)[0] + __builtin_bit_cast(f32x4, vf[jk][ks])[0];
                            continue;
                        }
                        s[qt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[jk][ks], s[qt][jk], 0, 0, 0);
                        dp[qt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[jk][ks], dp[qt][jk], 0, 0, 0);
                    }
                }
            }
        };
        int lp = lane;
        asm volatile("" : "+v"(lp));  // the K^T address is used here only: formed per pair instead of living in a register through the loops
        const int gp = lp >> 4, qpp = (lp & 15) >> 2, ppp = lp & 3;
        const unsigned tkw = lds_u32(lds) + KL + (4 * gp + qpp) * RB + ((dtw ^ ((2 * gp + (qpp >> 1)) & 3)) << 5) + (ppp >> 1) * 16 + 8 * (ppp & 1);
#define KT_RD(J_) ktf[J_].h[0] = tr_rd<(J_) * 32 * RB>(tkw); ktf[J_].h[1] = tr_rd<(J_) * 32 * RB + 16 * RB>(tkw)
        KT_RD(0); KT_RD(1); KT_RD(2); KT_RD(3);
        asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                     : "+v"(ktf[0].h[0]), "+v"(ktf[0].h[1]), "+v"(ktf[1].h[0]), "+v"(ktf[1].h[1]), "+v"(ktf[2].h[0]), "+v"(ktf[2].h[1]),
                       "+v"(ktf[3].h[0]), "+v"(ktf[3].h[1]));
        KT_RD(4); KT_RD(5); KT_RD(6); KT_RD(7);
        asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                     : "+v"(ktf[4].h[0]), "+v"(ktf[4].h[1]), "+v"(ktf[5].h[0]), "+v"(ktf[5].h[1]), "+v"(ktf[6].h[0]), "+v"(ktf[6].h[1]),
                       "+v"(ktf[7].h[0]), "+v"(ktf[7].h[1]));
#undef KT_RD
        f32x4 dk[4][2], dv[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt][0] = dk[dt][1] = dv[dt][0] = dv[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- dQ^T tile (head-dim block dtw, queries 32 tt + 16 jqw ..) of trip tt = K^T . dS^T over all 256 keys, from X[tt & 1]: all 16
        //      transposed reads are issued at once, the first four MFMAs start when half of them are back (LDS returns in order).
        //      (Running the tile one trip late, under the next trip's softmax, measured the same time at 15 registers more: not kept.)
        Frag xb[8];
        f32x4 dq, dq1;
        f32x4 qsum = f32x4{0.f, 0.f, 0.f, 0.f};  // sums over this wave's queries of the dQ values it stores (bias_rows)
#define X_RD(J_) xb[J_].h[0] = tr_rd<(J_) * 32 * XP>(xr); xb[J_].h[1] = tr_rd<(J_) * 32 * XP + 16 * XP>(xr)
        auto dq_tile = [&](int tt) {
            const unsigned xr = xr0 + (tt & 1) * XSZ;
            X_RD(0); X_RD(1); X_RD(2); X_RD(3);
            X_RD(4); X_RD(5); X_RD(6); X_RD(7);
            asm volatile("s_waitcnt lgkmcnt(8) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                         : "+v"(xb[0].h[0]), "+v"(xb[0].h[1]), "+v"(xb[1].h[0]), "+v"(xb[1].h[1]), "+v"(xb[2].h[0]), "+v"(xb[2].h[1]),
                           "+v"(xb[3].h[0]), "+v"(xb[3].h[1]));
            dq = dq1 = f32x4{0.f, 0.f, 0.f, 0.f};  // two chains of four: a dependent MFMA waits for its predecessor's result
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[0].v, xb[0].v, dq, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[1].v, xb[1].v, dq1, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[2].v, xb[2].v, dq, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[3].v, xb[3].v, dq1, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                         : "+v"(xb[4].h[0]), "+v"(xb[4].h[1]), "+v"(xb[5].h[0]), "+v"(xb[5].h[1]), "+v"(xb[6].h[0]), "+v"(xb[6].h[1]),
                           "+v"(xb[7].h[0]), "+v"(xb[7].h[1]));
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[4].v, xb[4].v, dq, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[5].v, xb[5].v, dq1, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[6].v, xb[6].v, dq, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[7].v, xb[7].v, dq1, 0, 0, 0);
            u32x2 w;  // 1 store
            w[0] = pack_bf16x2((dq[0] + dq1[0]) * scale, (dq[1] + dq1[1]) * scale);
            w[1] = pack_bf16x2((dq[2] + dq1[2]) * scale, (dq[3] + dq1[3]) * scale);
            char* ob = reinterpret_cast<char*>(dqkv) + (((size_t)b * PT + 32 * tt + 16 * jqw) * ld_dqkv + h * DH + 16 * dtw) * 2;
            asm volatile("" : "+s"(ob));  // a scalar base + the 32-bit lane offset (the per-lane 64-bit sum, hoisted out of both loops, was spilled)
            *reinterpret_cast<u32x2*>(ob + out_off) = w;
            if (BIAS_ON) {
                qsum[0] += __uint_as_float(w[0] << 16); qsum[1] += __uint_as_float(w[0] & 0xffff0000u);
                qsum[2] += __uint_as_float(w[1] << 16); qsum[3] += __uint_as_float(w[1] & 0xffff0000u);
            }
        };
        if constexpr (SKEW) {
            if (grp) {  // B: half a trip behind A from here on.  All eight waves' bias partials of the previous pair are written now
                PBARRIER();
                if (BIAS_ON && pprev >= 0 && tid < 448) bias_reduce(tid - 256, pprev / heads, pprev % heads);
            }
        }

        auto trip = [&](const int t, StatSet& S) {
            const int qc = 32 * t;
            m1(t);
            // SKEW: the next pair's V rows go out here, behind their last use (this trip's dP MFMAs), a slot and a half earlier than in lock
            // step: they are waited for at the next pair's first MFMA, and from the matrix slot of trip 7 that was less than the memory
            // latency under load.  (In issue order they now stand in front of trip 7's dQ store: every counted wait stays equal or stricter.)
            if constexpr (SKEW && !(ABL & 32)) {
                if (t == 7) issue_v(nxt);
            }
            // ---- the transposed Q / dO fragments (for dK / dV): all 16 reads go out together behind the softmax and each group of eight MFMAs
            //      waits only for its own eight.  (The first eight in FRONT of the softmax -- they do not depend on it -- measured 1 % faster
            //      at 16 registers more, which the instances with the bias sums do not have.)
            const unsigned qco = (unsigned)qc * RB;
            Frag dot[4], qt_[4];
#define QD_RD(DT_)                                                                                                   \
    dot[DT_].h[0] = tr_rd<DL>(tq[DT_] + qco); dot[DT_].h[1] = tr_rd<DL + 16 * RB>(tq[DT_] + qco);                   \
    qt_[DT_].h[0] = tr_rd<0>(tq[DT_] + qco);  qt_[DT_].h[1] = tr_rd<16 * RB>(tq[DT_] + qco)
            // ---- P = exp2(S * scale * log2e - lse * log2e), dS / scale = P * (dP - delta) (the softmax scale, a power of two, multiplies
            //      the dQ and dK tiles at their stores: bit-identical, one multiply per score element less); dropped P for dV
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const f32x4 lr = *reinterpret_cast<const f32x4*>(lse_s + pb * PT + qc + 16 * qt + 4 * g);
                const f32x4 dr = *reinterpret_cast<const f32x4*>(dlt_s + pb * PT + qc + 16 * qt + 4 * g);
#pragma unroll
                for (int jk = 0; jk < 2; ++jk) {
                    unsigned mbits = 0u;  // bit r: keep(query qc + 16 qt + 4 g + r, this lane's key r0 + 16 jk + c16)
                    if constexpr (DROP == 2) {
                        // word (query block qc / 16 + qt, key block 2 wave + jk, r = key & 3), its 16-bit group (key >> 2) & 3, bits 4 g ..
                        const char* mp = lds + MK + ((((qc >> 4) + qt) * 16 + 2 * wave + jk) << 5) + (c16 & 3) * 8 + (c16 >> 2) * 2;
                        mbits = (unsigned)*reinterpret_cast<const unsigned short*>(mp) >> (4 * g);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {  // packed arithmetic: two queries per instruction
                        if constexpr (ABL & 1) continue;
                        const f32x2_ arg = __builtin_elementwise_fma(f32x2_{s[qt][jk][r], s[qt][jk][r + 1]}, f32x2_{sl2, sl2}, f32x2_{-lr[r], -lr[r + 1]});
                        const f32x2_ pv = f32x2_{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                        f32x2_ dpv = f32x2_{dp[qt][jk][r], dp[qt][jk][r + 1]};
                        if constexpr (DROP == 2) {
                            const bool k0 = (mbits >> r) & 1u, k1 = (mbits >> (r + 1)) & 1u;
                            dpv = f32x2_{k0 ? dpv[0] : 0.0f, k1 ? dpv[1] : 0.0f};
                            s[qt][jk][r] = k0 ? pv[0] : 0.0f;  // 1 / (1 - p) multiplies the dV tile at the end
                            s[qt][jk][r + 1] = k1 ? pv[1] : 0.0f;
                            const f32x2_ ds_ = pv * __builtin_elementwise_fma(dpv, f32x2_{dc.scale, dc.scale}, f32x2_{-dr[r], -dr[r + 1]});
                            dp[qt][jk][r] = ds_[0];
                            dp[qt][jk][r + 1] = ds_[1];
                        } else {
                            s[qt][jk][r] = pv[0];
                            s[qt][jk][r + 1] = pv[1];
                            const f32x2_ ds_ = pv * (dpv - f32x2_{dr[r], dr[r + 1]});
                            dp[qt][jk][r] = ds_[0];
                            dp[qt][jk][r + 1] = ds_[1];
                        }
                    }
                    if constexpr (DROP == 2) {  // one (query block, key block) at a time: the selects' temporaries of all four do not fit
                        asm volatile("" : "+v"(s[qt][jk][0]), "+v"(s[qt][jk][1]), "+v"(s[qt][jk][2]), "+v"(s[qt][jk][3]), "+v"(dp[qt][jk][0]),
                                     "+v"(dp[qt][jk][1]), "+v"(dp[qt][jk][2]), "+v"(dp[qt][jk][3]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            Frag pf[2], dsf[2];
#pragma unroll
            for (int jk = 0; jk < 2; ++jk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[jk].v[r] = (__bf16)s[0][jk][r];
                    pf[jk].v[4 + r] = (__bf16)s[1][jk][r];
                    dsf[jk].v[r] = (__bf16)dp[0][jk][r];
                    dsf[jk].v[4 + r] = (__bf16)dp[1][jk][r];
                }
            // ---- dS^T block to the exchange buffer: X[key][query], 4 queries (8 B) per store
            if constexpr (!(ABL & 64)) {
                char* xw = lds + xw0 + (t & 1) * XSZ;
#pragma unroll
                for (int jk = 0; jk < 2; ++jk) {
                    *reinterpret_cast<u32x2*>(xw + jk * 16 * XP) = u32x2{dsf[jk].u[0], dsf[jk].u[1]};
                    *reinterpret_cast<u32x2*>(xw + jk * 16 * XP + 32) = u32x2{dsf[jk].u[2], dsf[jk].u[3]};
                }
            }
            if constexpr (SKEW && !(ABL & 8)) PBARRIER();  // end of H1: this group's dS block is in X[t & 1]
            // ---- dV^T += dO^T . P, dK^T += Q^T . dS
            if constexpr (ABL & 2) {
                dv[0][0][0] += __builtin_bit_cast(f32x4, pf[0].v)[0] + __builtin_bit_cast(f32x4, pf[1].v)[0];
                dk[0][0][0] += __builtin_bit_cast(f32x4, dsf[0].v)[0] + __builtin_bit_cast(f32x4, dsf[1].v)[0];
            } else {
                QD_RD(0); QD_RD(1); QD_RD(2); QD_RD(3);
                asm volatile("s_waitcnt lgkmcnt(8) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                             : "+v"(dot[0].h[0]), "+v"(dot[0].h[1]), "+v"(dot[1].h[0]), "+v"(dot[1].h[1]), "+v"(qt_[0].h[0]),
                               "+v"(qt_[0].h[1]), "+v"(qt_[1].h[0]), "+v"(qt_[1].h[1]));
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int jk = 0; jk < 2; ++jk) {
                        dv[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[dt].v, pf[jk].v, dv[dt][jk], 0, 0, 0);
                        dk[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_[dt].v, dsf[jk].v, dk[dt][jk], 0, 0, 0);
                    }
                asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7"
                             : "+v"(dot[2].h[0]), "+v"(dot[2].h[1]), "+v"(dot[3].h[0]), "+v"(dot[3].h[1]), "+v"(qt_[2].h[0]),
                               "+v"(qt_[2].h[1]), "+v"(qt_[3].h[0]), "+v"(qt_[3].h[1]));
#pragma unroll
                for (int dt = 2; dt < 4; ++dt)
#pragma unroll
                    for (int jk = 0; jk < 2; ++jk) {
                        dv[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[dt].v, pf[jk].v, dv[dt][jk], 0, 0, 0);
                        dk[dt][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_[dt].v, dsf[jk].v, dk[dt][jk], 0, 0, 0);
                    }
            }
#undef QD_RD
            if constexpr (!SKEW && !(ABL & 8)) PBARRIER();  // every wave's dS block is in X[t & 1]; the trip's Q / dO rows and dropout words are dead

            // ---- the next pair's share of this trip: statistics of four queries, the Q / dO rows and dropout words of this trip's slot
            //      (trip 0: also the K tile; trip 7: this wave's V rows; both IN FRONT of the statistics loads, so that one count serves
            //      every trip).
            //      ONE wait statement carries the loaded registers: with a second one in the other arm of an if / else, hipcc merged the
            //      arms by COPYING the loaded registers in front of one arm's wait -- stale statistics in a few pairs per thousand
            //      (tools/check_asm_loads.py scans for that pattern).
            if constexpr (!(ABL & 4)) {
                if constexpr (SKEW) {
                    const int tt = t - 1 + grp;  // A: the tile of the trip before (X[t & 1] still lacks B's block)
                    if (tt >= 0) dq_tile(tt);
                } else {
                    dq_tile(t);
                }
            }
            if constexpr (!(ABL & 32)) {
            if (!(ABL & 128) && t >= 2) {  // 128: no wait for / use of the statistics loads
                // this set's loads went out in trip t - 2; younger: that trip's rows + words, trip t - 1 whole (store, 3 loads, rows,
                // words), this trip's dQ store
                WAIT_S(S, 7 + 2 * NM);
                consume_stats(S, pb ^ 1);
            }
            if (!SKEW && t == 7) issue_v(nxt);
            if (!(ABL & 256) && t == 0) issue_k(nxt);  // 256: no LDS-DMA
            if constexpr (!(ABL & 512)) issue_stats(S, nxt, t);  // 512: no statistics loads
            if constexpr (!(ABL & 256)) {
                if constexpr (SKEW) {  // the slot of trip t - 1 (dead since the other group's H2(t - 1)); trip 0: slot 7, for this pair's trip 7
                    issue_qd(t ? nxt : pr, (t + 7) & 7);
                    issue_mask(t ? nxt : pr, (t + 7) & 7);
                } else {
                    issue_qd(nxt, t);
                    issue_mask(nxt, t);
                }
            }
            }
            if constexpr (SKEW && !(ABL & 8)) PBARRIER();  // end of H2
        };
#pragma unroll 1
        for (int t2 = 0; t2 < 8; t2 += 2) {
            trip(t2, S0);
            trip(t2 + 1, S1);
        }
        // ---- end of the pair: the statistics of trip 7, then dK and dV of this wave's keys
        if constexpr (SKEW && !(ABL & 4)) {
            if (!grp) dq_tile(7);  // A: X[1] is whole since the barrier that ended its H2(7)
        }
#undef X_RD
        if constexpr (!(ABL & (32 | 128))) {
            WAIT_S(S0, 10 + 2 * NM);  // trip 6's set: behind it its rows + words and trip 7's store, 4 V loads, 3 loads, rows, words
            consume_stats(S0, pb ^ 1);
            WAIT_S(S1, 1 + NM);       // trip 7's set: its rows + words (the one wait of a pair that meets loads less than a trip old)
            consume_stats(S1, pb ^ 1);
        }
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {  // 16 stores
            char* ob = reinterpret_cast<char*>(dqkv) + (((size_t)b * PT + r0 + 16 * jk) * ld_dqkv + h * DH) * 2;
            asm volatile("" : "+s"(ob));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 wk_, wv_;
                wk_[0] = pack_bf16x2(dk[dt][jk][0] * scale, dk[dt][jk][1] * scale);
                wk_[1] = pack_bf16x2(dk[dt][jk][2] * scale, dk[dt][jk][3] * scale);
                const float vs = DROP == 2 ? dc.scale : 1.0f;
                wv_[0] = pack_bf16x2(dv[dt][jk][0] * vs, dv[dt][jk][1] * vs);
                wv_[1] = pack_bf16x2(dv[dt][jk][2] * vs, dv[dt][jk][3] * vs);
                if constexpr (ABL & 2048) {  // 2048 (timing only): the same bytes as lane-linear 512-B pieces of this wave's rows
                    *reinterpret_cast<u32x2*>(ob + heads * DH * 2 + (size_t)(4 * dt) * ld_dqkv * 2 + lane * 8) = wk_;
                    *reinterpret_cast<u32x2*>(ob + 2 * heads * DH * 2 + (size_t)(4 * dt) * ld_dqkv * 2 + lane * 8) = wv_;
                } else if constexpr (!(ABL & 1024)) {  // 1024: no dK / dV stores
                    *reinterpret_cast<u32x2*>(ob + (heads * DH + 16 * dt) * 2 + out_off) = wk_;
                    *reinterpret_cast<u32x2*>(ob + (2 * heads * DH + 16 * dt) * 2 + out_off) = wv_;
                }
                if (BIAS_ON) {  // the accumulators are dead now: they carry the stored (rounded) values into the sums over jk below
                    dk[dt][jk] = f32x4{__uint_as_float(wk_[0] << 16), __uint_as_float(wk_[0] & 0xffff0000u), __uint_as_float(wk_[1] << 16),
                                       __uint_as_float(wk_[1] & 0xffff0000u)};
                    dv[dt][jk] = f32x4{__uint_as_float(wv_[0] << 16), __uint_as_float(wv_[0] & 0xffff0000u), __uint_as_float(wv_[1] << 16),
                                       __uint_as_float(wv_[1] & 0xffff0000u)};
                }
            }
        }
        // column sums over this wave's 32 keys (dK, dV) / its 128 queries (dQ): lane c16 = 0 of row group g holds d = 4 g + r
        auto partial_base = [&](bool& first) {
            int lb = lane;
            asm volatile("" : "+v"(lb));  // one per-lane base formed here + immediates (hoisted addresses cost registers through both loops)
            const int gb = lb >> 4;
            first = (lb & 15) == 0;
            return lds + XL + wave * BPW + (gb >> 1) * XP + 64 + (gb & 1) * 16;  // float 4 g of this wave's partials; + 16 dt floats = + 2 dt rows
        };
        auto q_partials = [&](char* bpl, bool first) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sq = row16_sum(qsum[r]);
                if (first) *reinterpret_cast<float*>(bpl + 2 * dtw * XP + 4 * r) = sq;
            }
        };
        auto kv_partials = [&](char* bpl, bool first) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sk = row16_sum(dk[dt][0][r] + dk[dt][1][r]), sv = row16_sum(dv[dt][0][r] + dv[dt][1][r]);
                    if (first) {
                        *reinterpret_cast<float*>(bpl + (8 + 2 * dt) * XP + 4 * r) = sk;   // floats 64 ..
                        *reinterpret_cast<float*>(bpl + (16 + 2 * dt) * XP + 4 * r) = sv;  // floats 128 ..
                    }
                }
        };
        if (BIAS_ON) {
            bool first;
            char* bpl = partial_base(first);
            q_partials(bpl, first);
            kv_partials(bpl, first);
        }
        if constexpr (SKEW) {
            if (!grp) PBARRIER();  // A: its pair epilogue ran next to B's H2(7); the next pair's first statistics (wave 0's) are written
            pprev = pr;
        } else {
            PBARRIER();  // the next pair's statistics are written, its K tile is whole (this wave's share was complete at the wait of trip 2)
            if (BIAS_ON && tid < 192) bias_reduce(tid, b, h);
        }
        if (pr + (int)gridDim.x >= pairs) break;
        pr += gridDim.x;
        pb ^= 1;
    }
    if constexpr (SKEW) {
        if (grp) {  // B's partials of the last pair are written behind this barrier (group A has left: not counted)
            PBARRIER();
            if (BIAS_ON && tid < 448) bias_reduce(tid - 256, pprev / heads, pprev % heads);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus fetches of the last pair land before the LDS is released
#undef WAIT_S
#undef WAIT_V
#undef PBARRIER
}

}  // namespace

static int g_attn_bwd_skew = -1;  // -1: not decided (BSI_ATTN_BWD_SKEW, default on)

// 1 (default): the single-sweep backward runs its wave groups half a trip apart; 0: in lock step (the A/B partner; bit-identical results).
// Returns the previous setting.
extern "C" int bsi_set_attention_bwd_skew(int on) {
    const int prev = g_attn_bwd_skew < 0 ? (getenv("BSI_ATTN_BWD_SKEW") ? atoi(getenv("BSI_ATTN_BWD_SKEW")) != 0 : 1) : g_attn_bwd_skew;
    g_attn_bwd_skew = on != 0;
    return prev;
}

// (the dispatcher in attention_bwd.hip decides when this kernel runs)
int bsi_attention_bwd_exchange(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse, int B,
                               int heads, void* dqkv, int ld_dqkv, DropCfg dc, const void* maskw, hipStream_t stream, float* bias_rows) {
    const int pairs = B * heads, ncu = compute_cus();
    const int grid = pairs < ncu ? pairs : ncu;
    const float sc = 1.0f / sqrtf((float)DH);
    auto go = [&](auto kern) {
        set_max_lds(reinterpret_cast<const void*>(kern), LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, stream, reinterpret_cast<const __bf16*>(qkv), ld_qkv,
                           reinterpret_cast<const __bf16*>(out), reinterpret_cast<const __bf16*>(dout), ld_o, lse, pairs, heads,
                           reinterpret_cast<__bf16*>(dqkv), ld_dqkv, sc, dc, reinterpret_cast<const char*>(maskw), bias_rows);
    };
    if (g_attn_bwd_skew < 0) g_attn_bwd_skew = getenv("BSI_ATTN_BWD_SKEW") ? atoi(getenv("BSI_ATTN_BWD_SKEW")) != 0 : 1;
    if (g_attn_bwd_skew) {
        if (dc.thr) go(attention_bwd_x_kernel<2, true>);
        else go(attention_bwd_x_kernel<0, true>);
    } else if (dc.thr) {
        go(attention_bwd_x_kernel<2, false>);
    } else {
        go(attention_bwd_x_kernel<0, false>);
    }
    BSI_CHECK_LAUNCH("bsi_attention_bwd(exchange)");
    return BSI_OK;
}
