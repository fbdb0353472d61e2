// Weight-gradient GEMM for gfx950:  C[N,K] (+)= P[M,N]^T . Q[M,K]   (P = dY, Q = X, both bf16 row-major with
// the contraction index m as the SLOW dimension), fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// This is the backward of every nn.Linear on the DiT path (autograd's `mm` of grad_output^T with the saved
// input; SURVEY §2.4 "backward: 14 mm per block").  The operands cannot be fed to the forward kernel without
// materialising transposes, so the tiles are staged as [32 m][256 cols] slabs (512-B rows, the natural global
// layout, full-line DMA) and the MFMA fragments (8 consecutive m for a fixed column) are produced by the
// hardware transpose read ds_read_b64_tr_b16, two per fragment — the same trick the attention kernel uses for V.
// 16-B chunks of a row are XOR-swizzled with ((m&3)<<1 | ((m>>3)&1)<<3) on the DMA source and on the read side,
// which makes the transposed reads conflict free.  Schedule: ping-pong wave groups over a 5-slot LDS ring (see
// gemm_bf16.hip variant 4).  The contraction (tokens) is long and the output small, so the M range is split
// over `splits` workgroups per tile; partial tiles go to fp32 slabs that bsi_reduce_slabs sums
// (deterministic, unlike float atomics, and ~5x cheaper than atomics at these shapes).
#include "common.h"

namespace {

struct TnParams {
    const __bf16* P;  // [M, ldp]  (n columns)
    const __bf16* Q;  // [M, ldq]  (k columns)
    float* out;       // [splits][N, ldc]
    int M, N, K;
    int ldp, ldq, ldc;
    int tiles_n, tiles_k, splits, m_per_split;
    size_t slab_stride;  // floats between slabs
    float* colsum;       // optional: [splits][colsum_stride] column sums of P (bias gradient), written by the k-tile-0 workgroups
    size_t colsum_stride;
    // paired launch (bsi_gemm_tn_pair_bf16): n tiles >= pair_tiles belong to a SECOND problem of the same M, K, ldq, ldc -- its P
    // operand, Q operand, N, output slabs -- so that two weight gradients with few output tiles each fill the chip together
    int pair_tiles;      // 0 = one problem
    const __bf16* P2;
    const __bf16* Q2;
    float* out2;
    int N2, ldp2;
    size_t slab_stride2;
    int xcd_map;         // XCD-aware workgroup order (off: BSI_TN_ABL & 1)
    int abl;             // laboratory (BSI_TN_ABL): 2 = no operand DMA after the prologue, 4 = fragment reads only in the first stage, 8 = no MFMAs
};

constexpr int T_RB = 512;                 // bytes per LDS row (256 bf16 columns)
constexpr int T_TILE = 32 * T_RB;         // one operand stage: 32 m-rows
constexpr int T_SLOT = 2 * T_TILE;        // 32 KB
constexpr int T_R = 5, T_D = T_R - 1;

__device__ __forceinline__ int tn_swz(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

__global__ __launch_bounds__(512) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave >> 2;   // ping-pong group; also the n half (128 columns) of the tile
    const int wk = wave & 3;    // 64-wide k slice

    // Workgroup -> (split, n tile, k tile).  Workgroups go to the 8 XCDs round robin (blockIdx % 8) and every XCD has its own L2,
    // so the workgroups of ONE XCD should share operand panels: XCD x takes a contiguous range of the order (split, band, slow
    // tile, fast tile), where `fast` is the smaller tile dimension walked in bands of <= 4 -- 32 consecutive entries are 8 x 4 tiles
    // of one split = 12 panel streams for 32 workgroups.  (Plain blockIdx order gave an XCD one k tile, 2-8 n tiles and ALL
    // splits: 36-48 streams, each P panel fetched by four XCDs.)
    const int tiles = p.tiles_n * p.tiles_k;
    int tile_n, tile_k, split;
    if (p.xcd_map) {
        const int L = (int)gridDim.x, b = (int)blockIdx.x;
        const int x = b & 7, lo = L >> 3, rem = L & 7;
        const int id = x * lo + min(x, rem) + (b >> 3);
        split = id / tiles;
        const int t = id - split * tiles;
        const bool k_fast = p.tiles_k <= p.tiles_n;
        const int F = k_fast ? p.tiles_k : p.tiles_n, S = k_fast ? p.tiles_n : p.tiles_k;
        const int band = t / (4 * S), r = t - band * 4 * S;
        const int w = min(4, F - 4 * band);  // the last band may be narrower
        const int sl = r / w, f = 4 * band + (r - sl * w);
        tile_n = k_fast ? sl : f;
        tile_k = k_fast ? f : sl;
    } else {
        const int tile = blockIdx.x % tiles;
        split = blockIdx.x / tiles;
        tile_n = tile / p.tiles_k;
        tile_k = tile % p.tiles_k;
    }
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 31) / 32;

    // staging: per stage 64 rows of 512 B (Q tile rows 0..31, then P tile rows 0..31); a wave-instruction covers 2 rows;
    // wave w issues instruction slots q*8 + w, q < 4  (slots 0..15 -> Q tile, 16..31 -> P tile)
    const int srow = lane >> 5, spos = lane & 31;
    const char* gsrc[4];
    int grow[4];  // m row (within the stage) of each slot, for the tail clamp
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = q * 8 + wave;
        const int r = (slot & 15) * 2 + srow;  // m row within the 32-row stage
        const int c = spos ^ tn_swz(r);
        grow[q] = r;
        if (slot < 16) {
            int col = k0 + c * 8;
            col = col < p.K ? col : 0;  // columns beyond K are never stored
            gsrc[q] = reinterpret_cast<const char*>(p.Q + col);
        } else {
            int col = n0 + c * 8;
            col = col < p.N ? col : 0;
            gsrc[q] = reinterpret_cast<const char*>(p.P + col);
        }
    }
    auto stage = [&](int v, int slot_i) {
        char* base = lds + slot_i * T_SLOT;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = q * 8 + wave;
            int m = mbeg + v * 32 + grow[q];
            m = m < mend ? m : mend - 1;  // tail rows are masked to zero contribution below (nk rounds up)
            const size_t ld = slot < 16 ? (size_t)p.ldq : (size_t)p.ldp;
            char* dst = base + slot * 1024;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc[q] + (size_t)m * ld * 2), LDS_PTR(dst), 16, 0, 0);
        }
    };

    // fragment addressing: lane = (q, q', pp): q = lane>>4 selects m block 8q..8q+7, q' = (lane&15)>>2 the row inside a
    // 4-row transpose block, pp = lane&3 the 4-column quad.
    const int q = lane >> 4, qp = (lane & 15) >> 2, pp = lane & 3;
    const int mA = 8 * q + qp, mB = mA + 4;
    const int swA = tn_swz(mA), swB = tn_swz(mB);
    const int rowA = mA * T_RB, rowB = mB * T_RB;
    const int half8 = 8 * (pp & 1), chp = pp >> 1;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4], bf[8];
    // fused bias gradient: column sums of the P tile = one extra MFMA with an all-ones A operand per 16-column unit.  The 16 units of
    // an n tile are dealt out over the workgroups that share it (k tiles 0 .. min(tiles_k, 16) - 1) and, inside a workgroup, over the
    // four waves of the group that holds the unit's fragments: at most one extra MFMA per wave and stage for tiles_k >= 2 (two when a
    // single workgroup owns the n tile).  With all 16 units in the k-tile-0 workgroups those ran 34 MFMAs per stage against 32
    // everywhere else, and the launch waits for its slowest workgroup: fc1 1068 -> 978 us without the bias gradient.  (Adding the
    // fragments on the vector ALU instead -- 32 unpack + 32 add instructions per stage -- was measured: 1055 us, no better.)
    int cs_j[2] = {-1, -1};  // this wave's units (fragment index j of its P half), wave-uniform
    if (p.colsum != nullptr) {
        const int cs = p.tiles_k < 16 ? p.tiles_k : 16;
        int seen = 0;
        for (int j = 0; j < 8; ++j) {
            if ((8 * wg + j) % cs != tile_k) continue;
            if ((seen & 3) == wk) cs_j[seen >> 2] = j;
            ++seen;
        }
    }
    const bool do_colsum = cs_j[0] >= 0;
    f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

#define TN_BARRIER()                             \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

    auto load_frags = [&](const char* b) {
        union U { bf16x8 v; s16x4 h[2]; };
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // Q tile, columns 64*wk + 16i + 4pp
            const int ch = (64 * wk + 16 * i) / 8 + chp;
            U u;
            u.h[0] = TR(b + rowA + ((ch ^ swA) << 4) + half8);
            u.h[1] = TR(b + rowB + ((ch ^ swB) << 4) + half8);
            af[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // P tile, columns 128*wg + 16j + 4pp
            const int ch = (128 * wg + 16 * j) / 8 + chp;
            U u;
            u.h[0] = TR(b + T_TILE + rowA + ((ch ^ swA) << 4) + half8);
            u.h[1] = TR(b + T_TILE + rowB + ((ch ^ swB) << 4) + half8);
            bf[j] = u.v;
        }
    };

#pragma unroll
    for (int d = 0; d < T_D; ++d)
        if (d < nk) stage(d, d);
    if (nk >= T_D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (T_D - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TN_BARRIER();
    if (wg == 1) TN_BARRIER();

    int slot = 0, pslot = T_D;
    for (int v = 0; v < nk; ++v) {
        if (!(BSI_ABL(p.abl, 4)) || v == 0) load_frags(lds + slot * T_SLOT);
        if (v + T_D < nk && !(BSI_ABL(p.abl, 2))) {
            stage(v + T_D, pslot);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (T_D - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mbeg + v * 32 + 32 > mend) {
            // ragged tail: rows >= mend were clamped to a valid address; zero their contribution (element e of a fragment
            // is contraction row 8q + e)
            const int valid = mend - (mbeg + v * 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (8 * q + e >= valid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i][e] = (__bf16)0.0f;
                }
            }
        }
        bf16x8 onesf;
        {
            const int valid = mend - (mbeg + v * 32);  // >= 32 except in the ragged tail stage
#pragma unroll
            for (int e = 0; e < 8; ++e) onesf[e] = (8 * q + e < valid) ? (__bf16)1.0f : (__bf16)0.0f;
        }
        TN_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        if (!(BSI_ABL(p.abl, 8))) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_colsum) {  // wave-uniform unit indices: branches, not register selects
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (cs_j[u] < 0) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (cs_j[u] == j) accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onesf, bf[j], accb[u], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        TN_BARRIER();
        slot = (slot == T_R - 1) ? 0 : slot + 1;
        pslot = (pslot == T_R - 1) ? 0 : pslot + 1;
    }
    if (wg == 0) TN_BARRIER();
#undef TN_BARRIER
#undef TR

    if (do_colsum && lane < 16) {  // D row 0 (lanes 0..15, register 0) holds the column sums
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (cs_j[u] < 0) continue;
            const int n = n0 + 128 * wg + 16 * cs_j[u] + lane;
            if (n < p.N) p.colsum[(size_t)split * p.colsum_stride + n] = accb[u][0];
        }
    }
    // D rows = k (4*(lane>>4) + reg inside A tile i), D cols = n (lane & 15 inside B tile j)
    float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 128 * wg + 16 * j + (lane & 15);
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 64 * wk + 16 * i + 4 * q;
            if (k < p.K) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(out + (size_t)n * p.ldc + k));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the same GEMM on an LDS image that needs NO swizzle and whose fragment addresses are one per-lane register + immediates.
// The kernel above reads its fragments through the ds_read_tr builtin, in front of which hipcc puts `s_waitcnt vmcnt(0)` whenever
// LDS-DMA is in flight: every load phase began by draining the whole operand pipeline (the four-stage look-ahead was a one-stage one),
// and the XOR swizzle made the 24 fragment addresses 24 registers plus a slot add each per phase.  Here:
//   * a DMA instruction writes ONE contraction row of each operand: lanes 0-31 the 512 B of the Q row, lanes 32-63 those of the P row
//     (contiguous in LDS, as the hardware requires); rows sit at a pitch of 1024 + 32 B, so eight consecutive row positions start in
//     eight disjoint 8-bank windows and the transposed reads (4 rows x 32 B per 16-lane group, two groups per LDS cycle) are conflict
//     free without any XOR;
//   * contraction row m of a stage sits at position (m with bits 2 and 3 swapped): the two 16-lane groups of an LDS cycle read rows
//     8q .. 8q+3 of q = 0, 1 -- positions 0..7 -- and the second read of a fragment (rows + 4) is 8 positions = an immediate further;
//   * fragment i / j of a wave is 32 B further in the row: the 24 reads of a stage are two per-lane base registers (Q side, P side)
//     + immediates, as inline asm, behind ONE counted wait; the operand DMA keeps its look-ahead (ring of 4 slots, 3 in flight).
constexpr int T2_PITCH = 1024 + 32, T2_SLOT = 32 * T2_PITCH, T2_R = 4, T2_D = T2_R - 1;  // 33,792 B per stage, 135 KB

template <int OFF>
__device__ __forceinline__ s16x4 tn_tr(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

// VAR (DMA placement): 0 = all four instructions of a stage in the load phase; otherwise NL = VAR & 3 of them in the load phase and the
// others inside the MFMA phase, behind the MFMA groups j = (VAR >> 4) & 7 and (VAR >> 8) & 7 (both behind the first when VAR & 0x1000)
template <int VAR>
__global__ __launch_bounds__(512) void gemm_tn2_kernel(const TnParams p) {
    constexpr int NL = VAR == 0 ? 4 : (VAR & 3), JA = (VAR >> 4) & 7, JB = (VAR >> 8) & 7;
    constexpr bool TOGETHER = (VAR & 0x1000) != 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave >> 2;   // ping-pong group; also the n half (128 columns) of the tile
    const int wk = wave & 3;    // 64-wide k slice

    const int tiles = p.tiles_n * p.tiles_k;
    int tile_n, tile_k, split;
    if (p.xcd_map) {  // XCD-aware order, as gemm_tn_kernel
        const int L = (int)gridDim.x, b = (int)blockIdx.x;
        const int x = b & 7, lo = L >> 3, rem = L & 7;
        const int id = x * lo + min(x, rem) + (b >> 3);
        split = id / tiles;
        const int t = id - split * tiles;
        const bool k_fast = p.tiles_k <= p.tiles_n;
        const int F = k_fast ? p.tiles_k : p.tiles_n, S = k_fast ? p.tiles_n : p.tiles_k;
        const int band = t / (4 * S), r = t - band * 4 * S;
        const int w = min(4, F - 4 * band);
        const int sl = r / w, f = 4 * band + (r - sl * w);
        tile_n = k_fast ? sl : f;
        tile_k = k_fast ? f : sl;
    } else {
        const int tile = blockIdx.x % tiles;
        split = blockIdx.x / tiles;
        tile_n = tile / p.tiles_k;
        tile_k = tile % p.tiles_k;
    }
    // the problem this n tile belongs to (paired launch: see TnParams)
    const bool second = p.pair_tiles > 0 && tile_n >= p.pair_tiles;
    const __bf16* const Pp = second ? p.P2 : p.P;
    const __bf16* const Qp = second ? p.Q2 : p.Q;
    const int Np = second ? p.N2 : p.N, ldpp = second ? p.ldp2 : p.ldp;
    float* const outp = second ? p.out2 : p.out;
    const size_t slabp = second ? p.slab_stride2 : p.slab_stride;
    if (second) tile_n -= p.pair_tiles;
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 31) / 32;

    // ---- staging: instruction t = q * 8 + wave (q < 4) fills position t; lanes 0-31: Q row, lanes 32-63: P row, 16 B per lane
    const bool pside = lane >= 32;
    const int c16 = lane & 31;
    const char* gbase;  // this lane's column of its operand
    unsigned ldb;       // row pitch in bytes of its operand
    {
        int col = (pside ? n0 : k0) + c16 * 8;
        col = col < (pside ? Np : p.K) ? col : 0;  // columns beyond the matrix are never stored
        gbase = reinterpret_cast<const char*>((pside ? Pp : Qp) + col);
        ldb = (unsigned)(pside ? ldpp : p.ldq) * 2u;
    }
    int mrow[4];  // contraction row (within the stage) of instruction q: position t holds row (t with bits 2 and 3 swapped)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = q * 8 + wave;
        mrow[q] = (t & 19) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3);
    }
    auto stage_one = [&](int v, int slot_i, int q) {  // instruction q of a stage
        char* base = lds + slot_i * T2_SLOT;
        int m = mbeg + v * 32 + mrow[q];
        m = m < mend ? m : mend - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(gbase + (size_t)m * ldb), LDS_PTR(base + (q * 8 + wave) * T2_PITCH), 16, 0, 0);
    };

    auto stage = [&](int v, int slot_i) {
        char* base = lds + slot_i * T2_SLOT;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int m = mbeg + v * 32 + mrow[q];
            m = m < mend ? m : mend - 1;  // tail rows are masked to zero contribution below (nk rounds up)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gbase + (size_t)m * ldb), LDS_PTR(base + (q * 8 + wave) * T2_PITCH), 16, 0, 0);
        }
    };

    // ---- fragments: lane = (q, q', pp): rows 8q + q' (+4), 4-column quad pp; position of row 8q + q' = q' + 4 (q & 1) + 16 (q >> 1)
    const int q = lane >> 4, qp = (lane & 15) >> 2, pp = lane & 3;
    const unsigned posb = (unsigned)(qp + 4 * (q & 1) + 16 * (q >> 1)) * T2_PITCH + 8u * pp;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned aq0 = lds0 + posb + 128u * wk, ap0 = lds0 + posb + 512u + 256u * wg;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    union U { bf16x8 v; s16x4 h[2]; };
    U af[4], bf[8];
    int cs_j[2] = {-1, -1};  // fused bias gradient: this wave's column units, as in gemm_tn_kernel
    if (p.colsum != nullptr) {
        const int cs = p.tiles_k < 16 ? p.tiles_k : 16;
        int seen = 0;
        for (int j = 0; j < 8; ++j) {
            if ((8 * wg + j) % cs != tile_k) continue;
            if ((seen & 3) == wk) cs_j[seen >> 2] = j;
            ++seen;
        }
    }
    const bool do_colsum = cs_j[0] >= 0;
    f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

#define TN_BARRIER()                             \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

#pragma unroll
    for (int d = 0; d < T2_D; ++d)
        if (d < nk) stage(d, d);
    if (nk >= T2_D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (T2_D - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TN_BARRIER();
    if (wg == 1) TN_BARRIER();

    int slot = 0, pslot = T2_D;
    for (int v = 0; v < nk; ++v) {
        if (!(BSI_ABL(p.abl, 4)) || v == 0) {  // (BSI_TN_ABL: laboratory flags, as in gemm_tn_kernel)
            const unsigned so = (unsigned)slot * T2_SLOT;
            const unsigned aq = aq0 + so, ap = ap0 + so;
            constexpr int HB = 8 * T2_PITCH;  // rows + 4 = 8 positions further
            af[0].h[0] = tn_tr<0>(aq);   af[0].h[1] = tn_tr<HB>(aq);
            af[1].h[0] = tn_tr<32>(aq);  af[1].h[1] = tn_tr<HB + 32>(aq);
            af[2].h[0] = tn_tr<64>(aq);  af[2].h[1] = tn_tr<HB + 64>(aq);
            af[3].h[0] = tn_tr<96>(aq);  af[3].h[1] = tn_tr<HB + 96>(aq);
            bf[0].h[0] = tn_tr<0>(ap);   bf[0].h[1] = tn_tr<HB>(ap);
            bf[1].h[0] = tn_tr<32>(ap);  bf[1].h[1] = tn_tr<HB + 32>(ap);
            bf[2].h[0] = tn_tr<64>(ap);  bf[2].h[1] = tn_tr<HB + 64>(ap);
            bf[3].h[0] = tn_tr<96>(ap);  bf[3].h[1] = tn_tr<HB + 96>(ap);
            bf[4].h[0] = tn_tr<128>(ap); bf[4].h[1] = tn_tr<HB + 128>(ap);
            bf[5].h[0] = tn_tr<160>(ap); bf[5].h[1] = tn_tr<HB + 160>(ap);
            bf[6].h[0] = tn_tr<192>(ap); bf[6].h[1] = tn_tr<HB + 192>(ap);
            bf[7].h[0] = tn_tr<224>(ap); bf[7].h[1] = tn_tr<HB + 224>(ap);
        }
        const bool split_issue = NL < 4 && v + T2_D < nk;
        if (split_issue) {
            // NL instructions now; outstanding then: stages v+1, v+2 (8) + NL -> stage v+1 has landed when at most 4 + NL are left
#pragma unroll
            for (int qq = 0; qq < NL; ++qq) stage_one(v + T2_D, pslot, qq);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + (NL < 4 ? NL : 0)) : "memory");
        } else if (v + T2_D < nk && !(BSI_ABL(p.abl, 2))) {
            stage(v + T2_D, pslot);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (T2_D - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0) ; data of %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15 %16 %17 %18 %19 %20 %21 %22 %23"
                     : "+v"(af[0].h[0]), "+v"(af[0].h[1]), "+v"(af[1].h[0]), "+v"(af[1].h[1]), "+v"(af[2].h[0]), "+v"(af[2].h[1]),
                       "+v"(af[3].h[0]), "+v"(af[3].h[1]), "+v"(bf[0].h[0]), "+v"(bf[0].h[1]), "+v"(bf[1].h[0]), "+v"(bf[1].h[1]),
                       "+v"(bf[2].h[0]), "+v"(bf[2].h[1]), "+v"(bf[3].h[0]), "+v"(bf[3].h[1]), "+v"(bf[4].h[0]), "+v"(bf[4].h[1]),
                       "+v"(bf[5].h[0]), "+v"(bf[5].h[1]), "+v"(bf[6].h[0]), "+v"(bf[6].h[1]), "+v"(bf[7].h[0]), "+v"(bf[7].h[1]));
        if (mbeg + v * 32 + 32 > mend) {
            // ragged tail: rows >= mend were clamped to a valid address; zero their contribution (element e of a fragment
            // is contraction row 8q + e)
            const int valid = mend - (mbeg + v * 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (8 * q + e >= valid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i].v[e] = (__bf16)0.0f;
                }
            }
        }
        bf16x8 onesf;
        {
            const int valid = mend - (mbeg + v * 32);  // >= 32 except in the ragged tail stage
#pragma unroll
            for (int e = 0; e < 8; ++e) onesf[e] = (8 * q + e < valid) ? (__bf16)1.0f : (__bf16)0.0f;
        }
        // priority goes to the LOAD phase (the phase that sets the interval: 24 transposed reads + its share of the DMA issue), not to
        // the MFMA phase as in the forward GEMM: +1.5-2.5 % (tools/tn_bench.py, profiles/r3/tn_ab.txt)
        __builtin_amdgcn_s_setprio(0);
        TN_BARRIER();
        if (!(BSI_ABL(p.abl, 8))) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i0 = 0; i0 < 4; ++i0) {  // boustrophedon: one operand changes per MFMA (power; gemm_bf16.hip k64r kernel)
                    const int i = (j & 1) ? 3 - i0 : i0;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i].v, bf[j].v, acc[i][j], 0, 0, 0);
                }
                if (NL < 4 && split_issue) {  // the other 4 - NL instructions, inside the MFMA phase (placement: template parameter)
                    if (j == JA) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage_one(v + T2_D, pslot, NL);
                        if constexpr (TOGETHER) {
#pragma unroll
                            for (int qq = NL + 1; qq < 4; ++qq) stage_one(v + T2_D, pslot, qq);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!TOGETHER && NL < 3 && j == JB) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int qq = NL + 1; qq < 4; ++qq) stage_one(v + T2_D, pslot, qq);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (do_colsum) {  // wave-uniform unit indices: branches, not register selects (selecting the fragment with 28
                                            // scalar-condition vector selects instead of the compare-and-branch chain measured 2-4 % SLOWER)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (cs_j[u] < 0) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (cs_j[u] == j) accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onesf, bf[j].v, accb[u], 0, 0, 0);
            }
        }
        TN_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        slot = (slot == T2_R - 1) ? 0 : slot + 1;
        pslot = (pslot == T2_R - 1) ? 0 : pslot + 1;
    }
    if (wg == 0) TN_BARRIER();
#undef TN_BARRIER

    if (do_colsum && lane < 16) {  // D row 0 (lanes 0..15, register 0) holds the column sums
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (cs_j[u] < 0) continue;
            const int n = n0 + 128 * wg + 16 * cs_j[u] + lane;
            if (n < p.N) p.colsum[(size_t)split * p.colsum_stride + n] = accb[u][0];
        }
    }
    float* out = outp + (size_t)split * slabp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 128 * wg + 16 * j + (lane & 15);
        if (n >= Np) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 64 * wk + 16 * i + 4 * q;
            if (k < p.K) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(out + (size_t)n * p.ldc + k));
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i]
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, size_t slab_stride, int splits, size_t n4,
                                    int accumulate, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 a = accumulate ? reinterpret_cast<const f32x4*>(out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slabs + (size_t)s * slab_stride) + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
        reinterpret_cast<f32x4*>(out)[i] = a;
    }
}

// Same sum for many slabs of a small tensor (conv weight gradients: 85 slabs of 147k floats): one output per thread would
// leave most CUs idle behind 85 dependent-latency loads, so 4 threads share an output (slabs s = g, g+4, ... each, 4 loads in
// flight) and their partial sums are added in fixed order through LDS -- still deterministic.
__global__ __launch_bounds__(256) void reduce_slabs_par_kernel(const float* __restrict__ slabs, size_t slab_stride, int splits, size_t n4,
                                                               int accumulate, float* __restrict__ out) {
    __shared__ f32x4 part[3][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = g;
        for (; s + 12 < splits; s += 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slabs + (size_t)(s + 4 * u) * slab_stride) + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += v[u][e];
        }
        for (; s < splits; s += 4) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slabs + (size_t)s * slab_stride) + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
    }
    if (g > 0) part[g - 1][o] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += part[k][o][e];
        if (accumulate) {
            const f32x4 prev = reinterpret_cast<const f32x4*>(out)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += prev[e];
        }
        reinterpret_cast<f32x4*>(out)[i] = a;
    }
}

// Two such reductions in one launch (a convolution's weight slabs and its bias slabs: the second job is a handful of workgroups
// that would otherwise cost a launch of their own).  Blocks [0, gA) work on job A, the rest on job B.
struct ReduceJob {
    const float* slabs;
    size_t slab_stride;
    int splits;
    size_t n4;
    float* out;
};
__global__ __launch_bounds__(256) void reduce_slabs_par2_kernel(ReduceJob ja, ReduceJob jb, int gA, int accumulate) {
    __shared__ f32x4 part[3][64];
    const bool second = (int)blockIdx.x >= gA;
    const ReduceJob j = second ? jb : ja;
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)(second ? blockIdx.x - gA : blockIdx.x) * 64 + o;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < j.n4) {
        int s = g;
        for (; s + 12 < j.splits; s += 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(j.slabs + (size_t)(s + 4 * u) * j.slab_stride) + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += v[u][e];
        }
        for (; s < j.splits; s += 4) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(j.slabs + (size_t)s * j.slab_stride) + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
    }
    if (g > 0) part[g - 1][o] = a;
    __syncthreads();
    if (g == 0 && i < j.n4) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += part[k][o][e];
        if (accumulate) {
            const f32x4 prev = reinterpret_cast<const f32x4*>(j.out)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += prev[e];
        }
        reinterpret_cast<f32x4*>(j.out)[i] = a;
    }
}

// column sums of a bf16 [M, ld] matrix (bias gradients): out[n] (+)= sum_m Y[m, n].
// A block owns a 256-column strip and a row range: 32 threads x 16 B cover the strip, 8 row lanes run in parallel and
// each walks its rows with 4 independent 16-B loads in flight; partial sums go to a slab per row split and are summed
// by reduce_slabs_kernel (deterministic, no atomics).  HBM-bound: one pass over Y.
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ Y, int M, int Ncols, int ld,
                                                     int rows_per_split, float* __restrict__ slabs, size_t slab_stride) {
    __shared__ float red[8][256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + tx * 8;
    const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < Ncols) {
        const __bf16* base = Y + c0;
        int m = m0 + ty;
        for (; m + 24 < m1; m += 32) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)(m + 8 * u) * ld));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[2 * e] += __uint_as_float(w[u][e] << 16);
                    a[2 * e + 1] += __uint_as_float(w[u][e] & 0xffff0000u);
                }
        }
        for (; m < m1; m += 8) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(base + (size_t)m * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[2 * e] += __uint_as_float(w[e] << 16);
                a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = a[e];
    __syncthreads();
    const int c = threadIdx.x;  // one column per thread
    if (blockIdx.x * 256 + c < Ncols) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][c];
        slabs[(size_t)blockIdx.y * slab_stride + blockIdx.x * 256 + c] = t;
    }
}

}  // namespace

// out[0..n) (+)= sum over `splits` slabs (n a multiple of 4); shared with conv_wgrad.hip
int bsi_reduce_slabs_launch(const float* slabs, size_t slab_stride, int splits, size_t n, int accumulate, float* out,
                            hipStream_t s) {
    const size_t n4 = n / 4;
    if (splits >= 8 && n4 <= (size_t)256 * 1024) {  // few outputs, many slabs: 4 threads per output
        hipLaunchKernelGGL(reduce_slabs_par_kernel, dim3((int)((n4 + 63) / 64)), dim3(256), 0, s, slabs, slab_stride, splits, n4, accumulate, out);
        BSI_CHECK_LAUNCH("bsi_reduce_slabs");
        return BSI_OK;
    }
    size_t g = (n4 + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((int)g), dim3(256), 0, s, slabs, slab_stride, splits, n4, accumulate, out);
    BSI_CHECK_LAUNCH("bsi_reduce_slabs");
    return BSI_OK;
}

// both reductions of a weight-gradient launch (weights: nA floats, bias: nB floats) -- ONE launch when both take the
// many-slabs form, two otherwise; same summation order as bsi_reduce_slabs_launch either way
int bsi_reduce_slabs2_launch(const float* slabsA, size_t strideA, size_t nA, float* outA, const float* slabsB, size_t strideB, size_t nB,
                             float* outB, int splits, int accumulate, hipStream_t s) {
    const size_t a4 = nA / 4, b4 = nB / 4;
    if (splits >= 8 && a4 <= (size_t)256 * 1024 && b4 <= (size_t)256 * 1024) {
        const int gA = (int)((a4 + 63) / 64), gB = (int)((b4 + 63) / 64);
        hipLaunchKernelGGL(reduce_slabs_par2_kernel, dim3(gA + gB), dim3(256), 0, s, ReduceJob{slabsA, strideA, splits, a4, outA},
                           ReduceJob{slabsB, strideB, splits, b4, outB}, gA, accumulate);
        BSI_CHECK_LAUNCH("bsi_reduce_slabs2");
        return BSI_OK;
    }
    const int rc = bsi_reduce_slabs_launch(slabsB, strideB, splits, nB, accumulate, outB, s);
    return rc ? rc : bsi_reduce_slabs_launch(slabsA, strideA, splits, nA, accumulate, outA, s);
}

extern "C" size_t bsi_gemm_tn_workspace_bytes(int M, int N, int K) {
    // worst case number of splits is 16; the fused bias gradient needs 16 x N more floats behind the tile slabs
    return (size_t)16 * (size_t)N * (size_t)K * sizeof(float) + (size_t)16 * (size_t)((N + 3) / 4 * 4) * sizeof(float);
}

static int tn_splits(int M, int N, int K, int num_cus) {
    const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
    const int max_by_m = M / 512 > 0 ? M / 512 : 1;  // keep at least 16 K steps per workgroup
    // one workgroup per CU at a time (160 KB of LDS): time ~ rounds(s) / s with rounds = ceil(tiles * s / CUs); the smallest s
    // among the best (rounding num_cus / tiles UP put the qkv weight gradient, 48 tiles, into 288 workgroups = two rounds)
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 16 && s <= max_by_m; ++s) {
        const double t = (double)((tiles * s + num_cus - 1) / num_cus) / s;
        if (t < best_t * 0.9) { best_t = t; best = s; }  // more splits only for a real gain: every split is a slab to write and reduce
    }
    return best;
}

static int gemm_tn_impl(const void* P, int ldp, const void* Q, int ldq, int M, int N, int K, float* out, int ldc, float* colsum_out,
                       int accumulate, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(P && Q && out && M > 0 && N > 0 && K > 0, "bsi_gemm_tn_bf16: bad args");
    BSI_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && ldp % 8 == 0 && ldq % 8 == 0 && ldp >= N && ldq >= K,
                  "bsi_gemm_tn_bf16: N=%d K=%d ldp=%d ldq=%d must be multiples of 8", N, K, ldp, ldq);
    BSI_CHECK_ARG(ldc == K, "bsi_gemm_tn_bf16: output must be dense (ldc == K)");
    BSI_CHECK_ARG(K % 4 == 0, "bsi_gemm_tn_bf16: K %% 4");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int cus = compute_cus();  // one workgroup per CU per round; CUs reserved for a communication kernel are left out
    TnParams p{};
    p.P = reinterpret_cast<const __bf16*>(P);
    p.Q = reinterpret_cast<const __bf16*>(Q);
    p.M = M; p.N = N; p.K = K; p.ldp = ldp; p.ldq = ldq; p.ldc = ldc;
    p.tiles_n = (N + 255) / 256;
    p.tiles_k = (K + 255) / 256;
    p.splits = tn_splits(M, N, K, cus);
    const int per = (M + p.splits - 1) / p.splits;
    p.m_per_split = (per + 31) / 32 * 32;
    p.splits = (M + p.m_per_split - 1) / p.m_per_split;
    const bool direct = p.splits == 1 && !accumulate;
    BSI_CHECK_ARG(direct || workspace, "bsi_gemm_tn_bf16: workspace required");
    p.out = direct ? out : reinterpret_cast<float*>(workspace);
    p.slab_stride = (size_t)N * ldc;
    const size_t n4 = (size_t)(N + 3) / 4 * 4;
    float* cs_slabs = reinterpret_cast<float*>(workspace) + (size_t)16 * N * K;
    if (colsum_out) {
        p.colsum = direct ? colsum_out : cs_slabs;
        p.colsum_stride = n4;
    }
    static const int abl = [] { const char* e = getenv("BSI_TN_ABL"); return e ? atoi(e) : 0; }();
    p.xcd_map = !(abl & 1);
    p.abl = abl;
    if (abl & 16) {  // BSI_TN_ABL & 16: the round-2 kernel (A/B partner)
        set_max_lds(reinterpret_cast<const void*>(gemm_tn_kernel), T_R * T_SLOT);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tiles_n * p.tiles_k * p.splits), dim3(512), T_R * T_SLOT, s, p);
    } else {
        auto go = [&](auto kern) {
            set_max_lds(reinterpret_cast<const void*>(kern), T2_R * T2_SLOT);
            hipLaunchKernelGGL(kern, dim3(p.tiles_n * p.tiles_k * p.splits), dim3(512), T2_R * T2_SLOT, s, p);
        };
        // DMA placement.  Without the fused bias gradient: two of a stage's four instructions per wave in the load phase, two behind
        // the second group of four MFMAs -- with all four in the load phase a wave sits in four back-to-back issues while the
        // addresser queue is full (the memory system delivers about 30 B/clk per CU to this stream: a DMA-only run takes 594 us for
        // fc1's 8.6 GB) before it can start its 24 fragment reads: +4-10 % (qkv 970 -> 1010, out 820 -> 880, fc1 1134 -> 1169, fc2
        // 1123 -> 1220 TFLOP/s on one box).  WITH the bias gradient (every Linear of the DiT) the MFMA phase carries the unit MFMAs and
        // their compare chain and the split measured 3-6 % SLOWER than all four in the load phase on three boxes of the pool and 8 %
        // faster on a fourth (a slower one): no clear winner, those launches keep all four in the load phase.  (The unit MFMAs moved
        // into the load phase instead: 10-12 % slower.)
        // Placements tried: behind groups 1 / 3 / 5 / 7, one behind 2 and one behind 6, 3 + 1, 1 + 3, 1 + 1 + 2 (profiles/r3/tn_ab.txt).
        if ((abl & 32) || (p.colsum != nullptr && !(abl & 64))) go(gemm_tn2_kernel<0>);  // BSI_TN_ABL & 64: split also with the bias gradient
        else go(gemm_tn2_kernel<0x1012>);
    }
    BSI_CHECK_LAUNCH("bsi_gemm_tn_bf16");
    if (!direct) {
        if (colsum_out) {
            int rc = bsi_reduce_slabs_launch(cs_slabs, n4, p.splits, n4 <= (size_t)N ? n4 : (size_t)N / 4 * 4, accumulate, colsum_out, s);
            if (rc) return rc;
        }
        return bsi_reduce_slabs_launch(reinterpret_cast<const float*>(workspace), p.slab_stride, p.splits, (size_t)N * ldc,
                                       accumulate, out, s);
    }
    return BSI_OK;
}

extern "C" int bsi_gemm_tn_pair_bf16(const void* P1, int ldp1, const void* Q1, int N1, float* out1, const void* P2, int ldp2, const void* Q2,
                                     int N2, float* out2, int ldq, int M, int K, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(P1 && Q1 && out1 && P2 && Q2 && out2 && workspace && M > 0 && N1 > 0 && N2 > 0 && K > 0, "bsi_gemm_tn_pair_bf16: bad args");
    BSI_CHECK_ARG(N1 % 256 == 0 && N2 % 8 == 0 && K % 8 == 0 && ldp1 % 8 == 0 && ldp2 % 8 == 0 && ldq % 8 == 0 && ldp1 >= N1 && ldp2 >= N2 && ldq >= K,
                  "bsi_gemm_tn_pair_bf16: N1=%d must be a multiple of 256 (whole tiles), N2=%d K=%d and the leading dimensions multiples of 8", N1, N2, K);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int cus = compute_cus();
    TnParams p{};
    p.P = reinterpret_cast<const __bf16*>(P1);
    p.Q = reinterpret_cast<const __bf16*>(Q1);
    p.P2 = reinterpret_cast<const __bf16*>(P2);
    p.Q2 = reinterpret_cast<const __bf16*>(Q2);
    p.M = M; p.N = N1; p.N2 = N2; p.K = K; p.ldp = ldp1; p.ldp2 = ldp2; p.ldq = ldq; p.ldc = K;
    p.pair_tiles = N1 / 256;
    p.tiles_n = p.pair_tiles + (N2 + 255) / 256;
    p.tiles_k = (K + 255) / 256;
    p.splits = tn_splits(M, p.tiles_n * 256, K, cus);
    const int per = (M + p.splits - 1) / p.splits;
    p.m_per_split = (per + 31) / 32 * 32;
    p.splits = (M + p.m_per_split - 1) / p.m_per_split;
    // slabs: problem 1's first, problem 2's behind them (bsi_gemm_tn_workspace_bytes(M, N1 + N2, K) covers both)
    p.out = reinterpret_cast<float*>(workspace);
    p.slab_stride = (size_t)N1 * K;
    p.out2 = p.out + (size_t)p.splits * p.slab_stride;
    p.slab_stride2 = (size_t)N2 * K;
    static const int abl = [] { const char* e = getenv("BSI_TN_ABL"); return e ? atoi(e) : 0; }();
    p.xcd_map = !(abl & 1);
    p.abl = abl;
    auto kern = gemm_tn2_kernel<0x1012>;
    set_max_lds(reinterpret_cast<const void*>(kern), T2_R * T2_SLOT);
    hipLaunchKernelGGL(kern, dim3(p.tiles_n * p.tiles_k * p.splits), dim3(512), T2_R * T2_SLOT, s, p);
    BSI_CHECK_LAUNCH("bsi_gemm_tn_pair_bf16");
    if (p.splits == 1) {  // (never at the engine's sizes) one slab each: plain copies through the reduction
        const int rc = bsi_reduce_slabs_launch(p.out, p.slab_stride, 1, (size_t)N1 * K, 0, out1, s);
        return rc ? rc : bsi_reduce_slabs_launch(p.out2, p.slab_stride2, 1, (size_t)N2 * K, 0, out2, s);
    }
    return bsi_reduce_slabs2_launch(p.out, p.slab_stride, (size_t)N1 * K, out1, p.out2, p.slab_stride2, (size_t)N2 * K, out2, p.splits, 0, s);
}

extern "C" int bsi_gemm_tn_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int N, int K, float* out, int ldc,
                                int accumulate, void* workspace, bsi_stream_t stream) {
    return gemm_tn_impl(P, ldp, Q, ldq, M, N, K, out, ldc, nullptr, accumulate, workspace, stream);
}

extern "C" int bsi_gemm_tn_bias_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int N, int K, float* out, int ldc,
                                     float* colsum_out, int accumulate, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(colsum_out && workspace && N % 4 == 0, "bsi_gemm_tn_bias_bf16: bad args");
    return gemm_tn_impl(P, ldp, Q, ldq, M, N, K, out, ldc, colsum_out, accumulate, workspace, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Column sums of fp32 row tables (bias gradients whose per-slab partial rows the producers of dY write, round 4): out[c] = sum_r
// src[r][c], rows added in a fixed order.  One kernel, run twice: stage 1 sums chunks of 64 rows into partial rows, stage 2 (the same
// kernel on the partial rows: one chunk) writes the result.  Up to four tables per launch.
namespace {
struct ColsumJobs {
    const float* src[4];
    float* dst[4];
    int rows[4], cols[4], ld[4], dst_ld[4];
    int blk0[5];  // first block of job j (blk0[njobs] = grid)
    int cblocks[4];  // column blocks (of 256 columns) of job j
    int njobs;
    int chunk;  // rows per block: CS_CHUNK in stage 1, all of them in stage 2
};
constexpr int CS_CHUNK = 64;

__global__ __launch_bounds__(256) void colsum_rows_kernel(const ColsumJobs jb) {
    __shared__ f32x4 part[3][64];  // (the three other row lanes of a column quad)
    int j = 0;
    while (j + 1 < jb.njobs && (int)blockIdx.x >= jb.blk0[j + 1]) ++j;
    const int lb = blockIdx.x - jb.blk0[j];
    const int cb = lb % jb.cblocks[j], chunk = lb / jb.cblocks[j];
    const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = cb * 256 + q * 4;
    const int r0 = chunk * jb.chunk, r1 = jb.chunk > 0 ? min(jb.rows[j], r0 + jb.chunk) : jb.rows[j];
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < jb.cols[j]) {
        const float* s = jb.src[j] + c;
        const size_t ld = (size_t)jb.ld[j];
        int r = r0 + g;
        for (; r + 12 < r1; r += 16) {  // four independent loads in flight
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(s + (size_t)(r + 4 * u) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += v[u][e];
        }
        for (; r < r1; r += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(s + (size_t)r * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
    }
    if (g > 0) part[g - 1][q] = a;
    __syncthreads();
    if (g == 0 && c < jb.cols[j]) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += part[k][q][e];
        *reinterpret_cast<f32x4*>(jb.dst[j] + (size_t)chunk * jb.dst_ld[j] + c) = a;
    }
}
}  // namespace

extern "C" size_t bsi_colsum_rows_scratch_bytes(int rows, int cols) {  // one partial row per chunk of 64 rows
    return rows > 0 && cols > 0 ? (size_t)((rows + CS_CHUNK - 1) / CS_CHUNK) * (size_t)cols * sizeof(float) : 0;
}

extern "C" int bsi_colsum_rows_f32(const bsi_colsum_job* jobs, int njobs, void* scratch, bsi_stream_t stream) {
    BSI_CHECK_ARG(jobs && njobs >= 1 && njobs <= 4 && scratch, "bsi_colsum_rows_f32: 1..4 jobs and a scratch buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ColsumJobs a{}, b{};
    a.njobs = b.njobs = njobs;
    float* sc = reinterpret_cast<float*>(scratch);
    int ga = 0, gb = 0;
    for (int j = 0; j < njobs; ++j) {
        const bsi_colsum_job& jo = jobs[j];
        BSI_CHECK_ARG(jo.src && jo.out && jo.rows > 0 && jo.cols > 0 && jo.cols % 4 == 0 && jo.ld % 4 == 0 && jo.ld >= jo.cols,
                      "bsi_colsum_rows_f32: job %d: rows=%d cols=%d ld=%d", j, jo.rows, jo.cols, jo.ld);
        const int chunks = (jo.rows + CS_CHUNK - 1) / CS_CHUNK, cblocks = (jo.cols + 255) / 256;
        a.src[j] = jo.src; a.rows[j] = jo.rows; a.cols[j] = jo.cols; a.ld[j] = jo.ld; a.cblocks[j] = cblocks; a.blk0[j] = ga;
        a.dst[j] = chunks > 1 ? sc : jo.out;  // a single chunk is the result already
        a.dst_ld[j] = jo.cols;
        ga += cblocks * chunks;
        b.src[j] = sc; b.rows[j] = chunks > 1 ? chunks : 0; b.cols[j] = jo.cols; b.ld[j] = jo.cols; b.cblocks[j] = cblocks; b.blk0[j] = gb;
        b.dst[j] = jo.out; b.dst_ld[j] = jo.cols;
        gb += chunks > 1 ? cblocks : 0;
        sc += (size_t)chunks * jo.cols;
    }
    a.blk0[njobs] = ga; b.blk0[njobs] = gb;
    a.chunk = CS_CHUNK;
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(ga), dim3(256), 0, s, a);
    BSI_CHECK_LAUNCH("bsi_colsum_rows_f32");
    if (gb > 0) {
        // stage 2 walks only the jobs that had more than one chunk: compact the table
        ColsumJobs c{};
        int n = 0, g2 = 0;
        for (int j = 0; j < njobs; ++j)
            if (b.rows[j] > 0) {
                c.src[n] = b.src[j]; c.dst[n] = b.dst[j]; c.rows[n] = b.rows[j]; c.cols[n] = b.cols[j]; c.ld[n] = b.ld[j];
                c.dst_ld[n] = b.dst_ld[j]; c.cblocks[n] = b.cblocks[j]; c.blk0[n] = g2;
                g2 += b.cblocks[j];
                ++n;
            }
        c.blk0[n] = g2; c.njobs = n;
        c.chunk = 0;
        hipLaunchKernelGGL(colsum_rows_kernel, dim3(g2), dim3(256), 0, s, c);
        BSI_CHECK_LAUNCH("bsi_colsum_rows_f32(stage 2)");
    }
    return BSI_OK;
}

extern "C" size_t bsi_colsum_workspace_bytes(int N) { return (size_t)64 * (size_t)((N + 3) / 4 * 4) * sizeof(float); }

extern "C" int bsi_colsum_bf16(const void* Y, int ld, int M, int N, float* out, int accumulate, void* workspace,
                               bsi_stream_t stream) {
    BSI_CHECK_ARG(Y && out && workspace && M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0 && ld >= N, "bsi_colsum_bf16: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int splits = M / 256;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const int per = (M + splits - 1) / splits;
    splits = (M + per - 1) / per;
    dim3 grid((N + 255) / 256, splits);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, reinterpret_cast<const __bf16*>(Y), M, N, ld, per,
                       reinterpret_cast<float*>(workspace), (size_t)N);
    BSI_CHECK_LAUNCH("bsi_colsum_bf16");
    size_t g = ((size_t)N / 4 + 255) / 256;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((int)g), dim3(256), 0, s, reinterpret_cast<const float*>(workspace),
                       (size_t)N, splits, (size_t)N / 4, accumulate, out);
    BSI_CHECK_LAUNCH("bsi_colsum_bf16(reduce)");
    return BSI_OK;
}
