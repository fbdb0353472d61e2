// 3x3 / 1x1 convolution (stride 1, zero padding) as an IMPLICIT GEMM on bf16 MFMA for gfx950.
// Replaces nn.Conv2d on the VDM-UNet path of the reference (bsi/models/vdm_unet.py:71-72,
// bsi/nn/residual_block.py:40-48, bsi/nn/attention.py:29-30), together with the elementwise ops torch runs after it.
//
// Activations are NHWC bf16 ([B*H*W pixels][C channels]); weights are pre-arranged [C_out][tap][C_in] so that the GEMM
// K index is (tap, channel).  No im2col buffer exists: the A tile of K step v (32 channels of one filter tap) is
// fetched by LDS-DMA (buffer_load ... lds) straight from the input pixel (y+dy, x+dx) of each output pixel; lanes whose
// tap falls outside the image use an out-of-range buffer offset, for which the DMA writes zeros.  An optional second source adds 1x1 "skip" K steps (the 1x1 conv on the residual path
// of the up blocks, residual_block.py:40,63, folded into the second 3x3 conv's accumulation).
//
// Schedule = the persistent deep-ring ping-pong of gemm_bf16.hip variant 6 with tile 512 (pixels) x 128 (C_out):
// C_out is 128 everywhere in the UNet, so the tile is made tall instead of wide; waves keep the 128 x 64 sub-tile.
// LDS: a ring of 4 stages of (512 + 128) rows x 64 B = 160 KB; the epilogue needs no scratch (DPP row-pair exchange).
#include <type_traits>

#include "common.h"

namespace {

struct ConvParams {
    const __bf16* A;      // NHWC input [B*H*W, Cin]
    const __bf16* A2;     // optional second source [B*H*W, Cin2] for the 1x1 skip K steps
    const __bf16* W;      // [N][taps*Cin + Cin2]
    const float* bias;    // [N]
    void* out;            // bf16 or fp32 [M, ldo]
    const float* film;    // FILM: [rows][2N] = (scale, shift) per image row
    const float* resid;   // BIAS_RESID_F32: fp32 [M, ldo] or null
    int film_rows, film_stride;
    int M, N, Cin, Cin2, taps, H, Wd, ldo;
    int K;                // taps*Cin + Cin2
    int tiles_m, tiles_n;
    int abl;              // experiments (bsi_conv_set_ablation): 1 no DMA, 2 no MFMA, 4 no epilogue, 8 every pixel row out of range, 16 no fragment reads
    int stagger;          // laboratory (BSI_CONV_STAGGER = sets * 1024 + step): workgroup set (blockIdx / 8) % sets starts set * step * 64 clocks late
    float* gn_part;       // BIAS_RESID_F32_GN: [M/128][N/4][2] = (mean, M2) of every 128-pixel x 4-channel block of the output
};

constexpr int C_BM = 512, C_BN = 128, C_RB = 64, C_R = 4, C_D = C_R - 1;
constexpr int C_SLOT = (C_BM + C_BN) * C_RB;  // 40 KB
constexpr int C_ABYTES = C_BM * C_RB;
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;   // buffer offset beyond num_records: the LDS-DMA writes zeros for that lane
constexpr unsigned BUF_RECORDS = 0x80000000u;  // every valid offset is below 2 GB (checked by the launcher)

// FILM_SILU_BF16: every wave's 128 rows lie in one image (H*W % 128 == 0, coefficients once per tile, applied in place);
// FILM_ROWS_SILU_BF16: any image size, coefficients looked up per row (ring kernel only)
enum { CEPI_BIAS_BF16 = 0, CEPI_FILM_SILU_BF16 = 1, CEPI_BIAS_RESID_F32 = 2, CEPI_BIAS_RESID_F32_GN = 3, CEPI_FILM_ROWS_SILU_BF16 = 4 };

// Padding code of a pixel: which of the four image borders it touches.  A tap (dy, dx) reads outside the image exactly
// when (code & tap_mask(dy, dx)) != 0.
__device__ __forceinline__ int border_code(int y, int x, int H, int W) {
    return (y == 0 ? 1 : 0) | (y == H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == W - 1 ? 8 : 0);
}
__device__ __forceinline__ int tap_mask(int dy, int dx) { return (dy < 0 ? 1 : 0) | (dy > 0 ? 2 : 0) | (dx < 0 ? 4 : 0) | (dx > 0 ? 8 : 0); }

// Store a wave's 16-row x 64-column bf16 block whose lane (rho, qd) holds row rho, columns 16 qd .. 16 qd + 15 as (w0 = first
// 8 columns, w1 = last 8): lanes rho and rho^1 swap one half (DPP quad permute), after which the 8 lanes of a row pair write
// one complete 128-B line per store instruction (even rows, then odd rows).  No LDS.
__device__ __forceinline__ void store_rows_dpp(__bf16* out, int ldo, int M, int m0, int nb, int rho, u32x4 w0, u32x4 w1) {
    const bool odd = rho & 1;
    u32x4 recv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned send = odd ? w0[e] : w1[e];
        recv[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xf, 0xf, true);
    }
    const u32x4 st_even = odd ? recv : w0, st_odd = odd ? w1 : recv;
    const int m_even = m0 + (rho & ~1);
    __bf16* dst = out + (size_t)m_even * ldo + nb + (odd ? 8 : 0);
    if (m_even < M) __builtin_nontemporal_store(st_even, reinterpret_cast<u32x4*>(dst));
    if (m_even + 1 < M) __builtin_nontemporal_store(st_odd, reinterpret_cast<u32x4*>(dst + ldo));
}

// FiLM + SiLU (residual_block.py:21-24,44-46: addcmul(shift, scale + 1, y), then SiLU) IN PLACE on a wave's accumulators when its
// 128 rows lie in one image (H*W % 128 == 0): the (scale, shift) of 4 columns are loaded, applied to the 8 rows, and dropped
// before the next 4 -- nothing but the accumulators stays live (the coefficient arrays of the earlier form cost 160 spilled
// registers in the ring kernel, some of them inside the K loop).
template <int TM>
__device__ __forceinline__ void film_silu_inplace(const ConvParams& p, f32x4 (&acc)[4][TM], int mw0, int nb, int HW) {
    const int m0w = mw0 < p.M ? mw0 : p.M - 1;
    asm volatile("" : "+v"(nb));  // keeps the lane's column offsets out of loop-invariant code motion (they would live across the K loop)
    const float* fr = p.film + (size_t)((m0w / HW) % p.film_rows) * p.film_stride + nb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 sc = *reinterpret_cast<const f32x4*>(fr + 4 * i);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(fr + p.N + 4 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] += 1.0f;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = silu_f(__fmaf_rn(sc[r], acc[i][j][r], sh[r]));
            if (j & 1) __builtin_amdgcn_sched_barrier(0);  // a few independent exp / rcp chains at a time, not all 128
        }
    }
}

// fp32 epilogue of a wave's 128-row x 64-column block (both kernels): lane (rho, qd) holds in acc[i][j] 4 consecutive floats
// (columns nb0 + 16 qd + 4 i ..) of row mw0 + 16 j + rho.  A 4 x 4 transpose across the four 16-lane groups (two rounds of gfx950
// lane-group swaps) makes the 4 lanes of a row write 64 contiguous bytes per store instruction instead of four 16-B pieces 64 B
// apart; the residual is read in the same pattern.
// GN: the block's GroupNorm partial statistics leave with it.  After the transpose a lane's float4 number s4 is exactly one
// 4-channel unit (columns nb0 + 16 s4 + 4 qd ..) -- one group of GroupNorm(32) at 128 channels, half a group of the 256-channel
// concatenation -- so the lane sums its 8 rows, the 16 lanes of a group are added by DPP, and (mean, M2) of the 128 x 4 block go
// to gn_part[(mw0 / 128) * (N / 4) + unit]; bsi_groupnorm_apply_nhwc merges the blocks of an image in a fixed order (Chan et
// al.).  Launcher: M % 128 == 0 and N % 64 == 0, so the row / column guards are wave-uniform here.  4 more store instructions
// per wave (NSTORE of the kernels' vmcnt allowances).
template <bool GN, int TM>
__device__ __forceinline__ void store_f32_rows(const ConvParams& p, f32x4 (&acc)[4][TM], int mw0, int nb0, int rho, int qd) {
    if constexpr (GN) asm volatile("" : "+s"(mw0), "+s"(nb0));  // row addresses depend on the tile only: keep them out of the K loop
    const int col = nb0 + 4 * qd;
    if constexpr (GN) {
        // The launcher guarantees whole blocks (M % 128 == 0, N % 64 == 0).  With a residual (every ResidualBlock's second
        // convolution) its rows are fetched TWO row blocks ahead by inline-asm loads with hand-counted waits: through ordinary loads
        // hipcc waits vmcnt(0) in front of every add (it assumes loads and stores may complete out of order with each other), i.e.
        // for the store of the previous 64 bytes -- 32 serial memory round trips per wave and tile.  Issue order: L0 L1 | block 0:
        // wait L0, add, L2, S0 | block 1: wait L1, add, L3, S1 | ... (L, S = 4 instructions each; vmcnt retires in issue order on
        // this hardware, MI355X_MICROARCH.md).
        if (mw0 >= p.M) return;
        // (one loop for both cases, the residual steps under wave-uniform branches: as two separate loops the accumulator array was
        // merged from two control-flow paths and the kernel spilled 190 registers)
        const bool has_res = p.resid != nullptr;
        f32x4 rs[2][4];
        auto load_res = [&](int j, f32x4 (&d)[4]) {
            const float* a = p.resid + (size_t)(mw0 + 16 * j + rho) * p.ldo + col;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                         "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                         : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(a) : "memory");
        };
        if (has_res) {
            load_res(0, rs[0]);
            load_res(1, rs[1]);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            f32x4 r[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
            transpose_lane_groups(r);
            if (has_res) {
                f32x4 (&d)[4] = rs[j & 1];
                if (j == 0) asm volatile("s_waitcnt vmcnt(4) ; data of %0 %1 %2 %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) :: "memory");
                else if (j == 1 || j == TM - 1) asm volatile("s_waitcnt vmcnt(8) ; data of %0 %1 %2 %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(12) ; data of %0 %1 %2 %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) :: "memory");
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[s4][e] += d[s4][e];
                if (j + 2 < TM) load_res(j + 2, d);
            }
            float* o = reinterpret_cast<float*>(p.out) + (size_t)(mw0 + 16 * j + rho) * p.ldo + col;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                *reinterpret_cast<f32x4*>(o + 16 * s4) = r[s4];
                acc[s4][j] = r[s4];  // the stored values stay in the (dead) accumulators for the statistics below
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        f32x4 r[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
        transpose_lane_groups(r);
        const int m = mw0 + 16 * j + rho;
        if (m >= p.M) continue;
        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + col;
        const float* rsd = p.resid ? p.resid + (size_t)m * p.ldo + col : nullptr;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (col + 16 * s4 >= p.N) continue;
            f32x4 v = r[s4];
            if (rsd) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(rsd + 16 * s4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
            *reinterpret_cast<f32x4*>(o + 16 * s4) = v;
        }
    }
    if constexpr (GN) {
        // two passes over the registers: mean, then the squares about it (no cancellation for any mean / std ratio)
        if (mw0 >= p.M) return;
        constexpr float inv_n = 1.0f / (16.0f * TM * 4.0f);
        float mean[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                a += acc[s4][j][0] + acc[s4][j][1];
                b += acc[s4][j][2] + acc[s4][j][3];
            }
            mean[s4] = row16_sum(a + b) * inv_n;
        }
        float* gp = p.gn_part + ((size_t)(mw0 >> 7) * (p.N >> 2) + (nb0 >> 2) + qd) * 2;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const float d0 = acc[s4][j][0] - mean[s4], d1 = acc[s4][j][1] - mean[s4], d2 = acc[s4][j][2] - mean[s4], d3 = acc[s4][j][3] - mean[s4];
                a = __fmaf_rn(d0, d0, a); b = __fmaf_rn(d1, d1, b);
                a = __fmaf_rn(d2, d2, a); b = __fmaf_rn(d3, d3, b);
            }
            const float m2 = row16_sum(a + b);
            if (rho == 0) {
                f32x2 st;
                st[0] = mean[s4];
                st[1] = m2;
                *reinterpret_cast<f32x2*>(gp + 8 * s4) = st;
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void conv_ring_kernel(const ConvParams p) {
    constexpr int TM = 8, NW = 8;
    constexpr bool BF16_OUT = (EPI < CEPI_BIAS_RESID_F32 || EPI == CEPI_FILM_ROWS_SILU_BF16);
    constexpr int NSTORE = BF16_OUT ? 2 * TM : 4 * TM + (EPI == CEPI_BIAS_RESID_F32_GN ? 4 : 0);  // store instructions of one wave's epilogue
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;          // ping-pong group = 256-row half of the tile
    const int wmm = (wave >> 1) & 1;   // 128-row quarter inside the group
    const int wn = wave & 1;           // 64-column half

    const int nwg = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int wpx = (gridDim.x + 7 - xcd) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int lo = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + wl;
    if (tile >= hi) return;
    if (p.stagger) {
        const int n = ((blockIdx.x >> 3) % (p.stagger >> 10)) * (p.stagger & 1023);
        for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);
    }
    const int nk = p.K / 32;
    const int HW = p.H * p.Wd;

    // ---- issue stream.  A stage = 40 wave-instructions (16 rows x 64 B): wave w issues instruction slots q*8 + w, q < 5:
    // slots 0..31 = activation rows (pixels), 32..39 = weight rows.  Pixel rows go through a buffer resource whose base is
    // moved by the tap's (wave-uniform) byte shift, so a lane's offset is a per-tile constant; padding lanes use an
    // out-of-range offset and the DMA writes zeros for them.  Per instruction: and, compare, select.
    const int srow = lane >> 2, spos = lane & 3;
    const int achunk = (spos ^ ((-(srow >> 2)) & 3)) * 16;  // swizzled 16-B chunk of this lane (same for every slot)
    // Per-lane source state of a tile, kept to THREE registers (the K loop holds 176 accumulator / fragment registers, and every
    // further loop-invariant one is a candidate for a spill inside that loop): the pixel row of slot 0 -- slot q lies
    // q * 128 rows further, which goes into the instruction's scalar offset --, the four border codes packed 5 bits each
    // (bit 4 = row beyond M: always out of range), and the weight row offset.
    int mrow;        // pixel row (global) of this lane in slot q = 0
    unsigned codes;  // border_code of the four slots' pixels, 5 bits each
    unsigned woffs;  // byte offset of this lane's weight row (+ chunk) from p.W
    auto set_sources = [&](int t) {
        const int m0 = (t / p.tiles_n) * C_BM, n0 = (t % p.tiles_n) * C_BN;
        mrow = m0 + wave * 16 + srow;
        codes = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mrow + q * NW * 16;
            const int rem = m % HW;
            const int cd = m < p.M ? border_code(rem / p.Wd, rem % p.Wd, p.H, p.Wd) : 16;
            codes |= (unsigned)cd << (5 * q);
        }
        const int rw = wave * 16 + srow;  // slot 32 + wave
        const int c = spos ^ ((-(rw >> 4)) & 3);
        int n = n0 + rw;
        n = n < p.N ? n : p.N - 1;
        woffs = (unsigned)n * (unsigned)(p.K * 2) + c * 16;
    };
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* A2b = reinterpret_cast<const char*>(p.A2);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    int itile = tile, ik = 0, itap = 0, icb = 0, islot = 0;  // next stage: K step, its tap / channel byte offset, ring slot
    auto issue_next = [&]() -> bool {
        if (itile >= hi) return false;
        if (BSI_ABL(p.abl, 1)) { if (++ik == nk) { ik = 0; itile += wpx; } return true; }
        char* base = lds + islot * C_SLOT;
        const bool src2 = itap >= p.taps;
        int dy = 0, dx = 0;
        if (!src2 && p.taps == 9) { dy = itap / 3 - 1; dx = itap - (itap / 3) * 3 - 1; }
        const int rowb = src2 ? p.Cin2 * 2 : p.Cin * 2;
        const long delta = (long)(dy * p.Wd + dx) * rowb + icb;
        const unsigned tmask = ((BSI_ABL(p.abl, 8)) ? 15u : (unsigned)tap_mask(dy, dx)) | 16u;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((src2 ? A2b : Ab) + delta), 0, BUF_RECORDS, 0x00020000);
        if (!(BSI_ABL(p.abl, 64))) {
            const unsigned vo0 = (unsigned)mrow * (unsigned)rowb + achunk;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned vo = ((codes >> (5 * q)) & tmask) ? OOB_OFFSET : vo0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(base + (q * NW + wave) * 1024), 16, vo, q * NW * 16 * rowb, 0, 0);
            }
        }
        if (!(BSI_ABL(p.abl, 32)))
            __builtin_amdgcn_global_load_lds(GLB_PTR(Wb + (size_t)ik * 64 + woffs), LDS_PTR(base + (32 + wave) * 1024), 16, 0, 0);
        icb += 64;
        if (!src2 && icb == p.Cin * 2) { icb = 0; ++itap; }
        if (++ik == nk) {
            ik = 0; itap = 0; icb = 0;
            itile += wpx;
            if (itile < hi) set_sources(itile);
        }
        islot = (islot + 1) & (C_R - 1);
        return true;
    };

    const int rho = lane & 15, qd = lane >> 4;
    const int cc = ((qd ^ ((-(rho >> 2)) & 3)) << 4);
    const int xoff = ((wm * 2 + wmm) * 128 + rho) * C_RB + cc;
    const int woff = C_ABYTES + (wn * 64 + 16 * (rho >> 2) + (rho & 3)) * C_RB + cc;

    f32x4 acc[4][TM];
    bf16x8 wf[4], xf[TM];
#define PHASE_BARRIER()                          \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

    auto init_acc = [&](int t) {  // accumulators start at the bias of their column (lane owns n = nb .. nb+15)
        f32x4 bv[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int nb = (t % p.tiles_n) * C_BN + wn * 64 + 16 * qd;
        if (p.bias && nb < p.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = bv[i];
    };
    auto epilogue = [&](int t) {
        asm volatile("" : "+s"(t));  // keeps the tile-dependent row addresses out of the K loop (see the slab kernel)
        const int mw0 = (t / p.tiles_n) * C_BM + (wm * 2 + wmm) * 128, nb0 = (t % p.tiles_n) * C_BN + wn * 64, nb = nb0 + 16 * qd;
        if (nb0 >= p.N || (BSI_ABL(p.abl, 4))) return;  // wave-uniform: the lane-group exchanges below need every lane
        if constexpr (BF16_OUT) {
            if (nb >= p.N) return;
            if constexpr (EPI == CEPI_FILM_SILU_BF16) film_silu_inplace(p, acc, mw0, nb, HW);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r];
                if constexpr (EPI == CEPI_FILM_ROWS_SILU_BF16) {
                    // y*(scale+1)+shift (FeatureModulation, residual_block.py:21-24: addcmul(shift, scale+1, y)), then SiLU
                    int m = mw0 + 16 * j + rho;
                    m = m < p.M ? m : p.M - 1;
                    const float* fr = p.film + (size_t)((m / HW) % p.film_rows) * p.film_stride;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(fr + nb + e);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(fr + p.N + nb + e);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[e + r] = silu_f(__fmaf_rn(sc[r] + 1.0f, v[e + r], sh[r]));
                    }
                }
                u32x4 w0, w1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w0[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    w1[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                }
                store_rows_dpp(reinterpret_cast<__bf16*>(p.out), p.ldo, p.M, mw0 + 16 * j, nb, rho, w0, w1);
            }
        } else {
            store_f32_rows<EPI == CEPI_BIAS_RESID_F32_GN>(p, acc, mw0, nb0, rho, qd);
        }
    };

    // prologue: C_D stages in flight; stage 0 must have landed before the first load phase
    set_sources(tile);
    {
        bool all = true;
#pragma unroll
        for (int d = 0; d < C_D; ++d) all = issue_next() && all;
        if (all) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PHASE_BARRIER();
    if (wm == 1) PHASE_BARRIER();  // group B runs one phase behind

    int slot = 0;
    int after_e = 0;
    while (true) {
        const int next = tile + wpx;
        const bool has_next = next < hi;
        init_acc(tile);
        for (int v = 0; v < nk; ++v) {
            if (!(BSI_ABL(p.abl, 16))) {
                const char* b = lds + slot * C_SLOT;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(b + woff + i * 4 * C_RB);
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(b + xoff + j * 16 * C_RB);
            }
            // stage v+1 must have landed before the barrier: the C_D - 1 younger stages may stay in flight, and so may the
            // stores of an epilogue that sit behind stage v+1 in issue order (true for the C_D - 1 phases after it)
            if (issue_next()) {
                if (after_e > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1) + NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (after_e > 0) --after_e;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PHASE_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            if (!(BSI_ABL(p.abl, 2))) {
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int i0 = 0; i0 < 4; ++i0) {  // boustrophedon, as in the slab kernel below
                        const int i = (j & 1) ? 3 - i0 : i0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
            if (v == nk - 1 && wm == 1) epilogue(tile);  // group B: before the barrier that ends its last C phase
            PHASE_BARRIER();
            slot = (slot + 1) & (C_R - 1);
        }
        if (wm == 0) epilogue(tile);  // group A: after that barrier, merged with its next L phase
        // the store allowance is valid only if every store instruction of the epilogue was issued (no row / column tail)
        after_e = ((tile / p.tiles_n) * C_BM + C_BM <= p.M && (tile % p.tiles_n) * C_BN + C_BN <= p.N && !(BSI_ABL(p.abl, 4))) ? C_D - 1 : 0;
        if (!has_next) break;
        tile = next;
    }
    if (wm == 0) PHASE_BARRIER();
#undef PHASE_BARRIER
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 convolution with the PIXEL SLAB kept in LDS across the 9 taps.
// conv_ring_kernel re-fetches its 512 pixel rows for every tap: 40 KB of LDS-DMA per 256 MFMAs per CU, which is what bounds
// it (the DMA engine delivers 55-75 GB/s per CU for 64-B segments at NHWC row pitch).  Here the K loop is ordered
// (32-channel chunk, tap): the chunk's slab = the tile's 512 pixels + a halo of 48 pixels on either side, 608 rows x 64 B =
// 38 KB, is DMA'd ONCE and the 9 taps read their fragments from it at row offset dy*W + dx (any 16 consecutive rows of the
// swizzled image are bank-conflict free).  Border handling moves to the read side: a fragment = 16 consecutive pixels of
// one image row (W % 16 == 0), so "tap row outside the image" is wave-uniform per fragment (registers zeroed) and "tap
// column outside" concerns lane 0 or 15 of fragments at a row end (those lanes read a zero row of LDS instead).
// DMA per phase: 8 KB of weights + 1/9 of the next slab = 12.3 KB instead of 40 KB.
// LDS: 2 slabs (double buffer) 76 KB + ring of 8 weight stages x 8 KB (6 in flight) + zero row.
// Issue order per phase and wave: [epilogue stores] [slab piece, taps 0..4] [weight stage P+6], in EVERY phase (past the end of a
// stream the instruction is a harmless dummy), so the vmcnt allowance -- the number of instructions issued after the one that
// must have landed -- is an immediate per tap (+ the stores of an epilogue inside the window).
constexpr int S_HALO = 48, S_ROWS = C_BM + 2 * S_HALO, S_INSTR = S_ROWS / 16;  // 608 rows, 38 DMA instructions
constexpr int S_SLAB = S_ROWS * C_RB;                                             // 38912 B
constexpr int S_WSLOTS = 8, S_WD = 6, S_WSTAGE = C_BN * C_RB;                     // 8 KB weight stages, 6 in flight
constexpr int S_WRING = 2 * S_SLAB, S_ZERO = S_WRING + S_WSLOTS * S_WSTAGE, S_LDS = S_ZERO + 256;  // S_ZERO is 256-B aligned
static_assert(S_ZERO % 256 == 0, "the zero region mirrors the bank position of the address it replaces");

// WD = image width (16 or 32: a template parameter, so that which fragments of a wave end at a row end is known to the compiler):
// a fragment is 16 pixels of one image row; at WD = 32 the even fragments of a wave start a row and the odd ones end it, at
// WD = 16 every fragment does both.
template <int EPI, int WD>
__global__ __launch_bounds__(512) void conv_slab_kernel(const ConvParams p) {
    constexpr int TM = 8, NW = 8;
    constexpr unsigned XL = WD == 32 ? 0x55u : 0xffu, XR = WD == 32 ? 0xaau : 0xffu;  // fragments with a pixel in column 0 / WD-1
    constexpr bool BF16_OUT = (EPI < CEPI_BIAS_RESID_F32 || EPI == CEPI_FILM_ROWS_SILU_BF16);
    constexpr int NSTORE = BF16_OUT ? 2 * TM : 4 * TM + (EPI == CEPI_BIAS_RESID_F32_GN ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wmm = (wave >> 1) & 1, wn = wave & 1;
    const int wrow0 = (wm * 2 + wmm) * 128;  // first tile row of this wave

    const int nwg = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int wpx = (gridDim.x + 7 - xcd) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int lo = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + wl;
    if (tile >= hi) return;
    const int nc = p.Cin / 32;
    const int HW = p.H * WD;

    if (tid < 64) reinterpret_cast<float*>(lds + S_ZERO)[tid] = 0.f;  // 256 B of zeros, visible after the prologue's barrier

    // ---- slab stream (runs one slab ahead of the compute): 5 DMA instructions per wave and slab, 16 rows x 64 B each.
    // The buffer base is the slab's first row, so a lane's offset never changes; rows past the end of the tensor fall outside
    // num_records (zeros), rows before its start (first tile only) are sent out of range explicitly.
    const int srow = lane >> 2, spos = lane & 3;
    // slab rows are swizzled with key (row >> 1) & 3: measured conflict free for fragments starting at ANY row
    // (tools/experiments/lds_shift_conflicts.hip: 7.2 cycles per ds_read_b128 for shifts 0..15; the aligned kernels' key
    // (-(row >> 2)) & 3 costs 9.3 cycles when the first row is not a multiple of 8, i.e. for the dx = +-1 taps)
    const int achunk = (spos ^ ((srow >> 1) & 3)) * 16;
    const int rowb = p.Cin * 2;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    asm volatile("" : "+s"(Ab), "+s"(Wb));  // keep the kernel arguments in SGPRs (no reloads inside the loop)
    unsigned soff[5];
    unsigned before = 0;  // bit q: this lane's row of piece q lies in the leading halo (before the tensor when the tile is the first)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int ins = q * NW + wave;
        ins = ins < S_INSTR ? ins : S_INSTR - 1;  // waves 6, 7: the 5th piece repeats the last rows (same data)
        const int r = ins * 16 + srow;
        soff[q] = (unsigned)r * (unsigned)rowb + achunk;
        before |= (r < S_HALO ? 1u : 0u) << q;
    }
    int st_tile = tile, st_c = 0, sg = 0;  // slab being filled: tile, chunk, running slab index (buffer = sg & 1)
    int st_m0 = (tile / p.tiles_n) * C_BM - S_HALO;  // its first pixel row (negative for the tensor's first tile)
    __amdgpu_buffer_rsrc_t srs;                      // buffer resource of that slab: base = its first row, chunk st_c
    auto slab_resource = [&]() {
        const long rec = ((long)p.M - st_m0) * rowb - st_c * 64;
        srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab + (long)st_m0 * rowb + st_c * 64), 0,
                                                (unsigned)(rec < (long)BUF_RECORDS ? rec : (long)BUF_RECORDS), 0x00020000);
    };
    slab_resource();
    // Both streams issue in EVERY phase, also past their end (st_tile / wt_tile >= hi): a slab piece then carries an out-of-range
    // offset (zeros into the slab buffer nobody reads any more), a weight stage re-reads the first one into a ring slot that is
    // free.  The number of instructions issued after the one a phase waits for is therefore a constant of the tap, and the
    // vmcnt allowance an immediate -- the run-time bookkeeping of irregular phases (five counters, a switch per phase) cost the
    // kernel 18 % (UNet sampling 648 -> 694 images/s with it compiled out).
    auto issue_slab_piece = [&](int q) {
        int ins = q * NW + wave;
        ins = ins < S_INSTR ? ins : S_INSTR - 1;
        unsigned vo = soff[q];
        if (st_m0 < 0) vo = ((before >> q) & 1) ? OOB_OFFSET : vo;
        if (st_tile >= hi) vo = OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, LDS_PTR(lds + (sg & 1) * S_SLAB + ins * 1024), 16, vo, 0, 0, 0);
    };
    auto advance_slab = [&]() {  // after piece 4
        ++sg;
        if (++st_c == nc) {
            st_c = 0;
            st_tile += wpx;
            st_m0 = (st_tile / p.tiles_n) * C_BM - S_HALO;
        }
        if (st_tile < hi) slab_resource();
    };
    // ---- weight stream: stage of global phase wP (tile, chunk, tap), 16 rows x 64 B per wave; N % 128 == 0 (launcher)
    const unsigned woffs = (unsigned)(wave * 16 + srow) * (unsigned)(p.K * 2) + (spos ^ ((-wave) & 3)) * 16;
    int wt_tile = tile, wt_c = 0, wP = 0;
    const char* wptr = Wb + (size_t)(tile % p.tiles_n) * C_BN * p.K * 2;  // + K offset of (chunk, tap), advanced incrementally
    auto issue_w = [&]() {
        const char* src = wt_tile < hi ? wptr : Wb;  // past the end: any valid weight row (see above)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + woffs), LDS_PTR(lds + S_WRING + (wP & (S_WSLOTS - 1)) * S_WSTAGE + wave * 1024), 16, 0, 0);
        ++wP;
        wptr += rowb;  // next tap: + Cin channels
    };
    auto advance_w_chunk = [&]() {  // after tap 8: the stream always runs S_WD phases ahead, so the tap is known at compile time
        if (wt_tile >= hi) return;
        wptr += 64 - 9 * rowb;  // next chunk: + 32 channels of tap 0
        if (++wt_c == nc) {
            wt_c = 0;
            wt_tile += wpx;
            wptr = Wb + (size_t)(wt_tile % p.tiles_n) * C_BN * p.K * 2;
        }
    };

    const int rho = lane & 15, qd = lane >> 4;
    const int wcc = ((qd ^ ((-(rho >> 2)) & 3)) << 4);
    const int woff = S_WRING + (wn * 64 + 16 * (rho >> 2) + (rho & 3)) * C_RB + wcc;
    const int xrow0 = S_HALO + wrow0 + rho;  // slab row of this lane's pixel in fragment 0, before the tap shift
    const bool edge_lane[2] = {rho == 0, rho == 15};  // the lane of a fragment that looks past the row end under dx = -1 / +1

    f32x4 acc[4][TM];
    bf16x8 wf[4], xf[TM];
#define PHASE_BARRIER()                          \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

    auto init_acc = [&](int t) {
        f32x4 bv[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int nb = (t % p.tiles_n) * C_BN + wn * 64 + 16 * qd;
        if (p.bias && nb < p.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = bv[i];
    };
    auto epilogue = [&](int t) {
        // fence: the epilogue's row addresses (16 per lane) depend on the tile only; group B's epilogue sits inside the tap loop, and
        // without the fence they are computed in front of it and live -- spilled -- through it
        asm volatile("" : "+s"(t));
        const int mw0 = (t / p.tiles_n) * C_BM + wrow0, nb0 = (t % p.tiles_n) * C_BN + wn * 64, nb = nb0 + 16 * qd;
        if (nb0 >= p.N || (BSI_ABL(p.abl, 4))) return;
        if constexpr (BF16_OUT) {
            if (nb >= p.N) return;
            // FiLM + SiLU per (image, channel); the wave's 128 rows lie in ONE image (launcher: H*W % 128 == 0)
            if constexpr (EPI == CEPI_FILM_SILU_BF16) film_silu_inplace(p, acc, mw0, nb, HW);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                f32x4 v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
                u32x4 w0, w1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    w0[e] = pack_bf16x2(v[0][2 * e], v[0][2 * e + 1]);
                    w0[2 + e] = pack_bf16x2(v[1][2 * e], v[1][2 * e + 1]);
                    w1[e] = pack_bf16x2(v[2][2 * e], v[2][2 * e + 1]);
                    w1[2 + e] = pack_bf16x2(v[3][2 * e], v[3][2 * e + 1]);
                }
                store_rows_dpp(reinterpret_cast<__bf16*>(p.out), p.ldo, p.M, mw0 + 16 * j, nb, rho, w0, w1);
            }
        } else {
            store_f32_rows<EPI == CEPI_BIAS_RESID_F32_GN>(p, acc, mw0, nb0, rho, qd);
        }
    };

    // the vmcnt allowance may count an epilogue's stores only when every one of them is issued (no row / column tail)
    auto full_tile = [&](int t) {
        return (t / p.tiles_n) * C_BM + C_BM <= p.M && (t % p.tiles_n) * C_BN + C_BN <= p.N && !(BSI_ABL(p.abl, 4));
    };
    // prologue: slab 0 and weight stages 0 .. S_WD-1; slab 0 and stage 0 must have landed before the first load phase
#pragma unroll
    for (int q = 0; q < 5; ++q) issue_slab_piece(q);
    advance_slab();
#pragma unroll
    for (int d = 0; d < S_WD; ++d) {
        issue_w();
        if (d % 9 == 8) advance_w_chunk();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_WD - 1) : "memory");  // slab 0 and weight stage 0 landed, stages 1 .. S_WD-1 in flight
    PHASE_BARRIER();
    if (wm == 1) PHASE_BARRIER();  // group B runs one phase behind

    // Weight stages 2 .. S_WD-1 count as issued in phases -4 .. -1: from phase 0 on, the instructions issued after the one a phase
    // waits for are those of a steady state.  after_e: phases whose allowance also covers the NSTORE stores of an epilogue issued
    // inside that window (only when every store of it was issued: no row / column tail; otherwise the wait simply includes them).
    int P = 0, cg = 0;    // global phase and slab counters of the compute side
    int after_e = 0;
    while (true) {
        const int next = tile + wpx;
        const bool has_next = next < hi;
        init_acc(tile);
        // border masks of this wave's 8 fragments (16 pixels of one image row each): lane j < 8 evaluates fragment j
        unsigned ytop, ybot;
        {
            const int R = (tile / p.tiles_n) * C_BM + wrow0 + 16 * (lane & 7);
            const int rem = R % HW, y = rem / WD;
            ytop = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(y == 0) & 0xff));
            ybot = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(y == p.H - 1) & 0xff));
        }
        for (int c = 0; c < nc; ++c) {
            const int slab_off = (cg & 1) * S_SLAB;
            const char* slab = lds + slab_off;
            auto tap = [&](auto T_) {
                constexpr int t = decltype(T_)::value;
                constexpr int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
                constexpr unsigned ZL = dx < 0 ? XL : dx > 0 ? XR : 0u;  // fragments whose lane 0 (dx < 0) / lane 15 (dx > 0) falls outside
                // ---- L phase: fragments of (chunk c, tap t)
                if (!(BSI_ABL(p.abl, 16)) || (c == 0 && t == 0)) {
                    const char* wb = lds + (P & (S_WSLOTS - 1)) * S_WSTAGE;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + woff + i * 4 * C_RB);
                    int row = xrow0 + dy * WD + dx;
                    asm volatile("" : "+v"(row));  // recompute per phase: hoisting the 9 taps' addresses out of the chunk loop costs 20+ VGPRs
                    const int xaddr = row * C_RB + ((qd ^ ((row >> 1) & 3)) << 4);
                    // the two row-border words are re-materialised per phase: left to itself the compiler derives a 64-bit lane mask
                    // for every (tap, fragment) once per tile -- scalar registers that are all spilled and read back in every phase
                    asm volatile("" : "+s"(ytop), "+s"(ybot));
                    const unsigned zall = dy < 0 ? ytop : dy > 0 ? ybot : 0u;
                    if (zall == 0) {  // no fragment of this wave lies in an image row whose tap row is outside (three quarters of the waves)
                        if constexpr (ZL == 0) {
#pragma unroll
                            for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(slab + xaddr + j * 16 * C_RB);
                        } else {
                            // column border: WHICH fragments have their edge lane outside is a constant of the tap (ZL), so only that
                            // lane's address is selected -- the zero region at the bank position of the address it replaces (the
                            // other 15 rows of its group leave exactly that slot free; j * 1 KB does not move the low 8 bits)
                            const int a0 = slab_off + xaddr;
                            const int za = edge_lane[dx < 0 ? 0 : 1] ? S_ZERO + (a0 & 255) : -1;
#pragma unroll
                            for (int j = 0; j < TM; ++j) {
                                if ((ZL >> j) & 1) {  // a constant once the loop is unrolled
                                    const int a = za >= 0 ? za : a0 + j * 16 * C_RB;
                                    xf[j] = *reinterpret_cast<const bf16x8*>(lds + a);
                                } else {
                                    xf[j] = *reinterpret_cast<const bf16x8*>(slab + xaddr + j * 16 * C_RB);
                                }
                            }
                        }
                    } else {
                        const bool el = dx == 0 ? false : edge_lane[dx < 0 ? 0 : 1];
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            const bool z = ((zall >> j) & 1) || (((ZL >> j) & 1) && el);
                            const int a0 = slab_off + xaddr + j * 16 * C_RB;
                            const int a = z ? S_ZERO + (a0 & 255) : a0;
                            xf[j] = *reinterpret_cast<const bf16x8*>(lds + a);
                        }
                    }
                }
                // ---- issue: next slab piece (taps 0..4), weight stage P + S_WD
                if (!(BSI_ABL(p.abl, 1))) {
                    if constexpr (t < 5) {
                        issue_slab_piece(t);
                        if constexpr (t == 4) advance_slab();
                    }
                    issue_w();
                    if constexpr ((t + S_WD) % 9 == 8) advance_w_chunk();
                }
                // must have landed: weight stage P+1 (issued in phase P-5) and, at the last tap, the whole next slab (its last
                // piece was the first instruction of phase P-4): everything issued after those may stay in flight -- a constant
                // of the tap, plus the stores of an epilogue issued in that window
                {
                    constexpr int pieces[9] = {1, 2, 3, 4, 5, 4, 3, 2, 0};
                    constexpr int allow = t == 8 ? 4 : 5 + pieces[t];
                    if (after_e > 0) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow + NSTORE) : "memory");
                        --after_e;
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow) : "memory");
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PHASE_BARRIER();
                // ---- C phase
                __builtin_amdgcn_s_setprio(1);
                if (!(BSI_ABL(p.abl, 2))) {  // (bsi_conv_set_ablation: kernel experiments)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
#pragma unroll
                        for (int i0 = 0; i0 < 4; ++i0) {  // boustrophedon: one operand changes per MFMA (power; gemm_bf16.hip)
                            const int i = (j & 1) ? 3 - i0 : i0;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_s_setprio(0);
                if (t == 8 && c == nc - 1 && wm == 1) { epilogue(tile); after_e = full_tile(tile) ? 5 : 0; }  // group B: before its last barrier
                PHASE_BARRIER();
                ++P;
            };
            tap(std::integral_constant<int, 0>{});
            tap(std::integral_constant<int, 1>{});
            tap(std::integral_constant<int, 2>{});
            tap(std::integral_constant<int, 3>{});
            tap(std::integral_constant<int, 4>{});
            tap(std::integral_constant<int, 5>{});
            tap(std::integral_constant<int, 6>{});
            tap(std::integral_constant<int, 7>{});
            tap(std::integral_constant<int, 8>{});
            ++cg;
        }
        if (wm == 0) { epilogue(tile); after_e = full_tile(tile) ? 5 : 0; }  // group A: merged with its next L phase
        if (!has_next) break;
        tile = next;
    }
    // the dummy DMAs of the last phases must have landed before the workgroup's LDS can be handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 0) PHASE_BARRIER();
#undef PHASE_BARRIER
}


// ---------------------------------------------------------------------------------------------------------
// The slab kernel WITH a second source: conv2 of the up blocks = 3 x 3 over h (Cin channels) + the folded 1 x 1 skip convolution
// over the raw concatenation cat(x, skip) (Cin2 = 2 Cin channels), which the ring kernel ran at 138 us against 101 us for the same
// 3 x 3 on the slab kernel.  A skip K step has ONE tap: its 32 KB of pixels are consumed a phase after they arrive, so it cannot
// join the two-slab rotation of the 3 x 3 chunks.  It gets ONE buffer of its own (Q, the tile's 512 pixels, no halo) and the two
// skip steps of a chunk are placed where Q can be refilled in between -- eleven phases per 32-channel chunk c of h:
//     position   0    1    2    3    4          5    6    7    8    9    10
//     K step     t0   t1   t2   t3   skip 2c    t4   t5   t6   t7   t8   skip 2c+1
//     issues     m s s  m s  m s  m   -          m k  k    k    k    -    -          (+ one weight stage each)
// m = a piece of the NEXT chunk's main slab (the other slab buffer, as in conv_slab_kernel), s = a piece of THIS chunk's first skip
// step (Q is free: the previous chunk's second skip step was read at position 10, by the lagging wave group half a phase before
// position 0's issue), k = a piece of this chunk's second skip step (Q was read at position 4).  The last s goes out at position 2
// and is read at 4, the last k at 8 and is read at 10: one and a half phases of lead, what the ring kernel gives its stages.  The
// weight stream runs five 8-KB stages ahead on a ring of six (a slot is rewritten the phase after both wave groups read it).
// LDS: 2 x 38 KB + 32 KB + 48 KB + the zero row = 156.25 KB.  The issue sequence is the same in every chunk of every tile (past
// the end of a stream the instructions are harmless dummies), so the vmcnt allowance of a phase -- the instructions issued after
// the youngest one it needs -- is a constant of its position: the weight stage of the next phase went out four phases ago
// (8 9 11 12 9 9 8 8 9 7 6 instructions since); position 3 also needs Q whole (3), position 9 Q again (2), position 10 the next
// main slab (10): the minimum of the two.
constexpr int S2_SKIP = C_BM * C_RB;                                   // 32 KB: the tile's 512 pixels, no halo
constexpr int S2_Q = 2 * S_SLAB, S2_WRING = S2_Q + S2_SKIP;
constexpr int S2_WSLOTS = 6, S2_WD = 5;
constexpr int S2_ZERO = S2_WRING + S2_WSLOTS * S_WSTAGE, S2_LDS = S2_ZERO + 256;
static_assert(S2_ZERO % 256 == 0 && S2_LDS <= 160 * 1024, "LDS image of the two-source slab kernel");

template <int EPI, int WD>
__global__ __launch_bounds__(512) void conv_slab2_kernel(const ConvParams p) {
    constexpr int TM = 8, NW = 8;
    constexpr unsigned XL = WD == 32 ? 0x55u : 0xffu, XR = WD == 32 ? 0xaau : 0xffu;
    constexpr bool BF16_OUT = (EPI < CEPI_BIAS_RESID_F32 || EPI == CEPI_FILM_ROWS_SILU_BF16);
    constexpr int NSTORE = BF16_OUT ? 2 * TM : 4 * TM + (EPI == CEPI_BIAS_RESID_F32_GN ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wmm = (wave >> 1) & 1, wn = wave & 1;
    const int wrow0 = (wm * 2 + wmm) * 128;

    const int nwg = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int wpx = (gridDim.x + 7 - xcd) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int lo = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + wl;
    if (tile >= hi) return;
    const int nc = p.Cin / 32;  // chunks of the 3 x 3 source; the skip source has 2 nc (launcher)
    const int HW = p.H * WD;

    if (tid < 64) reinterpret_cast<float*>(lds + S2_ZERO)[tid] = 0.f;

    const int srow = lane >> 2, spos = lane & 3;
    const int achunk = (spos ^ ((srow >> 1) & 3)) * 16;
    const int rowb = p.Cin * 2, rowb2 = p.Cin2 * 2;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* A2b = reinterpret_cast<const char*>(p.A2);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    asm volatile("" : "+s"(Ab), "+s"(A2b), "+s"(Wb));
    // ---- main slab stream, one slab ahead of the compute (conv_slab_kernel's)
    unsigned soff[5];
    unsigned before = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int ins = q * NW + wave;
        ins = ins < S_INSTR ? ins : S_INSTR - 1;
        const int r = ins * 16 + srow;
        soff[q] = (unsigned)r * (unsigned)rowb + achunk;
        before |= (r < S_HALO ? 1u : 0u) << q;
    }
    int st_tile = tile, st_c = 0, sg = 0;
    int st_m0 = (tile / p.tiles_n) * C_BM - S_HALO;
    __amdgpu_buffer_rsrc_t srs;
    auto slab_resource = [&]() {
        const long rec = ((long)p.M - st_m0) * rowb - st_c * 64;
        srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab + (long)st_m0 * rowb + st_c * 64), 0,
                                                (unsigned)(rec < (long)BUF_RECORDS ? rec : (long)BUF_RECORDS), 0x00020000);
    };
    slab_resource();
    auto issue_slab_piece = [&](int q) {
        int ins = q * NW + wave;
        ins = ins < S_INSTR ? ins : S_INSTR - 1;
        unsigned vo = soff[q];
        if (st_m0 < 0) vo = ((before >> q) & 1) ? OOB_OFFSET : vo;
        if (st_tile >= hi) vo = OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, LDS_PTR(lds + (sg & 1) * S_SLAB + ins * 1024), 16, vo, 0, 0, 0);
    };
    auto advance_slab = [&]() {
        ++sg;
        if (++st_c == nc) {
            st_c = 0;
            st_tile += wpx;
            st_m0 = (st_tile / p.tiles_n) * C_BM - S_HALO;
        }
        if (st_tile < hi) slab_resource();
    };
    // ---- skip steps of the chunk being computed: piece q of skip chunk sc -> rows 128 q + 16 wave .. + 15 of Q
    const unsigned koff = (unsigned)(wave * 16 + srow) * (unsigned)rowb2 + achunk;  // piece q lies 128 q rows further: the scalar offset
    __amdgpu_buffer_rsrc_t krs;
    auto skip_resource = [&](int sc) {  // the tile's pixels from its first row, channels 32 sc ..: rows past the tensor read zeros
        const int m0 = (tile / p.tiles_n) * C_BM;
        const long rec = ((long)p.M - m0) * rowb2 - sc * 64;
        krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A2b + (long)m0 * rowb2 + sc * 64), 0,
                                                (unsigned)(rec < (long)BUF_RECORDS ? rec : (long)BUF_RECORDS), 0x00020000);
    };
    auto issue_skip_piece = [&](int q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, LDS_PTR(lds + S2_Q + (q * NW + wave) * 1024), 16, koff, q * 128 * rowb2, 0, 0);
    };
    // ---- weight stream: the stage of the phase five ahead; (tile, chunk) of that phase, its position is a constant of this one's
    const unsigned woffs = (unsigned)(wave * 16 + srow) * (unsigned)(p.K * 2) + (spos ^ ((-wave) & 3)) * 16;
    int wt_tile = tile, wt_c = 0;
    int cslot = 0;  // ring slot of the stage this phase reads; the stage issued in it goes to slot cslot - 1 (mod 6)
    auto issue_w = [&](int pos) {  // pos: position of the issued stage in its chunk's eleven phases
        const int tap = pos < 4 ? pos : pos - 1;  // positions 0-3 -> taps 0-3, 5-9 -> taps 4-8 (4 and 10 are the skip steps)
        const char* base = wt_tile < hi ? Wb + (size_t)(wt_tile % p.tiles_n) * C_BN * p.K * 2 : Wb;
        const int off = wt_tile >= hi ? 0 : (pos == 4 || pos == 10) ? 9 * rowb + (2 * wt_c + (pos == 10 ? 1 : 0)) * 64 : tap * rowb + wt_c * 64;
        const int islot = cslot == 0 ? S2_WSLOTS - 1 : cslot - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off + woffs), LDS_PTR(lds + S2_WRING + islot * S_WSTAGE + wave * 1024), 16, 0, 0);
        if (pos == 10 && wt_tile < hi) {
            if (++wt_c == nc) {
                wt_c = 0;
                wt_tile += wpx;
            }
        }
    };

    const int rho = lane & 15, qd = lane >> 4;
    const int wcc = ((qd ^ ((-(rho >> 2)) & 3)) << 4);
    const int woff = S2_WRING + (wn * 64 + 16 * (rho >> 2) + (rho & 3)) * C_RB + wcc;
    const int xrow0 = S_HALO + wrow0 + rho;
    const bool edge_lane[2] = {rho == 0, rho == 15};
    const int krow = wrow0 + rho;  // Q holds the tile's rows from 0
    const int kaddr = S2_Q + krow * C_RB + ((qd ^ ((krow >> 1) & 3)) << 4);

    f32x4 acc[4][TM];
    bf16x8 wf[4], xf[TM];
#define PHASE_BARRIER()                          \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

    auto init_acc = [&](int t) {
        f32x4 bv[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int nb = (t % p.tiles_n) * C_BN + wn * 64 + 16 * qd;
        if (p.bias && nb < p.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = bv[i];
    };
    auto epilogue = [&](int t) {
        asm volatile("" : "+s"(t));
        const int mw0 = (t / p.tiles_n) * C_BM + wrow0, nb0 = (t % p.tiles_n) * C_BN + wn * 64, nb = nb0 + 16 * qd;
        if (nb0 >= p.N || (BSI_ABL(p.abl, 4))) return;
        if constexpr (BF16_OUT) {
            if (nb >= p.N) return;
            if constexpr (EPI == CEPI_FILM_SILU_BF16) film_silu_inplace(p, acc, mw0, nb, HW);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                f32x4 v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
                u32x4 w0, w1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    w0[e] = pack_bf16x2(v[0][2 * e], v[0][2 * e + 1]);
                    w0[2 + e] = pack_bf16x2(v[1][2 * e], v[1][2 * e + 1]);
                    w1[e] = pack_bf16x2(v[2][2 * e], v[2][2 * e + 1]);
                    w1[2 + e] = pack_bf16x2(v[3][2 * e], v[3][2 * e + 1]);
                }
                store_rows_dpp(reinterpret_cast<__bf16*>(p.out), p.ldo, p.M, mw0 + 16 * j, nb, rho, w0, w1);
            }
        } else {
            store_f32_rows<EPI == CEPI_BIAS_RESID_F32_GN>(p, acc, mw0, nb0, rho, qd);
        }
    };
    auto full_tile = [&](int t) {
        return (t / p.tiles_n) * C_BM + C_BM <= p.M && (t % p.tiles_n) * C_BN + C_BN <= p.N && !(BSI_ABL(p.abl, 4));
    };

    // prologue: the first main slab and weight stages 0 .. 4 (positions 0 .. 4 of chunk 0); the slab and stage 0 must have landed
#pragma unroll
    for (int q = 0; q < 5; ++q) issue_slab_piece(q);
    advance_slab();
#pragma unroll
    for (int d = 0; d < S2_WD; ++d) {
        cslot = d + 1;  // issue_w writes slot cslot - 1 = d
        issue_w(d);
    }
    cslot = 0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S2_WD - 1) : "memory");
    PHASE_BARRIER();
    if (wm == 1) PHASE_BARRIER();

    bool cold = true;  // the kernel's first phase: stages 1 .. 4 went out back to back in the prologue (7 instructions since stage 1, not 8)
    int cg = 0;
    int after_e = 0;
    while (true) {
        const int next = tile + wpx;
        const bool has_next = next < hi;
        init_acc(tile);
        unsigned ytop, ybot;
        {
            const int R = (tile / p.tiles_n) * C_BM + wrow0 + 16 * (lane & 7);
            const int rem = R % HW, y = rem / WD;
            ytop = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(y == 0) & 0xff));
            ybot = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(y == p.H - 1) & 0xff));
        }
        for (int c = 0; c < nc; ++c) {
            const int slab_off = (cg & 1) * S_SLAB;
            auto phase = [&](auto I_) {
                constexpr int I = decltype(I_)::value;  // position in the chunk: 4 and 10 are the skip steps
                constexpr bool SKIP = I == 4 || I == 10;
                // ---- L phase: fragments
                {
                    const char* wb = lds + cslot * S_WSTAGE;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + woff + i * 4 * C_RB);
                }
                if constexpr (!SKIP) {
                    constexpr int t = I < 4 ? I : I - 1;
                    constexpr int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
                    constexpr unsigned ZL = dx < 0 ? XL : dx > 0 ? XR : 0u;
                    int row = xrow0 + dy * WD + dx;
                    asm volatile("" : "+v"(row));
                    const int xaddr = slab_off + row * C_RB + ((qd ^ ((row >> 1) & 3)) << 4);
                    asm volatile("" : "+s"(ytop), "+s"(ybot));
                    const unsigned zall = dy < 0 ? ytop : dy > 0 ? ybot : 0u;
                    if (zall == 0) {
                        if constexpr (ZL == 0) {
#pragma unroll
                            for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(lds + xaddr + j * 16 * C_RB);
                        } else {
                            const int za = edge_lane[dx < 0 ? 0 : 1] ? S2_ZERO + (xaddr & 255) : -1;
#pragma unroll
                            for (int j = 0; j < TM; ++j) {
                                if ((ZL >> j) & 1) {
                                    const int a = za >= 0 ? za : xaddr + j * 16 * C_RB;
                                    xf[j] = *reinterpret_cast<const bf16x8*>(lds + a);
                                } else {
                                    xf[j] = *reinterpret_cast<const bf16x8*>(lds + xaddr + j * 16 * C_RB);
                                }
                            }
                        }
                    } else {
                        const bool el = dx == 0 ? false : edge_lane[dx < 0 ? 0 : 1];
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            const bool z = ((zall >> j) & 1) || (((ZL >> j) & 1) && el);
                            const int a0 = xaddr + j * 16 * C_RB;
                            const int a = z ? S2_ZERO + (a0 & 255) : a0;
                            xf[j] = *reinterpret_cast<const bf16x8*>(lds + a);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(lds + kaddr + j * 16 * C_RB);
                }
                // ---- issue: [main slab piece] [skip pieces] [weight stage of the phase five ahead]
                if constexpr (I < 4) issue_slab_piece(I);
                if constexpr (I == 5) { issue_slab_piece(4); advance_slab(); }
                if constexpr (I == 0) { skip_resource(2 * c); issue_skip_piece(0); issue_skip_piece(1); }
                if constexpr (I == 1) issue_skip_piece(2);
                if constexpr (I == 2) issue_skip_piece(3);
                if constexpr (I == 5) { skip_resource(2 * c + 1); issue_skip_piece(0); }
                if constexpr (I == 6) issue_skip_piece(1);
                if constexpr (I == 7) issue_skip_piece(2);
                if constexpr (I == 8) issue_skip_piece(3);
                issue_w((I + S2_WD) % 11);
                {
                    constexpr int allow_[11] = {8, 9, 11, 3, 9, 9, 8, 8, 9, 2, 6};
                    constexpr int allow = allow_[I];
                    if (I == 0 && cold) {
                        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                        cold = false;
                    } else if (I < 3 && after_e > 0) {  // positions 0 .. 2 after an epilogue: its stores are younger than what the phase needs
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow + NSTORE) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow) : "memory");
                    }
                    if (I == 2) after_e = 0;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PHASE_BARRIER();
                // ---- C phase
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int i0 = 0; i0 < 4; ++i0) {
                        const int i = (j & 1) ? 3 - i0 : i0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                if (I == 10 && c == nc - 1 && wm == 1) { epilogue(tile); after_e = full_tile(tile) ? 1 : 0; }  // group B: before its last barrier
                PHASE_BARRIER();
                cslot = cslot == S2_WSLOTS - 1 ? 0 : cslot + 1;
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            phase(std::integral_constant<int, 5>{});
            phase(std::integral_constant<int, 6>{});
            phase(std::integral_constant<int, 7>{});
            phase(std::integral_constant<int, 8>{});
            phase(std::integral_constant<int, 9>{});
            phase(std::integral_constant<int, 10>{});
            ++cg;
        }
        if (wm == 0) { epilogue(tile); after_e = full_tile(tile) ? 1 : 0; }  // group A: merged with its next load phase (position 0)
        if (!has_next) break;
        tile = next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 0) PHASE_BARRIER();
#undef PHASE_BARRIER
}

int g_conv_abl = 0;
int g_conv_grid_limit = 0;  // > 0: at most this many workgroups (tests: several tiles per workgroup on small inputs)

int conv_cus() {
    const int cus = compute_cus();
    if (g_conv_grid_limit > 0 && g_conv_grid_limit < cus) return g_conv_grid_limit;
    return cus;
}

template <int EPI>
int launch_conv_slab(const ConvParams& p, int grid, hipStream_t s) {
    if (p.Wd == 32) {
        auto kern = conv_slab_kernel<EPI, 32>;
        set_max_lds(reinterpret_cast<const void*>(kern), (int)S_LDS);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S_LDS, s, p);
    } else {
        auto kern = conv_slab_kernel<EPI, 16>;
        set_max_lds(reinterpret_cast<const void*>(kern), (int)S_LDS);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S_LDS, s, p);
    }
    BSI_CHECK_LAUNCH("bsi_conv_nhwc_bf16");
    return BSI_OK;
}

template <int EPI>
int launch_conv(ConvParams p, hipStream_t s) {
    p.tiles_m = (p.M + C_BM - 1) / C_BM;
    p.abl = g_conv_abl;
    static const int stag = [] { const char* e = getenv("BSI_CONV_STAGGER"); return e ? atoi(e) : 0; }();
    p.stagger = (stag >> 10) > 0 ? stag : 0;
    p.tiles_n = (p.N + C_BN - 1) / C_BN;
    const int nwg = p.tiles_m * p.tiles_n;
    const int grid = nwg < conv_cus() ? nwg : conv_cus();
    if constexpr (EPI != CEPI_FILM_ROWS_SILU_BF16) {
        // 3x3 without folded skip steps on images whose rows are whole fragments CAN run the slab kernel.  Which one does is
        // measured per epilogue (one MI355X, 256 images of 32 x 32; tools/conv_bench.py with ABL=0,512,256 after the slab kernel
        // lost its run-time issue bookkeeping and its spilled border masks):
        //   bias -> bf16:                 slab 89 us (128 -> 128), 133 us (256 -> 128), 208 us (128 -> 384), 175 us (384 -> 128);
        //                                 ring 102 / 175 / 235 / 252 us
        //   FiLM + SiLU -> bf16 (conv1):  slab (the ring kernel's FiLM instance spilled inside its K loop: 230 us average)
        //   bias + residual -> fp32 (+ GroupNorm partials, conv2): ring 141-143 us, slab 148 us (128 -> 128)
        // Re-measured once no instance spilled any more (the tile fence of the epilogues; UNet sampling, 512 images, k = 16): fp32
        // epilogues on the slab kernel 815 images/s against 772 on the ring kernel, FiLM on the ring kernel 670 -- everything that
        // can runs the slab kernel now.
        // Ablation flags (bsi_conv_set_ablation): 256 = never the slab kernel, 512 = the slab kernel wherever the shape allows,
        // 2048 / 4096 = ring kernel for the FiLM / fp32 epilogues.
        // 3 x 3 + folded 1 x 1 skip (conv2 of the up blocks): the two-source slab kernel when the skip source has exactly two chunks per
        // chunk of the 3 x 3 source (the UNet: 128 + 256 channels); ablation bits 256 / 8192 keep the ring kernel (8192: for these shapes only,
        // the A/B partner of tools/unet_bench.py)
        if (p.taps == 9 && p.Cin2 == 2 * p.Cin && p.Cin2 > 0 && (p.Wd == 16 || p.Wd == 32) && p.N % C_BN == 0 && !(g_conv_abl & (256 | 8192))) {
            if (p.Wd == 32) {
                auto kern = conv_slab2_kernel<EPI, 32>;
                set_max_lds(reinterpret_cast<const void*>(kern), (int)S2_LDS);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S2_LDS, s, p);
            } else {
                auto kern = conv_slab2_kernel<EPI, 16>;
                set_max_lds(reinterpret_cast<const void*>(kern), (int)S2_LDS);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S2_LDS, s, p);
            }
            BSI_CHECK_LAUNCH("bsi_conv_nhwc_bf16");
            return BSI_OK;
        }
        const bool can = p.taps == 9 && p.Cin2 == 0 && (p.Wd == 16 || p.Wd == 32) && p.N % C_BN == 0;
        constexpr bool F32 = (EPI == CEPI_BIAS_RESID_F32 || EPI == CEPI_BIAS_RESID_F32_GN);
        const bool want = (g_conv_abl & 512) ? true
                          : (g_conv_abl & 256) ? false
                          : F32 ? !(g_conv_abl & 4096)
                          : EPI == CEPI_FILM_SILU_BF16 ? !(g_conv_abl & 2048)
                                                       : true;
        if (can && want) return launch_conv_slab<EPI>(p, grid, s);
    }
    const size_t lds = (size_t)C_R * C_SLOT;
    auto kern = conv_ring_kernel<EPI>;
    set_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
    BSI_CHECK_LAUNCH("bsi_conv_nhwc_bf16");
    return BSI_OK;
}

}  // namespace

extern "C" int bsi_conv_set_ablation(int flags) {  // bits 256 / 512 / 2048 / 4096 / 8192: kernel choice; bits 1..64: laboratory build only
#ifndef BSI_LAB
    BSI_CHECK_ARG(!(flags & 127), "bsi_conv_set_ablation: bits 1..64 switch kernel parts off (wrong results) and exist only in a laboratory "
                                  "build (make -C bsi_amd/csrc LAB=1 OUTDIR=<dir>, BSI_HIP_LIB=<dir>/libbsi_hip.so); got %d", flags);
#endif
    g_conv_abl = flags;
    return BSI_OK;
}

extern "C" int bsi_conv_set_grid_limit(int max_workgroups) {
    if (max_workgroups < 0 || max_workgroups % 8) {  // tiles are partitioned over the 8 XCDs: every XCD needs a workgroup
        bsi_set_error("bsi_conv_set_grid_limit: limit must be 0 or a multiple of 8, got %d", max_workgroups);
        return BSI_EINVAL;
    }
    g_conv_grid_limit = max_workgroups;
    return BSI_OK;
}

extern "C" int bsi_conv_nhwc_bf16(const bsi_conv_args* a, bsi_stream_t stream) {
    BSI_CHECK_ARG(a && a->x && a->w && a->out, "bsi_conv_nhwc_bf16: null pointer");
    BSI_CHECK_ARG(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, "bsi_conv_nhwc_bf16: bad sizes");
    BSI_CHECK_ARG(a->taps == 9 || a->taps == 1, "bsi_conv_nhwc_bf16: taps must be 9 (3x3) or 1 (1x1), got %d", a->taps);
    BSI_CHECK_ARG(a->Cin % 32 == 0 && a->Cin2 % 32 == 0 && a->Cout % 16 == 0,
                  "bsi_conv_nhwc_bf16: Cin=%d, Cin2=%d must be multiples of 32 and Cout=%d of 16", a->Cin, a->Cin2, a->Cout);
    BSI_CHECK_ARG(a->Cin2 == 0 || a->x2, "bsi_conv_nhwc_bf16: second source missing");
    BSI_CHECK_ARG(a->ldo % 8 == 0 && a->ldo >= a->Cout, "bsi_conv_nhwc_bf16: bad ldo");
    ConvParams p{};
    p.A = reinterpret_cast<const __bf16*>(a->x);
    p.A2 = reinterpret_cast<const __bf16*>(a->x2);
    p.W = reinterpret_cast<const __bf16*>(a->w);
    p.bias = a->bias;
    p.out = a->out;
    p.film = a->film; p.film_rows = a->film_rows > 0 ? a->film_rows : 1; p.film_stride = a->film_stride;
    p.resid = a->resid;
    p.M = a->B * a->H * a->W; p.N = a->Cout; p.Cin = a->Cin; p.Cin2 = a->Cin2; p.taps = a->taps;
    p.H = a->H; p.Wd = a->W; p.ldo = a->ldo;
    p.K = a->taps * a->Cin + a->Cin2;
    BSI_CHECK_ARG(p.K / 32 >= C_D, "bsi_conv_nhwc_bf16: K=%d too small", p.K);
    BSI_CHECK_ARG((size_t)p.M * (a->Cin > a->Cin2 ? a->Cin : a->Cin2) * 2 < (size_t)BUF_RECORDS - 256 && (size_t)p.N * p.K * 2 < (1ull << 32),
                  "bsi_conv_nhwc_bf16: tensors exceed the 32-bit offset range");
    BSI_CHECK_ARG(a->H < 65536 && a->W < 65536, "bsi_conv_nhwc_bf16: image too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a->epilogue) {
        case BSI_CONV_BIAS_BF16: return launch_conv<CEPI_BIAS_BF16>(p, s);
        case BSI_CONV_FILM_SILU_BF16:
            BSI_CHECK_ARG(a->film, "bsi_conv_nhwc_bf16: FILM epilogue needs the (scale, shift) table");
            if ((a->H * a->W) % 128 == 0) return launch_conv<CEPI_FILM_SILU_BF16>(p, s);
            return launch_conv<CEPI_FILM_ROWS_SILU_BF16>(p, s);
        case BSI_CONV_BIAS_RESID_F32:
            if (a->gn_partial) {
                // a wave's 128-row x 64-column block must be whole and lie in one image
                BSI_CHECK_ARG((a->H * a->W) % 128 == 0 && a->Cout % 64 == 0,
                              "bsi_conv_nhwc_bf16: GroupNorm partials need H*W %% 128 == 0 and Cout %% 64 == 0 (H*W=%d, Cout=%d)", a->H * a->W, a->Cout);
                p.gn_part = a->gn_partial;
                return launch_conv<CEPI_BIAS_RESID_F32_GN>(p, s);
            }
            return launch_conv<CEPI_BIAS_RESID_F32>(p, s);
        default: bsi_set_error("bsi_conv_nhwc_bf16: unknown epilogue %d", a->epilogue); return BSI_EINVAL;
    }
}
