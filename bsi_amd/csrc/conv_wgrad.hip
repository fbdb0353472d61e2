// Weight gradient of the 3x3 / 1x1 convolutions of the VDM-UNet path as an IMPLICIT transposed GEMM on bf16 MFMA:
//   dW[co][tap][ci] = sum_m dY[m, co] * X[m + shift(tap), ci]        (taps that fall outside the image contribute 0)
//   dWskip[co][c]   = sum_m dY[m, co] * X2[m, c]                      (the 1x1 skip conv folded into conv2's K range)
// i.e. autograd's conv2d weight gradient (bsi/nn/residual_block.py:40-48, bsi/nn/attention.py:29-30,
// bsi/models/vdm_unet.py:71 of the reference).  Both operands have the contraction index (pixel m) as the SLOW
// dimension, exactly like the Linear weight gradient, so the machinery is gemm_tn.hip's: stages of 32 pixels are DMA'd
// as they lie in memory (NHWC rows) and the MFMA fragments come from ds_read_b64_tr_b16.  The im2col matrix
// X_col[m][tap*Cin + ci] exists only as source addresses of the DMA: each 16-B chunk of a stage row is fetched from the
// pixel (y+dy, x+dx) of its tap, or from a zero page.
//
// Tile: 128 output channels x 512 im2col columns (four 128-column units); per stage 4 unit sub-tiles + 1 dY sub-tile of
// [32 pixels][256 B] = 40 KB, 4-slot ring = all 160 KB of LDS.  8 waves = 8 slices of 64 im2col columns, every wave
// against all 128 dY columns.  The pixel range is split over workgroups; partial tiles go to fp32 slabs summed by
// reduce_slabs (deterministic).  Output layout = the forward's packed weight layout [Cout][taps*Cin + Cin2].
#include <stdlib.h>

#include "common.h"

int bsi_reduce_slabs_launch(const float* slabs, size_t slab_stride, int splits, size_t n, int accumulate, float* out,
                            hipStream_t s);
int bsi_reduce_slabs2_launch(const float* slabsA, size_t strideA, size_t nA, float* outA, const float* slabsB, size_t strideB, size_t nB,
                             float* outB, int splits, int accumulate, hipStream_t s);

namespace {

struct WgParams {
    const __bf16* dY;     // [M, ldy]
    const __bf16* X;      // [M, Cin]
    const __bf16* X2;     // [M, Cin2] or null
    const __bf16* zeros;  // >= 256 B of zeros
    float* out;           // [splits][Cout][Ktot]
    float* colsum;        // optional [splits][Cout]: column sums of dY (bias gradient), written by the first unit tile's workgroups
    int M, H, Wd, HW, Cin, Cin2, taps, Cout, ldy, Ktot;
    int tiles_u, tiles_n, splits, m_per_split, units;
    size_t slab_stride;
    int abl;  // experiments (BSI_WGRAD_ABL, tools/wgrad_bench.py): 1 no DMA after the prologue, 2 no fragment reads, 4 no MFMA
};

constexpr int G_RB = 256;          // bytes per sub-tile row (128 bf16 columns)
constexpr int G_SUB = 32 * G_RB;   // one sub-tile of a stage: 32 pixels
// A stage = UNITS im2col units + dY; UNITS = 4: 40 KB stages, ring of 4; UNITS = 3: 32 KB stages, ring of 5 (160 KB either way).
// UNITS = 3 when the K range is a multiple of 384 columns but not of 512 (K = 1152 = 9 units: 3 tiles of 3 units instead of
// 3 tiles of 4 with a quarter of the MFMAs wasted on padding; K = 2304: 6 x 3 instead of 5 x 4).

__device__ __forceinline__ int wg_swz(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

template <int UNITS>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgParams p) {
    constexpr int G_SLOT = (UNITS + 1) * G_SUB;
    constexpr int G_R = 160 * 1024 / G_SLOT, G_D = G_R - 1;  // 4 slots of 40 KB or 5 of 32 KB
    constexpr int NI = UNITS + 1;                           // DMA instructions per stage and wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // ping-pong group

    // XCD-aware order: workgroup b runs on XCD b % 8, which has its own L2.  The tiles of one split read the same dY rows and the
    // same X rows (shifted by their taps), so each XCD gets a contiguous range of (split, tile) pairs with the tile index fastest.
    const int tiles = p.tiles_n * p.tiles_u;
    const int total = tiles * p.splits, per_xcd = (total + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= total) return;
    const int tile = logical % tiles, split = logical / tiles;
    const int n0 = (tile / p.tiles_u) * 128, u0 = (tile % p.tiles_u) * UNITS;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 31) / 32;
    if (nk <= 0) return;  // uniform per workgroup

    // staging: a wave-instruction covers 4 pixel rows x 256 B; wave w fills rows 4w..4w+3 of every sub-tile (5 per stage).
    // Sources are buffer resources (X, X2, dY); a lane's byte offset is pixel*pitch + a per-lane constant (tap shift and channel
    // of its im2col column), and anything that must read as zero -- taps outside the image, columns beyond K, pixels beyond the
    // split -- gets an out-of-range offset, for which the LDS-DMA writes zeros (conv_igemm.hip).  The pixel's (y, x) is carried
    // from stage to stage instead of being divided out (stages are issued in order).
    constexpr unsigned OOB = 0xC0000000u;  // >= num_records even after adding a second OOB or a small negative constant
    const int srow = 4 * wave + (lane >> 4), spos = lane & 15;
    const int schunk = spos ^ wg_swz(srow);  // source chunk that lands at this lane's LDS position
    const int conv_k = p.taps * p.Cin;
    unsigned uconst[UNITS];  // per unit: shift*pitch + channel bytes of this lane's chunk, or OOB beyond Ktot
    int utap[UNITS];         // per unit: border bits that invalidate this lane's tap (conv_igemm.hip: tap_mask)
    bool lsrc2[UNITS];       // per unit: this lane's column belongs to the second source
    bool usrc2[UNITS];       // per unit (wave-uniform): every column belongs to the second source
    bool ustr[UNITS];        // per unit (wave-uniform): the unit straddles the two sources -> one DMA instruction per source
    int extra = 0;       // straddling units of this tile (0 or 1): DMA instructions per stage = NI + extra
#pragma unroll
    for (int q = 0; q < UNITS; ++q) {
        const int vc = (u0 + q) * 128 + schunk * 8;
        usrc2[q] = (u0 + q) * 128 >= conv_k;
        ustr[q] = p.Cin2 > 0 && (u0 + q) * 128 < conv_k && (u0 + q) * 128 + 128 > conv_k;
        extra += ustr[q] ? 1 : 0;
        lsrc2[q] = vc >= conv_k;
        uconst[q] = OOB; utap[q] = 0;
        if (vc < conv_k) {
            const int tap = vc / p.Cin, c = vc - tap * p.Cin;
            int dy = 0, dx = 0;
            if (p.taps == 9) { dy = tap / 3 - 1; dx = tap % 3 - 1; }
            uconst[q] = (unsigned)((dy * p.Wd + dx) * p.Cin * 2 + c * 2);
            utap[q] = (dy < 0 ? 1 : 0) | (dy > 0 ? 2 : 0) | (dx < 0 ? 4 : 0) | (dx > 0 ? 8 : 0);
        } else if (vc < p.Ktot) {
            uconst[q] = (unsigned)((vc - conv_k) * 2);
        }
    }
    const int ycol = n0 + schunk * 8;
    const unsigned yconst = ycol < p.Cout ? (unsigned)(ycol * 2) : OOB;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.X), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.X2 ? p.X2 : p.X), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dY), 0, 0x80000000u, 0x00020000);
    // running pixel of this lane's row in the stage to issue next
    int sm = mbeg + srow, sy, sx;
    {
        const int rem = sm % p.HW;
        sy = rem / p.Wd;
        sx = rem - sy * p.Wd;
    }
    const int q32 = 32 / p.Wd, r32 = 32 - q32 * p.Wd;
    const bool small_h = p.H < q32 + 2;  // rare: more than one image per step

    auto stage = [&](int slot_i) {  // stages are issued in order: v = 0, 1, 2, ...
        char* base = lds + slot_i * G_SLOT + wave * 1024;
        const bool m_ok = sm < mend;
        const int code = (sy == 0 ? 1 : 0) | (sy == p.H - 1 ? 2 : 0) | (sx == 0 ? 4 : 0) | (sx == p.Wd - 1 ? 8 : 0);
        const unsigned mpx = m_ok ? (unsigned)sm * (unsigned)(p.Cin * 2) : OOB;
        const unsigned mpx2 = m_ok ? (unsigned)sm * (unsigned)(p.Cin2 * 2) : OOB;
        const unsigned mpy = m_ok ? (unsigned)sm * (unsigned)(p.ldy * 2) : OOB;
#pragma unroll
        for (int q = 0; q < UNITS; ++q) {
            const unsigned vo = (code & utap[q]) ? OOB : (lsrc2[q] ? mpx2 : mpx) + uconst[q];
            if (ustr[q]) {  // divergent on purpose: two instructions with complementary exec masks fill disjoint lanes
                if (lsrc2[q]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, LDS_PTR(base + q * G_SUB), 16, vo, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(base + q * G_SUB), 16, vo, 0, 0, 0);
            } else if (usrc2[q]) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, LDS_PTR(base + q * G_SUB), 16, vo, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(base + q * G_SUB), 16, vo, 0, 0, 0);
            }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(base + UNITS * G_SUB), 16, mpy + yconst, 0, 0, 0);
        // next stage: 32 pixels on
        sm += 32;
        sx += r32;
        sy += q32;
        if (sx >= p.Wd) { sx -= p.Wd; ++sy; }
        if (small_h) sy %= p.H;
        else if (sy >= p.H) sy -= p.H;
    };

    // fragments (gemm_tn.hip): lane = (q4, qp, pp): q4 = lane>>4 selects pixels 8q4..8q4+7 of the stage, qp the row inside
    // a 4-row transpose block, pp the 4-column quad
    const int q4 = lane >> 4, qp = (lane & 15) >> 2, pp = lane & 3;
    const int mA = 8 * q4 + qp, mB = mA + 4;
    const int swA = wg_swz(mA), swB = wg_swz(mB);
    const int rowA = mA * G_RB, rowB = mB * G_RB;
    const int half8 = 8 * (pp & 1), chp = pp >> 1;
    const int wcol0 = 16 * UNITS * wave;  // first im2col column (inside the tile) of this wave's slice of 16 * UNITS columns

    f32x4 acc[UNITS][8];
#pragma unroll
    for (int i = 0; i < UNITS; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[UNITS], bf[8];
    // fused bias gradient (gemm_tn.hip): column sums of the dY tile = one extra MFMA per stage with an all-ones A operand
    // (pixels beyond the split are zero rows already); wave w takes output channels 16w .. 16w+15 of the tile
    const bool do_colsum = p.colsum != nullptr && (tile % p.tiles_u) == 0;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
    bf16x8 onesf;
#pragma unroll
    for (int e = 0; e < 8; ++e) onesf[e] = (__bf16)1.0f;

#define WG_BARRIER()                             \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

    auto load_frags = [&](const char* b) {
        union U { bf16x8 v; s16x4 h[2]; };
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {  // im2col columns wcol0 + 16i + 4pp: unit (wcol0 + 16i) / 128, 16-B chunk pair inside it
            const int col = wcol0 + 16 * i;
            const int xunit = (col >> 7) * G_SUB, ch = ((col & 127) >> 3) + chp;
            U u;
            u.h[0] = TR(b + xunit + rowA + ((ch ^ swA) << 4) + half8);
            u.h[1] = TR(b + xunit + rowB + ((ch ^ swB) << 4) + half8);
            af[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // dY columns 16j + 4pp
            const int ch = 2 * j + chp;
            U u;
            u.h[0] = TR(b + UNITS * G_SUB + rowA + ((ch ^ swA) << 4) + half8);
            u.h[1] = TR(b + UNITS * G_SUB + rowB + ((ch ^ swB) << 4) + half8);
            bf[j] = u.v;
        }
    };

#pragma unroll
    for (int d = 0; d < G_D; ++d)
        if (d < nk) stage(d);
    extra = __builtin_amdgcn_readfirstlane(extra);
    if (nk >= G_D && !extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (G_D - 1)) : "memory");
    else if (nk >= G_D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI + 1) * (G_D - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_BARRIER();
    if (grp == 1) WG_BARRIER();

    int slot = 0, pslot = G_D;
    for (int v = 0; v < nk; ++v) {
        if (!(p.abl & 2)) load_frags(lds + slot * G_SLOT);
        if (v + G_D < nk && !(p.abl & 1)) {
            stage(pslot);
            if (!extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (G_D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI + 1) * (G_D - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WG_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        if (!(p.abl & 4)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < UNITS; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_colsum) {
            bf16x8 bsel = bf[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) bsel = wave == j ? bf[j] : bsel;
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onesf, bsel, accb, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        WG_BARRIER();
        slot = (slot == G_R - 1) ? 0 : slot + 1;
        pslot = (pslot == G_R - 1) ? 0 : pslot + 1;
    }
    if (grp == 0) WG_BARRIER();
#undef WG_BARRIER
#undef TR

    if (do_colsum && lane < 16) {  // D row 0 (lanes 0..15, register 0) holds the column sums
        const int n = n0 + 16 * wave + lane;
        if (n < p.Cout) p.colsum[(size_t)split * p.Cout + n] = accb[0];
    }
    // D rows = im2col column (4*(lane>>4) + reg inside slice tile i), D cols = output channel (lane & 15 inside tile j)
    float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 16 * j + (lane & 15);
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int k = u0 * 128 + wcol0 + 16 * i + 4 * q4;
            if (k < p.Ktot) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(out + (size_t)n * p.Ktot + k));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3, image width 32, C_in a multiple of 128, no second source: the three taps of one filter ROW share their pixels.
// A stage is one image row of 32 output pixels; the im2col units of taps (dy, -1), (dy, 0), (dy, +1) for a 128-channel block are
// the SAME input row y + dy shifted by one pixel, so the stage DMAs that row ONCE into LDS rows 1..32 of a 34-row image whose rows
// 0 and 33 (x = -1 and x = 32) are zeros written at kernel start, and unit q reads its fragments at row offset q.  Per stage and
// wave: 1 X + 1 dY DMA instruction instead of 3 + 1 (the generic kernel's operand DMA is a third of its time,
// tools/wgrad_bench.py).  Tile = (filter row dy, channel block): 128 output channels x 3 x 128 im2col columns; everything else
// -- ping-pong groups, ring of 5 stages, transposed fragment reads, split-M slabs, fused bias gradient -- as above.
__global__ __launch_bounds__(512) void conv_wgrad_halo_kernel(const WgParams p) {
    constexpr int UNITS = 3;
    // a stage = TWO image rows (64 output pixels): X images of 34 rows each (36 allocated), then the 64 dY rows; ring of 4 stages.
    // Two rows per phase halve the barriers and loop overhead per MFMA (48 MFMAs per wave and phase instead of 24).
    constexpr int H_X = 36 * G_RB, H_Y = 2 * H_X, H_SLOT = H_Y + 64 * G_RB;  // 9216 + 9216 + 16384 = 34816 B
    constexpr int G_R = 4, G_D = G_R - 1;
    constexpr int NI = 4;  // DMA instructions per stage and wave: X and dY of both rows
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;

    const int tiles = p.tiles_n * p.tiles_u;
    const int total = tiles * p.splits, per_xcd = (total + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= total) return;
    const int tile = logical % tiles, split = logical / tiles;
    const int n0 = (tile / p.tiles_u) * 128, tu = tile % p.tiles_u;
    const int cblocks = p.Cin >> 7;
    const int cb = tu % cblocks, dyi = tu / cblocks, dy = dyi - 1;  // channel block, filter row
    const int mbeg = split * p.m_per_split;                          // multiple of 32 (plan): image rows
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int nk = (mend - mbeg + 63) / 64;
    if (nk <= 0) return;

    // zero rows 0 and 33 of both X images of every slot (never written again)
    for (int i = tid; i < G_R * 4 * (G_RB / 16); i += 512) {
        const int slot_i = i / (4 * (G_RB / 16)), which = (i / (G_RB / 16)) & 3, c16 = i % (G_RB / 16);
        *reinterpret_cast<u32x4*>(lds + slot_i * H_SLOT + (which >> 1) * H_X + ((which & 1) ? 33 : 0) * G_RB + c16 * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    constexpr unsigned OOB = 0xC0000000u;
    const int srow = 4 * wave + (lane >> 4), spos = lane & 15;  // pixel x of this lane's X / dY row, 16-B chunk position
    const unsigned xconst = (unsigned)(cb * 256 + ((spos ^ wg_swz(srow + 1)) << 4));  // X lands in LDS row srow + 1
    const int ycol = n0 + ((spos ^ wg_swz(srow)) << 3);
    const unsigned yconst = ycol < p.Cout ? (unsigned)(ycol * 2) : OOB;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.X), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dY), 0, 0x80000000u, 0x00020000);
    int sm = mbeg + srow;               // this lane's output pixel in the first row of the stage to issue next
    int sy = (mbeg >> 5) % p.H;         // image row of that row (wave-uniform)
    auto stage = [&](int slot_i) {
        char* base = lds + slot_i * H_SLOT + wave * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool m_ok = sm < mend;
            const bool row_ok = dy < 0 ? sy > 0 : dy > 0 ? sy < p.H - 1 : true;
            const unsigned vx = (m_ok && row_ok) ? (unsigned)(sm + dy * 32) * (unsigned)(p.Cin * 2) + xconst : OOB;
            const unsigned vy = m_ok ? (unsigned)sm * (unsigned)(p.ldy * 2) + yconst : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(base + h * H_X + G_RB), 16, vx, 0, 0, 0);  // rows 1 + 4 wave ..
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(base + H_Y + h * 32 * G_RB), 16, vy, 0, 0, 0);
            sm += 32;
            sy = sy + 1 == p.H ? 0 : sy + 1;
        }
    };

    const int q4 = lane >> 4, qp = (lane & 15) >> 2, pp = lane & 3;
    const int mA = 8 * q4 + qp, mB = mA + 4;
    const int half8 = 8 * (pp & 1), chp = pp >> 1;
    const int wcol0 = 16 * UNITS * wave;  // first of this wave's 48 im2col columns inside the tile (3 units x 128 channels)
    // X fragment i: unit q = column / 128 reads pixel rows shifted by q; addresses are per-lane constants of the kernel
    int xa[UNITS], xb[UNITS];
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
        const int col = wcol0 + 16 * i, q = col >> 7, ch = ((col & 127) >> 3) + chp;
        xa[i] = (mA + q) * G_RB + ((ch ^ wg_swz(mA + q)) << 4) + half8;
        xb[i] = (mB + q) * G_RB + ((ch ^ wg_swz(mB + q)) << 4) + half8;
    }
    const int swA = wg_swz(mA), swB = wg_swz(mB);
    const int rowA = mA * G_RB, rowB = mB * G_RB;

    f32x4 acc[UNITS][8];
#pragma unroll
    for (int i = 0; i < UNITS; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][UNITS], bf[2][8];
    const bool do_colsum = p.colsum != nullptr && tu == 0;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
    bf16x8 onesf;
#pragma unroll
    for (int e = 0; e < 8; ++e) onesf[e] = (__bf16)1.0f;

#define WG_BARRIER()                             \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr))

    auto load_frags = [&](const char* b) {
        union U { bf16x8 v; s16x4 h[2]; };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                U u;
                u.h[0] = TR(b + h * H_X + xa[i]);
                u.h[1] = TR(b + h * H_X + xb[i]);
                af[h][i] = u.v;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 2 * j + chp;
                U u;
                u.h[0] = TR(b + H_Y + h * 32 * G_RB + rowA + ((ch ^ swA) << 4) + half8);
                u.h[1] = TR(b + H_Y + h * 32 * G_RB + rowB + ((ch ^ swB) << 4) + half8);
                bf[h][j] = u.v;
            }
        }
    };

#pragma unroll
    for (int d = 0; d < G_D; ++d)
        if (d < nk) stage(d);
    if (nk >= G_D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (G_D - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zero rows
    WG_BARRIER();
    if (grp == 1) WG_BARRIER();

    int slot = 0, pslot = G_D;
    for (int v = 0; v < nk; ++v) {
        if (!(p.abl & 2)) load_frags(lds + slot * H_SLOT);
        if (v + G_D < nk && !(p.abl & 1)) {
            stage(pslot);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (G_D - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WG_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        if (!(p.abl & 4)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i0 = 0; i0 < UNITS; ++i0) {  // boustrophedon: one operand changes per MFMA (power; gemm_bf16.hip)
                        const int i = (j & 1) ? UNITS - 1 - i0 : i0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][i], bf[h][j], acc[i][j], 0, 0, 0);
                    }
        }
        if (do_colsum) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 bsel = bf[h][0];
#pragma unroll
                for (int j = 1; j < 8; ++j) bsel = wave == j ? bf[h][j] : bsel;
                accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onesf, bsel, accb, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        WG_BARRIER();
        slot = (slot == G_R - 1) ? 0 : slot + 1;
        pslot = (pslot == G_R - 1) ? 0 : pslot + 1;
    }
    if (grp == 0) WG_BARRIER();
#undef WG_BARRIER
#undef TR

    if (do_colsum && lane < 16) {
        const int n = n0 + 16 * wave + lane;
        if (n < p.Cout) p.colsum[(size_t)split * p.Cout + n] = accb[0];
    }
    // D rows = im2col column of the tile (4*(lane>>4) + reg inside slice tile i), D cols = output channel (lane & 15 inside tile j);
    // column c of unit q is K index (3 dyi + q) * Cin + 128 cb + c of the packed weight layout
    float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 16 * j + (lane & 15);
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int colt = wcol0 + 16 * i + 4 * q4, q = colt >> 7, c = colt & 127;
            const int k = (3 * dyi + q) * p.Cin + 128 * cb + c;
            __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(out + (size_t)n * p.Ktot + k));
        }
    }
}

// Finishing kernel of the UNet training engine: sums the split slabs in fixed order AND writes the torch Conv2d layout directly
//   w_out[o][c][tap] = sum_s slab[s][o][tap * cin_pad + c]   (c < Cin),   w2_out[o][c] = sum_s slab[s][o][taps * cin_pad + c],
//   dbias[o] = sum_s colsum[s][o]
// in one launch (the packed result + bsi_conv_wgrad_unpack + two reductions cost three launches and a pass more).  Threads walk
// the PACKED index (coalesced slab reads, 4 columns each, 4 threads sharing the slabs of an output as in reduce_slabs_par_kernel);
// the 4-B writes into the Conv2d layout are scattered but few.  Blocks [0, gW) weights + skip weights, the rest bias.
__global__ __launch_bounds__(256) void reduce_slabs_conv2d_kernel(const float* __restrict__ slabs, size_t slab_stride, int splits, int Cout,
                                                                  int Ktot, int Cin, int cin_pad, int taps, int Cin2,
                                                                  float* __restrict__ w_out, float* __restrict__ w2_out,
                                                                  const float* __restrict__ cs_slabs, float* __restrict__ dbias, int gW) {
    __shared__ f32x4 part[3][64];
    const int o64 = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool bias_job = (int)blockIdx.x >= gW;
    const size_t n4 = bias_job ? (size_t)(Cout / 4) : (size_t)Cout * Ktot / 4;
    const size_t i = (size_t)(bias_job ? blockIdx.x - gW : blockIdx.x) * 64 + o64;
    const float* src = bias_job ? cs_slabs : slabs;
    const size_t stride = bias_job ? (size_t)Cout : slab_stride;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = g;
        for (; s + 12 < splits; s += 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(s + 4 * u) * stride) + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += v[u][e];
        }
        for (; s < splits; s += 4) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)s * stride) + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
    }
    if (g > 0) part[g - 1][o64] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += part[k][o64][e];
        if (bias_job) {
            reinterpret_cast<f32x4*>(dbias)[i] = a;
        } else {
            const int o = (int)((i * 4) / Ktot), col = (int)((i * 4) % Ktot);  // Ktot % 4 == 0: the four columns share o
            const int conv_k = taps * cin_pad;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = col + e;
                if (k < conv_k) {
                    const int tap = k / cin_pad, c = k - tap * cin_pad;
                    if (c < Cin) w_out[((size_t)o * Cin + c) * taps + tap] = a[e];
                } else if (w2_out && k - conv_k < Cin2) {
                    w2_out[(size_t)o * Cin2 + (k - conv_k)] = a[e];
                }
            }
        }
    }
}

void plan(WgParams& p, int cus = 0) {
    const int g_wg_cus = cus > 0 ? cus : compute_cus();  // one workgroup per CU per round (minus the CUs reserved for communication)
    {   // 3 or 4 units of 128 im2col columns per tile: whichever pads the K range less (4 on a tie)
        const int nu = (p.Ktot + 127) / 128;
        const int pad4 = (nu + 3) / 4 * 4 - nu, pad3 = (nu + 2) / 3 * 3 - nu;
        p.units = pad3 < pad4 ? 3 : 4;
    }
    p.tiles_u = (p.Ktot + 128 * p.units - 1) / (128 * p.units);
    p.tiles_n = (p.Cout + 127) / 128;
    const int tiles = p.tiles_u * p.tiles_n;
    const int max_s = (p.M + 255) / 256;  // at least 8 stages per split
    // one workgroup per CU at a time (160 KB of LDS): time ~ rounds(s) / s with rounds = ceil(tiles * s / CUs); the smallest s
    // among the best (rounding CUs / tiles UP gave 3 x 86 = 258 workgroups = two rounds for the 128 -> 128 convolution)
    int s = 1;
    double best_t = 1e30;
    for (int c = 1; c <= g_wg_cus && c <= max_s; ++c) {
        const double t = (double)((tiles * c + g_wg_cus - 1) / g_wg_cus) / c;
        if (t < best_t * 0.97) { best_t = t; s = c; }  // more splits only for a real gain: every split is a slab to write and reduce
    }
    int mps = (p.M + s - 1) / s;
    mps = (mps + 31) / 32 * 32;
    p.m_per_split = mps;
    p.splits = (p.M + mps - 1) / mps;
    p.slab_stride = (size_t)p.Cout * p.Ktot;
}

// packed fp32 [Cout][ld] (K index (tap, channel), channels padded to cin_pad, columns from col0) -> torch Conv2d layout
// [Cout][Cin][taps]; accumulate != 0 adds.
__global__ void conv_wgrad_unpack_kernel(const float* __restrict__ packed, int Cout, int Cin, int taps, int cin_pad, int ld,
                                         int col0, int accumulate, float* __restrict__ out) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int c = (int)((i / taps) % Cin);
        const int o = (int)(i / ((size_t)taps * Cin));
        const float v = packed[(size_t)o * ld + col0 + tap * cin_pad + c];
        out[i] = accumulate ? out[i] + v : v;
    }
}

}  // namespace

extern "C" size_t bsi_conv_wgrad_workspace_bytes(int M, int Cin, int Cin2, int Cout, int taps) {
    if (M <= 0 || Cin <= 0 || Cout <= 0) return 0;
    WgParams p{};
    p.M = M; p.Cin = Cin; p.Cin2 = Cin2; p.Cout = Cout; p.taps = taps; p.Ktot = taps * Cin + Cin2;
    // the split count depends on the CUs the launch may fill: size for every reserve bsi_set_cu_reserve accepts, so that a
    // workspace allocated before the reserve changed still fits
    size_t worst = 0;
    for (int cus = device_cus(); cus >= 8 && cus >= device_cus() - BSI_MAX_CU_RESERVE; cus -= 8) {
        plan(p, cus);
        const size_t b = (size_t)p.splits * p.slab_stride * sizeof(float) + (size_t)p.splits * (size_t)Cout * sizeof(float);  // + bias-gradient slabs
        worst = b > worst ? b : worst;
    }
    return worst;
}

struct Conv2dOut {  // optional: write the torch Conv2d layout instead of the packed one (engine-internal entry below)
    float* w;   // [Cout][cin_logical][taps]
    float* w2;  // [Cout][Cin2] or null
    int cin_logical;
};

static int conv_wgrad_impl(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B, int H, int W, int Cin,
                           int Cin2, int Cout, int taps, float* out_packed, float* dbias, int accumulate, void* workspace,
                           bsi_stream_t stream, const Conv2dOut* c2d = nullptr) {
    BSI_CHECK_ARG(dy && x && zeros && (out_packed || c2d) && workspace, "bsi_conv_wgrad: null pointer");
    BSI_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (taps == 9 || taps == 1), "bsi_conv_wgrad: bad sizes");
    BSI_CHECK_ARG(Cin % 8 == 0 && Cin2 % 8 == 0 && Cout % 8 == 0 && ldy % 8 == 0 && ldy >= Cout,
                  "bsi_conv_wgrad: Cin=%d Cin2=%d Cout=%d ldy=%d must be multiples of 8", Cin, Cin2, Cout, ldy);
    BSI_CHECK_ARG(Cin2 == 0 || x2, "bsi_conv_wgrad: second source missing");
    BSI_CHECK_ARG((size_t)B * H * W * (size_t)(Cin > Cin2 ? (Cin > ldy ? Cin : ldy) : (Cin2 > ldy ? Cin2 : ldy)) * 2 < 0x7FF00000ull,
                  "bsi_conv_wgrad: tensors exceed the 31-bit buffer offset range");
    WgParams p{};
    p.dY = reinterpret_cast<const __bf16*>(dy);
    p.X = reinterpret_cast<const __bf16*>(x);
    p.X2 = reinterpret_cast<const __bf16*>(x2);
    p.zeros = reinterpret_cast<const __bf16*>(zeros);
    p.out = reinterpret_cast<float*>(workspace);
    p.M = B * H * W; p.H = H; p.Wd = W; p.HW = H * W; p.Cin = Cin; p.Cin2 = Cin2; p.taps = taps; p.Cout = Cout; p.ldy = ldy;
    p.Ktot = taps * Cin + Cin2;
    plan(p);
    {
        const char* e = getenv("BSI_WGRAD_ABL");  // kernel experiments only; results are wrong with flags != 0
        p.abl = e ? atoi(e) : 0;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    p.colsum = dbias ? p.out + (size_t)p.splits * p.slab_stride : nullptr;
    set_max_lds(reinterpret_cast<const void*>(conv_wgrad_kernel<4>), 160 * 1024);
    set_max_lds(reinterpret_cast<const void*>(conv_wgrad_kernel<3>), 160 * 1024);
    const dim3 grid(8 * ((p.tiles_n * p.tiles_u * p.splits + 7) / 8));  // whole rounds of the 8 XCDs (the kernel drops the excess)
    // one filter row's three taps share their pixels (conv_wgrad_halo_kernel): 3x3, width 32, C_in % 128 == 0, no second source
    static const bool no_halo = getenv("BSI_WGRAD_NO_HALO") != nullptr;
    const bool halo = !no_halo && taps == 9 && W == 32 && Cin % 128 == 0 && Cin2 == 0 && p.units == 3 && p.tiles_u == 3 * (Cin / 128) &&
                      p.m_per_split % 32 == 0;
    if (halo) {
        set_max_lds(reinterpret_cast<const void*>(conv_wgrad_halo_kernel), 160 * 1024);
        hipLaunchKernelGGL(conv_wgrad_halo_kernel, grid, dim3(512), 160 * 1024, s, p);
    } else if (p.units == 3) hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(512), 160 * 1024, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<4>, grid, dim3(512), 160 * 1024, s, p);
    BSI_CHECK_LAUNCH("bsi_conv_wgrad");
    if (c2d) {  // slabs -> Conv2d layout (+ skip weights, + bias) in one finishing launch
        const int gW = (int)(((size_t)Cout * p.Ktot / 4 + 63) / 64), gB = (Cout / 4 + 63) / 64;
        hipLaunchKernelGGL(reduce_slabs_conv2d_kernel, dim3(gW + gB), dim3(256), 0, s, p.out, p.slab_stride, p.splits, Cout, p.Ktot,
                           c2d->cin_logical, Cin, taps, Cin2, c2d->w, c2d->w2, p.colsum, dbias, gW);
        BSI_CHECK_LAUNCH("bsi_conv_wgrad_conv2d");
        return BSI_OK;
    }
    if (dbias)  // weight slabs and bias slabs summed by one launch
        return bsi_reduce_slabs2_launch(p.out, p.slab_stride, p.slab_stride, out_packed, p.colsum, (size_t)Cout, (size_t)Cout, dbias, p.splits,
                                        accumulate, s);
    return bsi_reduce_slabs_launch(p.out, p.slab_stride, p.splits, p.slab_stride, accumulate, out_packed, s);
}

// engine-internal (unet_ops.h): weight + bias gradient written in the torch Conv2d layout, w [Cout][cin_logical][taps] (the kernel
// runs on Cin >= cin_logical padded channels), w2 [Cout][Cin2] for folded 1x1 skip columns; outputs are WRITTEN
int bsi_conv_wgrad_conv2d_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B, int H, int W,
                                    int Cin, int cin_logical, int Cin2, int Cout, int taps, float* w, float* w2, float* dbias,
                                    void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(w && dbias && cin_logical > 0 && cin_logical <= Cin && Cout % 4 == 0 && (Cin2 == 0 || w2),
                  "bsi_conv_wgrad_conv2d: bad args");
    const Conv2dOut c2d{w, w2, cin_logical};
    return conv_wgrad_impl(dy, ldy, x, x2, zeros, B, H, W, Cin, Cin2, Cout, taps, nullptr, dbias, 0, workspace, stream, &c2d);
}

extern "C" int bsi_conv_wgrad_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B,
                                        int H, int W, int Cin, int Cin2, int Cout, int taps, float* out_packed, int accumulate,
                                        void* workspace, bsi_stream_t stream) {
    return conv_wgrad_impl(dy, ldy, x, x2, zeros, B, H, W, Cin, Cin2, Cout, taps, out_packed, nullptr, accumulate, workspace, stream);
}

extern "C" int bsi_conv_wgrad_bias_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B,
                                             int H, int W, int Cin, int Cin2, int Cout, int taps, float* out_packed, float* dbias,
                                             int accumulate, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(dbias && Cout % 4 == 0, "bsi_conv_wgrad_bias: bias gradient pointer missing or Cout %% 4");
    return conv_wgrad_impl(dy, ldy, x, x2, zeros, B, H, W, Cin, Cin2, Cout, taps, out_packed, dbias, accumulate, workspace, stream);
}

extern "C" int bsi_conv_wgrad_unpack(const float* packed, int Cout, int Cin, int taps, int cin_pad, int ld, int col0,
                                     int accumulate, float* out, bsi_stream_t stream) {
    BSI_CHECK_ARG(packed && out && Cout > 0 && Cin > 0 && taps > 0 && cin_pad >= Cin && ld >= col0 + taps * cin_pad,
                  "bsi_conv_wgrad_unpack: bad args");
    const size_t total = (size_t)Cout * Cin * taps;
    size_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), packed, Cout,
                       Cin, taps, cin_pad, ld, col0, accumulate, out);
    BSI_CHECK_LAUNCH("bsi_conv_wgrad_unpack");
    return BSI_OK;
}
