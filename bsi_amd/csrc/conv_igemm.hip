// 3x3 / 1x1 convolution (stride 1, zero padding) as an IMPLICIT GEMM on bf16 MFMA for gfx950.
// Replaces nn.Conv2d on the VDM-UNet path of the reference (bsi/models/vdm_unet.py:71-72,
// bsi/nn/residual_block.py:40-48, bsi/nn/attention.py:29-30), together with the elementwise ops torch runs after it.
//
// Activations are NHWC bf16 ([B*H*W pixels][C channels]); weights are pre-arranged [C_out][tap][C_in] so that the GEMM
// K index is (tap, channel).  No im2col buffer exists: the A tile of K step v (32 channels of one filter tap) is
// fetched by global_load_lds straight from the input pixel (y+dy, x+dx) of each output pixel; taps that fall outside the
// image read a zero page instead.  An optional second source adds 1x1 "skip" K steps (the 1x1 conv on the residual path
// of the up blocks, residual_block.py:40,63, folded into the second 3x3 conv's accumulation).
//
// Schedule = the persistent deep-ring ping-pong of gemm_bf16.hip variant 6 with tile 512 (pixels) x 128 (C_out):
// C_out is 128 everywhere in the UNet, so the tile is made tall instead of wide; waves keep the 128 x 64 sub-tile.
// LDS: 3 stages of (512 + 128) rows x 64 B = 120 KB + 32 KB epilogue scratch.
#include "common.h"

namespace {

struct ConvParams {
    const __bf16* A;      // NHWC input [B*H*W, Cin]
    const __bf16* A2;     // optional second source [B*H*W, Cin2] for the 1x1 skip K steps
    const __bf16* W;      // [N][taps*Cin + Cin2]
    const float* bias;    // [N]
    const __bf16* zeros;  // >= 64 B of zeros
    void* out;            // bf16 or fp32 [M, ldo]
    const float* film;    // FILM: [rows][2N] = (scale, shift) per image row
    const float* resid;   // BIAS_RESID_F32: fp32 [M, ldo] or null
    int film_rows, film_stride;
    int M, N, Cin, Cin2, taps, H, Wd, ldo;
    int K;                // taps*Cin + Cin2
    int tiles_m, tiles_n;
};

constexpr int C_BM = 512, C_BN = 128, C_RB = 64, C_R = 3, C_D = C_R - 1;
constexpr int C_SLOT = (C_BM + C_BN) * C_RB;  // 40 KB
constexpr int C_ABYTES = C_BM * C_RB;

enum { CEPI_BIAS_BF16 = 0, CEPI_FILM_SILU_BF16 = 1, CEPI_BIAS_RESID_F32 = 2 };

template <int EPI>
__global__ __launch_bounds__(512) void conv_ring_kernel(const ConvParams p) {
    constexpr int TM = 8, NW = 8;
    constexpr bool BF16_OUT = (EPI != CEPI_BIAS_RESID_F32);
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;          // ping-pong group = 256-row half of the tile
    const int wmm = (wave >> 1) & 1;   // 128-row quarter inside the group
    const int wn = wave & 1;           // 64-column half
    char* scratch = lds + C_R * C_SLOT + wave * 4096;

    const int nwg = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int wpx = (gridDim.x + 7 - xcd) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int lo = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + wl;
    if (tile >= hi) return;

    // staging: 40 wave-instructions (16 rows x 64 B) per stage, wave w issues instruction slots q*8 + w, q < 5:
    // slots 0..31 = activation rows (pixels), 32..39 = weight rows
    const int srow = lane >> 2, spos = lane & 3;
    const int HW = p.H * p.Wd;
    int pix_m[4], pix_yx[4];  // per A slot: (clamped) output pixel index of this lane's row and its packed (y << 16 | x)
    const int achunk = (spos ^ ((-(srow >> 2)) & 3)) * 16;  // swizzled 16-B chunk of this lane (same for every slot)
    unsigned woffs;           // byte offset of this lane's weight row (+ chunk) from p.W
    auto set_sources = [&](int t) {
        const int m0 = (t / p.tiles_n) * C_BM, n0 = (t % p.tiles_n) * C_BN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (q * NW + wave) * 16 + srow;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            const int rem = m % HW;
            pix_m[q] = m;
            pix_yx[q] = ((rem / p.Wd) << 16) | (rem % p.Wd);
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
    const char* Zb = reinterpret_cast<const char*>(p.zeros);
    auto stage = [&](int v, int slot) {
        char* base = lds + slot * C_SLOT;
        const int k0 = v * 32;
        const int conv_k = p.taps * p.Cin;
        int dy = 0, dx = 0, cbytes, rowb;
        const char* srcb;  // wave-uniform source tensor
        if (k0 < conv_k) {
            const int tap = k0 / p.Cin;
            cbytes = (k0 - tap * p.Cin) * 2;
            if (p.taps == 9) { dy = tap / 3 - 1; dx = tap % 3 - 1; }
            srcb = Ab;
            rowb = p.Cin * 2;
        } else {
            cbytes = (k0 - conv_k) * 2;
            srcb = A2b;
            rowb = p.Cin2 * 2;
        }
        const int shift = dy * p.Wd + dx;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = (pix_yx[q] >> 16) + dy, xx = (pix_yx[q] & 0xffff) + dx;
            const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
            const unsigned off = (unsigned)(pix_m[q] + shift) * (unsigned)rowb + cbytes + achunk;
            const char* src = ok ? srcb + off : Zb + achunk;
            char* dst = base + (q * NW + wave) * 1024;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
        }
        char* dstw = base + (32 + wave) * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(Wb + (size_t)v * 64 + woffs), LDS_PTR(dstw), 16, 0, 0);
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

    auto epilogue = [&](int t) {
        const int mw0 = (t / p.tiles_n) * C_BM + (wm * 2 + wmm) * 128, nw0 = (t % p.tiles_n) * C_BN + wn * 64;
        const int nb = nw0 + 16 * qd;
        const bool nb_ok = nb < p.N;
        float bias[16];
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            f32x4 bv = (p.bias && nb_ok) ? *reinterpret_cast<const f32x4*>(p.bias + nb + e) : f32x4{0.f, 0.f, 0.f, 0.f};
            bias[e] = bv[0]; bias[e + 1] = bv[1]; bias[e + 2] = bv[2]; bias[e + 3] = bv[3];
        }
        if constexpr (BF16_OUT) {
            const int rr = lane >> 3, ch = lane & 7;
            const bool cols_ok = nw0 + 8 * ch < p.N;
#pragma unroll
            for (int rnd = 0; rnd < TM / 2; ++rnd) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * rnd + jj;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r] + bias[4 * i + r];
                    if constexpr (EPI == CEPI_FILM_SILU_BF16) {
                        // y*(scale+1)+shift (FeatureModulation, residual_block.py:21-24: addcmul(shift, scale+1, y)), then SiLU
                        int m = mw0 + 16 * j + rho;
                        m = m < p.M ? m : p.M - 1;
                        const float* fr = p.film + (size_t)((m / HW) % p.film_rows) * p.film_stride;
                        if (nb_ok) {
#pragma unroll
                            for (int e = 0; e < 16; e += 4) {
                                const f32x4 sc = *reinterpret_cast<const f32x4*>(fr + nb + e);
                                const f32x4 sh = *reinterpret_cast<const f32x4*>(fr + p.N + nb + e);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[e + r] = silu_f(__fmaf_rn(sc[r] + 1.0f, v[e + r], sh[r]));
                            }
                        }
                    }
                    u32x4 w0, w1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w0[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                        w1[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                    }
                    const int r = 16 * jj + rho;
                    *reinterpret_cast<u32x4*>(scratch + r * 128 + (((2 * qd) ^ (r & 7)) << 4)) = w0;
                    *reinterpret_cast<u32x4*>(scratch + r * 128 + (((2 * qd + 1) ^ (r & 7)) << 4)) = w1;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                u32x4 d[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int r = 8 * t4 + rr;
                    d[t4] = *reinterpret_cast<const u32x4*>(scratch + r * 128 + ((ch ^ (r & 7)) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int m = mw0 + 32 * rnd + 8 * t4 + rr;
                    if (m < p.M && cols_ok)
                        __builtin_nontemporal_store(d[t4], reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.out) + (size_t)m * p.ldo + nw0 + 8 * ch));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = mw0 + 16 * j + rho;
                if (m >= p.M || !nb_ok) continue;
                float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + nb;
                const float* rs = p.resid ? p.resid + (size_t)m * p.ldo + nb : nullptr;
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[e / 4][j][r] + bias[e + r];
                    if (rs) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(rs + e);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    }
                    *reinterpret_cast<f32x4*>(o + e) = v;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    const int nk = p.K / 32;  // launcher guarantees nk >= D
    set_sources(tile);
#pragma unroll
    for (int d = 0; d < C_D; ++d) stage(d, d);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1)) : "memory");
    PHASE_BARRIER();
    if (wm == 1) PHASE_BARRIER();

    int slot = 0, pslot = C_D;
    int after_e = 0;
    while (true) {
        const int next = tile + wpx;
        const bool has_next = next < hi;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int v = 0; v < nk; ++v) {
            {
                const char* b = lds + slot * C_SLOT;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(b + woff + i * 4 * C_RB);
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(b + xoff + j * 16 * C_RB);
            }
            bool issued = false;
            if (v + C_D < nk) {
                stage(v + C_D, pslot);
                issued = true;
            } else if (has_next) {
                if (v + C_D == nk) set_sources(next);
                stage(v + C_D - nk, pslot);
                issued = true;
            }
            if (issued) {
                if (BF16_OUT && after_e > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1) + 16) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (C_D - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (after_e > 0) --after_e;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PHASE_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (v == nk - 1 && wm == 1) epilogue(tile);
            PHASE_BARRIER();
            slot = (slot == C_R - 1) ? 0 : slot + 1;
            pslot = (pslot == C_R - 1) ? 0 : pslot + 1;
        }
        if (wm == 0) epilogue(tile);
        after_e = 2;
        if (!has_next) break;
        tile = next;
    }
    if (wm == 0) PHASE_BARRIER();
#undef PHASE_BARRIER
}

int g_conv_cus = 0;

template <int EPI>
int launch_conv(ConvParams p, hipStream_t s) {
    p.tiles_m = (p.M + C_BM - 1) / C_BM;
    p.tiles_n = (p.N + C_BN - 1) / C_BN;
    if (g_conv_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_conv_cus = prop.multiProcessorCount;
        if (g_conv_cus <= 0) g_conv_cus = 256;
    }
    const int nwg = p.tiles_m * p.tiles_n;
    const int grid = nwg < g_conv_cus ? nwg : g_conv_cus;
    const size_t lds = (size_t)C_R * C_SLOT + 32768;
    auto kern = conv_ring_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
    BSI_CHECK_LAUNCH("bsi_conv_nhwc_bf16");
    return BSI_OK;
}

}  // namespace

extern "C" int bsi_conv_nhwc_bf16(const bsi_conv_args* a, bsi_stream_t stream) {
    BSI_CHECK_ARG(a && a->x && a->w && a->out && a->zeros, "bsi_conv_nhwc_bf16: null pointer");
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
    p.zeros = reinterpret_cast<const __bf16*>(a->zeros);
    p.out = a->out;
    p.film = a->film; p.film_rows = a->film_rows > 0 ? a->film_rows : 1; p.film_stride = a->film_stride;
    p.resid = a->resid;
    p.M = a->B * a->H * a->W; p.N = a->Cout; p.Cin = a->Cin; p.Cin2 = a->Cin2; p.taps = a->taps;
    p.H = a->H; p.Wd = a->W; p.ldo = a->ldo;
    p.K = a->taps * a->Cin + a->Cin2;
    BSI_CHECK_ARG(p.K / 32 >= C_D, "bsi_conv_nhwc_bf16: K=%d too small", p.K);
    BSI_CHECK_ARG((size_t)p.M * (a->Cin > a->Cin2 ? a->Cin : a->Cin2) * 2 < (1ull << 32) && (size_t)p.N * p.K * 2 < (1ull << 32),
                  "bsi_conv_nhwc_bf16: tensors exceed the 32-bit offset range");
    BSI_CHECK_ARG(a->H < 65536 && a->W < 65536, "bsi_conv_nhwc_bf16: image too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a->epilogue) {
        case BSI_CONV_BIAS_BF16: return launch_conv<CEPI_BIAS_BF16>(p, s);
        case BSI_CONV_FILM_SILU_BF16:
            BSI_CHECK_ARG(a->film, "bsi_conv_nhwc_bf16: FILM epilogue needs the (scale, shift) table");
            return launch_conv<CEPI_FILM_SILU_BF16>(p, s);
        case BSI_CONV_BIAS_RESID_F32: return launch_conv<CEPI_BIAS_RESID_F32>(p, s);
        default: bsi_set_error("bsi_conv_nhwc_bf16: unknown epilogue %d", a->epilogue); return BSI_EINVAL;
    }
}
