// bf16 GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T, fp32 accumulation on v_mfma_f32_16x16x32_bf16,
// fused epilogues (bias / GELU / SiLU / gated residual / positional embedding).
//
// Replaces every nn.Linear on the DiT path of the reference (bsi/models/dit.py:33-34,71-81,154,163-165,
// bsi/nn/mlp.py:34-38) together with the elementwise ops torch runs after it.
//
// Design (MI355X-first):
//   * workgroup tile BM x BN x 64 with 64-lane waves arranged WM x WN; every wave owns a
//     (16*TM) x 64 output sub-tile = TM x 4 MFMA tiles, accumulators stay in registers for the whole K loop;
//   * both operands are K-contiguous, so a tile row is 128 B = 8 chunks of 16 B; tiles are staged
//     HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered;
//   * LDS image is lane-linear per wave-instruction (hardware rule); bank conflicts are removed by
//     XOR-ing the chunk index on the *global source* side and again on the ds_read side:
//         activation rows:  chunk' = chunk ^ ((row >> 1) & 7)
//         weight rows:      chunk' = chunk ^ (((row >> 1) & 1) | (((row >> 4) & 3) << 1))
//     both of which reduce to `^ ((lane >> 1) & 7)` for the fragment reads below, making every
//     ds_read_b128 conflict free;
//   * the MFMA is issued as D = Wfrag x Xfrag, i.e. D rows are output columns n and D columns are output
//     rows m.  MFMA row rho of n-tile i is mapped to n = 16*(rho>>2) + 4*i + (rho&3), so after the K loop
//     a lane owns 16 *contiguous* n for each of its rows: the epilogue loads/stores 32-64 B per lane and
//     full 128-B lines per row per wave (no LDS transpose, no scattered 2-byte stores);
//   * workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles with n
//     fastest: the A panel of an m-tile is fetched from HBM once and re-used from that XCD's L2.
#include "common.h"

namespace {

constexpr int BK = 64;          // K elements per tile
constexpr int ROW_BYTES = 128;  // BK * sizeof(bf16)

struct GemmParams {
    const __bf16* A;
    const __bf16* W;
    const float* bias;
    void* out;
    int M, N, K;
    int lda, ldw, ldo;
    const float* gate;
    int gate_rows, gate_stride;
    int tokens;
    const float* pos;
    int tiles_m, tiles_n;
};

template <int EPI>
struct EpiTraits {
    static constexpr bool out_bf16 = (EPI == BSI_EPI_BIAS_BF16 || EPI == BSI_EPI_BIAS_GELU_BF16 ||
                                      EPI == BSI_EPI_BIAS_SILU_BF16);
};

// XCD-aware bijective remap of the linear workgroup id (guide §5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int TM, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int BM = WM * TM * 16;
    constexpr int BN = WN * 64;
    constexpr int ROWS = BM + BN;             // LDS rows per buffer (activation rows first, then weight rows)
    constexpr int BUF_BYTES = ROWS * ROW_BYTES;
    constexpr int STAGE_INSTR = ROWS * 8 / NT;  // global_load_lds per thread per K tile
    static_assert(ROWS * 8 % NT == 0, "tile does not divide over the workgroup");
    static_assert(BM % 8 == 0 && BN % 8 == 0, "8 rows per wave-instruction");

    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave-instruction q of this wave fills LDS rows [8*(q*NW+wave), +8) --------
    const int srow = lane >> 3;   // row within the 8-row group
    const int schunk = lane & 7;  // LDS chunk position within the row
    const char* gsrc[STAGE_INSTR];
#pragma unroll
    for (int q = 0; q < STAGE_INSTR; ++q) {
        const int r = (q * NW + wave) * 8 + srow;  // LDS row of this lane
        if (r < BM) {
            const int c = schunk ^ ((r >> 1) & 7);
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            gsrc[q] = reinterpret_cast<const char*>(p.A + (size_t)m * p.lda) + c * 16;
        } else {
            const int rw = r - BM;
            const int c = schunk ^ (((rw >> 1) & 1) | (((rw >> 4) & 3) << 1));
            int n = n0 + rw;
            n = n < p.N ? n : p.N - 1;
            gsrc[q] = reinterpret_cast<const char*>(p.W + (size_t)n * p.ldw) + c * 16;
        }
    }

    auto stage = [&](int kt, int buf) {
        char* base = lds + buf * BUF_BYTES;
#pragma unroll
        for (int q = 0; q < STAGE_INSTR; ++q) {
            char* dst = base + (q * NW + wave) * 8 * ROW_BYTES;  // wave-uniform; hardware adds lane*16
            __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc[q] + (size_t)kt * ROW_BYTES), LDS_PTR(dst), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes, relative to the buffer base) --------------------------------
    const int fx = (lane >> 1) & 7;
    // activation fragment of m-tile jm, k-step ks: row wm*TM*16 + 16*jm + (lane&15), chunk (4ks + lane>>4) ^ fx
    const int xoff = (wm * TM * 16 + (lane & 15)) * ROW_BYTES;
    // weight fragment of n-tile i: row BM + wn*64 + 16*((lane&15)>>2) + 4i + (lane&3)
    const int woff = (BM + wn * 64 + 16 * ((lane & 15) >> 2) + (lane & 3)) * ROW_BYTES;
    const int c0 = (((lane >> 4)) ^ fx) << 4;
    const int c1 = (((lane >> 4) + 4) ^ fx) << 4;

    f32x4 acc[4][TM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* b = lds + buf * BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cc = ks ? c1 : c0;
            bf16x8 wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wf[i] = *reinterpret_cast<const bf16x8*>(b + woff + i * 4 * ROW_BYTES + cc);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(b + xoff + j * 16 * ROW_BYTES + cc);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf, acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane owns rows m = m0 + wm*TM*16 + 16j + (lane&15), columns nb .. nb+15 ------------
    const int nb = n0 + wn * 64 + 16 * (lane >> 4);
    if (nb >= p.N) return;
    float bias[16];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        bias[e] = bv[0]; bias[e + 1] = bv[1]; bias[e + 2] = bv[2]; bias[e + 3] = bv[3];
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * TM * 16 + 16 * j + (lane & 15);
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r] + bias[4 * i + r];

        if constexpr (EPI == BSI_EPI_BIAS_GELU_BF16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = gelu_tanh_f(v[e]);
        } else if constexpr (EPI == BSI_EPI_BIAS_SILU_BF16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = silu_f(v[e]);
        }

        if constexpr (EpiTraits<EPI>::out_bf16) {
            __bf16* o = reinterpret_cast<__bf16*>(p.out) + (size_t)m * p.ldo + nb;
            u32x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w0[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                w1[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
            }
            *reinterpret_cast<u32x4*>(o) = w0;
            *reinterpret_cast<u32x4*>(o + 8) = w1;
        } else {
            float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + nb;
            if constexpr (EPI == BSI_EPI_GATE_RESID) {
                const int row = (m / p.tokens) % p.gate_rows;
                const float* g = p.gate + (size_t)row * p.gate_stride + nb;
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e);
                    f32x4 xv = *reinterpret_cast<const f32x4*>(o + e);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] = xv[r] + gv[r] * v[e + r];
                    *reinterpret_cast<f32x4*>(o + e) = xv;
                }
            } else {
                if constexpr (EPI == BSI_EPI_BIAS_POS_F32) {
                    const float* ps = p.pos + (size_t)(m % p.tokens) * p.N + nb;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(ps + e);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[e + r] += pv[r];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; e += 4)
                    *reinterpret_cast<f32x4*>(o + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
            }
        }
    }
}

template <int TM, int WM, int WN, int EPI>
int launch_cfg(const GemmParams& p0, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * 64;
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * ROW_BYTES;
    auto kern = gemm_bf16_kernel<TM, WM, WN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), lds, s, p);
    BSI_CHECK_LAUNCH("bsi_gemm_bf16");
    return BSI_OK;
}

template <int EPI>
int launch_epi(const GemmParams& p, hipStream_t s) {
    // Large problems: 256x256 tile, 8 waves (2x4), wave tile 128x64.
    // Small M (adaLN tables, tiny test models): 64x256 tile, 4 waves (1x4), wave tile 64x64... keeps N coverage.
    if (p.M > 128) return launch_cfg<8, 2, 4, EPI>(p, s);
    return launch_cfg<4, 2, 4, EPI>(p, s);
}

}  // namespace

extern "C" int bsi_gemm_bf16(const bsi_gemm_args* a, bsi_stream_t stream) {
    BSI_CHECK_ARG(a != nullptr, "bsi_gemm_bf16: null args");
    BSI_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "bsi_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    BSI_CHECK_ARG(a->K % BK == 0, "bsi_gemm_bf16: K=%d must be a multiple of %d", a->K, BK);
    BSI_CHECK_ARG(a->N % 16 == 0, "bsi_gemm_bf16: N=%d must be a multiple of 16", a->N);
    BSI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K,
                  "bsi_gemm_bf16: bad leading dimensions lda=%d ldw=%d", a->lda, a->ldw);
    BSI_CHECK_ARG(a->ldo % 8 == 0 && a->ldo >= a->N, "bsi_gemm_bf16: bad ldo=%d", a->ldo);
    BSI_CHECK_ARG(a->A && a->W && a->out, "bsi_gemm_bf16: null operand");
    GemmParams p{};
    p.A = reinterpret_cast<const __bf16*>(a->A);
    p.W = reinterpret_cast<const __bf16*>(a->W);
    p.bias = a->bias;
    p.out = a->out;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo;
    p.gate = a->gate; p.gate_rows = a->gate_rows; p.gate_stride = a->gate_stride;
    p.tokens = a->tokens > 0 ? a->tokens : 1;
    p.pos = a->pos;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (a->epilogue) {
        case BSI_EPI_BIAS_F32: return launch_epi<BSI_EPI_BIAS_F32>(p, s);
        case BSI_EPI_BIAS_BF16: return launch_epi<BSI_EPI_BIAS_BF16>(p, s);
        case BSI_EPI_BIAS_GELU_BF16: return launch_epi<BSI_EPI_BIAS_GELU_BF16>(p, s);
        case BSI_EPI_BIAS_SILU_BF16: return launch_epi<BSI_EPI_BIAS_SILU_BF16>(p, s);
        case BSI_EPI_GATE_RESID:
            BSI_CHECK_ARG(a->gate && a->gate_rows > 0, "bsi_gemm_bf16: GATE_RESID needs gate");
            return launch_epi<BSI_EPI_GATE_RESID>(p, s);
        case BSI_EPI_BIAS_POS_F32:
            BSI_CHECK_ARG(a->pos && a->tokens > 0, "bsi_gemm_bf16: BIAS_POS needs pos/tokens");
            return launch_epi<BSI_EPI_BIAS_POS_F32>(p, s);
        default:
            bsi_set_error("bsi_gemm_bf16: unknown epilogue %d", a->epilogue);
            return BSI_EINVAL;
    }
}
