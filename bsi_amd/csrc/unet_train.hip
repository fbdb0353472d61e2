// Training engine of the DenoisingVDMUNet: forward that records a tape, and the hand-written backward (replaces torch
// autograd over bsi/models/vdm_unet.py:92-100, bsi/nn/simplified_unet.py:33-48, bsi/nn/residual_block.py:61-64 and
// bsi/nn/attention.py:32-41 of the reference in `BSI.train_loss(...).mean().backward()`, bsi/tasks/bsi.py:187-194).
// Host-side sequencing only; no allocation, no synchronisation.
//
// Tape per residual block and pixel: a = bf16 silu(GN(x)) [Cx], (up blocks: raw = bf16 x [2 dim]), h1 = bf16 conv1 output
// [dim], y = bf16 dropout(silu(film(h1))) [dim], out fp32 [dim]  = 1.25 MB (2 MB for an up block) per 32x32 image at
// dim 128: 109 MB per image for the 67-block UNet of config/experiment/cifar10-vdm.yaml, 14 GB at 128 images per GPU.
#include "common.h"
#include "dit_ops.h"
#include "unet_ops.h"

extern "C" int bsi_silu_bf16(const float* pre, size_t n, void* out, bsi_stream_t stream);

namespace {

inline size_t au(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct UD {
    int nfreq, cin, cin_pad, HW, L, nblocks, dh, dim, F;
    size_t M;
};

inline UD ud(const bsi_unet_config* c, int B) {
    UD d;
    d.nfreq = (c->ff_nmax >= c->ff_nmin) ? (c->ff_nmax - c->ff_nmin + 1) : 0;
    d.cin = c->C + c->C * d.nfreq * 2;
    d.cin_pad = (int)au((size_t)d.cin, 32);
    d.HW = c->H * c->W;
    d.L = c->levels;
    d.nblocks = 2 * c->levels + 2;
    d.dim = c->dim;
    d.dh = c->dim / c->heads;
    d.F = d.nblocks * 2 * c->dim;
    d.M = (size_t)B * d.HW;
    return d;
}

struct BlockTape {
    char *a, *raw, *h1, *y;
    float* out;
};

struct UTape {
    char* zeros;
    char* xin;      // bf16 [M, cin_pad]
    char* emb;      // bf16 [B, 64]
    float* pre1;    // fp32 [B, c_dim]
    char* c1;       // bf16 [B, c_dim]
    float* pre2;    // fp32 [B, c_dim]
    char* c2;       // bf16 [B, c_dim]
    float* film;    // fp32 [B, nblocks, 2 dim]
    float* henc;    // fp32 [M, dim]
    char* agn;      // bf16 [M, dim]   GroupNorm output in front of the attention
    char* qkv;      // bf16 [M, 3 dim]
    char* ay;       // bf16 [M, dim]   attention output
    float* lse;     // fp32 [B, heads, HW]
    float* hatt;    // fp32 [M, dim]
    float* gnstats; // fp32 [nblocks + 1][B][32][2]: (mean, rstd) of every GroupNorm (block i; last = the attention's)
    float* gnpart;  // fp32 [nblocks + 2][M/128][dim/4][2]: GroupNorm partials of every fp32 feature map (henc, hatt, block outputs),
                    // written by the producing convolution's epilogue, read by bsi_groupnorm_apply_nhwc in the same forward pass
    size_t gnpart_stride;  // floats per map
    char* blocks;
    size_t plain_bytes, up_bytes, total;
};

inline UTape carve_tape(const bsi_unet_config* c, int B, void* base) {
    const UD d = ud(c, B);
    UTape t;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    const size_t M = d.M, dim = d.dim, cd = c->c_dim;
    t.zeros = p + off; off += 256;
    t.xin = p + off; off += au(M * d.cin_pad * 2);
    t.emb = p + off; off += au((size_t)B * 64 * 2);
    t.pre1 = reinterpret_cast<float*>(p + off); off += au((size_t)B * cd * 4);
    t.c1 = p + off; off += au((size_t)B * cd * 2);
    t.pre2 = reinterpret_cast<float*>(p + off); off += au((size_t)B * cd * 4);
    t.c2 = p + off; off += au((size_t)B * cd * 2);
    t.film = reinterpret_cast<float*>(p + off); off += au((size_t)B * d.F * 4);
    t.henc = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    t.agn = p + off; off += au(M * dim * 2);
    t.qkv = p + off; off += au(M * 3 * dim * 2);
    t.ay = p + off; off += au(M * dim * 2);
    t.lse = reinterpret_cast<float*>(p + off); off += au((size_t)B * c->heads * d.HW * 4);
    t.hatt = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    t.gnstats = reinterpret_cast<float*>(p + off); off += au((size_t)(d.nblocks + 1) * B * 64 * 4);
    t.gnpart_stride = au((M + 127) / 128 * (dim / 4) * 2 * 4) / 4;
    t.gnpart = reinterpret_cast<float*>(p + off); off += t.gnpart_stride * 4 * (d.nblocks + 2);
    t.blocks = p + off;
    t.plain_bytes = au(M * dim * 2) * 3 + au(M * dim * 4);
    t.up_bytes = au(M * 2 * dim * 2) * 2 + au(M * dim * 2) * 2 + au(M * dim * 4);
    off += t.plain_bytes * (d.L + 2) + t.up_bytes * d.L;
    t.total = off;
    return t;
}

inline BlockTape block_tape(const UTape& t, const UD& d, int blk) {
    BlockTape b;
    const size_t M = d.M, dim = d.dim;
    const bool up = blk >= d.L + 2;
    char* p = t.blocks + (up ? t.plain_bytes * (d.L + 2) + t.up_bytes * (blk - d.L - 2) : t.plain_bytes * blk);
    size_t off = 0;
    if (up) {
        b.a = p + off; off += au(M * 2 * dim * 2);
        b.raw = p + off; off += au(M * 2 * dim * 2);
    } else {
        b.a = p + off; off += au(M * dim * 2);
        b.raw = nullptr;
    }
    b.h1 = p + off; off += au(M * dim * 2);
    b.y = p + off; off += au(M * dim * 2);
    b.out = reinterpret_cast<float*>(p + off);
    return b;
}

struct UBwd {
    float* dcur[2];  // fp32 [M, dim]
    float* dskips;   // fp32 [L][M, dim]
    float* dcat;     // fp32 [M, 2 dim]
    char* g;         // bf16 [M, dim]
    char* dy;        // bf16 [M, dim]
    char* dh1;       // bf16 [M, dim]
    char* da;        // bf16 [M, 2 dim]
    char* dqkv;      // bf16 [M, 3 dim]
    float* dfilm;    // fp32 [HW / 64 planes][B, F]: per-slab sums of the FiLM gradients (reproducible: no atomics)
    float* gnpart;   // fp32 [B][2][2 dim]: per-image sums of a GroupNorm's affine gradients
    float* decparts; // per-block sums of the decode convolution's parameter gradients
    char* dfilm_bf;  // bf16 [B, F]
    float* dc;       // fp32 [B, c_dim]
    char* dpre_bf;   // bf16 [B, c_dim]
    char* wg;        // conv wgrad slabs
    char* tn;        // TN GEMM slabs
    char* cs;        // colsum slabs
    size_t skip_stride, total;
};

inline UBwd carve_bwd(const bsi_unet_config* c, int B, void* base) {
    const UD d = ud(c, B);
    UBwd w;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    const size_t M = d.M, dim = d.dim, cd = c->c_dim;
    for (int i = 0; i < 2; ++i) { w.dcur[i] = reinterpret_cast<float*>(p + off); off += au(M * dim * 4); }
    w.skip_stride = au(M * dim * 4);
    w.dskips = reinterpret_cast<float*>(p + off); off += w.skip_stride * d.L;
    w.dcat = reinterpret_cast<float*>(p + off); off += au(M * 2 * dim * 4);
    w.g = p + off; off += au(M * dim * 2);
    w.dy = p + off; off += au(M * dim * 2);
    w.dh1 = p + off; off += au(M * dim * 2);
    w.da = p + off; off += au(M * 2 * dim * 2);
    w.dqkv = p + off; off += au(M * 3 * dim * 2);
    w.dfilm = reinterpret_cast<float*>(p + off); off += au((size_t)(d.HW / 64 > 0 ? d.HW / 64 : 1) * B * d.F * 4);
    w.gnpart = reinterpret_cast<float*>(p + off); off += au((size_t)B * 2 * 2 * dim * 4);
    w.decparts = reinterpret_cast<float*>(p + off); off += au(bsi_unet_decode_bwd_parts_floats((int)M, (int)dim, c->C) * 4);
    w.dfilm_bf = p + off; off += au((size_t)B * d.F * 2);
    w.dc = reinterpret_cast<float*>(p + off); off += au((size_t)B * cd * 4);
    w.dpre_bf = p + off; off += au((size_t)B * cd * 2);
    size_t wg = bsi_conv_wgrad_workspace_bytes((int)M, 2 * (int)dim, 0, (int)dim, 9);
    size_t t2 = bsi_conv_wgrad_workspace_bytes((int)M, (int)dim, 2 * (int)dim, (int)dim, 9);
    if (t2 > wg) wg = t2;
    t2 = bsi_conv_wgrad_workspace_bytes((int)M, (int)dim, 0, 3 * (int)dim, 9);
    if (t2 > wg) wg = t2;
    t2 = bsi_conv_wgrad_workspace_bytes((int)M, d.cin_pad, 0, (int)dim, 9);
    if (t2 > wg) wg = t2;
    w.wg = p + off; off += au(wg);
    size_t tn = bsi_gemm_tn_workspace_bytes(B, d.F, (int)cd);
    t2 = bsi_gemm_tn_workspace_bytes(B, (int)cd, (int)cd);
    if (t2 > tn) tn = t2;
    w.tn = p + off; off += au(tn);
    w.cs = p + off; off += au(bsi_colsum_workspace_bytes(d.F > 3 * (int)dim ? d.F : 3 * (int)dim));
    w.total = off;
    return w;
}

#define TRY(expr)              \
    do {                       \
        int rc__ = (expr);     \
        if (rc__) return rc__; \
    } while (0)

int conv(const void* x, const void* x2, const void* w, const float* bias, const void* zeros, void* out, const float* resid, int B,
         int H, int W, int Cin, int Cin2, int Cout, int taps, int epi, bsi_stream_t s, float* gn_part = nullptr) {
    bsi_conv_args a{};
    a.gn_partial = gn_part;
    a.x = x; a.x2 = x2; a.w = w; a.bias = bias; a.zeros = zeros; a.out = out; a.resid = resid; a.B = B; a.H = H; a.W = W;
    a.Cin = Cin; a.Cin2 = Cin2; a.Cout = Cout; a.taps = taps; a.ldo = Cout; a.epilogue = epi;
    return bsi_conv_nhwc_bf16(&a, s);
}

int gemm(const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldo, int M, int N, int K, int epi,
         bsi_stream_t stream) {
    bsi_gemm_args g{};
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.epilogue = epi;
    return bsi_gemm_bf16(&g, stream);
}

int check_geometry(const bsi_unet_config* cfg, const UD& d, const char* who) {
    BSI_CHECK_ARG((cfg->dim == 64 || cfg->dim == 128) && cfg->dim % cfg->heads == 0 && (d.dh == 64 || d.dh == 128) && d.HW % 64 == 0 &&
                      cfg->emb_size <= 64 && cfg->emb_size % 2 == 0 && cfg->c_dim % 64 == 0,
                  "%s: unsupported geometry dim=%d heads=%d HW=%d emb=%d c_dim=%d", who, cfg->dim, cfg->heads, d.HW, cfg->emb_size,
                  cfg->c_dim);
    return BSI_OK;
}

}  // namespace

extern "C" size_t bsi_unet_tape_bytes(const bsi_unet_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve_tape(cfg, B, nullptr).total;
}

extern "C" size_t bsi_unet_backward_workspace_bytes(const bsi_unet_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve_bwd(cfg, B, nullptr).total;
}

extern "C" int bsi_unet_train_forward(const bsi_unet_config* cfg, const bsi_unet_weights* w, int B, const float* mu, const float* t,
                                      const float* c_in, const float* c_skip, const float* c_out, float* out, void* tape_mem,
                                      float dropout_p, unsigned long long seed, bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && w->blocks && mu && t && out && tape_mem && B > 0, "bsi_unet_train_forward: bad args");
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_unet_train_forward: dropout probability %g outside [0, 1)", (double)dropout_p);
    BSI_CHECK_ARG((c_in == nullptr) == (c_skip == nullptr) && (c_in == nullptr) == (c_out == nullptr),
                  "bsi_unet_train_forward: c_in/c_skip/c_out must be given together");
    const UD d = ud(cfg, B);
    TRY(check_geometry(cfg, d, "bsi_unet_train_forward"));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int dim = d.dim, H = cfg->H, W = cfg->W, L = d.L, M = (int)d.M, cd = cfg->c_dim;
    UTape tp = carve_tape(cfg, B, tape_mem);
    if (hipMemsetAsync(tp.zeros, 0, 256, s) != hipSuccess || hipMemsetAsync(tp.emb, 0, (size_t)B * 64 * 2, s) != hipSuccess) {
        bsi_set_error("bsi_unet_train_forward: memset failed");
        return BSI_ELAUNCH;
    }
    // c = pos_map(t) with its pre-activations kept (vdm_unet.py:62-69), FiLM table of every block (residual_block.py:39,62)
    {
        float* embf = tp.pre1;  // fp32 [B, emb_size] staging (overwritten by pre1 right after)
        TRY(bsi_nyquist_embed(t, B, w->pe_scale, w->pe_bias, cfg->emb_size, embf, nullptr, stream));
        TRY(bsi_cast_rows_bf16(embf, cfg->emb_size, B, cfg->emb_size, tp.emb, 64, stream));
        TRY(gemm(tp.emb, 64, w->pm1_w, 64, w->pm1_b, tp.pre1, cd, B, cd, 64, BSI_EPI_BIAS_F32, stream));
        TRY(bsi_silu_bf16(tp.pre1, (size_t)B * cd, tp.c1, stream));
        TRY(gemm(tp.c1, cd, w->pm3_w, cd, w->pm3_b, tp.pre2, cd, B, cd, cd, BSI_EPI_BIAS_F32, stream));
        TRY(bsi_silu_bf16(tp.pre2, (size_t)B * cd, tp.c2, stream));
        TRY(gemm(tp.c2, cd, w->film_w, cd, w->film_b, tp.film, d.F, B, d.F, cd, BSI_EPI_BIAS_F32, stream));
    }
    TRY(bsi_dit_prologue_launch(mu, c_in, 1, B, cfg->C, H, W, 1, cfg->ff_nmin, d.nfreq, d.cin_pad, tp.xin, s));
    // GroupNorm statistics from the producing convolution's epilogue + one streaming normalisation pass (unet_engine.hip);
    // partial slot of a feature map: 0 = encoder output, 1 = attention output, 2 + blk = block blk's output; x1p / x2p = slots of
    // a GroupNorm's inputs.  BSI_UNET_NO_GN_FUSE=1 keeps the reduce-then-normalise kernel.
    static const bool no_gn_fuse = getenv("BSI_UNET_NO_GN_FUSE") != nullptr;
    const bool gn_fuse = !no_gn_fuse && dim == 128 && d.HW % 128 == 0 && d.HW <= 1024;
    auto part = [&](int slot) -> float* { return gn_fuse ? tp.gnpart + (size_t)slot * tp.gnpart_stride : nullptr; };
    auto groupnorm = [&](const float* x1, int x1p, const float* x2, int x2p, const float* gw, const float* gb, int silu, void* a, void* raw,
                         float* stats) -> int {
        const int cin2 = x2 ? dim : 0;
        if (gn_fuse)
            return bsi_groupnorm_apply_nhwc(x1, dim, part(x1p), x2, cin2, x2 ? part(x2p) : nullptr, B, d.HW, gw, gb, 1e-5f, silu, a, raw, stats,
                                            stream);
        if (d.HW <= 1024)  // the register-resident kernel also saves (mean, rstd) for the backward pass
            return bsi_groupnorm_stats_nhwc(x1, dim, x2, cin2, B, d.HW, gw, gb, 1e-5f, silu, a, raw, stats, stream);
        return bsi_groupnorm_nhwc(x1, dim, x2, cin2, B, d.HW, gw, gb, 1e-5f, silu, a, raw, stream);
    };
    TRY(conv(tp.xin, nullptr, w->enc_w, w->enc_b, tp.zeros, tp.henc, nullptr, B, H, W, d.cin_pad, 0, dim, 9, BSI_CONV_BIAS_RESID_F32,
             stream, part(0)));
    auto resblock = [&](int blk, const float* x1, int x1p, const float* x2, int x2p) -> int {
        const bsi_unet_resblock_weights& rb = w->blocks[blk];
        BlockTape bt = block_tape(tp, d, blk);
        const int cin2 = x2 ? dim : 0;
        TRY(groupnorm(x1, x1p, x2, x2p, rb.gn_w, rb.gn_b, 1, bt.a, x2 ? bt.raw : nullptr, tp.gnstats + (size_t)blk * B * 64));
        TRY(conv(bt.a, nullptr, rb.conv1_w, rb.conv1_b, tp.zeros, bt.h1, nullptr, B, H, W, dim + cin2, 0, dim, 9, BSI_CONV_BIAS_BF16,
                 stream));
        TRY(bsi_film_silu_drop(bt.h1, M, dim, d.HW, tp.film + (size_t)blk * 2 * dim, B, d.F, make_drop(dropout_p, seed, blk), bt.y,
                               stream));
        return conv(bt.y, x2 ? bt.raw : nullptr, rb.conv2_w, rb.conv2_b, tp.zeros, bt.out, x2 ? nullptr : x1, B, H, W, dim,
                    x2 ? 2 * dim : 0, dim, 9, BSI_CONV_BIAS_RESID_F32, stream, part(2 + blk));
    };
    const float* h = tp.henc;
    int hp = 0;
    for (int i = 0; i < L; ++i) {
        TRY(resblock(i, h, hp, nullptr, 0));
        h = block_tape(tp, d, i).out;
        hp = 2 + i;
    }
    TRY(resblock(L, h, hp, nullptr, 0));
    h = block_tape(tp, d, L).out;
    hp = 2 + L;
    TRY(groupnorm(h, hp, nullptr, 0, w->agn_w, w->agn_b, 0, tp.agn, nullptr, tp.gnstats + (size_t)d.nblocks * B * 64));
    TRY(conv(tp.agn, nullptr, w->aqkv_w, w->aqkv_b, tp.zeros, tp.qkv, nullptr, B, H, W, dim, 0, 3 * dim, 9, BSI_CONV_BIAS_BF16, stream));
    TRY(bsi_attention_fwd_lse(tp.qkv, 3 * dim, B, d.HW, cfg->heads, d.dh, tp.ay, dim, tp.lse, stream));
    TRY(conv(tp.ay, nullptr, w->aout_w, w->aout_b, tp.zeros, tp.hatt, h, B, H, W, dim, 0, dim, 9, BSI_CONV_BIAS_RESID_F32, stream, part(1)));
    TRY(resblock(L + 1, tp.hatt, 1, nullptr, 0));
    h = block_tape(tp, d, L + 1).out;
    hp = 2 + L + 1;
    for (int i = 0; i < L; ++i) {
        TRY(resblock(L + 2 + i, h, hp, block_tape(tp, d, L - 1 - i).out, 2 + (L - 1 - i)));
        h = block_tape(tp, d, L + 2 + i).out;
        hp = 2 + L + 2 + i;
    }
    return bsi_unet_decode(h, B, d.HW, dim, w->dec_w, w->dec_b, cfg->C, mu, c_skip, c_out, 1, out, stream);
}

extern "C" int bsi_unet_backward(const bsi_unet_config* cfg, const bsi_unet_weights* w, const bsi_unet_weights_t* wT,
                                 const bsi_unet_grads* g, int B, const float* g_out, const float* c_out, void* tape_mem,
                                 void* workspace, float dropout_p, unsigned long long seed, bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && w->blocks && wT && wT->blocks && g && g->blocks && g_out && tape_mem && workspace && B > 0,
                  "bsi_unet_backward: bad args");
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_unet_backward: dropout probability %g outside [0, 1)", (double)dropout_p);
    const UD d = ud(cfg, B);
    TRY(check_geometry(cfg, d, "bsi_unet_backward"));
    const int dim = d.dim, H = cfg->H, W = cfg->W, L = d.L, M = (int)d.M, cd = cfg->c_dim;
    UTape tp = carve_tape(cfg, B, tape_mem);
    UBwd ws = carve_bwd(cfg, B, workspace);
    auto dskip = [&](int i) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws.dskips) + (size_t)i * ws.skip_stride); };

    // No atomics anywhere in this backward (bit reproducible): FiLM gradients go to per-slab planes that are summed in fixed order
    // at the end, GroupNorm affine and decode gradients to per-image / per-block rows summed by reduce_slabs -- nothing to zero.
    BSI_CHECK_ARG(d.HW % 64 == 0, "bsi_unet_backward: H*W=%d must be a multiple of 64", d.HW);
    const int fplanes = d.HW / 64;
    const size_t fplane = (size_t)B * d.F;

    // residual block backward: dOut fp32 [M, dim] -> out1 (gradient of x1, + add_b) and out2 (gradient of the skip tensor x2)
    // `g_ready`: ws.g already holds the bf16 copy of dOut (written by the GroupNorm backward that produced dOut)
    bool g_ready = false;
    auto resblock_bwd = [&](int blk, const float* dOut, const float* x1, const float* x2, const float* add_b, float* out1,
                            float* out2) -> int {
        const bsi_unet_resblock_weights& rb = w->blocks[blk];
        const bsi_unet_resblock_weights_t& rT = wT->blocks[blk];
        const bsi_unet_resblock_grads& rg = g->blocks[blk];
        BlockTape bt = block_tape(tp, d, blk);
        const int cin2 = x2 ? dim : 0, cx = dim + cin2;
        if (!g_ready) TRY(bsi_silu_bwd_bf16(dOut, nullptr, (size_t)M * dim, ws.g, stream));  // bf16 copy of dOut
        TRY(bsi_conv_wgrad_conv2d_nhwc_bf16(ws.g, dim, bt.y, x2 ? bt.raw : nullptr, tp.zeros, B, H, W, dim, dim, x2 ? 2 * dim : 0, dim, 9,
                                            rg.conv2_w, x2 ? rg.skip_w : nullptr, rg.conv2_b, ws.wg, stream));
        TRY(conv(ws.g, nullptr, rT.conv2_wT, nullptr, tp.zeros, ws.dy, nullptr, B, H, W, dim, 0, dim, 9, BSI_CONV_BIAS_BF16, stream));
        TRY(bsi_film_silu_bwd_drop(ws.dy, bt.h1, M, dim, d.HW, tp.film + (size_t)blk * 2 * dim, B, d.F, make_drop(dropout_p, seed, blk),
                                   ws.dh1, ws.dfilm + (size_t)blk * 2 * dim, d.F, stream, fplane));
        TRY(bsi_conv_wgrad_conv2d_nhwc_bf16(ws.dh1, dim, bt.a, nullptr, tp.zeros, B, H, W, cx, cx, 0, dim, 9, rg.conv1_w, nullptr, rg.conv1_b,
                                            ws.wg, stream));
        TRY(conv(ws.dh1, nullptr, rT.conv1_wT, nullptr, tp.zeros, ws.da, nullptr, B, H, W, dim, 0, cx, 9, BSI_CONV_BIAS_BF16, stream));
        const float* add = dOut;
        if (x2) {  // 1x1 skip conv on cat(x, x_skip) (residual_block.py:40,63): d cat = g . Wskip
            TRY(gemm(ws.g, dim, rT.skip_wT, dim, nullptr, ws.dcat, 2 * dim, M, 2 * dim, dim, BSI_EPI_BIAS_F32, stream));
            add = ws.dcat;
        }
        (void)rb;
        // out1 is the next block's dOut: its bf16 copy goes straight into ws.g (no longer read by this block)
        g_ready = true;
        return bsi_groupnorm_bwd_cast_det(ws.da, x1, dim, x2, cin2, B, d.HW, rb.gn_w, rb.gn_b, 1e-5f, 1, add, add_b, out1, out2, rg.gn_w,
                                          rg.gn_b, ws.g, d.HW <= 1024 ? tp.gnstats + (size_t)blk * B * 64 : nullptr, ws.gnpart, stream);
    };

    int cur = 0;
    const float* hlast = block_tape(tp, d, d.nblocks - 1).out;
    TRY(bsi_unet_decode_bwd_det(g_out, c_out, 1, hlast, B, d.HW, dim, w->dec_w, cfg->C, ws.dcur[cur], g->dec_w, g->dec_b, ws.decparts, stream));
    for (int i = L - 1; i >= 0; --i) {  // up blocks
        const int blk = L + 2 + i;
        const float* x1 = i == 0 ? block_tape(tp, d, L + 1).out : block_tape(tp, d, blk - 1).out;
        const float* x2 = block_tape(tp, d, L - 1 - i).out;
        TRY(resblock_bwd(blk, ws.dcur[cur], x1, x2, nullptr, ws.dcur[cur ^ 1], dskip(L - 1 - i)));
        cur ^= 1;
    }
    TRY(resblock_bwd(L + 1, ws.dcur[cur], tp.hatt, nullptr, nullptr, ws.dcur[cur ^ 1], nullptr));
    cur ^= 1;
    {   // Residual(GroupNorm -> Attention2D) (vdm_unet.py:83-87)
        const float* dOut = ws.dcur[cur];
        const float* hin = block_tape(tp, d, L).out;
        if (!g_ready) TRY(bsi_silu_bwd_bf16(dOut, nullptr, (size_t)M * dim, ws.g, stream));
        TRY(bsi_conv_wgrad_conv2d_nhwc_bf16(ws.g, dim, tp.ay, nullptr, tp.zeros, B, H, W, dim, dim, 0, dim, 9, g->aout_w, nullptr, g->aout_b, ws.wg,
                                            stream));
        TRY(conv(ws.g, nullptr, wT->aout_wT, nullptr, tp.zeros, ws.dy, nullptr, B, H, W, dim, 0, dim, 9, BSI_CONV_BIAS_BF16, stream));
        TRY(bsi_attention_bwd_long(tp.qkv, 3 * dim, tp.ay, ws.dy, dim, tp.lse, B, d.HW, cfg->heads, d.dh, ws.dqkv, 3 * dim, stream));
        TRY(bsi_conv_wgrad_conv2d_nhwc_bf16(ws.dqkv, 3 * dim, tp.agn, nullptr, tp.zeros, B, H, W, dim, dim, 0, 3 * dim, 9, g->aqkv_w, nullptr,
                                            g->aqkv_b, ws.wg, stream));
        TRY(conv(ws.dqkv, nullptr, wT->aqkv_wT, nullptr, tp.zeros, ws.da, nullptr, B, H, W, 3 * dim, 0, dim, 9, BSI_CONV_BIAS_BF16, stream));
        TRY(bsi_groupnorm_bwd_cast_det(ws.da, hin, dim, nullptr, 0, B, d.HW, w->agn_w, w->agn_b, 1e-5f, 0, dOut, nullptr, ws.dcur[cur ^ 1],
                                       nullptr, g->agn_w, g->agn_b, ws.g, d.HW <= 1024 ? tp.gnstats + (size_t)d.nblocks * B * 64 : nullptr,
                                       ws.gnpart, stream));
        g_ready = true;
        cur ^= 1;
    }
    // centre block 0 and the down path: the output of down block i is also the skip tensor of up block L-1-i
    TRY(resblock_bwd(L, ws.dcur[cur], L > 0 ? block_tape(tp, d, L - 1).out : tp.henc, nullptr, L > 0 ? dskip(L - 1) : nullptr,
                     ws.dcur[cur ^ 1], nullptr));
    cur ^= 1;
    for (int i = L - 1; i >= 0; --i) {
        TRY(resblock_bwd(i, ws.dcur[cur], i > 0 ? block_tape(tp, d, i - 1).out : tp.henc, nullptr, i > 0 ? dskip(i - 1) : nullptr,
                         ws.dcur[cur ^ 1], nullptr));
        cur ^= 1;
    }
    // encode convolution (vdm_unet.py:71,99)
    if (!g_ready) TRY(bsi_silu_bwd_bf16(ws.dcur[cur], nullptr, (size_t)M * dim, ws.g, stream));
    TRY(bsi_conv_wgrad_conv2d_nhwc_bf16(ws.g, dim, tp.xin, nullptr, tp.zeros, B, H, W, d.cin_pad, d.cin, 0, dim, 9, g->enc_w, nullptr, g->enc_b,
                                        ws.wg, stream));

    // FiLM projections and pos_map (rows = samples)
    TRY(bsi_sum_cast_rows_bf16(ws.dfilm, fplanes, fplane, d.F, B, d.F, ws.dfilm_bf, d.F, stream));
    TRY(bsi_gemm_tn_bias_bf16(ws.dfilm_bf, d.F, tp.c2, cd, B, d.F, cd, g->film_w, cd, g->film_b, 0, ws.tn, stream));
    TRY(gemm(ws.dfilm_bf, d.F, wT->film_wT, d.F, nullptr, ws.dc, cd, B, cd, d.F, BSI_EPI_BIAS_F32, stream));
    TRY(bsi_silu_bwd_bf16(ws.dc, tp.pre2, (size_t)B * cd, ws.dpre_bf, stream));
    TRY(bsi_gemm_tn_bias_bf16(ws.dpre_bf, cd, tp.c1, cd, B, cd, cd, g->pm3_w, cd, g->pm3_b, 0, ws.tn, stream));
    TRY(gemm(ws.dpre_bf, cd, wT->pm3_wT, cd, nullptr, ws.dc, cd, B, cd, cd, BSI_EPI_BIAS_F32, stream));
    TRY(bsi_silu_bwd_bf16(ws.dc, tp.pre1, (size_t)B * cd, ws.dpre_bf, stream));
    TRY(bsi_gemm_tn_bf16(ws.dpre_bf, cd, tp.emb, 64, B, cd, 64, g->pm1_w_padded, 64, 0, ws.tn, stream));
    return bsi_colsum_bf16(ws.dpre_bf, cd, B, cd, g->pm1_b, 0, ws.cs, stream);
}
