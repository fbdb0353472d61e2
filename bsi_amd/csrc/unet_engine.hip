// Host-side sequencing of one DenoisingVDMUNet evaluation (bsi/models/vdm_unet.py:92-100,
// bsi/nn/simplified_unet.py:33-48, bsi/nn/residual_block.py:61-64 of the reference) on a HIP stream.
// No allocation, no synchronisation: the caller owns the workspace.  Activations are NHWC.
#include <stdlib.h>

#include "common.h"
#include "dit_ops.h"
#include "unet_ops.h"

namespace {

inline size_t au(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct UDims {
    int nfreq, cin, cin_pad, HW, nblocks, dh;
    size_t M;
};

inline UDims udims(const bsi_unet_config* c, int B) {
    UDims d;
    d.nfreq = (c->ff_nmax >= c->ff_nmin) ? (c->ff_nmax - c->ff_nmin + 1) : 0;
    d.cin = c->C + c->C * d.nfreq * 2;
    d.cin_pad = (int)au((size_t)d.cin, 32);
    d.HW = c->H * c->W;
    d.nblocks = 2 * c->levels + 2;
    d.dh = c->dim / c->heads;
    d.M = (size_t)B * d.HW;
    return d;
}

struct UWs {
    char* zeros;
    char* xin;    // bf16 [M, cin_pad]
    char* a;      // bf16 [M, 2 dim]   GroupNorm output
    char* raw;    // bf16 [M, 2 dim]   un-normalised copy (skip conv operand)
    char* y;      // bf16 [M, dim]     FiLM + SiLU output / attention output
    char* h1;     // bf16 [M, dim]     conv1 output
    char* qkv;    // bf16 [M, 3 dim]
    float* h[3];  // fp32 [M, dim] rotating feature maps, each followed by its GroupNorm partials (map_bytes: conv epilogue -> bsi_groupnorm_apply_nhwc)
    float* skips; // fp32 [levels][M, dim], same
    size_t map_bytes, fmap_bytes;  // one feature map without / with its partials
    // split GroupNorm of the up blocks (gn_split): per level the normalised and the raw bf16 cat(x, skip), [M, 2 dim] each -- the skip
    // half (columns dim ..) is written on the way DOWN by the pass that reads the skip tensor anyway, the x half on the way up
    char* upa;
    char* upraw;
    size_t up_bytes;
    size_t total;
};

// GroupNorm statistics from the producing convolution's epilogue (one streaming normalisation pass instead of the register-resident
// reduce-then-normalise kernel); BSI_UNET_NO_GN_FUSE=1 keeps the separate kernel for comparison
inline bool gn_fuse_on(const bsi_unet_config* c) {
    static const bool no_gn_fuse = getenv("BSI_UNET_NO_GN_FUSE") != nullptr;
    return !no_gn_fuse && c->dim == 128 && (c->H * c->W) % 128 == 0;
}
// The 256-channel GroupNorm of an up block as two 128-channel halves (unet_ops.h, bsi_groupnorm_apply_split; round-3 review item 4):
// built, bit-identical results, and NOT the default -- BSI_UNET_GN_SPLIT=1 switches it on.  Measured (profiles/r4/unet_gn_split_ab.txt,
// 256 images): the up block's pass 86.4 -> 55.4 us, the pass in front of a down block 39.6 -> 69.4 us, UNet sampling +0.6-1.0 %:
// 18 % fewer bytes, but the halves are 256-B pieces at a 512-B pitch of the [M, 2 dim] operand of conv1 and stream at 4.8 TB/s where
// whole rows run at 6.2.  Contiguous halves need a two-source operand in the slab convolution; and the level buffers cost 17 GB of
// workspace at 512 images.
inline bool gn_split_on(const bsi_unet_config* c) {
    static const bool on = getenv("BSI_UNET_GN_SPLIT") != nullptr;
    return on && gn_fuse_on(c) && c->levels > 0;
}

inline UWs carve(const bsi_unet_config* c, int B, void* base) {
    const UDims d = udims(c, B);
    UWs w;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    const size_t M = d.M, dim = c->dim;
    w.zeros = p + off; off += 256;
    w.xin = p + off; off += au(M * d.cin_pad * 2);
    w.a = p + off; off += au(M * 2 * dim * 2);
    w.raw = p + off; off += au(M * 2 * dim * 2);
    w.y = p + off; off += au(M * dim * 2);
    w.h1 = p + off; off += au(M * dim * 2);
    w.qkv = p + off; off += au(M * 3 * dim * 2);
    w.map_bytes = au(M * dim * 4);
    w.fmap_bytes = w.map_bytes + au((M + 127) / 128 * (dim / 4) * 2 * 4);  // + (mean, M2) per 128 pixels x 4 channels
    for (int i = 0; i < 3; ++i) { w.h[i] = reinterpret_cast<float*>(p + off); off += w.fmap_bytes; }
    w.skips = reinterpret_cast<float*>(p + off); off += w.fmap_bytes * c->levels;
    w.up_bytes = gn_split_on(c) ? au(M * 2 * dim * 2) : 0;
    w.upa = p + off; off += w.up_bytes * c->levels;
    w.upraw = p + off; off += w.up_bytes * c->levels;
    w.total = off;
    return w;
}

#define TRY(expr)              \
    do {                       \
        int rc__ = (expr);     \
        if (rc__) return rc__; \
    } while (0)

int conv(const void* x, const void* x2, const void* w, const float* bias, const void* zeros, void* out, const float* film,
         int film_rows, int film_stride, const float* resid, int B, int H, int W, int Cin, int Cin2, int Cout, int taps, int epi,
         bsi_stream_t s, float* gn_part = nullptr) {
    bsi_conv_args a{};
    a.gn_partial = gn_part;
    a.x = x; a.x2 = x2; a.w = w; a.bias = bias; a.zeros = zeros; a.out = out; a.film = film; a.film_rows = film_rows;
    a.film_stride = film_stride; a.resid = resid; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cin2 = Cin2; a.Cout = Cout;
    a.taps = taps; a.ldo = Cout; a.epilogue = epi;
    return bsi_conv_nhwc_bf16(&a, s);
}

}  // namespace

extern "C" size_t bsi_unet_workspace_bytes(const bsi_unet_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve(cfg, B, nullptr).total;
}

extern "C" int bsi_unet_cin_pad(const bsi_unet_config* cfg) { return cfg ? udims(cfg, 1).cin_pad : 0; }

extern "C" size_t bsi_unet_film_scratch_bytes(const bsi_unet_config* cfg, int rows) {
    if (!cfg || rows <= 0) return 0;
    return 3 * au((size_t)rows * (cfg->c_dim > 64 ? cfg->c_dim : 64) * 2) + au((size_t)rows * cfg->emb_size * 4);
}

// c = pos_map(t) (vdm_unet.py:62-69) and the FiLM table of every residual block (residual_block.py:39,62):
// film[r, blk, 0:2dim] = Linear_blk(c[r]).  film: fp32 [rows, nblocks, 2*dim].
extern "C" int bsi_unet_film(const bsi_unet_config* cfg, const bsi_unet_weights* w, const float* t, int rows, float* film,
                             void* scratch, bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && t && film && scratch && rows > 0, "bsi_unet_film: bad args");
    BSI_CHECK_ARG(cfg->emb_size <= 64 && cfg->emb_size % 2 == 0 && cfg->c_dim % 64 == 0, "bsi_unet_film: emb_size=%d c_dim=%d unsupported",
                  cfg->emb_size, cfg->c_dim);
    const UDims d = udims(cfg, 1);
    char* p = reinterpret_cast<char*>(scratch);
    const size_t cb = au((size_t)rows * (cfg->c_dim > 64 ? cfg->c_dim : 64) * 2);
    char* emb = p;            // bf16 [rows, 64] (zero padded)
    char* c1 = p + cb;        // bf16 [rows, c_dim]
    char* c2 = p + 2 * cb;    // bf16 [rows, c_dim]
    float* embf = reinterpret_cast<float*>(p + 3 * cb);
    TRY(bsi_nyquist_embed(t, rows, w->pe_scale, w->pe_bias, cfg->emb_size, embf, nullptr, stream));
    TRY(bsi_cast_rows_bf16(embf, cfg->emb_size, rows, cfg->emb_size, emb, 64, stream));
    bsi_gemm_args g{};
    g.A = emb; g.W = w->pm1_w; g.bias = w->pm1_b; g.out = c1; g.M = rows; g.N = cfg->c_dim; g.K = 64; g.lda = 64; g.ldw = 64;
    g.ldo = cfg->c_dim; g.epilogue = BSI_EPI_BIAS_SILU_BF16;
    TRY(bsi_gemm_bf16(&g, stream));
    bsi_gemm_args g2{};
    g2.A = c1; g2.W = w->pm3_w; g2.bias = w->pm3_b; g2.out = c2; g2.M = rows; g2.N = cfg->c_dim; g2.K = cfg->c_dim;
    g2.lda = cfg->c_dim; g2.ldw = cfg->c_dim; g2.ldo = cfg->c_dim; g2.epilogue = BSI_EPI_BIAS_SILU_BF16;
    TRY(bsi_gemm_bf16(&g2, stream));
    bsi_gemm_args g3{};
    g3.A = c2; g3.W = w->film_w; g3.bias = w->film_b; g3.out = film; g3.M = rows; g3.N = d.nblocks * 2 * cfg->dim; g3.K = cfg->c_dim;
    g3.lda = cfg->c_dim; g3.ldw = cfg->c_dim; g3.ldo = g3.N; g3.epilogue = BSI_EPI_BIAS_F32;
    return bsi_gemm_bf16(&g3, stream);
}

extern "C" int bsi_unet_forward(const bsi_unet_config* cfg, const bsi_unet_weights* w, int B, const float* mu, const float* film,
                                int film_rows, const float* c_in, const float* c_skip, const float* c_out, int coef_stride,
                                float* out, void* workspace, bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && w->blocks && mu && film && out && workspace && B > 0, "bsi_unet_forward: bad args");
    BSI_CHECK_ARG(film_rows == 1 || film_rows == B, "bsi_unet_forward: film_rows=%d must be 1 or B=%d", film_rows, B);
    BSI_CHECK_ARG((c_in == nullptr) == (c_skip == nullptr) && (c_in == nullptr) == (c_out == nullptr),
                  "bsi_unet_forward: c_in/c_skip/c_out must be given together");
    const UDims d = udims(cfg, B);
    const int dim = cfg->dim, H = cfg->H, W = cfg->W, L = cfg->levels;
    BSI_CHECK_ARG(dim % 32 == 0 && (dim == 64 || dim == 128) && dim % cfg->heads == 0 && (d.dh == 64 || d.dh == 128) && d.HW % 64 == 0,
                  "bsi_unet_forward: unsupported geometry dim=%d heads=%d HW=%d", dim, cfg->heads, d.HW);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    UWs ws = carve(cfg, B, workspace);
    const int fstride = d.nblocks * 2 * dim;
    if (hipMemsetAsync(ws.zeros, 0, 256, s) != hipSuccess) {
        bsi_set_error("bsi_unet_forward: memset failed");
        return BSI_ELAUNCH;
    }
    // c_in*mu, Fourier features, NHWC bf16 with the channels padded to a multiple of 32 (vdm_unet.py:95-99)
    TRY(bsi_dit_prologue_launch(mu, c_in, coef_stride, B, cfg->C, H, W, 1, cfg->ff_nmin, d.nfreq, d.cin_pad, ws.xin, s));
    float* h = ws.h[0];
    int cur = 0;
    auto skip_buf = [&](int i) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws.skips) + (size_t)i * ws.fmap_bytes); };
    const bool gn_fuse = gn_fuse_on(cfg), gn_split = gn_split_on(cfg);
    auto part_of = [&](const float* map) -> float* {
        return gn_fuse ? reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<float*>(map)) + ws.map_bytes) : nullptr;
    };
    auto groupnorm = [&](const float* x1, const float* x2, int cin2, const float* gw, const float* gb, int silu, void* raw) -> int {
        if (gn_fuse)
            return bsi_groupnorm_apply_nhwc(x1, dim, part_of(x1), x2, cin2, x2 ? part_of(x2) : nullptr, B, d.HW, gw, gb, 1e-5f, silu, ws.a, raw,
                                            nullptr, stream);
        return bsi_groupnorm_nhwc(x1, dim, x2, cin2, B, d.HW, gw, gb, 1e-5f, silu, ws.a, raw, stream);
    };
    // residual block (residual_block.py:61-64): out = skip(x) + conv2(silu(film(conv1(silu(gn(x))))))
    // skip_level >= 0 (gn_split): x1 IS skip tensor `skip_level`; its pass also writes the skip half of that level's up block.
    // up_level >= 0: an up block, x2 = skip tensor `up_level`.
    auto resblock = [&](int blk, const float* x1, const float* x2, float* dst, int skip_level, int up_level) -> int {
        const bsi_unet_resblock_weights& rb = w->blocks[blk];
        const int cin2 = x2 ? dim : 0;
        const void* a_in = ws.a;      // conv1 operand
        const void* raw_in = ws.raw;  // folded 1x1 skip operand of conv2
        if (gn_split) {
            if (up_level >= 0) {  // the x half of cat(x, skip): groups of 8 channels, columns 0 .. dim of the level's buffers
                char* ua = ws.upa + (size_t)up_level * ws.up_bytes;
                char* ur = ws.upraw + (size_t)up_level * ws.up_bytes;
                TRY(bsi_groupnorm_apply_split(x1, part_of(x1), B, d.HW, 1e-5f, GnTarget{ua, ur, rb.gn_w, rb.gn_b, 2 * dim, 0, 8, 1}, GnTarget{}, stream));
                a_in = ua;
                raw_in = ur;
            } else {
                GnTarget second{};
                if (skip_level >= 0) {  // skip tensor j is popped by up iteration L - 1 - j = block L + 2 + (L - 1 - j)
                    const bsi_unet_resblock_weights& ub = w->blocks[L + 2 + (L - 1 - skip_level)];
                    second = GnTarget{ws.upa + (size_t)skip_level * ws.up_bytes, ws.upraw + (size_t)skip_level * ws.up_bytes, ub.gn_w + dim, ub.gn_b + dim,
                                      2 * dim, dim, 8, 1};
                }
                TRY(bsi_groupnorm_apply_split(x1, part_of(x1), B, d.HW, 1e-5f, GnTarget{ws.a, nullptr, rb.gn_w, rb.gn_b, dim, 0, 4, 1}, second, stream));
            }
        } else {
            TRY(groupnorm(x1, x2, cin2, rb.gn_w, rb.gn_b, 1, x2 ? ws.raw : nullptr));
        }
        // conv1 with the FiLM + SiLU epilogue of the slab kernel (one image per wave: the (scale, shift) coefficients are loaded once
        // per tile); BSI_UNET_SPLIT_FILM=1 keeps conv1 -> bf16 + a separate FiLM/SiLU pass for comparison
        static const bool split_film = getenv("BSI_UNET_SPLIT_FILM") != nullptr;
        if (split_film || d.HW % 128 != 0) {
            TRY(conv(a_in, nullptr, rb.conv1_w, rb.conv1_b, ws.zeros, ws.h1, nullptr, 0, 0, nullptr, B, H, W, dim + cin2, 0, dim, 9,
                     BSI_CONV_BIAS_BF16, stream));
            TRY(bsi_film_silu_drop(ws.h1, (int)d.M, dim, d.HW, film + (size_t)blk * 2 * dim, film_rows, fstride, DropCfg{}, ws.y, stream));
        } else {
            TRY(conv(a_in, nullptr, rb.conv1_w, rb.conv1_b, ws.zeros, ws.y, film + (size_t)blk * 2 * dim, film_rows, fstride, nullptr, B, H, W,
                     dim + cin2, 0, dim, 9, BSI_CONV_FILM_SILU_BF16, stream));
        }
        // conv2 (+ the 1x1 skip conv of cat(x, x_skip) folded in as extra K steps; its bias is folded into conv2_b)
        return conv(ws.y, x2 ? raw_in : nullptr, rb.conv2_w, rb.conv2_b, ws.zeros, dst, nullptr, 0, 0, x2 ? nullptr : x1, B, H, W, dim,
                    x2 ? 2 * dim : 0, dim, 9, BSI_CONV_BIAS_RESID_F32, stream, part_of(dst));
    };
    TRY(conv(ws.xin, nullptr, w->enc_w, w->enc_b, ws.zeros, h, nullptr, 0, 0, nullptr, B, H, W, d.cin_pad, 0, dim, 9,
             BSI_CONV_BIAS_RESID_F32, stream, part_of(h)));
    for (int i = 0; i < L; ++i) {  // down path: every block's output is also a skip tensor (simplified_unet.py:36-39)
        float* dst = skip_buf(i);
        TRY(resblock(i, h, nullptr, dst, i - 1, -1));  // block i > 0 reads skip tensor i - 1
        h = dst;
    }
    // centre: ResBlock, Residual(GroupNorm -> Attention2D), ResBlock (vdm_unet.py:80-89)
    TRY(resblock(L, h, nullptr, ws.h[cur], L - 1, -1));  // reads the last skip tensor
    h = ws.h[cur];
    TRY(groupnorm(h, nullptr, 0, w->agn_w, w->agn_b, 0, nullptr));
    TRY(conv(ws.a, nullptr, w->aqkv_w, w->aqkv_b, ws.zeros, ws.qkv, nullptr, 0, 0, nullptr, B, H, W, dim, 0, 3 * dim, 9,
             BSI_CONV_BIAS_BF16, stream));
    TRY(bsi_attention_fwd(ws.qkv, 3 * dim, B, d.HW, cfg->heads, d.dh, ws.y, dim, stream));
    {
        float* dst = ws.h[cur ^ 1];
        TRY(conv(ws.y, nullptr, w->aout_w, w->aout_b, ws.zeros, dst, nullptr, 0, 0, h, B, H, W, dim, 0, dim, 9, BSI_CONV_BIAS_RESID_F32,
                 stream, part_of(dst)));
        h = dst;
        cur ^= 1;
    }
    {
        float* dst = ws.h[2];
        TRY(resblock(L + 1, h, nullptr, dst, -1, -1));
        h = dst;
    }
    // up path: block(cat(x, skips.pop()))  (simplified_unet.py:43-46)
    int pp = 0;
    for (int i = 0; i < L; ++i) {
        float* dst = ws.h[pp];
        TRY(resblock(L + 2 + i, h, skip_buf(L - 1 - i), dst, -1, L - 1 - i));
        h = dst;
        pp ^= 1;
    }
    return bsi_unet_decode(h, B, d.HW, dim, w->dec_w, w->dec_b, cfg->C, mu, c_skip, c_out, coef_stride, out, stream);
}
