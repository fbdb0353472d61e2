// Host-side sequencing of one DenoisingDiT evaluation (bsi/models/dit.py:174-181,225-233 of the
// reference) on a HIP stream.  No allocation, no synchronisation: the caller owns the workspace.
#include "common.h"
#include "dit_ops.h"
#include "prof.h"

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DitDims {
    int tokens, kin, kpad, P, nfreq, cin;
};

inline DitDims dims_of(const bsi_dit_config* c) {
    DitDims d;
    d.nfreq = (c->ff_nmax >= c->ff_nmin) ? (c->ff_nmax - c->ff_nmin + 1) : 0;
    d.cin = c->C + c->C * d.nfreq * 2;
    d.tokens = (c->H / c->patch) * (c->W / c->patch);
    d.kin = c->patch * c->patch * d.cin;
    d.kpad = (int)align_up((size_t)d.kin, 64);
    d.P = c->patch * c->patch * c->C;
    return d;
}

inline int check_cfg(const bsi_dit_config* c) {
    BSI_CHECK_ARG(c != nullptr, "dit: null config");
    BSI_CHECK_ARG(c->C > 0 && c->H > 0 && c->W > 0 && c->patch > 0 && c->H % c->patch == 0 && c->W % c->patch == 0,
                  "dit: bad data shape %dx%dx%d / patch %d", c->C, c->H, c->W, c->patch);
    BSI_CHECK_ARG(c->dim > 0 && c->dim % 64 == 0 && c->depth > 0 && c->heads > 0 && c->dim % c->heads == 0,
                  "dit: dim=%d must be a multiple of 64 and divisible by heads=%d", c->dim, c->heads);
    const int dh = c->dim / c->heads;
    BSI_CHECK_ARG(dh == 64 || dh == 128, "dit: head dim %d unsupported (64 or 128)", dh);
    const DitDims d = dims_of(c);
    BSI_CHECK_ARG(d.tokens % 64 == 0, "dit: %d tokens per image; must be a multiple of 64", d.tokens);
    BSI_CHECK_ARG(d.P <= 64, "dit: patch*patch*C = %d > 64 unsupported", d.P);
    BSI_CHECK_ARG((c->H * c->W) % 4 == 0, "dit: H*W must be a multiple of 4");
    return BSI_OK;
}

struct Workspace {
    char* a0;   // bf16 [M, kpad]
    float* x;   // fp32 [M, dim]
    char* xn;   // bf16 [M, dim]
    char* big;  // bf16 [M, 4*dim]
    char* da;   // bf16 [M, dim]: the attention branch's delta, alive until the NEXT block's first LayerNorm pass stores the row
    char* splitk;  // fp32 partial sums of the split-K latency path (a few images per call), 0 bytes for large batches
    size_t splitk_bytes;
    size_t total;
};

inline Workspace carve(const bsi_dit_config* c, int B, void* base) {
    const DitDims d = dims_of(c);
    const size_t M = (size_t)B * d.tokens;
    Workspace w;
    size_t off = 0;
    char* p = reinterpret_cast<char*>(base);
    w.a0 = p + off; off += align_up(M * d.kpad * 2, 256);
    w.x = reinterpret_cast<float*>(p + off); off += align_up(M * c->dim * 4, 256);
    w.xn = p + off; off += align_up(M * c->dim * 2, 256);
    w.big = p + off; off += align_up(M * 4 * (size_t)c->dim * 2, 256);
    w.da = p + off; off += align_up(M * c->dim * 2, 256);
    {
        const int Mi = (int)M, dm = c->dim;
        size_t sk = bsi_gemm_splitk_workspace_bytes(Mi, 3 * dm, dm);
        const size_t c2 = bsi_gemm_splitk_workspace_bytes(Mi, dm, dm), c3 = bsi_gemm_splitk_workspace_bytes(Mi, 4 * dm, dm),
                     c4 = bsi_gemm_splitk_workspace_bytes(Mi, dm, 4 * dm);
        sk = sk > c2 ? sk : c2; sk = sk > c3 ? sk : c3; sk = sk > c4 ? sk : c4;
        w.splitk = p + off; w.splitk_bytes = sk; off += align_up(sk, 256);
    }
    w.total = off;
    return w;
}

}  // namespace

extern "C" int bsi_dit_kpad(const bsi_dit_config* cfg) { return cfg ? dims_of(cfg).kpad : 0; }
extern "C" int bsi_dit_tokens(const bsi_dit_config* cfg) { return cfg ? dims_of(cfg).tokens : 0; }

extern "C" size_t bsi_dit_workspace_bytes(const bsi_dit_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve(cfg, B, nullptr).total;
}

extern "C" size_t bsi_dit_adaln_scratch_bytes(const bsi_dit_config* cfg, int rows) {
    if (!cfg || rows <= 0) return 0;
    return 2 * align_up((size_t)rows * cfg->dim * 2, 256);
}

extern "C" int bsi_dit_adaln(const bsi_dit_config* cfg, const bsi_dit_weights* w, const float* t, int rows, float* mod,
                             void* scratch, bsi_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    BSI_CHECK_ARG(w && w->blocks && t && mod && scratch && rows > 0, "bsi_dit_adaln: bad args");
    const int dim = cfg->dim;
    char* emb = reinterpret_cast<char*>(scratch);
    char* hid = emb + align_up((size_t)rows * dim * 2, 256);
    // c = t_embedding(t)  (dit.py:177)
    ProfScope prof(BSI_PROF_ADALN, reinterpret_cast<hipStream_t>(stream));
    if (int rc = bsi_nyquist_embed(t, rows, w->t_scale, w->t_bias, dim, nullptr, emb, stream)) return rc;
    for (int l = 0; l < cfg->depth; ++l) {
        const bsi_dit_block_weights& bw = w->blocks[l];
        bsi_gemm_args g{};
        g.A = emb; g.W = bw.ada0_w; g.bias = bw.ada0_b; g.out = hid;
        g.M = rows; g.N = dim; g.K = dim; g.lda = dim; g.ldw = dim; g.ldo = dim;
        g.epilogue = BSI_EPI_BIAS_SILU_BF16;  // Linear -> SiLU  (dit.py:79-80)
        if (int rc = bsi_gemm_bf16(&g, stream)) return rc;
        bsi_gemm_args g2{};
        g2.A = hid; g2.W = bw.ada2_w; g2.bias = bw.ada2_b; g2.out = mod + (size_t)l * 6 * dim;
        g2.M = rows; g2.N = 6 * dim; g2.K = dim; g2.lda = dim; g2.ldw = dim; g2.ldo = cfg->depth * 6 * dim;
        g2.epilogue = BSI_EPI_BIAS_F32;  // Linear(size, 6*size)  (dit.py:80)
        if (int rc = bsi_gemm_bf16(&g2, stream)) return rc;
    }
    return BSI_OK;
}

extern "C" int bsi_dit_forward(const bsi_dit_config* cfg, const bsi_dit_weights* w, int B, const float* mu,
                               const float* mod, int mod_rows, const float* c_in, const float* c_skip,
                               const float* c_out, int coef_stride, float* out, void* workspace, float* tokens_out,
                               bsi_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    BSI_CHECK_ARG(w && w->blocks && mu && mod && out && workspace && B > 0, "bsi_dit_forward: bad args");
    BSI_CHECK_ARG(mod_rows == 1 || mod_rows == B, "bsi_dit_forward: mod_rows=%d must be 1 or B=%d", mod_rows, B);
    BSI_CHECK_ARG((c_in == nullptr) == (c_skip == nullptr) && (c_in == nullptr) == (c_out == nullptr),
                  "bsi_dit_forward: c_in/c_skip/c_out must be given together");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const DitDims d = dims_of(cfg);
    const int dim = cfg->dim;
    const int M = B * d.tokens;
    const int mod_stride = cfg->depth * 6 * dim;
    Workspace ws = carve(cfg, B, workspace);

    // 1. c_in*mu -> Fourier features -> patchify -> bf16 tokens
    {
        ProfScope prof(BSI_PROF_PROLOGUE, s);
        if (int rc = bsi_dit_prologue_launch(mu, c_in, coef_stride, B, cfg->C, cfg->H, cfg->W, cfg->patch,
                                             cfg->ff_nmin, d.nfreq, d.kpad, ws.a0, s))
            return rc;
    }
    // 2. patch encoder + positional embedding (dit.py:178)
    {
        bsi_gemm_args g{};
        g.A = ws.a0; g.W = w->enc_w; g.bias = w->enc_b; g.out = ws.x;
        g.M = M; g.N = dim; g.K = d.kpad; g.lda = d.kpad; g.ldw = d.kpad; g.ldo = dim;
        g.epilogue = BSI_EPI_BIAS_POS_F32; g.pos = w->pos; g.tokens = d.tokens;
        ProfScope prof(BSI_PROF_GEMM_ENC, s);
        if (int rc = bsi_gemm_bf16(&g, stream)) return rc;
    }
    // 3. blocks (dit.py:87-103).  Every branch GEMM stores its output (bias included) as a bf16 "delta"; the gated
    //    residual update x += gate*delta is fused into the NEXT LayerNorm+modulate pass (or the final kernel).
    //    The row is STORED once per block, not once per branch: the pass in front of the MLP applies the attention update in
    //    registers only (write_x = 0) and the next block's first pass applies both updates and stores the row -- the same two
    //    fp32 fmas per element in the same order, so results are bit-identical to the eager form (BSI_DIT_EAGER_RESID=1 keeps
    //    it for comparison), with 302 MB less traffic per block at 512 images.
    static const bool eager_resid = getenv("BSI_DIT_EAGER_RESID") != nullptr;
    const void* pend_delta = nullptr;  // delta of the previous branch, not yet added to x
    const float* pend_gate = nullptr;
    const void* pend_delta0 = nullptr;  // the update in front of it, applied by the previous pass in registers only
    const float* pend_gate0 = nullptr;
    for (int l = 0; l < cfg->depth; ++l) {
        const bsi_dit_block_weights& bw = w->blocks[l];
        const float* ml = mod + (size_t)l * 6 * dim;
        {
            ProfScope prof(BSI_PROF_LN, s);
            if (int rc = bsi_resid2_ln_modulate(ws.x, M, dim, 1e-5f, pend_delta0, pend_gate0, pend_delta, pend_gate, 1, ml, ml + dim,
                                                mod_rows, mod_stride, d.tokens, ws.xn, stream))
                return rc;
        }
        bsi_gemm_args g{};
        g.A = ws.xn; g.W = bw.qkv_w; g.bias = bw.qkv_b; g.out = ws.big;
        g.M = M; g.N = 3 * dim; g.K = dim; g.lda = dim; g.ldw = dim; g.ldo = 3 * dim;
        g.epilogue = BSI_EPI_BIAS_BF16;
        {
            ProfScope prof(BSI_PROF_GEMM_QKV, s);
            if (int rc = bsi_gemm_bf16_ws(&g, ws.splitk, ws.splitk_bytes, stream)) return rc;
        }
        {
            ProfScope prof(BSI_PROF_ATTN, s);
            if (int rc = bsi_attention_fwd(ws.big, 3 * dim, B, d.tokens, cfg->heads, dim / cfg->heads, ws.xn, dim,
                                           stream))
                return rc;
        }
        bsi_gemm_args go{};  // attention output projection -> delta
        go.A = ws.xn; go.W = bw.out_w; go.bias = bw.out_b; go.out = ws.da;
        go.M = M; go.N = dim; go.K = dim; go.lda = dim; go.ldw = dim; go.ldo = dim;
        go.epilogue = BSI_EPI_BIAS_BF16;
        {
            ProfScope prof(BSI_PROF_GEMM_OUT, s);
            if (int rc = bsi_gemm_bf16_ws(&go, ws.splitk, ws.splitk_bytes, stream)) return rc;
        }
        const bool lazy = !eager_resid && l + 1 < cfg->depth;  // the last block stores: the final kernel takes one pending update
        {   // x + gate_msa * delta (stored unless lazy); xn = LN(that) * (1 + scale_mlp) + shift_mlp
            ProfScope prof(BSI_PROF_LN, s);
            if (int rc = bsi_resid2_ln_modulate(ws.x, M, dim, 1e-5f, nullptr, nullptr, ws.da, ml + 2 * dim, lazy ? 0 : 1, ml + 3 * dim,
                                                ml + 4 * dim, mod_rows, mod_stride, d.tokens, ws.xn, stream))
                return rc;
        }
        pend_delta0 = lazy ? ws.da : nullptr;
        pend_gate0 = lazy ? ml + 2 * dim : nullptr;
        bsi_gemm_args g1{};
        g1.A = ws.xn; g1.W = bw.fc1_w; g1.bias = bw.fc1_b; g1.out = ws.big;
        g1.M = M; g1.N = 4 * dim; g1.K = dim; g1.lda = dim; g1.ldw = dim; g1.ldo = 4 * dim;
        g1.epilogue = BSI_EPI_BIAS_GELU_BF16;
        {
            ProfScope prof(BSI_PROF_GEMM_FC1, s);
            if (int rc = bsi_gemm_bf16_ws(&g1, ws.splitk, ws.splitk_bytes, stream)) return rc;
        }
        bsi_gemm_args g2{};  // MLP output -> delta (in the now dead xn buffer)
        g2.A = ws.big; g2.W = bw.fc2_w; g2.bias = bw.fc2_b; g2.out = ws.xn;
        g2.M = M; g2.N = dim; g2.K = 4 * dim; g2.lda = 4 * dim; g2.ldw = 4 * dim; g2.ldo = dim;
        g2.epilogue = BSI_EPI_BIAS_BF16;
        {
            ProfScope prof(BSI_PROF_GEMM_FC2, s);
            if (int rc = bsi_gemm_bf16_ws(&g2, ws.splitk, ws.splitk_bytes, stream)) return rc;
        }
        pend_delta = ws.xn;
        pend_gate = ml + 5 * dim;
    }
    if (tokens_out) {  // tests: materialise the final residual stream
        if (int rc = bsi_resid_ln_modulate(ws.x, M, dim, 1e-5f, pend_delta, pend_gate, nullptr, nullptr, mod_rows,
                                           mod_stride, d.tokens, nullptr, nullptr, nullptr, stream))
            return rc;
        pend_delta = nullptr;
        pend_gate = nullptr;
    }
    if (tokens_out) {
        hipError_t e = hipMemcpyAsync(tokens_out, ws.x, (size_t)M * dim * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) {
            bsi_set_error("bsi_dit_forward: tokens copy failed: %s", hipGetErrorString(e));
            return BSI_ELAUNCH;
        }
    }
    // 4. LayerNorm + Linear + unpatchify (+ x_hat = c_skip*mu + c_out*f)
    ProfScope prof(BSI_PROF_FINAL, s);
    return bsi_dit_final_launch(ws.x, M, dim, d.P, w->dec_ln_w, w->dec_ln_b, w->dec_w, w->dec_b, cfg->C, cfg->H, cfg->W,
                                cfg->patch, mu, c_skip, c_out, coef_stride, pend_delta, pend_gate, mod_rows, mod_stride, out,
                                s);
}
