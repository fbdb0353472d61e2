// Training engine of the DenoisingDiT: forward that records a tape, and the hand-written backward
// (replaces torch autograd over bsi/models/dit.py:87-103,174-181 of the reference in `BSI.train_loss(...).mean().backward()`,
// bsi/tasks/bsi.py:187-194).  Host-side sequencing only; no allocation, no synchronisation.
//
// Tape per block and token row (bf16 unless noted): xn1 [d], qkv [3d], lse (fp32 per head), ao [d], d1 [d], xn2 [d],
// hp [4d] (pre-GELU), h [4d], d2 [d], and the two LayerNorm inputs xa, xb (fp32 [d]) with their (mean, rstd)
// = 16 d bf16 + 2 d fp32 = 40 KB per token at d = 1024 (10 MB per image per block, 240 MB per image for DiT-L: sized for the
// 288 GB of HBM3E).  Keeping every LayerNorm input lets the backward fuse LayerNorm-backward with the gated-residual
// backward in one 18-B-per-element pass (bsi_ln_gate_bwd) instead of rewinding the stream (two passes, 30 B).
#include "common.h"
#include "dit_ops.h"
#include "prof.h"

int bsi_dit_final_bwd_launch(const float* x, int Mtok, int d, int P, const float* ln_w, const float* ln_b,
                             const float* dec_w, int C, int H, int W, int ps, const float* g_xhat, const float* c_out,
                             int coef_stride, float* dX, void* yb, void* dYb, float* d_dec_b, float* d_ln_w,
                             float* d_ln_b, float* parts, hipStream_t s);
size_t bsi_dit_final_bwd_parts_floats(int Mtok, int d);

namespace {

inline size_t au(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Dims {
    int tokens, kin, kpad, P, nfreq, cin, dim, depth, heads;
    size_t M;
};

inline Dims dims_of(const bsi_dit_config* c, int B) {
    Dims d;
    d.nfreq = (c->ff_nmax >= c->ff_nmin) ? (c->ff_nmax - c->ff_nmin + 1) : 0;
    d.cin = c->C + c->C * d.nfreq * 2;
    d.tokens = (c->H / c->patch) * (c->W / c->patch);
    d.kin = c->patch * c->patch * d.cin;
    d.kpad = (int)au((size_t)d.kin, 64);
    d.P = c->patch * c->patch * c->C;
    d.dim = c->dim; d.depth = c->depth; d.heads = c->heads;
    d.M = (size_t)B * d.tokens;
    return d;
}

struct BlockTape {
    char *xn1, *qkv, *ao, *d1, *xn2, *hp, *h, *d2;
    float* lse;
    float *xa, *xb;    // fp32 [M, dim]: inputs of LayerNorm 1 / 2 (xa of block 0 is written by the patch encoder)
    float *sa, *sb;    // fp32 [M, 2]: (mean, rstd) of those rows
    char* maskw;       // [B * heads][8 KB]: dropout-mask words of the attention weights (persistent kernels, 256 tokens x head dim 64)
};

struct Tape {
    char* a0;       // bf16 [M, kpad]
    float* x;       // fp32 [M, dim]  final residual stream (input of the decoder's LayerNorm)
    float* mod;     // fp32 [B, depth, 6 dim]
    char* emb;      // bf16 [B, dim]
    float* ada_pre; // fp32 [depth, B, dim]
    char* ada_s;    // bf16 [depth, B, dim]
    char* skws;     // split-K slabs of the per-sample adaLN GEMMs (bsi_gemm_splitk_f32_workspace_bytes)
    size_t skws_bytes;
    char* blocks;   // per block region
    size_t block_bytes, total;
};

// The dropout-mask words of the attention weights (8 KB per (image, head) and block) exist only for the geometry whose attention
// kernels consume them (bsi_attention_uses_mask_words: 256 tokens, head dim 64); elsewhere the region is empty and BlockTape::maskw
// is NULL, so no kernel can read words that were never written (forward and backward carve the tape with this one function).
inline size_t mask_words_bytes(const Dims& d, int B) {
    return bsi_attention_uses_mask_words(d.tokens, 64) ? au((size_t)B * d.heads * 8192) : 0;
}

inline Tape carve_tape(const Dims& d, int B, void* base) {
    Tape t;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    const size_t M = d.M, dim = d.dim;
    t.a0 = p + off; off += au(M * d.kpad * 2);
    t.x = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    t.mod = reinterpret_cast<float*>(p + off); off += au((size_t)B * d.depth * 6 * dim * 4);
    t.emb = p + off; off += au((size_t)B * dim * 2);
    t.ada_pre = reinterpret_cast<float*>(p + off); off += au((size_t)d.depth * B * dim * 4);
    t.ada_s = p + off; off += au((size_t)d.depth * B * dim * 2);
    {
        const size_t a0 = bsi_gemm_splitk_f32_workspace_bytes(B, (int)dim, (int)dim), a2 = bsi_gemm_splitk_f32_workspace_bytes(B, 6 * (int)dim, (int)dim);
        t.skws_bytes = a0 > a2 ? a0 : a2;
        t.skws = p + off; off += au(t.skws_bytes);
    }
    t.blocks = p + off;
    t.block_bytes = au(M * dim * 2) * 5 + au(M * 3 * dim * 2) + au(M * 4 * dim * 2) * 2 + au((size_t)B * d.heads * d.tokens * 4) +
                    au(M * dim * 4) * 2 + au(M * 2 * 4) * 2 + mask_words_bytes(d, B);
    off += t.block_bytes * d.depth;
    t.total = off;
    return t;
}

inline BlockTape block_tape(const Tape& t, const Dims& d, int B, int l) {
    BlockTape b;
    char* p = t.blocks + (size_t)l * t.block_bytes;
    const size_t M = d.M, dim = d.dim;
    size_t off = 0;
    b.xn1 = p + off; off += au(M * dim * 2);
    b.qkv = p + off; off += au(M * 3 * dim * 2);
    b.ao = p + off; off += au(M * dim * 2);
    b.d1 = p + off; off += au(M * dim * 2);
    b.xn2 = p + off; off += au(M * dim * 2);
    b.hp = p + off; off += au(M * 4 * dim * 2);
    b.h = p + off; off += au(M * 4 * dim * 2);
    b.d2 = p + off; off += au(M * dim * 2);
    b.lse = reinterpret_cast<float*>(p + off); off += au((size_t)B * d.heads * d.tokens * 4);
    b.xa = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    b.xb = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    b.sa = reinterpret_cast<float*>(p + off); off += au(M * 2 * 4);
    b.sb = reinterpret_cast<float*>(p + off); off += au(M * 2 * 4);
    b.maskw = mask_words_bytes(d, B) ? p + off : nullptr;
    return b;
}

struct BwdWs {
    float* dX;     // fp32 [M, dim]
    char* dd;      // bf16 [M, dim]     gradient of a branch delta
    char* dbig;    // bf16 [M, 4 dim]   dhp / dqkv
    char* dsmall;  // bf16 [M, dim]     dxn / dao
    float* dmod;   // fp32 [2 (block parity)][tokens / 64 planes][B, 6 dim]: per-slab sums of the modulation gradients (reproducible: no atomics)
    char* dmod_bf; // bf16 [B, 6 dim]
    float* fin_parts;  // per-block sums of the final layer's parameter gradients (bsi_dit_final_bwd_launch)
    float* ds;     // fp32 [B, dim]
    char* dpre_bf; // bf16 [B, dim]
    char* tn;      // TN GEMM slabs
    char* cs;      // colsum slabs
    float* decw;   // fp32 [64, dim]   decoder weight gradient with its rows padded to a multiple of 8
    // bias gradients taken where dY is produced (round 4): per-slab column sums, reduced once per block by bsi_colsum_rows_f32
    float* rows_fc2;  // fp32 [M / 64][dim]      sums of the dd2 rows (written by the kernel that writes ws.dd for the MLP branch)
    float* rows_out;  // fp32 [M / 64][dim]      sums of the dd1 rows
    float* rows_fc1;  // fp32 [ceil(M / 128)][4 dim]  sums of the dhp rows (GEMM epilogue)
    float* rows_qkv;  // fp32 [B][3 dim]          sums of the dqkv rows of an image (single-sweep attention backward)
    char* rows_scratch;
    size_t total;
};

inline BwdWs carve_bwd(const Dims& d, int B, void* base) {
    BwdWs w;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    const size_t M = d.M, dim = d.dim;
    w.dX = reinterpret_cast<float*>(p + off); off += au(M * dim * 4);
    w.dd = p + off; off += au(M * dim * 2);
    w.dbig = p + off; off += au(M * 4 * dim * 2);
    w.dsmall = p + off; off += au(M * dim * 2);
    w.dmod = reinterpret_cast<float*>(p + off); off += au((size_t)2 * (d.tokens / 64 > 0 ? d.tokens / 64 : 1) * B * 6 * dim * 4);
    w.dmod_bf = p + off; off += au((size_t)B * 6 * dim * 2);
    w.fin_parts = reinterpret_cast<float*>(p + off); off += au(bsi_dit_final_bwd_parts_floats((int)d.M, (int)dim) * 4);
    w.ds = reinterpret_cast<float*>(p + off); off += au((size_t)B * dim * 4);
    w.dpre_bf = p + off; off += au((size_t)B * dim * 2);
    w.tn = p + off; off += au(bsi_gemm_tn_workspace_bytes((int)M, 4 * (int)dim, (int)dim));
    w.cs = p + off; off += au(bsi_colsum_workspace_bytes(6 * (int)dim));
    w.decw = reinterpret_cast<float*>(p + off); off += au((size_t)64 * dim * 4);
    const size_t r64 = (M + 63) / 64, r128 = (M + 127) / 128;
    w.rows_fc2 = reinterpret_cast<float*>(p + off); off += au(r64 * dim * 4);
    w.rows_out = reinterpret_cast<float*>(p + off); off += au(r64 * dim * 4);
    w.rows_fc1 = reinterpret_cast<float*>(p + off); off += au(r128 * 4 * dim * 4);
    w.rows_qkv = reinterpret_cast<float*>(p + off); off += au((size_t)B * 3 * dim * 4);
    w.rows_scratch = p + off;
    off += au(2 * bsi_colsum_rows_scratch_bytes((int)r64, (int)dim) + bsi_colsum_rows_scratch_bytes((int)r128, 4 * (int)dim) +
              bsi_colsum_rows_scratch_bytes(B, 3 * (int)dim));
    w.total = off;
    return w;
}

int gemm(const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldo, int M, int N, int K, int epi,
         const void* aux, void* out2, const float* pos, int tokens, bsi_stream_t stream, float* colsum_rows = nullptr) {
    bsi_gemm_args g{};
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.epilogue = epi; g.aux = aux; g.out2 = out2; g.pos = pos; g.tokens = tokens; g.colsum_rows = colsum_rows;
    return bsi_gemm_bf16(&g, stream);
}

// per-sample GEMM (M = images) with an fp32 result: split over K through `ws` when the shape qualifies (gemm_bf16.hip, splitk_plan_f32)
int gemm_rows(const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldo, int M, int N, int K, void* ws,
              size_t ws_bytes, bsi_stream_t stream) {
    bsi_gemm_args g{};
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
    g.epilogue = BSI_EPI_BIAS_F32;
    return bsi_gemm_bf16_ws(&g, ws, ws_bytes, stream);
}

void* const* g_block_events = nullptr;
int g_block_events_n = 0;

#define TRY(expr)            \
    do {                     \
        int rc__ = (expr);   \
        if (rc__) return rc__; \
    } while (0)

}  // namespace

// Gates of the NEXT forwards (bsi_dit_train_forward_set_gates): [0] what the front of the network needs (patch encoder), [1 + l]
// block l, [depth + 1] the decoder.
const bsi_fwd_gate* g_fwd_gates = nullptr;
int g_fwd_gates_n = 0;

extern "C" int bsi_dit_train_forward_set_gates(const bsi_fwd_gate* gates, int n) {
    g_fwd_gates = gates;
    g_fwd_gates_n = gates ? n : 0;
    return BSI_OK;
}

namespace {
// wait for the gate's event (the parameters behind it have arrived), then refresh the bf16 shadows that are cast from them
int pass_gate(const bsi_fwd_gate& g, hipStream_t s) {
    if (g.event) {
        hipError_t e = hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(g.event), 0);
        if (e != hipSuccess) {
            bsi_set_error("bsi_dit_train_forward: waiting for a parameter gate failed: %s", hipGetErrorString(e));
            return BSI_ELAUNCH;
        }
    }
    if (g.cast && g.n_cast > 0 && g.cast_tiles > 0) return bsi_cast_batch_bf16(g.cast, g.n_cast, g.cast_tiles, reinterpret_cast<bsi_stream_t>(s));
    return BSI_OK;
}
}  // namespace

extern "C" int bsi_dit_backward_set_events(void* const* events, int depth) {
    g_block_events = events;
    g_block_events_n = events ? depth : 0;
    return BSI_OK;
}

extern "C" size_t bsi_dit_tape_bytes(const bsi_dit_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve_tape(dims_of(cfg, B), B, nullptr).total;
}

extern "C" size_t bsi_dit_backward_workspace_bytes(const bsi_dit_config* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return carve_bwd(dims_of(cfg, B), B, nullptr).total;
}

extern "C" int bsi_silu_bf16(const float* pre, size_t n, void* out, bsi_stream_t stream);

extern "C" int bsi_dit_train_forward(const bsi_dit_config* cfg, const bsi_dit_weights* w, int B, const float* mu,
                                     const float* t, const float* c_in, const float* c_skip, const float* c_out,
                                     float* out, void* tape_mem, float dropout_p, unsigned long long seed,
                                     bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && w->blocks && mu && t && out && tape_mem && B > 0, "bsi_dit_train_forward: bad args");
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_dit_train_forward: dropout probability %g outside [0, 1)", (double)dropout_p);
    BSI_CHECK_ARG((c_in == nullptr) == (c_skip == nullptr) && (c_in == nullptr) == (c_out == nullptr),
                  "bsi_dit_train_forward: c_in/c_skip/c_out must be given together");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Dims d = dims_of(cfg, B);
    BSI_CHECK_ARG(d.tokens % 64 == 0 && d.dim % 64 == 0 && d.dim / d.heads == 64 && d.tokens <= 256 && d.dim <= 1024,
                  "bsi_dit_train_forward: unsupported geometry (tokens %d, dim %d, head dim %d)", d.tokens, d.dim, d.dim / d.heads);
    const int dim = d.dim, M = (int)d.M;
    const int mod_stride = d.depth * 6 * dim;
    Tape tp = carve_tape(d, B, tape_mem);

    // Gated forward (data-parallel sharded step, bsi_dit_train_forward_set_gates): the parameters of block l may still be on their
    // way (all-gather on another stream) when this function is called; block l waits for ITS gate only, then casts its own bf16
    // shadows and runs its own adaLN MLP (the grouped launches below would need every block's parameters before the first block).
    const bsi_fwd_gate* gates = g_fwd_gates;
    BSI_CHECK_ARG(!gates || g_fwd_gates_n == d.depth + 2, "bsi_dit_train_forward: %d gates set, the model needs depth + 2 = %d",
                  g_fwd_gates_n, d.depth + 2);
    if (gates) TRY(pass_gate(gates[0], s));
    // adaLN tables, keeping the intermediate activations (dit.py:77-81)
    TRY(bsi_nyquist_embed(t, B, w->t_scale, w->t_bias, dim, nullptr, tp.emb, stream));
    // The adaLN MLP of every block depends on t only.  When the blocks' matrices and biases lie at uniform strides (the model's cast
    // plan allocates the shadows that way; the biases do when the parameters live in one flat buffer, as under DPTrainer) the 2 x depth
    // per-sample GEMMs are TWO grouped launches; otherwise one split-K GEMM per block and matrix as before.
    bool grouped = d.depth > 1 && dim >= 96;
    ptrdiff_t sw0 = 0, sb0 = 0, sw2 = 0, sb2 = 0;
    if (grouped) {
        auto diff = [](const void* a, const void* b) { return reinterpret_cast<const char*>(a) - reinterpret_cast<const char*>(b); };
        const bsi_dit_block_weights* bk = w->blocks;
        sw0 = diff(bk[1].ada0_w, bk[0].ada0_w); sb0 = diff(bk[1].ada0_b, bk[0].ada0_b);
        sw2 = diff(bk[1].ada2_w, bk[0].ada2_w); sb2 = diff(bk[1].ada2_b, bk[0].ada2_b);
        for (int l = 1; l < d.depth && grouped; ++l)
            grouped = diff(bk[l].ada0_w, bk[0].ada0_w) == l * sw0 && diff(bk[l].ada0_b, bk[0].ada0_b) == l * sb0 &&
                      diff(bk[l].ada2_w, bk[0].ada2_w) == l * sw2 && diff(bk[l].ada2_b, bk[0].ada2_b) == l * sb2;
        // positive strides only: they travel as size_t (a block order that runs DOWN in memory takes the per-block path)
        grouped = grouped && sw0 > 0 && sb0 > 0 && sw2 > 0 && sb2 > 0 && sw0 % 16 == 0 && sb0 % 16 == 0 && sw2 % 16 == 0 && sb2 % 16 == 0;
    }
    // one block's adaLN MLP: as a group of ONE of the grouped kernel when the whole model would take the grouped path (the same
    // arithmetic per block: a gated forward returns the bits of the ungated one), else the split-K GEMMs
    auto adaln_block = [&](int l) -> int {
        const bsi_dit_block_weights& bw = w->blocks[l];
        float* pre = tp.ada_pre + (size_t)l * B * dim;
        char* sl = tp.ada_s + (size_t)l * B * dim * 2;
        if (grouped) {
            bsi_gemm_args g{};
            g.A = tp.emb; g.W = bw.ada0_w; g.bias = bw.ada0_b; g.out = pre;
            g.M = B; g.N = dim; g.K = dim; g.lda = dim; g.ldw = dim; g.ldo = dim; g.epilogue = BSI_EPI_BIAS_F32;
            TRY(bsi_gemm_bf16_grouped(&g, 1, 0, 0, 0, 0, stream));
            TRY(bsi_silu_bf16(pre, (size_t)B * dim, sl, stream));
            g.A = sl; g.W = bw.ada2_w; g.bias = bw.ada2_b; g.out = tp.mod + (size_t)l * 6 * dim;
            g.N = 6 * dim; g.ldo = mod_stride;
            return bsi_gemm_bf16_grouped(&g, 1, 0, 0, 0, 0, stream);
        }
        TRY(gemm_rows(tp.emb, dim, bw.ada0_w, dim, bw.ada0_b, pre, dim, B, dim, dim, tp.skws, tp.skws_bytes, stream));
        TRY(bsi_silu_bf16(pre, (size_t)B * dim, sl, stream));
        return gemm_rows(sl, dim, bw.ada2_w, dim, bw.ada2_b, tp.mod + (size_t)l * 6 * dim, mod_stride, B, 6 * dim, dim, tp.skws, tp.skws_bytes, stream);
    };
    if (gates) {
        // per block, inside the loop below
    } else if (grouped) {
        bsi_gemm_args g{};
        g.A = tp.emb; g.W = w->blocks[0].ada0_w; g.bias = w->blocks[0].ada0_b; g.out = tp.ada_pre;
        g.M = B; g.N = dim; g.K = dim; g.lda = dim; g.ldw = dim; g.ldo = dim; g.epilogue = BSI_EPI_BIAS_F32;
        TRY(bsi_gemm_bf16_grouped(&g, d.depth, 0, (size_t)sw0, (size_t)sb0, (size_t)B * dim * 4, stream));
        TRY(bsi_silu_bf16(tp.ada_pre, (size_t)d.depth * B * dim, tp.ada_s, stream));
        g.A = tp.ada_s; g.W = w->blocks[0].ada2_w; g.bias = w->blocks[0].ada2_b; g.out = tp.mod;
        g.N = 6 * dim; g.ldo = mod_stride;
        TRY(bsi_gemm_bf16_grouped(&g, d.depth, (size_t)B * dim * 2, (size_t)sw2, (size_t)sb2, (size_t)6 * dim * 4, stream));
    } else {
        for (int l = 0; l < d.depth; ++l) TRY(adaln_block(l));
    }
    TRY(bsi_dit_prologue_launch(mu, c_in, 1, B, cfg->C, cfg->H, cfg->W, cfg->patch, cfg->ff_nmin, d.nfreq, d.kpad, tp.a0, s));
    TRY(gemm(tp.a0, d.kpad, w->enc_w, d.kpad, w->enc_b, block_tape(tp, d, B, 0).xa, dim, M, dim, d.kpad, BSI_EPI_BIAS_POS_F32,
             nullptr, nullptr, w->pos, d.tokens, stream));
    const void* pend_delta = nullptr;
    const float* pend_gate = nullptr;
    float* x_prev = nullptr;  // residual stream before the pending branch is added
    for (int l = 0; l < d.depth; ++l) {
        const bsi_dit_block_weights& bw = w->blocks[l];
        BlockTape bt = block_tape(tp, d, B, l);
        const float* ml = tp.mod + (size_t)l * 6 * dim;
        if (gates) {
            TRY(pass_gate(gates[1 + l], s));
            TRY(adaln_block(l));
        }
        // xa = x_prev + gate * d2 of the block below (block 0: the encoder wrote xa), xn1 = LN(xa) * (1 + scale) + shift
        // (with dropout on the DiT geometry, this HBM-bound pass also computes the attention layer's dropout-mask words: row m ->
        //  block m, M = B * tokens = B * heads * 16 blocks when tokens == 16 * heads)
        const DropCfg adc = make_drop(dropout_p, seed, 2 * l);
        const bool ride = adc.thr != 0 && bsi_attention_uses_mask_words(d.tokens, 64) && d.tokens == 16 * d.heads && dim > 256 && dim <= 1024;
        TRY(bsi_resid_ln_modulate_drop(l == 0 ? bt.xa : x_prev, M, dim, 1e-5f, pend_delta, pend_gate, ml, ml + dim, B, mod_stride,
                                       d.tokens, nullptr, nullptr, bt.xn1, DropCfg{}, stream, bt.xa, bt.sa, nullptr, nullptr, 1,
                                       ride ? adc : DropCfg{}, ride ? bt.maskw : nullptr));
        TRY(gemm(bt.xn1, dim, bw.qkv_w, dim, bw.qkv_b, bt.qkv, 3 * dim, M, 3 * dim, dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));
        TRY(bsi_attention_fwd_train(bt.qkv, 3 * dim, B, d.tokens, d.heads, 64, bt.ao, dim, bt.lse,
                                    adc, stream, bt.maskw, ride));
        TRY(gemm(bt.ao, dim, bw.out_w, dim, bw.out_b, bt.d1, dim, M, dim, dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));
        TRY(bsi_resid_ln_modulate_drop(bt.xa, M, dim, 1e-5f, bt.d1, ml + 2 * dim, ml + 3 * dim, ml + 4 * dim, B, mod_stride,
                                       d.tokens, nullptr, nullptr, bt.xn2, make_drop(dropout_p, seed, 2 * l + 1), stream, bt.xb,
                                       bt.sb));
        TRY(gemm(bt.xn2, dim, bw.fc1_w, dim, bw.fc1_b, bt.h, 4 * dim, M, 4 * dim, dim, BSI_EPI_BIAS_GELU_DUAL, nullptr, bt.hp, nullptr, 0, stream));
        TRY(gemm(bt.h, 4 * dim, bw.fc2_w, 4 * dim, bw.fc2_b, bt.d2, dim, M, dim, 4 * dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));
        pend_delta = bt.d2;
        pend_gate = ml + 5 * dim;
        x_prev = bt.xb;
    }
    if (gates) TRY(pass_gate(gates[d.depth + 1], s));
    // materialise the final residual stream (kept for the decoder's backward), then the decoder
    TRY(bsi_resid_ln_modulate_drop(x_prev, M, dim, 1e-5f, pend_delta, pend_gate, nullptr, nullptr, B, mod_stride, d.tokens, nullptr,
                                   nullptr, nullptr, DropCfg{}, stream, tp.x, nullptr));
    return bsi_dit_final_launch(tp.x, M, dim, d.P, w->dec_ln_w, w->dec_ln_b, w->dec_w, w->dec_b, cfg->C, cfg->H, cfg->W,
                                cfg->patch, mu, c_skip, c_out, 1, nullptr, nullptr, 1, 0, out, s);
}

extern "C" int bsi_dit_backward(const bsi_dit_config* cfg, const bsi_dit_weights* w, const bsi_dit_weights_t* wT,
                                const bsi_dit_grads* g, int B, const float* g_out, const float* c_out, void* tape_mem,
                                void* workspace, float dropout_p, unsigned long long seed, bsi_stream_t stream) {
    BSI_CHECK_ARG(cfg && w && w->blocks && wT && wT->blocks && g && g->blocks && g_out && tape_mem && workspace && B > 0,
                  "bsi_dit_backward: bad args");
    BSI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bsi_dit_backward: dropout probability %g outside [0, 1)", (double)dropout_p);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Dims d = dims_of(cfg, B);
    const int dim = d.dim, M = (int)d.M;
    const int mod_stride = d.depth * 6 * dim;
    Tape tp = carve_tape(d, B, tape_mem);
    BwdWs ws = carve_bwd(d, B, workspace);

    // modulation gradients: every 64-token slab of an image stores its sums into its own plane (bsi_ln_gate_bwd_drop with a plane
    // stride), the planes are added in fixed order when the block's adaLN backward needs them: no atomics, no zero fill, bit
    // reproducible.  Two blocks are in flight at a time (the fused kernel of block l also writes g_m of block l - 1): parity slots.
    BSI_CHECK_ARG(d.tokens % 64 == 0, "bsi_dit_backward: tokens=%d must be a multiple of 64", d.tokens);
    const int nplanes = d.tokens / 64;
    const size_t plane = (size_t)B * 6 * dim;  // floats per plane
    const int dstride = 6 * dim;               // row stride inside a plane
    auto dmod_of = [&](int l) { return ws.dmod + (size_t)(l & 1) * nplanes * plane; };
    // Bias gradients of out-projection, fc2 and fc1: the kernels that WRITE those layers' output gradients (the LayerNorm / gate backward
    // for dd1 and dd2, the GELU'-epilogue GEMM for dhp) also leave per-slab column sums; one small reduction per block adds the slabs in
    // fixed order.  The weight-gradient GEMMs then run without the bias rider (gemm_tn.hip: -9..-14 % per launch).  The qkv bias follows
    // where the single-sweep attention backward runs (per-image sums of the dqkv rows it stores); the adaLN biases stay fused.
    static const bool fused_bias = [] { const char* e = getenv("BSI_TRAIN_FUSED_BIAS"); return e && *e == '1'; }();
    const bool rel_dd = !fused_bias && dim > 256;
    const bool rel_fc1 = !fused_bias && bsi_gemm_emits_colsum(M, dim);
    const int r64 = M / 64, r128 = (M + 127) / 128;
    auto rel_qkv_of = [&](int l) {  // depends on the block only through its dropout site (on or off for all)
        return !fused_bias && bsi_attention_bwd_emits_bias(d.tokens, 64, make_drop(dropout_p, seed, 2 * l), block_tape(tp, d, B, l).maskw);
    };
    // decoder: dX = d/dx_final, parameter gradients of patch_decoder (dit.py:163-165)
    {
        const int Pp = (d.P + 7) / 8 * 8;  // yb -> ws.dd, dYb -> ws.dsmall (Pp <= 64 <= dim columns)
        BSI_CHECK_ARG(Pp <= dim, "bsi_dit_backward: decoder width %d exceeds dim %d", Pp, dim);
        TRY(bsi_dit_final_bwd_launch(tp.x, M, dim, d.P, w->dec_ln_w, w->dec_ln_b, w->dec_w, cfg->C, cfg->H, cfg->W,
                                     cfg->patch, g_out, c_out, 1, ws.dX, ws.dd, ws.dsmall, g->dec_b, g->dec_ln_w,
                                     g->dec_ln_b, ws.fin_parts, s));
        TRY(bsi_gemm_tn_bf16(ws.dsmall, Pp, ws.dd, dim, M, Pp, dim, ws.decw, dim, 0, ws.tn, stream));
        if (hipMemcpyAsync(g->dec_w, ws.decw, (size_t)d.P * dim * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            bsi_set_error("bsi_dit_backward: decoder gradient copy failed");
            return BSI_ELAUNCH;
        }
    }

    {   // top of the stack: x_final = xb + g_m * d2 of the last block
        const int l = d.depth - 1;
        BlockTape bt = block_tape(tp, d, B, l);
        TRY(bsi_ln_gate_bwd_drop(nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, ws.dX, bt.d2,
                                 tp.mod + (size_t)l * 6 * dim + 5 * dim, mod_stride, dmod_of(l) + 5 * dim, dstride, ws.dd, M, dim,
                                 d.tokens, DropCfg{}, stream, plane, rel_dd ? ws.rows_fc2 : nullptr));
    }
    for (int l = d.depth - 1; l >= 0; --l) {
        const bsi_dit_block_weights& bw = w->blocks[l];
        const bsi_dit_block_weights_t& bT = wT->blocks[l];
        const bsi_dit_block_grads& bg = g->blocks[l];
        BlockTape bt = block_tape(tp, d, B, l);
        const float* ml = tp.mod + (size_t)l * 6 * dim;
        float* dml = dmod_of(l);
        (void)bw;
        // ---- MLP branch: x2 = xb + g_m * d2.  ws.dd = g_m * dX was produced by the fused kernel of the block above (or by the
        //      gate-only launch in front of the loop).  dh = dd2 . W2, times gelu'(hp)  -> dhp
        TRY(gemm(ws.dd, dim, bT.fc2_wT, dim, nullptr, ws.dbig, 4 * dim, M, 4 * dim, dim, BSI_EPI_MUL_GELUGRAD_BF16, bt.hp, nullptr, nullptr, 0, stream,
                 rel_fc1 ? ws.rows_fc1 : nullptr));
        if (rel_dd) TRY(bsi_gemm_tn_bf16(ws.dd, dim, bt.h, 4 * dim, M, dim, 4 * dim, bg.fc2_w, 4 * dim, 0, ws.tn, stream));
        else TRY(bsi_gemm_tn_bias_bf16(ws.dd, dim, bt.h, 4 * dim, M, dim, 4 * dim, bg.fc2_w, 4 * dim, bg.fc2_b, 0, ws.tn, stream));
        // dxn2 = dhp . W1
        TRY(gemm(ws.dbig, 4 * dim, bT.fc1_wT, 4 * dim, nullptr, ws.dsmall, dim, M, dim, 4 * dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));
        if (rel_fc1) TRY(bsi_gemm_tn_bf16(ws.dbig, 4 * dim, bt.xn2, dim, M, 4 * dim, dim, bg.fc1_w, dim, 0, ws.tn, stream));
        else TRY(bsi_gemm_tn_bias_bf16(ws.dbig, 4 * dim, bt.xn2, dim, M, 4 * dim, dim, bg.fc1_w, dim, bg.fc1_b, 0, ws.tn, stream));
        // LayerNorm 2 backward (dX becomes dL/dxb) + attention branch xb = xa + g_a * d1: dd = g_a * dX, dg_a
        TRY(bsi_ln_gate_bwd_drop(ws.dsmall, bt.xb, bt.sb, ml + 4 * dim, mod_stride, dml + 3 * dim, dml + 4 * dim, dstride, ws.dX,
                                 bt.d1, ml + 2 * dim, mod_stride, dml + 2 * dim, dstride, ws.dd, M, dim, d.tokens,
                                 make_drop(dropout_p, seed, 2 * l + 1), stream, plane, rel_dd ? ws.rows_out : nullptr));
        TRY(gemm(ws.dd, dim, bT.out_wT, dim, nullptr, ws.dsmall, dim, M, dim, dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));  // dao
        const bool rel_qkv = rel_qkv_of(l);
        // The out-projection's weight gradient (16 output tiles) waits for the qkv projection's (48) when neither carries a bias rider:
        // one launch of 64 tiles x 4 token ranges instead of 16 x 15 and 48 x 5 (ws.dd and the tape's ao live until LayerNorm 1 below)
        static const bool no_pair = [] { const char* e = getenv("BSI_TRAIN_NO_TN_PAIR"); return e && *e == '1'; }();
        const bool pair = rel_dd && rel_qkv && !no_pair && dim % 256 == 0;
        if (pair) {}
        else if (rel_dd) TRY(bsi_gemm_tn_bf16(ws.dd, dim, bt.ao, dim, M, dim, dim, bg.out_w, dim, 0, ws.tn, stream));
        else TRY(bsi_gemm_tn_bias_bf16(ws.dd, dim, bt.ao, dim, M, dim, dim, bg.out_w, dim, bg.out_b, 0, ws.tn, stream));
        TRY(bsi_attention_bwd_drop(bt.qkv, 3 * dim, bt.ao, ws.dsmall, dim, bt.lse, B, d.tokens, d.heads, 64, ws.dbig, 3 * dim,
                                   make_drop(dropout_p, seed, 2 * l), stream, bt.maskw, rel_qkv ? ws.rows_qkv : nullptr));
        if (rel_dd || rel_fc1 || rel_qkv) {  // the slab tables of this block are complete (rows_fc2 is rewritten by the LayerNorm-1 launch below)
            bsi_colsum_job jobs[4];
            int nj = 0;
            if (rel_dd) {
                jobs[nj++] = bsi_colsum_job{ws.rows_fc2, r64, dim, dim, bg.fc2_b};
                jobs[nj++] = bsi_colsum_job{ws.rows_out, r64, dim, dim, bg.out_b};
            }
            if (rel_fc1) jobs[nj++] = bsi_colsum_job{ws.rows_fc1, r128, 4 * dim, 4 * dim, bg.fc1_b};
            if (rel_qkv) jobs[nj++] = bsi_colsum_job{ws.rows_qkv, B, 3 * dim, 3 * dim, bg.qkv_b};
            TRY(bsi_colsum_rows_f32(jobs, nj, ws.rows_scratch, stream));
        }
        TRY(gemm(ws.dbig, 3 * dim, bT.qkv_wT, 3 * dim, nullptr, ws.dsmall, dim, M, dim, 3 * dim, BSI_EPI_BIAS_BF16, nullptr, nullptr, nullptr, 0, stream));  // dxn1
        if (pair) TRY(bsi_gemm_tn_pair_bf16(ws.dbig, 3 * dim, bt.xn1, 3 * dim, bg.qkv_w, ws.dd, dim, bt.ao, dim, bg.out_w, dim, M, dim, ws.tn, stream));
        else if (rel_qkv) TRY(bsi_gemm_tn_bf16(ws.dbig, 3 * dim, bt.xn1, dim, M, 3 * dim, dim, bg.qkv_w, dim, 0, ws.tn, stream));
        else TRY(bsi_gemm_tn_bias_bf16(ws.dbig, 3 * dim, bt.xn1, dim, M, 3 * dim, dim, bg.qkv_w, dim, bg.qkv_b, 0, ws.tn, stream));
        // LayerNorm 1 backward (dX becomes dL/dxa) + the MLP branch of the block below: xa = xb' + g_m' * d2'
        if (l > 0) {
            BlockTape below = block_tape(tp, d, B, l - 1);
            const float* mlb = tp.mod + (size_t)(l - 1) * 6 * dim;
            float* dmlb = dmod_of(l - 1);
            TRY(bsi_ln_gate_bwd_drop(ws.dsmall, bt.xa, bt.sa, ml + dim, mod_stride, dml, dml + dim, dstride, ws.dX, below.d2,
                                     mlb + 5 * dim, mod_stride, dmlb + 5 * dim, dstride, ws.dd, M, dim, d.tokens, DropCfg{}, stream, plane,
                                     rel_dd ? ws.rows_fc2 : nullptr));
        } else {
            TRY(bsi_ln_gate_bwd_drop(ws.dsmall, bt.xa, bt.sa, ml + dim, mod_stride, dml, dml + dim, dstride, ws.dX, nullptr, nullptr,
                                     0, nullptr, 0, nullptr, M, dim, d.tokens, DropCfg{}, stream, plane));
        }
        {   // adaLN MLP of this block (dit.py:77-81): mod_l = W2 silu(W1 c + b1) + b2, rows = samples
            const float* pre = tp.ada_pre + (size_t)l * B * dim;
            const char* sl = tp.ada_s + (size_t)l * B * dim * 2;
            // bf16 dmod[:, l, :] = the slab planes of this block summed in fixed order
            TRY(bsi_sum_cast_rows_bf16(dml, nplanes, plane, dstride, B, 6 * dim, ws.dmod_bf, 6 * dim, stream));
            TRY(bsi_gemm_tn_bias_bf16(ws.dmod_bf, 6 * dim, sl, dim, B, 6 * dim, dim, bg.ada2_w, dim, bg.ada2_b, 0, ws.tn, stream));
            TRY(gemm_rows(ws.dmod_bf, 6 * dim, bT.ada2_wT, 6 * dim, nullptr, ws.ds, dim, B, dim, 6 * dim, ws.tn,
                          bsi_gemm_tn_workspace_bytes((int)M, 4 * dim, dim), stream));  // the weight-gradient slabs are free between its launches
            TRY(bsi_silu_bwd_bf16(ws.ds, pre, (size_t)B * dim, ws.dpre_bf, stream));
            TRY(bsi_gemm_tn_bias_bf16(ws.dpre_bf, dim, tp.emb, dim, B, dim, dim, bg.ada0_w, dim, bg.ada0_b, 0, ws.tn, stream));
        }
        if (g_block_events && l < g_block_events_n && g_block_events[l]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(g_block_events[l]), s);
    }
    // patch encoder (dit.py:154,178): x0 = A0 . Wenc^T + b + pos
    TRY(bsi_silu_bwd_bf16(ws.dX, nullptr, (size_t)M * dim, ws.dd, stream));  // bf16 copy of dX
    TRY(bsi_gemm_tn_bias_bf16(ws.dd, dim, tp.a0, d.kpad, M, dim, d.kpad, g->enc_w_padded, d.kpad, g->enc_b, 0, ws.tn, stream));

    return BSI_OK;
}
