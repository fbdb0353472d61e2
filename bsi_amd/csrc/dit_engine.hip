// Host-side sequencing of one DenoisingDiT evaluation (bsi/models/dit.py:174-181,225-233 of the
// reference) on a HIP stream.  No allocation, no synchronisation: the caller owns the workspace.
#include <functional>
#include <vector>

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
    // enough for one chain over B images and for bsi_dit_forward_pair's two half-batch chains
    const size_t one = carve(cfg, B, nullptr).total;
    const size_t two = B >= 2 ? carve(cfg, (B + 1) / 2, nullptr).total + carve(cfg, B / 2, nullptr).total : 0;
    return one > two ? one : two;
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

// ---- one evaluation as a CHAIN of launches -------------------------------------------------------------------------------------
// bsi_dit_forward enqueues the chain on one stream.  bsi_dit_forward_pair splits the batch into two halves with a chain each and
// interleaves them over two CU-masked streams: the matrix-pipe-bound launches (class G: the GEMMs, and by default attention) on the
// large partition, the HBM-bound ones (class H: prologue, LayerNorm passes, final kernel) of the OTHER half on the small one.
namespace {

enum { SC_G = 0, SC_H = 1 };

struct Op {
    int sc;    // stream class
    int prof;  // BSI_PROF_* class
    std::function<int(hipStream_t)> run;
};

struct ChainArgs {
    const bsi_dit_config* cfg;
    const bsi_dit_weights* w;
    int B;
    const float* mu;
    const float* mod;
    int mod_rows;
    const float *c_in, *c_skip, *c_out;
    int coef_stride;
    float* out;
    void* workspace;
    float* tokens_out;
    bool attn_on_h;
};

// The launches of dit.py:174-181,225-233 for one (half) batch, in dependency order: every op depends on the one before it.
void build_chain(const ChainArgs& a, std::vector<Op>& ops) {
    const bsi_dit_config* cfg = a.cfg;
    const bsi_dit_weights* w = a.w;
    const DitDims d = dims_of(cfg);
    const int dim = cfg->dim, B = a.B;
    const int M = B * d.tokens;
    const int mod_rows = a.mod_rows, mod_stride = cfg->depth * 6 * dim;
    const Workspace ws = carve(cfg, B, a.workspace);
    const float* mod = a.mod;

    // 1. c_in*mu -> Fourier features -> patchify -> bf16 tokens
    ops.push_back({SC_H, BSI_PROF_PROLOGUE, [=](hipStream_t s) {
                       return bsi_dit_prologue_launch(a.mu, a.c_in, a.coef_stride, B, cfg->C, cfg->H, cfg->W, cfg->patch, cfg->ff_nmin,
                                                      d.nfreq, d.kpad, ws.a0, s);
                   }});
    // 2. patch encoder + positional embedding (dit.py:178)
    ops.push_back({SC_G, BSI_PROF_GEMM_ENC, [=](hipStream_t s) {
                       bsi_gemm_args g{};
                       g.A = ws.a0; g.W = w->enc_w; g.bias = w->enc_b; g.out = ws.x;
                       g.M = M; g.N = dim; g.K = d.kpad; g.lda = d.kpad; g.ldw = d.kpad; g.ldo = dim;
                       g.epilogue = BSI_EPI_BIAS_POS_F32; g.pos = w->pos; g.tokens = d.tokens;
                       return bsi_gemm_bf16(&g, reinterpret_cast<bsi_stream_t>(s));
                   }});
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
        const bsi_dit_block_weights bw = w->blocks[l];
        const float* ml = mod + (size_t)l * 6 * dim;
        {
            const void *pd0 = pend_delta0, *pd = pend_delta;
            const float *pg0 = pend_gate0, *pg = pend_gate;
            ops.push_back({SC_H, BSI_PROF_LN, [=](hipStream_t s) {
                               return bsi_resid2_ln_modulate(ws.x, M, dim, 1e-5f, pd0, pg0, pd, pg, 1, ml, ml + dim, mod_rows, mod_stride,
                                                             d.tokens, ws.xn, reinterpret_cast<bsi_stream_t>(s));
                           }});
        }
        ops.push_back({SC_G, BSI_PROF_GEMM_QKV, [=](hipStream_t s) {
                           bsi_gemm_args g{};
                           g.A = ws.xn; g.W = bw.qkv_w; g.bias = bw.qkv_b; g.out = ws.big;
                           g.M = M; g.N = 3 * dim; g.K = dim; g.lda = dim; g.ldw = dim; g.ldo = 3 * dim;
                           g.epilogue = BSI_EPI_BIAS_BF16;
                           return bsi_gemm_bf16_ws(&g, ws.splitk, ws.splitk_bytes, reinterpret_cast<bsi_stream_t>(s));
                       }});
        ops.push_back({a.attn_on_h ? SC_H : SC_G, BSI_PROF_ATTN, [=](hipStream_t s) {
                           return bsi_attention_fwd(ws.big, 3 * dim, B, d.tokens, cfg->heads, dim / cfg->heads, ws.xn, dim,
                                                    reinterpret_cast<bsi_stream_t>(s));
                       }});
        ops.push_back({SC_G, BSI_PROF_GEMM_OUT, [=](hipStream_t s) {  // attention output projection -> delta
                           bsi_gemm_args go{};
                           go.A = ws.xn; go.W = bw.out_w; go.bias = bw.out_b; go.out = ws.da;
                           go.M = M; go.N = dim; go.K = dim; go.lda = dim; go.ldw = dim; go.ldo = dim;
                           go.epilogue = BSI_EPI_BIAS_BF16;
                           return bsi_gemm_bf16_ws(&go, ws.splitk, ws.splitk_bytes, reinterpret_cast<bsi_stream_t>(s));
                       }});
        const bool lazy = !eager_resid && l + 1 < cfg->depth;  // the last block stores: the final kernel takes one pending update
        // x + gate_msa * delta (stored unless lazy); xn = LN(that) * (1 + scale_mlp) + shift_mlp
        ops.push_back({SC_H, BSI_PROF_LN, [=](hipStream_t s) {
                           return bsi_resid2_ln_modulate(ws.x, M, dim, 1e-5f, nullptr, nullptr, ws.da, ml + 2 * dim, lazy ? 0 : 1,
                                                         ml + 3 * dim, ml + 4 * dim, mod_rows, mod_stride, d.tokens, ws.xn,
                                                         reinterpret_cast<bsi_stream_t>(s));
                       }});
        pend_delta0 = lazy ? ws.da : nullptr;
        pend_gate0 = lazy ? ml + 2 * dim : nullptr;
        ops.push_back({SC_G, BSI_PROF_GEMM_FC1, [=](hipStream_t s) {
                           bsi_gemm_args g1{};
                           g1.A = ws.xn; g1.W = bw.fc1_w; g1.bias = bw.fc1_b; g1.out = ws.big;
                           g1.M = M; g1.N = 4 * dim; g1.K = dim; g1.lda = dim; g1.ldw = dim; g1.ldo = 4 * dim;
                           g1.epilogue = BSI_EPI_BIAS_GELU_BF16;
                           return bsi_gemm_bf16_ws(&g1, ws.splitk, ws.splitk_bytes, reinterpret_cast<bsi_stream_t>(s));
                       }});
        ops.push_back({SC_G, BSI_PROF_GEMM_FC2, [=](hipStream_t s) {  // MLP output -> delta (in the now dead xn buffer)
                           bsi_gemm_args g2{};
                           g2.A = ws.big; g2.W = bw.fc2_w; g2.bias = bw.fc2_b; g2.out = ws.xn;
                           g2.M = M; g2.N = dim; g2.K = 4 * dim; g2.lda = 4 * dim; g2.ldw = 4 * dim; g2.ldo = dim;
                           g2.epilogue = BSI_EPI_BIAS_BF16;
                           return bsi_gemm_bf16_ws(&g2, ws.splitk, ws.splitk_bytes, reinterpret_cast<bsi_stream_t>(s));
                       }});
        pend_delta = ws.xn;
        pend_gate = ml + 5 * dim;
    }
    if (a.tokens_out) {  // tests: materialise the final residual stream
        const void* pd = pend_delta;
        const float* pg = pend_gate;
        float* tokens_out = a.tokens_out;
        ops.push_back({SC_H, BSI_PROF_LN, [=](hipStream_t s) {
                           if (int rc = bsi_resid_ln_modulate(ws.x, M, dim, 1e-5f, pd, pg, nullptr, nullptr, mod_rows, mod_stride, d.tokens,
                                                              nullptr, nullptr, nullptr, reinterpret_cast<bsi_stream_t>(s)))
                               return rc;
                           hipError_t e = hipMemcpyAsync(tokens_out, ws.x, (size_t)M * dim * sizeof(float), hipMemcpyDeviceToDevice, s);
                           if (e != hipSuccess) {
                               bsi_set_error("bsi_dit_forward: tokens copy failed: %s", hipGetErrorString(e));
                               return (int)BSI_ELAUNCH;
                           }
                           return (int)BSI_OK;
                       }});
        pend_delta = nullptr;
        pend_gate = nullptr;
    }
    // 4. LayerNorm + Linear + unpatchify (+ x_hat = c_skip*mu + c_out*f)
    {
        const void* pd = pend_delta;
        const float* pg = pend_gate;
        ops.push_back({SC_H, BSI_PROF_FINAL, [=](hipStream_t s) {
                           return bsi_dit_final_launch(ws.x, M, dim, d.P, w->dec_ln_w, w->dec_ln_b, w->dec_w, w->dec_b, cfg->C, cfg->H,
                                                       cfg->W, cfg->patch, a.mu, a.c_skip, a.c_out, a.coef_stride, pd, pg, mod_rows,
                                                       mod_stride, a.out, s);
                       }});
    }
}

inline int run_op(const Op& op, hipStream_t s) {
    ProfScope prof(op.prof, s);
    return op.run(s);
}

int check_forward_args(const bsi_dit_config* cfg, const bsi_dit_weights* w, int B, const float* mu, const float* mod, int mod_rows,
                       const float* c_in, const float* c_skip, const float* c_out, const float* out, const void* workspace) {
    if (int rc = check_cfg(cfg)) return rc;
    BSI_CHECK_ARG(w && w->blocks && mu && mod && out && workspace && B > 0, "bsi_dit_forward: bad args");
    BSI_CHECK_ARG(mod_rows == 1 || mod_rows == B, "bsi_dit_forward: mod_rows=%d must be 1 or B=%d", mod_rows, B);
    BSI_CHECK_ARG((c_in == nullptr) == (c_skip == nullptr) && (c_in == nullptr) == (c_out == nullptr),
                  "bsi_dit_forward: c_in/c_skip/c_out must be given together");
    return BSI_OK;
}

}  // namespace

extern "C" int bsi_dit_forward(const bsi_dit_config* cfg, const bsi_dit_weights* w, int B, const float* mu,
                               const float* mod, int mod_rows, const float* c_in, const float* c_skip,
                               const float* c_out, int coef_stride, float* out, void* workspace, float* tokens_out,
                               bsi_stream_t stream) {
    if (int rc = check_forward_args(cfg, w, B, mu, mod, mod_rows, c_in, c_skip, c_out, out, workspace)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    std::vector<Op> ops;
    ops.reserve(8 + 7 * (size_t)cfg->depth);
    build_chain(ChainArgs{cfg, w, B, mu, mod, mod_rows, c_in, c_skip, c_out, coef_stride, out, workspace, tokens_out, false}, ops);
    for (const Op& op : ops)
        if (int rc = run_op(op, s)) return rc;
    return BSI_OK;
}

// ---- CU-partitioned stream pair ---------------------------------------------------------------------------------------------------
struct bsi_cu_pair {
    hipStream_t g = nullptr, h = nullptr;
    int h_cus = 0, dev = 0;
    std::vector<hipEvent_t> ev;  // ring of timing-less events for the hand-overs between the two streams
    size_t next = 0;
    hipEvent_t take() {
        hipEvent_t e = ev[next];
        next = (next + 1) % ev.size();
        return e;
    }
};

extern "C" int bsi_cu_pair_create(int h_cus, bsi_cu_pair** out) {
    BSI_CHECK_ARG(out, "bsi_cu_pair_create: null output");
    *out = nullptr;
    const int all = device_cus();
    BSI_CHECK_ARG(h_cus >= 8 && h_cus % 8 == 0 && h_cus <= BSI_MAX_CU_RESERVE && h_cus < all,
                  "bsi_cu_pair_create: h_cus=%d must be a multiple of 8 in [8, %d] (CU mask bits are dealt to the 8 XCDs round robin)", h_cus,
                  BSI_MAX_CU_RESERVE);
    // Bit i of a queue's CU mask is CU (i / 8) of XCD (i % 8) (the driver deals the bits to the XCCs round robin), so the low
    // all - h_cus bits are an XCD-balanced set of (all - h_cus) / 8 CUs per XCD and the remaining bits the other h_cus / 8 per XCD.
    const int words = (all + 31) / 32;
    std::vector<uint32_t> mg(words, 0u), mh(words, 0u);
    for (int i = 0; i < all; ++i) (i < all - h_cus ? mg : mh)[i / 32] |= 1u << (i % 32);
    bsi_cu_pair* p = new bsi_cu_pair;
    p->h_cus = h_cus;
    (void)hipGetDevice(&p->dev);
    hipError_t e = hipExtStreamCreateWithCUMask(&p->g, (uint32_t)words, mg.data());
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&p->h, (uint32_t)words, mh.data());
    if (e != hipSuccess) {
        bsi_set_error("bsi_cu_pair_create: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        if (p->g) (void)hipStreamDestroy(p->g);
        delete p;
        return BSI_ELAUNCH;
    }
    p->ev.assign(2048, nullptr);
    for (hipEvent_t& ev : p->ev) {
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            bsi_set_error("bsi_cu_pair_create: hipEventCreate failed");
            (void)bsi_cu_pair_destroy(p);
            return BSI_ELAUNCH;
        }
    }
    *out = p;
    return BSI_OK;
}

extern "C" int bsi_cu_pair_destroy(bsi_cu_pair* p) {
    if (!p) return BSI_OK;
    (void)hipStreamSynchronize(p->g);
    (void)hipStreamSynchronize(p->h);
    for (hipEvent_t e : p->ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(p->g);
    (void)hipStreamDestroy(p->h);
    delete p;
    return BSI_OK;
}

extern "C" int bsi_cu_pair_streams(const bsi_cu_pair* p, bsi_stream_t* g, bsi_stream_t* h, int* h_cus) {
    BSI_CHECK_ARG(p, "bsi_cu_pair_streams: null pair");
    if (g) *g = reinterpret_cast<bsi_stream_t>(p->g);
    if (h) *h = reinterpret_cast<bsi_stream_t>(p->h);
    if (h_cus) *h_cus = p->h_cus;
    return BSI_OK;
}

extern "C" int bsi_dit_forward_pair(const bsi_dit_config* cfg, const bsi_dit_weights* w, int B, const float* mu,
                                    const float* mod, int mod_rows, const float* c_in, const float* c_skip,
                                    const float* c_out, int coef_stride, float* out, void* workspace, bsi_cu_pair* pair, int flags,
                                    bsi_stream_t stream) {
    if (int rc = check_forward_args(cfg, w, B, mu, mod, mod_rows, c_in, c_skip, c_out, out, workspace)) return rc;
    BSI_CHECK_ARG(pair, "bsi_dit_forward_pair: null pair");
    BSI_CHECK_ARG(B >= 2, "bsi_dit_forward_pair: B=%d cannot be split", B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipStream_t st[2] = {pair->g, pair->h};
    const DitDims d = dims_of(cfg);
    const size_t D = (size_t)cfg->C * cfg->H * cfg->W;
    const int Bh[2] = {(B + 1) / 2, B / 2};
    // the two halves' chains: images [0, Bh0) and [Bh0, B), each with a workspace of its own inside the caller's
    std::vector<Op> ch[2];
    const size_t ws0 = carve(cfg, Bh[0], nullptr).total;
    for (int h = 0; h < 2; ++h) {
        const size_t b0 = h ? (size_t)Bh[0] : 0;
        ChainArgs a{cfg, w, Bh[h], mu + b0 * D, mod_rows == 1 ? mod : mod + b0 * (size_t)cfg->depth * 6 * cfg->dim, mod_rows == 1 ? 1 : Bh[h],
                    c_in ? c_in + b0 * coef_stride : nullptr, c_skip ? c_skip + b0 * coef_stride : nullptr,
                    c_out ? c_out + b0 * coef_stride : nullptr, coef_stride, out + b0 * D,
                    reinterpret_cast<char*>(workspace) + (h ? ws0 : 0), nullptr, (flags & BSI_PAIR_ATTN_ON_H) != 0};
        ch[h].reserve(8 + 7 * (size_t)cfg->depth);
        build_chain(a, ch[h]);
    }
    (void)d;
    // The persistent kernels size their grids with compute_cus(): G has all CUs but the pair's H share, as a hard limit (CU mask).
    // The LayerNorm passes run as persistent kernels sized for H (BSI_PAIR_LN_CLASSIC=1: the one-row-per-wave kernel, for A/B runs).
    const bool ln_classic = getenv("BSI_PAIR_LN_CLASSIC") != nullptr;
    const int saved_reserve = g_bsi_cu_reserve, saved_masked = g_bsi_cu_masked, saved_ln = g_bsi_ln_stream_cus;
    g_bsi_cu_reserve = pair->h_cus;
    g_bsi_cu_masked = 1;
    if (!ln_classic) g_bsi_ln_stream_cus = pair->h_cus;
    struct Restore {
        int r, m, l;
        ~Restore() { g_bsi_cu_reserve = r; g_bsi_cu_masked = m; g_bsi_ln_stream_cus = l; }
    } restore{saved_reserve, saved_masked, saved_ln};

#define PAIR_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            bsi_set_error("bsi_dit_forward_pair: %s: %s", #call, hipGetErrorString(e__));       \
            return BSI_ELAUNCH;                                                                 \
        }                                                                                       \
    } while (0)
    // fork: both streams start behind everything already enqueued on the caller's stream
    {
        hipEvent_t e = pair->take();
        PAIR_HIP(hipEventRecord(e, s));
        PAIR_HIP(hipStreamWaitEvent(pair->g, e, 0));
        PAIR_HIP(hipStreamWaitEvent(pair->h, e, 0));
    }
    // Interleave the chains SEGMENT by segment (a segment = an H-class op and the G-class ops behind it): A1 B1 A2 B2 ...  On G
    // that is [qkv attn out]A [qkv attn out]B [fc1 fc2]A [fc1 fc2]B, on H LN1A LN1B LN2A LN2B: while G works on one half's segment, H
    // prepares the other half's next one.  Hand-overs inside a chain are events recorded right behind the producing launch.
    size_t pos[2] = {0, 0};
    int last_sc[2] = {-1, -1};
    hipEvent_t pending[2] = {nullptr, nullptr};  // recorded behind the chain's last launch when the next one runs on the other stream
    auto segment = [&](int h) -> int {
        std::vector<Op>& ops = ch[h];
        bool first = true;
        while (pos[h] < ops.size()) {
            const Op& op = ops[pos[h]];
            if (!first && op.sc == SC_H && last_sc[h] == SC_G) break;  // the next segment starts here
            first = false;
            if (pending[h]) {
                PAIR_HIP(hipStreamWaitEvent(st[op.sc], pending[h], 0));
                pending[h] = nullptr;
            }
            if (int rc = run_op(op, st[op.sc])) return rc;
            last_sc[h] = op.sc;
            ++pos[h];
            const bool more = pos[h] < ops.size();
            if (!more || ops[pos[h]].sc != op.sc) {  // hand-over (or the chain's end: the join below waits for it)
                pending[h] = pair->take();
                PAIR_HIP(hipEventRecord(pending[h], st[op.sc]));
            }
        }
        return BSI_OK;
    };
    int rc_all = BSI_OK;
    while (rc_all == BSI_OK && (pos[0] < ch[0].size() || pos[1] < ch[1].size())) {
        for (int h = 0; h < 2 && rc_all == BSI_OK; ++h)
            if (pos[h] < ch[h].size()) rc_all = segment(h);
    }
    // join: the caller's stream continues behind both chains -- also when a launch failed half way (whatever was enqueued on G and H
    // still reads and writes the caller's buffers; the caller must not get ahead of it)
    for (int h = 0; h < 2; ++h) {
        hipEvent_t e = pair->take();
        PAIR_HIP(hipEventRecord(e, st[h]));
        PAIR_HIP(hipStreamWaitEvent(s, e, 0));
    }
#undef PAIR_HIP
    return rc_all;
}
