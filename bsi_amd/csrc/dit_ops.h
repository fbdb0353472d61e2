// Internal launchers shared between dit_ops.hip and the engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

int bsi_dit_prologue_launch(const float* mu, const float* c_in, int coef_stride, int B, int C, int H, int W, int ps,
                            int nmin, int nfreq, int kpad, void* out, hipStream_t s);
int bsi_dit_final_launch(const float* x, int Mtok, int d, int P, const float* ln_w, const float* ln_b,
                         const float* dec_w, const float* dec_b, int C, int H, int W, int ps, const float* mu,
                         const float* c_skip, const float* c_out, int coef_stride, const void* delta, const float* gate,
                         int gate_rows, int gate_stride, float* out, hipStream_t s);

#include "common.h"
int bsi_resid_ln_modulate_drop(float* x, int M, int d, float eps, const void* delta, const float* gate,
                               const float* shift, const float* scale, int mod_rows, int mod_stride, int tokens,
                               const float* ln_w, const float* ln_b, void* out_bf16, DropCfg dc, bsi_stream_t stream,
                               float* x_out = nullptr, float* stats = nullptr, const void* delta0 = nullptr,
                               const float* gate0 = nullptr, int write_x = 1, DropCfg mask_dc = DropCfg{}, void* maskw = nullptr);
int bsi_ln_gate_bwd_drop(const void* dxn, const float* x, const float* stats, const float* scale, int mod_stride, float* dshift,
                         float* dscale, int dmod_stride, float* dX, const void* delta, const float* gate, int gate_stride,
                         float* dgate, int dgate_stride, void* ddelta, int M, int d, int tokens, DropCfg dc, bsi_stream_t stream, size_t part_stride = 0,
                         float* dbias_rows = nullptr /* [M / 64][d]: column sums of ddelta per 64-row slab (bias gradient), gate part only */);
// out_bf16[r][c] = bf16(sum over nparts planes), fixed order: the reproducible counterpart of the atomics (part_stride above)
int bsi_sum_cast_rows_bf16(const float* parts, int nparts, size_t part_stride, int row_stride, int rows, int cols, void* out, int ld_out,
                           bsi_stream_t stream);
int bsi_ln_mod_bwd_drop(const void* dxn, const float* x, const float* scale, int mod_stride, float* dshift,
                        float* dscale, int dmod_stride, float* dX, int M, int d, int tokens, float eps, DropCfg dc,
                        bsi_stream_t stream);
// maskw: optional [B * heads][8 KB] dropout-mask words (256 tokens, head dim 64, dropout on: bsi_attention_uses_mask_words) that
// the forward writes and the backward reads instead of evaluating the hash again; NULL: both sides hash.
bool bsi_attention_uses_mask_words(int tokens, int dh);
// mask_ready: the words were already written (by the LayerNorm pass in front of the qkv projection, bsi_resid_ln_modulate_drop)
int bsi_attention_fwd_train(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out,
                            float* lse, DropCfg dc, bsi_stream_t stream, void* maskw = nullptr, bool mask_ready = false);
int bsi_attention_bwd_drop(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                           int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, DropCfg dc, bsi_stream_t stream,
                           const void* maskw = nullptr, float* bias_rows = nullptr);
// gemm_bf16.hip: can the MUL_GELUGRAD GEMM of this shape write bsi_gemm_args::colsum_rows?
bool bsi_gemm_emits_colsum(int M, int K);
// attention_bwd_x.hip: the single-sweep backward (256 tokens, head dim 64; dropout off, or on with the mask words)
// bias_rows (or null): [B][3 * heads * 64] fp32 per-image column sums of dqkv (the qkv bias gradient's slabs)
int bsi_attention_bwd_exchange(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse, int B,
                               int heads, void* dqkv, int ld_dqkv, DropCfg dc, const void* maskw, hipStream_t stream, float* bias_rows = nullptr);
// does bsi_attention_bwd_drop with these arguments run the kernel that can write bias_rows?
bool bsi_attention_bwd_emits_bias(int tokens, int dh, DropCfg dc, const void* maskw);
