// Internal launchers of the UNet training path shared between unet_bwd_ops.hip and the engine (not part of the C ABI).
#pragma once
#include "common.h"

int bsi_film_silu_drop(const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride, DropCfg dc,
                       void* y, bsi_stream_t stream);
int bsi_film_silu_bwd_drop(const void* dy, const void* h1, int M, int N, int HW, const float* film, int film_rows,
                           int film_stride, DropCfg dc, void* dh1, float* dfilm, int dfilm_stride, bsi_stream_t stream,
                           size_t part_stride = 0);
// reproducible forms of the C-ABI entries that accumulate with atomics (scratch from the engine's workspace, outputs WRITTEN)
int bsi_groupnorm_bwd_cast_det(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                               const float* beta, float eps, int silu, const float* add, const float* add_b, float* out1, float* out2,
                               float* dgamma, float* dbeta, void* out1_bf16, const float* stats, float* partials, bsi_stream_t stream);
size_t bsi_unet_decode_bwd_parts_floats(int M, int C, int Cout);
int bsi_unet_decode_bwd_det(const float* g_xhat, const float* c_out, int coef_stride, const float* h, int B, int HW, int C, const float* w,
                            int Cout, float* dh, float* dw, float* db, float* parts, bsi_stream_t stream);
int bsi_sum_cast_rows_bf16(const float* parts, int nparts, size_t part_stride, int row_stride, int rows, int cols, void* out, int ld_out,
                           bsi_stream_t stream);
int bsi_conv_wgrad_conv2d_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B, int H, int W,
                                    int Cin, int cin_logical, int Cin2, int Cout, int taps, float* w, float* w2, float* dbias,
                                    void* workspace, bsi_stream_t stream);

// GroupNorm of ONE 128-channel fp32 map into up to two bf16 targets (inference engine, round 4).  GroupNorm(32) over the 256
// channels of cat(x, skip) (residual_block.py:39-64 on simplified_unet.py:43-46) is two independent halves of 16 groups of 8
// channels: the skip half is normalised (with the UP block's affine, channels 128..255) by the pass that reads the skip tensor
// anyway -- the GroupNorm in front of the next DOWN block -- so the up block's pass reads only its own input.  A target is the
// column range [col0, col0 + 128) of a bf16 [M, ld] matrix; cpg = channels per group (4: GroupNorm(32) of 128 channels, 8: a half
// of the 256-channel one); raw (optional) receives the un-normalised bf16 copy (operand of the folded 1x1 skip convolution).
struct GnTarget {
    void* out;
    void* raw;
    const float* gamma;  // [128], already offset to this half's channels
    const float* beta;
    int ld, col0, cpg, silu;
};
int bsi_groupnorm_apply_split(const float* x, const float* part, int B, int HW, float eps, GnTarget first, GnTarget second /* out == NULL: none */,
                              bsi_stream_t stream);
