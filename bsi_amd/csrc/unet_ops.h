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
