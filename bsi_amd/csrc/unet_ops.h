// Internal launchers of the UNet training path shared between unet_bwd_ops.hip and the engine (not part of the C ABI).
#pragma once
#include "common.h"

int bsi_film_silu_drop(const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride, DropCfg dc,
                       void* y, bsi_stream_t stream);
int bsi_film_silu_bwd_drop(const void* dy, const void* h1, int M, int N, int HW, const float* film, int film_rows,
                           int film_stride, DropCfg dc, void* dh1, float* dfilm, int dfilm_stride, bsi_stream_t stream);
