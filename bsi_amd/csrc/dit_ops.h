// Internal launchers shared between dit_ops.hip and the engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

int bsi_dit_prologue_launch(const float* mu, const float* c_in, int coef_stride, int B, int C, int H, int W, int ps,
                            int nmin, int nfreq, int kpad, void* out, hipStream_t s);
int bsi_dit_final_launch(const float* x, int Mtok, int d, int P, const float* ln_w, const float* ln_b,
                         const float* dec_w, const float* dec_b, int C, int H, int W, int ps, const float* mu,
                         const float* c_skip, const float* c_out, int coef_stride, const void* delta, const float* gate,
                         int gate_rows, int gate_stride, float* out, hipStream_t s);
