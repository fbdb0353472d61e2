/* bsi_hip.h — C ABI of the MI355X-native BSI hot path (libbsi_hip.so).
 *
 * The reference (martenlienen/bsi) has no FFI: its hot path is the Python surface
 * bsi.bsi.BSI + bsi.models.{dit,vdm_unet} executing torch ops.  This header declares the
 * native entry points the Python mirror (bsi_amd/) binds with ctypes; each entry cites the
 * reference lines it replaces.  Conventions:
 *   - plain pointers and sizes; all pointers are DEVICE pointers unless marked `host`;
 *   - fp32 tensors are `float*`, bf16 tensors are `void*` (raw bfloat16 bits);
 *   - the library allocates nothing persistent: callers pass outputs and workspaces;
 *   - every call enqueues on `stream` (a hipStream_t; NULL = the default stream) and returns
 *     immediately; no host synchronisation happens inside;
 *   - return value 0 = ok, negative = error (message via bsi_last_error()); never aborts.
 */
#ifndef BSI_HIP_H
#define BSI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bsi_stream_t; /* hipStream_t */

enum {
    BSI_OK = 0,
    BSI_EINVAL = -1,       /* bad argument / unsupported shape */
    BSI_ELAUNCH = -2,      /* HIP launch error */
    BSI_EUNSUPPORTED = -3, /* feature not built */
};

int bsi_version(void);
const char* bsi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * BSI algorithm wrapper — bsi/bsi.py
 * ---------------------------------------------------------------------------------------- */

/* LogUniform constants held by BSI.p_lambda (bsi.py:67-84,135): ln_low = log(float(lambda_0)),
 * delta = log(float(lambda_0 + alpha_M)) - ln_low, both Python doubles in the reference and cast
 * to fp32 when they meet an fp32 tensor. */
typedef struct bsi_params {
    float lambda_0;
    float alpha_M;
    float alpha_R;
    float ln_low; /* (float)ln_low  */
    float delta;  /* (float)delta   */
} bsi_params;

/* bsi.py:390-403 (_edm_preconditioning) with bsi.py:83-84 (icdf): for every t[i]
 *   lam = exp(delta*t + ln_low); alpha = lam - lambda_0; kappa = 1 + alpha*(alpha/lam);
 *   c_skip = alpha/kappa; c_out = rsqrt(kappa); c_in = sqrt(lam/kappa).
 * Any output pointer may be NULL. */
int bsi_edm_coeffs(const bsi_params* p /*host*/, const float* t, int n, float* lam, float* c_skip,
                   float* c_out, float* c_in, bsi_stream_t stream);

/* bsi.py:80-81 / 76-78: t = cdf(lam) = (log(lam) - ln_low)/delta and w = lam*delta (reciprocal pdf). */
int bsi_lambda_to_t(const bsi_params* p /*host*/, const float* lam, int n, float* t, float* rpdf,
                    bsi_stream_t stream);

/* bsi.py:322-323 (sample), 352-353, 261-262: lam[i] = icdf(t[i]) (i < k1), alpha[i] = lam[i+1]-lam[i]. */
int bsi_schedule(const bsi_params* p /*host*/, const float* t, int k1, float* lam, float* alpha,
                 bsi_stream_t stream);

/* bsi.py:422-440 (_sample_lambda, low-discrepancy branch): t = (perm/(1+total) + offset) mod 1,
 * lam = icdf(t).  perm: int64[total], offset: device scalar. */
int bsi_lambda_grid(const bsi_params* p /*host*/, const int64_t* perm, const float* offset, int total,
                    float* lam, bsi_stream_t stream);

/* bsi.py:405-420 (_sample_q_mu_lambda): mu[r] = ((lam[r]-lambda_0)/lam[r]) * x[r % B] + rsqrt(lam[r]) * eps[r],
 * r < rows (rows = n_samples*B), each row D floats. */
int bsi_q_sample(const bsi_params* p /*host*/, const float* x, const float* lam, const float* eps, int rows,
                 int B, int D, float* mu, bsi_stream_t stream);

/* bsi.py:325 / 356: mu_0[r] = rsqrt(lam[0]) * eps0[r]  (start of the sampling chain; lam = bsi_schedule output). */
int bsi_sample_init(const float* eps0, const float* lam, int rows, int D, float* mu, bsi_stream_t stream);

/* bsi.py:385: out[r] = c[r*c_stride] * mu[r]  (the c_in * mu input scaling for a generic denoiser). */
int bsi_scale_rows(const float* mu, const float* c, int c_stride, int rows, int D, float* out,
                   bsi_stream_t stream);

/* bsi.py:382-386: x_hat = c_skip*mu + c_out*f (torch.addcmul). c_* indexed [r*c_stride]. */
int bsi_predict_combine(const float* mu, const float* f, const float* c_skip, const float* c_out,
                        int c_stride, int rows, int D, float* x_hat, bsi_stream_t stream);
/* backward of the above w.r.t. f (and optionally mu): g_f = c_out*g, g_mu = c_skip*g. */
int bsi_predict_combine_bwd(const float* g, const float* c_skip, const float* c_out, int c_stride, int rows,
                            int D, float* g_f, float* g_mu /*nullable*/, bsi_stream_t stream);

/* bsi.py:331-335 / 364-368: one measure/refine step of Algorithm 3 for `rows` images of D floats:
 *   x_hat = c_skip*mu + c_out*f        (skipped when f_is_xhat != 0: f already holds x_hat)
 *   y     = x_hat + rsqrt(alpha[i]) * eps
 *   mu'   = (alpha[i]*y + lam[i]*mu) / lam[i+1]
 * lam/alpha/c_skip/c_out are the device vectors of bsi_schedule / bsi_edm_coeffs over the schedule,
 * `i` the step index.  x_hat_out / y_out may be NULL (history capture of sample_history). */
int bsi_refine_step(const float* mu, const float* f, const float* eps, const float* lam, const float* alpha,
                    const float* c_skip, const float* c_out, int i, int f_is_xhat, int rows, int D,
                    float* x_hat_out, float* y_out, float* mu_next, bsi_stream_t stream);

/* The same step with the Gaussian measurement noise generated IN the kernel (opt-in, BSI.sample(device_noise=True)):
 * Philox4x32-10 keyed by the 64-bit `seed` in device memory, counter = (group of four elements, noise stream `i`), Box-Muller.
 * bsi_philox_normal writes the same stream to memory (out[4g .. 4g+3] = the four normals of group g of stream `stream_id`),
 * used for mu_0 (bsi.py:325) and by the tests: bsi_refine_step with eps = bsi_philox_normal(seed, i) is bit-identical. */
int bsi_refine_step_philox(const float* mu, const float* f, const unsigned long long* seed, const float* lam,
                           const float* alpha, const float* c_skip, const float* c_out, int i, int f_is_xhat, int rows, int D,
                           float* x_hat_out, float* y_out, float* mu_next, bsi_stream_t stream);
int bsi_philox_normal(const unsigned long long* seed, unsigned stream_id, size_t n, float* out, bsi_stream_t stream);
/* Known-answer access to that generator (it replaces ATen's generator behind torch.randn, bsi.py:325,332-334, so the reference
 * holds no vectors for it; the published ones are Random123's, tests/golden/philox_kat.json).
 * bsi_philox4x32_10: out[4b .. 4b+3] = Philox4x32-10(counter = ctr_key[6b .. 6b+3], key = ctr_key[6b+4 .. 6b+5]).
 * bsi_philox_uint32: the integer stream behind bsi_philox_normal: out[4g .. 4g+3] = Philox4x32-10(counter = (g & 0xffffffff,
 * g >> 32, stream_id, 0x42534931), key = (seed & 0xffffffff, seed >> 32)); the normals are Box-Muller on the top 24 bits:
 * u1 = ((x0 >> 8) + 1) / 2^24, u2 = (x1 >> 8) / 2^24 -> sqrt(-2 ln u1) * (cos, sin)(2 pi u2), likewise (x2, x3). */
int bsi_philox4x32_10(const unsigned* ctr_key, size_t nblocks, unsigned* out, bsi_stream_t stream);
int bsi_philox_uint32(const unsigned long long* seed, unsigned stream_id, size_t n, unsigned* out, bsi_stream_t stream);

/* bsi.py:309-310, 273-274, 288-289: out[r] = w[r] * scale * reduce_D((x[r % B] - x_hat[r])^2),
 * reduce = mean if mean != 0 else sum.  diff_out (nullable) receives x - x_hat for the backward. */
int bsi_sqerr_rows(const float* x, const float* x_hat, const float* w, float scale, int mean, int rows,
                   int B, int D, float* out, bsi_stream_t stream);
/* gradient of the above w.r.t. x_hat: g_xhat[r] = -2 * g[r] * w[r] * scale * (x - x_hat) / (mean ? D : 1). */
int bsi_sqerr_rows_bwd(const float* x, const float* x_hat, const float* w, const float* g, float scale,
                       int mean, int rows, int B, int D, float* g_xhat, bsi_stream_t stream);

/* bsi.py:230-247 (reconstruction_loss): Normal(x_hat, alpha_R^-1/2) integrated over the bin of x.
 * bounds: device vector of the k+1 bin boundaries (Discretization.bin_boundaries, bsi.py:29-30, built by the
 * caller with torch.linspace exactly as the reference does); bucket index = clamp(trunc((x - lo_edge)/dx), 0, k-1)
 * with lo_edge = min - dx/2 (bsi.py:32-35); outer bins are open (bsi.py:241-242), floor 1e-20 (244).
 * k == 0 (bounds NULL): continuous -log N(x; x_hat, 1/alpha_R) (bsi.py:235).  out[r] = -sum_D log p. */
int bsi_recon_nll(const float* x, const float* x_hat, float alpha_R, const float* bounds, float lo_edge, float dx,
                  int k, int rows, int B, int D, float* out, bsi_stream_t stream);

/* bsi.py:41-48 (Discretization.to_8bit_image) on device: uint8 = clamp(((x-lo)/(hi-lo))*255, 0, 255). */
int bsi_to_uint8(const float* x, float lo, float hi, size_t n, uint8_t* out, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The other two algorithm wrappers that share the denoisers with BSI — bsi/vdm.py (Variational Diffusion Models) and
 * bsi/bfn.py (Bayesian Flow Networks); SURVEY §8(f) rank 4.  Row r of an [n_samples, batch] quantity is r = s*B + b.
 * ---------------------------------------------------------------------------------------- */
/* vdm.py:385-397 / bfn.py:313-325: t = (perm/(1+total) + offset) mod 1 (low-discrepancy time grid). */
int bsi_tgrid(const int64_t* perm, const float* offset, int total, float* t, bsi_stream_t stream);
/* vdm.py:343-347, bfn.py:302-309: out[r] = addcmul(a[r]*x[r % B], b[r], eps[r]). */
int bsi_affine_noise(const float* x, const float* a, const float* b, const float* eps, int rows, int B, int D, float* out,
                     bsi_stream_t stream);
/* vdm.py:365-379: out = a[idx]*z + b[idx]*xh + c[idx]*eps over n floats (one ancestral sampling step). */
int bsi_axpbypcz(const float* z, const float* xh, const float* eps, const float* a, const float* b, const float* c, int idx, size_t n,
                 float* out, bsi_stream_t stream);
/* bfn.py:291 (.clip) and its backward (gradient passes where lo <= raw <= hi). */
int bsi_clip(const float* x, float lo, float hi, size_t n, float* out, bsi_stream_t stream);
int bsi_clip_bwd(const float* g, const float* raw, float lo, float hi, size_t n, float* out, bsi_stream_t stream);
/* vdm.py:138-150,324-329: gamma = lerp(gamma_0, gamma_1, t); alpha = sqrt(sigmoid(-gamma)), sigma = sqrt(sigmoid(gamma)),
 * snr = exp(-gamma), c_skip = 1/alpha, c_out = -sigma/alpha (x_hat = (z - sigma f)/alpha).  Outputs nullable. */
int bsi_vdm_coeffs(const float* t, int n, float gamma_0, float gamma_1, float* alpha, float* sigma, float* snr, float* c_skip,
                   float* c_out, bsi_stream_t stream);
/* vdm.py:350-379 over a schedule t[0..k]: z_s = c_z[i] z_t + c_x[i] x_hat + std[i] eps for the step t[i] -> t[i+1];
 * dsnr[i] = snr(t[i+1]) - snr(t[i]) (vdm.py:231, nullable). */
int bsi_vdm_step_coeffs(const float* t, int k, float gamma_0, float gamma_1, float* c_z, float* c_x, float* std_, float* dsnr,
                        bsi_stream_t stream);
/* vdm.py:167-195: Normal(x_hat, std) evaluated at the k bin centres, normalised over bins; out[r] = -sum_D log p[bin(x)]. */
int bsi_vdm_recon_nll(const float* x, const float* x_hat, float std_, const float* bounds, float lo_edge, float dx, int k, int rows,
                      int B, int D, float* out, bsi_stream_t stream);
/* vdm.py:127-136: out[r] = 0.5 * sum_D (var_1 + (1 - var_1) x^2 - log var_1 - 1). */
int bsi_vdm_prior(const float* x, float var_1, int rows, int D, float* out, bsi_stream_t stream);
/* bfn.py:282-309,197-198 per time t[i] (gamma = 1 - sigma_1^(2t)): fa = gamma, fb = sqrt(gamma(1-gamma)) (flow distribution);
 * c_skip = 1/g, c_out = -sqrt((1-g)/g) with g at max(t, t_min), both 0 where t < t_min (x_hat before the clip);
 * w = sigma_1^(-2t).  Outputs nullable. */
int bsi_bfn_coeffs(const float* t, int n, float sigma_1, float t_min, float* fa, float* fb, float* c_skip, float* c_out, float* w,
                   bsi_stream_t stream);
/* bfn.py:215-226: alpha[i] = sigma_1^(-2 t[i+1]) (1 - sigma_1^(2 (t[i+1]-t[i]))), rho[0] = 1, rho[i+1] = rho[i] + alpha[i]
 * (rho, alpha feed bsi_refine_step as lam, alpha); wdisc[i] = sigma_1^((-2/k)(i+1)) (bfn.py:176-181, nullable). */
int bsi_bfn_schedule(const float* t, int k, float sigma_1, float* alpha, float* rho, float* wdisc, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks of the denoisers — bsi/models/dit.py, bsi/models/pos_emb.py, bsi/nn
 * ---------------------------------------------------------------------------------------- */

/* fp32 -> bf16 (round to nearest even); used to refresh the bf16 weight shadows.
 * Copies `rows` rows of `cols` floats into rows of `ld_out` bf16 (zero-filling cols..ld_out). */
int bsi_cast_bf16(const float* in, int rows, int cols, void* out, int ld_out, bsi_stream_t stream);
/* same with an input row stride (ld_in floats); out_bf16 = bf16(silu(pre)) elementwise */
int bsi_cast_rows_bf16(const float* in, int ld_in, int rows, int cols, void* out, int ld_out, bsi_stream_t stream);
int bsi_silu_bf16(const float* pre, size_t n, void* out, bsi_stream_t stream);

/* fourier_features.py:21-36 (FourierFeatures.forward): x fp32 viewed as [outer, C, inner] ->
 * out [outer, C*nf*2, inner], nf = n_max-n_min+1, channel (c*nf + n)*2 + o = sin(o*pi/2 + fl32(2*pi*2^n) * x[c]). */
int bsi_fourier_features(const float* x, int outer, int C, int inner, int n_min, int n_max, float* out,
                         bsi_stream_t stream);

/* pos_emb.py:78-84: out[r, j] = sin(bias[j] + scale[j]*t[r]) (fp32, accurate sin). out_bf16 nullable. */
int bsi_nyquist_embed(const float* t, int rows, const float* scale, const float* bias, int size,
                      float* out_f32, void* out_bf16, bsi_stream_t stream);

/* GEMM  C[M,N] = A[M,K] . W[N,K]^T  with bf16 operands, fp32 accumulation (MFMA 16x16x32) and a fused
 * epilogue.  Replaces every nn.Linear of the DiT (dit.py:33-34,71-81,154,163-165; mlp.py:34-38).
 * K % 64 == 0, N % 16 == 0, lda/ldw in elements (multiples of 8). */
enum {
    BSI_EPI_BIAS_F32 = 0,       /* out_f32[m,n] = acc + bias[n]                                       */
    BSI_EPI_BIAS_BF16 = 1,      /* out_bf16[m,n] = bf16(acc + bias[n])                                */
    BSI_EPI_BIAS_GELU_BF16 = 2, /* out_bf16 = bf16(gelu_tanh(acc + bias))           (mlp.0, dit.py:71-76) */
    BSI_EPI_BIAS_SILU_BF16 = 3, /* out_bf16 = bf16(silu(acc + bias))       (adaLN_modulation.0-1, :79-81) */
    BSI_EPI_GATE_RESID = 4,     /* out_f32[m,n] += gate[row(m), n] * (acc + bias[n])    (dit.py:93-102) */
    BSI_EPI_BIAS_POS_F32 = 5,   /* out_f32[m,n] = acc + bias[n] + pos[m % tokens, n]     (dit.py:178)  */
    BSI_EPI_BIAS_GELU_DUAL = 6, /* training: out2_bf16 = bf16(acc + bias) (saved for backward), out_bf16 = bf16(gelu(.)) */
    BSI_EPI_MUL_GELUGRAD_BF16 = 7, /* backward: out_bf16 = bf16((acc + bias) * gelu_tanh'(aux[m,n])), aux bf16 like out */
};
typedef struct bsi_gemm_args {
    const void* A;  /* bf16 [M, lda] */
    const void* W;  /* bf16 [N, ldw] */
    const float* bias; /* [N] or NULL */
    void* out;      /* f32 or bf16 [M, ldo] */
    int M, N, K;
    int lda, ldw, ldo;
    int epilogue;
    /* GATE_RESID: gate value for row m is gate[(m / tokens) % gate_rows * gate_stride + n] */
    const float* gate;
    int gate_rows, gate_stride;
    int tokens;       /* tokens per sample (GATE_RESID, BIAS_POS) */
    const float* pos; /* [tokens, N] (BIAS_POS) */
    const void* aux;  /* bf16 [M, ldo] (MUL_GELUGRAD) */
    void* out2;       /* bf16 [M, ldo] (BIAS_GELU_DUAL) */
    float* colsum_rows; /* optional, MUL_GELUGRAD with M > 128 only: fp32 [ceil(M / 128)][N]; row s receives the column sums of output rows
                           128 s .. 128 s + 127 (of the bf16 values written) -- summed over s (bsi_colsum_rows_f32) this is the bias
                           gradient of the Linear whose output gradient `out` is (autograd's grad_output.sum(0) of fc1) */
} bsi_gemm_args;
int bsi_gemm_bf16(const bsi_gemm_args* a /*host*/, bsi_stream_t stream);
/* Small-M latency path: same contract, plus a caller-owned scratch buffer.  When M <= 2048, K >= 2048 and the epilogue is a plain
 * bf16 one, the K range is split over workgroups (fp32 partial sums in `workspace`, summed in fixed order) so that a GEMM of a few
 * tiles does not walk its whole K loop on a handful of CUs; otherwise (or with too small a workspace) it is bsi_gemm_bf16.
 * bsi_gemm_splitk_workspace_bytes returns the bytes that make the split possible for a shape (0 = never split). */
size_t bsi_gemm_splitk_workspace_bytes(int M, int N, int K);
/* `groups` GEMMs of one shape (a->M, N, K, leading dimensions; epilogue BSI_EPI_BIAS_F32) in one launch: A, W, bias and out of group g
 * lie g * stride bytes behind those of `a` (stride_a = 0: a shared A).  dit.py:77-81 for all blocks at once: the adaLN MLP depends on t only. */
int bsi_gemm_bf16_grouped(const bsi_gemm_args* a, int groups, size_t stride_a, size_t stride_w, size_t stride_bias, size_t stride_out,
                          bsi_stream_t stream);
/* The same for BSI_EPI_BIAS_F32 on per-sample GEMMs (M <= 2048 rows, K >= 1024: the adaLN MLP of a train step, dit.py:77-81): the
 * slice count depends on K only, so a result does not depend on how many rows were computed with it. */
size_t bsi_gemm_splitk_f32_workspace_bytes(int M, int N, int K);
int bsi_gemm_bf16_ws(const bsi_gemm_args* a /*host*/, void* workspace, size_t workspace_bytes, bsi_stream_t stream);
/* Weight-gradient GEMM (backward of nn.Linear w.r.t. its weight, autograd `grad_output.T @ input`):
 *   out[N,K] (+)= P[M,N]^T . Q[M,K],  P = dY and Q = X are bf16 row-major (token index m slow), fp32 result.
 * The token range is split over workgroups; partial tiles are summed deterministically from `workspace`
 * (bsi_gemm_tn_workspace_bytes).  accumulate != 0 adds to `out` (gradient accumulation). */
size_t bsi_gemm_tn_workspace_bytes(int M, int N, int K);
int bsi_gemm_tn_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int N, int K, float* out, int ldc,
                     int accumulate, void* workspace, bsi_stream_t stream);
/* Same, plus colsum_out[N] (+)= sum_m P[m, :] from the same launch (the bias gradient of the Linear: one extra MFMA with an
 * all-ones operand per stage in the workgroups of the first k tile, instead of a second pass over dY). */
/* TWO weight gradients of one token count in ONE launch: out1[N1,K] = P1^T Q1 and out2[N2,K] = P2^T Q2 (M rows each, the Q operands
 * share ldq, the outputs are dense: ldc = K; N1 a multiple of 256).  The qkv and out-projection gradients of a DiT block have 48 and
 * 16 output tiles: together they are the fc1 gradient's 64, split over M four ways = one workgroup per CU, instead of 5- and 15-way
 * splits of their own.  workspace: bsi_gemm_tn_workspace_bytes(M, N1 + N2, K).  Replaces two torch.mm(dY.t(), X) of autograd over
 * nn.Linear (bsi/models/dit.py:33-34). */
int bsi_gemm_tn_pair_bf16(const void* P1, int ldp1, const void* Q1, int N1, float* out1, const void* P2, int ldp2, const void* Q2, int N2,
                          float* out2, int ldq, int M, int K, void* workspace, bsi_stream_t stream);
int bsi_gemm_tn_bias_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int N, int K, float* out, int ldc,
                          float* colsum_out, int accumulate, void* workspace, bsi_stream_t stream);
/* out[c] = sum over r < rows of src[r * ld + c] (fp32, c < cols, cols % 4 == 0), rows added in index order (deterministic): the second
 * stage of the bias gradients whose per-slab column sums the producers of dY write (bsi_gemm_args::colsum_rows, the LayerNorm /
 * gate backward, the attention backward).  Up to four jobs in one pair of launches.  scratch: the sum of bsi_colsum_rows_scratch_bytes(rows, cols) over the
 * jobs.  ld % 4 == 0, src and out 16-byte aligned. */
typedef struct bsi_colsum_job { const float* src; int rows, cols, ld; float* out; } bsi_colsum_job;
size_t bsi_colsum_rows_scratch_bytes(int rows, int cols);
int bsi_colsum_rows_f32(const bsi_colsum_job* jobs /*host*/, int njobs, void* scratch, bsi_stream_t stream);
/* Bias gradient: out[n] (+)= sum_m Y[m,n] for bf16 Y [M, ld]. */
size_t bsi_colsum_workspace_bytes(int N);
int bsi_colsum_bf16(const void* Y, int ld, int M, int N, float* out, int accumulate, void* workspace,
                    bsi_stream_t stream);

/* Testing hook: schedule of the large-tile GEMM.  12 (default) = bf16-output epilogues on the K = 64 half-stage ring, fp32-output
 * ones on the K = 32 ring; 6 = the K = 32 ring for every epilogue (A/B partner of the full-size tests).  Bits 16..23: band
 * height of the tile walk (0 = the default 4).  The two schedules agree to one bf16 ulp. */
int bsi_gemm_set_variant(int variant);

/* dit.py:50-55,66,96: out_bf16[m,:] = LayerNorm(x[m,:]; eps, no affine) * (1 + scale[row]) + shift[row]
 * with row = (m / tokens) % mod_rows; shift/scale point into the adaLN chunk table (stride mod_stride).
 * With shift == scale == NULL it is a plain LayerNorm with optional affine weight/bias (dit.py:163). */
int bsi_ln_modulate(const float* x, int M, int d, float eps, const float* shift, const float* scale,
                    int mod_rows, int mod_stride, int tokens, const float* ln_w, const float* ln_b,
                    void* out_bf16, bsi_stream_t stream);

/* The same with the block's gated residual update fused in front (dit.py:93-102: torch.addcmul(x, gate, branch)):
 *   x[m,:] += gate[row] * delta[m,:]   (delta: bf16 [M,d] branch output incl. bias, gate: adaLN chunk, same row
 *   indexing as shift/scale), written back to x, then LayerNorm+modulate of the updated row into out_bf16.
 * delta == NULL skips the update; out_bf16 == NULL skips the norm (pure residual update).  delta may alias out_bf16. */
int bsi_resid_ln_modulate(float* x, int M, int d, float eps, const void* delta, const float* gate,
                          const float* shift, const float* scale, int mod_rows, int mod_stride, int tokens,
                          const float* ln_w, const float* ln_b, void* out_bf16, bsi_stream_t stream);

/* Two consecutive updates of dit.py:93-102 with ONE store of the residual row between them.  A pass with write_x = 0 applies
 * (delta, gate) in registers only: out_bf16 = LayerNorm(x + gate * delta) * (1 + scale) + shift, x untouched.  The next pass
 * names that update as (delta0, gate0) -- the caller keeps delta0 alive until then -- in front of its own (delta, gate):
 *   x[m,:] = fma(gate, delta, fma(gate0, delta0, x[m,:]))   (the same two fp32 fmas as two stored passes: bit-identical rows),
 * stored when write_x != 0, then normalised as above.  delta0 == NULL: as bsi_resid_ln_modulate without affine weights. */
int bsi_resid2_ln_modulate(float* x, int M, int d, float eps, const void* delta0, const float* gate0, const void* delta,
                           const float* gate, int write_x, const float* shift, const float* scale, int mod_rows,
                           int mod_stride, int tokens, void* out_bf16, bsi_stream_t stream);

/* dit.py:39-46 / attention.py:34-40: softmax(q k^T / sqrt(dh)) v per (batch, head), non-causal.
 * qkv: bf16 [B, tokens, 3, heads, dh] (row stride ld_qkv elements); out: bf16 [B, tokens, heads*dh]
 * (row stride ld_out).  dh in {64, 128}; tokens % 64 == 0. */
int bsi_attention_fwd(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out,
                      bsi_stream_t stream);

/* As bsi_attention_fwd, additionally saving lse[b,h,q] = log sum_k exp(q.k/sqrt(dh)) (fp32 [B,heads,tokens]). */
int bsi_attention_fwd_lse(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out,
                          float* lse, bsi_stream_t stream);
/* Backward of the attention (autograd of dit.py:43-44): dqkv (bf16, layout of qkv) from qkv, the forward output
 * `out`, its gradient `dout` (both bf16 [B,tokens,heads*dh], row stride ld_o) and lse.  dh = 64, tokens <= 256. */
int bsi_attention_bwd(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                      int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, bsi_stream_t stream);
/* Same contract for long sequences / wide heads (UNet centre attention, attention.py:18,38: 1024 positions, dh 128):
 * the other side of each pass is streamed through LDS; tokens % 64 == 0, dh 64 or 128, no dropout. */
int bsi_attention_bwd_long(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse,
                           int B, int tokens, int heads, int dh, void* dqkv, int ld_dqkv, bsi_stream_t stream);

/* Backward building blocks of the DiT block (autograd of dit.py:50-55,87-103):
 * bsi_gate_bwd: for x2 = x1 + gate*delta and dX = dL/dx2: ddelta = bf16(gate*dX), dgate[b] += sum_tokens dX*delta,
 *   and x is rewound in place to x1 = x2 - gate*delta (the forward keeps only the final residual stream).
 * bsi_ln_mod_bwd: for xn = LN(x)*(1+scale)+shift and dxn: dshift[b] += sum dxn, dscale[b] += sum dxn*LN(x),
 *   dX += LayerNorm-backward(dxn*(1+scale)).  Per-sample rows: gate/scale/dgate/... indexed [b*stride]. */
int bsi_gate_bwd(const float* dX, const void* delta, float* x, const float* gate, int gate_stride, float* dgate,
                 int dgate_stride, int M, int d, int tokens, void* ddelta, bsi_stream_t stream);
int bsi_ln_mod_bwd(const void* dxn, const float* x, const float* scale, int mod_stride, float* dshift, float* dscale,
                   int dmod_stride, float* dX, int M, int d, int tokens, float eps, bsi_stream_t stream);
/* Fused form used by the training engine, whose tape keeps every LayerNorm input x (fp32) and its per-row (mean, rstd)
 * `stats` [M][2]: LayerNorm-modulate backward (as bsi_ln_mod_bwd, statistics read instead of recomputed) followed, on the
 * same row, by the gated-residual backward of the branch below it (ddelta = bf16(gate*dX), dgate[b] += sum dX*delta; no
 * rewinding of x).  dxn == NULL: gate part only; delta == NULL: LayerNorm part only.  d <= 1024, tokens % 64 == 0. */
int bsi_ln_gate_bwd(const void* dxn, const float* x, const float* stats, const float* scale, int mod_stride, float* dshift,
                    float* dscale, int dmod_stride, float* dX, const void* delta, const float* gate, int gate_stride,
                    float* dgate, int dgate_stride, void* ddelta, int M, int d, int tokens, bsi_stream_t stream);
/* out_bf16[i] = ds[i] * silu'(pre[i]) (pre == NULL: plain cast); fp32 [rows,cols] -> bf16 [cols, ld] transposed. */
int bsi_silu_bwd_bf16(const float* ds, const float* pre, size_t n, void* out, bsi_stream_t stream);
int bsi_cast_transpose_bf16(const float* in, int rows, int cols, void* out, int ld_out, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DenoisingDiT engine — bsi/models/dit.py:106-233
 * ---------------------------------------------------------------------------------------- */
typedef struct bsi_dit_config {
    int C, H, W;     /* data_shape */
    int patch;       /* patch_size */
    int dim, depth, heads;
    int ff_nmin, ff_nmax; /* FourierFeatures n_min..n_max (fourier_features.py:11-19); ff_nmin > ff_nmax = none */
} bsi_dit_config;

typedef struct bsi_dit_block_weights {
    const void *qkv_w, *out_w, *fc1_w, *fc2_w, *ada0_w, *ada2_w;    /* bf16 [N][K] shadows */
    const float *qkv_b, *out_b, *fc1_b, *fc2_b, *ada0_b, *ada2_b;   /* fp32 biases */
} bsi_dit_block_weights;

typedef struct bsi_dit_weights {
    const void* enc_w;     /* bf16 [dim][kpad], kpad = bsi_dit_kpad(cfg) (zero padded)        */
    const float* enc_b;    /* [dim]                                                           */
    const float* pos;      /* [tokens, dim] fp32 patch_pos_embedding (dit.py:135-146)         */
    const float* t_scale;  /* [dim] NyquistPositionalEmbedding(dim,1000).scale (dit.py:147)   */
    const float* t_bias;   /* [dim]                                                           */
    const float* dec_ln_w; /* [dim] patch_decoder.0 */
    const float* dec_ln_b;
    const float* dec_w;    /* fp32 [patch*patch*C][dim] patch_decoder.1 */
    const float* dec_b;
    const bsi_dit_block_weights* blocks; /* host array [depth] */
} bsi_dit_weights;

int bsi_dit_kpad(const bsi_dit_config* cfg);
int bsi_dit_tokens(const bsi_dit_config* cfg);
/* bytes of scratch needed by bsi_dit_forward for a batch of B images */
size_t bsi_dit_workspace_bytes(const bsi_dit_config* cfg, int B);

/* adaLN tables (dit.py:77-81,90-92) for `rows` conditioning times: mod[r, l, 0:6*dim] =
 * Linear2_l(SiLU(Linear1_l(emb(t[r])))) for every block l.  mod: fp32 [rows, depth, 6*dim].
 * In `sample` t is shared by the whole batch, so the table is built once for all k+1 steps.
 * scratch: bsi_dit_adaln_scratch_bytes(cfg, rows). */
size_t bsi_dit_adaln_scratch_bytes(const bsi_dit_config* cfg, int rows);
int bsi_dit_adaln(const bsi_dit_config* cfg, const bsi_dit_weights* w /*host*/, const float* t, int rows,
                  float* mod, void* scratch, bsi_stream_t stream);

/* One denoiser evaluation f(c_in*mu, t) (dit.py:225-233) with the BSI preconditioning fused around it
 * (bsi.py:381-386).  mu: fp32 [B,C,H,W].  mod: adaLN table rows for this call, `mod_rows` in {1, B}
 * (row b % mod_rows).  c_in/c_skip/c_out: device vectors indexed [b*coef_stride] (coef_stride 0 = shared
 * scalar); pass NULL for all three to evaluate the bare denoiser.  out: fp32 [B,C,H,W] =
 * c_skip*mu + c_out*f  (or f).  tokens_out (nullable): fp32 [B*tokens, dim] copy of the final residual
 * stream (tests). */
int bsi_dit_forward(const bsi_dit_config* cfg, const bsi_dit_weights* w /*host*/, int B, const float* mu,
                    const float* mod, int mod_rows, const float* c_in, const float* c_skip, const float* c_out,
                    int coef_stride, float* out, void* workspace, float* tokens_out, bsi_stream_t stream);

/* Two-resource evaluation: the same result as bsi_dit_forward (bit for bit: every row's arithmetic is independent of the batch it
 * is launched in), with the batch split into two halves whose launches are interleaved over a PAIR of CU-masked streams
 * (hipExtStreamCreateWithCUMask): the matrix-pipe-bound launches (GEMMs, attention) on the large partition G, the HBM-bound ones
 * (prologue, LayerNorm+modulate passes, final kernel) of the OTHER half on the small partition H (h_cus compute units, a multiple
 * of 8 = the same number on every XCD), handed over by events.  While G multiplies for one half, H streams for the other: the loop
 * of bsi.py:312-336 / dit.py:96-103 no longer alternates between an idle memory system and idle matrix pipes.  The caller's stream
 * forks into the pair at entry and joins at exit; workspace: bsi_dit_workspace_bytes(cfg, B) bytes as for bsi_dit_forward.
 * flags: BSI_PAIR_ATTN_ON_H = attention launches on H instead of G. */
typedef struct bsi_cu_pair bsi_cu_pair;
#define BSI_PAIR_ATTN_ON_H 1
int bsi_cu_pair_create(int h_cus, bsi_cu_pair** pair);
int bsi_cu_pair_destroy(bsi_cu_pair* pair);
int bsi_cu_pair_streams(const bsi_cu_pair* pair, bsi_stream_t* g, bsi_stream_t* h, int* h_cus);
int bsi_dit_forward_pair(const bsi_dit_config* cfg, const bsi_dit_weights* w /*host*/, int B, const float* mu,
                         const float* mod, int mod_rows, const float* c_in, const float* c_skip, const float* c_out,
                         int coef_stride, float* out, void* workspace, bsi_cu_pair* pair, int flags, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * VDM-UNet building blocks — bsi/models/vdm_unet.py, bsi/nn/residual_block.py, bsi/nn/attention.py.
 * Activations are NHWC: [B*H*W pixels][channels].
 * ---------------------------------------------------------------------------------------- */
enum {
    BSI_CONV_BIAS_BF16 = 0,      /* out_bf16 = bf16(conv + bias)                                   (attention.py:29) */
    BSI_CONV_FILM_SILU_BF16 = 1, /* out_bf16 = bf16(silu((conv + bias)*(scale+1) + shift))  (residual_block.py:44-46) */
    BSI_CONV_BIAS_RESID_F32 = 2, /* out_f32 = conv + bias (+ resid)                    (residual_block.py:48,63) */
};
typedef struct bsi_conv_args {
    const void* x;     /* bf16 NHWC [B*H*W, Cin] */
    const void* x2;    /* optional bf16 NHWC [B*H*W, Cin2]: extra 1x1 K steps (skip conv folded in) */
    const void* w;     /* bf16 [Cout][taps*Cin + Cin2] from bsi_conv_weight_pack */
    const float* bias; /* [Cout] */
    const void* zeros; /* unused by the forward kernel (padding taps use out-of-range buffer offsets); wgrad reads 256 zero bytes */
    void* out;         /* bf16 or fp32 [B*H*W, ldo] */
    const float* film; /* FILM: fp32 [film_rows][film_stride], scale at [0, Cout), shift at [Cout, 2 Cout) */
    const float* resid;/* BIAS_RESID_F32: fp32 [B*H*W, ldo] or NULL */
    int film_rows, film_stride;
    int B, H, W, Cin, Cin2, Cout, taps /* 9 = 3x3 pad 1, 1 = 1x1 */, ldo, epilogue;
    float* gn_partial; /* BIAS_RESID_F32, optional: [B*H*W/128][Cout/4][2] = (mean, M2) of every 128-pixel x 4-channel block of
                          `out`, written by the epilogue for bsi_groupnorm_apply_nhwc (needs H*W % 128 == 0, Cout % 64 == 0) */
} bsi_conv_args;
/* Conv2d(stride 1, zero padding) as implicit GEMM on bf16 MFMA; Cin, Cin2 multiples of 32, Cout of 16. */
int bsi_conv_nhwc_bf16(const bsi_conv_args* a /*host*/, bsi_stream_t stream);
/* > 0 (a multiple of 8, one share per XCD): launch at most this many (persistent) workgroups, 0 = one per CU.  Lets small inputs exercise several tiles per workgroup. */
int bsi_conv_set_grid_limit(int max_workgroups);
/* Kernel experiments only: switch parts of the convolution kernel off (ConvParams::abl in conv_igemm.hip); 0 = normal. */
int bsi_conv_set_ablation(int flags);
/* fp32 Conv2d weight [Cout][Cin][kh][kw] -> bf16 [Cout][ld] at column col0 with K index (tap, channel), Cin padded. */
int bsi_conv_weight_pack(const float* w, int Cout, int Cin, int taps, int cin_pad, int ld, int col0, void* out,
                         bsi_stream_t stream);
/* Weights of the INPUT-gradient convolution: fp32 [Cout][Cin][taps] -> bf16 [Cin][ld],
 * out[ci][(taps-1-tap)*Cout + co] = w[co][ci][tap] (taps rotated by 180 degrees, channels swapped); feeding it to
 * bsi_conv_nhwc_bf16 with dY as the input computes autograd's conv2d input gradient. */
int bsi_conv_weight_pack_t(const float* w, int Cout, int Cin, int taps, int ld, void* out, bsi_stream_t stream);
/* Both re-arrangements for many convolutions in ONE launch (a train step re-packs every convolution weight after the optimizer
 * update).  descs: n descriptors in DEVICE memory; transposed = 0: bsi_conv_weight_pack(w, Cout, Cin, taps, cin_pad, ld, col0, out),
 * transposed = 1: bsi_conv_weight_pack_t(w, Cout, Cin, taps, ld, out) (cin_pad, col0 ignored). */
/* All bf16 shadows of a model's fp32 weight matrices in one launch: for descriptor i, dst (or NULL) = row-major [rows][ld] (columns
 * cols .. ld-1 zero: the padded patch-encoder weight), dst_t (or NULL) = the transpose [cols][ld_t] (ld_t >= rows).  descs is a DEVICE
 * array; tile0 = the sum of bsi_cast_batch_tiles(rows, cols, dst ? ld : cols) over the descriptors in front, tiles = the total.
 * Replaces 265 launches of bsi_cast_bf16 / bsi_cast_transpose_bf16 per DiT-L optimizer step. */
typedef struct bsi_cast_desc { const float* src; void* dst; void* dst_t; int rows, cols, ld, ld_t, tile0, reserved; } bsi_cast_desc;
int bsi_cast_batch_tiles(int rows, int cols, int ld);
int bsi_cast_batch_bf16(const bsi_cast_desc* descs /*device*/, int n, int tiles, bsi_stream_t stream);
typedef struct bsi_conv_pack_desc {
    const float* w;
    void* out;
    int Cout, Cin, taps, cin_pad, ld, col0;
} bsi_conv_pack_desc;
int bsi_conv_weight_pack_batch(const bsi_conv_pack_desc* descs /*device*/, int n, int transposed, bsi_stream_t stream);
/* WEIGHT gradient of bsi_conv_nhwc_bf16 (autograd of nn.Conv2d w.r.t. weight, residual_block.py:40-48, attention.py:29-30):
 *   out_packed[co][tap*Cin + ci] (+)= sum_m dy[m, co] * x[pixel m shifted by tap, ci]   (zero outside the image)
 *   out_packed[co][taps*Cin + c] (+)= sum_m dy[m, co] * x2[m, c]                        (folded 1x1 skip conv)
 * dy bf16 [B*H*W, ldy], x bf16 NHWC [B*H*W, Cin], x2 bf16 [B*H*W, Cin2] or NULL; channel counts multiples of 8;
 * zeros: >= 256 bytes of zeros.  fp32 result in the packed layout of bsi_conv_weight_pack; bsi_conv_wgrad_unpack
 * converts to the nn.Conv2d layout [Cout][Cin][kh][kw].  Deterministic (split-M slabs in `workspace`). */
size_t bsi_conv_wgrad_workspace_bytes(int M, int Cin, int Cin2, int Cout, int taps);
int bsi_conv_wgrad_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B, int H, int W,
                             int Cin, int Cin2, int Cout, int taps, float* out_packed, int accumulate, void* workspace,
                             bsi_stream_t stream);
/* Same, plus the bias gradient dbias[co] (+)= sum_m dY[m, co] from one extra MFMA per stage (all-ones operand) in the same
 * kernel: autograd of nn.Conv2d w.r.t. bias without a second pass over dY. */
int bsi_conv_wgrad_bias_nhwc_bf16(const void* dy, int ldy, const void* x, const void* x2, const void* zeros, int B, int H, int W,
                                  int Cin, int Cin2, int Cout, int taps, float* out_packed, float* dbias, int accumulate,
                                  void* workspace, bsi_stream_t stream);
int bsi_conv_wgrad_unpack(const float* packed, int Cout, int Cin, int taps, int cin_pad, int ld, int col0, int accumulate,
                          float* out, bsi_stream_t stream);
/* GroupNorm(32 groups, affine, eps) over cat(x1, x2) channels (x2 nullable) per image, optional SiLU -> bf16 NHWC
 * (residual_block.py:42-43, vdm_unet.py:52,84); raw_bf16 (nullable) receives the un-normalised bf16 copy. */
int bsi_groupnorm_nhwc(const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                       const float* beta, float eps, int silu, void* out_bf16, void* raw_bf16, bsi_stream_t stream);
/* Same, and saves (mean, rstd) of every (image, group) to stats[B][32][2] for the backward pass (H*W <= 1024). */
int bsi_groupnorm_stats_nhwc(const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma, const float* beta,
                             float eps, int silu, void* out_bf16, void* raw_bf16, float* stats, bsi_stream_t stream);
/* The same GroupNorm as ONE streaming pass: the statistics come from the partials the producing convolutions wrote
 * (bsi_conv_args::gn_partial: part1 [B*HW/128][C1/4][2], part2 [B*HW/128][C2/4][2] = (mean, M2) of 128 pixels x 4 channels),
 * merged per (image, group) in a fixed order, so x is read once and never held.  C1 + C2 = 128 or 256, H*W % 128 == 0;
 * stats (nullable) receives (mean, rstd) as bsi_groupnorm_stats_nhwc writes them. */
int bsi_groupnorm_apply_nhwc(const float* x1, int C1, const float* part1, const float* x2, int C2, const float* part2, int B, int HW,
                             const float* gamma, const float* beta, float eps, int silu, void* out_bf16, void* raw_bf16,
                             float* stats, bsi_stream_t stream);
/* Backward of bsi_groupnorm_nhwc: da bf16 [B*HW, C1+C2] is the gradient of the (SiLU'd) output.
 *   out1 [B*HW, C1] = dx1 (+ add[:, :C1]) (+ add_b),  out2 [B*HW, C2] = dx2 (+ add[:, C1:]);  add: fp32 [B*HW, C1+C2] or NULL,
 *   add_b: fp32 [B*HW, C1] or NULL (out1 may alias add when C2 == 0);  dgamma, dbeta [C1+C2] are ACCUMULATED (atomics). */
int bsi_groupnorm_bwd_nhwc(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW,
                           const float* gamma, const float* beta, float eps, int silu, const float* add, const float* add_b,
                           float* out1, float* out2, float* dgamma, float* dbeta, bsi_stream_t stream);
/* Same, and also writes the x1 gradient as bf16 [B*HW, C1] (the dY operand of the next block's weight / input gradient
 * convolutions: saves a separate fp32 -> bf16 pass over it); `stats` = the (mean, rstd) pairs bsi_groupnorm_stats_nhwc saved in the
 * forward pass ([B][32][2] floats), NULL = recompute them. */
int bsi_groupnorm_bwd_cast_nhwc(const void* da, const float* x1, int C1, const float* x2, int C2, int B, int HW, const float* gamma,
                                const float* beta, float eps, int silu, const float* add, const float* add_b, float* out1,
                                float* out2, float* dgamma, float* dbeta, void* out1_bf16, const float* stats /* or NULL */,
                                bsi_stream_t stream);
/* Training form of the FiLM stage (residual_block.py:21-24,44-46): y = Dropout_p(SiLU(h1*(scale+1)+shift)), h1 and y bf16
 * [M, N], film fp32 rows (scale at [0,N), shift at [N,2N)) selected by (m / HW) % film_rows; the dropout mask is the
 * counter hash of (seed, site, row m, column n) (bsi_dropout_mask exports the same mask).  _bwd: dh1 = bf16(dU*(scale+1)) with
 * dU = dy*mask/(1-p)*silu'(u); dfilm[b, n] += sum dU*h1, dfilm[b, N+n] += sum dU (atomics; N 64 or 128, HW % 64 == 0). */
int bsi_film_silu(const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride, float dropout_p,
                  unsigned long long seed, unsigned site, void* y, bsi_stream_t stream);
int bsi_film_silu_bwd(const void* dy, const void* h1, int M, int N, int HW, const float* film, int film_rows, int film_stride,
                      float dropout_p, unsigned long long seed, unsigned site, void* dh1, float* dfilm, int dfilm_stride,
                      bsi_stream_t stream);
/* Backward of bsi_unet_decode: dh fp32 [B*HW, C] is written; dw [Cout, C] and db [Cout] are ACCUMULATED. */
int bsi_unet_decode_bwd(const float* g_xhat, const float* c_out, int coef_stride, const float* h, int B, int HW, int C,
                        const float* w, int Cout, float* dh, float* dw, float* db, bsi_stream_t stream);
/* decode Conv2d(C -> Cout, 1x1) in fp32 on NHWC fp32 h, NCHW output, fused x_hat = c_skip*mu + c_out*f. */
int bsi_unet_decode(const float* h, int B, int HW, int C, const float* w, const float* bias, int Cout, const float* mu,
                    const float* c_skip, const float* c_out, int coef_stride, float* out, bsi_stream_t stream);

/* DenoisingVDMUNet engine — bsi/models/vdm_unet.py:20-100 (no down/up-sampling; levels residual blocks down, centre
 * [ResBlock, Residual(GroupNorm -> Attention2D), ResBlock], levels blocks up on cat(x, skip)). */
typedef struct bsi_unet_config {
    int C, H, W;      /* data_shape */
    int dim, levels, heads;
    int ff_nmin, ff_nmax;
    int emb_size;     /* pos_emb.size (32) */
    int c_dim;        /* pos_emb.size * pos_emb_mult (128) */
} bsi_unet_config;
typedef struct bsi_unet_resblock_weights {
    const float *gn_w, *gn_b;                  /* layers.0: GroupNorm(32, Cin) */
    const void* conv1_w; const float* conv1_b; /* layers.2: bf16 [dim][9*Cin] (bsi_conv_weight_pack) */
    const void* conv2_w; const float* conv2_b; /* layers.5|6 (+ skip 1x1 appended as K columns, its bias added): bf16 [dim][9*dim (+2*dim)] */
} bsi_unet_resblock_weights;
typedef struct bsi_unet_weights {
    const void* enc_w; const float* enc_b;     /* encode: bf16 [dim][9*cin_pad] */
    const float *dec_w, *dec_b;                /* decode: fp32 [C][dim], [C] */
    const float *pe_scale, *pe_bias;           /* NyquistPositionalEmbedding tables [emb_size] */
    const void* pm1_w; const float* pm1_b;     /* pos_map.1: bf16 [c_dim][64] (K zero padded) */
    const void* pm3_w; const float* pm3_b;     /* pos_map.3: bf16 [c_dim][c_dim] */
    const void* film_w; const float* film_b;   /* all project_onto_scale_shift stacked: bf16 [nblocks*2*dim][c_dim], fp32 [nblocks*2*dim] */
    const bsi_unet_resblock_weights* blocks;   /* host array [2*levels+2]: down 0..L-1, centre 0, centre 2, up 0..L-1 */
    const float *agn_w, *agn_b;                /* center_block.1.fn.0 */
    const void* aqkv_w; const float* aqkv_b;   /* center_block.1.fn.1.to_qkv: bf16 [3*dim][9*dim] */
    const void* aout_w; const float* aout_b;   /* center_block.1.fn.1.to_out: bf16 [dim][9*dim] */
} bsi_unet_weights;
int bsi_unet_cin_pad(const bsi_unet_config* cfg);
size_t bsi_unet_workspace_bytes(const bsi_unet_config* cfg, int B);
size_t bsi_unet_film_scratch_bytes(const bsi_unet_config* cfg, int rows);
/* film[r, blk, 0:2*dim] = project_onto_scale_shift_blk(pos_map(t[r]))  (fp32 [rows, nblocks, 2*dim]). */
int bsi_unet_film(const bsi_unet_config* cfg, const bsi_unet_weights* w /*host*/, const float* t, int rows, float* film,
                  void* scratch, bsi_stream_t stream);
/* out = c_skip*mu + c_out*f(c_in*mu, t) (or f(mu,t) with NULL coefficients); film rows: 1 (shared t) or B. */
int bsi_unet_forward(const bsi_unet_config* cfg, const bsi_unet_weights* w /*host*/, int B, const float* mu, const float* film,
                     int film_rows, const float* c_in, const float* c_skip, const float* c_out, int coef_stride, float* out,
                     void* workspace, bsi_stream_t stream);

/* DenoisingVDMUNet training engine — forward with a tape + hand-written backward (replaces torch autograd over
 * vdm_unet.py:92-100, simplified_unet.py:33-48, residual_block.py:61-64, attention.py:32-41 inside
 * `BSI.train_loss(...).mean().backward()`, bsi/tasks/bsi.py:187-194). */
typedef struct bsi_unet_resblock_weights_t { /* bf16 shadows for the input-gradient products */
    const void* conv1_wT; /* [Cin][9*dim]  bsi_conv_weight_pack_t(layers.2.weight) */
    const void* conv2_wT; /* [dim][9*dim]  bsi_conv_weight_pack_t(layers.5|6.weight) */
    const void* skip_wT;  /* up blocks: [2*dim][dim] = bsi_conv_weight_pack_t(skip.weight, taps 1); else NULL */
} bsi_unet_resblock_weights_t;
typedef struct bsi_unet_weights_t {
    const bsi_unet_resblock_weights_t* blocks; /* host array [2*levels+2], order of bsi_unet_weights.blocks */
    const void* aqkv_wT;  /* [dim][9*3*dim] */
    const void* aout_wT;  /* [dim][9*dim] */
    const void* film_wT;  /* [c_dim][nblocks*2*dim]  transpose of the stacked project_onto_scale_shift weights */
    const void* pm3_wT;   /* [c_dim][c_dim] */
} bsi_unet_weights_t;
typedef struct bsi_unet_resblock_grads { /* fp32, WRITTEN; conv weights in the torch Conv2d layout [Cout][Cin][kh][kw] */
    float *gn_w, *gn_b;
    float* conv1_w; float* conv1_b; /* [dim][Cin][3][3] */
    float* conv2_w; float* conv2_b; /* [dim][dim][3][3]; conv2_b is also the skip conv's bias gradient */
    float* skip_w;                  /* up blocks: [dim][2*dim] (the folded 1x1 skip convolution), else NULL */
} bsi_unet_resblock_grads;
typedef struct bsi_unet_grads {
    float* enc_w; float* enc_b;         /* [dim][C + Fourier channels][3][3] (unpadded) */
    float *dec_w, *dec_b;
    float* pm1_w_padded; float* pm1_b;  /* [c_dim][64] */
    float* pm3_w; float* pm3_b;
    float* film_w; float* film_b;       /* stacked [nblocks*2*dim][c_dim], [nblocks*2*dim] */
    const bsi_unet_resblock_grads* blocks;
    float *agn_w, *agn_b;
    float* aqkv_w; float* aqkv_b;       /* [3*dim][dim][3][3] */
    float* aout_w; float* aout_b;       /* [dim][dim][3][3] */
} bsi_unet_grads;
size_t bsi_unet_tape_bytes(const bsi_unet_config* cfg, int B);
size_t bsi_unet_backward_workspace_bytes(const bsi_unet_config* cfg, int B);
/* out = c_skip*mu + c_out*f(c_in*mu, t) with per-sample t [B]; records the tape.  dropout_p > 0 applies nn.Dropout after the
 * FiLM stage of every residual block (residual_block.py:46) with the counter mask (seed, site = block index). */
int bsi_unet_train_forward(const bsi_unet_config* cfg, const bsi_unet_weights* w /*host*/, int B, const float* mu, const float* t,
                           const float* c_in, const float* c_skip, const float* c_out, float* out, void* tape,
                           float dropout_p, unsigned long long seed, bsi_stream_t stream);
/* Parameter gradients of sum(g_out * out) into `g` (every field is WRITTEN). */
int bsi_unet_backward(const bsi_unet_config* cfg, const bsi_unet_weights* w /*host*/, const bsi_unet_weights_t* wT /*host*/,
                      const bsi_unet_grads* g /*host*/, int B, const float* g_out, const float* c_out, void* tape,
                      void* workspace, float dropout_p, unsigned long long seed, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DenoisingDiT training engine — forward with a tape + hand-written backward (replaces torch autograd over
 * dit.py:87-103,174-181 inside `BSI.train_loss(...).mean().backward()`, bsi/tasks/bsi.py:187-194).
 * ---------------------------------------------------------------------------------------- */
typedef struct bsi_dit_block_weights_t { /* transposed bf16 shadows W^T ([in][out] row-major) for the input gradients */
    const void *qkv_wT, *out_wT, *fc1_wT, *fc2_wT, *ada2_wT;
} bsi_dit_block_weights_t;
typedef struct bsi_dit_weights_t {
    const bsi_dit_block_weights_t* blocks; /* host array [depth] */
} bsi_dit_weights_t;

typedef struct bsi_dit_block_grads { /* fp32 gradient buffers, shapes of the parameters */
    float *qkv_w, *qkv_b, *out_w, *out_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ada0_w, *ada0_b, *ada2_w, *ada2_b;
} bsi_dit_block_grads;
typedef struct bsi_dit_grads {
    float* enc_w_padded; /* [dim, kpad]: the caller keeps columns 0..patch*patch*Cin-1 */
    float* enc_b;
    float *dec_ln_w, *dec_ln_b, *dec_w, *dec_b;
    const bsi_dit_block_grads* blocks; /* host array [depth] */
} bsi_dit_grads;

size_t bsi_dit_tape_bytes(const bsi_dit_config* cfg, int B);
size_t bsi_dit_backward_workspace_bytes(const bsi_dit_config* cfg, int B);
/* out = c_skip*mu + c_out*f(c_in*mu, t) (or f(mu, t) with NULL coefficients), per-sample t/coefficients [B];
 * records the activations the backward needs in `tape` (bsi_dit_tape_bytes). */
int bsi_dit_train_forward(const bsi_dit_config* cfg, const bsi_dit_weights* w /*host*/, int B, const float* mu,
                          const float* t, const float* c_in, const float* c_skip, const float* c_out, float* out,
                          void* tape, float dropout_p, unsigned long long seed, bsi_stream_t stream);
/* Gradients of every parameter given g_out = dL/d(out) [B,C,H,W].  Overwrites the buffers of `g`; consumes the tape
 * (the residual stream in it is rewound in place). */
int bsi_dit_backward(const bsi_dit_config* cfg, const bsi_dit_weights* w /*host*/, const bsi_dit_weights_t* wT /*host*/,
                     const bsi_dit_grads* g /*host*/, int B, const float* g_out, const float* c_out, void* tape,
                     void* workspace, float dropout_p, unsigned long long seed, bsi_stream_t stream);
/* Dropout of the DiT blocks in training (dit.py:43-44 attention-weight dropout, dit.py:70,101 nn.Dropout before the
 * MLP) is a counter-based mask: element (row, col) of site s is kept iff a 16-bit field of hash(seed, s, row, col / 4) is
 * >= p*2^16 (one hash per aligned quad of columns), re-evaluated in the backward kernels; only the attention weights of the
 * DiT geometry proper (256 tokens, head dim 64) keep their mask on the tape, as 64-bit lane-mask words (8 KB per (image, head)
 * and block) that the forward and backward attention kernels use as select masks.  Sites: 2*block (attention weights: row = (b*heads+h)*T + query, col = key) and
 * 2*block+1 (MLP input: row = token, col = feature); UNet: site = residual block, row = pixel, col = channel.
 * bsi_dropout_mask exposes the mask of a [rows, cols] site (uint8 keep flags) for tests. */
int bsi_dropout_mask(float p, unsigned long long seed, unsigned site, unsigned rows, unsigned cols, uint8_t* out /*[rows*cols]*/,
                     bsi_stream_t stream);
/* The same mask of an attention site (rows = pairs * 256, cols = 256) as the lane-mask words the 256-token attention kernels consume:
 * words[pair][qb 0..15][kt 0..15][r 0..3] (64 bit each), bit 16 g + c = keep(query 16 qb + c, key 16 kt + 4 g + r); 8 KB per pair. */
int bsi_attention_dropout_words(float p, unsigned long long seed, unsigned site, int pairs, void* words, bsi_stream_t stream);
/* dit.py:43-44 in training: F.scaled_dot_product_attention(q, k, v, dropout_p = p) and its autograd, with the mask above (row =
 * (b * heads + h) * tokens + query, col = key).  words: NULL (both sides evaluate the hash), or -- 256 tokens, head dim 64 only -- a
 * buffer of 8 KB per (image, head) that the forward FILLS with the lane-mask words and the backward READS (what the training engine
 * keeps on its tape); pass the same value to both.  lse as bsi_attention_fwd_lse.  p = 0 is plain attention. */
int bsi_attention_fwd_dropout(const void* qkv, int ld_qkv, int B, int tokens, int heads, int dh, void* out, int ld_out, float* lse,
                              float p, unsigned long long seed, unsigned site, void* words, bsi_stream_t stream);
int bsi_attention_bwd_dropout(const void* qkv, int ld_qkv, const void* out, const void* dout, int ld_o, const float* lse, int B,
                              int tokens, int heads, int dh, void* dqkv, int ld_dqkv, float p, unsigned long long seed, unsigned site,
                              const void* words, bsi_stream_t stream);
/* Optional hook for the data-parallel SHARDED step (replaces what DDP gets from hiding its traffic behind the backward,
 * /root/reference bsi/tasks/bsi.py:163-166): the all-gather of the updated parameters may still be running, bucket by bucket in
 * forward order on another stream, when the next forward is enqueued.  gates: host array of depth + 2 entries -- [0] the front of the
 * network (patch encoder), [1 + l] block l, [depth + 1] the decoder -- each an event (hipEvent_t or NULL) that the forward's stream
 * waits for before it touches that part's parameters, and the part's descriptors of the bf16 shadow cast (a DEVICE sub-table whose
 * tile0 start at 0, or NULL), launched right behind the wait.  With gates set every block runs its own adaLN MLP (same arithmetic as
 * the grouped launches: bit-identical results).  The array must stay valid until cleared; pass NULL to clear. */
typedef struct bsi_fwd_gate { void* event; const bsi_cast_desc* cast; int n_cast; int cast_tiles; } bsi_fwd_gate;
int bsi_dit_train_forward_set_gates(const bsi_fwd_gate* gates /*host*/, int n);
/* Optional hook for data-parallel overlap: events[l] (hipEvent_t, host array [depth], entries may be NULL) is recorded
 * on the stream as soon as every parameter gradient of block l has been enqueued, so the caller can start that
 * block's gradient all-reduce on another stream while the backward continues.  Pass NULL to clear. */
int bsi_dit_backward_set_events(void* const* events, int depth);

/* ------------------------------------------------------------------------------------------
 * Train-step tail on flat fp32 buffers (bsi/tasks/bsi.py:187-198 + Lightning's clip/optimizer/EMA hooks):
 * config/train.yaml:40 gradient_clip_val, config/task/optimizer/adamw.yaml (torch.optim.AdamW),
 * bsi/tasks/ema_pytorch.py:316-434 (EMA.update -> copy while step <= update_after_step, else lerp_).
 * ---------------------------------------------------------------------------------------- */
size_t bsi_sqnorm_workspace_bytes(void);
/* out_sq[0] = sum g[i]^2 (device scalar). */
int bsi_grad_sqnorm(const float* g, size_t n, float* out_sq, void* workspace, bsi_stream_t stream);
/* g' = grad_scale*g (e.g. 1/world after a sum all-reduce); clip: g' *= min(1, max_norm/(grad_scale*sqrt(sqnorm)+1e-6))
 * (max_norm <= 0: off); AdamW step `step` (1-based) with decoupled weight decay; EMA: ema_weight in [0,1) -> ema +=
 * ema_weight*(p-ema), ema_weight >= 1 -> ema = p (warm-up copy), ema_weight < 0 or ema NULL -> untouched. */
int bsi_clip_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, const float* sqnorm,
                       float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, float ema_weight, bsi_stream_t stream);

/* Segment forms for the data-parallel step (bsi_amd/dp.py; replaces DDP's reducer + optimizer hooks, bsi/tasks/bsi.py:163-198, for
 * a reduce-scatter / sharded-update / all-gather step as well as the all-reduce step).  A segment is one (bucket, rank) slice of
 * the flat buffers: parameters / moments / EMA at element offset p_off, its gradient at g_off of the gradient buffer (the flat
 * gradient, or the rank's compact reduce-scatter output), len elements (a multiple of 4, offsets too), cut into chunks of
 * BSI_SQNORM_CHUNK elements.  my_chunk = chunks of the table's earlier segments (ascending), out_chunk = index of the segment's
 * first chunk in the GLOBAL partial array, which is laid out for the segments of ALL ranks: bsi_sqnorm_segments writes one fp32
 * partial per chunk of the table, bsi_sqnorm_finish sums `nchunks` partials in index order -- the same bits whether a chunk was
 * computed by every rank (all-reduced gradient) or by its owner only (partials of the other ranks added as exact zeros).
 * Table in DEVICE memory.  bsi_clip_adamw_ema_segments = bsi_clip_adamw_ema on the table's segments (same arithmetic). */
#define BSI_SQNORM_CHUNK 16384
typedef struct bsi_seg {
    size_t p_off, g_off, len, my_chunk, out_chunk;
} bsi_seg;
/* n fp32 copies (src -> dst, len elements, non-overlapping) in ONE launch: descriptor table in DEVICE memory, tile0 = the sum of
 * bsi_copy_batch_tiles(len) over the descriptors in front (ascending), tiles = the total.  The UNet backward uses it to put the stacked
 * FiLM gradients and the shared conv2 / skip bias gradients at their parameters' places in the flat gradient buffer (what autograd's
 * per-parameter accumulation does for residual_block.py:39,41,63). */
typedef struct bsi_copy_desc { const float* src; float* dst; size_t len; unsigned tile0, reserved; } bsi_copy_desc;
int bsi_copy_batch_tiles(size_t len);
int bsi_copy_batch_f32(const bsi_copy_desc* descs /*device*/, int n, int tiles, bsi_stream_t stream);
int bsi_sqnorm_segments(const float* g, const bsi_seg* segs, int nseg, size_t nchunks, float* partials, bsi_stream_t stream);
int bsi_sqnorm_finish(const float* partials, size_t nchunks, float* out_sq, bsi_stream_t stream);
int bsi_clip_adamw_ema_segments(float* p, const float* g, float* m, float* v, float* ema, const bsi_seg* segs, int nseg,
                                size_t nchunks, const float* sqnorm, float max_norm, float grad_scale, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float ema_weight, bsi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): per-kernel-class timing with HIP events on the launch stream.
 * ---------------------------------------------------------------------------------------- */
enum {
    BSI_PROF_GEMM_QKV = 0, BSI_PROF_GEMM_OUT = 1, BSI_PROF_GEMM_FC1 = 2, BSI_PROF_GEMM_FC2 = 3,
    BSI_PROF_ATTN = 4, BSI_PROF_LN = 5, BSI_PROF_PROLOGUE = 6, BSI_PROF_FINAL = 7, BSI_PROF_GEMM_ENC = 8,
    BSI_PROF_ADALN = 9, BSI_PROF_NCLASS = 16
};
/* Data-parallel step (replaces what DDP's reducer gets from running on its own stream, /root/reference bsi/tasks/bsi.py:163-166):
 * leave `cus` compute units (0 or a multiple of 8, at most BSI_MAX_CU_RESERVE) out of every persistent kernel's grid -- the
 * GEMMs, weight-gradient GEMMs, convolutions and attention kernels launch one 160-KB-LDS workgroup per CU with a static share
 * of the tiles, so a workgroup whose CU is held by an RCCL kernel would start a whole kernel late.  Per CALLING THREAD (launches of
 * other threads -- an evaluation beside a training step -- keep their own setting), takes effect at that thread's next launch; 0 (the
 * default) = all CUs.  bsi_compute_cus: the CU count those kernels size their grids with. */
#define BSI_MAX_CU_RESERVE 64
int bsi_set_cu_reserve(int cus);
int bsi_compute_cus(void);
/* LayerNorm + modulate passes of the inference engine (one shared modulation row, no dropout): cus > 0 = launch them as PERSISTENT
 * kernels sized for `cus` compute units (every wave prefetches its next row; modulation vectors in LDS), the form
 * bsi_dit_forward_pair uses on its small partition; 0 (default; env BSI_LN_STREAM_CUS) = one row per wave over the whole chip.
 * Bit-identical results.  Per calling thread, takes effect at its next launch. */
int bsi_set_ln_stream_cus(int cus);
/* Tile queue of the persistent bf16 GEMM (K >= 512, more tiles than CUs): 1 = workgroups draw tile tickets from per-XCD counters in
 * device memory (and from the other XCDs' once their own is empty) instead of taking a static share, and the grid ignores the CU
 * reserve: a workgroup whose CU is held by a kernel of another stream leaves its share to the others, and no CU idles while RCCL
 * is quiet.  Same tiles, same arithmetic per tile: results are bit-identical to the static schedule.  Per calling thread, takes
 * effect at its next launch; default 0 (env BSI_TILE_QUEUE overrides).  A launch that is being captured into a HIP graph always
 * takes the static schedule.  DPTrainer(tile_queue=True) applies it to the launches of the backward only. */
int bsi_set_tile_queue(int on);
/* Schedule of the single-sweep attention backward (256 tokens, head dim 64: autograd of dit.py:43-44).  1 (default; env
 * BSI_ATTN_BWD_SKEW=0 overrides): the workgroup's two wave groups run half a trip apart, so that the softmax of one shares a SIMD with
 * the matrix products of the other; 0: both in lock step (the A/B partner).  Same arithmetic in the same order: bit-identical
 * results.  Process-wide, takes effect at the next launch; returns the previous setting. */
int bsi_set_attention_bwd_skew(int on);
/* bit i of mask enables class i; 0 disables.  Events are recorded around every launch of an enabled class
 * made through bsi_dit_forward / bsi_dit_adaln. */
int bsi_prof_enable(unsigned mask);
/* Diagnostic: shader clock seen by a one-wave kernel spinning `us` microseconds: out = {shader cycles, 100 MHz ticks}. */
int bsi_clock_probe(unsigned long long* out /*device*/, int us, bsi_stream_t stream);
/* Box yardstick: one 512-thread workgroup per CU runs iters x 64 v_mfma_f32_16x16x32_bf16 per wave on register operands with
 * hashed (random) values -- the GEMMs' instruction without LDS, memory or barriers.  workspace (bsi_mfma_probe_workspace_bytes,
 * device): after the launch, for wave v of workgroup w, words [4 (8 w + v) ..] = {shader cycles, 100 MHz ticks, first tick, last
 * tick} of its loop (the constant 100 MHz counter is chip-wide).  Returns the workgroup count and the FLOP each one executed:
 * TFLOP/s = workgroups * flop / ((max last tick - min first tick) * 10 ns), MHz = 100 * cycles / ticks of a wave. */
size_t bsi_mfma_probe_workspace_bytes(void);
int bsi_mfma_probe(int iters, void* workspace, int* workgroups, double* flop_per_workgroup, bsi_stream_t stream);
/* Waits for the recorded launches of `cls`, returns their count and summed duration, and clears them. */
int bsi_prof_read(int cls, int* count, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* BSI_HIP_H */
