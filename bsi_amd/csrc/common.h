// Shared device/host helpers for the BSI gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsi_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Status/err plumbing: every C-ABI entry returns 0 on success, a negative BSI_E* code otherwise,
// and records a message retrievable with bsi_last_error().
void bsi_set_error(const char* fmt, ...);

#define BSI_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            bsi_set_error(__VA_ARGS__);          \
            return BSI_EINVAL;                   \
        }                                        \
    } while (0)

#define BSI_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            bsi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return BSI_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

// Laboratory ablations (kernel parts switched off: results are WRONG) exist only in a laboratory build of the library
// (make -C bsi_amd/csrc LAB=1 OUTDIR=...): in the product build BSI_ABL() is the constant 0, the branches vanish from the hot loops
// and the setters refuse those bits.  Kernel-CHOICE bits (which of two correct kernels runs) are ordinary run-time flags.
#ifdef BSI_LAB
#define BSI_ABL(flags, bits) ((flags) & (bits))
#else
#define BSI_ABL(flags, bits) 0
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, device): remember the pairs already set (round 1 kept one
// flag per process, which left the kernels of a second device without the attribute).  A lost race only repeats the call.
inline void set_max_lds(const void* kern, int bytes) {
    struct Seen { const void* k; int dev; };
    static Seen seen[512];
    static int n_seen = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const int n = n_seen;
    for (int i = 0; i < n; ++i)
        if (seen[i].k == kern && seen[i].dev == dev) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (n < 512) {
        seen[n] = Seen{kern, dev};
        n_seen = n + 1;
    }
}

// Compute units of the CURRENT device (persistent kernels launch one workgroup per CU).
inline int device_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// Compute units the persistent kernels may fill: device_cus() minus the CUs reserved for kernels of another stream that must
// be able to run BESIDE them -- the RCCL all-reduce of the data-parallel step (bsi_set_cu_reserve, bsi_amd/dp.py).  Every
// kernel of this library that launches "one workgroup per CU" with a static partition of its tiles sizes its grid (and the
// split counts derived from it) with this: a 160-KB-LDS workgroup that cannot be placed because a communication kernel holds
// its CU would otherwise start a whole kernel late and double the launch's duration.
// (thread_local: a setting belongs to the thread that launches -- an evaluation or a sampling call on another thread keeps its own)
extern thread_local int g_bsi_cu_reserve;  // prof.hip
extern thread_local int g_bsi_cu_masked;   // prof.hip: 1 while launches go to a CU-masked stream (bsi_dit_forward_pair): the reserve is then a HARD limit --
                              // a workgroup beyond compute_cus() cannot be placed beside the others, tile queue or not
extern thread_local int g_bsi_ln_stream_cus;  // prof.hip: > 0 = the LayerNorm passes of the inference engine run as persistent kernels sized for this many CUs
inline int compute_cus() {
    const int c = device_cus() - g_bsi_cu_reserve;
    return c < 8 ? 8 : c;
}

// Tile queue of the persistent GEMM (gemm_bf16.hip, DYN instances): instead of a static share of the tiles a workgroup draws
// tickets from per-XCD counters in device memory, so a workgroup that starts late -- its CU held by a communication kernel of
// another stream (the RCCL all-reduce of the data-parallel step) -- leaves its share to the others instead of running it alone
// at the end.  Layout of a control block (unsigned words, zero between launches: the last workgroup to leave resets it):
// [0..7] next ticket of XCD x, [8..15] workgroups of XCD x that have left, [16] XCDs whose workgroups have all left (leaving is
// counted in two levels: 256 returning atomics on ONE word serialise into a 10-20 us tail of the kernel), [32 + 16 * blockIdx.x]
// this workgroup's mailbox (one 64-byte line: wave 0 draws the ticket, the other waves read it there).  One block per stream
// (kernels of a stream are serialised).
constexpr int BSI_TQ_GONE = 8, BSI_TQ_XCDS = 16, BSI_TQ_MBOX = 32, BSI_TQ_MAX_WG = 512;
constexpr size_t BSI_TQ_WORDS = BSI_TQ_MBOX + 16 * (size_t)BSI_TQ_MAX_WG;
extern thread_local int g_bsi_tile_queue;       // prof.hip: 0 = static shares, 1 = tickets where a kernel supports them
unsigned* bsi_tile_queue_block(hipStream_t s);  // prof.hip: the stream's control block, or nullptr (queue off / none available)
#ifdef BSI_LAB
extern unsigned* g_lab_static_block;            // prof.hip (laboratory build): where the static schedule leaves its stamps
#endif

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}

// Round-to-nearest-even fp32 -> bf16 through the hardware conversion (keeps NaN a NaN).
__device__ __forceinline__ __bf16 f32_to_bf16(float x) { return (__bf16)x; }

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return *reinterpret_cast<unsigned int*>(&v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave-wide sums on the DPP / lane-swap path (no LDS traffic, ~10x shorter than the __shfl_xor chain): quad permutes and
// row mirrors sum each 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950) combine the four rows.  Every lane ends
// with the total.  The association differs from wave_sum, so results agree with it to fp32 rounding, not bit for bit.
template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_xadd<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_xadd<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_xadd<0x141>(v);  // row_half_mirror
    v = dpp_xadd<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float rows_sum(float v) {  // sum over the four 16-lane rows of values that are uniform per row
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float wave_sum_fast(float v) { return rows_sum(row16_sum(v)); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// tanh-approximated GELU, 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3))) == x * sigmoid(2u) == x / (1 + 2^(-2u*log2 e)):
// 2 transcendentals (v_exp_f32, v_rcp_f32), ~1e-6 relative error.  (A transcendental-free packed-fp32 polynomial form was
// measured in the fc1 epilogue: no gain -- the epilogue's cost is its 16 store instructions on the VMEM path)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2*sqrt(2/pi)*log2(e)
    const float b = a * 0.044715f;
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(x * __fmaf_rn(b, x2, a));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// The same arithmetic on two values at once (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: five packed instructions per pair instead
// of six scalar ones per value; the two transcendentals stay scalar).  Bit-identical to gelu_tanh_f per element.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ gelu_tanh_f2(f32x2_ x) {
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float b = a * 0.044715f;
    const f32x2_ x2 = x * x;
    const f32x2_ t = __builtin_elementwise_fma(x2, f32x2_{b, b}, f32x2_{a, a});
    const f32x2_ arg = x * t;
    const f32x2_ d = f32x2_{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + f32x2_{1.0f, 1.0f};
    return x * f32x2_{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// d/dx of the tanh-GELU: with s = sigmoid(2u), u = k0*(x + k1 x^3):  gelu = x*s,  gelu' = s + x*s*(1-s)*2*k0*(1+3*k1*x^2)
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float a = -2.0f * k0 * 1.4426950408889634f;
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(x * __fmaf_rn(a * k1, x2, a));
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return __fmaf_rn((x * s) * (1.0f - s), __fmaf_rn(6.0f * k0 * k1, x2, 2.0f * k0), s);
}
// two values at once, bit-identical per element (see gelu_tanh_f2)
__device__ __forceinline__ f32x2_ gelu_tanh_grad_f2(f32x2_ x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float a = -2.0f * k0 * 1.4426950408889634f;
    const f32x2_ x2 = x * x;
    const f32x2_ arg = x * __builtin_elementwise_fma(x2, f32x2_{a * k1, a * k1}, f32x2_{a, a});
    const f32x2_ d = f32x2_{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + f32x2_{1.0f, 1.0f};
    const f32x2_ s = f32x2_{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_ w = __builtin_elementwise_fma(x2, f32x2_{6.0f * k0 * k1, 6.0f * k0 * k1}, f32x2_{2.0f * k0, 2.0f * k0});
    return __builtin_elementwise_fma((x * s) * (f32x2_{1.0f, 1.0f} - s), w, s);
}

// 4 x 4 transpose of r[0..3] across the four 16-lane groups of a wave: afterwards lane group g holds in r[s] what lane group s
// held in r[g].  v_permlane32_swap exchanges (vdst lanes 32..63) <-> (vsrc lanes 0..31), v_permlane16_swap exchanges
// (vdst odd groups) <-> (vsrc even groups).
__device__ __forceinline__ void transpose_lane_groups(f32x4 (&r)[4]) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const u2v t = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[pr][d]), __float_as_uint(r[pr + 2][d]), false, false);
            r[pr][d] = __uint_as_float(t[0]);
            r[pr + 2][d] = __uint_as_float(t[1]);
        }
#pragma unroll
    for (int pr = 0; pr < 4; pr += 2)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const u2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[pr][d]), __float_as_uint(r[pr + 1][d]), false, false);
            r[pr][d] = __uint_as_float(t[0]);
            r[pr + 1][d] = __uint_as_float(t[1]);
        }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// Counter-based dropout (training only): element `idx` of site `seed` is kept iff hash(seed, idx) >= p * 2^32.
// The same function is evaluated in the forward and the backward kernels, so no mask is stored.
struct DropCfg {
    unsigned thr;      // p * 2^32 (0 = dropout off)
    unsigned s0, s1;   // seed words
    float scale;       // 1 / (1 - p)
};
__host__ __device__ __forceinline__ unsigned bsi_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ bool drop_keep_rc(const DropCfg& c, unsigned rowh, unsigned col);
__host__ __device__ __forceinline__ bool drop_keep(const DropCfg& c, unsigned long long idx) {  // (row, column) = (high, low) word
    return drop_keep_rc(c, bsi_mix32((unsigned)(idx >> 32) + c.s0) ^ c.s1, (unsigned)idx);
}
// Elements are addressed as (row, column) = the (high, low) words of the index: the inner hash depends on the row only, so
// kernels hoist it per row (drop_row) and pay one cheap mix per aligned quad of columns (drop_keep_rc) with no 64-bit arithmetic.
__host__ __device__ __forceinline__ unsigned drop_row(const DropCfg& c, unsigned row) { return bsi_mix32(row + c.s0) ^ c.s1; }
// One hash serves FOUR neighbouring columns (an aligned quad): 64 bits = four 16-bit fields, each compared with the 16-bit
// threshold; the kernels visit columns in aligned quads (a lane's MFMA registers, a float4 of an elementwise pass), so the
// compiler shares the hash.  The mixer uses 24-bit multiplies only (v_mad_u32_u24: full rate; the two 32-bit multiplies of
// bsi_mix32 are quarter rate on the vector ALU and were most of what dropout cost the attention kernels, round 3: forward 240 ->
// 373 us, backward 873 -> 1062 us): 13 full-rate instructions per quad.  Statistics (tools/experiments/drop_hash_stats.py):
// every input bit flips 16.0-17.3 of the 32 output bits of either word; fields uniform (chi-square 242-298 for 255 dof);
// neighbouring columns / rows uncorrelated (|r| < 1e-3); row and column drop counts dispersed as the binomial.  p is resolved to 2^-16.
struct DropQuad { unsigned h1, h2; };
__host__ __device__ __forceinline__ unsigned bsi_mad24(unsigned a, unsigned b24, unsigned c) { return (a & 0xffffffu) * b24 + c; }
__host__ __device__ __forceinline__ DropQuad drop_quad(unsigned rowh, unsigned quad) {
    const unsigned x = quad ^ rowh;
    unsigned t = bsi_mad24(x, 0x9E3779u, x >> 12);
    t ^= t >> 15;
    t = bsi_mad24(t, 0x85EBCBu, t >> 10);
    DropQuad q;
    q.h1 = t ^ (t >> 14);
    const unsigned u = bsi_mad24(q.h1, 0xC2B2AFu, q.h1 >> 6);
    q.h2 = u ^ (u >> 13);
    return q;
}
__host__ __device__ __forceinline__ unsigned drop_field(const DropQuad& q, unsigned col) {  // the 16-bit field of column col (its low 2 bits)
    const unsigned h = (col & 2u) ? q.h2 : q.h1;
    return (col & 1u) ? (h >> 16) : (h & 0xffffu);
}
__host__ __device__ __forceinline__ bool drop_keep_rc(const DropCfg& c, unsigned rowh, unsigned col) {
    return drop_field(drop_quad(rowh, col >> 2), col) >= (c.thr >> 16);
}
#ifdef __HIPCC__
// Dropout-mask WORDS of the attention weights (256 tokens, head dim 64; consumers: attention_persist.hip, attention_bwd.hip).
// Block `wid` = pair * 16 + qb covers queries 16 qb .. 16 qb + 15 of (batch, head) pair `pair` against all 256 keys = 64 words:
// word 4 kt + r, bit 16 g + c = keep(row = pair * 256 + 16 qb + c, column = 16 kt + 4 g + r) -- the lane <-> (query, key) map of
// the S^T = K.Q^T accumulators.  One wave computes a block: a lane's four keys of a key block are an aligned quad of mask columns
// = one hash, the four comparisons' lane masks are the words; lane 4 kt + r returns word (kt, r).
__device__ __forceinline__ unsigned long long drop_mask_block(const DropCfg& dc, unsigned wid, int lane) {
    const int g = lane >> 4, c16 = lane & 15;
    const unsigned rowh = drop_row(dc, wid * 16u + (unsigned)c16);
    const unsigned thr16 = dc.thr >> 16;
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
        const DropQuad dq = drop_quad(rowh, (unsigned)(4 * kt + g));
        unsigned long long w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = __builtin_amdgcn_ballot_w64(drop_field(dq, (unsigned)r) >= thr16);
        // v_writelane: the four scalar words into lanes 4 kt .. 4 kt + 3 (2 instructions per word; a compare + two selects otherwise).
        // ONE statement per key block, opened by wait states: a v_writelane that reads a scalar register right behind the v_cmp
        // that wrote it got the OLD value (5 % of the mask bits wrong with one statement per word) -- hipcc pads nothing inside asm.
        asm("s_nop 4\n\t"
            "v_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\t"
            "v_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
            "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\t"
            "v_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
            : "+v"(lo), "+v"(hi)
            : "s"((unsigned)w[0]), "s"((unsigned)(w[0] >> 32)), "s"((unsigned)w[1]), "s"((unsigned)(w[1] >> 32)), "s"((unsigned)w[2]),
              "s"((unsigned)(w[2] >> 32)), "s"((unsigned)w[3]), "s"((unsigned)(w[3] >> 32)), "n"(4 * kt), "n"(4 * kt + 1), "n"(4 * kt + 2),
              "n"(4 * kt + 3));
    }
    return ((unsigned long long)hi << 32) | lo;
}
#endif

inline DropCfg make_drop(float p, unsigned long long seed, unsigned site) {
    DropCfg c{};
    if (p <= 0.f) return c;
    const unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(site + 1);
    double t = (double)p * 4294967296.0 + 32768.0;  // the kernels compare 16-bit hash halves with thr >> 16: round p to 2^-16
    c.thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (c.thr < 65536u) c.thr = 65536u;            // dropout ON means at least 2^-16
    c.s0 = bsi_mix32((unsigned)s);
    c.s1 = bsi_mix32((unsigned)(s >> 32) ^ 0x85ebca6bu);
    // the survivors' scale follows the QUANTISED probability the kernels apply (thr >> 16 of 65536), so that E[dropout(x)] = x
    // exactly; p >= 1 is rejected at the entry points (a scale of 1 / 0)
    if ((c.thr >> 16) > 65535u) c.thr = 65535u << 16;
    c.scale = 65536.0f / (float)(65536u - (c.thr >> 16));
    return c;
}
