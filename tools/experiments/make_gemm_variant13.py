"""Laboratory generator (round 3): derives `gemm_bf16_k64s_kernel` (variant 13: operand DMA split by wave group) from the text of
`gemm_bf16_k64r_kernel` in bsi_amd/csrc/gemm_bf16.hip and makes it selectable with bsi_gemm_set_variant(13).  Result of the A/B
(tools/experiments/gemm_variant_ab.py, profiles/r3/gemm_variant13_ab.txt): bit-identical outputs, +-1 % -- not kept in the product.
    python tools/experiments/make_gemm_variant13.py && make -C bsi_amd/csrc      # apply;  git checkout bsi_amd/csrc/gemm_bf16.hip  # undo"""
p = 'bsi_amd/csrc/gemm_bf16.hip'
s = open(p).read()
start = s.index("// Variant 12: 128-B tile rows (whole cache lines per DMA row, see variant 10) on a RING OF FIVE HALF-STAGES.")
end = s.index("template <int TM, int WM, int WN, int EPI>\nint launch_cfg(")
n = s[start:end]
n = n.replace("// Variant 12: 128-B tile rows (whole cache lines per DMA row, see variant 10) on a RING OF FIVE HALF-STAGES.",
"""// Variant 13 = variant 12 with the operand DMA split BY WAVE GROUP (round 3, late).  In variant 12 every wave issues its quarter of every
// half-stage and waits for it at the end of its next load phase: for the W half-stage of stage G+1 that is two phases (~1400 cycles)
// after the issue, and the in-kernel stamps (tools/experiments/gemm_lab.hip, LAB_SEG=1) show 135 of a load phase's 650 cycles spent in
// that wait.  Here group A (waves 0-3, the group that reads a stage first) issues the whole W half-stage of stage G+1 in its first
// load phase of stage G and waits for it BEHIND its second MFMA phase (3.5 phases later: the wait runs under the MFMAs), and group B
// (waves 4-7) issues the A half-stage of stage G+2, four instructions in each of its load phases, and waits for it at the end of the
// second load phase of stage G+1 (4 phases later).  Same slots, same LDS images, same arithmetic: bit-identical results.
// (variant 12's description:) 128-B tile rows on a RING OF FIVE HALF-STAGES.""")
n = n.replace("void gemm_bf16_k64r_kernel(const GemmParams p) {", "void gemm_bf16_k64s_kernel(const GemmParams p) {")
old_issue = n[n.index("    // ---- issue stream: half-stages in the order A(0) W(0) A(1) W(1) ... over this workgroup's tiles"):n.index("    const int rho = lane & 15, qd = lane >> 4;")]
new_issue = '''    // ---- issue streams: group A walks the W half-stages W(0) W(1) ..., group B the A half-stages A(0) A(1) ..., each over this
    // workgroup's tiles with its own tile pointer; slot of A(G) = 2G % 5, of W(G) = (2G + 1) % 5 as in variant 12
    const int srow = lane >> 3, spos = lane & 7;
    const int wg = wave & 3;  // index inside the group: instruction (q, wg) of a half-stage moves tile rows 8 (4q + wg) .. + 7
    unsigned goff[8];         // source byte offsets of this lane's chunk for the wave's 8 instructions of a half-stage
    const char* Sb = reinterpret_cast<const char*>(wm == 0 ? p.W : p.A);
    asm volatile("" : "+s"(Sb));
    int s_tile = tile, s_v = 0, s_slot = wm == 0 ? 1 : 0;  // next half-stage of this group's stream and its slot
    auto set_sources = [&](int t) {
        int tm_, tn_;
        tile_coords(p, t, tm_, tn_);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (q * 4 + wg) * 8 + srow;
            if (wm == 0) {
                int n = tn_ * BN + r;
                n = n < p.N ? n : p.N - 1;
                goff[q] = (unsigned)n * (unsigned)(p.ldw * 2) + ((spos ^ (((r >> 1) & 1) | (((r >> 4) & 3) << 1))) << 4);
            } else {
                int m = tm_ * BM + r;
                m = m < p.M ? m : p.M - 1;
                goff[q] = (unsigned)m * (unsigned)(p.lda * 2) + ((spos ^ ((r >> 1) & 7)) << 4);
            }
        }
    };
    auto stream_advance = [&]() {
        s_slot = s_slot >= 3 ? s_slot - 3 : s_slot + 2;
        if (++s_v == nk) {
            s_v = 0;
            s_tile += wpx;
            if (s_tile < hi) set_sources(s_tile);
        }
    };
    // instructions [Q0, Q1) of the stream's current half-stage; false when the stream is exhausted
    auto issue_part = [&](auto Q0_, auto Q1_) -> bool {
        constexpr int Q0 = decltype(Q0_)::value, Q1 = decltype(Q1_)::value;
        if (s_tile >= hi) return false;
        char* base = lds + s_slot * HALF + wg * 1024;
        const char* src = Sb + (size_t)s_v * 128;
#pragma unroll
        for (int q = Q0; q < Q1; ++q) __builtin_amdgcn_global_load_lds(GLB_PTR(src + goff[q]), LDS_PTR(base + q * 4 * 1024), 16, 0, 0);
        if (Q1 == 8) stream_advance();
        return true;
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;

'''
n = n.replace(old_issue, new_issue)
old_pro = n[n.index("    // prologue: A(0), W(0), A(1) in flight; stage 0 must have landed before the first load phase"):n.index("    int sa = 0, sw = 1;  // slots of the A and W halves of the stage being consumed")]
new_pro = '''    // prologue: group A issues W(0); group B issues A(0) and A(1), of which A(1) may stay in flight
    set_sources(tile);
    if (wm == 0) {
        issue_part(I0{}, I8{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_part(I0{}, I8{});
        if (issue_part(I0{}, I8{})) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PHASE_BARRIER();
    if (wm == 1) PHASE_BARRIER();  // group B runs one phase behind

'''
n = n.replace(old_pro, new_pro)
old_loop = n[n.index("    // Epilogue stores and the in-order vmcnt: a DMA half-stage issued AFTER the stores cannot be waited for without draining"):n.index("#undef PHASE_BARRIER")]
new_loop = '''    // Epilogue stores and the in-order vmcnt (as in variant 12): the DMA instructions of the load phase that follows an epilogue are
    // issued IN FRONT of the epilogue's stores, and the first wait behind them leaves the NST stores in flight.
    // Waits.  Group A: behind the second MFMA phase of stage G, for W(G+1) (its 8 instructions of this stage; nothing younger but
    // the stores of an epilogue).  Group B: at the end of the second load phase of stage G, for A(G+1) (issued during stage G-1):
    // the 8 instructions of stage G -- A(G+2) -- may stay in flight (+ the stores).  A stream that has run out waits for everything.
    constexpr int NST = EPI == BSI_EPI_BIAS_GELU_DUAL ? 32 : 16;
    bool after_e = false;     // the next wait may leave the stores of the last epilogue in flight
    bool pre = false, pre_status = false;  // the next load phase's instructions were issued in front of the epilogue
    while (true) {
        const int next = tile + wpx;
        const bool has_next = next < hi;
        init_acc(tile);
        for (int v = 0; v < nk; ++v) {
            const char* ba = lds + sa * HALF;
            const char* bw = lds + sw * HALF;
            bool issued = false;  // this wave issued its share of this stage
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // ---- L(v, ks): DMA issue (unless it went out in front of an epilogue), fragment reads
                if (ks == 0) {
                    if (pre) { issued = pre_status; pre = false; }
                    else issued = wm == 0 ? issue_part(I0{}, I8{}) : issue_part(I0{}, I4{});
                } else if (wm == 1 && issued) {
                    issue_part(I4{}, I8{});
                }
                {
                    const int cx = ((ks * 4 + qd) ^ xkey) << 4, cw = ((ks * 4 + qd) ^ wkey) << 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(bw + wrow + i * 4 * RB + cw);
#pragma unroll
                    for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(ba + xrow + j * 16 * RB + cx);
                }
                if (ks == 1 && wm == 1) {  // group B: A(G+1) has landed
                    if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (BF16_OUT && after_e) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    after_e = false;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PHASE_BARRIER();
                // ---- C(v, ks)
                __builtin_amdgcn_s_setprio(1);
                // boustrophedon over the 4 x TM fragment grid (see variant 12)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int i0 = 0; i0 < 4; ++i0) {
                        const int i = (j & 1) ? 3 - i0 : i0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                if (ks == 1 && wm == 0) {  // group A: W(G+1) has landed -- the wait runs under the MFMAs just issued
                    if (BF16_OUT && after_e && issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    after_e = false;
                }
                if (ks == 1 && v == nk - 1 && wm == 1) {  // group B: before the barrier that ends its last C phase
                    pre_status = issue_part(I0{}, I4{});
                    pre = true;
                    epilogue(tile);
                }
                PHASE_BARRIER();
            }
            sa = sa >= 3 ? sa - 3 : sa + 2;
            sw = sw >= 3 ? sw - 3 : sw + 2;
        }
        if (wm == 0) {  // group A: after that barrier, i.e. at the start of its next load phase
            pre_status = issue_part(I0{}, I8{});
            pre = true;
            epilogue(tile);
        }
        {   // the store allowance of the next wait is valid only if every store of the epilogue is issued (no M / N tail)
            int tm_, tn_;
            tile_coords(p, tile, tm_, tn_);
            after_e = tm_ * 256 + 256 <= p.M && tn_ * 256 + 256 <= p.N;
        }
        if (!has_next) break;
        tile = next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 0) PHASE_BARRIER();
'''
n = n.replace(old_loop, new_loop)
assert "issue_next" not in n
s = s[:end] + n + s[end:]
old = '''    auto kern = gemm_bf16_k64r_kernel<EPI>;
    set_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
'''
new = '''    auto kern = g_variant == 13 ? gemm_bf16_k64s_kernel<EPI> : gemm_bf16_k64r_kernel<EPI>;
    set_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
'''
assert old in s
s = s.replace(old, new, 1)
s = s.replace("if (g_variant == 12 && EpiTraits<EPI>::out_bf16 && p.M > 128", "if ((g_variant == 12 || g_variant == 13) && EpiTraits<EPI>::out_bf16 && p.M > 128", 1)
s = s.replace("BSI_CHECK_ARG(v >= 0 && ((v & 0xff) == 12 || (v & 0xff) == 6),", "BSI_CHECK_ARG(v >= 0 && ((v & 0xff) == 12 || (v & 0xff) == 13 || (v & 0xff) == 6),", 1)
s = s.replace('#include <cstdlib>\n', '#include <cstdlib>\n#include <type_traits>\n', 1)
open(p, 'w').write(s)
print("variant 13 written into", p)
