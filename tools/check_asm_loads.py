#!/usr/bin/env python
"""Lint for the hand-counted vector-memory waits of the HIP kernels.

Several epilogues fetch rows with INLINE-ASM global loads and wait for them with an inline-asm `s_waitcnt vmcnt(N)` that carries the
loaded registers as "+v" operands (gemm_bf16.hip wave_tile_epilogue_train, conv_igemm.hip store_f32_rows, attention_persist.hip,
attention_bwd.hip attention_bwd_p_kernel, gemm_tn.hip gemm_tn2_kernel):
hipcc's own waits would be vmcnt(0), it assumes loads and stores can complete out of order with each other.  The compiler does not
know that the registers are not valid between the load and the wait.  The "+v" operands keep their USES behind the wait, but hipcc may
still COPY such a register in front of the wait (it did, when two waits sat in the arms of an if / else).  This script compiles the
given .hip files to gfx950 assembly and reports, per kernel, every instruction outside an asm block that touches the destination
registers of an inline-asm load before a wait statement names them (`s_waitcnt vmcnt(N) ; data of v[..] v[..]`): forward data-flow
over the function's basic blocks (labels, s_branch / s_cbranch edges), so loads carried around a loop are followed.

usage: check_asm_loads.py file.hip [file.hip ...]      exit status 1 if anything was found"""
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^s_(c?branch\w*)\s+(\.LBB\d+_\d+)")


def blocks_of(func):
    """[(label or None, [(line number, kind, payload)]), ...], successors by index: basic blocks of one function in text order."""
    blocks, cur, in_asm = [], [None, []], False
    for ln, line in enumerate(func.split("\n")):
        code = line.strip()
        m = LABEL.match(code)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        if code.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if code.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not code or code.startswith((";", ".")) or code.endswith(":"):
            continue
        if in_asm:
            if (code.startswith(("global_load", "buffer_load")) and " lds" not in code) or \
                    (code.startswith("global_atomic") and code.rstrip().endswith("sc0")):  # returning atomics write their first operand like a load
                cur[1].append((ln, "load", (regs_of(code.split(",")[0]), "vm")))
            elif code.startswith("ds_read"):  # inline-asm LDS reads (transposed reads the compiler must not wait vmcnt(0) for)
                cur[1].append((ln, "load", (regs_of(code.split(",")[0]), "lgkm")))
            elif code.startswith("s_waitcnt") and "; data of" in code:
                cur[1].append((ln, "wait", (regs_of(code.split("; data of")[1]), code)))
            elif code.startswith("s_waitcnt") and ("vmcnt(0)" in code or "lgkmcnt(0)" in code):
                cur[1].append((ln, "drain", "vm" if "vmcnt(0)" in code else "lgkm"))
            continue
        body = code.split(";")[0].strip()
        if body.startswith("s_waitcnt"):  # a full drain the compiler wrote is as good as one of ours
            for cnt, kind in (("vmcnt(0)", "vm"), ("lgkmcnt(0)", "lgkm")):
                if cnt in body:
                    cur[1].append((ln, "drain", kind))
            continue
        m = BRANCH.match(body)
        if m:
            cur[1].append((ln, "branch", (m.group(1), m.group(2))))
            if m.group(1) == "branch":  # unconditional: the text that follows starts a new block without a fall-through edge
                blocks.append(cur)
                cur = [None, []]
                cur.append("nofall")
            continue
        if body.startswith("s_endpgm"):
            cur[1].append((ln, "end", None))
            blocks.append(cur)
            cur = [None, []]
            cur.append("nofall")
            continue
        cur[1].append((ln, "inst", (regs_of(body), body)))
    blocks.append(cur)
    return blocks


RESERVED = re.compile(r"\bv255\b|\bv\[(\d+):255\]")


def check_reserved_register(name, func):
    """Kernels whose inline asm lands vector-memory results in the FIXED register v255 (gemm_bf16.hip, the tile queue's ticket) rely on
    the compiler never naming that register itself (it is a clobber of those statements, not an operand): report every instruction
    outside the asm blocks that does."""
    lines = func.split("\n")
    in_asm, uses_fixed, out = False, False, []
    for ln, line in enumerate(lines):
        code = line.strip()
        if code.startswith(";;#ASMSTART"):
            in_asm = True
        elif code.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            uses_fixed |= bool(re.match(r"(global_atomic\w*|global_load_dword\w*)\s+v255\b", code))
        elif code and not code.startswith((";", ".")) and RESERVED.search(code.split(";")[0]):
            out.append(f"{name}: line {ln}: `{code}` names v255 outside the asm statements that reserve it")
    return out if uses_fixed else []


def check(asm_text):
    problems, n_loads, n_waits = [], 0, 0
    for func in re.split(r"\n(?=_Z\w+:)", asm_text):
        name = func.split(":")[0]
        problems += check_reserved_register(name, func)
        blocks = blocks_of(func)
        index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
        succ = [set() for _ in blocks]
        for i, b in enumerate(blocks):
            ends = False
            for _, kind, payload in b[1]:
                if kind == "branch":
                    if payload[1] in index:
                        succ[i].add(index[payload[1]])
                    ends = payload[0] == "branch"
                elif kind == "end":
                    ends = True
            if not ends and i + 1 < len(blocks) and not (len(blocks[i + 1]) > 2):
                succ[i].add(i + 1)
        # Two forward analyses over the blocks.  MUST (intersection at joins): registers an inline-asm load is still writing on
        # EVERY path into a block -- an instruction that touches one of them is a bug (the conditions of `if (has_res) load` ...
        # `if (has_res) wait` are correlated, a union at the joins would report the infeasible load-without-wait path).  MAY (union):
        # kept for symmetry; a wait statement must name registers that SOME inline-asm load of the function writes (hipcc copied them
        # otherwise).
        def transfer(i, pend, report, may):
            pend = dict(pend)
            for ln, kind, payload in blocks[i][1]:
                if kind == "load":
                    for r in payload[0]:
                        pend[r] = (ln, payload[1])
                elif kind == "drain":
                    for r in [r for r, (_, k) in pend.items() if k == payload]:
                        del pend[r]
                elif kind == "wait":
                    named, code = payload
                    if report and may and any(r not in ever_loaded for r in named):
                        problems.append(f"{name}: line {ln}: wait names registers no inline-asm load writes (copied?): {code}")
                    for r in named:
                        pend.pop(r, None)
                elif kind == "inst" and report and not may:
                    hit = payload[0] & set(pend)
                    if hit:
                        r = sorted(hit)[0]
                        problems.append(f"{name}: line {ln}: `{payload[1]}` touches v{r} (inline-asm load at line {pend[r][0]}) before its wait")
            return pend

        ever_loaded = set()
        for b in blocks:
            for _, kind, payload in b[1]:
                if kind == "load":
                    ever_loaded |= payload[0]
        pred = [set() for _ in blocks]
        for i in range(len(blocks)):
            for j in succ[i]:
                pred[j].add(i)
        for may in (False, True):
            out = [None] * len(blocks)  # None = not computed yet (top)
            inn = [dict() for _ in blocks]
            changed = True
            while changed:
                changed = False
                for i in range(len(blocks)):
                    known = [out[p] for p in pred[i] if out[p] is not None]
                    if i == 0 or not pred[i]:
                        new_in = {}
                    elif not known:
                        continue
                    elif may:
                        new_in = {}
                        for o in known:
                            new_in.update(o)
                    else:
                        new_in = {r: v for r, v in known[0].items() if all(r in o for o in known[1:])}
                    new_out = transfer(i, new_in, False, may)
                    if out[i] is None or set(new_out) != set(out[i]) or set(new_in) != set(inn[i]):
                        out[i], inn[i] = new_out, new_in
                        changed = True
            for i in range(len(blocks)):
                transfer(i, inn[i], True, may)
        for b in blocks:
            n_loads += sum(1 for _, k, _p in b[1] if k == "load")
            n_waits += sum(1 for _, k, _p in b[1] if k == "wait")
    return problems, n_loads, n_waits


def main(files):
    bad = 0
    for src in files:
        out = tempfile.mktemp(suffix=".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        problems, n_loads, n_waits = check(open(out).read())
        print(f"{src}: {n_loads} inline-asm loads, {n_waits} wait statements, {len(problems)} problems")
        for p in problems[:20]:
            print("   ", p)
        bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
