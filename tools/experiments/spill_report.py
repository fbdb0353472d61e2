#!/usr/bin/env python
"""Where do a HIP file's register spills execute?  Compiles <file.hip> to gfx950 assembly and prints, per kernel, the innermost
loop that contains MFMAs with its count of scratch (VGPR spill) accesses and v_readlane / v_writelane (SGPR spill) instructions.
Spills outside that loop cost once per tile; inside it they sit on the critical path of every K step (a convolution kernel with
6 scratch loads per step ran 27 % slower than the same kernel without).  usage: spill_report.py bsi_amd/csrc/conv_igemm.hip"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                       src, "-o", out], stderr=subprocess.DEVNULL)
text = open(out).read()
for f in re.split(r"\n(?=_Z\w+:)", text):
    name = f.split(":")[0]
    lines = f.split("\n")
    if not any("v_mfma" in l for l in lines):
        continue
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            loops.append((labels[m.group(1)], i))
    def count(a, b, pat):
        return sum(1 for l in lines[a:b + 1] if re.search(pat, l))
    cand = [(b - a, a, b) for a, b in loops if count(a, b, "v_mfma")]
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
    total_sc, total_ln = count(0, len(lines) - 1, "scratch_"), count(0, len(lines) - 1, "v_(read|write)lane")
    if not cand:
        print(f"{demangled}: no loop with MFMAs; scratch {total_sc}, lane ops {total_ln}")
        continue
    nmin = min(count(a, b, "v_mfma") for _, a, b in cand)
    print(f"{demangled}   | whole kernel: scratch {total_sc}, lane ops {total_ln}")
    seen = set()
    for _, a, b in sorted(cand):
        if count(a, b, "v_mfma") != nmin or any(a <= x <= b for x in seen):  # leaf loops only (one unrolled K step / tap sequence)
            continue
        seen.add(a)
        valu, salu = count(a, b, r"^\s+v_(?!mfma)"), count(a, b, r"^\s+s_(?!waitcnt|barrier|nop)")
        print(f"    MFMA loop of {b - a} lines: {nmin} MFMAs, scratch {count(a, b, 'scratch_')}, read/writelane "
              f"{count(a, b, 'v_(read|write)lane')}, VALU {valu}, SALU {salu}")
