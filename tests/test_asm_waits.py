"""Static check of the hand-counted vector-memory waits (tools/check_asm_loads.py): in the gfx950 code hipcc generates, nothing may
touch the destination registers of an inline-asm global load before the wait statement that names them.  Runs on the CPU
(hipcc cross-compiles); a failure here is a data race the GPU tests may or may not catch."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "tools", "check_asm_loads.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,loads", [("gemm_bf16.hip", 32), ("conv_igemm.hip", 128), ("attention_persist.hip", 16),
                                       ("attention_bwd.hip", 100), ("gemm_tn.hip", 40), ("attention_bwd_x.hip", 100)])  # round 3 / 4: asm fragment loads, transposed reads
def test_inline_asm_loads_are_waited_for(src, loads):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "bsi_amd", "csrc", src), "-o", out], stderr=subprocess.DEVNULL)
    problems, n_loads, n_waits = lint.check(open(out).read())
    os.remove(out)
    assert n_loads >= loads and n_waits > 0, (n_loads, n_waits)  # the scan really saw the asm blocks
    assert not problems, "\n".join(problems[:10])


def test_lint_reports_a_copy_in_front_of_the_wait():
    """The pattern hipcc produced once: the loaded registers copied in front of the wait of one branch arm."""
    asm = """_Z1kv:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[0:1], off
	;;#ASMEND
	s_cbranch_vccnz .LBB0_2
	v_mov_b64_e32 v[8:9], v[4:5]
	;;#ASMSTART
	s_waitcnt vmcnt(0) ; data of v[8:11]
	;;#ASMEND
	s_branch .LBB0_3
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(2) ; data of v[4:7]
	;;#ASMEND
	v_add_f32_e32 v12, v4, v5
.LBB0_3:
	s_endpgm
"""
    problems, n_loads, n_waits = lint.check(asm)
    assert n_loads == 1 and n_waits == 2
    assert any("v_mov_b64_e32" in p and "before its wait" in p for p in problems), problems
    assert any("no inline-asm load writes" in p for p in problems), problems
    assert not any("v_add_f32" in p for p in problems), problems
