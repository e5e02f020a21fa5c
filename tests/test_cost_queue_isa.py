"""k_cost2's row loop (gray cost, D <= 128) issues its loads and stores from inline assembly and waits for them by count.  The
compiler does not know that the loads' destination registers are written later than the instruction, so one thing must never
appear in its output: an instruction that copies or overwrites such a register while its load may be in flight (hipcc did emit
such copies for a loop with several exits).  This test compiles those kernels for gfx950 and reads the listing."""
import os
import shutil
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "isa"))
import check_cost_queue

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_copy_of_an_in_flight_queue_register(tmp_path):
    out = str(tmp_path / "cq.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
                           "--cuda-device-only", "-S", "-o", out, os.path.join(HERE, "isa", "cost_queue_tu.hip")],
                          stderr=subprocess.DEVNULL)
    seen, bad = check_cost_queue.check(out)
    assert seen == 8, "expected the 8 gray cost kernels with a queue, found %d" % seen
    assert not bad, "\n".join("%s:%d %s: %s" % b for b in bad)


def test_checker_sees_a_copy(tmp_path):
    listing = """_ZN3svk7k_cost2ILi2ELi512ELi1EEEvX:
	;;#ASMSTART
	global_load_dwordx2 v[30:31], v[16:17], off
	;;#ASMEND
	v_mov_b64_e32 v[54:55], v[30:31]
	v_add_u32_e32 v1, v2, v3
	;;#ASMSTART
	s_waitcnt vmcnt(5)
	;;#ASMEND
	v_mul_u32_u24_e32 v2, v30, v4
	v_mov_b32_e32 v31, 0
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
.Lfunc_end0:
"""
    p = tmp_path / "x.s"
    p.write_text(listing)
    seen, bad = check_cost_queue.check(str(p))
    assert seen == 1 and [b[2] for b in bad] == ["copies a queue register", "writes a queue register"]
