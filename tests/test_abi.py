"""CPU tests of the drop-in boundary: libsv_hip.so loads without a GPU, exports every symbol that
include/sv_hip.h declares, validates parameters, and fails loudly (no CPU fallback) when asked to
compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import svhip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sv_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = svhip.load()
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(svhip.EXPORTS) == names, "python binding list and header disagree"
    assert b"gfx950" in lib.sv_version()


def test_params_struct_layout_matches_header():
    assert C.sizeof(svhip.Params) == 14 * 4
    p = svhip.default_params(128, 8)
    assert (p.num_disparities, p.block_size, p.P1, p.P2) == (128, 7, 392, 1568)
    assert (p.pre_filter_cap, p.uniqueness_ratio, p.disp12_max_diff) == (63, 15, 0)  # DispModule.cpp:369-394
    assert p.path_mask == svhip.PATHS_8 and p.flags == svhip.FLAGS_HH
    q = svhip.default_params(96, 5)
    assert q.path_mask == svhip.PATHS_5 and q.flags == svhip.FLAGS_REPO
    with pytest.raises(svhip.SvError):
        svhip.default_params(96, 6)


def test_null_context_is_an_error_not_a_crash():
    lib = svhip.load()
    assert lib.sv_destroy(None) == -1
    assert lib.sv_set_params(None, None) == -1
    assert lib.sv_compute_gray_dev(None, None, None, None) == -1
    assert lib.sv_last_error(None) == b"null context"


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    with pytest.raises(svhip.SvError) as e:
        svhip.Matcher(640, 480, svhip.default_params(128, 8))
    assert e.value.code == -3 and "device" in str(e.value)


def test_python_binding_has_no_cpu_fallback():
    src = open(os.path.join(ROOT, "stereo-vision_amd", "svhip", "__init__.py")).read()
    assert "svref" not in src and "oracle" not in src.replace("no CPU fallback", "")
    csrc = os.path.join(ROOT, "stereo-vision_amd", "csrc")
    for f in os.listdir(csrc):
        assert "sv_ref" not in open(os.path.join(csrc, f), errors="ignore").read(), f


@pytest.mark.skipif(_gpu_present(), reason="checks the no-device behaviour")
def test_bench_refuses_to_run_without_a_gpu():
    """bench.py has no CPU path to fall back to: without a device it stops with a message, it does not time the oracle."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no HIP device" in (r.stderr + r.stdout)
    assert "frames/s" not in r.stdout
