"""CPU test of the host mirror's SFM.ini `params (...)` line (DispModule.cpp:896-918,
1054-1071): built by `make -C stereo-vision_amd/host`, pure C++ (no GPU, no YARP)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereo-vision_amd", "host")


def test_params_line_round_trip():
    exe = os.path.join(HOST, "sv_params_io_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", HOST, "sv_params_io_test"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=30)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sv_params_io ok" in out.stdout
