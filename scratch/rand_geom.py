"""Runs the cases of tests/test_gpu_parity.py::test_random_geometries_all_variants one by one and reports each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import svhip, svref
from svhip import synth
from test_gpu_parity import _params
rng = np.random.default_rng(20240611)
n = 0
while n < 28:
    D = int(rng.choice([32, 64, 96, 128, 160, 224, 256]))
    bs = int(rng.choice([1, 3, 5, 7, 9, 11]))
    paths = int(rng.choice([5, 8]))
    flags = int(rng.choice([svhip.FLAGS_REPO, svhip.FLAGS_HH]))
    W = D + bs + int(rng.integers(0, 140))
    H = bs + int(rng.integers(0, 150))
    cap = int(rng.choice([15, 31, 63]))
    uniq = int(rng.choice([0, 5, 15, 40]))
    P1, P2 = 8 * bs * bs, 32 * bs * bs
    if (2 * cap + 63) * bs * bs + P2 > 32767:
        continue
    n += 1
    what = "W=%d H=%d D=%d bs=%d paths=%d flags=%#x cap=%d uniq=%d" % (W, H, D, bs, paths, flags, cap, uniq)
    if len(sys.argv) > 1 and sys.argv[1] == "list":
        print(n, what); continue
    left, right, _ = synth.stereo_pair(1000 + n, W, H, D)
    ph, po = _params(D, paths, flags=flags, block_size=bs, P1=P1, P2=P2, pre_filter_cap=cap, uniqueness_ratio=uniq)
    ref = svref.sgbm(left, right, po)
    m = svhip.Matcher(W, H, ph)
    for mode in (0, 4, 5, 3):
        m.set_debug(mode)
        try:
            out = m.compute_gray(left, right)
            r = "ok" if np.array_equal(out, ref) else "MISMATCH %d px" % int((out != ref).sum())
        except svhip.SvError as e:
            r = "ERROR %s" % e
        print(n, what, "mode", mode, r, flush=True)
    m.close()
