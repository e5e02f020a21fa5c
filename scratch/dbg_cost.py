"""Where the cost volume of one geometry differs from the oracle's: dbg_cost.py W H D bs cap flags seed"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "stereo-vision_amd")); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import svhip, svref
from svhip import synth
from test_gpu_parity import _params
W, H, D, bs, cap, flags, seed = [int(x, 0) for x in sys.argv[1:8]]
left, right, _ = synth.stereo_pair(seed, W, H, D)
ph, po = _params(D, 8, flags=flags, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, pre_filter_cap=cap)
ref, st = svref.sgbm(left, right, po, stages=("C",))
m = svhip.Matcher(W, H, ph)
m.set_debug(True)
for rep in range(3):
    got = m.compute_gray(left, right)
    C = m.debug_volume(0)
    bad = C != st["C"]
    print("rep", rep, "C shape", C.shape, "bad", int(bad.sum()), "final equal", np.array_equal(got, ref))
    if bad.any():
        ys, xs, ds = np.nonzero(bad)
        print(" rows", np.unique(ys)[:40], "\n cols", np.unique(xs)[:40], "\n d", np.unique(ds)[:40])
        y, x, d = ys[0], xs[0], ds[0]
        print(" first", y, x, d, "got", C[y, x, :8], "ref", st["C"][y, x, :8])
        print(" bad per row", [(int(r), int(bad[r].sum())) for r in range(min(H, 40))])
        R = st["C"]
        for yy in np.unique(ys)[:6]:
            print("  row", yy, "matches ref row", [dy for dy in range(-4, 5) if 0 <= yy + dy < H and np.array_equal(C[yy], R[yy + dy])],
                  "bad cols", np.unique(np.nonzero(bad[yy])[0])[:12], "...", int(bad[yy].any(axis=1).sum()), "of", C.shape[1])
