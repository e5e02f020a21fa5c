"""Ad-hoc: extreme geometries (very tall, very wide, large ragged) through every aggregation variant vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import svhip, svref
from svhip import synth
cases = [(200, 8000, 32), (60, 20000, 32), (9000, 40, 32), (30000, 12, 64), (1283, 721, 160), (2049, 1537, 256), (40000, 8, 32), (40001, 8, 32), (40, 65535, 32)]
for W, H, D in cases:
    t = time.time()
    left, right, _ = synth.stereo_pair(7, W, H, D)
    ph = svhip.default_params(D, 8)
    po = svref.Params()
    for f, _t in svhip.Params._fields_: setattr(po, f, getattr(ph, f))
    ref = svref.sgbm(left, right, po)
    try:
        m = svhip.Matcher(W, H, ph)
    except svhip.SvError as e:
        print(W, H, D, "create:", e, flush=True); continue
    for mode in (0, 4, 5, 3, 2):
        m.set_debug(mode)
        try:
            out = m.compute_gray(left, right)
            r = "ok" if np.array_equal(out, ref) else "MISMATCH %d px" % int((out != ref).sum())
        except svhip.SvError as e:
            r = "ERROR %s" % e
        print("%dx%d D=%d mode %d: %s  agg=%s" % (W, H, D, mode, r, m.last_aggregation()), flush=True)
    m.close()
    print("   (%.1f s)" % (time.time() - t), flush=True)
