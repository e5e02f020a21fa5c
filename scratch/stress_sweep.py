"""One-off stress of the fused sweeps' winner-take-all variants (LDS neighbour / uniqueness forms): random geometry, D,
uniqueness ratio incl. 0 / 99 / 100, minDisparity, both workgroup shapes, every frame against the oracle.
   python scratch/stress_sweep.py [cases] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "stereo-vision_amd")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
import svref, svhip
from svhip import synth
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = 0
for case in range(ncase):
    D = int(rng.choice([32, 64, 96, 128, 160, 192, 256]))
    bs = int(rng.choice([3, 5, 7]))
    minD = int(rng.choice([0, 0, 0, 7, -5, 40, -70]))
    W = D + abs(minD) + bs + int(rng.integers(2, 100))
    H = bs + int(rng.integers(1, 150))
    nb = int(rng.integers(2, 7))
    uniq = int(rng.choice([0, 1, 5, 15, 50, 99, 100]))
    paths = int(rng.choice([5, 8]))
    flags = int(rng.choice([svhip.FLAGS_REPO, svhip.FLAGS_HH]))
    cw = 7 if D > 128 else int(rng.choice([7, 15]))
    os.environ["SV_SWEEP_NCU"] = str(int(rng.integers(2, 13))); os.environ["SV_SWEEP_CW"] = str(cw)
    ph = svhip.default_params(D, paths, flags=flags, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, uniqueness_ratio=uniq)
    ph.min_disparity = minD
    po = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    pairs = [synth.stereo_pair(9000 + 16 * case + i, W, H, D) for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(); rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.full((nb, H, W), 12345, dtype=torch.int16, device="cuda")
    m = svhip.Matcher(W, H, ph); m.set_batch(nb); m.set_debug(3)
    m.compute_gray_batch_dev(nb, lt, rt, out); m.synchronize()
    got = out.cpu().numpy()
    ok = all(np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)) for i in range(nb))
    agg = m.last_aggregation()
    print("case %2d W=%d H=%d D=%d bs=%d minD=%d nb=%d uniq=%d paths=%d cw=%d agg=%s %s" % (case, W, H, D, bs, minD, nb, uniq, paths, cw, agg, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    m.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
