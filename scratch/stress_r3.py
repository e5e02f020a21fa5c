"""One-off stress of round 3's kernel changes (cost kernel: LDS stage, counted memory queue, disparity parts; sweeps: binary16
three-input minimum): random geometry, D, block, cap, P1 / P2 (up to the sweeps' value gate), minDisparity, batches on both sides
of 8 frames (32- and 64-column cost tiles), noise images among the synthetic ones, every frame against the oracle.
   python scratch/stress_r3.py [cases] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "stereo-vision_amd")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
import svref, svhip
from svhip import synth
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
bad = done = 0
while done < ncase:
    D = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256]))
    bs = int(rng.choice([1, 3, 5, 7, 9, 11]))
    cap = int(rng.choice([15, 31, 63, 100]))
    P1 = int(rng.choice([8 * bs * bs, 1, 300]))
    csum = (2 * cap + 63) * bs * bs
    room = 0x7BFF - P1 - csum                     # what P2 may take before the sweeps hand over
    if room < P1 + 1:
        continue
    P2 = int(rng.choice([min(32 * bs * bs, room), room, max(P1 + 1, room // 2)]))
    P2 = max(P2, P1, 1)
    minD = int(rng.choice([0, 0, 0, 7, -5, 40]))
    W = D + abs(minD) + bs + int(rng.integers(2, 120))
    H = bs + int(rng.integers(1, 90))
    nb = int(rng.choice([2, 5, 8, 9, 13]))
    uniq = int(rng.choice([0, 5, 15, 50]))
    paths = int(rng.choice([5, 8]))
    flags = int(rng.choice([svhip.FLAGS_REPO, svhip.FLAGS_HH]))
    cw = 7 if D > 128 else int(rng.choice([7, 15]))
    noise = bool(rng.integers(0, 3) == 0)
    os.environ["SV_SWEEP_NCU"] = str(int(rng.integers(2, 13))); os.environ["SV_SWEEP_CW"] = str(cw)
    ph = svhip.default_params(D, paths, flags=flags, block_size=bs, P1=P1, P2=P2, uniqueness_ratio=uniq, pre_filter_cap=cap)
    ph.min_disparity = minD
    po = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    if noise:
        pairs = [(rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)) for i in range(nb)]
    else:
        pairs = [synth.stereo_pair(9000 + 16 * done + i, W, H, D)[:2] for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(); rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.full((nb, H, W), 12345, dtype=torch.int16, device="cuda")
    try:
        m = svhip.Matcher(W, H, ph)
    except svhip.SvError as e:
        print("skipped (%s)" % str(e)[:80]); continue
    m.set_batch(nb); m.set_debug(3)
    m.compute_gray_batch_dev(nb, lt, rt, out); m.synchronize()
    got = out.cpu().numpy()
    ok = all(np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)) for i in range(nb))
    agg = m.last_aggregation()
    print("case %2d W=%d H=%d D=%d bs=%d cap=%d P1=%d P2=%d minD=%d nb=%d uniq=%d paths=%d cw=%d noise=%d agg=%s %s" % (
        done, W, H, D, bs, cap, P1, P2, minD, nb, uniq, paths, cw, noise, agg, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok; done += 1
    m.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
