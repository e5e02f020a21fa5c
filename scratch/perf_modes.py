"""Which aggregation variant wins at which batch (C2): mode 0 = heuristics, 5 = all paths in one launch, 3 = fused sweeps, 2 = per-path + fused WTA."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(4)]
m = svhip.Matcher(W, H, svhip.default_params(D, 8))
m.set_stream(torch.cuda.current_stream().cuda_stream)
for nb in (4, 8, 12, 16, 24, 32, 40, 48):
    m.set_batch(nb)
    lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
    res = []
    for mode in (2, 5, 3):
        m.set_debug(mode)
        for i in range(3): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        reps = max(3, min(100, 1024 // nb))
        t = time.perf_counter()
        for i in range(reps): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        res.append(nb / ((time.perf_counter() - t) / reps))
    print("batch %3d: per-path %7.1f  all-paths-one-launch %7.1f  sweeps %7.1f frames/s" % (nb, *res), flush=True)
m.close()
