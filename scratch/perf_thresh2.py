"""Where the fused sweeps (8-wave workgroups) pass the all-paths kernels at small batches (C2): production choice against forced sweeps."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(4)]
os.environ["SV_SWEEP_CW"] = "7"
for nb in (6, 8, 10, 12, 14, 16, 18, 20):
    res = []
    lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
    for mode in (0, 3, 5):
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.set_batch(nb); m.set_debug(mode)
        for i in range(3): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        reps = 40
        t = time.perf_counter()
        for i in range(reps): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        res.append((nb / ((time.perf_counter() - t) / reps), m.last_aggregation()))
        m.close()
    print("batch %3d: production %7.1f %s   forced sweeps (8-wave) %7.1f %s   all-paths %7.1f %s" % (nb, *res[0], *res[1], *res[2]), flush=True)
