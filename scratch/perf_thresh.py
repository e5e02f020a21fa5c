"""Per-path kernels against forced sweeps (both shapes) at small batches, C2."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(4)]
for nb in (6, 8, 12, 16, 20, 24):
    for mode, cw in ((0, "0"), (3, "7"), (3, "15")):
        os.environ["SV_SWEEP_CW"] = cw
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.set_batch(nb); m.set_debug(mode)
        lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
        rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
        out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
        for i in range(3): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(40): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 40
        print("batch %3d mode %d cw %2s: %7.3f ms  %7.1f frames/s  agg %s" % (nb, mode, cw, dt * 1e3, nb / dt, m.last_aggregation()), flush=True)
        m.close()
