"""Both sweep workgroup shapes against the production choice at mid-size batches, C2."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(4)]
for nb in (28, 32, 40, 48, 56, 64, 80, 96):
    res = []
    for cw in ("0", "7", "15"):
        os.environ["SV_SWEEP_CW"] = cw
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.set_batch(nb)
        lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
        rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
        out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
        for i in range(3): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(30): m.compute_gray_batch_dev(nb, lt, rt, out)
        torch.cuda.synchronize()
        res.append(nb / ((time.perf_counter() - t) / 30))
        m.close()
    print("batch %3d: auto %7.1f  8-wave %7.1f  16-wave %7.1f frames/s" % (nb, *res), flush=True)
